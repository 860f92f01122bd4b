#!/usr/bin/env python
"""Benchmark of the batched ALGAMES Newton / augmented-Lagrangian hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks (it re-executes itself
under torch.distributed.run on 127.0.0.1) and FAILS -- it never degrades to fewer GPUs -- when the node has fewer
than N devices.

A "step" is one pass of the hot path over one batch: a full batched `newton_solve!` (init_traj! + RK3 rollout,
AL outer loop, Newton inner loop, structured KKT solve, line search, dual/penalty updates) with all inputs
already resident in HBM.  Default workload = BASELINE config C2: 3-player DoubleIntegrator, N = 40, 4096
synthetic scenarios per GPU; C4 is the same problem with 8192 scenarios per GPU (65 536 over
8 GPUs); C3 = 4-player Unicycle N = 50 (1024 per GPU); C5 = 3-player Unicycle N = 30 (with --mpc-steps T one
step is a T-step receding-horizon loop per scenario).  Every step re-initialises the iterate from the counter
RNG, so all K steps do identical work.  metric = game-Newton-iterations per second (inner iterations that
performed a linear solve, summed over games and ranks) -- BASELINE.json's "Newton iters/sec (batch)";
games-to-convergence/s is reported next to it.  Weak scaling: the scenario batch is sharded by contiguous
global scenario ids, one rank per GPU, no data-path collective (games are independent); the only collectives
are the timing barrier / max and the final count reduction.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
F64_PEAK = 78.6e12         # FLOP/s, dense f64 vector = f64 MFMA peak (same guide)

# configuration -> (scenario family of algames.jl_amd/scenarios.py, default scenarios per GPU)
CONFIGS = {"C2": ("C2", 4096), "C3": ("C3", 1024), "C4": ("C2", 8192), "C5": ("C5", 64),
           # not BASELINE configurations: the QuadrotorGame scenario of scenarios.quadrotor_crossing (dense Newton direction)
           "Q2": ("Q", 4096), "Q4": ("Q", 1024)}
CONFIG_KW = {"Q2": {"p": 2}, "Q4": {"p": 4}}
WORKLOADS = {"C2": "C2: 3-player DoubleIntegrator (d=2), N=40, collision cost + collision avoidance, 4096 scenarios/GPU",
             "C3": "C3: 4-player Unicycle, N=50, collision avoidance + control bounds, 1024 scenarios/GPU",
             "C4": "C4: 3-player DoubleIntegrator (d=2), N=40, 65536 scenarios sharded over 8 GPUs (8192/GPU)",
             "C5": "C5: 3-player Unicycle, N=30, collision avoidance + control bounds, receding-horizon seeds",
             "Q2": "Q2 (not a BASELINE configuration): 2-player Quadrotor, N=20, planar collision avoidance + rotor bounds",
             "Q4": "Q4 (not a BASELINE configuration): 4-player Quadrotor, N=20, planar collision avoidance + rotor bounds"}


def survey_balg(N, n, m, p):
    """bytes per game-Newton-iteration of the dense block-tridiagonal LU the survey priced,
    8*[2(S+n) + 2(N-1)(b^2 + b*p*n)].  Context only: the structured elimination never performs that spill."""
    b = n + m + p * n
    S = n * p * (N - 1) + m * (N - 1) + n * (N - 1)
    return 8 * (2 * (S + n) + 2 * (N - 1) * (b * b + b * p * n))


def structured_bytes(family, N, n, m, p, ls_trials_per_iter=1.0, gate=True, waves_per_game=1):
    """Algorithmic HBM bytes per game-Newton-iteration of THIS implementation (DESIGN.md "Roofline accounting"): what the passes of
    one iteration have to move when every array crosses the memory system once per pass that needs it -- the floor of this
    algorithm; re-reads, partial lines and the per-outer-iteration record pass are not included.
      trial pass   double integrator (C2 / C4): the FUSED pass of round 4 -- read z and dz, write the trial, read the multipliers, write the
                   step records; other models: update_traj! (read z, dz, write the trial) + assemble pass (read trial and proximal
                   reference, multipliers, write the records incl. the pair-gradient table)
      backward     read the record slice [.. rd], write the gains
      forward      read gains and record slice, write dx, du (double integrator: also reads [.. rx] and parks w = rx + Q dx for the costate)
      costate      read [.. rx] and dx (double integrator: w only), write dlambda
      gate         the opt-u rows of the refinement gate: du, two dlambda entries per control, R^, ru
    The accepted trial becomes pdtraj by exchanging buffers (no traffic; a solve that ends on the exchanged side copies pdtraj home once: 2 x 17 KB per
    C2 solve, in traffic_over_model)."""
    S = n * p * (N - 1) + m * (N - 1) + n * (N - 1)
    it, K = S + n, N - 1
    nc = {"C2": 0, "C3": 4 * p, "C5": 4 * p, "Q": 204 * p}[family]      # RK2 Jacobian coefficients per step (quadrotor: dense blocks)
    npair = p * (p - 1)
    len_costate = nc + 3 * npair + 3 * p + p * n                         # [coef | Hh | Hd | rx]
    len_sweep = len_costate + 2 * m + n                                  # + [R^ | ru | rd]
    len_rec = len_sweep + 2 * p * p                                      # + pair-gradient table
    con = K * npair + (2 * m * K if family in ("C3", "C5", "Q") else 0)  # constraint rows touched (lam, mu read)
    gains = K * m * (n + 1)
    # the fused trial pass (AsmLds::FUSED): double integrator and unicycle, base constraint set, ONE wavefront per game (C2 / C4, and the
    # unicycle shapes when a batch is large enough for one-wavefront kernels); teams keep the two passes.  (The pair-gradient tables the
    # fused pass still writes in its phase A and re-reads per chunk -- 2 K p^2 doubles -- are an artefact of the implementation, not of the
    # algorithm: they are left out of this floor and show up in traffic_over_model.)
    fused = family in ("C2", "C3", "C5") and waves_per_game == 1
    if fused:
        trial = ls_trials_per_iter * (it + S + S + 2 * con + K * len_sweep)
    else:
        trial = ls_trials_per_iter * ((it + S + it) + (2 * it + 2 * con + K * len_rec))   # axpy + assemble pass
    if family == "C2":                                                   # double integrator: the forward sweep parks w = rx + Q dx (FWDW)
        forward = gains + K * (len_costate + n) + K * (n + m) + K * p * n
        costate = 2 * K * p * n
    else:
        forward = gains + K * (nc + n) + K * (n + m)
        costate = K * len_costate + K * n + K * p * n
    backward = K * len_sweep + gains
    gate_b = K * 5 * m if gate else 0
    return 8 * (trial + backward + forward + costate + gate_b)


def outer_pass_bytes(family, N, n, m, p, fused_dual=False):
    """Algorithmic bytes of the two passes newton_solve! makes once per OUTER iteration, which structured_bytes() (per Newton iteration)
    leaves out: (record pass, dual / penalty update pass).
      record pass   the first inner iteration after a dual update cannot reuse the accepted trial's records (the multipliers moved):
                    read the iterate and the multipliers / penalties, write the step records and the pair-gradient tables
      dual update   dual_update! / penalty_update! (constraints_methods.jl:295-445): read the iterate (positions and controls touch every
                    line of it), lambda and mu, write the constraint values, lambda and mu
    Per-sweep counters (tests/probes/phase_bytes.sh with the ALG_DIR_STOP builds) put the three sweeps and the gate
    within 7 % of structured_bytes(); what the fabric moves beyond that model is these passes and re-reads inside the trial pass."""
    S = n * p * (N - 1) + m * (N - 1) + n * (N - 1)
    it, K = S + n, N - 1
    nc = {"C2": 0, "C3": 4 * p, "C5": 4 * p, "Q": 204 * p}[family]
    npair = p * (p - 1)
    len_rec = nc + 3 * npair + 3 * p + p * n + 2 * m + n + 2 * p * p
    con = K * npair + (2 * m * K if family in ("C3", "C5", "Q") else 0)
    # round 6, fused kernels (one wavefront per game, double integrator / unicycle): the dual / penalty update rides on the record pass that
    # follows it -- its reads ARE the record pass's, what it adds is the three stores (values, lambda, mu)
    return 8 * (it + 2 * con + K * len_rec), (8 * 3 * con if fused_dual else 8 * (it + 2 * con + 3 * con))


# --------------------------------------------------------------------------------------------------
# Sharding + reduction: the N > 1 path lives in the package (algames.jl_amd/sharding.py); bench.py only maps its
# configuration names.  tests/test_sharding_gloo.py runs the package functions with world_size 2 on gloo.
# --------------------------------------------------------------------------------------------------
def make_shard(alg, config, games_per_rank, rank, world, backend=None, device=0, **kw):
    """The rank's shard of a bench configuration: a thin caller of the package's `sharding.make_shard` (contiguous global
    scenario ids)."""
    return alg.sharding.make_shard(CONFIGS[config][0], games_per_rank, rank, world, backend=backend, device=device,
                                   **{**CONFIG_KW.get(config, {}), **kw})


def reduce_counters(alg, counts, elapsed, world, device, use_dist=None):
    """Sum of the per-rank counters, max of the per-rank wall time: the package's `sharding.reduce_counters` (the only collectives).
    `use_dist` = whether this run created a process group (a world of one under --force-dist / torchrun still reduces through it)."""
    return alg.sharding.reduce_counters(counts, elapsed, world, device, use_dist=use_dist)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# Test hook (tests/test_gpu_bench_ranks.py): ALGAMES_BENCH_SHARED_DEVICE=1 lets the ranks of an N > 1 run share device 0 with gloo
# carrying the barrier and the counter reduction, so that the multi-process path of this script (shards, barrier, max over ranks,
# reduction, the one JSON line) runs on a one-GPU box.  The line it prints is labelled and is not a multi-GPU measurement.
SHARED_DEVICE = os.environ.get("ALGAMES_BENCH_SHARED_DEVICE") == "1"


def launch_ranks(ngpu):
    """Re-executes this script as `ngpu` ranks (one per GPU) under torch.distributed.run.  Fails when the node cannot
    supply `ngpu` devices."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < ngpu and not (SHARED_DEVICE and have >= 1):
        raise SystemExit(f"bench.py: --gpus {ngpu} requested but this node exposes {have} GPU(s); refusing to run a "
                         f"smaller job under the same label")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpu}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def committed_profile(config, games_per_gpu, mpc_steps):
    """Newest committed PMC summary (profiles/rNN_*_pmc*.json) of this workload, or None.  These numbers come from
    separate rocprofv3 --pmc passes of this same command and are labelled as such in the output."""
    import re
    best = None
    for pf in glob.glob(os.path.join(ROOT, "profiles", "*_pmc*.json")):
        try:
            pj = json.load(open(pf))
        except Exception:
            continue
        if pj.get("config") != config or pj.get("games_per_gpu") != games_per_gpu or int(pj.get("mpc_steps", 0)) != mpc_steps:
            continue
        key = [int(t) for t in re.findall(r"\d+", os.path.basename(pf))]
        if best is None or key > best[0]:
            best = (key, pf, pj)
    return best


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
               "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE"))


def inrun_pmc(argv_tail, kernel, iters_per_launch, own_bytes, timeout_s=150):
    """Hardware counters of THIS run's workload, measured now: after the timed region rank 0 starts one short child of this very
    script per counter group under `rocprofv3 --pmc ... --kernel-trace` (counter passes on their own, as the MI355X guide
    prescribes; FETCH_SIZE x 2 = the gfx950 correction calibrated with tests/probes/pmc_calib.hip, KB units) and averages the solver
    kernel's launches.  Returns None when rocprofv3 is missing or a pass fails (the caller then falls back to the committed
    profile and says so)."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    cnt, dur_ns, ncalls = {}, [], 0
    work = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    try:
        for i, group in enumerate(PMC_PASSES):
            out = os.path.join(work, f"p{i}")
            cmd = [exe, "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"] + argv_tail
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            acc, disp = {}, {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel not in row["Kernel_Name"]:
                        continue
                    acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    disp.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
            if not acc:
                return None
            for k, v in acc.items():
                cnt[k] = v / len(disp[k])
            for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel in row["Kernel_Name"]:
                        dur_ns.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F64")
    if any(k not in cnt for k in need) or not dur_ns:
        return None
    t = sum(dur_ns) / len(dur_ns) * 1e-9
    hbm = 1024.0 * (2.0 * cnt["FETCH_SIZE"] + cnt["WRITE_SIZE"])
    simd_quads = cnt["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0      # 8 XCDs report their cycles; 1024 SIMDs; quad-cycle = 4 clocks
    res = {"traffic": hbm, "traffic_over_model": hbm / (own_bytes * iters_per_launch),
           "valu_issue_frac": cnt["SQ_ACTIVE_INST_VALU"] / simd_quads,
           "mfma_frac": cnt["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / t / F64_PEAK,
           "kernel_ms_profiled": 1e3 * t,
           "pmc_source": "in-run: %d rocprofv3 --pmc passes of a 2-step child of this command, started after the timed region" % len(PMC_PASSES)}
    if "SQ_WAVE_CYCLES" in cnt and cnt["SQ_WAVE_CYCLES"] > 0:
        res["wave_issue_frac"] = cnt.get("SQ_ACTIVE_INST_ANY", 0.0) / cnt["SQ_WAVE_CYCLES"]
        res["wave_wait_frac"] = cnt.get("SQ_WAIT_ANY", 0.0) / cnt["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in cnt:
        res["valu_insts_per_game_iter"] = cnt["SQ_INSTS_VALU"] / iters_per_launch
        res["salu_insts_per_game_iter"] = cnt.get("SQ_INSTS_SALU", 0.0) / iters_per_launch
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8,
                    help="untimed launches first (default 8: the shader clock needs ~5 launches of this kernel to settle, profiles/r05_clock_probe.txt)")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--games-per-gpu", type=int, default=0, help="scenarios per GPU (default: the config's BASELINE batch)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU gets the config's per-GPU batch (C4 = 8192 x 8 = north_star's 65 536); "
                         "strong: the config's ONE-GPU batch (or --games-total) is split over the GPUs (C2: 4096 -> 4096 / N per GPU)")
    ap.add_argument("--games-total", type=int, default=0, help="strong scaling: total scenarios of the job (default: the config's one-GPU batch)")
    ap.add_argument("--mpc-steps", type=int, default=0,
                    help="C5 receding-horizon mode: one bench step = this many warm-started MPC solves per game")
    ap.add_argument("--waves-per-game", type=int, default=0, choices=[0, 1, 2, 4],
                    help="kernel shape of the fused solver: wavefronts per game (0 = the library's automatic choice)")
    ap.add_argument("--refine-steps", type=int, default=-1, help="alg_set_refinement max_steps (default: the library's; 0 = gate and refinement off)")
    ap.add_argument("--refine-tol", type=float, default=-1.0, help="alg_set_refinement tol (default: the library's)")
    ap.add_argument("--perturb", type=float, default=0.0,
                    help="heterogeneous batch (NOT the BASELINE workload; labelled in config.workload): every start position coordinate of every "
                         "scenario is moved by U(-PERTURB, PERTURB), keyed by the global scenario id -- the games then need different numbers "
                         "of Newton iterations (C2 at 0.3: 11 ... 109) and one launch lasts as long as its slowest game")
    ap.add_argument("--handoff", type=int, default=-1,
                    help="straggler hand-off (alg_set_handoff): games that exceed this many Newton iterations in the one-wavefront kernel park "
                         "their state and a second launch finishes them with the team kernel (-1: the library's default, 0: off)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed with the nccl (= RCCL) backend even for one rank, so that communicator creation, the "
                         "barrier and the two all-reduces of the counter reduction run on the device (tests/test_gpu_bench_ranks.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (the children of a run use this)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.mpc_steps and args.config != "C5":
        raise SystemExit("bench.py: --mpc-steps is the C5 receding-horizon mode")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        launch_ranks(args.gpus)                                     # does not return
    world = int(world_env) if world_env is not None else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torchrun: bench.py starts its own ranks)")

    import numpy as np
    import torch
    import torch.distributed as dist
    import algames_jl_amd as alg

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if SHARED_DEVICE:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} (node exposes {torch.cuda.device_count()})")
    torch.cuda.set_device(local_rank)
    # one rank: no process group unless asked for (--force-dist) or launched by torchrun with WORLD_SIZE=1 -- then the same RCCL path
    # as N > 1 runs with a world of one
    use_dist = world > 1 or args.force_dist or (world_env is not None and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world_env is None:                                        # --force-dist outside torchrun: a rendezvous of our own
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        if SHARED_DEVICE and world > 1:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI

    family, default_games = CONFIGS[args.config]
    cfg_kw = CONFIG_KW.get(args.config, {})
    if args.scaling == "strong":
        total = args.games_total or default_games
        if total % world:
            raise SystemExit(f"bench.py: --scaling strong needs a total ({total}) divisible by --gpus ({world})")
        G = total // world                                           # equal shards: every rank picks the same kernel shape
    else:
        G = args.games_per_gpu or default_games
    if os.environ.get("ALGAMES_BENCH_FAIL_RANK") == str(rank):       # test hook (tests/test_gpu_bench_ranks.py): a rank that dies must fail the job
        raise SystemExit(f"bench.py: rank {rank} fails on request (ALGAMES_BENCH_FAIL_RANK)")
    prob, ids = make_shard(alg, args.config, G, rank, world, device=local_rank)
    b = prob.batch
    if args.perturb:
        # heterogeneous batch: start positions spread by +-perturb, drawn from the counter generator keyed by the GLOBAL scenario id (stream
        # apart from the scenario's own draws), so the shard layout does not change them
        npos = 2 * b.p
        x0p = prob.x0.copy()
        x0p[:, :npos] += alg.scenarios._uniform(0x9E27 + 7919, ids, npos, -args.perturb, args.perturb)
        prob.x0 = x0p
        b.set_x0(x0p)
    stream = torch.cuda.Stream()                                    # a real (non-NULL) HIP stream owned by torch
    torch.cuda.set_stream(stream)
    b.set_stream(stream.cuda_stream)                                # the library launches on this stream
    b.set_waves_per_game(args.waves_per_game)
    waves_per_game = b.get_waves_per_game()
    if args.refine_steps >= 0 or args.refine_tol >= 0:
        b.set_refinement(args.refine_steps if args.refine_steps >= 0 else None, args.refine_tol if args.refine_tol >= 0 else None)
    if args.handoff > 0:
        b.set_handoff(args.handoff)
    refine_steps, refine_tol, refine_mu = b.get_refinement()
    prob._sync_options()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mpc_steps:
        x0_start = prob.x0.copy()

    def step():
        if args.mpc_steps:
            b.set_x0(x0_start)                                      # restart the same closed-loop rollout
            alg.mpc_solve(prob, args.mpc_steps)
        else:
            b.newton_solve_async(init=True, game_id0=int(ids[0]))

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(args.steps):
        step()
        ev[i + 1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]   # HIP events on the launch stream

    st = b.get_stats()
    if args.mpc_steps:
        it_, cv_ = b.mpc_totals()
        iters_rank, conv_rank = int(it_.sum()), int(cv_.sum())
        per_game = it_
    else:
        iters_rank = int(st["newton_iters"].sum())
        conv_rank = int(st["converged"].sum())
        per_game = st["newton_iters"]
    # one launch lasts as long as its slowest game: mean / max of the per-game iteration counts (1.0 = homogeneous batch)
    balance = float(per_game.mean() / max(1, per_game.max()))
    bad_rank = int((st["status"] != 0).sum())
    refinements_rank = int(st["refinements"].sum())          # correction solves of the last launch (mpc mode: of every game's last solve)
    shard_ranges = alg.sharding.gather_shard_ranges(int(ids[0]), int(ids[-1]) + 1, world, "cuda", use_dist=use_dist)
    (iters_all, conv_all, bad_all), elapsed = reduce_counters(alg, [iters_rank, conv_rank, bad_rank], elapsed, world, "cuda", use_dist=use_dist)

    if rank == 0:
        K = args.steps
        value = iters_all * K / elapsed
        kern_s = float(np.mean(kernel_ms)) * 1e-3
        p, n, m, N = b.p, b.n, b.m, b.N
        own = structured_bytes(family, N, n, m, p, gate=refine_steps > 0, waves_per_game=waves_per_game)
        kernel = "k_mpc_loop" if args.mpc_steps else "k_newton_solve"
        achieved = own * iters_rank / kern_s                         # B/s of this rank's launch
        roof = {
            # the contract's bound is the HBM roofline: achieved = algorithmic bytes per launch / launch duration, where
            # the algorithmic bytes are THIS kernel's (DESIGN.md "Roofline accounting": structured_bytes(), per
            # game-Newton-iteration) x the game-iterations one launch performs; frac <= 1 by construction.
            "bound": "hbm", "kernel": kernel,
            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
            "bytes_per_game_iter": own, "game_iters_per_launch": iters_rank, "kernel_ms_avg": 1e3 * kern_s,
            "traffic": None,
            "limiter": "instruction issue + dependency latency (not HBM): see valu_issue_frac / mfma_frac",
            # context only (SURVEY.md 8(d) priced a dense block-LU with factor spill; this kernel never performs it)
            "survey_dense_lu_bytes_per_game_iter": survey_balg(N, n, m, p),
        }
        if not args.mpc_steps:
            # passes made once per outer iteration, amortised over the Newton iterations with this launch's own counts: every outer
            # iteration starts with a record pass, every one but the last ends with the dual / penalty update
            rec_b, dual_b = outer_pass_bytes(family, N, n, m, p, fused_dual=(family in ("C2", "C3", "C5") and waves_per_game == 1))
            n_out = int(st["outer_iters"].sum()); n_dual = int(np.maximum(st["outer_iters"] - 1, 0).sum())
            roof["outer_passes_per_launch"] = {"record": n_out, "dual_update": n_dual}
            roof["bytes_per_game_iter_outer_passes"] = (n_out * rec_b + n_dual * dual_b) / max(1, iters_rank)
        pmc = None
        if world == 1 and not args.no_pmc:
            tail = ["--config", args.config, "--games-per-gpu", str(G), "--waves-per-game", str(args.waves_per_game),     # (a 1-GPU child: weak)
                    "--refine-steps", str(refine_steps), "--refine-tol", repr(refine_tol)]
            if args.mpc_steps:
                tail += ["--mpc-steps", str(args.mpc_steps)]
            if args.perturb:
                tail += ["--perturb", repr(args.perturb)]
            if args.handoff >= 0:
                tail += ["--handoff", str(args.handoff)]
            pmc = inrun_pmc(tail, kernel, iters_rank, own)
        if pmc is not None:
            roof.update(pmc)
            if "bytes_per_game_iter_outer_passes" in roof:
                roof["traffic_over_model_incl_outer_passes"] = roof["traffic"] / ((own + roof["bytes_per_game_iter_outer_passes"]) * iters_rank)
        elif not args.no_pmc:
            prof = committed_profile(args.config, G, args.mpc_steps)
            if prof is not None:
                _, pf, pj = prof
                roof["traffic"] = pj.get("hbm_bytes_per_launch")
                for k in ("valu_issue_frac", "mfma_frac", "wave_issue_frac", "traffic_over_model"):
                    if k in pj:
                        roof[k] = pj[k]
                roof["pmc_source"] = "committed: " + os.path.relpath(pf, ROOT) + " (the in-run rocprofv3 passes were not available; not re-measured in this run)"
        out = {
            "metric": "newton_iters_per_sec", "value": value, "unit": "game-Newton-iterations/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.config] + f" [{args.scaling} scaling: {G} scenarios/GPU x {world} GPU = {G * world} scenarios]"
                                   + (f" [HETEROGENEOUS variant, not the BASELINE workload: start positions perturbed by +-{args.perturb}]" if args.perturb else ""),
                       "perturb": args.perturb,
                       "name": args.config,
                       "games_per_gpu": G, "games_total": G * world, "newton_iters_per_solve_total": iters_all,
                       "shard_ranges": shard_ranges,          # [lo, hi) global scenario ids of every rank, gathered from the ranks
                       "mpc_steps": args.mpc_steps,
                       "parallelism": f"scenario-sharded x{world}" + (" (TEST HOOK: the ranks share device 0, gloo; not a multi-GPU measurement)" if SHARED_DEVICE and world > 1 else ""), "wavefronts_per_game": waves_per_game,
                       "collectives": (("gloo" if SHARED_DEVICE and world > 1 else "nccl (RCCL)") + f", world {world}: barrier + all_gather of the shard ranges + 2 all_reduce of the counters") if use_dist else "none (one rank, no process group)",
                       "iters_per_game_mean_over_max_rank0": balance,
                       "handoff": dict(zip(("budget_iters", "games_handed_over_rank0"), b.get_handoff())),
                       "direction_refinement": {"max_steps": refine_steps, "tol": refine_tol, "mu_tight": refine_mu, "correction_solves_rank0": refinements_rank},
                       "solver": ("fused per-game receding-horizon loop kernel (alg_mpc_solve)" if args.mpc_steps
                                  else "fused per-game newton_solve! kernel")},
            "games_to_convergence_per_sec": conv_all * K / elapsed,
            "games_converged": conv_all, "games_failed": bad_all,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(alg, family, G, args.mpc_steps, cfg_kw)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def usable_cpus():
    """CPUs this process may actually run on: min(logical CPUs, affinity mask, cgroup CPU quota).  The GPU boxes expose 256
    logical CPUs but run the container under a cgroup quota (cpu.max = 16 CPUs); more OpenMP threads than that only add
    throttling."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota)))
    return n, (os.cpu_count() or 1), quota


def cpu_baseline(alg, family, G, mpc_steps=0, cfg_kw=None):
    """The oracle (literal CPU restatement of the reference algorithm: global KKT assembly + general partial-pivot
    LU per game, OpenMP over games) timed on a bounded sample of the same workload: all host cores, and one core
    (the closest analogue of the single-threaded Julia solver)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as orc
    cores, logical, quota = usable_cpus()
    prev = orc.set_threads(cores)

    def run(nsample):
        prob = alg.scenarios.make_problem(family, np.arange(nsample), backend=orc.lib(), **(cfg_kw or {}))
        t0 = time.perf_counter()
        if mpc_steps:
            steps = min(mpc_steps, 8)
            it, cv, _ = alg.mpc_solve(prob, steps)
            dt = time.perf_counter() - t0
            return int(it.sum()), int(cv.sum()), dt, f"{steps} receding-horizon steps each"
        alg.newton_solve(prob)
        dt = time.perf_counter() - t0
        s = prob.stats.summary
        return int(s["newton_iters"].sum()), int(s["converged"].sum()), dt, "one newton_solve! each"

    nsample = int(max(8, 16 * cores))
    it, cv, dt, what = run(nsample)
    if dt < 4.0:                                                    # a bigger sample first (bounded: host memory)
        nsample = int(min(64 * cores, max(nsample, nsample * 10.0 / max(dt, 1e-3))))
        it, cv, dt, what = run(nsample)
    reps = 1
    while dt < 6.0 and reps < 64:                                   # ... then repeat it until >= 6 s of OpenMP work are timed
        it2, cv2, dt2, _ = run(nsample)
        it += it2; cv += cv2; dt += dt2; reps += 1
    if reps > 1:
        what += f", sample solved {reps} times"
    orc.set_threads(1)
    n1 = 24 if not mpc_steps else 8
    it1, cv1, dt1, _ = run(n1)
    orc.set_threads(prev)
    return {"value": float(it / dt), "unit": "game-Newton-iterations/s", "cores": cores, "threads_used": cores,
            "host_logical_cpus": logical, "cgroup_cpu_quota": quota,
            "kind": "port", "sample": f"first {nsample} scenarios of the same workload, {what}, {dt:.1f} s wall, OpenMP over games",
            "games_to_convergence_per_sec": float(cv / dt),
            "single_core_value": float(it1 / dt1),
            "single_core_sample": f"first {n1} scenarios, 1 thread, {dt1:.1f} s wall",
            "scaling_all_cores_over_one": float((it / dt) / (it1 / dt1))}


if __name__ == "__main__":
    main()
