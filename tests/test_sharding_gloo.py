"""N > 1 path on CPU: world_size-2 `gloo` run of the scenario sharding.
Games are independent, so the multi-GPU path is: contiguous shards of global scenario ids, one rank per
device, no data-path collective, one final reduction of the counters.  Here each rank solves its shard with
the CPU oracle standing in for the device (the oracle is test infrastructure) and the reduced counters and
gathered trajectories must equal the single-process solve of the whole batch -- i.e. inputs depend only on
global scenario ids and the reduction is what bench.py does."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.distributed as dist
import algames_jl_amd as alg, oracle as orc
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
TOTAL = 10
import bench                                   # the N > 1 path under test is the package's sharding module; bench.py is a thin caller of it
prob, ids = alg.sharding.make_shard("C4", TOTAL // world, rank, world, backend=orc.lib(), N=10)
p2, ids2 = bench.make_shard(alg, "C4", TOTAL // world, rank, world, backend=orc.lib(), N=10)
assert np.array_equal(ids, ids2) and np.array_equal(prob.x0, p2.x0)
lo, hi = int(ids[0]), int(ids[-1]) + 1
assert (lo, hi) == alg.scenarios.shard_range(TOTAL, rank, world)
alg.newton_solve(prob)
s = prob.stats.summary
assert alg.sharding.local_counters(prob)[:2] == [int(s["newton_iters"].sum()), int(s["converged"].sum())]
cnt, tmax = alg.sharding.reduce_counters([int(s["newton_iters"].sum()), int(s["converged"].sum()), hi - lo], 1.0 + rank, world, "cpu")
assert tmax == float(world)                    # max over ranks of the per-rank time
rng = alg.sharding.gather_shard_ranges(lo, hi, world, "cpu")
assert rng == [list(alg.scenarios.shard_range(TOTAL, r, world)) for r in range(world)], rng      # the shards tile the job in rank order
# a caller's own process group must not be touched by a one-rank reduction (ADVICE r5): world = 1 means no collective, whatever is initialised
one, t1 = alg.sharding.reduce_counters([rank + 1], 2.0, 1, "cpu")
assert one == [rank + 1] and t1 == 2.0 and alg.sharding.gather_shard_ranges(lo, hi, 1, "cpu") == [[lo, hi]]
cnt = torch.tensor(cnt)
z = torch.from_numpy(prob.batch.get_traj())
per = (TOTAL + world - 1) // world
pad = torch.zeros(per, z.shape[1], dtype=torch.float64); pad[: z.shape[0]] = z
out = [torch.zeros_like(pad) for _ in range(world)]
dist.all_gather(out, pad)
if rank == 0:
    full = alg.scenarios.make_problem("C2", np.arange(TOTAL), N=10, backend=orc.lib())
    alg.newton_solve(full)
    fs = full.stats.summary
    zall = torch.cat(out)[:TOTAL].numpy()
    assert cnt.tolist() == [int(fs["newton_iters"].sum()), int(fs["converged"].sum()), TOTAL], (cnt, fs["newton_iters"].sum())
    assert np.array_equal(zall, full.batch.get_traj())
    print("SHARDING_OK", cnt.tolist())
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    import algames_jl_amd as alg
    for total, world in ((4096, 1), (65536, 8), (10, 3), (7, 8)):
        cuts = [alg.scenarios.shard_range(total, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert all(lo <= hi for lo, hi in cuts)


def test_scenarios_depend_only_on_global_id():
    import algames_jl_amd as alg
    a = alg.scenarios.c2_double_integrator(np.arange(0, 64))[3]
    b = alg.scenarios.c2_double_integrator(np.arange(32, 64))[3]
    assert np.array_equal(a[32:], b)
    c = alg.scenarios.c3_unicycle(np.arange(5, 9))[3]
    d = alg.scenarios.c3_unicycle(np.arange(0, 9))[3]
    assert np.array_equal(c, d[5:])


def test_bench_refuses_to_degrade_the_gpu_count():
    """`python bench.py --gpus 2` with no torchrun environment starts its own ranks -- and must fail (not run a smaller
    job) when the node has fewer devices; under torchrun a WORLD_SIZE that disagrees with --gpus is an error too."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    assert out.returncode != 0 and "refusing" in out.stderr, out.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr, out.stderr[-2000:]


def test_world_size_2_gloo_sharded_solve(orc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    code = "ROOT = %r\n" % ROOT + WORKER
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", "--no-python", sys.executable, "-c", code]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDING_OK" in out.stdout


def test_sharded_problem_in_one_process_equals_the_whole_batch(orc):
    """`ShardedGameProblem` (one process, several handles): the same contiguous-shard rule, every launch before the first
    synchronisation.  With the oracle standing in for the device the sharded solve must equal the single-handle solve bit for
    bit (inputs depend only on global scenario ids) -- also with an uneven split and more devices than games."""
    import algames_jl_amd as alg
    full = alg.scenarios.make_problem("C2", np.arange(3, 13), N=10, backend=orc.lib())
    alg.newton_solve(full)
    for devs in ([0, 0], [0, 0, 0], [0] * 16):
        sh = alg.scenarios.make_problem("C2", np.arange(3, 13), N=10, backend=orc.lib(), devices=devs)
        assert isinstance(sh, alg.ShardedGameProblem) and sum(hi - lo for lo, hi in sh.cuts) == 10
        alg.newton_solve(sh)
        assert np.array_equal(sh.get_traj(), full.batch.get_traj())
        assert np.array_equal(sh.stats.summary["newton_iters"], full.stats.summary["newton_iters"])
        assert alg.sharding.local_counters(sh) == alg.sharding.local_counters(full)
        assert np.array_equal(sh.stats.history(7)["res"], full.stats.history(7)["res"])
        assert np.array_equal(sh.pdtraj.states, full.pdtraj.states)
