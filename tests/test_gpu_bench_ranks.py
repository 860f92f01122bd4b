"""bench.py's N > 1 path on a one-GPU box: two ranks started by bench.py itself (torch.distributed.run on 127.0.0.1), sharing
device 0 through the script's test hook, gloo carrying the barrier and the counter reduction.  Checks what the driver's multi-GPU
runs rely on -- contiguous global scenario ids per rank, whole-job totals, one JSON line from rank 0, weak and strong splits -- with
the HIP path doing the solves in both processes.  RCCL itself needs a second device and is not exercised here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, shared=False):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    if shared:
        env["ALGAMES_BENCH_SHARED_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", *args],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints the one line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_weak_and_strong_split_equal_the_single_rank_job():
    one = _bench("--games-per-gpu", "512")
    weak = _bench("--gpus", "2", "--games-per-gpu", "256", shared=True)
    strong = _bench("--gpus", "2", "--scaling", "strong", "--games-total", "512", shared=True)
    assert one["n_gpus"] == 1 and weak["n_gpus"] == 2 and strong["n_gpus"] == 2
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    for d in (weak, strong):
        c = d["config"]
        assert c["games_total"] == 512 and c["games_per_gpu"] == 256 and "TEST HOOK" in c["parallelism"]
        # scenario ids are global: two 256-game shards do the work of the 512-game batch, game for game
        assert c["newton_iters_per_solve_total"] == one["config"]["newton_iters_per_solve_total"]
        assert d["games_converged"] == one["games_converged"] == 512 and d["games_failed"] == 0
        assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - c["newton_iters_per_solve_total"]) <= 1e-6 * c["newton_iters_per_solve_total"]


@pytest.mark.gpu
def test_more_ranks_than_devices_is_refused_without_the_hook():
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("ALGAMES_BENCH_SHARED_DEVICE", None)
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pmc"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_rccl_communicator_barrier_and_reduction_run_on_the_device_with_a_world_of_one():
    """the nccl (= RCCL) backend has to create a communicator and carry the barrier and the two all-reduces of
    `sharding.reduce_counters` on the MI355X at least once; a one-GPU box can do that with a world of one.  Both launch forms:
    `--force-dist` (bench.py makes its own rendezvous) and torchrun with `--nproc-per-node 1` (WORLD_SIZE=1 in the environment)."""
    plain = _bench("--games-per-gpu", "512")
    forced = _bench("--games-per-gpu", "512", "--force-dist")
    assert plain["config"]["collectives"].startswith("none") and forced["config"]["collectives"].startswith("nccl (RCCL), world 1")
    env = dict(os.environ); env.pop("ALGAMES_BENCH_SHARED_DEVICE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--games-per-gpu", "512"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    torchrun = json.loads(lines[0])
    assert torchrun["config"]["collectives"].startswith("nccl (RCCL), world 1")
    for d in (forced, torchrun):
        assert d["n_gpus"] == 1 and d["games_converged"] == plain["games_converged"] == 512 and d["games_failed"] == 0
        assert d["config"]["newton_iters_per_solve_total"] == plain["config"]["newton_iters_per_solve_total"]
        assert set(d) == set(plain) and set(d["config"]) == set(plain["config"]) and d["metric"] == plain["metric"] and d["unit"] == plain["unit"]


@pytest.mark.gpu
def test_eight_ranks_of_the_c4_job_tile_65536_scenarios_and_print_one_line():
    """BASELINE configs[3] as the driver launches it -- `bench.py --config C4 --gpus 8`, eight processes, 8192 scenarios each -- on a one-GPU
    box: the ranks share device 0 through the script's test hook (gloo carries the barrier, the gather of the shard ranges and the counter
    reduction; RCCL needs eight devices).  The multi-process path is what is checked: one JSON line, 65 536 scenarios, contiguous shards
    in rank order, whole-job totals; no PMC / CPU-baseline legs in the ranks."""
    d = _bench("--config", "C4", "--gpus", "8", shared=True)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["games_per_gpu"] == 8192 and c["games_total"] == 65536 and "TEST HOOK" in c["parallelism"]
    assert c["shard_ranges"] == [[r * 8192, (r + 1) * 8192] for r in range(8)]
    assert d["games_converged"] == 65536 and d["games_failed"] == 0
    assert c["newton_iters_per_solve_total"] >= 65536 * 5
    assert "cpu_baseline" not in d and d["roofline"].get("traffic") is None       # rank 0 of a multi-rank job runs no profiler child and no CPU leg
    assert c["collectives"].startswith("gloo, world 8")


@pytest.mark.gpu
def test_a_failing_rank_fails_the_job():
    env = dict(os.environ, ALGAMES_BENCH_SHARED_DEVICE="1", ALGAMES_BENCH_FAIL_RANK="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--games-per-gpu", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pmc"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]                # no result line from a job that lost a rank
