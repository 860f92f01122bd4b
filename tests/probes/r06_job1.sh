#!/bin/bash
# round 6, job 1: baseline on this round's first box (HEAD of round 5): GPU suite, default bench line, phase-cycle profile of C2, heterogeneous C2
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/gputest.txt; tail -2 $O/gputest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 4096 > $O/phase_cycles_c2_4096.txt 2>&1; cat $O/phase_cycles_c2_4096.txt
python - > $O/hetero_c2.txt 2>&1 <<'PY'
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch, algames_jl_amd as alg
for spread in (0.0, 0.3):
    prob = alg.scenarios.make_problem("C2", np.arange(4096)); prob.batch.set_waves_per_game(1)
    if spread:
        rng = np.random.default_rng(5); x0 = prob.x0.copy(); x0[:, :6] += rng.uniform(-spread, spread, (4096, 6)); prob.batch.set_x0(x0)
    b = prob.batch; prob._sync_options()
    for _ in range(3): b.newton_solve_async(init=True, game_id0=0)
    b.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.newton_solve_async(init=True, game_id0=0)
    b.synchronize(); dt = (time.perf_counter() - t0) / 5
    it = b.get_stats()["newton_iters"]
    print("C2 4096 games spread %.1f: %.3f ms, %.3g game-iterations/s, iterations mean %.1f max %d; histogram" % (spread, dt * 1e3, it.sum() / dt, it.mean(), it.max()), {int(k): int(v) for k, v in enumerate(np.bincount(it)) if v})
PY
cat $O/hetero_c2.txt | cut -c1-400
