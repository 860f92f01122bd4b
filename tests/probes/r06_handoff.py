"""Straggler hand-off on heterogeneous batches: rate of a perturbed batch against the hand-off budget.  usage: python tests/probes/r06_handoff.py [CONFIG GAMES SPREAD]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, algames_jl_amd as alg
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; spread = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
def problem():
    prob = alg.scenarios.make_problem(cfg, np.arange(B)); prob.batch.set_waves_per_game(1)
    rng = np.random.default_rng(5); x0 = prob.x0.copy(); npos = 2 * prob.model.p
    x0[:, :npos] += rng.uniform(-spread, spread, (B, npos)); prob.batch.set_x0(x0); prob._sync_options()
    return prob
ref = None
for K in [0, 8, 10, 12, 13, 14, 16, 20, 24, 32, 48]:
    prob = problem(); b = prob.batch
    if K: b.set_handoff(K)
    for _ in range(3): b.newton_solve_async(init=True, game_id0=0)
    b.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.newton_solve_async(init=True, game_id0=0)
    b.synchronize(); dt = (time.perf_counter() - t0) / 5
    st = b.get_stats(); it = st["newton_iters"]
    if ref is None: ref = it.copy()
    print("%s %d games +-%.1f, hand-off budget %2d: %7.3f ms  %.3g game-iterations/s  handed over %4d games  iterations mean %.1f max %d  counts differ from the plain solve in %d games  status != 0: %d"
          % (cfg, B, spread, K, dt * 1e3, it.sum() / dt, b.get_handoff()[1], it.mean(), it.max(), int((it != ref).sum()), int((st["status"] != 0).sum())), flush=True)
