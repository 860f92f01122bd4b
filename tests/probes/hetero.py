"""Tail / imbalance behaviour of the one-game-per-wavefront design: a batch whose games need different numbers
of Newton iterations.  C5 problems (3-player Unicycle, N=30) at 4096 games: iteration counts spread; compare the measured rate with
the rate the same kernel reaches on a homogeneous batch (every game = a copy of one median game)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import algames_jl_amd as alg
def rate(prob, reps=5):
    b = prob.batch; prob._sync_options()
    for _ in range(2): b.newton_solve_async(init=True, game_id0=prob.game_id0)
    torch.cuda.synchronize(); b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): b.newton_solve_async(init=True, game_id0=prob.game_id0)
    b.synchronize(); dt = (time.perf_counter() - t0) / reps
    it = b.get_stats()["newton_iters"]
    return it, dt
for cfg, B, spread in (("C2", 4096, 0.0), ("C2", 4096, 0.3), ("C2", 4096, 0.6), ("C5", 4096, 0.3), ("C5", 4096, 0.6)):
    prob = alg.scenarios.make_problem(cfg, np.arange(B)); prob.batch.set_waves_per_game(1)
    if spread:
        rng = np.random.default_rng(5)
        x0 = prob.x0.copy(); npos = 2 * prob.model.p
        x0[:, :npos] += rng.uniform(-spread, spread, (B, npos))
        prob.batch.set_x0(x0)
    print("spread", spread, end=" ")
    it, dt = rate(prob)
    st = prob.batch.get_stats(); print("converged %d / %d, status!=0: %d" % (st["converged"].sum(), B, (st["status"] != 0).sum()), end=" ")
    print("%s B=%d: iterations min %d median %d mean %.1f max %d | %.3f ms per solve | %.3g game-iterations/s | ideal if the slowest game alone set the time: mean/max = %.2f"
          % (cfg, B, it.min(), np.median(it), it.mean(), it.max(), dt * 1e3, it.sum() / dt, it.mean() / it.max()))
    if spread:
        o = np.argsort(-it)[:3]
        for g in o:
            h = prob.batch.get_history(int(g)); js = h["ls_j"][h["ls_j"] > 0]
            print("   game %d: %d iterations, %.2f line-search trials per iteration (max %d), %d failed searches" % (g, it[g], js.mean() if len(js) else 0, js.max() if len(js) else 0, st["ls_failures"][g]))
    hist = np.bincount(it)
    print("   histogram (iters:count)", {int(k): int(v) for k, v in enumerate(hist) if v})
# ---- what removes the tail: more games than resident slots (the dispatcher backfills), or a second batch in flight on another stream
print("batch-size sweep, C2 +-0.3:")
def spread_problem(B, seed0=0):
    prob = alg.scenarios.make_problem("C2", np.arange(seed0, seed0 + B)); prob.batch.set_waves_per_game(1)
    rng = np.random.default_rng(5 + seed0); x0 = prob.x0.copy(); npos = 2 * prob.model.p
    x0[:, :npos] += rng.uniform(-0.3, 0.3, (B, npos)); prob.batch.set_x0(x0)
    return prob
for B in (4096, 8192, 16384, 32768):
    prob = spread_problem(B); it, dt = rate(prob, reps=3)
    print("  %6d games in one launch: mean %.1f max %d iterations | %.2f ms | %.3g game-iterations/s" % (B, it.mean(), it.max(), dt * 1e3, it.sum() / dt))
for K in (2, 4):
    probs = [spread_problem(4096, seed0=4096 * q) for q in range(K)]
    for p_ in probs: p_._sync_options()
    def go():
        for p_ in probs: p_.batch.newton_solve_async(init=True, game_id0=p_.game_id0)
        for p_ in probs: p_.batch.synchronize()
    go(); go()
    t0 = time.perf_counter(); go(); go(); go(); dt = (time.perf_counter() - t0) / 3
    its = sum(int(p_.batch.get_stats()["newton_iters"].sum()) for p_ in probs)
    print("  %d handles x 4096 games on their own streams, launched together: %.2f ms | %.3g game-iterations/s" % (K, dt * 1e3, its / dt))
