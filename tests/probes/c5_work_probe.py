"""Work accounting of the C5 receding-horizon loop at 64 seeds: Newton directions, record! passes and line-search trial passes per
game (step-wise loop, statistics read back per MPC step), against the fused loop's wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import algames_jl_amd as alg
ids = np.arange(0, 64); T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pg = alg.scenarios.make_problem("C5", ids)
b = pg.batch; pg._sync_options()
its = np.zeros((T, 64), int); recs = np.zeros((T, 64), int); trials = np.zeros((T, 64), int); outers = np.zeros((T, 64), int); lsf = np.zeros((T, 64), int)
for t in range(T):
    if t == 1:
        pg.opts.shift, pg.opts.dual_reset = 1, False; pg._sync_options()
    st = b.newton_solve(init=True, game_id0=pg.game_id0 + t * 1000003)
    its[t] = st["newton_iters"]; recs[t] = st["records"]; outers[t] = st["outer_iters"]; lsf[t] = st["ls_failures"]
    for g in range(64):
        h = b.get_history(g); trials[t, g] = h["ls_j"].sum()
    b.mpc_advance()
print("per game over %d steps: directions mean %.0f max %d | records mean %.0f max %d | trial passes mean %.0f max %d | outer its mean %.0f | ls failures total %d" % (
    T, its.sum(0).mean(), its.sum(0).max(), recs.sum(0).mean(), recs.sum(0).max(), trials.sum(0).mean(), trials.sum(0).max(), outers.sum(0).mean(), lsf.sum()))
print("per MPC step: directions mean %.2f, records %.2f, trial passes %.2f, outer iterations %.2f" % (its.mean(), recs.mean(), trials.mean(), outers.mean()))
print("histogram of ls_j sum / direction: %.2f trials per direction" % (trials.sum() / its.sum()))
pf = alg.scenarios.make_problem("C5", ids)
for nw in (1, 4):
    pf = alg.scenarios.make_problem("C5", ids); pf.batch.set_waves_per_game(nw)
    alg.mpc_solve(pf, 5)
    pf = alg.scenarios.make_problem("C5", ids); pf.batch.set_waves_per_game(nw)
    t0 = time.time(); it, cv, _ = alg.mpc_solve(pf, T); t1 = time.time()
    print("fused loop nw=%d: %.3f s, %d directions total (max game %d) -> %.0f game-iters/s ; per direction of the slowest game %.1f us" % (nw, t1 - t0, it.sum(), it.max(), it.sum() / (t1 - t0), 1e6 * (t1 - t0) / it.max()))
