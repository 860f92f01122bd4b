"""Step-wise timing of the quadrotor configurations: assemble pass (record!), assemble + Newton direction, one line-search call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import algames_jl_amd as alg
def tm(f, reps=5):
    f(); t0 = time.time()
    for _ in range(reps): f()
    return (time.time() - t0) / reps
for p, B in ((2, 4096), (4, 1024), (1, 4096)):
    pg = alg.scenarios.make_problem("Q", np.arange(B), p=p)
    b = pg.batch
    alg.newton_solve(pg)
    s = pg.stats.summary
    t_solve = tm(lambda: alg.newton_solve(pg), 3)
    b.init_traj(game_id0=0)
    t_rec = tm(lambda: b.record())
    t_dir = tm(lambda: b.newton_direction(1e-5))
    rn = b.residual()[1]
    t_res = tm(lambda: b.residual())
    t_ls = tm(lambda: b.line_search(rn, 1e-5))
    print("Q p=%d B=%d: solve %.1f ms (%.1f iters/game) | record %.2f ms | record+direction %.2f ms | residual(+D2H) %.2f ms | line search %.2f ms" % (
        p, B, 1e3 * t_solve, s["newton_iters"].mean(), 1e3 * t_rec, 1e3 * t_dir, 1e3 * t_res, 1e3 * t_ls))
