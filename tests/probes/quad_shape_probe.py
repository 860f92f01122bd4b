import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import algames_jl_amd as alg
for p, B in ((2, 4096), (2, 1024), (3, 2048), (4, 1024)):
    for nw in (1, 4):
        pg = alg.scenarios.make_problem("Q", np.arange(B), p=p); pg.batch.set_waves_per_game(nw)
        alg.newton_solve(pg)
        t0 = time.time()
        for _ in range(3): alg.newton_solve(pg)
        t = (time.time() - t0) / 3
        print("Q p=%d B=%d nw=%d: %.1f ms  %.3f M game-iters/s" % (p, B, nw, 1e3 * t, pg.stats.summary["newton_iters"].sum() / t / 1e6))
