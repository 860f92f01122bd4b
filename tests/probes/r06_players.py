"""Round-6 probe (GPU box): solve rate of the dense-direction player counts, five to ten DoubleIntegrator (d = 2) / Unicycle players crossing a circle,
N = 20, 256 games (one game per CU from seven players on).  Prints game-Newton-iterations/s per shape.
usage: python tests/probes/r06_players.py [games]"""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, dt = 20, 0.1
for model, name in ((0, "DoubleIntegrator"), (1, "Unicycle")):
    for p in (5, 6, 7, 8, 9, 10):
        g = alg.Batch(alg.hip_lib(), model, p, N, dt, B)
        rng = np.random.default_rng(p)
        ni = g.n // p
        ang = 2 * np.pi * np.arange(p) / p
        x0 = np.zeros((B, g.n)); xf = np.zeros((B, p, ni))
        x0[:, 0:p] = np.cos(ang) + 0.02 * rng.normal(size=(B, p)); x0[:, p:2 * p] = np.sin(ang) + 0.02 * rng.normal(size=(B, p))
        xf[:, :, 0] = -np.cos(ang); xf[:, :, 1] = -np.sin(ang)
        if model == 1:
            x0[:, 2 * p:3 * p] = ang + np.pi; xf[:, :, 2] = ang + np.pi
        g.set_x0(x0); g.set_lqr(np.ones((B, p, ni)), 0.1 * np.ones((B, p, g.mi)), xf, np.zeros((B, p, g.mi)))
        g.add_collision_avoidance(np.full(p, 0.15)); g.add_control_bound(np.full(g.m, 5.0), np.full(g.m, -5.0))
        g.newton_solve(init=True, game_id0=0); g.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            g.newton_solve_async(init=True, game_id0=0)
        g.synchronize(); t = (time.perf_counter() - t0) / 2
        s = g.get_stats(); it = int(s["newton_iters"].sum())
        print("%-16s p = %2d  n = %2d  %d games  %7.1f ms per solve  %6d game-iterations  %8.1f K game-iterations/s  converged %d / %d" %
              (name, p, g.n, B, 1e3 * t, it, 1e-3 * it / t, int((s["status"] == 0).sum()), B), flush=True)
