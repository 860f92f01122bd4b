"""Quad-team kernels against the one-wavefront kernels and the oracle (C2 scenarios), then timing at 4096 games."""
import sys, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import algames_jl_amd as alg
import oracle as orc
import torch
def run(ids, N, qt, backend=None):
    p = alg.scenarios.make_problem("C2", ids, N=N, backend=backend) if backend else alg.scenarios.make_problem("C2", ids, N=N)
    if backend is None:
        p.batch.set_waves_per_game(1); p.batch.set_quad_team(qt)
        assert p.batch.get_quad_team() == qt
    alg.newton_solve(p)
    return p
for N in (12, 40):
    ids = np.arange(8)
    a = run(ids, N, 1); b = run(ids, N, 0); c = run(ids, N, 0, backend=orc.lib())
    sa, sb, sc = a.stats.summary, b.stats.summary, c.stats.summary
    print("N", N, "iters qt", sa["newton_iters"].tolist(), "1w", sb["newton_iters"].tolist(), "orc", sc["newton_iters"].tolist())
    print("  status qt", sa["status"].tolist(), "outer", sa["outer_iters"].tolist(), sc["outer_iters"].tolist())
    za, zb, zc = a.batch.get_traj(), b.batch.get_traj(), c.batch.get_traj()
    print("  max|qt-1w| %.3e  max|qt-orc| %.3e  max|1w-orc| %.3e" % (np.abs(za - zb).max(), np.abs(za - zc).max(), np.abs(zb - zc).max()))
if len(sys.argv) > 1:
  for nb in (4096, 8192, 16384):
    ids = np.arange(nb)
    for qt in (0, 1):
        p = alg.scenarios.make_problem("C2", ids, N=40)
        p.batch.set_waves_per_game(1); p.batch.set_quad_team(qt)
        alg.newton_solve(p); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            p = alg.scenarios.make_problem("C2", ids, N=40); p.batch.set_waves_per_game(1); p.batch.set_quad_team(qt)
            torch.cuda.synchronize(); t0 = time.perf_counter(); alg.newton_solve(p); p.batch.synchronize() if hasattr(p.batch, "synchronize") else None; torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        it = p.stats.summary["newton_iters"].sum()
        print(nb, "games qt", qt, "ms", [round(t * 1e3, 3) for t in ts], "iters", int(it), "M/s %.2f" % (it / min(ts) / 1e6))
