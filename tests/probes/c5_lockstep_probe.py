"""C5 closed loop, lock-step: the HIP path drives the loop; before every MPC step the oracle receives the HIP path's complete
solver state (x0, warm-start trajectory, multipliers, penalties) and both run that one newton_solve!."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
ids = np.arange(128, 192); T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pg = alg.scenarios.make_problem("C5", ids); po = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
bg, bo = pg.batch, po.batch
import os
bg.set_waves_per_game(int(os.environ.get("NW", "0")))
print("waves per game", bg.get_waves_per_game())
rows = []
for t in range(T):
    if t == 1:
        for p_ in (pg, po):
            p_.opts.shift, p_.opts.dual_reset = 1, False; p_._sync_options()
    z = bg.get_traj(0); lam, mu = bg.get_con_duals()
    bo.set_x0(z[:, :bg.n].copy()); bo.set_traj(z, 0); bo.set_con_duals(lam, mu)
    sg = bg.newton_solve(init=True, game_id0=pg.game_id0 + t * 1000003)
    so = bo.newton_solve(init=True, game_id0=pg.game_id0 + t * 1000003)
    zg, zo = bg.get_traj(0), bo.get_traj(0)
    err = np.abs(zg - zo).max(axis=1)
    for g in range(len(ids)):
        hg, ho = bg.get_history(g, 4)[:1], bo.get_history(g, 4)[:1]
        first = max(abs(hg[f][0] - ho[f][0]) / max(1e-300, abs(ho[f][0]), 1e-12) for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio")) if len(hg) and len(ho) else np.nan
        rows.append((t, g, sg["newton_iters"][g], so["newton_iters"][g], sg["ls_failures"][g], so["ls_failures"][g], sg["converged"][g], so["converged"][g], err[g], first))
    bg.mpc_advance()
r = np.array(rows)
same = r[:, 2] == r[:, 3]
clean = (r[:, 4] == 0) & (r[:, 5] == 0) & (r[:, 6] == 1) & (r[:, 7] == 1)
print("solves", len(r), "identical iteration counts", int(same.sum()), "mismatching", int((~same).sum()))
print("clean solves (converged, no failed line search, both)", int(clean.sum()), "of which mismatching", int((clean & ~same).sum()))
print("max traj err over solves with identical counts %.3e ; over clean+identical %.3e" % (r[same, 8].max(), r[clean & same, 8].max()))
print("max first-record rel diff %.3e" % np.nanmax(r[:, 9]))
bad = r[~same]
print("mismatching solves: iters gpu/orc, lsfail gpu/orc, conv gpu/orc")
for x in bad[:40]: print(int(x[0]), int(x[1]), int(x[2]), int(x[3]), int(x[4]), int(x[5]), int(x[6]), int(x[7]), "%.2e" % x[8])
big = r[same & (r[:, 8] > 1e-8)]
print("identical counts but traj err > 1e-8:", len(big))
for x in big[:20]: print(int(x[0]), int(x[1]), int(x[2]), int(x[4]), int(x[6]), "%.2e" % x[8])
