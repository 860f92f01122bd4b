"""Seed 101156 of the long base family (4-player unicycle, N = 14, collision cost + avoidance + control bounds, 3 outer x 5 inner iterations, none of the
three games converged): the one case of 2000 outside the rule of tests/test_gpu_fuzz.py (job 32).  (1) record by record, the deviation of the HIP path and
of the double oracle from the long-double arbiter; (2) along the arbiter's iterates: condition of the Jacobian, error and backward error of each double
program's Newton direction.   usage: python tests/probes/r06_seed_101156.py [SEED]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 101156
print("library:", os.environ.get("ALGAMES_HIP_LIB", "shipped"))
g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), False, arb="x")
print(seed, tag)
for b in (g, o, x): b.newton_solve(init=True, game_id0=7)
zg, zo, zx = g.get_traj(0), o.get_traj(0), x.get_traj(0)
for game in range(g.B):
    print("game", game, "|hip-x| %.2e |orc-x| %.2e |hip-orc| %.2e scale %.1f" % (np.abs(zg[game] - zx[game]).max(), np.abs(zo[game] - zx[game]).max(), np.abs(zg[game] - zo[game]).max(), np.abs(zx[game]).max()))
    hg, ho, hx = g.get_history(game), o.get_history(game), x.get_history(game)
    s0 = 1e-3 * abs(hx["res"][0])
    for rec in range(min(len(hg), len(ho), len(hx))):
        eg = max(abs(hg[f][rec] - hx[f][rec]) / max(abs(hx[f][rec]), s0) for f in F.ARB_FIELDS)
        eo = max(abs(ho[f][rec] - hx[f][rec]) / max(abs(hx[f][rec]), s0) for f in F.ARB_FIELDS)
        print("   rec %2d  ls_j hip/orc/x %d %d %d  alpha %.3g  res_x %.4g  dev hip %.1e orc %.1e" % (rec, hg["ls_j"][rec], ho["ls_j"][rec], hx["ls_j"][rec], hx["alpha"][rec], hx["res"][rec], eg, eo))
# (2) directions along a common path: every program is set to the arbiter's iterate, the arbiter's direction with alpha = 0.5 moves it
g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), False, arb="x")
reg = float(tag[5]["reg_0"])
for b in (g, o, x): b.init_traj(game_id0=7); b.rollout()
for it in range(10):
    Jx = x.residual_jacobian(reg); rx = x.residual(reg=reg)[0]
    dg, do_, dx = g.newton_direction(reg)[0], o.newton_direction(reg)[0], x.newton_direction(reg)[0]
    for game in range(g.B):
        sd = np.abs(dx[game]).max()
        be = lambda d: np.abs(Jx[game] @ d + rx[game]).max() / (np.abs(Jx[game]).sum(1).max() * np.abs(d).max() + np.abs(rx[game]).max())
        print("it", it, "game", game, "cond %.1e |res| %.3g |d| %.3g" % (np.linalg.cond(np.asarray(Jx[game], float)), np.abs(rx[game]).sum(), sd),
              "dir err: hip %.1e orc %.1e" % (np.abs(dg[game] - dx[game]).max() / sd, np.abs(do_[game] - dx[game]).max() / sd),
              "bwd err: hip %.1e orc %.1e x %.1e" % (be(dg[game]), be(do_[game]), be(dx[game])))
    x.update_traj(0.5)
    z = x.get_traj()
    for b in (g, o): b.set_traj(z)
