"""Full solve of a quadrotor fuzz seed under several refinement settings: per-game distance of the final trajectory / opt_vio from the arbiter.
usage: python tests/probes/r06_seed_solve.py SEED"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
np.set_printoptions(linewidth=220, precision=3)
seed = int(sys.argv[1])
def make():
    rng = np.random.default_rng(seed)
    return F._random_pair(alg, orc, rng, True, d3=True, force=(3, 2), arb="x")
g, o, x, tag = make()
so, sx = o.newton_solve(init=True, game_id0=7), x.newton_solve(init=True, game_id0=7)
zo, zx = o.get_traj(0), x.get_traj(0)
print(tag[:4], "oracle vs arbiter |dz|", np.abs(zo - zx).max(axis=1), "|z|", np.abs(zx).max(axis=1), "opt_vio diff", np.abs(so["last"]["opt_vio"] - sx["last"]["opt_vio"]), "iters", so["newton_iters"])
for name, rs in (("gate off", (0, 2.0 ** -34)), ("1 / 2^-34", (1, 2.0 ** -34)), ("default 6 / 2^-34", None), ("6 / 1e-30", (6, 1e-30)), ("4 forced", (4, 0.0)), ("8 forced", (8, 0.0))):
    g, _, _, _ = make()
    if rs is not None: g.set_refinement(rs[0], rs[1], 1.6e5)
    sg = g.newton_solve(init=True, game_id0=7)
    zg = g.get_traj(0)
    print("  %-18s |z - zx| %s  opt_vio diff %s  status %s iters %s corrections %s" % (name, np.abs(zg - zx).max(axis=1), np.abs(sg["last"]["opt_vio"] - sx["last"]["opt_vio"]), sg["status"], sg["newton_iters"], sg["refinements"]))
