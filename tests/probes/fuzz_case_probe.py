"""One random-problem case of tests/test_gpu_fuzz.py against the oracle AND the long-double arbiter, with several refinement settings.
usage: python tests/probes/fuzz_case_probe.py family seed   (family: base | extended | dense)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
import algames_jl_amd as alg
import test_gpu_fuzz as F
fam, seed = sys.argv[1], int(sys.argv[2])
def make():
    if fam == "base": return F._random_pair(alg, orc, np.random.default_rng(1000 + seed), ext=False, arb="x")
    if fam == "extended": return F._random_pair(alg, orc, np.random.default_rng(5000 + seed), ext=True, arb="x")
    rng = np.random.default_rng(13000 + seed); famd = F.DENSE_FAMILIES[seed % len(F.DENSE_FAMILIES)]
    return F._random_pair(alg, orc, rng, ext=bool(seed % 2), force=famd, arb="x")
for setting in ((0, None, None), (2, None, None), (2, None, 0.0), (2, 2.0 ** -40, 0.0)):
    g, o, x, tag = make()
    g.set_refinement(*setting)
    sg, so, sx = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7), x.newton_solve(init=True, game_id0=7)
    zg, zo, zx = g.get_traj(0), o.get_traj(0), x.get_traj(0)
    sc = max(1.0, np.abs(zx).max())
    print("refinement", g.get_refinement(), "| iters g/o/x", sg["newton_iters"], so["newton_iters"], sx["newton_iters"], "corrections", sg["refinements"],
          "| |g-o| %.2e |g-x| %.2e |o-x| %.2e (scale %.1f)" % (np.abs(zg - zo).max(), np.abs(zg - zx).max(), np.abs(zo - zx).max(), sc), "status", sg["status"], so["status"])
print(tag)
