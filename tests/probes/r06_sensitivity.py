"""CPU only: the oracle against itself with x0 (1 + eps s), s = +-1 per entry, on seeds of the long generator (tests/probes/fuzz_long_r6.py) -- does the PROBLEM
amplify a sub-ulp perturbation as far as the HIP path and the oracle part on the GPU?   usage: python tests/probes/r06_sensitivity.py SEED [SEED ...]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
class FakeAlg:
    def __getattr__(self, k): return getattr(alg, k)
    def hip_lib(self): return orc.lib()       # both sides on the oracle: the first one gets a perturbed x0
def pair(seed):
    rng = np.random.default_rng(seed); fa = FakeAlg()
    if seed >= 900000: i = seed - 900000; model = (F.DI, F.UNI, F.BIC)[i % 3]; return F._random_pair(fa, orc, rng, ext=(model == F.BIC or bool((i // 3) % 2)), force=(model, 10), force_d3=False)
    if seed >= 800000: return F._random_pair(fa, orc, rng, ext=False, force=(F.DI, 1 + (seed - 800000) % 4), force_d3=False, d_override=1)
    if seed >= 700000: i = seed - 700000; model, p = F.P789_FAMILIES[i % 9]; return F._random_pair(fa, orc, rng, ext=(model == F.BIC or bool(i % 2)), force=(model, p), force_d3=False)
    if seed >= 600000: i = seed - 600000; model, p = F.P56_FAMILIES[i % 6]; return F._random_pair(fa, orc, rng, ext=(model == F.BIC or bool(i % 2)), force=(model, p), force_d3=False)
    if seed >= 500000: return F._random_pair(fa, orc, rng, True, d3=True, force=(0, 3))
    if seed >= 400000: return F._random_pair(fa, orc, rng, True, d3=True, force=(3, 2))
    if seed >= 300000: return F._random_pair(fa, orc, rng, True, d3=True)
    return F._random_pair(fa, orc, rng, seed >= 200000)
for seed in [int(a) for a in sys.argv[1:]]:
    for eps in (1e-15, 1e-13):
        g, o, tag = pair(seed)
        x0 = o.get_x0(); g.set_x0(x0 * (1 + eps * np.sign(np.sin(np.arange(x0.size).reshape(x0.shape)))))
        sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
        zg, zo = g.get_traj(0), o.get_traj(0)
        same = all(np.array_equal(sg[f], so[f]) for f in ("status", "outer_iters", "newton_iters", "ls_failures"))
        hs = all(np.array_equal(g.get_history(q)["ls_j"], o.get_history(q)["ls_j"]) if len(g.get_history(q)) == len(o.get_history(q)) else False for q in range(g.B))
        print(seed, tag[:4], "eps %.0e" % eps, "counts equal" if same else "COUNTS DIFFER", "step sizes equal" if hs else "STEP SIZES DIFFER", "max|dz|/scale per game", ["%.1e" % v for v in np.abs(zg - zo).max(axis=1) / np.maximum(1.0, np.abs(zo).max(axis=1))], "status", so["status"].tolist(), "scale", ["%.3g" % v for v in np.abs(zo).max(axis=1)], flush=True)
