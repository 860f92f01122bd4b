"""How much does the CPU oracle itself amplify a 1e-13 relative perturbation of x0 on a fuzz problem?  (CPU only.)  On the seeds where
the HIP path and the oracle end outside the comparison tolerances the answer is 1e6 .. 1e10 in exactly the games that differ; on
ordinary seeds it is ~1: those mismatches are conditioning of the (diverging) problem, not an arithmetic difference that matters.
usage: python tests/probes/fuzz_sensitivity.py [seed ...]   (five- / six-player family, seeds 500000 + i)"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
class FakeAlg:
    def __getattr__(self, k): return getattr(alg, k)
    def hip_lib(self): return orc.lib()       # both sides on the oracle: second one gets a perturbed x0
fams = [(F.DI, 5), (F.DI, 6), (F.UNI, 5), (F.UNI, 6), (F.BIC, 5), (F.BIC, 6)]
for seed in ([int(a) for a in sys.argv[1:]] or [500258, 500365, 500262, 500001, 500002, 500003]):
    rng = np.random.default_rng(seed)
    model, p = fams[(seed - 500000) % 6]
    g, o, tag = F._random_pair(FakeAlg(), orc, rng, ext=(model == F.BIC or bool((seed - 500000) % 2)), force=(model, p), force_d3=False)
    x0 = o.get_x0(); g.set_x0(x0 * (1 + 1e-13 * np.sign(np.sin(np.arange(x0.size).reshape(x0.shape)))))
    sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
    zg, zo = g.get_traj(0), o.get_traj(0)
    print(seed, (model, p), "iters", so["newton_iters"], "amplification of a 1e-13 relative x0 perturbation: max|dz|/scale per game", np.abs(zg - zo).max(axis=1) / np.abs(zo).max(axis=1))
