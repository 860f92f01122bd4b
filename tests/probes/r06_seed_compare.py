"""One seed of the base family of tests/probes/fuzz_long_r6.py under the acceptance rule of tests/test_gpu_fuzz.py, with the library named by
ALGAMES_HIP_LIB (default: the shipped one).  usage: python tests/probes/r06_seed_compare.py SEED [SEED ...]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
print("library:", os.environ.get("ALGAMES_HIP_LIB", "shipped"))
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    g, o, x, tag = F._random_pair(alg, orc, rng, seed >= 200000, arb="x")
    try:
        F._compare_solve(g, o, tag, x=x); print(seed, "inside the rule")
    except AssertionError as e:
        print(seed, "OUTSIDE the rule:", str(e)[:3000])
    sg, so = g.get_stats(), o.get_stats()
    for f in ("status", "outer_iters", "newton_iters", "converged", "ls_failures"):
        print("  ", f, sg[f].tolist(), so[f].tolist())
