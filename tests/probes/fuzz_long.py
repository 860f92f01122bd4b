"""One-off long differential fuzz run (GPU box): python tests/probes/fuzz_long.py [n_seeds] -- same generator as tests/test_gpu_fuzz.py."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = []; tot_it = 0; tot_fail = 0; t0 = time.time()
only3d = os.environ.get("FUZZ_ONLY_3D") == "1"
for ext, base, d3 in (((True, 300000, True),) if only3d else ((False, 100000, False), (True, 200000, False), (True, 300000, True))):
    for seed in range(n):
        rng = np.random.default_rng(base + seed)
        g, o, tag = F._random_pair(alg, orc, rng, ext, d3=d3)
        try:
            F._compare_solve(g, o, tag)
            s = o.get_stats(); tot_it += int(s["newton_iters"].sum()); tot_fail += int(s["ls_failures"].sum())
        except AssertionError as e:
            bad.append((base + seed, str(e)[:300]))
print("cases", (1 if only3d else 3) * n, "mismatches", len(bad), "iters", tot_it, "ls_failures", tot_fail, "sec %.0f" % (time.time() - t0))
for b in bad[:20]: print(b)
