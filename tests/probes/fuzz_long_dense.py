"""One-off long differential fuzz run of the dense-direction families (GPU box): python tests/probes/fuzz_long_dense.py [n_seeds] --
generator and comparison of tests/test_gpu_fuzz.py::test_fuzz_dense_direction_instantiations, seeds 400000 + i."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = []; tot_it = 0; tot_fail = 0; t0 = time.time(); per = {}
for seed in range(n):
    rng = np.random.default_rng(400000 + seed)
    fam = F.DENSE_FAMILIES[seed % len(F.DENSE_FAMILIES)]
    g, o, tag = F._random_pair(alg, orc, rng, ext=bool(seed % 2), force=fam)
    if seed % 3 == 0:
        g.set_waves_per_game(1)
    try:
        F._compare_solve(g, o, tag)
        s = o.get_stats(); tot_it += int(s["newton_iters"].sum()); tot_fail += int(s["ls_failures"].sum())
        per[fam] = per.get(fam, 0) + 1
    except AssertionError as e:
        bad.append((400000 + seed, fam, str(e)[:300]))
print("cases", n, "mismatches", len(bad), "iters", tot_it, "ls_failures", tot_fail, "sec %.0f" % (time.time() - t0), "ok per family", per)
for b in bad[:30]: print(b)
