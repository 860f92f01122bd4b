"""One-off long differential fuzz run of the five- / six-player families (GPU box): python tests/probes/fuzz_long_p56.py [n] -- generator and
comparison of tests/test_gpu_fuzz.py::test_fuzz_five_and_six_players, seeds 500000 + i."""
import sys, os, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = []; tot_it = 0; t0 = time.time()
fams = [(F.DI, 5), (F.DI, 6), (F.UNI, 5), (F.UNI, 6), (F.BIC, 5), (F.BIC, 6)]
for seed in range(n):
    rng = np.random.default_rng(500000 + seed)
    model, p = fams[seed % 6]
    g, o, tag = F._random_pair(alg, orc, rng, ext=(model == F.BIC or bool(seed % 2)), force=(model, p), force_d3=False)
    try:
        F._compare_solve(g, o, tag)
        tot_it += int(o.get_stats()["newton_iters"].sum())
    except AssertionError as e:
        bad.append((500000 + seed, (model, p), str(e)[:200]))
print("cases", n, "mismatches", len(bad), "iters", tot_it, "sec %.0f" % (time.time() - t0))
for b in bad[:20]: print(b)
