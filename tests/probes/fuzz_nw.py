"""One fuzz case (generator of tests/test_gpu_fuzz.py) under both kernel shapes: python tests/probes/fuzz_nw.py SEED [ext] [d3]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
for seed in [int(a) for a in sys.argv[1].split(",")]:
    ext = seed >= 200000; d3 = seed >= 300000
    out = {}
    for nw in (1, 0):
        rng = np.random.default_rng(seed)
        g, o, tag = F._random_pair(alg, orc, rng, ext, d3=d3)
        g.set_waves_per_game(nw)
        w = g.get_waves_per_game()
        sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
        zg, zo = g.get_traj(0), o.get_traj(0)
        hg, ho = g.get_history(0), o.get_history(0)
        n = min(len(hg), len(ho))
        first_bad = next((i for i in range(n) if not np.isclose(hg["res"][i], ho["res"][i], rtol=1e-6, atol=1e-12)), None)
        print(seed, tag[:4], "nw", w, "iters gpu", sg["newton_iters"], "orc", so["newton_iters"], "status", sg["status"], so["status"], "zerr %.2e" % np.abs(zg - zo).max(), "first bad record", first_bad, "of", len(hg), len(ho))
        if first_bad is not None:
            print("   gpu res", hg["res"][max(0, first_bad - 1):first_bad + 2], "alpha", hg["alpha"][max(0, first_bad - 1):first_bad + 2], "ls_j", hg["ls_j"][max(0, first_bad - 1):first_bad + 2])
            print("   orc res", ho["res"][max(0, first_bad - 1):first_bad + 2], "alpha", ho["alpha"][max(0, first_bad - 1):first_bad + 2], "ls_j", ho["ls_j"][max(0, first_bad - 1):first_bad + 2])
