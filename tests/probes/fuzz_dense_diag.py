import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
np.set_printoptions(linewidth=200, precision=4)
for seed in [int(a) for a in sys.argv[1:]]:
    for nwforce in (None, 1):
        rng = np.random.default_rng(seed)
        fam = F.DENSE_FAMILIES[(seed - 400000) % len(F.DENSE_FAMILIES)]
        g, o, tag = F._random_pair(alg, orc, rng, ext=bool((seed - 400000) % 2), force=fam)
        if nwforce is not None or (seed - 400000) % 3 == 0: g.set_waves_per_game(1)
        print("=== seed", seed, "nw", g.get_waves_per_game(), tag[:5])
        sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
        for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
            print("  ", f, sg[f], so[f])
        zg, zo = g.get_traj(0), o.get_traj(0)
        print("   max|dz| per game", np.abs(zg - zo).max(axis=1), "scale", np.abs(zo).max(axis=1))
        for game in range(3):
            hg, ho = g.get_history(game), o.get_history(game)
            L = min(len(hg), len(ho))
            d = [(i, hg["ls_j"][i], ho["ls_j"][i], hg["res"][i], ho["res"][i]) for i in range(L) if hg["ls_j"][i] != ho["ls_j"][i] or abs(hg["res"][i] - ho["res"][i]) > 1e-9 * abs(ho["res"][i])]
            print("   game", game, "history len", len(hg), len(ho), "first diffs", d[:3])
