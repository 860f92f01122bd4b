import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    g, o, tag = F._random_pair(alg, orc, rng, seed >= 200000)
    sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
    print(seed, tag[:4], [k for k, v in tag[4] if v])
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        print("  ", f, sg[f].tolist(), so[f].tolist())
    for game in range(3):
        hg, ho = g.get_history(game), o.get_history(game)
        m = min(len(hg), len(ho))
        d = [i for i in range(m) if hg["ls_j"][i] != ho["ls_j"][i] or abs(hg["res"][i] - ho["res"][i]) > 1e-6 * max(1, abs(ho["res"][i]))]
        print("   game", game, "len", len(hg), len(ho), "first diff at", d[:1])
        if d:
            i = d[0]
            for j in range(max(0, i - 2), min(m, i + 2)):
                print("      rec", j, "gpu", hg[j], "\n             orc", ho[j])
    hg, ho = g.get_history(0), o.get_history(0)
    print("   rel diff of res per record, game 0:", ["%.1e" % (abs(a - b) / max(abs(b), 1e-300)) for a, b in zip(hg["res"], ho["res"])])
    print("   res:", ["%.3g" % b for b in ho["res"]], "opt_vio", ["%.3g" % b for b in ho["opt_vio"]])
    zg, zo = g.get_traj(0), o.get_traj(0)
    print("   max |z_gpu - z_orc| per game:", np.abs(zg - zo).max(axis=1), "scale", np.abs(zo).max(axis=1))
    for game in range(3):
        hg, ho = g.get_history(game), o.get_history(game)
        print("   game", game, "rel diff res:", ["%.1e" % (abs(a - b) / max(abs(b), 1e-300)) for a, b in zip(hg["res"], ho["res"])])
