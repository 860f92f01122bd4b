"""max |difference| between two builds of the library on the BASELINE shapes (complements bitwise_ab.py).  usage: r05_diff.py LIB_A LIB_B"""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    import numpy as np, algames_jl_amd as alg
    out = {}
    for cfg, B in (("C2", 256), ("C3", 128), ("C5", 64)):
        for nw in (1, 0):
            prob = alg.scenarios.make_problem(cfg, np.arange(B)); prob.batch.set_waves_per_game(nw); alg.newton_solve(prob)
            out[(cfg, nw)] = (prob.batch.get_traj(), prob.stats.summary["newton_iters"], prob.batch.get_waves_per_game())
    pickle.dump(out, open(sys.argv[2], "wb")); sys.exit(0)
import numpy as np
res = []
for i, lib in enumerate(sys.argv[1:3]):
    f = "/tmp/df_%d.pkl" % i
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f], env=dict(os.environ, ALGAMES_HIP_LIB=os.path.abspath(lib)))
    res.append(pickle.load(open(f, "rb")))
for k in res[0]:
    a, b = res[0][k], res[1][k]
    print(k, "waves", a[2], "max |dz| %.3e" % np.abs(a[0] - b[0]).max(), "iters equal", np.array_equal(a[1], b[1]), "games differing", int((np.abs(a[0] - b[0]).max(axis=1) > 0).sum()), "of", len(a[0]))
