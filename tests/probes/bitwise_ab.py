"""Are two builds of the library bit-identical on a workload?  usage: python tests/probes/bitwise_ab.py LIB_A LIB_B [CONFIG GAMES]"""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, algames_jl_amd as alg
    out = {}
    for cfg, B in (("C2", 256), ("C3", 128), ("C5", 64)):
        for nw in (1, 0):
            prob = alg.scenarios.make_problem(cfg, np.arange(B)); prob.batch.set_waves_per_game(nw); alg.newton_solve(prob)
            out[(cfg, nw)] = (prob.batch.get_traj().tobytes(), prob.stats.summary["newton_iters"].tobytes())
    pickle.dump(out, open(sys.argv[2], "wb")); sys.exit(0)
res = []
for i, lib in enumerate(sys.argv[1:3]):
    f = "/tmp/bw_%d.pkl" % i
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f], env=dict(os.environ, ALGAMES_HIP_LIB=os.path.abspath(lib)))
    res.append(pickle.load(open(f, "rb")))
for k in res[0]: print(k, "bit-identical" if res[0][k] == res[1][k] else "DIFFERENT")
