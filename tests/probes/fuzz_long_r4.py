"""Round-4 long differential fuzz run (GPU box): the generator and the acceptance rule of tests/test_gpu_fuzz.py (1e-8, arbiter rule where
a problem amplifies rounding) over n seeds per family: base / extended / 3-D, plus the dense-direction families.
usage: python tests/probes/fuzz_long_r4.py [n_seeds] [refine_max]     (refine_max 0 = the round-3 arithmetic, for comparison)"""
import sys, os, time, io, contextlib
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rmax = int(sys.argv[2]) if len(sys.argv) > 2 else -1
t0 = time.time()
for name, ext, base, d3, force in (("base", False, 100000, False, None), ("extended", True, 200000, False, None), ("3-D", True, 300000, True, None),
                                   ("quadrotor x2", True, 400000, True, (3, 2)), ("DI d=3 x3", True, 500000, True, (0, 3))):
    bad = []; consulted = 0; it = 0; fails = 0; corr = 0; nn = n if force is None else max(1, n // 4)
    for seed in range(nn):
        rng = np.random.default_rng(base + seed)
        g, o, x, tag = F._random_pair(alg, orc, rng, ext, d3=d3, force=force, arb="x")
        if rmax >= 0: g.set_refinement(rmax)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                F._compare_solve(g, o, tag, x=x)
        except AssertionError as e:
            bad.append((base + seed, str(e)[:200] + " ... " + str(e)[-260:].replace("\n", " ")))
        consulted += "arbiter consulted" in buf.getvalue()
        s = g.get_stats(); it += int(s["newton_iters"].sum()); fails += int(s["ls_failures"].sum()); corr += int(s["refinements"].sum())
    print("%-13s cases %d outside the rule %d, arbiter consulted %d, Newton iterations %d, failed line searches %d, correction solves %d, %.0f s"
          % (name, nn, len(bad), consulted, it, fails, corr, time.time() - t0), flush=True)
    for b in bad[:6]: print("   ", b, flush=True)
