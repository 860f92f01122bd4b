"""Round-6 long differential fuzz run (GPU box): the generator and the acceptance rule of tests/test_gpu_fuzz.py (1e-8, arbiter rules) over n seeds per
tile-path family and n / 4 per dense family, incl. the round-6 families (seven to ten players, DoubleIntegrator d = 1).
usage: python tests/probes/fuzz_long_r6.py [n_seeds [family name]]"""
import sys, os, time, io, contextlib
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.time()
fams = [("base", False, 100000, False, None, None, n), ("extended", True, 200000, False, None, None, n), ("3-D", True, 300000, True, None, None, n),
        ("quadrotor x2", True, 400000, True, (3, 2), None, n // 4), ("DI d=3 x3", True, 500000, True, (0, 3), None, n // 4),
        ("5-6 players", None, 600000, False, "p56", None, n // 4), ("7-9 players", None, 700000, False, "p789", None, n // 8), ("DI d=1", False, 800000, False, "d1", 1, n // 4),
        ("10 players", None, 900000, False, "p10", None, n // 16)]
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, ext, base, d3, force, dov, nn in fams:
    if only and name != only: continue
    bad = []; consulted = 0; noise = 0; it = 0; fails = 0; corr = 0
    for seed in range(max(1, nn)):
        rng = np.random.default_rng(base + seed)
        if force == "p56": model, p = F.P56_FAMILIES[seed % 6]; g, o, x, tag = F._random_pair(alg, orc, rng, ext=(model == F.BIC or bool(seed % 2)), force=(model, p), force_d3=False, arb="x")
        elif force == "p789": model, p = F.P789_FAMILIES[seed % 9]; g, o, x, tag = F._random_pair(alg, orc, rng, ext=(model == F.BIC or bool(seed % 2)), force=(model, p), force_d3=False, arb="x")
        elif force == "p10": model = (F.DI, F.UNI, F.BIC)[seed % 3]; g, o, x, tag = F._random_pair(alg, orc, rng, ext=(model == F.BIC or bool((seed // 3) % 2)), force=(model, 10), force_d3=False, arb="x")
        elif force == "d1": g, o, x, tag = F._random_pair(alg, orc, rng, ext=False, force=(F.DI, 1 + seed % 4), force_d3=False, arb="x", d_override=1)
        else: g, o, x, tag = F._random_pair(alg, orc, rng, ext, d3=d3, force=force, arb="x")
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                F._compare_solve(g, o, tag, x=x)
        except AssertionError as e:
            bad.append((base + seed, str(e)[:200] + " ... " + str(e)[-260:].replace("\n", " ")))
        consulted += "arbiter consulted" in buf.getvalue(); noise += "status differs" in buf.getvalue()
        s = g.get_stats(); it += int(s["newton_iters"].sum()); fails += int(s["ls_failures"].sum()); corr += int(s["refinements"].sum())
    print("%-13s cases %d outside the rule %d, arbiter consulted for trajectories %d, diverged-game status rule %d, Newton iterations %d, failed line searches %d, correction solves %d, %.0f s"
          % (name, max(1, nn), len(bad), consulted, noise, it, fails, corr, time.time() - t0), flush=True)
    for b in bad[:40]: print("   ", b, flush=True)
