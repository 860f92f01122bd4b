"""The fuzz tail of round 4 under different gate settings: the extended-bicycle and two-quadrotor seeds that
tests/probes/fuzz_long_r4.py left outside the acceptance rule, solved with (a) the library's defaults, (b) the gate without the penalty
relaxation, (c) two forced corrections on every direction, (d) refinement off; per seed: do the discrete histories agree with the
oracle's, |hip - oracle|, |hip - arbiter|, |oracle - arbiter| (games with the arbiter's history), correction solves.
usage (GPU box): python tests/probes/r05_fuzz_tail.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
orc.build()
import test_gpu_fuzz as F
SEEDS = [(200036, True, False, None), (200041, True, False, None), (200085, True, False, None), (200280, True, False, None), (200290, True, False, None),
         (200302, True, False, None), (400011, True, True, (3, 2)), (400040, True, True, (3, 2)), (400051, True, True, (3, 2)), (400059, True, True, (3, 2)), (400074, True, True, (3, 2))]
SETTINGS = [("default", None), ("no relaxation", (2, 2.0 ** -34, 0.0)), ("tol 2^-44, no relaxation", (2, 2.0 ** -44, 0.0)), ("forced x2", (2, 0.0, 0.0)), ("off", (0, 2.0 ** -34, 1.6e5))]
F_HIST = ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures")
for seed, ext, d3, force in SEEDS:
    ref = None
    for name, st in SETTINGS:
        g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), ext, d3=d3, force=force, arb="x")
        if st is not None: g.set_refinement(*st)
        sg = g.newton_solve(init=True, game_id0=7)
        if ref is None:
            so, sx = o.newton_solve(init=True, game_id0=7), x.newton_solve(init=True, game_id0=7)
            zo, zx = o.get_traj(0), x.get_traj(0); ref = (so, sx, zo, zx)
        so, sx, zo, zx = ref
        zg = g.get_traj(0)
        hist_ok = all(np.array_equal(sg[f], so[f]) for f in F_HIST)
        ok = so["status"] == 0
        same = np.all([sx[f] == so[f] for f in ("status", "outer_iters", "newton_iters", "ls_failures")], axis=0) & ok & np.all([sg[f] == so[f] for f in ("status", "outer_iters", "newton_iters", "ls_failures")], axis=0)
        scale = max(1.0, np.abs(zo[ok]).max()) if ok.any() else 1.0
        err = np.abs(zg[ok] - zo[ok]).max(initial=0.0)
        eg, eo = np.abs(zg[same] - zx[same]).max(initial=0.0), np.abs(zo[same] - zx[same]).max(initial=0.0)
        inside = hist_ok and (err <= 1e-8 * scale or (same[ok].all() and eg <= 4 * eo + 1e-8 * scale))
        print("%d %-26s model %d p %d N %d | history %s | hip-orc %.1e hip-x %.1e orc-x %.1e scale %.1f | games same %d/%d | corrections %d iters %d | %s"
              % (seed, name, tag[0], tag[1], tag[2], "same" if hist_ok else "DIFF", err, eg, eo, scale, same.sum(), len(same), int(sg["refinements"].sum()), int(sg["newton_iters"].sum()), "inside" if inside else "OUTSIDE"), flush=True)
