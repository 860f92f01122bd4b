"""Distances of full solves from the oracle with the refinement gate off / at its defaults / forced (tol = 0): random `_pair`
problems of tests/test_gpu_parity.py and IBR.  usage: python tests/probes/refine_forced_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
import algames_jl_amd as alg
from test_gpu_parity import DI, UNI, _pair
FAM = [(DI, 3, 2, 12), (DI, 2, 3, 8), (UNI, 3, 2, 10), (UNI, 4, 2, 8), (DI, 5, 2, 6), (DI, 3, 3, 6)]
SET = {"off": (0, 0.0, 0.0), "default": (2, 2.0 ** -34, 1.6e5), "forced": (2, 0.0, 0.0)}
for case in FAM:
    for mode in ("newton", "ibr"):
        if mode == "ibr" and case[1] > 4: continue
        out = []
        for name, st in SET.items():
            g, o = _pair(alg, orc, *case, B=5)
            g.set_refinement(*st)
            if mode == "newton":
                sg, so = g.newton_solve(init=True, game_id0=3), o.newton_solve(init=True, game_id0=3)
            else:
                sg, so = g.ibr_newton_solve(6, init=True, game_id0=1), o.ibr_newton_solve(6, init=True, game_id0=1)
            same = all(np.array_equal(sg[f], so[f]) for f in ("status", "outer_iters", "newton_iters", "ls_failures"))
            out.append("%s: %.2e%s ref %d/%d conv %d" % (name, np.abs(g.get_traj(0) - o.get_traj(0)).max(), "" if same else " COUNTS", int(sg["refinements"].sum()), int(sg["newton_iters"].sum()), int(sg["converged"].sum())))
        print(case, mode, " | ".join(out), flush=True)
