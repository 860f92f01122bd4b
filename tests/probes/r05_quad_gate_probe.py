"""What the dense direction's gate sees against what the arbiter test asks for (quadrotor): for the quadrotor seeds of
tests/test_gpu_fuzz.py::BWD_SEEDS and the Q2 bench scenario, the direction with 0 / 1 / 2 forced corrections: normwise backward error
in the arbiter's Jacobian (the test's bound is 1e-15), and the gate's own figures of the FIRST solve (max |rho|, row-wise omega, largest
row scale).  usage (GPU box): python tests/probes/r05_quad_gate_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import algames_jl_amd as alg
import oracle as orc
orc.build()
import test_gpu_fuzz as F

def bwd(J, r, d):
    return np.abs(J @ d + r).max() / (np.abs(J).sum(1).max() * np.abs(d).max() + np.abs(r).max())

rows = []
for seed, fam in F.BWD_SEEDS:
    if fam is None or fam[0] != 3:
        continue
    g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), ext=bool(seed % 2), force=fam, arb="x")
    reg = 1e-6
    for b in (g, o, x):
        b.init_traj(game_id0=7); b.rollout()
    for it in range(2):
        Jx, rx = x.residual_jacobian(reg), x.residual(reg=reg)[0]
        res = {}
        for steps in (0, 1, 2):
            g.set_refinement(steps, 0.0 if steps else None)          # tol 0: every allowed correction is made
            d = g.newton_direction(reg)[0]
            gate = g.get_direction_gate()
            res[steps] = [bwd(Jx[q], rx[q], d[q]) for q in range(g.B)]
            if steps == 1: g1 = gate.copy()
        for q in range(g.B):
            print("seed %d p=%d it %d game %d: bwd 0/1/2 corrections %.1e %.1e %.1e | first solve: rho/smax %.1e omega %.1e" %
                  (seed, fam[1], it, q, res[0][q], res[1][q], res[2][q], g1[q, 0] / max(g1[q, 2], 1e-300), g1[q, 1]))
            rows.append((res[0][q], res[1][q], res[2][q], g1[q, 0] / max(g1[q, 2], 1e-300), g1[q, 1]))
        for b in (g, o, x):
            b.update_traj(0.5)
        z = x.get_traj()
        for b in (g, o):
            b.set_traj(z)
        g.set_refinement(2, 2.0 ** -34)
a = np.array(rows)
print("n", len(a), "max bwd with 0/1/2 corrections: %.1e %.1e %.1e" % tuple(a[:, :3].max(0)))
for thr in (2.0 ** -50, 2.0 ** -51, 2.0 ** -52, 2.0 ** -53):
    need = a[:, 3] > thr
    print("gate rho/smax > %.1e: %d of %d would be corrected; worst bwd among the uncorrected %.1e" % (thr, need.sum(), len(a), a[~need, 0].max() if (~need).any() else 0.0))
