"""NumPy prototype of the refinement of the structured Newton direction (round 4): one correction solve J e = -(J d + res) with the
same elimination on a heavily penalised KKT system, against the pivoted LU of the same system.  The u-rows carry the whole residual
(the forward / costate sweeps satisfy the dyn / opt-x rows by construction)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests/probes')
import numpy as np
import oracle as orc
from riccati_proto import structured_direction

def bwd(J, d, r):
    return np.abs(J @ d + r).max() / (np.abs(J).sum(1).max() * np.abs(d).max() + np.abs(r).max())

def rows_u(b):
    n, m, p, N, mi = b.n, b.m, b.p, b.N, b.mi
    idx = []
    for i in range(p):
        for k in range(N - 1):
            s = i * (N - 1) * (n + mi) + k * (n + mi) + n
            idx += list(range(s, s + mi))
    return np.array(idx)

rng = np.random.default_rng(1)
for model, p, N, mu in [(1, 3, 30, 1e3), (1, 3, 30, 1e5), (1, 3, 30, 1e7), (0, 3, 40, 1e7), (1, 4, 20, 1e7)]:
    b = orc.OracleBatch(model, p, N, 0.1, 1)
    ni = b.n // p
    b.set_lqr(np.ones((p, ni)), 0.5 * np.ones((p, b.mi)), rng.random((p, ni)), np.zeros((p, b.mi)))
    b.set_x0(0.3 * rng.random(b.n))
    b.add_collision_avoidance(np.full(p, 0.4))
    b.add_control_bound(np.full(b.m, 0.5), np.full(b.m, -0.5))
    b.set_traj(0.3 * rng.random((1, b.traj_len)))
    mus = np.full((1, b.con_len), mu); mus[0, ::3] = 1.0
    b.set_con_duals(rng.random((1, b.con_len)) * (rng.random((1, b.con_len)) < 0.5), mus)
    reg = 1e-3
    d0, st = b.newton_direction(reg)
    J = b.residual_jacobian(reg)[0]; res = b.residual()[0][0]
    d1 = structured_direction(b, J, res)
    r1 = J @ d1 + res
    ur = rows_u(b); mask = np.ones(len(r1), bool); mask[ur] = False
    e = structured_direction(b, J, np.where(mask, 0.0, r1))
    d2 = d1 + e
    efull = structured_direction(b, J, r1)
    d3 = d1 + efull
    sc = np.abs(d0).max()
    print(f"model {model} p {p} mu {mu:.0e} cond {np.linalg.cond(J):.1e} | bwd LU {bwd(J, d0[0], res):.1e} struct {bwd(J, d1, res):.1e} refined(u rows) {bwd(J, d2, res):.1e} refined(all rows) {bwd(J, d3, res):.1e}"
          f" | err vs LU: struct {np.abs(d1 - d0[0]).max() / sc:.1e} refined {np.abs(d2 - d0[0]).max() / sc:.1e} | res share of non-u rows {np.abs(r1[mask]).max():.1e} u rows {np.abs(r1[ur]).max():.1e}")
