"""Gate statistic of the refinement: row-wise (Oettli-Prager) backward error of the u-rows, omega = max_c |rho_c| / (|J_c| |d| + |res_c|),
for the structured direction, the refined one and the pivoted LU -- quadrotor seeds and ordinary seeds."""
import sys, types
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests/probes'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle as orc
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    from riccati_proto import structured_direction
import test_gpu_fuzz as F
from refine_proto import bwd, rows_u

def omega(J, d, r, rows):
    rho = J[rows] @ d + r[rows]
    sc = np.abs(J[rows]) @ np.abs(d) + np.abs(r[rows])
    return (np.abs(rho) / np.maximum(sc, 1e-300)).max()

fake = types.SimpleNamespace(Batch=lambda lib, model, p, N, dt, B, d=2: orc.OracleBatch(model, p, N, dt, B, d=d), hip_lib=lambda: None)
for seed, fam in F.BWD_SEEDS:
    kw = dict(ext=False) if fam is None else dict(ext=bool(seed % 2), force=fam)
    g, o, x, tag = F._random_pair(fake, orc, np.random.default_rng(seed), arb="x", **kw)
    reg = 1e-6
    for b in (o, x): b.init_traj(game_id0=7); b.rollout()
    Jx, rx = x.residual_jacobian(reg), x.residual(reg=reg)[0]
    do = o.newton_direction(reg)[0]
    for game in range(o.B):
        J, r = Jx[game], rx[game]
        d1 = structured_direction(o, J, r)
        r1 = J @ d1 + r
        ur = rows_u(o); mask = np.ones(len(r1), bool); mask[ur] = False
        d2 = d1 + structured_direction(o, J, np.where(mask, 0.0, r1))
        print(seed, tag[:3], game, f"bwd: LU {bwd(J, do[game], r):.1e} struct {bwd(J, d1, r):.1e} refined {bwd(J, d2, r):.1e} | omega_u: LU {omega(J, do[game], r, ur):.1e} struct {omega(J, d1, r, ur):.1e} refined {omega(J, d2, r, ur):.1e}")
