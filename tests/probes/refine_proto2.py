"""Refinement prototype on the quadrotor seeds of test_direction_backward_error_against_the_arbiter (CPU only: the structured
elimination emulated in NumPy on the oracle's Jacobian)."""
import sys, types
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests/probes'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle as orc
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    from riccati_proto import structured_direction
import test_gpu_fuzz as F
from refine_proto import bwd, rows_u

fake = types.SimpleNamespace(Batch=lambda lib, model, p, N, dt, B, d=2: orc.OracleBatch(model, p, N, dt, B, d=d), hip_lib=lambda: None)
for seed, fam in F.BWD_SEEDS[6:]:
    if fam[0] != 3: continue
    g, o, x, tag = F._random_pair(fake, orc, np.random.default_rng(seed), arb="x", ext=bool(seed % 2), force=fam)
    reg = 1e-6
    for b in (o, x): b.init_traj(game_id0=7); b.rollout()
    for it in range(2):
        Jx, rx = x.residual_jacobian(reg), x.residual(reg=reg)[0]
        do = o.newton_direction(reg)[0]
        for game in range(o.B):
            J, r = Jx[game], rx[game]
            d1 = structured_direction(o, J, r)
            r1 = J @ d1 + r
            ur = rows_u(o); mask = np.ones(len(r1), bool); mask[ur] = False
            d2 = d1 + structured_direction(o, J, np.where(mask, 0.0, r1))
            r2 = J @ d2 + r
            d3 = d2 + structured_direction(o, J, np.where(mask, 0.0, r2))
            print(seed, tag[:3], it, game, f"LU {bwd(J, do[game], r):.1e} struct {bwd(J, d1, r):.1e} refined {bwd(J, d2, r):.1e} twice {bwd(J, d3, r):.1e}  | res non-u {np.abs(r1[mask]).max():.1e} u {np.abs(r1[ur]).max():.1e}")
        for b in (o, x): b.update_traj(0.5)
        o.set_traj(x.get_traj())
