"""CPU-only study of the C5 hard solves (round 4): the oracle runs the C5 closed loop; solves with many Newton iterations are replayed
step-wise, and at every iterate the structured elimination (NumPy emulation, riccati_proto) is compared with the pivoted LU (oracle) and
the long-double arbiter: forward error of the direction, normwise backward error, candidate gate statistics, effect of one refinement."""
import sys, io, contextlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests/probes')
import numpy as np
import oracle as orc
import algames_jl_amd as alg
with contextlib.redirect_stdout(io.StringIO()):
    from riccati_proto import structured_direction
from refine_proto import bwd, rows_u

T = int(sys.argv[1]) if len(sys.argv) > 1 else 14
HARD = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ids = np.arange(128, 192)
po = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
bo = po.batch
saved = []
for t in range(T):
    if t == 1:
        po.opts.shift, po.opts.dual_reset = 1, False; po._sync_options()
    z = bo.get_traj(0); lam, mu = bo.get_con_duals()
    gid = po.game_id0 + t * 1000003
    so = bo.newton_solve(init=True, game_id0=gid)
    hard = np.nonzero(so["newton_iters"] >= HARD)[0]
    print("step", t, "iters max", so["newton_iters"].max(), "hard games", hard.tolist(), flush=True)
    for g in hard: saved.append((t, int(g), gid + int(g), z[g].copy(), lam[g].copy(), mu[g].copy(), int(so["newton_iters"][g])))
    bo.mpc_advance()

def one(ids1):
    p1 = alg.scenarios.make_problem("C5", ids1, backend=orc.lib()); return p1
rows = []
for (t, g, gid, z, lam, mu, nit) in saved[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]:
    p1 = alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib()); b = p1.batch
    px = alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib("x")); x = px.batch
    for q in (p1, px):
        if t >= 1: q.opts.shift, q.opts.dual_reset = 1, False; q._sync_options()
        q.batch.set_x0(z[None, :b.n].copy()); q.batch.set_traj(z[None], 0); q.batch.set_con_duals(lam[None], mu[None])
    o = p1.opts
    b.init_traj(game_id0=gid, use_shift=True); b.rollout()
    if t == 0: b.reset_con()
    ur = rows_u(b)
    delta = 0.0; it = 0; done = False
    for k in range(1, o.outer_iter + 1):
        ls_count = 0
        for l in range(1, o.inner_iter + 1):
            reg = o.reg_0 * l ** 4
            zz = b.get_traj(0); la, m_ = b.get_con_duals()
            x.set_traj(zz, 0); x.set_con_duals(la, m_)
            J = x.residual_jacobian(reg)[0]; r = x.residual()[0][0]
            d0 = b.newton_direction(reg)[0][0]; dx = x.newton_direction(reg)[0][0]
            d1 = structured_direction(b, J, r)
            r1 = J @ d1 + r
            mask = np.ones(len(r1), bool); mask[ur] = False
            d2 = d1 + structured_direction(b, J, np.where(mask, 0.0, r1))
            sc = np.abs(dx).max()
            fe = lambda d: np.abs(d - dx).max() / sc
            rho = r1[ur]; scu = np.abs(J[ur]) @ np.abs(d1) + np.abs(r[ur])
            print(f"t{t} g{g} k{k} l{l} mu_max {m_.max():.0e} | fwd err LU {fe(d0):.1e} struct {fe(d1):.1e} refined {fe(d2):.1e} | bwd LU {bwd(J, d0, r):.1e} struct {bwd(J, d1, r):.1e} refined {bwd(J, d2, r):.1e}"
                  f" | |rho|max {np.abs(rho).max():.1e} nonu {np.abs(r1[mask]).max():.1e} omega_u {(np.abs(rho) / scu).max():.1e} |d| {sc:.1e} |r| {np.abs(r).max():.1e}", flush=True)
            info = b.newton_step(k, l, delta)
            delta = float(info["delta"][0]); it += 1
            if info["status"][0] != 0: done = True; break
            ls_count = ls_count + 1 if info["ls_failed"][0] else 0
            if ls_count >= 1 or info["control_flow"][0] == 1: break
        if done: break
        rec = info["rec"][0]
        conv = rec["dyn_vio"] < o.ϵ_dyn and rec["con_vio"] < o.ϵ_con and rec["sta_vio"] < o.ϵ_sta and rec["opt_vio"] < o.ϵ_opt
        if k == o.outer_iter or conv: break
        b.dual_penalty_update()
    print(f"== t{t} g{g}: fused iters {nit}, replay iters {it}")
