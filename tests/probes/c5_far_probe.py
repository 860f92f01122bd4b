"""Round 4: find the C5 lock-step solves where the HIP path is the far one (or its counts differ from the oracle's and the arbiter's) and
replay them iteration by iteration with the HIP path as the master: before every Newton iteration the oracle and the long-double
arbiter receive the HIP path's state; all three compute the direction (forward error against the arbiter) and take the step (distance
of the new iterates, line-search decisions).  Shows WHICH iteration and WHICH component loses the digits.
usage: python tests/probes/c5_far_probe.py [T] [max_replays] [refine_max]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
import algames_jl_amd as alg

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
MAXR = int(sys.argv[2]) if len(sys.argv) > 2 else 12
RMAX = int(sys.argv[3]) if len(sys.argv) > 3 else -1
RTOL = float(sys.argv[4]) if len(sys.argv) > 4 else None
ids = np.arange(128, 192)
pg = alg.scenarios.make_problem("C5", ids); po = alg.scenarios.make_problem("C5", ids, backend=orc.lib()); px = alg.scenarios.make_problem("C5", ids, backend=orc.lib("x"))
bg, bo, bx = pg.batch, po.batch, px.batch
bg.set_waves_per_game(1)
if RMAX >= 0 and hasattr(bg, "set_refinement"): bg.set_refinement(RMAX, RTOL)
print("refinement setting", bg.get_refinement())
CNT = ("status", "outer_iters", "newton_iters", "ls_failures", "converged", "records")
flagged = []; tot = dict(hip_far=0, orc_far=0, n_diff=0, hip_right=0, orc_right=0, neither=0, worst_eg=0.0, worst_eo=0.0, refinements=0)
t0 = time.time()
for t in range(T):
    if t == 1:
        for q in (pg, po, px): q.opts.shift, q.opts.dual_reset = 1, False; q._sync_options()
    z = bg.get_traj(0); lam, mu = bg.get_con_duals()
    for b_ in (bo, bx): b_.set_x0(z[:, :bg.n].copy()); b_.set_traj(z, 0); b_.set_con_duals(lam, mu)
    gid = pg.game_id0 + t * 1000003
    sg, so, sx = bg.newton_solve(init=True, game_id0=gid), bo.newton_solve(init=True, game_id0=gid), bx.newton_solve(init=True, game_id0=gid)
    if "refinements" in sg.dtype.names: tot["refinements"] += int(sg["refinements"].sum())
    same = np.all([sg[f] == so[f] for f in CNT], axis=0)
    gx = np.all([sg[f] == sx[f] for f in CNT], axis=0); ox = np.all([so[f] == sx[f] for f in CNT], axis=0)
    zx = bx.get_traj(0)
    eg, eo = np.abs(bg.get_traj(0) - zx).max(axis=1), np.abs(bo.get_traj(0) - zx).max(axis=1)
    all3 = gx & ox & (sx["converged"] == 1)
    far = all3 & (eg > 1e-8 + 100.0 * eo); ofar = all3 & (eo > 1e-8 + 100.0 * eg)
    tot["hip_far"] += int(far.sum()); tot["orc_far"] += int(ofar.sum()); tot["n_diff"] += int((~same).sum())
    tot["hip_right"] += int((gx & ~ox).sum()); tot["orc_right"] += int((ox & ~gx).sum()); tot["neither"] += int((~gx & ~ox).sum())
    tot["worst_eg"] = max(tot["worst_eg"], float(eg[all3].max(initial=0.0))); tot["worst_eo"] = max(tot["worst_eo"], float(eo[all3].max(initial=0.0)))
    for g in np.nonzero(far | ~same)[0]:
        flagged.append((t, int(g), gid + int(g), z[g].copy(), lam[g].copy(), mu[g].copy(), "far" if far[g] else "diff", int(sg["newton_iters"][g]), int(so["newton_iters"][g]), float(eg[g]), float(eo[g])))
        print("flag", t, int(g), flagged[-1][6], "iters hip/orc/x", int(sg["newton_iters"][g]), int(so["newton_iters"][g]), int(sx["newton_iters"][g]), "eg %.2e eo %.2e" % (eg[g], eo[g]), flush=True)
    bg.mpc_advance()
print("lock-step totals", tot, "time %.0f s" % (time.time() - t0), flush=True)

def bwd(J, d, r): return np.abs(J @ d + r).max() / (np.abs(J).sum(1).max() * np.abs(d).max() + np.abs(r).max())
def urows(b):
    n, m, p, N, mi = b.n, b.m, b.p, b.N, b.mi
    return np.array([i * (N - 1) * (n + mi) + k * (n + mi) + n + j for i in range(p) for k in range(N - 1) for j in range(mi)])
def gate_stats(J, d, r, ur):
    """candidate gate statistics of the opt-u rows: normwise, row-wise (Oettli-Prager), against the largest row of |J||d|, against |r|"""
    rho = np.abs(J[ur] @ d + r[ur]); rowsc = np.abs(J) @ np.abs(d)
    return (rho.max() / (np.abs(J).sum(1).max() * np.abs(d).max()), (rho / (rowsc[ur] + np.abs(r[ur]) + 1e-300)).max(), rho.max() / rowsc.max(), rho.max() / np.abs(r).max())
for (t, g, gid, z, lam, mu, kind, nh, no, egv, eov) in flagged[:MAXR]:
    probs = [alg.scenarios.make_problem("C5", ids[g:g + 1]), alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib()), alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib("x"))]
    h, o, x = (q.batch for q in probs)
    h.set_waves_per_game(1)
    if RMAX >= 0 and hasattr(h, "set_refinement"): h.set_refinement(*((RMAX, RTOL) if RMAX > 0 else (1, 1e300)))
    for q in probs:
        if t >= 1: q.opts.shift, q.opts.dual_reset = 1, False; q._sync_options()
        q.batch.set_x0(z[None, :h.n].copy()); q.batch.set_traj(z[None], 0); q.batch.set_con_duals(lam[None], mu[None])
        q.batch.init_traj(game_id0=gid, use_shift=True); q.batch.rollout()
        if t == 0: q.batch.reset_con()
    op = probs[0].opts; ur = urows(h)
    print(f"=== replay t{t} g{g} ({kind}; fused iters hip {nh} orc {no}; eg {egv:.2e} eo {eov:.2e})", flush=True)
    delta = 0.0; done = False; it = 0
    for k in range(1, op.outer_iter + 1):
        ls_count = 0
        for l in range(1, op.inner_iter + 1):
            reg = op.reg_0 * l ** 4
            zz = h.get_traj(0); la, m_ = h.get_con_duals()
            for q in (o, x): q.set_traj(zz, 0); q.set_con_duals(la, m_)
            J = x.residual_jacobian(reg)[0]; r = x.residual()[0][0]
            dh, do_, dx = h.newton_direction(reg)[0][0], o.newton_direction(reg)[0][0], x.newton_direction(reg)[0][0]
            dg = h.get_direction_gate()[0]; dev = (dg[0], dg[1], dg[0] / max(dg[2], 1e-300))
            sc = np.abs(dx).max(); fe = lambda d: np.abs(d - dx).max() / sc
            ih, io, ix = h.newton_step(k, l, delta), o.newton_step(k, l, delta), x.newton_step(k, l, delta)
            zh, zo, zxx = h.get_traj(0)[0], o.get_traj(0)[0], x.get_traj(0)[0]
            it += 1
            gh, go = gate_stats(J, dh, r, ur), gate_stats(J, do_, r, ur)
            print(f"  k{k} l{l} mu {m_.max():.0e} | dir fwd LU {fe(do_):.1e} HIP {fe(dh):.1e} bwd LU {bwd(J, do_, r):.1e} HIP {bwd(J, dh, r):.1e} | DEV rho {dev[0]:.1e} omega {dev[1]:.1e} mix {dev[2]:.1e} | gate HIP norm {gh[0]:.1e} row {gh[1]:.1e} mix {gh[2]:.1e} rel {gh[3]:.1e} ; LU norm {go[0]:.1e} row {go[1]:.1e} mix {go[2]:.1e} rel {go[3]:.1e} | step: ls_j h/o/x {int(ih['ls_j'][0])}/{int(io['ls_j'][0])}/{int(ix['ls_j'][0])}"
                  f" |zh-zx| {np.abs(zh - zxx).max():.1e} |zo-zx| {np.abs(zo - zxx).max():.1e} | res {ix['rec']['res'][0]:.2e} opt {ix['rec']['opt_vio'][0]:.2e} |d| {sc:.1e}", flush=True)
            info = ih
            delta = float(info["delta"][0])
            if info["status"][0] != 0: done = True; break
            ls_count = ls_count + 1 if info["ls_failed"][0] else 0
            if ls_count >= 1 or info["control_flow"][0] == 1: break
        if done: break
        rec = info["rec"][0]
        conv = rec["dyn_vio"] < op.ϵ_dyn and rec["con_vio"] < op.ϵ_con and rec["sta_vio"] < op.ϵ_sta and rec["opt_vio"] < op.ϵ_opt
        if k == op.outer_iter or conv: break
        for q in (h, o, x): q.dual_penalty_update()
    print(f"  replayed {it} iterations", flush=True)
