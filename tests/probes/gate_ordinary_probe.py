"""Round 4: the candidate gate statistics of the refinement on ORDINARY solves (C2 / C3 scenario games, HIP path as the master): what
would a threshold cost in false triggers?   usage: python tests/probes/gate_ordinary_probe.py CFG first_game n_games [refine_max]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "probes"))
import numpy as np
import oracle as orc
import algames_jl_amd as alg
import importlib.util
cfg, g0, ng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); RMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 0
def bwd(J, d, r): return np.abs(J @ d + r).max() / (np.abs(J).sum(1).max() * np.abs(d).max() + np.abs(r).max())
def urows(b):
    n, m, p, N, mi = b.n, b.m, b.p, b.N, b.mi
    return np.array([i * (N - 1) * (n + mi) + k * (n + mi) + n + j for i in range(p) for k in range(N - 1) for j in range(mi)])
def gate_stats(J, d, r, ur):
    rho = np.abs(J[ur] @ d + r[ur]); rowsc = np.abs(J) @ np.abs(d)
    return (rho.max() / (np.abs(J).sum(1).max() * np.abs(d).max()), (rho / (rowsc[ur] + np.abs(r[ur]) + 1e-300)).max(), rho.max() / rowsc.max(), rho.max() / np.abs(r).max())
for g in range(g0, g0 + ng):
    ids = np.arange(g, g + 1)
    probs = [alg.scenarios.make_problem(cfg, ids), alg.scenarios.make_problem(cfg, ids, backend=orc.lib()), alg.scenarios.make_problem(cfg, ids, backend=orc.lib("x"))]
    h, o, x = (q.batch for q in probs)
    h.set_waves_per_game(1); h.set_refinement(*((RMAX,) if RMAX > 0 else (1, 1e300)))
    for q in probs:
        q.batch.init_traj(game_id0=q.game_id0, use_shift=True); q.batch.rollout(); q.batch.reset_con()
    op = probs[0].opts; ur = urows(h)
    delta = 0.0; done = False
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
            gh, go = gate_stats(J, dh, r, ur), gate_stats(J, do_, r, ur)
            info = h.newton_step(k, l, delta)
            print(f"{cfg} g{g} k{k} l{l} mu {m_.max():.0e} | dir fwd LU {fe(do_):.1e} HIP {fe(dh):.1e} bwd LU {bwd(J, do_, r):.1e} HIP {bwd(J, dh, r):.1e} | DEV rho {dev[0]:.1e} omega {dev[1]:.1e} mix {dev[2]:.1e} | gate HIP norm {gh[0]:.1e} row {gh[1]:.1e} mix {gh[2]:.1e} rel {gh[3]:.1e} ; LU norm {go[0]:.1e} row {go[1]:.1e} mix {go[2]:.1e} rel {go[3]:.1e} | res {info['rec']['res'][0]:.2e}", flush=True)
            delta = float(info["delta"][0])
            if info["status"][0] != 0: done = True; break
            ls_count = ls_count + 1 if info["ls_failed"][0] else 0
            if ls_count >= 1 or info["control_flow"][0] == 1: break
        if done: break
        rec = info["rec"][0]
        conv = rec["dyn_vio"] < op.ϵ_dyn and rec["con_vio"] < op.ϵ_con and rec["sta_vio"] < op.ϵ_sta and rec["opt_vio"] < op.ϵ_opt
        if k == op.outer_iter or conv: break
        h.dual_penalty_update()
