"""Round 4, the missing probe of VERDICT r3 item 1: accuracy of the HIP Newton direction exactly where the C5 lock-step run loses its
digits (long solves, penalties at their ceiling), on the quadrotor seeds and on the two FUZZ_TOL_LOOSE cases.  The double oracle drives
the solve step-wise; at every iterate the HIP handle and the long-double arbiter receive the oracle's state and all three compute the
Newton direction: forward error against the arbiter's direction and normwise backward error in the arbiter's Jacobian.
usage: python tests/probes/dir_accuracy_probe.py [T_mpc_steps] [min_iters] [max_solves]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
import algames_jl_amd as alg

def bwd(J, d, r):
    return np.abs(J @ d + r).max() / (np.abs(J).sum(1).max() * np.abs(d).max() + np.abs(r).max())

def c5_part(T, HARD, MAXS):
    ids = np.arange(128, 192)
    po = alg.scenarios.make_problem("C5", ids, backend=orc.lib()); bo = po.batch
    saved = []
    for t in range(T):
        if t == 1:
            po.opts.shift, po.opts.dual_reset = 1, False; po._sync_options()
        z = bo.get_traj(0); lam, mu = bo.get_con_duals()
        gid = po.game_id0 + t * 1000003
        so = bo.newton_solve(init=True, game_id0=gid)
        hard = np.nonzero(so["newton_iters"] >= HARD)[0]
        for g in hard[:2]: saved.append((t, int(g), gid + int(g), z[g].copy(), lam[g].copy(), mu[g].copy(), int(so["newton_iters"][g])))
        bo.mpc_advance()
    worst = 0.0
    for (t, g, gid, z, lam, mu, nit) in saved[:MAXS]:
        probs = [alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib()), alg.scenarios.make_problem("C5", ids[g:g + 1], backend=orc.lib("x")),
                 alg.scenarios.make_problem("C5", ids[g:g + 1])]
        b, x, h = (q.batch for q in probs)
        h.set_waves_per_game(1)
        for q in probs:
            if t >= 1: q.opts.shift, q.opts.dual_reset = 1, False; q._sync_options()
            q.batch.set_x0(z[None, :b.n].copy()); q.batch.set_traj(z[None], 0); q.batch.set_con_duals(lam[None], mu[None])
        o = probs[0].opts
        b.init_traj(game_id0=gid, use_shift=True); b.rollout()
        if t == 0: b.reset_con()
        delta = 0.0; done = False
        for k in range(1, o.outer_iter + 1):
            ls_count = 0
            for l in range(1, o.inner_iter + 1):
                reg = o.reg_0 * l ** 4
                zz = b.get_traj(0); la, m_ = b.get_con_duals()
                for q in (x, h): q.set_traj(zz, 0); q.set_con_duals(la, m_)
                J = x.residual_jacobian(reg)[0]; r = x.residual()[0][0]
                d0 = b.newton_direction(reg)[0][0]; dx = x.newton_direction(reg)[0][0]; dh = h.newton_direction(reg)[0][0]
                sc = np.abs(dx).max(); fe = lambda d: np.abs(d - dx).max() / sc
                worst = max(worst, fe(dh))
                print(f"C5 t{t} g{g} k{k} l{l} mu_max {m_.max():.0e} | fwd err LU {fe(d0):.1e} HIP {fe(dh):.1e} | bwd LU {bwd(J, d0, r):.1e} HIP {bwd(J, dh, r):.1e} x {bwd(J, dx, r):.1e} | |d| {sc:.1e} |r| {np.abs(r).max():.1e}", flush=True)
                info = b.newton_step(k, l, delta)
                delta = float(info["delta"][0])
                if info["status"][0] != 0: done = True; break
                ls_count = ls_count + 1 if info["ls_failed"][0] else 0
                if ls_count >= 1 or info["control_flow"][0] == 1: break
            if done: break
            rec = info["rec"][0]
            conv = rec["dyn_vio"] < o.ϵ_dyn and rec["con_vio"] < o.ϵ_con and rec["sta_vio"] < o.ϵ_sta and rec["opt_vio"] < o.ϵ_opt
            if k == o.outer_iter or conv: break
            b.dual_penalty_update()
    print("C5 worst HIP forward error", worst)

def fuzz_part():
    import test_gpu_fuzz as F
    for seed, fam in F.BWD_SEEDS:
        if fam is not None and fam[0] == 3:
            g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), arb="x", ext=bool(seed % 2), force=fam)
            reg = 1e-6
            for b in (g, o, x): b.init_traj(game_id0=7); b.rollout()
            Jx, rx = x.residual_jacobian(reg), x.residual(reg=reg)[0]
            dg, do, dx = g.newton_direction(reg)[0], o.newton_direction(reg)[0], x.newton_direction(reg)[0]
            for game in range(g.B):
                sc = np.abs(dx[game]).max()
                print(f"QUAD {seed} {tag[:3]} game {game} | fwd err LU {np.abs(do[game] - dx[game]).max() / sc:.1e} HIP {np.abs(dg[game] - dx[game]).max() / sc:.1e} | bwd LU {bwd(Jx[game], do[game], rx[game]):.1e} HIP {bwd(Jx[game], dg[game], rx[game]):.1e} x {bwd(Jx[game], dx[game], rx[game]):.1e}", flush=True)

if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    HARD = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    MAXS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    t0 = time.time()
    fuzz_part()
    print("fuzz part", time.time() - t0, "s", flush=True)
    c5_part(T, HARD, MAXS)
    print("total", time.time() - t0, "s")
