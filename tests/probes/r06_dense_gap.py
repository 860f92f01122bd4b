"""Where does the dense direction lose its digits?  Replays a quadrotor fuzz seed step by step on the HIP path and on the oracle (from the
oracle's iterate each time, so that the two see the same state), and at every state compares the HIP direction with the arbiter's and
splits the residual J d + res (arbiter's Jacobian, long double) by row kind: opt-x | opt-u | dyn.
usage: python tests/probes/r06_dense_gap.py SEED [GAME]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
np.set_printoptions(linewidth=220, precision=3)
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
g, o, x, tag = F._random_pair(alg, orc, rng, True, d3=True, force=(3, 2), arb="x")
print(tag)
B, n, m, p, N, mi = 3, g.n, g.m, g.p, g.N, g.mi
S = n * p * (N - 1) + m * (N - 1) + n * (N - 1)
nx = p * (N - 1) * (n + mi)
rows_x = np.array([i * (N - 1) * (n + mi) + k * (n + mi) + a for i in range(p) for k in range(N - 1) for a in range(n)])
rows_u = np.array([i * (N - 1) * (n + mi) + k * (n + mi) + n + j for i in range(p) for k in range(N - 1) for j in range(mi)])
rows_d = np.arange(nx, S)
opts = tag[-1]
for b in (g, o, x): b.init_traj(game_id0=7); b.rollout(0)
print("init diff", np.abs(g.get_traj(0) - o.get_traj(0)).max())
it = 0
for k in range(1, opts["outer_iter"] + 1):
    for l in range(1, opts["inner_iter"] + 1):
        reg = opts["reg_0"] * l ** 4 if opts["regularize"] else 0.0
        z = o.get_traj(0); lam, mu = o.get_con_duals()
        for b in (g, x): b.set_traj(z); b.set_con_duals(lam, mu)
        J = x.residual_jacobian(reg).astype(np.longdouble); res = x.residual(0, 0.0)[0].astype(np.longdouble)
        dx, sx = x.newton_direction(reg); do, so = o.newton_direction(reg)
        out = []
        for name, rs in (("gate off", 0), ("default", None), ("2 forced", (2, 0.0)), ("6, tol 1e-30", (6, 1e-30)), ("8 forced", (8, 0.0))):
            if rs == 0: g.set_refinement(0)
            elif rs is None: g.set_refinement(6, 2.0 ** -34, 1.6e5)
            else: g.set_refinement(rs[0], rs[1], 1.6e5)
            dg, sg = g.newton_direction(reg)
            if rs is None: print("      gate of the first solve [max |rho|, omega, row scale]:", g.get_direction_gate().tolist())
            sc = np.abs(dx).max(axis=1)
            r = np.einsum("brc,bc->br", J, dg.astype(np.longdouble)) + res
            out.append((name, sg, np.abs(dg - dx).max(axis=1) / sc, [np.abs(r[:, rr]).max(axis=1).astype(float) for rr in (rows_x, rows_u, rows_d)]))
        ro = np.einsum("brc,bc->br", J, do.astype(np.longdouble)) + res
        print("k %d l %d  |z| %.2e  |d| %s  oracle err %s" % (k, l, np.abs(z).max(), np.abs(dx).max(axis=1), np.abs(do - dx).max(axis=1) / np.abs(dx).max(axis=1)))
        print("      oracle residual rows x|u|d:", [np.abs(ro[:, rr]).max(axis=1).astype(float) for rr in (rows_x, rows_u, rows_d)])
        for name, sg, e, rr in out: print("   %-9s status %s  err vs arbiter %s  residual rows x %s u %s d %s" % (name, sg, e, rr[0], rr[1], rr[2]))
        g.set_refinement(6, 2.0 ** -34, 1.6e5)
        # advance all three with the ORACLE's step (one inner iteration of the oracle), so that the next state is shared
        info = o.newton_step(k, l)
        if np.all(info["control_flow"] == 1): break
    o.dual_penalty_update() if hasattr(o, "dual_penalty_update") else None
