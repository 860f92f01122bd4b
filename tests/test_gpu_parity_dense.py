"""Parity of the instantiations that take the dense Newton direction (Cfg::DENSE, newton_direction_dense) beyond the base
QuadrotorGame cases of tests/test_gpu_parity_quad.py:
  * DoubleIntegratorGame d = 3 with p = 1, 3, 4 (n = 6, 18, 24 -- outside the single 16 x 16 tile), base and extended sets;
  * QuadrotorGame p = 1..4 with the extended ingredient set: state bounds, walls, circles (planar, on px[i]) and the 3-D half on
    pz[i][1:3] -- add_spherical_collision_avoidance! (constraints_methods.jl:45-81), Wall3DConstraint (wall_constraint.jl:127-236),
    CylinderConstraint (cylinder_constraint.jl:35-127).
HIP path through the C ABI against the CPU oracle on the same seeded inputs; tolerances of tests/test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DI, QUAD = 0, 3
ALL = ("cost", "sph", "ctl", "sb", "wall", "circ", "wall3", "cyl")
BASE = ("cost", "avoid", "ctl")
S2 = np.sqrt(0.5)


def _pair(alg, orc, model, p, N, B, seed=0, ingredients=ALL, dt=0.1):
    g = alg.Batch(alg.hip_lib(), model, p, N, dt, B, d=3)
    o = orc.OracleBatch(model, p, N, dt, B, d=3)
    ni, mi = g.n // p, g.mi
    rng = np.random.default_rng(seed)
    Q, R = 1 + rng.random((B, p, ni)), 0.5 + rng.random((B, p, mi))
    xf, uf = rng.random((B, p, ni)), rng.random((B, p, mi)) - 0.5
    x0 = rng.random((B, g.n))
    xmax = np.where(rng.random((p, g.n)) < 0.6, 0.3 + 0.5 * rng.random((p, g.n)), np.inf)
    xmin = np.minimum(np.where(rng.random((p, g.n)) < 0.6, 0.5 * rng.random((p, g.n)) - 0.1, -np.inf), xmax)
    for b in (g, o):
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        if "cost" in ingredients and p > 1:
            b.add_collision_cost(np.full(p, 3.0), 1.0 + np.arange(p))
        if "sph" in ingredients and p > 1:
            b.add_spherical_collision_avoidance(0.35 + 0.05 * np.arange(p))
        if "avoid" in ingredients and p > 1:
            b.add_collision_avoidance(0.3 + 0.05 * np.arange(p))
        if "ctl" in ingredients:
            umax = np.full(b.m, 0.6); umin = np.full(b.m, -0.4); umax[0] = np.inf
            b.add_control_bound(umax, umin)
        if "sb" in ingredients:
            b.add_state_bound(p - 1, xmax[p - 1], xmin[p - 1])
            if p > 2:
                b.add_state_bound(0, xmax[0], xmin[0])
        if "wall" in ingredients:
            b.add_wall_constraint([0.0, 0.2], [0.5, 1.0], [1.0, 0.9], [0.5, 0.1], [0.0, 0.6], [1.0, 0.8])
        if "circ" in ingredients:
            b.add_circle_constraint([0.5, 0.2], [0.5, 0.8], [0.3, 0.25])
        if "wall3" in ingredients:
            b.add_wall3d_constraint([[0.0, 0.0, 0.5], [0.0, 0.0, 0.1]], [[1.0, 0.0, 0.5], [1.0, 0.2, 0.3]],
                                    [[1.0, 1.0, 0.5], [0.8, 1.0, 0.9]], [[0.0, 0.6, 0.8], [S2, 0.0, -S2]])
        if "cyl" in ingredients:
            b.add_cylinder_constraint([[0.5, 0.5, 0.0], [0.0, 0.4, 0.6], [0.3, -0.2, 0.3]], [2, 0, 1], [1.5, 2.0, 0.9], [0.45, 0.5, 0.35])
    assert g.con_len == o.con_len
    z = rng.random((B, g.traj_len)); z[:, :g.n] = x0
    lam, mu = rng.random((B, g.con_len)), 1.0 + 2.0 * rng.random((B, g.con_len))
    lam[rng.random((B, g.con_len)) < 0.3] = 0.0
    for b in (g, o):
        b.set_traj(z); b.set_con_duals(lam, mu)
    return g, o


CASES = ([(DI, p, N, ing) for p, N in ((1, 9), (3, 7), (4, 6)) for ing in (BASE, ALL)]
         + [(QUAD, p, N, ALL) for p, N in ((1, 7), (2, 7), (3, 5), (4, 4))]
         + [(QUAD, 2, 6, ("sph", "wall3")), (QUAD, 3, 5, ("sb", "wall", "circ", "cyl"))])
IDS = [f"{'DI3' if c[0] == DI else 'QUAD'}-p{c[1]}-{'+'.join(c[3])}" for c in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_residual_record_jacobian_direction_parity(alg, orc, case):
    model, p, N, ing = case
    g, o = _pair(alg, orc, model, p, N, B=2, seed=len(ing) + p, ingredients=ing)
    for which, reg in ((0, 0.0), (0, 1e-3)):
        rg, ng = g.residual(which, reg); ro, no = o.residual(which, reg)
        assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
        assert np.allclose(ng, no, rtol=1e-13, atol=0)
    a, b = g.record(), o.record()
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(a[f], b[f], rtol=1e-12, atol=1e-15), f
    for reg in (0.0, 1e-3 * 3 ** 4):
        Jg, Jo = g.residual_jacobian(reg), o.residual_jacobian(reg)
        assert np.abs(Jg - Jo).max() <= 1e-12 * np.abs(Jo).max()
    for reg in (1e-3, 1e-7 * 2 ** 4):
        dg, sg = g.newton_direction(reg); do, so = o.newton_direction(reg)
        assert np.all(sg == 0) and np.all(so == 0)
        assert (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9
        J = o.residual_jacobian(reg); res = o.residual()[0]
        lin = np.einsum("brc,bc->br", J, dg) + res
        assert np.abs(lin).max() <= 1e-8 * max(1.0, np.abs(res).max())


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_inner_iteration_dual_update_and_solve_parity(alg, orc, case):
    model, p, N, ing = case
    g, o = _pair(alg, orc, model, p, N, B=3, seed=5 + p, ingredients=ing)
    for l in (1, 2):
        ig, io = g.newton_step(1, l), o.newton_step(1, l)
        for f in ("status", "control_flow", "ls_j", "ls_failed"):
            assert np.array_equal(ig[f], io[f]), f
        assert np.array_equal(ig["alpha"], io["alpha"])
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(ig["rec"][f], io["rec"][f], rtol=1e-9, atol=1e-14), f
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())
    for b in (g, o):
        b.set_options(rho_increase=7.0, rho_max=50.0, lambda_max=1.5, alpha_dual=0.7, alphax_dual=[0.5, 1.5, 0.8, 1.2] + [1.0] * 6)
    vg, vo = g.dual_penalty_update(), o.dual_penalty_update()
    fin = np.isfinite(vo)
    # (the iterates of the two inner iterations above agree to 1e-9 relative; the constraint values inherit that)
    assert np.array_equal(np.isfinite(vg), fin) and np.abs(vg[fin] - vo[fin]).max() < 1e-10
    (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
    assert np.abs(lg - lo).max() < 1e-10 and np.array_equal(mg, mo)
    # a short fused solve from the current iterate (two outer iterations): identical control flow, same iterate
    for b in (g, o):
        b.set_options(outer_iter=2, inner_iter=3, dual_reset=0, rho_increase=10.0, rho_max=1e7, lambda_max=1e7, alpha_dual=1.0, alphax_dual=[1.0] * 10)
    if p == 3:
        g.set_waves_per_game(1)            # p = 3: one wavefront per game; p = 2, 4: the automatic team of four (p = 1 has no team kernel)
    assert g.get_waves_per_game() == (4 if p in (2, 4) else 1)
    sg, so = g.newton_solve(init=False), o.newton_solve(init=False)
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), (f, sg[f], so[f])
    zg, zo = g.get_traj(0), o.get_traj(0)
    assert np.abs(zg - zo).max() <= 1e-7 * max(1.0, np.abs(zo).max())
    assert g.lib.debug_check_guards(g.h) == 0


@pytest.mark.parametrize("model,p", [(DI, 3), (QUAD, 2)])
def test_ibr_and_mpc_loop_parity(alg, orc, model, p):
    g, o = _pair(alg, orc, model, p, 6, B=3, seed=13, ingredients=ALL)
    for b in (g, o):
        b.set_options(outer_iter=1, inner_iter=1, dual_reset=0, reg_0=1e-3)
    for player in range(p):
        sg, so = g.ibr_solve_player(player), o.ibr_solve_player(player)
        for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
            assert np.array_equal(sg[f], so[f]), (player, f, sg[f], so[f])
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-9, atol=1e-12), (player, f)
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())
    for b in (g, o):
        b.set_options(outer_iter=2, inner_iter=4, dual_reset=1, reg_0=1e-3)
    g.mpc_totals(reset=True); o.mpc_totals(reset=True)
    sg, so = g.mpc_solve(3, game_id0=7, record_states=True), o.mpc_solve(3, game_id0=7, record_states=True)
    (ig, cg), (io, co) = g.mpc_totals(), o.mpc_totals()
    assert np.array_equal(ig, io) and np.array_equal(cg, co) and ig.min() >= 3
    zo = o.get_traj(0)
    assert np.abs(sg - so).max() < 1e-7 and np.abs(g.get_traj(0) - zo).max() <= 1e-8 * max(1.0, np.abs(zo).max())
