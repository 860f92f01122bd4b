"""Parity of the 3-D half of SURVEY.md 8(f) rank 3 -- add_spherical_collision_avoidance!, Wall3DConstraint, CylinderConstraint on a
DoubleIntegratorGame with d = 3 (the EXT instantiation with 3-D position blocks, Cfg::PD = 3), mixed with the planar
ingredients (collision cost, state bounds, walls, circles act on x, y as in the reference) -- HIP path through the C ABI
against the CPU oracle on the same seeded inputs.  Same tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DI = 0
ALL = ("cost", "sph", "ctl", "sb", "wall", "circ", "wall3", "cyl")
S2 = np.sqrt(0.5)


def _pair(alg, orc, N, B, seed=0, ingredients=ALL, dt=0.1):
    p = 2
    g = alg.Batch(alg.hip_lib(), DI, p, N, dt, B, d=3)
    o = orc.OracleBatch(DI, p, N, dt, B, d=3)
    rng = np.random.default_rng(seed)
    Q, R = 1 + rng.random((B, p, 6)), 0.5 + rng.random((B, p, 3))
    xf, uf = rng.random((B, p, 6)), rng.random((B, p, 3)) - 0.5
    x0 = rng.random((B, g.n))
    xmax = np.where(rng.random((p, g.n)) < 0.6, 0.3 + 0.5 * rng.random((p, g.n)), np.inf)
    xmin = np.minimum(np.where(rng.random((p, g.n)) < 0.6, 0.5 * rng.random((p, g.n)) - 0.1, -np.inf), xmax)
    for b in (g, o):
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        if "cost" in ingredients:
            b.add_collision_cost(np.full(p, 3.0), 1.0 + np.arange(p))
        if "sph" in ingredients:
            b.add_spherical_collision_avoidance([0.35, 0.45])
        if "avoid" in ingredients:
            b.add_collision_avoidance([0.3, 0.4])
        if "ctl" in ingredients:
            umax = np.full(b.m, 0.6); umin = np.full(b.m, -0.4); umax[0] = np.inf
            b.add_control_bound(umax, umin)
        if "sb" in ingredients:
            b.add_state_bound(1, xmax[1], xmin[1])
        if "wall" in ingredients:
            b.add_wall_constraint([0.0, 0.2], [0.5, 1.0], [1.0, 0.9], [0.5, 0.1], [0.0, 0.6], [1.0, 0.8])
        if "circ" in ingredients:
            b.add_circle_constraint([0.5, 0.2], [0.5, 0.8], [0.3, 0.25])
        if "wall3" in ingredients:
            # a floor panel under the unit cube (normal up-and-sideways) and a slanted panel through it
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


SUBSETS = [ALL, ("sph",), ("wall3",), ("cyl",), ("cost", "avoid", "wall3"), ("cost", "sph", "ctl"), ("sb", "wall", "circ")]


@pytest.mark.parametrize("ingredients", SUBSETS)
def test_residual_record_jacobian_direction_parity(alg, orc, ingredients):
    g, o = _pair(alg, orc, 9, B=3, seed=len(ingredients), ingredients=ingredients)
    for which, reg in ((0, 0.0), (0, 1e-3)):
        rg, ng = g.residual(which, reg); ro, no = o.residual(which, reg)
        assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
        assert np.allclose(ng, no, rtol=1e-13, atol=0)
    a, b = g.record(), o.record()
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(a[f], b[f], rtol=1e-12, atol=1e-15), f
    if len(ingredients) > 1 or ingredients[0] != "wall3":
        assert np.all(a["sta_vio"] > 0)
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


def test_3d_terms_are_exercised(alg, orc):
    """Every 3-D constraint family is active somewhere in the seeded inputs (so the parity above is not vacuous) and the
    z-coupling entries of the position block are non-zero."""
    g, o = _pair(alg, orc, 9, B=3, seed=8)
    vals = o.kat_evaluate_con()
    K, p, n = o.N - 1, 2, o.n
    off = p * (p - 1) * K + 2 * o.m * K + p * 2 * n * K + p * 2 * K + p * 2 * K      # collision, control, state bound, 2 walls, 2 circles
    w3 = vals[:, off:off + p * K * 2]; cy = vals[:, off + p * K * 2:]
    assert cy.shape[1] == p * K * 3
    assert (w3 > 0).any() and (w3 < 0).any() and (cy > 0).any() and (cy == 0).any()
    assert (vals[:, :p * (p - 1) * K] > 0).any()
    J = o.residual_jacobian(0.0)[0]
    blk = J[np.ix_(np.arange(n), np.arange(n))]                  # Q^_1 at knot 2: rows opt_1,x_2 x columns x_2
    assert np.abs(blk[0, 4]) > 0 and np.abs(blk[2, 4]) > 0       # x1-z1 and y1-z1 couplings (indices i + a p)


@pytest.mark.parametrize("ingredients", [ALL, ("cost", "sph", "ctl", "wall3", "cyl")])
def test_inner_iteration_dual_update_and_ibr_parity(alg, orc, ingredients):
    g, o = _pair(alg, orc, 8, B=4, seed=5, ingredients=ingredients)
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
        b.set_options(rho_increase=7.0, rho_max=50.0, lambda_max=1.5, alpha_dual=0.7, alphax_dual=[0.5, 1.5] + [1.0] * 8)
    vg, vo = g.dual_penalty_update(), o.dual_penalty_update()
    fin = np.isfinite(vo)
    assert np.array_equal(np.isfinite(vg), fin) and np.abs(vg[fin] - vo[fin]).max() < 1e-13
    (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
    assert np.abs(lg - lo).max() < 1e-13 and np.array_equal(mg, mo)
    for b in (g, o):
        b.set_options(outer_iter=1, inner_iter=1, dual_reset=0, reg_0=1e-3)
    for player in range(2):
        sg, so = g.ibr_solve_player(player), o.ibr_solve_player(player)
        for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
            assert np.array_equal(sg[f], so[f]), (player, f, sg[f], so[f])
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-9, atol=1e-12), (player, f)
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())


def _drone_problem(alg, backend, X0):
    """Two point-mass drones swap places through a gap: a cylinder pillar between them, a ceiling panel, spherical collision
    avoidance, control bounds and a velocity-free state bound on player 1 (host-side mirror of the reference's builders)."""
    p, N, dt = 2, 20, 0.1
    model = alg.DoubleIntegratorGame(p=p, d=3)
    game_obj = alg.GameObjective([np.array([10.0, 10, 10, 1, 1, 1])] * p, [0.1 * np.ones(3)] * p,
                                 [np.array([1.0, 0.05, 0.5, 0, 0, 0]), np.array([-1.0, -0.05, 0.5, 0, 0, 0])], [np.zeros(3)] * p, N, model)
    alg.add_collision_cost(game_obj, 0.6 * np.ones(p), 2.0 * np.ones(p))
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_spherical_collision_avoidance(con, 0.15)
    alg.add_control_bound(con, 4 * np.ones(model.m), -4 * np.ones(model.m))
    alg.add_state_bound(con, 1, 3 * np.ones(model.n), -3 * np.ones(model.n))
    alg.add_wall_constraint(con, [alg.CylinderWall([0.0, 0.0, 0.0], "z", 2.0, 0.2)])
    alg.add_wall_constraint(con, [alg.Wall3D([-2.0, -2.0, 0.9], [2.0, -2.0, 0.9], [2.0, 2.0, 0.9], [0.0, 0.0, 1.0])])
    return alg.GameProblem(N, dt, X0, model, alg.Options(inner_print=False, outer_print=False), game_obj, con, backend=backend)


def test_3d_solve_parity_and_convergence(alg, orc):
    # state order: [x1 x2 | y1 y2 | z1 z2 | velocities]
    x0 = np.array([-1.0, 1.0, 0.02, -0.02, 0.5, 0.55, 0, 0, 0, 0, 0, 0])
    X0 = np.tile(x0, (10, 1)); X0[1:, :6] += 0.1 * (np.random.default_rng(4).random((9, 6)) - 0.5)
    pg, po = _drone_problem(alg, None, X0), _drone_problem(alg, orc.lib(), X0)
    alg.newton_solve(pg); alg.newton_solve(po)
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), (f, sg[f], so[f])
    assert sg["converged"].sum() >= 8 and np.all(sg["status"] == 0)
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-8, atol=1e-9), f
    Xg, Ug, Lg = pg.batch.split_traj(pg.batch.get_traj()); Xo, Uo, Lo = po.batch.split_traj(po.batch.get_traj())
    assert np.abs(Xg - Xo).max() <= 1e-8 and np.abs(Ug - Uo).max() <= 1e-8
    assert np.abs(Lg - Lo).max() <= 1e-6 * max(1.0, np.abs(Lo).max())
    (lg, mg), (lo, mo) = pg.batch.get_con_duals(), po.batch.get_con_duals()
    assert np.array_equal(mg, mo) and np.abs(lg - lo).max() <= 1e-6 * max(1.0, np.abs(lo).max())
    # the converged games respect the obstacles: outside the pillar, below the ceiling, apart from each other
    ok = sg["converged"] == 1
    X = Xg[ok]
    for i in range(2):
        assert np.all(X[:, 1:, i] ** 2 + X[:, 1:, 2 + i] ** 2 >= 0.2 ** 2 - 2e-3)
        assert np.all(X[:, 1:, 4 + i] <= 0.9 + 1e-3)
    d = np.sqrt(((X[:, 1:, [0, 2, 4]] - X[:, 1:, [1, 3, 5]]) ** 2).sum(-1))
    assert d.min() >= 0.3 - 2e-3


def test_3d_builders_reject_models_without_three_position_dimensions(alg):
    b = alg.Batch(alg.hip_lib(), DI, 2, 6, 0.1, 1)               # d = 2
    for call in (lambda: b.add_spherical_collision_avoidance(0.1),
                 lambda: b.add_wall3d_constraint([[0, 0, 0]], [[1, 0, 0]], [[1, 1, 0]], [[0, 0, 1]]),
                 lambda: b.add_cylinder_constraint([[0, 0, 0]], [2], [1.0], [0.1])):
        with pytest.raises(alg.AlgamesError):
            call()
    b3 = alg.Batch(alg.hip_lib(), DI, 2, 6, 0.1, 1, d=3)
    with pytest.raises(alg.AlgamesError):
        b3.add_cylinder_constraint([[0, 0, 0]], [3], [1.0], [0.1])


@pytest.mark.timeout(120)
def test_no_kernel_writes_outside_its_buffers_3d(alg):
    import ctypes
    g, _ = None, None
    g = alg.Batch(alg.hip_lib(), DI, 2, 30, 0.1, 5, d=3)
    rng = np.random.default_rng(3)
    g.set_x0(rng.normal(size=(5, g.n)) * 0.5)
    g.set_lqr(1 + rng.random((5, 2, 6)), 0.5 + rng.random((5, 2, 3)), rng.normal(size=(5, 2, 6)), np.zeros((5, 2, 3)))
    g.add_collision_cost(np.full(2, 2.0), np.ones(2)); g.add_spherical_collision_avoidance(np.full(2, 0.2))
    g.add_control_bound(np.full(g.m, 2.0), np.full(g.m, -2.0))
    g.add_wall3d_constraint([[-2.0, -2, 1]], [[2.0, -2, 1]], [[2.0, 2, 1]], [[0.0, 0, 1]])
    g.add_cylinder_constraint([[0.0, 0, 0]], [2], [2.0], [0.2])
    g.add_wall_constraint([0.0], [-2.0], [1.0], [-2.0], [0.0], [-1.0]); g.add_circle_constraint([3.0], [3.0], [0.5])
    g.add_state_bound(0, np.full(g.n, 50.0), np.full(g.n, -50.0))
    g.set_options(outer_iter=3, inner_iter=4)
    g.newton_solve(init=True, game_id0=11)
    g.residual(); g.residual_jacobian(1e-3); g.newton_direction(1e-3); g.record()
    g.newton_step(1, 1); g.dual_penalty_update(); g.rollout(0)
    for player in range(2):
        g.ibr_solve_player(player)
    g.ibr_newton_solve(init=True, game_id0=3, ibr_iter=2, ordering=[0, 1], delta_min=1e-9)
    g.mpc_totals(reset=True); g.mpc_solve(3, 5, record_states=True)
    fn = g.lib.debug_check_guards
    assert fn(g.h) == 0


def test_3d_receding_horizon_and_ibr_solve_parity(alg, orc):
    """The fused receding-horizon loop (k_mpc_loop) and the iterated-best-response solve of the 3-D instantiation."""
    x0 = np.array([-1.0, 1.0, 0.02, -0.02, 0.5, 0.55, 0, 0, 0, 0, 0, 0])
    X0 = np.tile(x0, (6, 1)); X0[1:, :6] += 0.1 * (np.random.default_rng(6).random((5, 6)) - 0.5)
    pg, po = _drone_problem(alg, None, X0), _drone_problem(alg, orc.lib(), X0)
    ig, cg, sg = alg.mpc_solve(pg, 5, record_states=True)
    io, co, so = alg.mpc_solve(po, 5, record_states=True)
    assert np.array_equal(ig, io) and np.array_equal(cg, co) and np.abs(sg - so).max() < 1e-7
    assert np.abs(sg[-1] - sg[0]).max() > 0.05
    pg, po = _drone_problem(alg, None, X0), _drone_problem(alg, orc.lib(), X0)
    ibr = alg.IBROptions(); ibr.ibr_iter = 3
    alg.ibr_newton_solve(pg, ibr_opts=ibr); alg.ibr_newton_solve(po, ibr_opts=ibr)
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "newton_iters", "records", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), (f, sg[f], so[f])
    zg, zo = pg.batch.get_traj(), po.batch.get_traj()
    assert np.abs(zg - zo).max() <= 1e-7 * max(1.0, np.abs(zo).max())


@pytest.mark.parametrize("model,p,N", [(DI, 2, 7), (3, 2, 5)])
def test_per_player_wall3d_and_cylinder_parity(alg, orc, model, p, N):
    """add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) / (game_con, i, walls::Vector{CylinderWall})
    (constraints_methods.jl:208-247, 256-299): DIFFERENT 3-D wall / cylinder sets per player with one shared entry each, HIP path vs
    oracle on a DoubleIntegrator (d = 3) and a Quadrotor game: residual, Jacobian, direction, an inner iteration, the dual / penalty
    update and a short full solve; the rows of a player an entry does not constrain stay inert."""
    B = 3
    g = alg.Batch(alg.hip_lib(), model, p, N, 0.1, B, d=3)
    o = orc.OracleBatch(model, p, N, 0.1, B, d=3)
    rng = np.random.default_rng(33)
    ni = g.n // p
    Q, R = 1 + rng.random((B, p, ni)), 0.5 + rng.random((B, p, g.mi))
    xf, uf = rng.random((B, p, ni)), rng.random((B, p, g.mi)) - 0.5
    x0 = rng.random((B, g.n))
    if model == 3:
        x0[:, 3 * p:] *= 0.1                                   # small attitudes / velocities: a flyable start
        uf = 1.2 + 0.1 * uf
    wa = ([[0.0, 0.0, 0.5]], [[1.0, 0.0, 0.5]], [[1.0, 1.0, 0.5]], [[0.0, 0.6, 0.8]])
    wb = ([[0.0, 0.0, 0.1]], [[1.0, 0.2, 0.3]], [[0.8, 1.0, 0.9]], [[S2, 0.0, -S2]])
    both = tuple(a + b_ for a, b_ in zip(wa, wb))
    for b in (g, o):
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        b.add_spherical_collision_avoidance(0.3 + 0.1 * np.arange(p))
        b.add_wall3d_constraint_player(0, *both)                                       # player 0: both panels
        b.add_wall3d_constraint_player(p - 1, *wb)                                     # last player: the slanted one only (shared entry)
        b.add_cylinder_constraint_player(p - 1, [[0.5, 0.5, 0.0], [0.0, 0.4, 0.6]], [2, 0], [1.5, 2.0], [0.45, 0.5])
        b.add_cylinder_constraint_player(0, [[0.0, 0.4, 0.6], [0.3, -0.2, 0.3]], [0, 1], [2.0, 0.9], [0.5, 0.35])   # shares the x-axis cylinder
    assert g.con_len == o.con_len
    z = rng.random((B, g.traj_len)); z[:, :g.n] = x0
    if model == 3:
        X, U, L = g.split_traj(z); X[:, :, 3 * p:] *= 0.1; U[:] = 1.2 + 0.1 * U; z = g.join_traj(X, U, L)
    lam, mu = rng.random((B, g.con_len)), 1.0 + 2.0 * rng.random((B, g.con_len))
    lam[rng.random((B, g.con_len)) < 0.3] = 0.0
    for b in (g, o):
        b.set_traj(z); b.set_con_duals(lam, mu)
    rg, ng = g.residual(0, 0.0); ro, no = o.residual(0, 0.0)
    assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max()) and np.allclose(ng, no, rtol=1e-13, atol=0)
    Jg, Jo = g.residual_jacobian(1e-3), o.residual_jacobian(1e-3)
    assert np.abs(Jg - Jo).max() <= 1e-12 * np.abs(Jo).max()
    dg, sg = g.newton_direction(1e-3); do, so = o.newton_direction(1e-3)
    assert np.all(sg == 0) and np.all(so == 0)
    assert (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9
    ig, io = g.newton_step(1, 1), o.newton_step(1, 1)
    assert np.array_equal(ig["ls_j"], io["ls_j"]) and np.array_equal(ig["alpha"], io["alpha"])
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(ig["rec"][f], io["rec"][f], rtol=1e-9, atol=1e-14), f
    vg, vo = g.dual_penalty_update(), o.dual_penalty_update()
    assert np.abs(vg - vo).max() <= 1e-9 * max(1.0, np.abs(vo).max())
    # inert rows: 3-D wall 0 (the floor panel) does not constrain the last player, cylinder 0 (z axis) not player 0
    K = N - 1
    nw3, ncy = 2, 3
    base = g.con_len - p * K * (nw3 + ncy)
    w3 = vg[:, base:base + p * K * nw3].reshape(B, p, K, nw3); cy = vg[:, base + p * K * nw3:].reshape(B, p, K, ncy)
    assert np.all(w3[:, p - 1, :, 0] == 0.0) and np.all(cy[:, 0, :, 0] == 0.0) and np.all(cy[:, p - 1, :, 2] == 0.0)
    (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
    assert np.array_equal(mg, mo) and np.abs(lg - lo).max() <= 1e-9 * max(1.0, np.abs(lo).max())
    for b in (g, o):
        b.set_options(outer_iter=3, inner_iter=5)
    sg, so = g.newton_solve(init=True, game_id0=5), o.newton_solve(init=True, game_id0=5)
    for f in ("status", "outer_iters", "newton_iters", "records", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), f
    assert np.abs(g.get_traj() - o.get_traj()).max() <= 1e-7 * max(1.0, np.abs(o.get_traj()).max())
