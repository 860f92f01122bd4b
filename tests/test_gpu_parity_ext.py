"""Parity of the extended ingredient set of examples/intro_example.jl -- BicycleGame,
StateBoundConstraint, WallConstraint, CircleConstraint -- HIP path (EXT kernel instantiations, through the C ABI)
against the CPU oracle on the same seeded inputs.  Same tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DI, UNI, BIC = 0, 1, 2
CASES = [  # (model, p, N)
    (DI, 1, 6), (DI, 3, 10), (DI, 4, 6), (UNI, 1, 7), (UNI, 2, 12), (UNI, 3, 9), (UNI, 4, 6),
    (BIC, 1, 8), (BIC, 2, 12), (BIC, 3, 10), (BIC, 4, 6),
    # five and six players (n = 20 / 24: dense Newton direction, algames_p5.hip / algames_p6.hip)
    (DI, 5, 5), (DI, 6, 4), (UNI, 5, 5), (UNI, 6, 4), (BIC, 5, 5), (BIC, 6, 4),
    # seven to ten players (round 6: algames_p7.hip ... algames_p10.hip; ten = the reference's cap, TIGHT LDS layout of the dense direction)
    (DI, 7, 4), (UNI, 8, 4), (BIC, 9, 4), (BIC, 7, 5), (DI, 9, 3), (UNI, 9, 4), (DI, 10, 4), (UNI, 10, 3), (BIC, 10, 4),
]
ALL = ("cost", "avoid", "ctl", "sb", "wall", "circ")


def _pair(alg, orc, model, p, N, B, seed=0, ingredients=ALL, dt=0.1, lf=0.07, lr=0.04):
    g = alg.Batch(alg.hip_lib(), model, p, N, dt, B)
    o = orc.OracleBatch(model, p, N, dt, B)
    rng = np.random.default_rng(seed)
    ni = g.n // p
    Q, R = 1 + rng.random((B, p, ni)), 0.5 + rng.random((B, p, g.mi))
    xf, uf = rng.random((B, p, ni)), rng.random((B, p, g.mi)) - 0.5
    x0 = rng.random((B, g.n))
    xmax = np.where(rng.random((p, g.n)) < 0.6, 0.3 + 0.5 * rng.random((p, g.n)), np.inf)
    xmin = np.where(rng.random((p, g.n)) < 0.6, 0.5 * rng.random((p, g.n)) - 0.1, -np.inf)
    xmin = np.minimum(xmin, xmax)
    for b in (g, o):
        if model == BIC:
            b.set_bicycle(lf, lr)
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        if "cost" in ingredients and p > 1:
            b.add_collision_cost(np.full(p, 3.0), 1.0 + np.arange(p))
        if "avoid" in ingredients and p > 1:
            b.add_collision_avoidance(0.3 + 0.1 * np.arange(p))
        if "ctl" in ingredients:
            umax = np.full(b.m, 0.6); umin = np.full(b.m, -0.4); umax[0] = np.inf
            b.add_control_bound(umax, umin)
        if "sb" in ingredients:
            for i in range(0, p, 2):                       # players 0, 2 only
                b.add_state_bound(i, xmax[i], xmin[i])
        if "wall" in ingredients:
            # two walls crossing the unit square where the random positions live
            b.add_wall_constraint([0.0, 0.2], [0.5, 1.0], [1.0, 0.9], [0.5, 0.1], [0.0, 0.6], [1.0, 0.8])
        if "circ" in ingredients:
            b.add_circle_constraint([0.5, 0.2, 0.9], [0.5, 0.8, 0.1], [0.3, 0.25, 0.2])
    assert g.con_len == o.con_len
    z = rng.random((B, g.traj_len)); z[:, :g.n] = x0
    lam, mu = rng.random((B, g.con_len)), 1.0 + 2.0 * rng.random((B, g.con_len))
    lam[rng.random((B, g.con_len)) < 0.3] = 0.0
    for b in (g, o):
        b.set_traj(z); b.set_con_duals(lam, mu)
    return g, o


@pytest.mark.parametrize("case", CASES)
def test_residual_and_record_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=4)
    for which, reg in ((0, 0.0), (0, 1e-3)):
        rg, ng = g.residual(which, reg); ro, no = o.residual(which, reg)
        assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
        assert np.allclose(ng, no, rtol=1e-13, atol=0)
    zt = np.random.default_rng(9).random((g.B, g.traj_len)); zt[:, :g.n] = g.get_traj()[:, :g.n]
    g.set_traj(zt, 1); o.set_traj(zt, 1)
    rg, ng = g.residual(1, 0.37); ro, no = o.residual(1, 0.37)
    assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
    a, b = g.record(), o.record()
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(a[f], b[f], rtol=1e-12, atol=1e-15), f
    assert np.all(a["sta_vio"] > 0)                          # the extended constraints are violated somewhere


@pytest.mark.parametrize("case", CASES)
def test_jacobian_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=2)
    for reg in (0.0, 1e-3 * 3 ** 4):
        Jg, Jo = g.residual_jacobian(reg), o.residual_jacobian(reg)
        assert np.abs(Jg - Jo).max() <= 1e-12 * np.abs(Jo).max()


@pytest.mark.parametrize("case", CASES)
def test_newton_direction_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=3)
    for reg in (1e-3, 1e-7 * 2 ** 4):
        dg, sg = g.newton_direction(reg); do, so = o.newton_direction(reg)
        assert np.all(sg == 0) and np.all(so == 0)
        scale = np.abs(do).max(axis=1, keepdims=True)
        assert (np.abs(dg - do) / scale).max() < 1e-9
        J = o.residual_jacobian(reg); res = o.residual()[0]
        lin = np.einsum("brc,bc->br", J, dg) + res
        assert np.abs(lin).max() <= 1e-8 * max(1.0, np.abs(res).max())


@pytest.mark.parametrize("ingredients", [(), ("sb",), ("wall",), ("circ",), ("cost", "avoid", "ctl")])
@pytest.mark.parametrize("case", [CASES[1], CASES[5], CASES[9]])
def test_ingredient_subsets_parity(alg, orc, case, ingredients):
    if not ingredients and case[0] != BIC:
        pytest.skip("covered by tests/test_gpu_parity.py (base instantiation)")
    g, o = _pair(alg, orc, *case, B=2, seed=2, ingredients=ingredients)
    rg, _ = g.residual(); ro, _ = o.residual()
    assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
    dg, sg = g.newton_direction(1e-3); do, so = o.newton_direction(1e-3)
    assert np.all(sg == 0) and (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9


@pytest.mark.parametrize("case", [CASES[1], CASES[4], CASES[9], CASES[10]])
def test_inner_iteration_and_line_search_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=4, seed=5)
    for l in (1, 2):
        ig, io = g.newton_step(1, l), o.newton_step(1, l)
        for f in ("status", "control_flow", "ls_j", "ls_failed"):
            assert np.array_equal(ig[f], io[f]), f
        assert np.array_equal(ig["alpha"], io["alpha"])
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(ig["rec"][f], io["rec"][f], rtol=1e-9, atol=1e-14), f
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())


def test_dual_penalty_update_and_reset_parity(alg, orc):
    for case in (CASES[2], CASES[5], CASES[9]):
        g, o = _pair(alg, orc, *case, B=3, seed=7)
        for b in (g, o):
            b.set_options(rho_increase=7.0, rho_max=50.0, lambda_max=1.5, alpha_dual=0.7, alphax_dual=[0.5, 1.0, 1.5, 2.0] + [1.0] * 6)
        vg, vo = g.dual_penalty_update(), o.dual_penalty_update()
        fin = np.isfinite(vo)
        assert np.array_equal(np.isfinite(vg), fin) and np.abs(vg[fin] - vo[fin]).max() < 1e-14
        (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
        assert np.abs(lg - lo).max() < 1e-14 and np.array_equal(mg, mo)
        g.reset_con(); o.reset_con()
        (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
        assert np.all(lg == 0) and np.array_equal(mg, mo) and np.all(mg == 1.0)


def test_bicycle_init_rollout_and_mpc_advance_parity(alg, orc):
    g, o = _pair(alg, orc, BIC, 3, 10, B=4, seed=11)
    g.init_traj(game_id0=1000); o.init_traj(game_id0=1000)
    Xg, Ug, Lg = g.split_traj(g.get_traj(0)); Xo, Uo, Lo = o.split_traj(o.get_traj(0))
    assert np.array_equal(Ug, Uo) and np.array_equal(Lg, Lo)
    assert np.abs(Xg - Xo).max() < 1e-14                                  # RK3 rollout of the bicycle
    z = np.random.default_rng(3).random((g.B, g.traj_len))
    for b in (g, o):
        b.set_traj(z); b.rollout(0)
    assert np.abs(g.get_traj(0) - o.get_traj(0)).max() < 1e-13
    for b in (g, o):
        b.set_traj(z); b.mpc_advance()                                    # x0 <- RK2(x_1, u_1)
    assert np.abs(g.get_x0() - o.get_x0()).max() < 1e-14


@pytest.mark.parametrize("case", [CASES[2], CASES[5], CASES[9]])
def test_ibr_best_response_step_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=3, seed=13)
    for b in (g, o):
        b.set_options(outer_iter=1, inner_iter=1, dual_reset=0, reg_0=1e-3)
    for player in range(case[1]):
        sg, so = g.ibr_solve_player(player), o.ibr_solve_player(player)
        for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
            assert np.array_equal(sg[f], so[f]), (player, f, sg[f], so[f])
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-9, atol=1e-12), (player, f)
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())


def _intro_problem(alg, backend, x0):
    # examples/intro_example.jl:10-74 (same construction as tests/test_oracle_kat.py::_intro_problem)
    p, N, dt = 3, 20, 0.1
    model = alg.BicycleGame(p=p)
    game_obj = alg.GameObjective([10.0 * np.ones(4)] * p, [0.1 * np.ones(2)] * p,
                                 [np.array([2, 0.4, 0, 0.0]), np.array([2, 0.0, 0, 0.0]), np.array([3, -0.4, 0, 0.0])],
                                 [np.zeros(2)] * p, N, model)
    alg.add_collision_cost(game_obj, 1.0 * np.ones(p), 5.0 * np.ones(p))
    game_con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(game_con, 0.08)
    alg.add_control_bound(game_con, 5 * np.ones(model.m), -5 * np.ones(model.m))
    alg.add_state_bound(game_con, 1, 5 * np.ones(model.n), -5 * np.ones(model.n))
    alg.add_wall_constraint(game_con, [alg.Wall([0.0, -0.4], [1.0, -0.4], [0.0, -1.0])])
    alg.add_circle_constraint(game_con, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], [0.1, 0.2, 0.3])
    return alg.GameProblem(N, dt, x0, model, alg.Options(inner_print=False, outer_print=False), game_obj, game_con, backend=backend)


def test_intro_example_solve_parity(alg, orc):
    """examples/intro_example.jl (3-player bicycle, every constraint type) as a batch: game 0 is the example's x0, the
    others perturb the start positions."""
    x0 = np.array([0.1, 0.0, 0.5, -0.4, 0.0, 0.7, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    X0 = np.tile(x0, (12, 1)); X0[1:, :6] += 0.05 * (np.random.default_rng(4).random((11, 6)) - 0.5)
    pg, po = _intro_problem(alg, None, X0), _intro_problem(alg, orc.lib(), X0)
    alg.newton_solve(pg); alg.newton_solve(po)
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), (f, sg[f], so[f])
    assert sg["converged"][0] == 1 and sg["status"][0] == 0
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-8, atol=1e-9), f
    Xg, Ug, Lg = pg.batch.split_traj(pg.batch.get_traj()); Xo, Uo, Lo = po.batch.split_traj(po.batch.get_traj())
    assert np.abs(Xg - Xo).max() <= 1e-8 and np.abs(Ug - Uo).max() <= 1e-8
    assert np.abs(Lg - Lo).max() <= 1e-6 * max(1.0, np.abs(Lo).max())
    (lg, mg), (lo, mo) = pg.batch.get_con_duals(), po.batch.get_con_duals()
    assert np.array_equal(mg, mo) and np.abs(lg - lo).max() <= 1e-6 * max(1.0, np.abs(lo).max())


def test_reference_constrained_unicycle_with_circles_on_gpu(alg):
    """test/problem/solver_methods.jl:132-182 in full (collision avoidance + control bounds + circle constraints) through
    the product path, with the reference's thresholds."""
    N, dt = 20, 0.1
    model = alg.UnicycleGame(p=2)
    opts = alg.Options(inner_print=False, outer_print=False)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 7, 20, 25, 1e-7, 1e-10, 1e-10
    obj = alg.GameObjective([np.ones(4)] * 2, [0.5 * np.ones(2)] * 2, [np.zeros(4)] * 2, [-np.ones(2)] * 2, N, model)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(con, 0.05)
    alg.add_control_bound(con, np.ones(4), -np.ones(4))
    alg.add_circle_constraint(con, [1.50, 0.2, 0.3], [1.25, 0.2, 0.3], [0.2, 0.2, 0.3])
    prob = alg.GameProblem(N, dt, [1.0, 2.0, 1.1, 2.0, 0.0, 0.0, 0.9, 0.9], model, opts, obj, con)
    alg.newton_solve(prob)
    last = prob.stats.summary["last"][0]
    res = alg.residual(prob)
    assert np.abs(res).sum() / res.shape[1] < 1e-3
    for f in ("dyn_vio", "sta_vio", "con_vio", "opt_vio"):
        assert last[f] < 1e-3, (f, last[f])


def test_extended_constraints_unsupported_configuration_fails_loudly(alg):
    with pytest.raises(alg.AlgamesError):
        alg.Batch(alg.hip_lib(), DI, 11, 6, 0.1, 1, d=2)        # eleven players: beyond the reference's cap (options.jl:68)
    with pytest.raises(alg.AlgamesError):
        alg.Batch(alg.hip_lib(), DI, 5, 6, 0.1, 1, d=3)         # DoubleIntegrator d = 3 with five players (n = 30): no kernel instantiation
    b = alg.Batch(alg.hip_lib(), DI, 2, 6, 0.1, 1, d=1)         # DoubleIntegrator d = 1: base constraint set only
    with pytest.raises(alg.AlgamesError):
        b.add_state_bound(0, np.ones(b.n), -np.ones(b.n))


def _guards_ok(batch):
    """Diagnostic entry of the ABI (alg_debug_check_guards): every device buffer is followed by a 4 KiB guard zone;
    returns the number of buffers whose guard was overwritten."""
    import ctypes
    fn = batch.lib.debug_check_guards
    return fn(batch.h)


# every compiled kernel instantiation (ALG_CFGS_BASE / ALG_CFGS_EXT of algames_kernels.hpp): (model, p, N, extended constraints)
ALL_INSTANTIATIONS = ([(DI, p, N, False) for p, N in ((1, 5), (2, 13), (3, 40), (4, 9))] + [(UNI, p, N, False) for p, N in ((1, 6), (2, 12), (3, 30), (4, 50))]
                      + [(DI, p, N, True) for p, N in ((1, 7), (2, 9), (3, 12), (4, 6))] + [(UNI, p, N, True) for p, N in ((1, 9), (2, 8), (3, 11), (4, 7))]
                      + [(BIC, p, N, True) for p, N in ((1, 8), (2, 7), (3, 20), (4, 11))]
                      + [(DI, 5, 7, False), (DI, 6, 5, True), (UNI, 5, 6, True), (UNI, 6, 5, False), (BIC, 5, 6, True), (BIC, 6, 5, True)]
                      # (round 5: the other four five- / six-player instantiations, so that every kernel of ALG_CFGS_P56 is reached by a test)
                      + [(DI, 5, 5, True), (DI, 6, 6, False), (UNI, 5, 5, False), (UNI, 6, 4, True)]
                      # (round 6: seven to ten players, every instantiation of ALG_CFGS_P789)
                      + [(m_, p_, 4, e_) for p_ in (7, 8, 9, 10) for m_, e_ in ((DI, False), (DI, True), (UNI, False), (UNI, True), (BIC, True))])


@pytest.mark.timeout(120)
@pytest.mark.parametrize("case", ALL_INSTANTIATIONS)
def test_no_kernel_writes_outside_its_buffers(alg, case):
    """Runs every kernel family of every instantiation (solve, step-wise entry points, IBR, receding-horizon loop, dense Jacobian)
    and checks the guard zones behind all device buffers; the timeout turns a kernel that never returns into a failure."""
    model, p, N, ext = case
    g = alg.Batch(alg.hip_lib(), model, p, N, 0.1, 5)
    rng = np.random.default_rng(3)
    ni = g.n // p
    g.set_x0(rng.normal(size=(5, g.n)) * 0.5)
    g.set_lqr(1 + rng.random((5, p, ni)), 0.5 + rng.random((5, p, g.mi)), rng.normal(size=(5, p, ni)), np.zeros((5, p, g.mi)))
    if p > 1:
        g.add_collision_cost(np.full(p, 2.0), np.ones(p)); g.add_collision_avoidance(np.full(p, 0.2))
    g.add_control_bound(np.full(g.m, 2.0), np.full(g.m, -2.0))
    if ext:
        g.add_wall_constraint([0.0], [-2.0], [1.0], [-2.0], [0.0], [-1.0])
        g.add_circle_constraint([3.0], [3.0], [0.5])
        g.add_state_bound(0, np.full(g.n, 50.0), np.full(g.n, -50.0))
    g.set_options(outer_iter=3, inner_iter=4)
    g.newton_solve(init=True, game_id0=11)
    g.residual(); g.residual_jacobian(1e-3); g.newton_direction(1e-3); g.record()
    g.newton_step(1, 1); g.dual_penalty_update(); g.rollout(0)
    for player in range(p):
        g.ibr_solve_player(player)
    g.ibr_newton_solve(init=True, game_id0=3, ibr_iter=2, ordering=list(range(p)), delta_min=1e-9)
    g.mpc_totals(reset=True); g.mpc_solve(3, 5, record_states=True)
    assert _guards_ok(g) == 0


@pytest.mark.parametrize("model,p,N", [(UNI, 3, 9), (BIC, 2, 8), (DI, 4, 6)])
def test_per_player_wall_and_circle_parity(alg, orc, model, p, N):
    """add_wall_constraint!(game_con, i, walls) / add_circle_constraint!(game_con, i, ...) (constraints_methods.jl:121-139, 161-187):
    different sets for different players (one shared entry), HIP path vs oracle: residual, Jacobian, direction, an inner iteration,
    the dual / penalty update and a short full solve."""
    B = 3
    g = alg.Batch(alg.hip_lib(), model, p, N, 0.1, B)
    o = orc.OracleBatch(model, p, N, 0.1, B)
    rng = np.random.default_rng(21)
    ni = g.n // p
    Q, R = 1 + rng.random((B, p, ni)), 0.5 + rng.random((B, p, g.mi))
    xf, uf = rng.random((B, p, ni)), rng.random((B, p, g.mi)) - 0.5
    x0 = rng.random((B, g.n))
    for b in (g, o):
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        if p > 1:
            b.add_collision_avoidance(0.3 + 0.1 * np.arange(p))
        b.add_wall_constraint_player(0, [0.0, 0.2], [0.5, 1.0], [1.0, 0.9], [0.5, 0.1], [0.0, 0.6], [1.0, 0.8])
        b.add_wall_constraint_player(p - 1, [0.2, -1.0], [1.0, 0.3], [0.9, 2.0], [0.1, 0.3], [0.6, 0.0], [0.8, 1.0])   # shares the second wall
        b.add_circle_constraint_player(p - 1, [0.5], [0.5], [0.3])
        if p > 2:
            b.add_circle_constraint_player(1, [0.2, 0.5], [0.8, 0.5], [0.25, 0.3])
    assert g.con_len == o.con_len
    z = rng.random((B, g.traj_len)); z[:, :g.n] = x0
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
    (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
    assert np.array_equal(mg, mo) and np.abs(lg - lo).max() <= 1e-9 * max(1.0, np.abs(lo).max())
    for b in (g, o):
        b.set_options(outer_iter=3, inner_iter=5)
    sg, so = g.newton_solve(init=True, game_id0=5), o.newton_solve(init=True, game_id0=5)
    for f in ("status", "outer_iters", "newton_iters", "records", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), f
    assert np.abs(g.get_traj() - o.get_traj()).max() <= 1e-7 * max(1.0, np.abs(o.get_traj()).max())
    assert _guards_ok(g) == 0
