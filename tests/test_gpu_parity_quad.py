"""QuadrotorGame (src/dynamics/quadrotor.jl) on the HIP path against the CPU oracle: the Cfg::DENSE
instantiations (forward-mode RK2 Jacobian blocks per player, LDS-resident dense Newton direction with tiled f64 MFMA products).
Same tolerances as tests/test_gpu_parity.py.  The reference itself holds no solver test for this model (test/dynamics/quadrotor.jl
checks the index sets and allocation-freeness only): the oracle's restatement is pinned by tests/test_oracle_kat.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QUAD = 3
CASES = [(1, 7), (2, 8), (3, 6), (4, 5)]          # (p, N)


def _pair(alg, orc, p, N, B, seed=0, ingredients=("cost", "avoid", "ctl"), dt=0.1):
    g = alg.Batch(alg.hip_lib(), QUAD, p, N, dt, B, d=3)
    o = orc.OracleBatch(QUAD, p, N, dt, B, d=3)
    assert (g.n, g.m, g.mi) == (12 * p, 4 * p, 4) and g.traj_len == o.traj_len and g.con_len == o.con_len
    rng = np.random.default_rng(seed)
    Q, R = 1 + rng.random((B, p, 12)), 0.5 + rng.random((B, p, 4))
    xf, uf = rng.random((B, p, 12)), rng.random((B, p, 4)) - 0.5
    x0 = rng.random((B, g.n)) - 0.3
    z = rng.random((B, g.traj_len)) - 0.3; z[:, :g.n] = x0                 # negative rotor commands: the max(0, kf w) branch
    lam, mu = rng.random((B, g.con_len)), 1.0 + 2.0 * rng.random((B, g.con_len))
    lam[rng.random((B, g.con_len)) < 0.3] = 0.0
    for b in (g, o):
        b.set_x0(x0); b.set_lqr(Q, R, xf, uf)
        if "cost" in ingredients and p > 1:
            b.add_collision_cost(np.full(p, 3.0), 1.0 + np.arange(p))
        if "avoid" in ingredients and p > 1:
            b.add_collision_avoidance(0.3 + 0.1 * np.arange(p))
        if "ctl" in ingredients:
            umax = np.full(b.m, 0.6); umin = np.full(b.m, -0.4); umax[0] = np.inf
            b.add_control_bound(umax, umin)
        b.set_traj(z); b.set_con_duals(lam, mu)
    return g, o


@pytest.mark.parametrize("case", CASES)
def test_residual_and_record_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=3)
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


@pytest.mark.parametrize("case", CASES)
def test_jacobian_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=2)
    for reg in (0.0, 1e-3 * 3 ** 4):
        Jg, Jo = g.residual_jacobian(reg), o.residual_jacobian(reg)
        assert (Jg != 0).sum() > 0
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
        zd = g.get_traj(2)
        assert np.all(zd[:, :g.n] == 0) and np.array_equal(zd[:, g.n:], dg)


@pytest.mark.parametrize("case", CASES)
def test_line_search_and_inner_iteration_parity(alg, orc, case):
    g, o = _pair(alg, orc, *case, B=4, seed=3)
    reg = 1e-3
    do, _ = o.newton_direction(reg); g.newton_direction(reg)
    zd = np.zeros((g.B, g.traj_len)); zd[:, g.n:] = do
    g.set_traj(zd, 2); o.set_traj(zd, 2)
    rn = o.residual()[1]
    ag, jg = g.line_search(rn, reg); ao, jo = o.line_search(rn, reg)
    assert np.array_equal(jg, jo) and np.array_equal(ag, ao)
    g.update_traj(ao, 0, 0); o.update_traj(ao, 0, 0)
    assert np.array_equal(g.get_traj(0), o.get_traj(0))
    for l in (1, 2):
        ig, io = g.newton_step(1, l), o.newton_step(1, l)
        for f in ("status", "control_flow", "ls_j", "ls_failed"):
            assert np.array_equal(ig[f], io[f]), f
        assert np.array_equal(ig["alpha"], io["alpha"])
        assert np.allclose(ig["delta"], io["delta"], rtol=1e-9)
        for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(ig["rec"][f], io["rec"][f], rtol=1e-9, atol=1e-14), f
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())


@pytest.mark.parametrize("ingredients", [(), ("cost",), ("avoid",), ("ctl",)])
def test_ingredient_subsets_parity(alg, orc, ingredients):
    g, o = _pair(alg, orc, 2, 6, B=2, seed=17, ingredients=ingredients)
    rg, _ = g.residual(0, 0.0); ro, _ = o.residual(0, 0.0)
    assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
    dg, sg = g.newton_direction(1e-3); do, so = o.newton_direction(1e-3)
    assert np.all(sg == 0) and np.all(so == 0)
    assert (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9


def test_init_rollout_dual_update_and_guards(alg, orc):
    for case in (CASES[1], CASES[3]):
        g, o = _pair(alg, orc, *case, B=3, seed=11)
        g.init_traj(game_id0=1000); o.init_traj(game_id0=1000)
        Xg, Ug, Lg = g.split_traj(g.get_traj(0)); Xo, Uo, Lo = o.split_traj(o.get_traj(0))
        assert np.array_equal(Ug, Uo) and np.array_equal(Lg, Lo)
        assert np.abs(Xg - Xo).max() < 1e-13                                  # RK3 rollout, free fall from x0 (rotors ~ off)
        z = np.random.default_rng(5).random((3, g.traj_len)); z[:, :g.n] = g.get_traj()[:, :g.n]
        g.set_traj(z); o.set_traj(z)
        vg, vo = g.dual_penalty_update(), o.dual_penalty_update()
        fin = np.isfinite(vo)
        assert np.array_equal(np.isfinite(vg), fin) and np.abs(vg[fin] - vo[fin]).max() < 1e-14
        (lg, mg), (lo, mo) = g.get_con_duals(), o.get_con_duals()
        assert np.abs(lg - lo).max() < 1e-14 and np.array_equal(mg, mo)
        g.newton_solve(init=True)
        assert g.lib.debug_check_guards(g.h) == 0


def _assert_solve_parity(pg, po):
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), (f, sg[f], so[f])
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-9, atol=1e-9), f
    Xg, Ug, Lg = pg.batch.split_traj(pg.batch.get_traj()); Xo, Uo, Lo = po.batch.split_traj(po.batch.get_traj())
    assert np.abs(Xg - Xo).max() <= 1e-8 and np.abs(Ug - Uo).max() <= 1e-8
    assert np.abs(Lg - Lo).max() <= 1e-6 * max(1.0, np.abs(Lo).max())
    (lg, mg), (lo, mo) = pg.batch.get_con_duals(), po.batch.get_con_duals()
    assert np.array_equal(mg, mo) and np.abs(lg - lo).max() <= 1e-6 * max(1.0, np.abs(lo).max())
    for game in (0, len(sg) - 1):
        hg, ho = pg.stats.history(game), po.stats.history(game)
        assert len(hg) == len(ho)
        assert np.array_equal(hg["outer"], ho["outer"]) and np.array_equal(hg["ls_j"], ho["ls_j"]) and np.array_equal(hg["alpha"], ho["alpha"])
        assert np.allclose(hg["res"], ho["res"], rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("nw", [0, 1])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_newton_solve_parity_quadrotor_crossing(alg, orc, p, nw):
    """scenarios.quadrotor_crossing: p quadrotors fly across a circle holding height, planar collision avoidance on px[i],
    rotor commands in [0, 3]: the fused solver against the oracle, converged to the reference's exit test."""
    ids = np.arange(16, 16 + (12 if p < 4 else 6))
    pg = alg.scenarios.make_problem("Q", ids, p=p)
    po = alg.scenarios.make_problem("Q", ids, p=p, backend=orc.lib())
    # kernel shape: automatic (a team of four wavefronts per game for p >= 2 at this batch size) or one wavefront per game
    pg.batch.set_waves_per_game(nw)
    assert pg.batch.get_waves_per_game() == (1 if nw == 1 or p == 1 else 4)
    alg.newton_solve(pg); alg.newton_solve(po)
    _assert_solve_parity(pg, po)
    s = pg.stats.summary
    assert np.all(s["status"] == 0) and np.all(s["converged"] == 1)
    for f in ("opt_vio", "sta_vio", "dyn_vio", "con_vio"):
        assert np.all(s["last"][f] < 1e-3), f
    X = pg.batch.split_traj(pg.batch.get_traj())[0]
    assert np.hypot(X[:, -1, :p] - X[:, 0, :p], X[:, -1, p:2 * p] - X[:, 0, p:2 * p]).min() > 0.5     # they really fly across
    if p > 1:
        assert pg.batch.get_con_duals()[0].max() > 1e-4                        # the avoidance constraints are active
    z1 = pg.batch.get_traj()
    alg.newton_solve(pg)
    assert np.array_equal(pg.batch.get_traj(), z1)                             # deterministic


def test_ibr_and_mpc_parity_quadrotor(alg, orc):
    g, o = _pair(alg, orc, 2, 7, B=3, seed=13)
    for b in (g, o):
        b.set_options(outer_iter=1, inner_iter=1, dual_reset=0, reg_0=1e-3)
    for player in range(2):
        sg, so = g.ibr_solve_player(player), o.ibr_solve_player(player)
        for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
            assert np.array_equal(sg[f], so[f]), (player, f, sg[f], so[f])
        zg, zo = g.get_traj(0), o.get_traj(0)
        assert np.abs(zg - zo).max() <= 1e-9 * max(1.0, np.abs(zo).max())
        dg, do = g.get_traj(2), o.get_traj(2)
        assert np.abs(dg - do).max() <= 1e-9 * max(1.0, np.abs(do).max())
    ids = np.arange(300, 304)
    pg = alg.scenarios.make_problem("Q", ids, p=2, N=10)
    po = alg.scenarios.make_problem("Q", ids, p=2, N=10, backend=orc.lib())
    ig, cg, sg = alg.mpc_solve(pg, 4, record_states=True)
    io, co, so = alg.mpc_solve(po, 4, record_states=True)
    assert np.array_equal(ig, io) and np.array_equal(cg, co)
    assert sg.shape == so.shape == (5, 4, 24) and np.abs(sg - so).max() < 1e-7
    ps = alg.scenarios.make_problem("Q", ids, p=2, N=10)
    is_, cs, ss = alg.mpc_solve(ps, 4, record_states=True, fused=False)
    assert np.array_equal(is_, ig) and np.abs(ss - sg).max() < 1e-9


def test_quadrotor_mass_parameter_parity(alg, orc):
    """alg_set_quadrotor (QuadrotorGame(; mass), quadrotor.jl:20): residual, Jacobian and direction with a non-default mass."""
    g, o = _pair(alg, orc, 2, 6, B=2, seed=21)
    g.set_quadrotor(0.8); o.set_quadrotor(0.8)
    rg, _ = g.residual(0, 0.0); ro, _ = o.residual(0, 0.0)
    assert np.abs(rg - ro).max() <= 1e-12 * (1 + np.abs(ro).max())
    g2, o2 = _pair(alg, orc, 2, 6, B=2, seed=21)
    assert np.abs(ro - o2.residual(0, 0.0)[0]).max() > 1e-3                     # the mass matters
    Jg, Jo = g.residual_jacobian(1e-3), o.residual_jacobian(1e-3)
    assert np.abs(Jg - Jo).max() <= 1e-12 * np.abs(Jo).max()
    dg, sg = g.newton_direction(1e-3); do, so = o.newton_direction(1e-3)
    assert np.all(sg == 0) and (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9
    ig, io = g.newton_step(1, 1), o.newton_step(1, 1)
    assert np.array_equal(ig["ls_j"], io["ls_j"]) and np.abs(g.get_traj(0) - o.get_traj(0)).max() <= 1e-9 * max(1.0, np.abs(o.get_traj(0)).max())


def test_quadrotor_rejections(alg):
    """quadrotor.jl:22: at most four players; anything else is refused with a message."""
    with pytest.raises(alg.AlgamesError):
        alg.Batch(alg.hip_lib(), QUAD, 5, 6, 0.1, 2, d=3)
