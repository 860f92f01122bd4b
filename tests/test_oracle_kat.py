"""Pins the CPU oracle (and the host-side index logic) to every literal known-answer value the
reference's own tests hold for the Newton / augmented-Lagrangian path.
Each test cites the reference test file:line (relative to /root/reference) it restates.
The reference itself (Julia) cannot run here, so these literals + the end-to-end thresholds of
test/problem/solver_methods.jl are the pinning."""
import numpy as np
import pytest

DI, UNI, BIC, QUAD = 0, 1, 2, 3


# ---------------------------------------------------------------- layout / indexing (host logic)
def test_vertical_indices_literals(alg):
    # test/core/newton_core.jl:4-16
    model = alg.UnicycleGame(p=2)
    ps = alg.ProblemSize(3, model)
    n, mi = ps.n, ps.mi
    v = alg.vertical_indices(ps)
    assert v[alg.stampify("opt", 1, "x", 1, 2)] == list(range(1, n + 1))
    assert v[alg.stampify("opt", 1, "u", 1, 1)] == [n + i for i in range(1, mi[0] + 1)]
    assert v[alg.stampify("opt", 1, "x", 1, 3)] == [n + mi[0] + i for i in range(1, n + 1)]
    assert v[alg.stampify("opt", 1, "u", 1, 2)] == [2 * n + mi[0] + i for i in range(1, mi[0] + 1)]
    assert v[alg.stampify("opt", 2, "x", 1, 2)] == [2 * n + 2 * mi[0] + i for i in range(1, n + 1)]
    # test/core/newton_core.jl:18-41: a permutation of 1:S
    allinds = sorted(sum(v.values(), []))
    assert allinds == list(range(1, ps.S + 1))
    assert ps.S == ps.p * n * 2 + ps.m * 2 + n * 2


def test_horizontal_indices_literals(alg):
    # test/core/newton_core.jl:44-59
    model = alg.UnicycleGame(p=2)
    ps = alg.ProblemSize(3, model)
    n, m, mi = ps.n, ps.m, ps.mi
    h = alg.horizontal_indices(ps)
    assert h[alg.stampify("x", 1, 2)] == list(range(1, n + 1))
    assert h[alg.stampify("u", 1, 1)] == [n + i for i in range(1, mi[0] + 1)]
    assert h[alg.stampify("u", 2, 1)] == [n + mi[0] + i for i in range(1, mi[1] + 1)]
    assert h[alg.stampify("λ", 1, 1)] == [n + m + i for i in range(1, n + 1)]
    assert h[alg.stampify("λ", 2, 1)] == [2 * n + m + i for i in range(1, n + 1)]
    assert h[alg.stampify("x", 1, 3)] == [3 * n + m + i for i in range(1, n + 1)]
    assert sorted(sum(h.values(), [])) == list(range(1, ps.S + 1))


def test_stamp_validity_truth_table(alg):
    # test/core/stamp.jl:8-104 (N=10, p=3)
    N, p = 10, 3
    V = lambda *a: alg.valid(alg.stampify(*a), N, p)
    assert V("opt", 1, "x", 1, 5)
    assert not V("opt", 4, "x", 1, 5)
    assert not V("opt", 2, "u", 2, 10)
    assert V("opt", 2, "u", 2, 9)
    assert not V("opt", 2, "u", 3, 9)
    assert not V("opt", 2, "z", 1, 1)
    assert not V("dyn", 1, "u", 2, 1)
    assert V("dyn", 1, "x", 1, 1)
    assert not V("dyn", 2, "x", 1, 1)
    assert not V("dyn", 2, "x", 2, 1)
    assert V("opt", 2, "u", 2, 4)
    assert V("u", 2, 3)
    assert not V("u", 0, 3)
    assert not V("u", 1, 100)
    assert not V("x", 2, 2)
    assert not V("x", 1, 1)
    assert V("x", 1, 2)
    assert V("λ", 1, 2)
    assert not V("λ", 0, 2)
    assert V("λ", 2, 4)
    assert not V("λ", 2, 40)
    assert V("opt", 1, "x", 1, 5, "x", 1, 3)
    assert not V("opt", 1, "x", 1, 5, "u", 0, 3)
    assert V("opt", 1, "x", 1, 5, "u", 3, 3)
    assert V("opt", 1, "x", 1, 5, "u", 1, 3)
    assert not V("opt", 1, "x", 1, 5, "λ", 3, 3)
    assert V("opt", 1, "x", 1, 5, "λ", 1, 3)
    assert V("dyn", 1, "x", 1, 5, "x", 1, 3)
    assert V("dyn", 1, "x", 1, 5, "u", 2, 3)


def test_model_index_sets(alg):
    # test/dynamics/double_integrator.jl:3-18
    m = alg.DoubleIntegratorGame(p=2, d=3)
    assert (m.n, m.m, m.p) == (12, 6, 2)
    assert m.ni == [6, 6] and m.mi == [3, 3]
    assert m.pu == [[1, 3, 5], [2, 4, 6]]
    assert m.px == [[1, 3], [2, 4]]
    assert m.pz == [[1, 3, 5, 7, 9, 11], [2, 4, 6, 8, 10, 12]]
    # test/dynamics/unicycle.jl:3-20
    u = alg.UnicycleGame(p=3)
    assert (u.n, u.m, u.p) == (12, 6, 3)
    assert u.pu == [[1, 4], [2, 5], [3, 6]]
    assert u.px == [[1, 4], [2, 5], [3, 6]]
    assert u.pz == [[1, 4, 7, 10], [2, 5, 8, 11], [3, 6, 9, 12]]
    # test/dynamics/bicycle.jl:3-24
    bm = alg.BicycleGame(p=3, lr=1.0, lf=2.0)
    assert (bm.lr, bm.lf) == (1.0, 2.0)
    assert (bm.n, bm.m, bm.p) == (12, 6, 3)
    assert bm.ni == [4, 4, 4] and bm.mi == [2, 2, 2]
    assert bm.pu == [[1, 4], [2, 5], [3, 6]]
    assert bm.px == [[1, 4], [2, 5], [3, 6]]
    assert bm.pz == [[1, 4, 7, 10], [2, 5, 8, 11], [3, 6, 9, 12]]
    # ProblemSize.S, src/struct/problem_size.jl:22
    assert alg.ProblemSize(40, alg.DoubleIntegratorGame(p=3)).S == 2106
    assert alg.ProblemSize(50, alg.UnicycleGame(p=4)).S == 4312


def test_oracle_layout_matches_reference_indices(alg, orc):
    """The oracle's flat layouts ARE the reference's vertical / horizontal orders: plant one value per
    block through split/join and recover it at the literal offsets of newton_core.jl."""
    model = alg.UnicycleGame(p=3)
    N = 10
    ps = alg.ProblemSize(N, model)
    b = orc.OracleBatch(UNI, 3, N, 0.1, 1)
    h = alg.horizontal_indices(ps)
    rng = np.random.default_rng(0)
    dtraj = rng.random(ps.S)
    z = np.concatenate([np.zeros(ps.n), dtraj])[None]
    X, U, L = b.split_traj(z)
    # test/struct/primal_dual_traj.jl:48-63 (set_traj!)
    ix = lambda s: np.array(h[s]) - 1
    assert np.array_equal(X[0, 1], dtraj[ix(alg.stampify("x", 1, 2))])
    assert np.array_equal(X[0, N - 1], dtraj[ix(alg.stampify("x", 1, N))])
    for (i, k) in [(1, 1), (2, 1), (1, 2), (2, 2), (3, 2), (1, N - 1)]:
        assert np.array_equal(U[0, k - 1][np.array(model.pu[i - 1]) - 1], dtraj[ix(alg.stampify("u", i, k))])
    assert np.array_equal(L[0, 0, 0], dtraj[ix(alg.stampify("λ", 1, 1))])
    assert np.array_equal(L[0, 0, N - 2], dtraj[ix(alg.stampify("λ", 1, N - 1))])
    assert np.array_equal(L[0, 2, N - 2], dtraj[ix(alg.stampify("λ", 3, N - 1))])
    # test/struct/primal_dual_traj.jl:65-84 (get_traj! round trip, exact)
    assert np.array_equal(b.join_traj(X, U, L), z)
    b.set_traj(z); assert np.array_equal(b.get_traj(), z)


def test_update_traj_and_delta_step(orc):
    # test/struct/primal_dual_traj.jl:86-107: 10 + 0.5*100 = 60 everywhere, x_1 untouched
    b = orc.OracleBatch(UNI, 3, 10, 0.1, 1)
    x0 = np.random.default_rng(1).random(b.n)
    src = np.full((1, b.traj_len), 10.0); src[0, :b.n] = x0
    dlt = np.full((1, b.traj_len), 100.0); dlt[0, :b.n] = 0
    tgt = np.zeros((1, b.traj_len)); tgt[0, :b.n] = x0
    b.set_traj(src, 0); b.set_traj(tgt, 1); b.set_traj(dlt, 2)
    b.update_traj(0.5, target=1, source=0)
    out = b.get_traj(1)
    assert np.array_equal(out[0, :b.n], x0)
    assert np.all(out[0, b.n:] == 60.0)
    # test/struct/primal_dual_traj.jl:109-123: Δ_step == 10*α (duals ignored)
    ten = np.full((1, b.traj_len), 10.0)
    b.set_traj(ten, 2)
    assert b.kat_delta_step(0.5) == 10.0 * 0.5


# ---------------------------------------------------------------- dynamics
def test_dynamics_and_rk2(orc):
    rng = np.random.default_rng(2)
    # src/dynamics/double_integrator.jl:27-31 ; test/problem/local_quantities.jl:4-14 (RK2 within 1e-3 of Euler at dt=0.01)
    b = orc.OracleBatch(DI, 3, 10, 0.01, 1)
    x, u = rng.random(b.n), rng.random(b.m)
    xd, x2, x3, J = b.kat_dynamics(x, u)
    assert np.array_equal(xd, np.concatenate([x[b.m:], u]))
    assert np.abs(x2 - (x + 0.01 * xd)).sum() < 1e-3
    # exact DI maps: A = [[I, dt I],[0, I]], B = [[dt^2/2 I],[dt I]]; RK3 gives the same map
    dt, m = 0.01, b.m
    A = np.block([[np.eye(m), dt * np.eye(m)], [np.zeros((m, m)), np.eye(m)]])
    Bm = np.vstack([dt * dt / 2 * np.eye(m), dt * np.eye(m)])
    assert np.allclose(J, np.hstack([A, Bm]), atol=1e-15)
    assert np.allclose(x2, A @ x + Bm @ u, atol=1e-15) and np.allclose(x3, x2, atol=1e-15)
    # src/dynamics/unicycle.jl:27-32
    b = orc.OracleBatch(UNI, 3, 10, 0.2, 1)
    x, u = rng.random(b.n), rng.random(b.m)
    xd, x2, x3, J = b.kat_dynamics(x, u)
    p = 3
    ref = np.concatenate([np.cos(x[6:9]) * x[9:12], np.sin(x[6:9]) * x[9:12], u])
    assert np.allclose(xd, ref, atol=1e-16)
    # test/problem/local_quantities.jl:24-57: discrete_jacobian!(RK2) == derivative of discrete_dynamics(RK2)
    Jfd = np.zeros_like(J); eps = 1e-6
    for c in range(b.n + b.m):
        zp = np.concatenate([x, u]); zm = zp.copy(); zp[c] += eps; zm[c] -= eps
        Jfd[:, c] = (b.kat_dynamics(zp[:b.n], zp[b.n:])[1] - b.kat_dynamics(zm[:b.n], zm[b.n:])[1]) / (2 * eps)
    assert np.abs(J - Jfd).sum() < 1e-7
    # RK3 formula (RobotDynamics 0.3.1) restated independently
    f = lambda xx: np.concatenate([np.cos(xx[6:9]) * xx[9:12], np.sin(xx[6:9]) * xx[9:12], u])
    k1 = 0.2 * f(x); k2 = 0.2 * f(x + k1 / 2); k3 = 0.2 * f(x - k1 + 2 * k2)
    assert np.allclose(x3, x + (k1 + 4 * k2 + k3) / 6, atol=1e-15)
    assert np.allclose(x2, x + 0.2 * f(x + 0.1 * f(x)), atol=1e-15)


def test_bicycle_dynamics(orc):
    # src/dynamics/bicycle.jl:28-41 restated independently (lf = lr = 0.05, the BicycleGame defaults)
    rng = np.random.default_rng(5)
    b = orc.OracleBatch(BIC, 3, 10, 0.1, 1)
    x, u = rng.random(b.n), rng.random(b.m) - 0.5
    lf = lr = 0.05

    def f(xx, uu):
        beta = np.arctan2(lr * np.tan(uu[3:6]), lr + lf)
        return np.concatenate([xx[6:9] * np.cos(beta + xx[9:12]), xx[6:9] * np.sin(beta + xx[9:12]), uu[0:3], xx[6:9] * np.sin(beta) / lr])
    xd, x2, x3, J = b.kat_dynamics(x, u)
    assert np.allclose(xd, f(x, u), atol=1e-15)
    assert np.allclose(x2, x + 0.1 * f(x + 0.05 * f(x, u), u), atol=1e-15)
    k1 = 0.1 * f(x, u); k2 = 0.1 * f(x + k1 / 2, u); k3 = 0.1 * f(x - k1 + 2 * k2, u)
    assert np.allclose(x3, x + (k1 + 4 * k2 + k3) / 6, atol=1e-15)
    Jfd = np.zeros_like(J); eps = 1e-6
    for c in range(b.n + b.m):
        zp = np.concatenate([x, u]); zm = zp.copy(); zp[c] += eps; zm[c] -= eps
        Jfd[:, c] = (b.kat_dynamics(zp[:b.n], zp[b.n:])[1] - b.kat_dynamics(zm[:b.n], zm[b.n:])[1]) / (2 * eps)
    assert np.abs(J - Jfd).sum() < 1e-6


def test_quadrotor_index_literals(alg):
    # test/dynamics/quadrotor.jl:4-25
    p = 3
    model = alg.QuadrotorGame(p=p)
    assert (model.n, model.m, model.p) == (12 * p, 4 * p, p)
    assert model.ni == [12] * p and model.mi == [4] * p
    assert model.pu == [[1, 4, 7, 10], [2, 5, 8, 11], [3, 6, 9, 12]]
    assert model.px == [[1, 4], [2, 5], [3, 6]]
    assert model.pz[0] == [1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 34]
    assert model.pz[1] == [2, 5, 8, 11, 14, 17, 20, 23, 26, 29, 32, 35]
    assert model.pz[2] == [3, 6, 9, 12, 15, 18, 21, 24, 27, 30, 33, 36]
    assert alg.dim(model) == 3                                  # quadrotor.jl:208
    with pytest.raises(AssertionError):
        alg.QuadrotorGame(p=5)                                   # quadrotor.jl:22


def _quad_f(x, u, P, mass=0.5):
    """src/dynamics/quadrotor.jl:49-121 restated independently: attitude through the unit quaternion of the MRP
    (Rotations.jl 1.0: q = ((1 - |g|^2), 2 g) / (1 + |g|^2)) and the generic quaternion rotation matrix, MRP rate through
    B(g) w / 4 with B = (1 - |g|^2) I + 2 [g x] + 2 g g'."""
    J, grav, L, kf, km = np.array([0.0023, 0.0023, 0.004]), np.array([0, 0, -9.81]), 0.175, 1.245, 1.0
    xd = np.zeros_like(x)
    for i in range(P):
        g = x[[3 * P + i, 4 * P + i, 5 * P + i]]; v = x[[6 * P + i, 7 * P + i, 8 * P + i]]; w = x[[9 * P + i, 10 * P + i, 11 * P + i]]
        wm = u[[i, P + i, 2 * P + i, 3 * P + i]]
        s = g @ g
        qw, qv = (1 - s) / (1 + s), 2 * g / (1 + s)
        K = np.array([[0, -qv[2], qv[1]], [qv[2], 0, -qv[0]], [-qv[1], qv[0], 0]])
        Rm = np.eye(3) + 2 * qw * K + 2 * K @ K
        F = np.maximum(0, kf * wm)
        f = mass * grav + Rm @ np.array([0, 0, F.sum()])
        tau = np.array([L * (F[1] - F[3]), L * (F[2] - F[0]), km * (wm[0] - wm[1] + wm[2] - wm[3])])
        G = np.array([[0, -g[2], g[1]], [g[2], 0, -g[0]], [-g[1], g[0], 0]])
        qd = 0.25 * ((1 - s) * np.eye(3) + 2 * G + 2 * np.outer(g, g)) @ w
        wd = (tau - np.cross(w, J * w)) / J
        for a in range(3):
            xd[a * P + i] = v[a]; xd[(3 + a) * P + i] = qd[a]; xd[(6 + a) * P + i] = f[a] / mass; xd[(9 + a) * P + i] = wd[a]
    return xd


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_quadrotor_dynamics(orc, p):
    rng = np.random.default_rng(40 + p)
    dt = 0.05
    b = orc.OracleBatch(QUAD, p, 6, dt, 1)
    assert (b.n, b.m) == (12 * p, 4 * p)
    x, u = rng.random(b.n) - 0.3, rng.random(b.m) - 0.2       # some rotor commands negative: the max(0, .) branch
    xd, x2, x3, J = b.kat_dynamics(x, u)
    assert np.allclose(xd, _quad_f(x, u, p), rtol=1e-13, atol=1e-13)
    f = lambda xx: _quad_f(xx, u, p)
    assert np.allclose(x2, x + dt * f(x + dt / 2 * f(x)), rtol=1e-13, atol=1e-13)
    k1 = dt * f(x); k2 = dt * f(x + k1 / 2); k3 = dt * f(x - k1 + 2 * k2)
    assert np.allclose(x3, x + (k1 + 4 * k2 + k3) / 6, rtol=1e-13, atol=1e-13)
    # discrete_jacobian!(RK2) == derivative of discrete_dynamics(RK2) (test/problem/local_quantities.jl:24-57 pattern)
    Jfd = np.zeros_like(J); eps = 1e-6
    for c in range(b.n + b.m):
        zp = np.concatenate([x, u]); zm = zp.copy(); zp[c] += eps; zm[c] -= eps
        Jfd[:, c] = (b.kat_dynamics(zp[:b.n], zp[b.n:])[1] - b.kat_dynamics(zm[:b.n], zm[b.n:])[1]) / (2 * eps)
    assert np.abs(J - Jfd).max() < 1e-6
    # hover: identity attitude, zero rates, each rotor carrying a quarter of the weight -> xdot = 0
    x = np.zeros(b.n); u = np.full(b.m, 0.5 * 9.81 / 4 / 1.245)
    assert np.abs(b.kat_dynamics(x, u)[0]).max() < 1e-14


def test_quadrotor_game_end_to_end_on_the_oracle(alg, orc):
    """No solver test of the reference uses the QuadrotorGame; what can be held against it is its own exit test
    (solver_methods.jl:49-53): the crossing scenario of scenarios.quadrotor_crossing converges with every violation < 1e-3,
    the quadrotors reach the far side, planar collision avoidance (px[i], quadrotor.jl:34) is active and respected."""
    p = 2
    prob = alg.scenarios.make_problem("Q", np.arange(3), p=p, backend=orc.lib())
    assert isinstance(prob.model, alg.QuadrotorGame) and prob.probsize.n == 24 and prob.probsize.m == 8
    alg.newton_solve(prob)
    s = prob.stats.summary
    assert np.all(s["status"] == 0) and np.all(s["converged"] == 1)
    for f in ("opt_vio", "sta_vio", "dyn_vio", "con_vio"):
        assert np.all(s["last"][f] < 1e-3), f
    X, U, _ = prob.batch.split_traj(prob.batch.get_traj())
    d = np.hypot(X[:, :, 0] - X[:, :, 1], X[:, :, 2] - X[:, :, 3])              # planar distance of the two vehicles per knot
    assert d.min() > 0.2 - 2e-3 and d.min() < 0.25                               # pair radius 0.1 + 0.1: the constraint binds
    assert prob.batch.get_con_duals()[0].max() > 1e-4
    assert np.all(U > -1e-3) and np.all(U < 3.0 + 1e-3)                          # rotor commands inside their bounds
    assert np.abs(X[:, -1, 4:6] - 0.5).max() < 0.15                              # height held
    with pytest.raises(alg.AlgamesError):
        alg.velocity_index(prob.model, 1)                                        # velocity_constraint.jl:30-43: no method for this model


def test_quadrotor_mass_parameter(alg, orc):
    # QuadrotorGame(; p, mass) (quadrotor.jl:20): the mass enters vdot = g + R F / m only
    rng = np.random.default_rng(77)
    b = orc.OracleBatch(QUAD, 2, 6, 0.05, 1)
    b.set_quadrotor(0.8)
    x, u = rng.random(b.n) - 0.3, rng.random(b.m)
    xd = b.kat_dynamics(x, u)[0]
    assert np.allclose(xd, _quad_f(x, u, 2, mass=0.8), rtol=1e-13, atol=1e-13)
    assert np.abs(xd - _quad_f(x, u, 2, mass=0.5))[12:18].max() > 1e-3          # the velocity rows really changed
    with pytest.raises(alg.AlgamesError):
        b.set_quadrotor(0.0)
    with pytest.raises(alg.AlgamesError):
        orc.OracleBatch(UNI, 2, 6, 0.05, 1).set_quadrotor(0.8)
    model = alg.QuadrotorGame(p=2, mass=0.8)
    assert model.mass == 0.8
    with pytest.raises(alg.AlgamesError):
        alg.QuadrotorGame(p=2, mass=-1.0)


def test_quadrotor_rotation_is_orthogonal():
    # the restated MRP rotation (oracle: I + (4 (1 - s) [g x] + 8 [g x]^2) / (1 + s)^2) is a proper rotation and matches
    # Rodrigues for angle 4 atan(|g|) about g / |g|
    rng = np.random.default_rng(9)
    for _ in range(20):
        g = rng.normal(size=3) * 0.6
        s = g @ g
        G = np.array([[0, -g[2], g[1]], [g[2], 0, -g[0]], [-g[1], g[0], 0]])
        Rm = np.eye(3) + (4 * (1 - s) * G + 8 * G @ G) / (1 + s) ** 2
        assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-13) and abs(np.linalg.det(Rm) - 1) < 1e-13
        th, ax = 4 * np.arctan(np.sqrt(s)), g / np.sqrt(s)
        A = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        assert np.allclose(Rm, np.eye(3) + np.sin(th) * A + (1 - np.cos(th)) * A @ A, atol=1e-13)


# ---------------------------------------------------------------- objective
def test_lqr_gradient_hessian_scaling(orc):
    # test/objective/objective.jl:11-64: zero cost at (xf, uf) through the pz/pu padding; q=(Qx+q)dt,
    # r=(Ru+r)dt for stage knots, terminal unscaled, r_N = 0, Q dt / Q
    rng = np.random.default_rng(3)
    N, dt, p = 10, 0.1, 3
    b = orc.OracleBatch(UNI, p, N, dt, 1)
    Q, R = rng.random((p, 4)), rng.random((p, 2))
    xf = np.array([[i] * 4 for i in (1, 2, 3)], float); uf = np.array([[2 * i] * 2 for i in (1, 2, 3)], float)
    b.set_lqr(Q, R, xf, uf)
    X = np.array([1, 2, 3] * 4, float); U = np.array([2, 4, 6] * 2, float)
    for i in range(p):
        q, r, _ = b.kat_cost(i, 0, X, U)
        assert np.abs(q).max() <= 1e-10 and np.abs(r).max() <= 1e-10
    x, u = 10 * rng.random(b.n), 10 * rng.random(b.m)
    Qfull = np.zeros(b.n); Qfull[[0, 3, 6, 9]] = Q[0]; xff = np.zeros(b.n); xff[[0, 3, 6, 9]] = xf[0]
    q, r, Qm = b.kat_cost(0, 0, x, u)
    assert np.abs(q - Qfull * (x - xff) * dt).sum() < 1e-10
    assert np.abs(r - R[0] * (u[[0, 3]] - uf[0]) * dt).sum() < 1e-10
    assert np.abs(Qm - np.diag(Qfull) * dt).sum() < 1e-10
    q, r, Qm = b.kat_cost(0, N - 1, x, u)
    assert np.abs(q - Qfull * (x - xff)).sum() < 1e-10
    assert np.abs(Qm - np.diag(Qfull)).sum() < 1e-10


def test_collision_cost_value_gradient_hessian(orc):
    # test/objective/objective.jl:108-147: x=[1,1.1,2,2,0,0,0,0], μ=10, r=0.2 -> 0.05
    x = np.array([1.0, 1.1, 2.0, 2.0, 0, 0, 0, 0])
    assert abs(orc.collision_cost_value(10.0, 0.2, x[[0, 2]], x[[1, 3]]) - 0.05) < 1e-10
    # test/objective/objective.jl:155-200: gradient/Hessian vs derivatives of stage_cost (active and inactive)
    rng = np.random.default_rng(4)
    dt = 0.1
    for rad, tolg in ((1e3, 1e-7), (1e-3, 1e-7)):
        b = orc.OracleBatch(DI, 2, 10, dt, 1)
        b.set_lqr(np.zeros((2, 4)), np.zeros((2, 2)), np.zeros((2, 4)), np.zeros((2, 2)))
        b.add_collision_cost([rad, rad], [10.0, 10.0])
        x, u = rng.random(b.n), rng.random(b.m)
        val = lambda xx: orc.collision_cost_value(10.0, rad, xx[[0, 2]], xx[[1, 3]])
        q, r, Qm = b.kat_cost(0, b.N - 1, x, u)       # terminal knot: unscaled
        g = np.zeros(b.n); Hfd = np.zeros((b.n, b.n)); e = 1e-5
        for c in range(b.n):
            xp, xm = x.copy(), x.copy(); xp[c] += e; xm[c] -= e
            g[c] = (val(xp) - val(xm)) / (2 * e)
            Hfd[:, c] = (b.kat_cost(0, b.N - 1, xp, u)[0] - b.kat_cost(0, b.N - 1, xm, u)[0]) / (2 * e)
        assert np.abs(q - g).sum() / max(np.abs(q).sum(), 1e-300) < 1e-6 or np.abs(q - g).sum() < tolg
        assert np.abs(Qm - Hfd).sum() < 1e-2
        assert np.all(r == 0)
        qs = b.kat_cost(0, 0, x, u)[0]                # stage knot: scaled by dt
        assert np.allclose(qs, q * dt, rtol=1e-15, atol=0)


# ---------------------------------------------------------------- constraints
def test_control_bound_evaluate_literal(orc):
    # test/constraints/control_bound_constraint.jl:3-15
    b = orc.OracleBatch(DI, 3, 5, 0.1, 1)      # m = 6
    U = np.array([13.0, 1.0, -12.0, 1.0, 2.0, 30.0])
    u_max = np.array([np.inf, np.inf, -11.0, 15.0, 2.0, 30.0])
    u_min = np.array([-np.inf, -10.0, -np.inf, 1.0, -2.0, -30.0])
    b.add_control_bound(u_max, u_min)
    X, _, L = b.split_traj(b.get_traj())
    b.set_traj(b.join_traj(X, np.broadcast_to(U, (1, 4, 6)).copy(), L))
    vals = b.kat_evaluate_con()[0]
    ctl = vals[b.p * (b.p - 1) * (b.N - 1):].reshape(b.N - 1, 12)
    finite = ctl[0][np.isfinite(ctl[0])]
    assert np.array_equal(finite, np.array([-1.0, -14.0, 0.0, 0.0, -11.0, 0.0, -4.0, -60.0]))
    with pytest.raises(Exception):              # checkBounds, control_bound_constraint.jl:69-75
        b.add_control_bound(np.zeros(6), np.ones(6))


def _ext_off(b):
    return b.p * (b.p - 1) * (b.N - 1) + 2 * b.m * (b.N - 1)


def test_state_bound_evaluate_literal(orc):
    # test/constraints/state_bound_constraint.jl:3-15 (n = 6 -> one DoubleIntegrator player in d = 3)
    X6 = np.array([13.0, 1.0, -12.0, 1.0, 2.0, 30.0])
    x_max = np.array([np.inf, np.inf, -11.0, 15.0, 2.0, 30.0])
    x_min = np.array([-np.inf, -10.0, -np.inf, 1.0, -2.0, -30.0])
    b = orc.OracleBatch(DI, 1, 4, 0.1, 1, d=3)
    b.set_lqr(np.zeros((1, 6)), np.zeros((1, 3)), np.zeros((1, 6)), np.zeros((1, 3)))
    b.add_state_bound(0, x_max, x_min)
    assert b.con_len == _ext_off(b) + 12 * (b.N - 1)
    X, U, L = b.split_traj(b.get_traj())
    X[0, 1:] = X6
    b.set_traj(b.join_traj(X, U, L))
    sb = b.kat_evaluate_con()[0][_ext_off(b):].reshape(b.N - 1, 12)
    for k in range(b.N - 1):
        assert np.array_equal(sb[k][np.isfinite(sb[k])], np.array([-1.0, -14.0, 0.0, 0.0, -11.0, 0.0, -4.0, -60.0]))
    assert np.array_equal(np.nonzero(np.isfinite(sb[0]))[0] + 1, [3, 4, 5, 6, 8, 10, 11, 12])     # con.inds, :16
    with pytest.raises(Exception):
        b.add_state_bound(0, np.zeros(6), np.ones(6))


def test_wall_evaluate_literal(orc):
    # test/constraints/wall_constraint.jl:3-18: position (x, y) = (1, 1)
    s2 = np.sqrt(2.0)
    x1 = [0.0, 0.0, 1.0, 3.0, -2.0]; y1 = [1.0, -1.0, 2.0, 2.0, 0.0]
    x2 = [1.0, 1.0, 2.0, 2.0, 0.0]; y2 = [0.0, 0.0, 1.0, 1.0, 0.0]
    xv = np.array([1.0, 1.0, 1.0, 1.0, 0.0]) / s2; yv = np.array([1.0, -1.0, 1.0, -1.0, s2]) / s2
    b = orc.OracleBatch(DI, 1, 3, 0.1, 1)
    b.set_lqr(np.zeros((1, 4)), np.zeros((1, 2)), np.zeros((1, 4)), np.zeros((1, 2)))
    b.add_wall_constraint(x1, y1, x2, y2, xv, yv)
    X, U, L = b.split_traj(b.get_traj())
    X[0, 1:] = [1.0, 1.0, -12.0, 13.0]
    b.set_traj(b.join_traj(X, U, np.zeros_like(L)))
    w = b.kat_evaluate_con()[0][_ext_off(b):].reshape(b.N - 1, 5)
    assert np.abs(w[0] - np.array([s2 / 2, 0.0, -s2 / 2, 0.0, 0.0])).sum() < 1e-10
    # jacobian (wall_constraint.jl:79-96; test :20-31): opt rows of a pure-constraint problem = C'(lambda + a mu c)
    lam = np.zeros((1, b.con_len)); mu = np.ones((1, b.con_len))
    lam[0, _ext_off(b):] = np.tile([0.5, 0.25, 2.0, 1.0, 3.0], b.N - 1)
    b.set_con_duals(lam, mu)
    res = b.residual()[0][0]
    left_right = np.array([1.0, 0.0, 1.0, 0.0, 0.0])           # inside the slab of walls 1 and 3 only
    wgt = lam[0, _ext_off(b):_ext_off(b) + 5] + 1.0 * w[0]     # a = 1: c >= 0 or lambda > 0 for every row here
    gx = (left_right * xv * wgt).sum(); gy = (left_right * yv * wgt).sum()
    assert np.allclose(res[0:4], [gx, gy, 0.0, 0.0], atol=1e-14)


def test_wall_state_violation_literal(orc):
    # test/struct/violations.jl:36-48: Unicycle p = 3, N = 10, all-ones trajectory from x0 = 0, wall (0,1)-(1,0) with v = (1,1)/sqrt(2)
    N, p = 10, 3
    b = orc.OracleBatch(UNI, p, N, 0.1, 1)
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    s2 = np.sqrt(2.0)
    b.add_wall_constraint([0.0], [1.0], [1.0], [0.0], [1 / s2], [1 / s2])
    z = np.ones((1, b.traj_len)); z[0, :b.n] = 0.0
    b.set_traj(z)
    w = b.kat_evaluate_con()[0][_ext_off(b):].reshape(p, N - 1)
    assert np.abs(w - s2 / 2).sum() <= 1e-10                    # sqrt(2)/2 on knots 2..N for every player (knot 1 carries no constraint)
    assert abs(b.record()["sta_vio"][0] - s2 / 2) < 1e-10


def test_velocity_bound_adders(alg):
    # test/constraints/velocity_constraint.jl:3-46
    N = 10
    model = alg.UnicycleGame(p=3)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_velocity_bound(model, con, np.ones(3), -np.ones(3))
    assert len(con.state_conval) == 3 and len(con.state_conval[0]) == 3
    assert [alg.velocity_index(model, i) for i in (1, 2, 3)] == [10, 11, 12]
    mx, mn = con.state_bounds[2]
    assert np.array_equal(np.nonzero(np.isfinite(mx))[0], [9, 10, 11]) and np.all(mx[9:] == 1) and np.all(mn[9:] == -1)
    model = alg.BicycleGame(p=3)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_velocity_bound(model, con, [1, np.inf, np.inf], [1, -1, -np.inf])
    assert len(con.state_conval) == 3 and len(con.state_conval[0]) == 2
    assert [alg.velocity_index(model, i) for i in (1, 2, 3)] == [7, 8, 9]
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_velocity_bound(model, con, np.full(3, np.inf), np.full(3, -np.inf))
    assert len(con.state_conval) == 3 and len(con.state_conval[0]) == 0
    with pytest.raises(Exception):
        alg.velocity_index(alg.DoubleIntegratorGame(p=2), 1)


def test_circle_and_extended_constraints_fd(orc):
    # CircleConstraint (TrajectoryOptimization 0.4.1, un-vendored): c = r^2 - (x-xc)^2 - (y-yc)^2 <= 0 keeps the
    # player outside the disc.  FD check: opt_x rows = d/dx of each player's AL penalty with the active set frozen,
    # jacobian = its derivative (state bound + wall + circle together, unicycle p = 2).
    rng = np.random.default_rng(9)
    N, p = 5, 2
    b = orc.OracleBatch(UNI, p, N, 0.1, 1)
    n = b.n
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    xmax = np.full(n, np.inf); xmin = np.full(n, -np.inf); xmax[[0, 3, 6]] = [0.5, 0.4, 0.6]; xmin[[1, 7]] = [0.3, 0.55]
    b.add_state_bound(1, xmax, xmin)
    b.add_wall_constraint([-1.0], [0.5], [2.0], [0.5], [0.0], [1.0])
    xc, yc, rad = np.array([0.5, 0.2]), np.array([0.5, 0.9]), np.array([0.4, 0.3])
    b.add_circle_constraint(xc, yc, rad)
    z = rng.random((1, b.traj_len)); X, U, L = b.split_traj(z)
    z = b.join_traj(X, U, np.zeros_like(L)); b.set_traj(z)
    off = _ext_off(b); K = N - 1
    vals = b.kat_evaluate_con()[0]
    circ = vals[off + p * 2 * n * K + p * K:].reshape(p, K, 2)
    for i in range(p):
        for k in range(1, N):
            assert np.allclose(circ[i, k - 1], rad ** 2 - (X[0, k, i] - xc) ** 2 - (X[0, k, i + p] - yc) ** 2, atol=1e-15)
    lam = rng.random((1, b.con_len)) * (rng.random((1, b.con_len)) > 0.5); mu = np.full((1, b.con_len), 3.0)
    b.set_con_duals(lam, mu)
    res0 = b.residual()[0][0]; J = b.residual_jacobian()[0]
    act = ((vals >= 0) | (lam[0] > 0)) & np.isfinite(vals)

    def pen(xk, i, k):          # player i's AL penalty at knot k (1-based knot index k+1), active set frozen
        s = 0.0
        def term(e, c):
            return lam[0, e] * c + 0.5 * 3.0 * act[e] * c * c if np.isfinite(c) else 0.0
        if i == 1:
            for r in range(2 * n):
                c = xk[r] - xmax[r] if r < n else xmin[r - n] - xk[r - n]
                s += term(off + (i * K + k - 1) * 2 * n + r, c)
        inside = -1.0 < xk[i] < 2.0
        s += term(off + p * 2 * n * K + i * K + (k - 1), (xk[i + p] - 0.5) * inside)
        for c2 in range(2):
            s += term(off + p * 2 * n * K + p * K + (i * K + k - 1) * 2 + c2, rad[c2] ** 2 - (xk[i] - xc[c2]) ** 2 - (xk[i + p] - yc[c2]) ** 2)
        return s
    eps = 1e-6
    for i in range(p):
        for k in range(1, N):
            rows = i * K * (n + 2) + (k - 1) * (n + 2) + np.arange(n)
            g = np.zeros(n)
            for a in range(n):
                xp, xm = X[0, k].copy(), X[0, k].copy(); xp[a] += eps; xm[a] -= eps
                g[a] = (pen(xp, i, k) - pen(xm, i, k)) / (2 * eps)
            assert np.allclose(res0[rows], g, atol=1e-7), (i, k)
    # jacobian: the Gauss-Newton block C' I_mu C (constraint_derivatives.jl:10-19; no second derivative of c)
    def grads(xk, i, k):
        out = []
        if i == 1:
            for r in range(2 * n):
                g = np.zeros(n); g[r % n] = 1.0 if r < n else -1.0
                out.append((off + (i * K + k - 1) * 2 * n + r, g))
        g = np.zeros(n); g[i + p] = 1.0 * (-1.0 < xk[i] < 2.0)
        out.append((off + p * 2 * n * K + i * K + (k - 1), g))
        for c2 in range(2):
            g = np.zeros(n); g[i] = -2 * (xk[i] - xc[c2]); g[i + p] = -2 * (xk[i + p] - yc[c2])
            out.append((off + p * 2 * n * K + p * K + (i * K + k - 1) * 2 + c2, g))
        return out
    for i in range(p):
        for k in range(1, N):
            rows = i * K * (n + 2) + (k - 1) * (n + 2) + np.arange(n)
            cols = (k - 1) * b.b + np.arange(n)
            Hx = sum(3.0 * act[e] * np.outer(g, g) for e, g in grads(X[0, k], i, k))
            assert np.allclose(J[np.ix_(rows, cols)], Hx, atol=1e-13), (i, k)


def test_al_expansion_formula(orc):
    # test/constraints/constraint_derivatives.jl:3-34: vals = [-0.9; -1.1], grad = C'λ + C' Iρ c,
    # hess = C' Iρ C with Iρ = diag((c>=0)|(λ>0)) μ
    N, p = 10, 3
    b = orc.OracleBatch(UNI, p, N, 0.1, 1)
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    b.add_control_bound(np.ones(b.m), -np.ones(b.m))
    z = np.full((1, b.traj_len), 0.1)
    X, U, L = b.split_traj(z)
    b.set_traj(b.join_traj(X, U, np.zeros_like(L)))      # duals 0 so only the constraint term is in opt_u rows
    lam = np.zeros((1, b.con_len)); mu = np.full((1, b.con_len), 1.0)
    off = p * (p - 1) * (N - 1)
    lamc = lam[0, off:].reshape(N - 1, 2 * b.m)
    for k in range(N - 2):
        lamc[k] = k + 1
    b.set_con_duals(lam, mu)
    vals = b.kat_evaluate_con()[0, off:].reshape(N - 1, 2 * b.m)
    assert np.allclose(vals[0], np.r_[-0.9 * np.ones(b.m), -1.1 * np.ones(b.m)], atol=1e-16)
    assert np.allclose(vals[-1], np.r_[-0.9 * np.ones(b.m), -1.1 * np.ones(b.m)], atol=1e-16)
    res = b.residual()[0][0]
    J = b.residual_jacobian()[0]
    C = np.vstack([np.eye(b.m), -np.eye(b.m)])
    for k in (0, N - 2):
        Irho = np.diag(((vals[k] >= 0) | (lamc[k] > 0)) * 1.0)
        grad = C.T @ lamc[k] + C.T @ Irho @ vals[k]
        hess = C.T @ Irho @ C
        for i in range(p):
            rows = i * (N - 1) * (b.n + 2) + k * (b.n + 2) + b.n + np.arange(2)
            pu = np.array([i, i + p])
            assert np.allclose(res[rows], grad[pu], atol=1e-15)
            cols = k * b.b + b.n + i * 2 + np.arange(2)
            assert np.allclose(J[np.ix_(rows, cols)], hess[np.ix_(pu, pu)], atol=1e-15)


def test_penalty_schedule_and_dual_update(orc):
    # test/constraints/constraints_methods.jl:135-229
    N, p = 20, 3
    b = orc.OracleBatch(DI, p, N, 0.1, 1)
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    b.add_control_bound(10 * np.ones(b.m), -10 * np.ones(b.m))
    b.set_options(rho_0=1e-3, rho_increase=1e1, rho_max=1e-1, lambda_max=1e1)
    off = p * (p - 1) * (N - 1)
    b.reset_con()
    assert np.all(b.get_con_duals()[1][0, off:] == 1e-3)
    b.dual_penalty_update(); assert np.allclose(b.get_con_duals()[1][0, off:], 1e-2, rtol=1e-15)
    b.dual_penalty_update(); assert np.allclose(b.get_con_duals()[1][0, off:], 1e-1, rtol=1e-15)
    for _ in range(4):
        b.dual_penalty_update()
    assert np.all(b.get_con_duals()[1][0, off:] == 1e-1)          # capped at μ_max
    b.reset_con()
    lam, mu = b.get_con_duals()
    assert np.all(mu[0, off:] == 1e-3) and np.all(lam == 0)
    # dual update at the zero trajectory: c < 0 -> λ stays 0 (:203-205)
    b.set_traj(np.zeros((1, b.traj_len)))
    b.dual_penalty_update()
    assert np.all(b.get_con_duals()[0] == 0)
    # all-100 trajectory: vals = [90; -110], λ = 1e-3 [90; 0] (:207-212)
    b.reset_con()
    b.set_traj(np.full((1, b.traj_len), 1e2))
    vals = b.dual_penalty_update()[0, off:].reshape(N - 1, 2 * b.m)
    assert np.array_equal(vals[0], np.r_[90 * np.ones(b.m), -110 * np.ones(b.m)])
    lam = b.get_con_duals()[0][0, off:].reshape(N - 1, 2 * b.m)
    assert np.allclose(lam[0], 1e-3 * np.r_[90 * np.ones(b.m), np.zeros(b.m)], rtol=1e-15)
    # all-1e5 trajectory: clamp to λ_max = 10 (:214-219)
    b.reset_con()
    b.set_traj(np.full((1, b.traj_len), 1e5))
    vals = b.dual_penalty_update()[0, off:].reshape(N - 1, 2 * b.m)
    assert np.array_equal(vals[0], np.r_[(1e5 - 10) * np.ones(b.m), -(1e5 + 10) * np.ones(b.m)])
    lam = b.get_con_duals()[0][0, off:].reshape(N - 1, 2 * b.m)
    assert np.array_equal(lam[0], np.r_[10.0 * np.ones(b.m), np.zeros(b.m)])
    b.reset_con()
    assert np.all(b.get_con_duals()[0] == 0)


def test_violations(orc):
    # test/struct/violations.jl:3-33: zero trajectory -> zero dynamics violation; control violation 0.9
    N, p = 10, 3
    b = orc.OracleBatch(UNI, p, N, 0.1, 1)
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    b.set_traj(np.zeros((1, b.traj_len)))
    assert b.record()["dyn_vio"][0] == 0.0
    b.add_control_bound(0.1 * np.ones(b.m), -0.1 * np.ones(b.m))
    z = np.ones((1, b.traj_len)); z[0, :b.n] = 0
    b.set_traj(z)
    rec = b.record()
    assert rec["con_vio"][0] == 0.9
    # dynamics violation == max |RK2(z_1) - x_2| (violations.jl:18-26)
    X, U, L = b.split_traj(z)
    x2 = b.kat_dynamics(X[0, 0], U[0, 0])[1]
    viol = max(np.abs(b.kat_dynamics(X[0, k], U[0, k])[1] - X[0, k + 1]).max() for k in range(N - 1))
    assert abs(rec["dyn_vio"][0] - viol) < 1e-15
    # optimality violation = max |res| over the opt rows (violations.jl:153-168; test :62-68)
    res = b.residual()[0][0]
    nopt = p * (N - 1) * (b.n + 2)
    assert rec["opt_vio"][0] == np.abs(res[:nopt]).max()
    assert abs(rec["res"][0] - np.abs(res).sum() / b.S) < 1e-15


def test_collision_avoidance_constraint_terms(orc):
    """CollisionConstraint (TrajectoryOptimization 0.4.1; formula restated, parity unpinned):
    consistency of value / AL gradient / Gauss-Newton Hessian through finite differences."""
    N, p = 6, 3
    b = orc.OracleBatch(DI, p, N, 0.1, 1)
    b.set_lqr(np.zeros((p, 4)), np.zeros((p, 2)), np.zeros((p, 4)), np.zeros((p, 2)))
    b.add_collision_avoidance([0.6, 0.7, 0.8])
    rng = np.random.default_rng(5)
    z = rng.random((1, b.traj_len))
    X, U, L = b.split_traj(z)
    z = b.join_traj(X, U, np.zeros_like(L)); b.set_traj(z)
    vals = b.kat_evaluate_con()[0][:p * (p - 1) * (N - 1)].reshape(p * (p - 1), N - 1)
    q = 0
    for i in range(p):
        for j in range(p):
            if j == i:
                continue
            R = [0.6, 0.7, 0.8][i] + [0.6, 0.7, 0.8][j]
            for k in range(1, N):
                dl = X[0, k][[i, i + p]] - X[0, k][[j, j + p]]
                assert abs(vals[q, k - 1] - (R * R - dl @ dl)) < 1e-15
            q += 1
    # AL term: opt_i,x rows = d/dx [ λ c + 1/2 μ a c^2 ] with a frozen; check against FD of that scalar
    lam = rng.random((1, b.con_len)); mu = 3.0 * np.ones((1, b.con_len))
    b.set_con_duals(lam, mu)
    res0 = b.residual()[0][0]
    # residual of pure-constraint problem in row block opt_1,x_2 equals gradient of player-1 penalty at knot 2
    def pen(xk, i, k):
        s = 0.0
        for j in range(p):
            if j == i:
                continue
            qq = i * (p - 1) + (j if j < i else j - 1)
            R = [0.6, 0.7, 0.8][i] + [0.6, 0.7, 0.8][j]
            dl = xk[[i, i + p]] - xk[[j, j + p]]
            c = R * R - dl @ dl
            lm = lam[0, qq * (N - 1) + (k - 1)]
            a = 1.0 if (c >= 0 or lm > 0) else 0.0
            s += lm * c + 0.5 * 3.0 * a * c * c
        return s
    for i in range(p):
        for k in (1, N - 1):
            g = np.zeros(b.n); e = 1e-6
            for c in range(b.n):
                xp, xm = X[0, k].copy(), X[0, k].copy(); xp[c] += e; xm[c] -= e
                g[c] = (pen(xp, i, k) - pen(xm, i, k)) / (2 * e)
            rows = i * (N - 1) * (b.n + 2) + (k - 1) * (b.n + 2) + np.arange(b.n)
            assert np.allclose(res0[rows], g, atol=1e-6)


# ---------------------------------------------------------------- linear algebra of the oracle itself
@pytest.mark.parametrize("model,p,N", [(DI, 2, 8), (UNI, 2, 6), (UNI, 3, 5)])
def test_banded_lu_equals_dense_lu_in_reference_order(orc, model, p, N):
    """Δtraj from the banded partial-pivot LU == dense partial-pivot LU of the literal S x S matrix in
    the reference's own (vertical, horizontal) order, and J Δ = -res."""
    b = orc.OracleBatch(model, p, N, 0.1, 1)
    rng = np.random.default_rng(6)
    ni = b.n // p
    b.set_lqr(1 + rng.random((p, ni)), 0.5 + rng.random((p, b.mi)), rng.random((p, ni)), rng.random((p, b.mi)))
    b.set_x0(rng.random(b.n))
    b.add_collision_cost(np.full(p, 3.0), np.full(p, 2.0))
    b.add_collision_avoidance(np.full(p, 0.4))
    b.add_control_bound(np.full(b.m, 0.5), np.full(b.m, -0.5))
    z = rng.random((1, b.traj_len)); b.set_traj(z)
    b.set_con_duals(rng.random((1, b.con_len)), 2.0 * np.ones((1, b.con_len)))
    reg = 1e-3
    d_band, st = b.newton_direction(reg)
    assert st[0] == 0
    d_dense = b.kat_dense_direction(reg)
    assert np.allclose(d_band[0], d_dense, rtol=1e-9, atol=1e-10)
    J = b.residual_jacobian(reg)[0]
    res = b.residual()[0][0]
    assert np.abs(J @ d_band[0] + res).max() < 1e-9 * max(1.0, np.abs(res).max())


def test_jacobian_is_derivative_of_residual_for_linear_dynamics(orc):
    """For the double integrator without collision terms the Gauss-Newton Jacobian is exact:
    finite differences of residual! reproduce residual_jacobian! (structure + ordering check)."""
    p, N = 2, 5
    b = orc.OracleBatch(DI, p, N, 0.1, 1)
    rng = np.random.default_rng(7)
    b.set_lqr(1 + rng.random((p, 4)), 0.5 + rng.random((p, 2)), rng.random((p, 4)), rng.random((p, 2)))
    x0 = rng.random(b.n); b.set_x0(x0)
    z = rng.random((1, b.traj_len)); z[0, :b.n] = x0; b.set_traj(z)
    J = b.residual_jacobian(0.0)[0]
    r0 = b.residual()[0][0]
    Jfd = np.zeros_like(J); e = 1e-6
    for c in range(b.S):
        zp = z.copy(); zp[0, b.n + c] += e; b.set_traj(zp)
        Jfd[:, c] = (b.residual()[0][0] - r0) / e
    assert np.abs(J - Jfd).max() < 1e-6


# ---------------------------------------------------------------- end-to-end (test/problem/solver_methods.jl)
def _problem(alg, orc, model, x0, opts, constrained=False, circles=None):
    N, dt, p = 20, 0.1, model.p
    Q = [np.ones(model.ni[i]) for i in range(p)]
    R = [0.5 * np.ones(model.mi[i]) for i in range(p)]
    xf = [np.zeros(model.ni[i]) for i in range(p)]
    uf = [-np.ones(model.mi[i]) for i in range(p)]
    game_obj = alg.GameObjective(Q, R, xf, uf, N, model)
    game_con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    if constrained:
        alg.add_collision_avoidance(game_con, 0.05)
        alg.add_control_bound(game_con, np.ones(model.m), -np.ones(model.m))
    if circles is not None:
        alg.add_circle_constraint(game_con, *circles)
    return alg.GameProblem(N, dt, x0, model, opts, game_obj, game_con, backend=orc.lib())


def _check(alg, prob, tol):
    res = alg.residual(prob)
    assert np.abs(res).sum() / res.shape[1] < tol
    assert alg.dynamics_violation(prob)[0] < tol


def test_e2e_linear_one_player_one_newton_step(alg, orc):
    # test/problem/solver_methods.jl:6-34
    opts = alg.Options(inner_print=False, outer_print=False)
    prob = _problem(alg, orc, alg.DoubleIntegratorGame(p=1), [1.0, 1.0, 0.0, 0.9], opts)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 1, 1, 25, 1e-7, 1e-10, 1e-10
    alg.newton_solve(prob)
    _check(alg, prob, 1e-6)
    assert prob.stats.summary["newton_iters"][0] == 1


def test_e2e_unicycle_one_player(alg, orc):
    # test/problem/solver_methods.jl:36-65
    opts = alg.Options(inner_print=False, outer_print=False)
    prob = _problem(alg, orc, alg.UnicycleGame(p=1), [1.0, 1.0, 0.0, 0.9], opts)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 7, 20, 25, 1e-7, 1e-10, 1e-10
    alg.newton_solve(prob)
    _check(alg, prob, 1e-6)


def test_e2e_linear_two_players_one_newton_step(alg, orc):
    # test/problem/solver_methods.jl:68-97  (BASELINE config C1)
    opts = alg.Options(inner_print=False, outer_print=False)
    prob = _problem(alg, orc, alg.DoubleIntegratorGame(p=2), [1.0, 2.0, 1.0, 2.0, 0.0, 0.0, 0.9, 0.9], opts)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 1, 1, 25, 1e-7, 1e-10, 1e-10
    alg.newton_solve(prob)
    _check(alg, prob, 1e-6)


def test_e2e_unicycle_two_players(alg, orc):
    # test/problem/solver_methods.jl:100-129
    opts = alg.Options(inner_print=False, outer_print=False)
    prob = _problem(alg, orc, alg.UnicycleGame(p=2), [1.0, 2.0, 1.0, 2.0, 0.0, 0.0, 0.9, 0.9], opts)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 7, 20, 25, 1e-7, 1e-10, 1e-10
    alg.newton_solve(prob)
    _check(alg, prob, 1e-6)


def test_e2e_unicycle_two_players_constrained(alg, orc):
    # test/problem/solver_methods.jl:132-182 with its *effective* options (SURVEY.md section 4 warning: the problem is
    # built with the previous block's opts object).  Without the circle constraints of :156-160 (the C3/C5 ingredient set).
    opts = alg.Options(inner_print=False, outer_print=False)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 7, 20, 25, 1e-7, 1e-10, 1e-10
    prob = _problem(alg, orc, alg.UnicycleGame(p=2), [1.0, 2.0, 1.1, 2.0, 0.0, 0.0, 0.9, 0.9], opts, constrained=True)
    alg.newton_solve(prob)
    last = prob.stats.summary["last"][0]
    res = alg.residual(prob)
    assert np.abs(res).sum() / res.shape[1] < 1e-3
    for f in ("dyn_vio", "sta_vio", "con_vio", "opt_vio"):
        assert last[f] < 1e-3, (f, last[f])


def test_e2e_unicycle_two_players_constrained_with_circles(alg, orc):
    # test/problem/solver_methods.jl:132-182 in full: collision avoidance + control bounds + the circle constraints of :156-160
    opts = alg.Options(inner_print=False, outer_print=False)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = 7, 20, 25, 1e-7, 1e-10, 1e-10
    prob = _problem(alg, orc, alg.UnicycleGame(p=2), [1.0, 2.0, 1.1, 2.0, 0.0, 0.0, 0.9, 0.9], opts, constrained=True,
                    circles=([1.50, 0.2, 0.3], [1.25, 0.2, 0.3], [0.2, 0.2, 0.3]))
    alg.newton_solve(prob)
    last = prob.stats.summary["last"][0]
    res = alg.residual(prob)
    assert np.abs(res).sum() / res.shape[1] < 1e-3
    for f in ("dyn_vio", "sta_vio", "con_vio", "opt_vio"):
        assert last[f] < 1e-3, (f, last[f])
    # the circles are felt: player trajectories stay outside every disc
    X = prob.pdtraj.states[0]
    for xc, yc, r in zip([1.50, 0.2, 0.3], [1.25, 0.2, 0.3], [0.2, 0.2, 0.3]):
        for i in range(2):
            assert np.all((X[1:, i] - xc) ** 2 + (X[1:, i + 2] - yc) ** 2 >= r * r - 2e-3)


def _intro_problem(alg, backend, x0=None, device=0):
    # examples/intro_example.jl:10-74
    p, N, dt = 3, 20, 0.1
    model = alg.BicycleGame(p=p)
    Q = [10.0 * np.ones(4) for _ in range(p)]; R = [0.1 * np.ones(2) for _ in range(p)]
    xf = [np.array([2, 0.4, 0, 0.0]), np.array([2, 0.0, 0, 0.0]), np.array([3, -0.4, 0, 0.0])]
    uf = [np.zeros(2) for _ in range(p)]
    game_obj = alg.GameObjective(Q, R, xf, uf, N, model)
    alg.add_collision_cost(game_obj, 1.0 * np.ones(p), 5.0 * np.ones(p))
    game_con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(game_con, 0.08)
    alg.add_control_bound(game_con, 5 * np.ones(model.m), -5 * np.ones(model.m))
    alg.add_state_bound(game_con, 1, 5 * np.ones(model.n), -5 * np.ones(model.n))
    alg.add_wall_constraint(game_con, [alg.Wall([0.0, -0.4], [1.0, -0.4], [0.0, -1.0])])
    alg.add_circle_constraint(game_con, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], [0.1, 0.2, 0.3])
    if x0 is None:
        x0 = np.array([0.1, 0.0, 0.5, -0.4, 0.0, 0.7, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    return alg.GameProblem(N, dt, x0, model, alg.Options(inner_print=False, outer_print=False), game_obj, game_con,
                           backend=backend, device=device)


def test_e2e_intro_example(alg, orc):
    # examples/intro_example.jl with Options() defaults: the solve must converge to the default tolerances (eps = 1e-3)
    prob = _intro_problem(alg, orc.lib())
    alg.newton_solve(prob)
    s = prob.stats.summary
    last = s["last"][0]
    assert s["status"][0] == 0
    for f in ("dyn_vio", "sta_vio", "con_vio", "opt_vio"):
        assert last[f] < 1e-3, (f, last[f])
    X = prob.pdtraj.states[0]
    assert np.all(X[1:, 3:6][(X[1:, 0:3] > 0) & (X[1:, 0:3] < 1)] >= -0.4 - 2e-3)     # wall: y >= -0.4 while 0 < x < 1


# ---------------------------------------------------------------- iterated best response (test/problem/solver_methods.jl:185-314)
def test_ibr_mask_sizes_and_structure(alg):
    # test/core/newton_core.jl:115-160 (p = 5, N = 6): |vertical mask| = |horizontal mask| = (N-1)(2n + mi); with
    # splitted_state = false the masks are the player's opt rows + all dyn rows / x + u_i + lambda_i columns
    model = alg.DoubleIntegratorGame(p=5)
    ps = alg.ProblemSize(6, model)
    v, h = alg.vertical_indices(ps), alg.horizontal_indices(ps)
    for i in range(1, 6):
        vm = sum([v[alg.stampify("opt", i, "x", 1, k)] + v[alg.stampify("opt", i, "u", i, k - 1)] for k in range(2, 7)], []) \
            + sum([v[alg.stampify("dyn", 1, "x", 1, k)] for k in range(1, 6)], [])
        hm = sum([h[alg.stampify("x", 1, k)] for k in range(2, 7)], []) + sum([h[alg.stampify("u", i, k)] + h[alg.stampify("λ", i, k)] for k in range(1, 6)], [])
        assert len(vm) == len(set(vm)) == (ps.N - 1) * (2 * ps.n + ps.mi[i - 1]) == len(hm) == len(set(hm))


def _ibr_problem(alg, orc, model, x0, opts):
    N, dt, p = 20, 0.1, model.p
    obj = alg.GameObjective([np.ones(model.ni[i]) for i in range(p)], [0.5 * np.ones(model.mi[i]) for i in range(p)],
                            [np.zeros(model.ni[i]) for i in range(p)], [-np.ones(model.mi[i]) for i in range(p)], N, model)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    return alg.GameProblem(N, dt, x0, model, opts, obj, con, backend=orc.lib())


@pytest.mark.parametrize("which", [0, 1, 2, 3])
def test_ibr_e2e(alg, orc, which):
    model, x0, outer, inner, single, tol = [
        (alg.DoubleIntegratorGame(p=1), [1.0, 1.0, 0.0, 0.9], 1, 1, True, 1e-6),                         # :187-216
        (alg.UnicycleGame(p=1), [1.0, 1.0, 0.0, 0.9], 7, 20, True, 1e-6),                                # :218-247
        (alg.DoubleIntegratorGame(p=2), [1.0, 2.0, 1.0, 2.0, 0.0, 0.0, 0.9, 0.9], 1, 1, False, 5e-2),    # :250-279
        (alg.UnicycleGame(p=2), [1.0, 2.0, 1.0, 2.0, 0.0, 0.0, 0.9, 0.9], 7, 20, False, 5e-2)][which]     # :282-311
    opts = alg.Options(inner_print=False, outer_print=False)
    prob = _ibr_problem(alg, orc, model, x0, opts)
    opts.outer_iter, opts.inner_iter, opts.ls_iter, opts.reg_0, opts.ϵ_dyn, opts.ϵ_opt = outer, inner, 25, 1e-7, 1e-10, 1e-10
    if single:
        alg.ibr_newton_solve(prob, 1)
    else:
        alg.ibr_newton_solve(prob)
    res = alg.residual(prob)
    assert np.abs(res).sum() / res.shape[1] < tol
    assert alg.dynamics_violation(prob)[0] < 1e-6


def test_ibr_direction_moves_only_the_players_variables(orc):
    """Δtraj[horiz_mask] = -lu(jac[verti_mask, horiz_mask]) \\ res[verti_mask] (solver_methods.jl:249-251): after one best
    response of player 2 only x, u_2 and lambda_2 have moved."""
    b = orc.OracleBatch(UNI, 3, 8, 0.1, 2)
    rng = np.random.default_rng(21)
    b.set_lqr(1 + rng.random((3, 4)), 0.5 + rng.random((3, 2)), rng.random((3, 4)), rng.random((3, 2)))
    x0 = rng.random((2, b.n)); b.set_x0(x0)
    b.add_collision_avoidance(0.3 * np.ones(3)); b.add_control_bound(0.6 * np.ones(b.m), -0.6 * np.ones(b.m))
    z = rng.random((2, b.traj_len)); z[:, :b.n] = x0; b.set_traj(z)
    b.set_options(outer_iter=1, inner_iter=1, dual_reset=0)
    b.ibr_solve_player(1)
    X0, U0, L0 = b.split_traj(z); X1, U1, L1 = b.split_traj(b.get_traj())
    moved_u = np.abs(U1 - U0).max(axis=(0, 1)) > 0
    assert np.array_equal(moved_u, np.array([False, True, False, False, True, False]))     # pu[2] = {2, 5}
    assert np.array_equal(L1[:, [0, 2]], L0[:, [0, 2]]) and np.abs(L1[:, 1] - L0[:, 1]).max() > 0
    assert np.abs(X1[:, 1:] - X0[:, 1:]).max() > 0 and np.array_equal(X1[:, 0], X0[:, 0])


# ---------------------------------------------------------------- committed fixtures (tests/golden/)
def _golden_literals():
    import json, os
    def num(v):
        return float(v) if isinstance(v, str) else v
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat_literals.json")) as f:
        return json.load(f), num


def test_golden_reference_literals(alg, orc):
    """tests/golden/reference_kat_literals.json (the literal values the reference's tests hold) against the oracle / host."""
    G, num = _golden_literals()
    # control bound / state bound evaluate
    for key, add, field in (("control_bound_evaluate", "ctl", "u"), ("state_bound_evaluate", "sb", "x")):
        g = G[key]
        b = orc.OracleBatch(DI, 1, 4, 0.1, 1, d=3) if add == "sb" else orc.OracleBatch(DI, 2, 5, 0.1, 1, d=3)
        b.set_lqr(np.zeros((b.p, 6)), np.zeros((b.p, 3)), np.zeros((b.p, 6)), np.zeros((b.p, 3)))
        hi = np.array([num(v) for v in g[field + "_max"]]); lo = np.array([num(v) for v in g[field + "_min"]])
        X, U, L = b.split_traj(b.get_traj())
        if add == "ctl":
            b.add_control_bound(hi, lo); U[0, :] = g[field]
        else:
            b.add_state_bound(0, hi, lo); X[0, 1:] = g[field]
        b.set_traj(b.join_traj(X, U, L))
        vals = b.kat_evaluate_con()[0]
        rows = vals[b.p * (b.p - 1) * (b.N - 1):][:12] if add == "ctl" else vals[_ext_off(b):][:12]
        assert np.array_equal(rows[np.isfinite(rows)], np.array(g["finite_values"]))
        if "finite_indices_1based" in g:
            assert np.array_equal(np.nonzero(np.isfinite(rows))[0] + 1, g["finite_indices_1based"])
    # wall evaluate
    g = G["wall_evaluate"]; s2 = np.sqrt(2.0)
    b = orc.OracleBatch(DI, 1, 3, 0.1, 1)
    b.set_lqr(np.zeros((1, 4)), np.zeros((1, 2)), np.zeros((1, 4)), np.zeros((1, 2)))
    b.add_wall_constraint(g["x1"], g["y1"], g["x2"], g["y2"], np.array(g["xv_times_sqrt2"]) / s2, np.array(g["yv_times_sqrt2"]) / s2)
    X, U, L = b.split_traj(b.get_traj()); X[0, 1:, 0:2] = g["position"]
    b.set_traj(b.join_traj(X, U, L))
    w = b.kat_evaluate_con()[0][_ext_off(b):][:5]
    assert np.abs(w - np.array(g["values"])).sum() < g["tol_l1"]
    # layout / index sets
    assert alg.ProblemSize(40, alg.DoubleIntegratorGame(p=3)).S == G["layout"]["S"]["DoubleIntegrator p=3 d=2 N=40"]
    assert alg.ProblemSize(50, alg.UnicycleGame(p=4)).S == G["layout"]["S"]["Unicycle p=4 N=50"]
    for name, model in (("unicycle_p3", alg.UnicycleGame(p=3)), ("bicycle_p3", alg.BicycleGame(p=3)), ("double_integrator_p2_d3", alg.DoubleIntegratorGame(p=2, d=3))):
        for f in ("pu", "px", "pz"):
            assert getattr(model, f) == G["model_index_sets"][name][f], (name, f)


def test_golden_solutions_oracle(alg, orc):
    """The oracle reproduces the committed solution vectors (tests/golden/oracle_solutions.npz, made by make_golden.py)."""
    import os, sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    import make_golden
    ref = np.load(os.path.join(gdir, "oracle_solutions.npz"))
    for name in ("c2_n12", "c5", "q2", "intro"):
        got = make_golden.solve(name, alg, orc.lib())
        for k, v in got.items():
            r = ref[f"{name}.{k}"]
            if v.dtype.kind in "iu":
                assert np.array_equal(v, r), (name, k)
            else:
                assert np.allclose(v, r, rtol=1e-9, atol=1e-11), (name, k)


def test_mpc_loop_fused_equals_stepwise_oracle(alg, orc):
    # builder-defined receding-horizon loop: orc_mpc_solve == per-step newton_solve! + advance
    ids = np.arange(300, 303)
    pa = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
    pb = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
    ia, ca, sa = alg.mpc_solve(pa, 4, record_states=True)
    ib, cb, sb = alg.mpc_solve(pb, 4, record_states=True, fused=False)
    assert np.array_equal(ia, ib) and np.array_equal(ca, cb) and np.array_equal(sa, sb)
    assert ia.sum() > 0 and np.abs(sa[-1] - sa[0]).max() > 0.05


# ---- 3-D half of SURVEY 8(f) rank 3: Wall3DConstraint, CylinderConstraint, spherical collision avoidance ---------------
def _one_player_3d(orc, N=3):
    b = orc.OracleBatch(DI, 1, N, 0.1, 1, d=3)
    b.set_lqr(np.zeros((1, 6)), np.zeros((1, 3)), np.zeros((1, 6)), np.zeros((1, 3)))
    return b


def _put_position(b, xyz):
    X, U, L = b.split_traj(b.get_traj())
    X[0, 1:, :3] = xyz
    b.set_traj(b.join_traj(X, U, np.zeros_like(L)))


def test_wall3d_evaluate_literal(orc):
    # test/constraints/wall_constraint.jl:34-64; the test's (x, y, z) = X[4], X[2], X[1] become the player's position
    s2 = np.sqrt(2.0)
    p1 = np.zeros((5, 3))
    p2 = np.array([[1.0, 0, 0], [1, 0, 0], [1, 0, 1], [1, 0, 0], [1, 0, 0]])
    p3 = np.array([[1.0, 1, 0], [1, 1, 1], [1, 1, 1], [0, 1, 0], [0, 1, 1]])
    v = np.array([[0.0, 0, 1], [0, -1 / s2, 1 / s2], [-1 / s2, 0, 1 / s2], [0, 0, 1], [0, -1 / s2, 1 / s2]])
    b = _one_player_3d(orc)
    b.add_wall3d_constraint(p1, p2, p3, v)
    assert b.con_len == _ext_off(b) + 5 * (b.N - 1)
    for X6, want in (([0.0, 0.10, -12.0, 0.10, 12.0, 11.0], [0.0, -0.1 / s2, -0.1 / s2, 0.0, -0.1 / s2]),
                     ([1.0, 0.55, -12.0, 0.55, 12.0, 11.0], [1.0, 0.45 / s2, 0.45 / s2, 1.0, 0.45 / s2]),
                     ([1.0, 1.25, -12.0, 0.75, 12.0, 11.0], [0.0, 0.0, 0.0, 1.0, -0.25 / s2])):
        _put_position(b, [X6[3], X6[1], X6[0]])
        w = b.kat_evaluate_con()[0][_ext_off(b):].reshape(b.N - 1, 5)
        assert np.abs(w[0] - np.array(want)).sum() < 1e-10 and np.array_equal(w[0], w[1])
    with pytest.raises(Exception):
        orc.OracleBatch(DI, 2, 3, 0.1, 1).add_wall3d_constraint(p1, p2, p3, v)      # no third position dimension


def test_cylinder_evaluate_literal(orc):
    # test/constraints/cylinder_constraint.jl:3-22: (x, y, z) = (1, 1, 2)
    p = np.array([[1.0, 0, 1], [1, 0, 1], [1, 0, 3], [0, 1, 1], [1, 0, 2]])
    axis = [2, 2, 2, 0, 1]; l = [5.0, 2.0, 0.5, 2.0, 10.0]; r = [3.0, 2.0, 7.0, 3.0, 1.0]
    b = _one_player_3d(orc)
    b.add_cylinder_constraint(p, axis, l, r)
    _put_position(b, [1.0, 1.0, 2.0])
    c = b.kat_evaluate_con()[0][_ext_off(b):].reshape(b.N - 1, 5)
    assert np.abs(c[0] - np.array([8.0, 3.0, 0.0, 8.0, 1.0])).sum() < 1e-10
    with pytest.raises(Exception):
        b.add_cylinder_constraint(p, [0, 1, 2, 3, 0], l, r)


def test_per_player_wall3d_and_cylinder_sets(orc):
    """add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) / (game_con, i, walls::Vector{CylinderWall}) (constraints_methods.jl:208-247,
    256-299) attach the constraint to state_conlist[i] of ONE player.  The literal values of test/constraints/wall_constraint.jl:34-64 and
    cylinder_constraint.jl:3-22 for the player that carries a set, exact zeros (inert rows) for the other one; an identical entry is shared."""
    s2 = np.sqrt(2.0)
    b = orc.OracleBatch(DI, 2, 3, 0.1, 1, d=3)
    b.set_lqr(np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 6)), np.zeros((2, 3)))
    p1 = np.zeros((2, 3)); p2 = np.array([[1.0, 0, 0], [1, 0, 0]]); p3 = np.array([[1.0, 1, 0], [1, 1, 1]]); v = np.array([[0.0, 0, 1], [0, -1 / s2, 1 / s2]])
    off0 = b.con_len
    b.add_wall3d_constraint_player(1, p1, p2, p3, v)                       # player 2 (1-based): both walls
    b.add_wall3d_constraint_player(0, p1[1:], p2[1:], p3[1:], v[1:])       # player 1: the second wall only -- shared table entry
    assert b.con_len == off0 + 2 * 2 * (b.N - 1)                           # two table entries, p = 2 players, knots 2..N
    b.add_cylinder_constraint_player(0, [[1.0, 0, 1]], [2], [5.0], [3.0])
    assert b.con_len == off0 + (2 * 2 + 2 * 1) * (b.N - 1)
    X, U, L = b.split_traj(b.get_traj())
    X[0, 1:, 0:2] = 0.10; X[0, 1:, 2:4] = 0.10; X[0, 1:, 4:6] = 0.0        # both players at (x, y, z) = (0.10, 0.10, 0): wall_constraint.jl:49-53
    b.set_traj(b.join_traj(X, U, np.zeros_like(L)))
    vals = b.kat_evaluate_con()[0][off0:]
    K = b.N - 1
    w = vals[:2 * 2 * K].reshape(2, K, 2); c = vals[2 * 2 * K:].reshape(2, K, 1)
    assert np.all(w[0, :, 0] == 0.0)                                       # wall 0 does not constrain player 1
    assert np.abs(w[0, :, 1] + 0.1 / s2).max() < 1e-12 and np.abs(w[1, :, 1] + 0.1 / s2).max() < 1e-12 and np.abs(w[1, :, 0]).max() < 1e-12
    assert np.all(c[1] == 0.0)                                             # the cylinder belongs to player 1 only
    X[0, 1:, 0:2] = 1.0; X[0, 1:, 2:4] = 1.0; X[0, 1:, 4:6] = 2.0           # (1, 1, 2): cylinder_constraint.jl:9-14 -> 8
    b.set_traj(b.join_traj(X, U, np.zeros_like(L)))
    c = b.kat_evaluate_con()[0][off0 + 2 * 2 * K:].reshape(2, K, 1)
    assert np.abs(c[0] - 8.0).max() < 1e-12 and np.all(c[1] == 0.0)
    # the inert rows stay out of the residual: player 2's opt rows see no cylinder term
    lam = np.zeros((1, b.con_len)); lam[0, off0 + 2 * 2 * K:] = 1.0
    b.set_con_duals(lam, np.full((1, b.con_len), 2.0))
    r = b.residual()[0][0]
    n, mi = b.n, b.mi
    rows2 = r[(b.N - 1) * (n + mi):2 * (b.N - 1) * (n + mi)]
    assert np.all(rows2 == 0.0) and np.abs(r[:(b.N - 1) * (n + mi)]).max() > 0.0
    with pytest.raises(Exception):
        orc.OracleBatch(DI, 2, 3, 0.1, 1).add_wall3d_constraint_player(0, p1, p2, p3, v)   # no third position dimension


def test_3d_constraints_gradient_and_gauss_newton_block(orc):
    # jacobian! of both constraints equals ForwardDiff of evaluate (wall_constraint.jl test :67-70, cylinder test :25-34):
    # opt_x rows = d/dx of the AL penalty with the active set frozen; jacobian = C' I_mu C; spherical collision avoidance on
    rng = np.random.default_rng(21)
    N, p = 4, 2
    b = orc.OracleBatch(DI, p, N, 0.1, 1, d=3)
    n, K = b.n, N - 1
    b.set_lqr(np.zeros((p, 6)), np.zeros((p, 3)), np.zeros((p, 6)), np.zeros((p, 3)))
    b.add_spherical_collision_avoidance([0.4, 0.5])
    w3 = (np.array([[0.0, 0, 0.2]]), np.array([[1.0, 0, 0.2]]), np.array([[1.0, 1, 0.2]]), np.array([[0.0, 0.6, 0.8]]))
    b.add_wall3d_constraint(*w3)
    cy = (np.array([[0.5, 0.5, 0.0], [0.0, 0.4, 0.6]]), [2, 0], [1.5, 2.0], [0.45, 0.5])
    b.add_cylinder_constraint(*cy)
    z = rng.random((1, b.traj_len)); X, U, L = b.split_traj(z)
    z = b.join_traj(X, U, np.zeros_like(L)); b.set_traj(z)
    vals = b.kat_evaluate_con()[0]
    lam = rng.random((1, b.con_len)) * (rng.random((1, b.con_len)) > 0.5); mu = np.full((1, b.con_len), 3.0)
    b.set_con_duals(lam, mu)
    res0 = b.residual()[0][0]; J = b.residual_jacobian()[0]
    act = ((vals >= 0) | (lam[0] > 0))
    off = _ext_off(b)

    def cons(xk, i, k):
        """(row, value) of every constraint attached to player i at knot k."""
        out = []
        q = xk[[i, p + i, 2 * p + i]]
        j = 1 - i
        qj = xk[[j, p + j, 2 * p + j]]
        out.append((i * K + (k - 1), (0.4 + 0.5) ** 2 - ((q - qj) ** 2).sum()))                 # pairq(i, j) = i for p = 2
        p1, p2, p3, v = (a[0] for a in w3)
        inside = ((q - p1) @ (p2 - p1) > 0) * ((q - p2) @ (p1 - p2) > 0) * ((q - p3) @ (p2 - p3) > 0) * ((q - p2) @ (p3 - p2) > 0)
        out.append((off + i * K + (k - 1), ((q - p1) @ v) * inside))
        for c in range(2):
            t0 = q - cy[0][c]; ax = cy[1][c]
            valid = 0.0 < t0[ax] < cy[2][c]
            out.append((off + p * K + (i * K + k - 1) * 2 + c, (cy[3][c] ** 2 - (t0 ** 2).sum() + t0[ax] ** 2) * valid))
        return out
    for i in range(p):
        for k in range(1, N):
            for e, c in cons(X[0, k], i, k):
                assert abs(vals[e] - c) < 1e-14, (i, k, e)
    eps = 1e-6
    for i in range(p):
        for k in range(1, N):
            rows = i * K * (n + 3) + (k - 1) * (n + 3) + np.arange(n)
            cols = (k - 1) * b.b + np.arange(n)
            g = np.zeros(n); Hx = np.zeros((n, n))
            for idx, (e, c0) in enumerate(cons(X[0, k], i, k)):
                gc = np.zeros(n)
                for a in range(n):
                    xp, xm = X[0, k].copy(), X[0, k].copy(); xp[a] += eps; xm[a] -= eps
                    gc[a] = (cons(xp, i, k)[idx][1] - cons(xm, i, k)[idx][1]) / (2 * eps)
                g += gc * (lam[0, e] + 3.0 * act[e] * c0)
                Hx += 3.0 * act[e] * np.outer(gc, gc)
            assert np.allclose(res0[rows], g, atol=1e-7), (i, k)
            assert np.allclose(J[np.ix_(rows, cols)], Hx, atol=1e-6), (i, k)


def test_e2e_3d_host_builders_on_the_oracle(alg, orc):
    """Host mirror of add_spherical_collision_avoidance! / add_wall_constraint!(::Vector{Wall3D}) / (::Vector{CylinderWall})
    (constraints_methods.jl:45-81,201-284) through GameProblem on the oracle backend: two point masses swap places around a
    pillar under a ceiling; the solve converges and the solution respects the obstacles."""
    p, N, dt = 2, 20, 0.1
    model = alg.DoubleIntegratorGame(p=p, d=3)
    obj = alg.GameObjective([np.array([10.0, 10, 10, 1, 1, 1])] * p, [0.1 * np.ones(3)] * p,
                            [np.array([1.0, 0.05, 0.5, 0, 0, 0]), np.array([-1.0, -0.05, 0.5, 0, 0, 0])], [np.zeros(3)] * p, N, model)
    alg.add_collision_cost(obj, 0.6 * np.ones(p), 2.0 * np.ones(p))
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_spherical_collision_avoidance(con, 0.15)
    alg.add_control_bound(con, 4 * np.ones(model.m), -4 * np.ones(model.m))
    alg.add_wall_constraint(con, [alg.CylinderWall([0.0, 0.0, 0.0], "z", 2.0, 0.2)])
    alg.add_wall_constraint(con, [alg.Wall3D([-2.0, -2.0, 0.9], [2.0, -2.0, 0.9], [2.0, 2.0, 0.9], [0.0, 0.0, 1.0])])
    with pytest.raises(TypeError):
        alg.add_wall_constraint(con, [alg.Wall([0, 0], [1, 0], [0, 1]), alg.CylinderWall([0, 0, 0], "x", 1, 1)])
    with pytest.raises(alg.AlgamesError):
        alg.add_wall_constraint(con, [alg.CylinderWall([0, 0, 0], "x", 1, 1)])           # second cylinder set
    x0 = np.array([-1.0, 1.0, 0.02, -0.02, 0.5, 0.55, 0, 0, 0, 0, 0, 0])
    prob = alg.GameProblem(N, dt, x0, model, alg.Options(inner_print=False, outer_print=False), obj, con, backend=orc.lib())
    assert prob.batch.con_len == 2 * (N - 1) + 2 * model.m * (N - 1) + p * (N - 1) + p * (N - 1)
    alg.newton_solve(prob)
    s = prob.stats.summary
    assert s["converged"][0] == 1 and s["status"][0] == 0
    X, U, L = prob.batch.split_traj(prob.batch.get_traj())
    for i in range(2):
        assert np.all(X[0, 1:, i] ** 2 + X[0, 1:, 2 + i] ** 2 >= 0.2 ** 2 - 2e-3) and np.all(X[0, 1:, 4 + i] <= 0.9 + 1e-3)
    d = np.sqrt(((X[0, 1:, [0, 2, 4]] - X[0, 1:, [1, 3, 5]]) ** 2).sum(0))
    assert d.min() >= 0.3 - 2e-3
    # a 2-D model has no third position dimension: the build of the problem fails loudly
    con2 = alg.GameConstraintValues(alg.ProblemSize(N, alg.UnicycleGame(p=2)))
    alg.add_spherical_collision_avoidance(con2, 0.1)
    m2 = alg.UnicycleGame(p=2)
    obj2 = alg.GameObjective([np.ones(4)] * 2, [np.ones(2)] * 2, [np.zeros(4)] * 2, [np.zeros(2)] * 2, N, m2)
    with pytest.raises(alg.AlgamesError):
        alg.GameProblem(N, dt, np.zeros(8), m2, alg.Options(inner_print=False, outer_print=False), obj2, con2, backend=orc.lib())


def test_scn_and_printers(alg, orc, capsys):
    # test/problem/global_quantities.jl:30-40 (literal strings)
    assert alg.scn(1234.0) == " 1.2e+3" and alg.scn(-1234.0) == "-1.2e+3" and alg.scn(-0.1234) == "-1.2e-1" and alg.scn(0.1234) == " 1.2e-1"
    assert alg.scn(0) == " 0.0e+0" and alg.scn(-0) == " 0.0e+0" and alg.scn(0, digits=3) == " 0.000e+0"
    assert alg.scn(1234, digits=3) == " 1.234e+3" and alg.scn(1234, digits=0) == " 1e+3"
    with pytest.raises(AssertionError):
        alg.scn(1234, digits=-1)
    # opts.inner_print (solver_methods.jl:36,100): header + one line per Newton iteration, replayed from the history
    N, model = 10, alg.DoubleIntegratorGame(p=2)
    opts = alg.Options(outer_print=False)                      # inner_print defaults to true like the reference
    obj = alg.GameObjective([np.ones(4)] * 2, [0.5 * np.ones(2)] * 2, [np.zeros(4)] * 2, [-np.ones(2)] * 2, N, model)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    prob = alg.GameProblem(N, 0.1, [1.0, 2.0, 1.0, 2.0, 0.0, 0.0, 0.9, 0.9], model, opts, obj, con, backend=orc.lib())
    alg.newton_solve(prob)
    out = capsys.readouterr().out.splitlines()
    assert out[0].split() == ["out", "in", "α", "Δ", "res", "reg"]
    assert len(out) - 1 == int(prob.stats.summary["newton_iters"][0]) >= 1
    k, l, j = out[1].split()[:3]
    assert (k, l, j) == ("1", "1", "1")
    st = prob.stats
    assert len(st.res) == int(st.iter[0]) == len(st.outer_iter) == len(st.Δ_traj) == len(st.dyn_vio) == len(st.t_elap)
    assert st.res[-1] < 1e-3 and st.res[-1] < st.res[0] and st.outer_iter[0] == 1
    # plot recipes as data (src/plots/solver_plots.jl; test/plots/solver_plots.jl only checks that they run)
    X = prob.pdtraj.states[0]
    (x1, x2), (y1, y2) = alg.recipe_traj(model, X)
    assert len(x1) == 2 and np.array_equal(x1[1], X[:, 1]) and np.array_equal(y1[0], X[:, 2]) and x2 is x1
    xs, ys, labels = alg.recipe_violation(st)
    assert labels[-4:] == ["dyn", "con", "sta", "opt"] and len(xs) == len(ys) == int(st.outer_iter[-1]) + 4
    assert np.all(ys[-4] >= -10) and len(ys[-1]) == int(st.iter[0])


def test_per_player_wall_and_circle_sets(alg, orc):
    """add_wall_constraint!(game_con, i, walls) / add_circle_constraint!(game_con, i, ...) (constraints_methods.jl:121-139, 161-187):
    the constraint object is pushed to state_conval[i] only.  (1) giving every player the same set one by one equals the
    all-player call; (2) a set given to one player leaves the other players' rows inert: residual rows, constraint values and
    dual updates of the others equal those of the problem without the set."""
    p, N = 3, 6
    walls = ([0.0, 0.2], [0.5, 1.0], [1.0, 0.9], [0.5, 0.1], [0.0, 0.6], [1.0, 0.8])
    circ = ([0.5, 0.2], [0.5, 0.8], [0.3, 0.25])
    rng = np.random.default_rng(3)

    def batch(setup):
        b = orc.OracleBatch(UNI, p, N, 0.1, 2)
        b.set_x0(rng_state["x0"]); b.set_lqr(*rng_state["lqr"])
        setup(b)
        z = rng_state["z"].copy()
        b.set_traj(z)
        lam = np.full((2, b.con_len), 0.3); mu = np.full((2, b.con_len), 1.5)
        b.set_con_duals(lam, mu)
        return b

    n = 4 * p
    rng_state = dict(x0=rng.random((2, n)), z=None,
                     lqr=(1 + rng.random((2, p, 4)), 1 + rng.random((2, p, 2)), rng.random((2, p, 4)), rng.random((2, p, 2))))
    probe = orc.OracleBatch(UNI, p, N, 0.1, 2)
    z = rng.random((2, probe.traj_len)); z[:, :n] = rng_state["x0"]; rng_state["z"] = z

    everyone = batch(lambda b: (b.add_wall_constraint(*walls), b.add_circle_constraint(*circ)))
    one_by_one = batch(lambda b: [(b.add_wall_constraint_player(i, *walls), b.add_circle_constraint_player(i, *circ)) for i in range(p)])
    assert everyone.con_len == one_by_one.con_len
    assert np.array_equal(everyone.residual()[0], one_by_one.residual()[0])
    assert np.array_equal(everyone.residual_jacobian(1e-3), one_by_one.residual_jacobian(1e-3))

    plain = batch(lambda b: None)
    only1 = batch(lambda b: (b.add_wall_constraint_player(1, *walls), b.add_circle_constraint_player(1, *circ)))
    r0, r1 = plain.residual()[0], only1.residual()[0]
    S = r0.shape[1]; rows_per_player = (N - 1) * (n + 2)
    other = np.r_[0:rows_per_player, 2 * rows_per_player:S]            # opt rows of players 0 and 2, dynamics rows
    assert np.array_equal(r0[:, other], r1[:, other])
    mine = np.arange(rows_per_player, 2 * rows_per_player)
    assert np.abs(r0[:, mine] - r1[:, mine]).max() > 1e-3             # player 1 feels its walls / circles
    vals = only1.dual_penalty_update()
    lam, mu = only1.get_con_duals()
    ext = vals[:, plain.con_len:].reshape(2, 2, p, N - 1, 2) if False else vals[:, plain.con_len:]
    nw, nc = 2, 2
    wall_rows = ext[:, :p * (N - 1) * nw].reshape(2, p, N - 1, nw)
    circ_rows = ext[:, p * (N - 1) * nw:].reshape(2, p, N - 1, nc)
    assert np.all(wall_rows[:, [0, 2]] == 0.0) and np.all(circ_rows[:, [0, 2]] == 0.0)        # inert rows: value 0
    assert np.abs(circ_rows[:, 1]).min() > 0.0
    lam_ext = lam[:, plain.con_len:]
    lw = lam_ext[:, :p * (N - 1) * nw].reshape(2, p, N - 1, nw)
    assert np.all(lw[:, [0, 2]] == 0.3)                                # lambda + mu * 0, clamped: unchanged
    # a third distinct wall for player 0, then too many
    b = batch(lambda b: b.add_wall_constraint_player(1, *walls))
    b.add_wall_constraint_player(0, [9.0], [9.0], [8.0], [8.0], [0.0], [1.0])
    assert b.con_len == plain.con_len + p * (N - 1) * 3
    with pytest.raises(alg.AlgamesError):
        b.add_wall_constraint_player(0, *[np.arange(6.0) + 20 + f for f in range(6)])
    # host mirror: add_wall_constraint(game_con, i, walls) / add_circle_constraint(game_con, i, ...)
    con = alg.GameConstraintValues(alg.ProblemSize(N, alg.UnicycleGame(p=p)))
    alg.add_wall_constraint(con, 2, [alg.Wall([0.0, 0.5], [1.0, 0.5], [0.0, 1.0])])
    alg.add_circle_constraint(con, 3, [0.5], [0.5], [0.2])
    assert list(con.player_walls) == [2] and list(con.player_circles) == [3]
    with pytest.raises(alg.AlgamesError):
        alg.add_wall_constraint(con, [alg.Wall([0.0, 0.5], [1.0, 0.5], [0.0, 1.0])])


def test_pair_collision_avoidance_adders(alg, orc):
    """add_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:5-19): one CollisionConstraint per ordered pair
    with its own radius.  (1) every ordered pair added one by one with r_i + r_j is the vector form (:21-33), bit for bit;
    (2) an asymmetric subset: c = radius^2 - |px_i - px_j|^2 on the pairs that were added, the others are inert (value 0, zero
    residual / Jacobian contribution, multiplier never moves); (3) the ABI refuses a second constraint on a pair and a mix of
    planar and spherical sets."""
    import oracle as orcmod
    p, N, B = 3, 8, 3
    rng = np.random.default_rng(5)

    def batch():
        b = orcmod.OracleBatch(0, p, N, 0.1, B, d=2)
        b.set_x0(rng0["x0"]); b.set_lqr(rng0["Q"], rng0["R"], rng0["xf"], rng0["uf"])
        return b
    ni = 4
    rng0 = dict(x0=rng.random((B, 4 * p)), Q=1 + rng.random((B, p, ni)), R=0.5 + rng.random((B, p, 2)), xf=rng.random((B, p, ni)), uf=rng.random((B, p, 2)) - 0.5)
    r = np.array([0.3, 0.4, 0.55])
    a, b = batch(), batch()
    a.add_collision_avoidance(r)
    for i in range(p):
        for j in range(p):
            if i != j:
                b.add_collision_avoidance_pair(i, j, r[i] + r[j])
    z = rng.random((B, a.traj_len)); lam = rng.random((B, a.con_len)); mu = 1 + rng.random((B, a.con_len))
    for t in (a, b):
        t.set_traj(z); t.set_con_duals(lam, mu)
    assert np.array_equal(a.residual(0, 0.0)[0], b.residual(0, 0.0)[0])
    assert np.array_equal(a.residual_jacobian(1e-3), b.residual_jacobian(1e-3))
    # asymmetric subset: (0 -> 1) with radius 0.9, (2 -> 0) with radius 0.2; nothing else
    c = batch(); c.add_collision_avoidance_pair(0, 1, 0.9); c.add_collision_avoidance_pair(2, 0, 0.2)
    c.set_traj(z); c.set_con_duals(lam, mu)
    res, _ = c.residual(0, 0.0)
    e = batch(); e.add_collision_avoidance_pair(0, 1, 0.9); e.add_collision_avoidance_pair(2, 0, 0.2)
    e.set_traj(z); e.set_con_duals(lam, mu)
    vals = e.dual_penalty_update()                                          # evaluate! at pdtraj: the constraint values
    K = N - 1
    def q(i, j): return i * (p - 1) + (j if j < i else j - 1)
    X = z[:, 4 * p:].reshape(B, K, -1)[:, :, :4 * p]                     # x_2 .. x_N
    for (i, j, R_) in ((0, 1, 0.9), (2, 0, 0.2)):
        d2 = (X[:, :, i] - X[:, :, j]) ** 2 + (X[:, :, p + i] - X[:, :, p + j]) ** 2
        if vals is not None:
            assert np.allclose(vals[:, q(i, j) * K:(q(i, j) + 1) * K], R_ ** 2 - d2, rtol=1e-14, atol=1e-15)
    if vals is not None:
        for (i, j) in ((1, 0), (0, 2), (1, 2), (2, 1)):
            assert np.all(vals[:, q(i, j) * K:(q(i, j) + 1) * K] == 0.0)
    # the absent pairs contribute nothing: removing them from the vector form's residual = zeroing their multipliers AND penalties
    # cannot be expressed through the ABI, so compare with a batch whose only pairs are the same two, radii as r_i + r_j
    d = batch(); d.add_collision_avoidance_pair(0, 1, 0.9); d.add_collision_avoidance_pair(2, 0, 0.2)
    lam2 = lam.copy(); mu2 = mu.copy()
    for (i, j) in ((1, 0), (0, 2), (1, 2), (2, 1)):
        lam2[:, q(i, j) * K:(q(i, j) + 1) * K] = 7.0; mu2[:, q(i, j) * K:(q(i, j) + 1) * K] = 123.0      # must not matter
    d.set_traj(z); d.set_con_duals(lam2, mu2)
    assert np.array_equal(d.residual(0, 0.0)[0], res)
    assert np.array_equal(d.residual_jacobian(0.0), c.residual_jacobian(0.0))
    d.dual_penalty_update()
    lam_after = d.get_con_duals()[0]
    for (i, j) in ((1, 0), (0, 2), (1, 2), (2, 1)):
        assert np.all(lam_after[:, q(i, j) * K:(q(i, j) + 1) * K] == 7.0)           # inert rows: no dual ascent
    with pytest.raises(alg.AlgamesError, match="already carries"):
        c.add_collision_avoidance_pair(0, 1, 0.5)
    with pytest.raises(alg.AlgamesError, match="i != j"):
        c.add_collision_avoidance_pair(1, 1, 0.5)
    # host mirror of the reference signature (1-based players)
    model = alg.DoubleIntegratorGame(p=3)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(con, 1, 2, 0.9); alg.add_collision_avoidance(con, 3, 1, 0.2)
    assert con.collision_pairs == {(1, 2): 0.9, (3, 1): 0.2}
    with pytest.raises(alg.AlgamesError):
        alg.add_collision_avoidance(con, 0.3)


def test_arbiter_builds(alg, orc):
    """The arbiter of the parity tests is the oracle's own source with the scalar type swapped (oracle/Makefile): long double
    (liboracle_x.so) and __float128 (liboracle_q.so) behind the same double ABI.  On a well-conditioned problem the three agree to
    the double oracle's rounding, the two extended builds to long-double rounding; the discrete history is identical."""
    import oracle as orcmod
    import subprocess
    try:
        orcmod.lib("q")
    except (subprocess.CalledProcessError, OSError) as e:      # a toolchain without libquadmath: only this test needs the __float128 build
        pytest.skip(f"__float128 arbiter not buildable here: {e}")
    ids = np.arange(40, 43)
    probs = {k: alg.scenarios.make_problem("C2", ids, N=8, backend=orcmod.lib(k)) for k in ("", "x", "q")}
    for p_ in probs.values():
        alg.newton_solve(p_)
    z = {k: p_.batch.get_traj() for k, p_ in probs.items()}
    s = {k: p_.stats.summary for k, p_ in probs.items()}
    for k in ("x", "q"):
        for f in ("status", "outer_iters", "newton_iters", "ls_failures", "converged"):
            assert np.array_equal(s[k][f], s[""][f]), (k, f)
    assert np.abs(z[""] - z["q"]).max() <= 1e-10 and np.abs(z["x"] - z["q"]).max() <= 1e-13
    # residual of one fixed iterate: double rounding vs extended (the ABI rounds the result to double)
    rng = np.random.default_rng(2)
    zz = rng.random(z[""].shape); zz[:, :12] = probs[""].x0
    r = {}
    for k, p_ in probs.items():
        p_.batch.set_traj(zz); r[k] = p_.batch.residual(0, 1e-3)[0]
    assert np.abs(r["x"] - r["q"]).max() <= 4e-16 * (1 + np.abs(r["q"]).max())
    assert 0 < np.abs(r[""] - r["q"]).max() <= 1e-12 * (1 + np.abs(r["q"]).max())


def test_violation_profiles(alg, orc):
    """The .vio vectors of the four violation objects (violations.jl:5-26, 41-67, 86-114, 140-168) through orc_get_violation_profile:
    shapes (N-1, N-1, N, N), no state violation at knot 1, the maxima over the knots are what record! stores as *_vio, the
    dynamics profile of a rolled-out trajectory is zero, and a knot moved by hand shows up in exactly the two dynamics steps
    around it and in the state profile of its own knot."""
    import oracle as orcmod
    p, N, B = 3, 9, 2
    rng = np.random.default_rng(9)
    b = orcmod.OracleBatch(1, p, N, 0.1, B)                       # Unicycle
    b.set_x0(rng.random((B, 4 * p))); b.set_lqr(1 + rng.random((B, p, 4)), 0.5 + rng.random((B, p, 2)), rng.random((B, p, 4)), np.zeros((B, p, 2)))
    b.add_collision_avoidance(np.full(p, 0.8)); b.add_control_bound(np.full(2 * p, -0.01), np.full(2 * p, -0.5))       # upper bound below the initial controls: violated
    b.init_traj(game_id0=3); b.rollout()
    v = b.violation_profile()
    assert v["dyn"].shape == (B, N - 1) and v["con"].shape == (B, N - 1) and v["sta"].shape == (B, N) and v["opt"].shape == (B, N)
    assert np.abs(v["dyn"]).max() <= 1e-14 and np.all(v["sta"][:, 0] == 0.0)
    rec = b.record()
    for f in ("dyn", "con", "sta", "opt"):
        assert np.array_equal(v[f].max(axis=1), rec[f + "_vio"]), f
    assert v["con"].max() > 0 and v["sta"].max() > 0 and v["opt"].max() > 0
    # move the state of knot 5 (1-based) of game 0: dynamics steps 4 and 5 (1-based) see it, the others do not
    z = b.get_traj(); X, U, L = b.split_traj(z); X = X.copy(); X[0, 4, :] += 0.25
    b.set_traj(b.join_traj(X, U, L))
    w = b.violation_profile()
    changed = np.nonzero(w["dyn"][0] > 1e-12)[0]
    assert list(changed) == [3, 4] and np.abs(w["dyn"][1]).max() <= 1e-14
    assert np.array_equal(w["sta"][0, np.arange(N) != 4], v["sta"][0, np.arange(N) != 4])


def test_violation_profiles_extended_constraints(alg, orc):
    """The same with state bounds, a wall and a circle (the per-player constraint blocks behind the collision / control rows): the
    state profile's maximum is the record's sta_vio, and a knot pushed through the wall by hand changes the state profile at that
    knot only."""
    import oracle as orcmod
    p, N, B = 2, 7, 2
    rng = np.random.default_rng(4)
    b = orcmod.OracleBatch(2, p, N, 0.1, B)                       # Bicycle (extended instantiations)
    b.set_x0(0.2 * rng.random((B, 4 * p))); b.set_lqr(1 + rng.random((B, p, 4)), 0.5 + rng.random((B, p, 2)), rng.random((B, p, 4)), np.zeros((B, p, 2)))
    xmax = np.full(b.n, np.inf); xmin = np.full(b.n, -np.inf); xmax[0] = 0.05
    b.add_state_bound(0, xmax, xmin)
    b.add_wall_constraint([0.0], [0.5], [1.0], [0.5], [0.0], [1.0]); b.add_circle_constraint([0.5], [0.5], [0.1])
    b.init_traj(game_id0=1); b.rollout()
    v = b.violation_profile(); rec = b.record()
    assert np.array_equal(v["sta"].max(axis=1), rec["sta_vio"]) and np.all(v["sta"][:, 0] == 0.0)
    z = b.get_traj(); X, U, L = b.split_traj(z); X = X.copy()
    X[1, 3, 0] = 0.5; X[1, 3, p] = 0.9                              # player 0 of game 1 at knot 4: x = 0.5, y = 0.9, above the wall y = 0.5
    b.set_traj(b.join_traj(X, U, L))
    w = b.violation_profile()
    assert np.array_equal(w["sta"][0], v["sta"][0])
    assert np.array_equal(w["sta"][1, np.arange(N) != 3], v["sta"][1, np.arange(N) != 3])
    assert abs(w["sta"][1, 3] - max(0.5 - 0.05, 0.9 - 0.5)) < 1e-12  # the larger of the state-bound and the wall violation
