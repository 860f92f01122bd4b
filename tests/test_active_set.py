"""Active-set / null-space analysis (src/active_set/*.jl), host mirror algames.jl_amd/active_set.py.
The literals are those of the reference's tests (test/active_set/active_set_stamp.jl, active_set_core.jl, active_set_methods.jl).
CPU tests run the analysis on top of the oracle's ABI (it is test infrastructure); the `gpu` test runs it on the HIP path."""
import numpy as np
import pytest


def test_cstamp_validity_truth_table(alg):
    A = alg.active_set
    N, p = 10, 4                                              # test/active_set/active_set_stamp.jl:3-36
    table = [(("v", 1, 2, 3), True), (("v", 1, 1, 3), False), (("v", 1, 3, 3), True), (("v", 1, 5, 3), False), (("v", 0, 3, 3), False),
             (("v", 1, 3, 1), False), (("v", 3, 1, 3), False), (("v", 1, 2, 11), False),
             (("h", 1, 3, 3), True), (("h", 2, 2, 3), False), (("h", 3, 1, 3), True), (("h", 3, 1, 1), False), (("h", 3, 1, 11), False),
             (("h", 5, 1, 10), False), (("h", 4, 1, 10), True)]
    for (dim, i, j, k), want in table:
        assert A.valid_c(A.CStamp(dim, "col", i, j, k), N, p) is want, (dim, i, j, k)
    s0, s1 = A.CStamp("h", "col", 4, 1, 10), A.CStamp()      # :38-47
    assert s1 != s0
    s2 = A.stampify_c("v", "col", 1, 2, 10)
    s1 = A.CStamp("v", "col", 1, 2, 10)
    assert s2 == s1 and hash(s2) == hash(s1)


def test_active_set_core_sizes_and_index_maps(alg):
    A = alg.active_set
    N, p = 10, 3                                              # test/active_set/active_set_core.jl:3-9
    ps = alg.ProblemSize(N, alg.UnicycleGame(p=p))
    core = A.ActiveSetCore(ps)
    assert core.Sv == ps.S + (N - 1) * p * (p - 1) // 2 and core.Sh == ps.S + (N - 1) * p * (p - 1)
    assert core.res.shape == (core.Sv,) and core.jac.shape == (core.Sv, core.Sh)
    # active_set_core.jl:113-121 / :144-152: extra rows / columns follow the solver's, ordered by (k, i, j)
    assert core.verti_inds[A.CStamp("v", "col", 1, 2, 2)] == [ps.S + 1] and core.verti_inds[A.CStamp("v", "col", 2, 3, 2)] == [ps.S + 3]
    assert core.verti_inds[A.CStamp("v", "col", 1, 2, 3)] == [ps.S + 4] and core.verti_inds[A.CStamp("v", "col", 2, 3, N)] == [core.Sv]
    assert core.horiz_inds[A.CStamp("h", "col", 1, 2, 2)] == [ps.S + 1] and core.horiz_inds[A.CStamp("h", "col", 2, 1, 2)] == [ps.S + 3]
    assert core.horiz_inds[A.CStamp("h", "col", 3, 2, N)] == [core.Sh]
    assert core.vmask == list(range(1, core.Sv + 1)) and core.hmask == list(range(1, core.Sh + 1))


def test_active_rule_literals(alg):
    A = alg.active_set
    N, p = 10, 3                                              # test/active_set/active_set_methods.jl:3-34
    con = alg.GameConstraintValues(alg.ProblemSize(N, alg.UnicycleGame(p=p)))
    alg.add_collision_avoidance(con, 1.0)
    assert A.active(con, A.stampify_c("v", "col", 1, 2, 12)) == [0]
    cv = A.collision_convals(con)[(1, 2)]
    cv.λ[0], cv.λ[1], cv.λ[2] = 10.0, -20.0, -20.0
    cv.vals[2], cv.vals[3], cv.vals[4] = 1.0, 2.0, -2.0
    A.update_active_set(con, tol=0.0)                         # Altro.update_active_set!(conval, Val(0.0))
    assert [A.active(con, A.stampify_c("v", "col", 1, 2, k))[0] for k in (2, 3, 4, 5, 6)] == [1, 1, 1, 1, 0]


def _problem(alg, backend, N=10, p=3, seed=0):
    rng = np.random.default_rng(seed)
    model = alg.UnicycleGame(p=p)
    Q = [rng.random(4) for _ in range(p)]; R = [rng.random(2) for _ in range(p)]
    xf = [(i + 1) * np.ones(4) for i in range(p)]; uf = [2.0 * (i + 1) * np.ones(2) for i in range(p)]
    obj = alg.GameObjective(Q, R, xf, uf, N, model)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(con, 1.0)
    return alg.GameProblem(N, 0.1, rng.random(model.n), model, alg.Options(inner_print=False, outer_print=False), obj, con, backend=backend)


def _masks_and_nullspace(alg, prob):
    A = alg.active_set
    ps = prob.probsize
    N, p, S = ps.N, ps.p, ps.S
    core = A.ActiveSetCore(ps)
    # masks (test/active_set/active_set_methods.jl:37-88): everything active at the zero trajectory with radius 1 ...
    A.update_active_set(prob.game_con, np.zeros((N, ps.n)), tol=0.0)
    A.active_vertical_mask(core, prob.game_con); A.active_horizontal_mask(core, prob.game_con)
    assert core.vmask == list(range(1, S + (N - 1) * p * (p - 1) // 2 + 1)) and core.hmask == list(range(1, S + (N - 1) * p * (p - 1) + 1))
    # ... nothing active on a trajectory spread over 1e3
    far = 1e3 * np.random.default_rng(100).random((N, ps.n))
    A.update_active_set(prob.game_con, far, tol=0.0)
    A.active_vertical_mask(core, prob.game_con); A.active_horizontal_mask(core, prob.game_con)
    assert core.vmask == list(range(1, S + 1)) and core.hmask == list(range(1, S + 1))
    # nullspace (:91-116): x0 in [0,1]^n with radius 1 -> every pair is inside the radius at the initial (x_k = 1e-8 scale) trajectory
    A.residual(core, prob); A.residual_jacobian(core, prob)
    assert np.abs(core.res[:S] - alg.residual(prob)[0]).max() == 0.0
    assert np.abs(core.jac[:S, :S] - alg.residual_jacobian(prob, 0.0)[0]).max() == 0.0
    assert np.all(core.jac[S:] == 0.0) and np.abs(core.jac[:S, S:]).max() > 0.0        # the reference's dead branch: no constraint rows
    A.update_nullspace(core, prob)
    Sh = S + (N - 1) * p * (p - 1)
    assert core.null.mat.shape == (Sh, (N - 1) * p) and len(core.null.vec) == (N - 1) * p and len(core.null.vec[0]) == Sh
    for v in core.null.vec[:3]:
        assert abs(np.mean(np.abs(v)) - 1.0) < 1e-12
    # with the constraint rows really written (what the dead branch was meant to do) the null vectors annihilate the active system
    A.update_nullspace(core, prob, constraint_rows=True)
    d = core.jac[np.ix_(np.asarray(core.vmask) - 1, np.asarray(core.hmask) - 1)]
    assert core.null.mat.shape[1] == d.shape[1] - np.linalg.matrix_rank(d)
    assert np.abs(d @ core.null.mat).max() < 1e-8 * max(1.0, np.abs(d).max())
    return core


def test_masks_and_nullspace_on_the_oracle_abi(alg, orc):
    prob = _problem(alg, orc.lib())
    prob.batch.init_traj(0)                                   # pdtraj at the 1e-8-scale initial guess, x_k = rollout
    _masks_and_nullspace(alg, prob)


@pytest.mark.gpu
def test_masks_and_nullspace_on_the_hip_path(alg, orc):
    pg, po = _problem(alg, None), _problem(alg, orc.lib())
    pg.batch.init_traj(0); po.batch.init_traj(0)
    cg, co = _masks_and_nullspace(alg, pg), _masks_and_nullspace(alg, po)
    assert np.abs(cg.jac - co.jac).max() <= 1e-12 * np.abs(co.jac).max()
    # and after a solve: the active set at the generalized Nash equilibrium, multipliers from the device
    alg.newton_solve(pg); alg.newton_solve(po)
    A = alg.active_set
    for prob, core in ((pg, cg), (po, co)):
        A.update_nullspace(core, prob, constraint_rows=True)
    assert cg.vmask == co.vmask and cg.hmask == co.hmask
    assert cg.null.mat.shape == co.null.mat.shape


def test_active_set_on_a_pair_only_problem(alg, orc):
    """constraints added with add_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:5-19) must reach the
    active-set analysis too -- own radius per ordered pair, absent pairs inert; the spherical form measures the 3-D distance."""
    A = alg.active_set
    N, p = 6, 3
    model = alg.UnicycleGame(p=p)
    con = alg.GameConstraintValues(alg.ProblemSize(N, model))
    alg.add_collision_avoidance(con, 1, 2, 0.5)
    alg.add_collision_avoidance(con, 3, 1, 2.0)
    cvs = A.collision_convals(con)
    assert sorted(cvs) == [(1, 2), (3, 1)] and cvs[(1, 2)].radius == 0.5 and cvs[(3, 1)].radius == 2.0
    X = np.zeros((N, model.n)); X[:, 0] = 1.0                  # player 1 at (1, 0), players 2 and 3 at the origin
    A.update_active_set(con, X, tol=0.0)
    assert np.allclose(cvs[(1, 2)].vals, 0.25 - 1.0) and np.allclose(cvs[(3, 1)].vals, 4.0 - 1.0)
    assert A.active(con, A.stampify_c("v", "col", 1, 2, 3)) == [0] and A.active(con, A.stampify_c("h", "col", 3, 1, 3)) == [1]
    assert A.active(con, A.stampify_c("h", "col", 2, 1, 3)) == [0]           # no such constraint
    assert np.allclose(cvs[(3, 1)].jac[0, [2, 2 + p]], [2.0, 0.0]) and np.allclose(cvs[(3, 1)].jac[0, [0, p]], [-2.0, 0.0])
    # the augmented system of a pair-only problem carries the pair's multiplier columns
    obj = alg.GameObjective([np.ones(4)] * p, [np.ones(2)] * p, [np.zeros(4)] * p, [np.zeros(2)] * p, N, model)
    prob = alg.GameProblem(N, 0.1, np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), model, alg.Options(inner_print=False, outer_print=False), obj, con, backend=orc.lib())
    prob.batch.init_traj(0)
    core = A.ActiveSetCore(prob.probsize)
    A.update_active_set(con, np.tile(prob.x0 if prob.x0.ndim == 1 else prob.x0[0], (N, 1)), tol=0.0)
    A.active_vertical_mask(core, con); A.active_horizontal_mask(core, con)
    S = prob.probsize.S
    assert len(core.hmask) == S + (N - 1)                       # only (3, 1) is inside its radius: one multiplier column per knot
    A.residual_jacobian(core, prob)
    assert np.abs(core.jac[:S, S:]).max() > 0.0
    # spherical: the third position coordinate counts
    m3 = alg.DoubleIntegratorGame(p=2, d=3)
    c3 = alg.GameConstraintValues(alg.ProblemSize(N, m3))
    alg.add_spherical_collision_avoidance(c3, 1, 2, 1.0)
    X3 = np.zeros((N, m3.n)); X3[:, 2 * 2] = 3.0               # z of player 1 (pz[1][3] = 1 + 2 p)
    A.update_active_set(c3, X3, tol=0.0)
    assert np.allclose(A.collision_convals(c3)[(1, 2)].vals, 1.0 - 9.0)
