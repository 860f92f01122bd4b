"""`opts.f_init` as a caller-supplied generator (src/struct/options.jl:11; init_traj!, src/struct/primal_dual_traj.jl:29-44, called at
src/problem/solver_methods.jl:13).  The device generates the default `rand` itself; any other generator is applied on the host
(`host.init_traj_host`) and the solve runs on the resulting guess.  CPU tests run on the oracle's ABI (test infrastructure); the
`gpu` test compares the HIP path with the oracle on the same generator."""
import numpy as np
import pytest


def _prob(alg, backend, ids=(7, 8, 9), **kw):
    return alg.scenarios.make_problem("C5", np.arange(ids[0], ids[0] + len(ids)), backend=backend, **kw)


def test_f_init_zeros_and_ones_are_the_reference_guess(alg, orc):
    p = _prob(alg, orc.lib()); b = p.batch
    calls = []
    def ones(size): calls.append(size); return np.ones(size)
    p.opts.f_init, p.opts.amplitude_init = ones, 1e-3
    alg.host.init_traj_host(p)
    X, U, L = b.split_traj(b.get_traj(0))
    assert np.array_equal(X[:, 0], b.get_x0())                                  # set_state!(pdtraj.pr[1], x0)
    assert np.all(U == 1e-3) and np.all(L == 1e-3) and np.all(X[:, 1:] == 1e-3)
    # the reference's call order, game by game: N knots of n + m, then p x (N - 1) dual vectors of n
    per_game = [b.n + b.m] * b.N + [b.n] * (b.p * (b.N - 1))
    assert calls == per_game * X.shape[0]
    # the solve on that guess = an explicit warm start from the same controls / duals
    alg.newton_solve(p)
    q = _prob(alg, orc.lib()); qb = q.batch
    Xq, Uq, Lq = qb.split_traj(qb.get_traj(0))
    Uq[:], Lq[:] = 1e-3, 1e-3
    qb.set_traj(qb.join_traj(Xq, Uq, Lq))
    alg.newton_solve(q, init=False)
    assert np.array_equal(b.get_traj(0), qb.get_traj(0))
    for f in ("status", "newton_iters", "outer_iters", "converged"):
        assert np.array_equal(p.stats.summary[f], q.stats.summary[f])
    assert p.stats.summary["converged"].all()


def test_f_init_with_shift_copies_the_tail_and_draws_the_rest(alg, orc):
    p = _prob(alg, orc.lib()); b = p.batch
    alg.newton_solve(p)                                                           # default rand on the "device", then a shifted warm start
    X0, U0, L0 = b.split_traj(b.get_traj(0))
    p.opts.f_init, p.opts.amplitude_init, p.opts.shift = np.zeros, 1.0, 3
    alg.host.init_traj_host(p)
    X, U, L = b.split_traj(b.get_traj(0))
    N, s = b.N, 3
    assert np.array_equal(U[:, :N - 1 - s], U0[:, s:]) and np.all(U[:, N - 1 - s:] == 0.0)     # k + s <= N - 1 copied, the rest drawn
    assert np.array_equal(L[:, :, :N - 1 - s], L0[:, :, s:]) and np.all(L[:, :, N - 1 - s:] == 0.0)
    assert np.array_equal(X[:, 1:N - s], X0[:, 1 + s:]) and np.all(X[:, N - s:] == 0.0) and np.array_equal(X[:, 0], b.get_x0())


def test_f_init_generator_on_a_sharded_problem(alg, orc):
    rng = np.random.default_rng(5)
    draws = rng.standard_normal(100000); pos = [0]
    def randn(size): v = draws[pos[0]:pos[0] + size]; pos[0] += size; return v
    whole = _prob(alg, orc.lib(), ids=(20, 21, 22, 23)); whole.opts.f_init = randn
    alg.newton_solve(whole)
    pos[0] = 0
    parts = alg.scenarios.make_problem("C5", np.arange(20, 24), backend=orc.lib(), devices=[0, 0]); parts.opts.f_init = randn
    for q in parts.shards: q.opts.f_init = randn
    alg.newton_solve(parts)
    assert np.array_equal(whole.batch.get_traj(0), parts.get_traj(0))


@pytest.mark.gpu
def test_f_init_generator_parity(alg, orc):
    def gen():
        rng = np.random.default_rng(11)
        return lambda size: rng.standard_normal(size)
    g, o = _prob(alg, None, ids=(40, 41, 42, 43, 44, 45)), _prob(alg, orc.lib(), ids=(40, 41, 42, 43, 44, 45))
    for q in (g, o):
        q.opts.f_init, q.opts.amplitude_init = gen(), 1e-2
        alg.newton_solve(q)
    for f in ("status", "newton_iters", "outer_iters", "ls_failures", "converged"):
        assert np.array_equal(g.stats.summary[f], o.stats.summary[f]), f
    assert np.abs(g.batch.get_traj(0) - o.batch.get_traj(0)).max() <= 1e-8
