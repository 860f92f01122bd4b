"""Straggler hand-off (alg_set_handoff; no reference counterpart -- the reference solves one game at a time, solver_methods.jl:5-65):
the budgeted one-wavefront kernel parks the games that need more than K inner iterations, the team kernel resumes them from the state
in their arena chunk and continues the same outer / inner loops.  What has to hold: every game is finished (no PARKED status left), the
games that never parked are bit-identical to the plain solve, the parked games take the discrete path of the plain solve and of the CPU
oracle (iteration counts, outer iterations, convergence flags), and their trajectories agree with both to the team kernels' rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALG_STATUS_PARKED = 3


def _perturbed(alg, cfg, games, spread, backend=None, seed=5):
    prob = alg.scenarios.make_problem(cfg, np.arange(games), backend=backend)
    rng = np.random.default_rng(seed)
    x0 = prob.x0.copy(); npos = 2 * prob.model.p
    x0[:, :npos] += rng.uniform(-spread, spread, (games, npos))
    prob.x0 = x0
    prob.batch.set_x0(x0)
    return prob


def _solve(alg, prob, handoff=0):
    b = prob.batch
    if hasattr(b, "set_waves_per_game"):
        b.set_waves_per_game(1)
    if handoff:
        b.set_handoff(handoff)
    alg.newton_solve(prob)
    return prob.stats.summary.copy(), b.get_traj().copy(), b.get_con_duals()


@pytest.mark.parametrize("cfg,games,spread,budget", [("C2", 512, 0.3, 16), ("C5", 256, 0.3, 14), ("C3", 128, 0.2, 12)])
def test_parked_games_take_the_plain_solve_s_path_and_match_the_oracle(alg, orc, cfg, games, spread, budget):
    plain = _perturbed(alg, cfg, games, spread)
    s0, z0, (lam0, mu0) = _solve(alg, plain)
    ho = _perturbed(alg, cfg, games, spread)
    s1, z1, (lam1, mu1) = _solve(alg, ho, handoff=budget)
    k, parked = ho.batch.get_handoff()
    assert k == budget
    # The budget counts inner iterations BEGUN (solver_methods.jl:38-44): every one of them makes a record!, the last one of an outer
    # iteration usually ends at the optimality test without a linear solve, and newton_solve! adds one final record -- a game parks exactly
    # when the plain solve made more than budget + 1 records.
    over = (s0["records"] - 1) > budget
    assert parked == int(over.sum()) > 0, (parked, int(over.sum()))
    assert not (s1["status"] == ALG_STATUS_PARKED).any()
    # games that finished inside the budget never left the one-wavefront kernel: the same bits as the plain solve
    early = ~over
    assert early.sum() > 0
    assert np.array_equal(z0[early].view(np.uint64), z1[early].view(np.uint64))
    for f in ("newton_iters", "outer_iters", "converged", "status", "records", "ls_failures"):
        assert np.array_equal(s0[f][early], s1[f][early]), f
    # the parked games: same discrete path as the plain solve wherever the plain solve's own path is the oracle's (a handful of games of
    # a +-0.3 batch amplify rounding by 1e8 and more: the two double programs part ways there already, tests/probes/fuzz_sensitivity.py)
    cpu = _perturbed(alg, cfg, games, spread, backend=orc.lib())
    alg.newton_solve(cpu)
    sc, zc = cpu.stats.summary, cpu.batch.get_traj()
    same_plain = np.ones(games, bool)
    for f in ("newton_iters", "outer_iters", "converged", "status"):
        same_plain &= s0[f] == sc[f]
    late = ~early
    ok = late & same_plain
    assert ok.sum() >= 0.9 * late.sum(), (int(ok.sum()), int(late.sum()))          # the amplifying games are the exception
    for f in ("newton_iters", "outer_iters", "converged", "status", "ls_failures"):
        assert np.array_equal(s1[f][ok], sc[f][ok]), (f, np.nonzero(ok & (s1[f] != sc[f]))[0][:8])
    conv = ok & (sc["converged"] == 1)
    assert conv.sum() > 0
    err = np.abs(z1[conv] - zc[conv]).max(axis=1)
    assert np.median(err) < 1e-9 and err.max() < 1e-6, (np.median(err), err.max())
    assert np.abs(mu1[conv] - cpu.batch.get_con_duals()[1][conv]).max() == 0.0      # penalties: bit-equal (powers of rho_increase)


def test_homogeneous_batch_parks_nothing_and_is_bit_identical(alg):
    a = alg.scenarios.make_problem("C2", np.arange(256)); b = alg.scenarios.make_problem("C2", np.arange(256))
    s0, z0, d0 = _solve(alg, a)
    s1, z1, d1 = _solve(alg, b, handoff=16)
    assert b.batch.get_handoff() == (16, 0)
    assert np.array_equal(z0.view(np.uint64), z1.view(np.uint64))
    assert np.array_equal(d0[0].view(np.uint64), d1[0].view(np.uint64)) and np.array_equal(d0[1].view(np.uint64), d1[1].view(np.uint64))
    assert s0.tobytes() != b"" and np.array_equal(s0["newton_iters"], s1["newton_iters"])


def test_budget_one_parks_every_game_and_the_team_kernel_finishes_them(alg, orc):
    """The extreme split: one inner iteration on the one-wavefront kernel, everything else on the team kernel -- including the dual /
    penalty update the fused kernel leaves to its next record! when a game parks at the top of an outer iteration."""
    for budget in (1, 4):
        p = alg.scenarios.make_problem("C2", np.arange(64))
        s1, z1, (lam1, mu1) = _solve(alg, p, handoff=budget)
        assert p.batch.get_handoff() == (budget, 64)
        cpu = alg.scenarios.make_problem("C2", np.arange(64), backend=orc.lib()); alg.newton_solve(cpu)
        sc = cpu.stats.summary
        for f in ("newton_iters", "outer_iters", "converged", "status", "records"):
            assert np.array_equal(s1[f], sc[f]), f
        assert np.abs(z1 - cpu.batch.get_traj()).max() < 1e-8
        lc, mc = cpu.batch.get_con_duals()
        assert np.array_equal(mu1, mc) and np.abs(lam1 - lc).max() <= 1e-6 * (1 + np.abs(lc).max())


def test_handoff_is_refused_where_no_kernel_pair_exists(alg):
    p = alg.scenarios.make_problem("C2", np.arange(8), p=2)
    with pytest.raises(Exception):
        p.batch.set_handoff(8)
    p.batch.set_handoff(0)                                     # switching it off is always accepted
    assert p.batch.get_handoff() == (0, 0)
