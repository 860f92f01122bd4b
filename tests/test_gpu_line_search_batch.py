"""Line search of the team kernels and of the one-wavefront unicycle kernels: step sizes tried in groups (trial_norms_multi, algames_assemble.hpp) against the one-by-one search of
solver_methods.jl:105-125 -- same binary, alg_set_line_search_groups(h, 0 / 1).  The group pass reproduces the norms of the one-by-one trials bit for bit
(the device counts disagreements in alg_game_stats.reserved), so iterates, step sizes and iteration counts are identical."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(alg, cfg, games, waves, multi):
    prob = alg.scenarios.make_problem(cfg, np.arange(games))
    prob.batch.set_line_search_groups(multi)                    # alg_set_line_search_groups: the two searches in the same binary
    assert prob.batch.get_line_search_groups() == bool(multi)
    prob.batch.set_waves_per_game(waves)
    return prob


@pytest.mark.parametrize("waves", [4, 1])
def test_receding_horizon_loop_is_bitwise_the_one_by_one_search(alg, waves):
    out = []
    for multi in (False, True):
        prob = _problem(alg, "C5", 16, waves, multi)
        it, cv, states = alg.mpc_solve(prob, 40, record_states=True)
        out.append((it.copy(), cv.copy(), states.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2].view(np.uint64), out[1][2].view(np.uint64))
    assert out[0][0].sum() > 16 * 40          # the loops did iterate


@pytest.mark.parametrize("waves", [4, 1])
def test_step_by_step_solves_same_trials_and_no_norm_disagreement(alg, waves):
    G, steps = 8, 25
    hist = []
    for multi in (False, True):
        prob = _problem(alg, "C5", G, waves, multi); b = prob.batch
        rows = []
        for t in range(steps):
            if t == 1:
                prob.opts.shift, prob.opts.dual_reset = 1, False; prob._sync_options()
            b.newton_solve_async(init=True, game_id0=prob.game_id0 + t * 1000003)
            st = b.get_stats()
            assert int(st["reserved"].sum()) == 0                       # every group-pass norm equalled the ordinary pass's, bit for bit
            for g in range(G):
                h = b.get_history(g)
                rows.append((t, g, h["ls_j"].copy(), h["alpha"].copy(), h["res"].copy()))
            traj = b.get_traj().copy() if hasattr(b, "get_traj") else None
            rows.append(traj)
            b.mpc_advance()
        hist.append(rows)
    deep = 0
    for a, c in zip(hist[0], hist[1]):
        if a is None or isinstance(a, np.ndarray):
            if a is not None: assert np.array_equal(a.view(np.uint64), c.view(np.uint64))
            continue
        assert np.array_equal(a[2], c[2]) and np.array_equal(a[3].view(np.uint64), c[3].view(np.uint64)) and np.array_equal(a[4].view(np.uint64), c[4].view(np.uint64))
        deep += int((a[2] >= 3).sum())
    assert deep > 20              # searches that went past the second step size (where the groups start) did occur


@pytest.mark.parametrize("cfg,games,waves", [("C3", 64, 2), ("C3", 32, 4), ("C2", 64, 4), ("C5", 64, 4), ("C5", 64, 1), ("C3", 64, 1)])
def test_perturbed_solves_bitwise(alg, cfg, games, waves):
    res = []
    for multi in (False, True):
        prob = _problem(alg, cfg, games, waves, multi)
        rng = np.random.default_rng(5)
        x0 = prob.batch.get_x0(); prob.batch.set_x0(x0 + 0.3 * rng.standard_normal(x0.shape))
        alg.newton_solve(prob)
        st = prob.batch.get_stats()
        assert int(st["reserved"].sum()) == 0
        res.append((prob.batch.get_traj().copy(), st["newton_iters"].copy(), st["ls_failures"].copy()))
    assert np.array_equal(res[0][0].view(np.uint64), res[1][0].view(np.uint64))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
