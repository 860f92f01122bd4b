"""Full-size parity: every game of every BASELINE configuration (BASELINE.json `configs`) solved by the HIP path (through
the C ABI) and by the CPU oracle (OpenMP over games, all host cores) on the same seeded scenarios -- not a slice.

  C2  3-player DoubleIntegrator N=40, all 4096 scenarios
  C3  4-player Unicycle N=50, all 1024 scenarios
  C4  one 8192-scenario shard (rank 3 of 8) of the 65 536-scenario batch
  C5  receding-horizon loop, 64 seeds (the per-GPU share of 512 over 8 GPUs) x 200 MPC steps, fused kernel

Tolerances are those of tests/test_gpu_parity.py: identical iteration / line-search counts and penalties,
primal trajectories <= 1e-8, multipliers <= 1e-6 relative, final statistics within 1e-9.
The oracle side needs a few seconds per test on the GPU box's host cores."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1800)]


def _solve_both(alg, orc, cfg, ids):
    pg = alg.scenarios.make_problem(cfg, ids)
    po = alg.scenarios.make_problem(cfg, ids, backend=orc.lib())
    alg.newton_solve(pg)
    alg.newton_solve(po)
    return pg, po


def _assert_full_parity(pg, po):
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        bad = np.nonzero(sg[f] != so[f])[0]
        assert bad.size == 0, (f, bad[:8], sg[f][bad[:8]], so[f][bad[:8]])
    for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
        assert np.allclose(sg["last"][f], so["last"][f], rtol=1e-9, atol=1e-9), f
    Xg, Ug, Lg = pg.batch.split_traj(pg.batch.get_traj())
    Xo, Uo, Lo = po.batch.split_traj(po.batch.get_traj())
    ex = np.abs(Xg - Xo).reshape(len(Xg), -1).max(axis=1)
    eu = np.abs(Ug - Uo).reshape(len(Ug), -1).max(axis=1)
    assert ex.max() <= 1e-8 and eu.max() <= 1e-8, (int(ex.argmax()), ex.max(), int(eu.argmax()), eu.max())
    assert np.abs(Lg - Lo).max() <= 1e-6 * max(1.0, np.abs(Lo).max())
    (lg, mg), (lo, mo) = pg.batch.get_con_duals(), po.batch.get_con_duals()
    assert np.array_equal(mg, mo)
    assert np.abs(lg - lo).max() <= 1e-6 * max(1.0, np.abs(lo).max())
    return ex.max(), eu.max()


def test_c2_all_4096_games_against_the_oracle(alg, orc):
    pg, po = _solve_both(alg, orc, "C2", np.arange(4096))
    _assert_full_parity(pg, po)
    s = pg.stats.summary
    assert np.all(s["converged"] == 1) and np.all(s["status"] == 0)


def test_c3_all_1024_games_against_the_oracle(alg, orc):
    pg, po = _solve_both(alg, orc, "C3", np.arange(1024))
    _assert_full_parity(pg, po)
    assert np.all(pg.stats.summary["converged"] == 1)


def test_c4_one_8192_game_shard_against_the_oracle(alg, orc):
    lo, hi = alg.scenarios.shard_range(65536, 3, 8)
    assert hi - lo == 8192
    pg, po = _solve_both(alg, orc, "C4", np.arange(lo, hi))
    _assert_full_parity(pg, po)
    assert np.all(pg.stats.summary["converged"] == 1)


def test_c5_receding_horizon_64_seeds_x_200_steps_against_the_oracle(alg, orc):
    """The stated C5 shape per GPU (512 seeds over 8 GPUs = 64 seeds, 200 MPC steps): 12 800 warm-started newton_solve!s.

    The closed loop feeds every solve's output into the next one and the scenario contains hard solves (vehicles crossing:
    20-140 Newton iterations, failed line searches), where a 1e-10 difference is amplified until a discrete decision flips;
    free-running loops of the two implementations therefore separate after a few dozen steps for a third of the seeds
    (scratch/c5_loop_probe.py).  SURVEY.md 8(d) defines C5 parity per individual solve, so the comparison is lock-step: the HIP
    path drives the loop, and before every MPC step the oracle receives its complete solver state (x0, warm-start trajectory,
    multipliers, penalties); both then run that solve.  Bounds (measured: 7 of 12 800 solves differ in their counts, all of
    them >= 14-iteration solves at steps 10-13):
      * the first record! of every solve (same inputs, pure arithmetic) agrees to 1e-9 relative / 1e-12 absolute;
      * >= 99.8 % of the solves have identical outer / Newton / line-search-failure counts, every exception is a solve of
        >= 10 Newton iterations;
      * converged solves with identical counts, <= 10 Newton iterations and no failed line search agree to 1e-8 in the
        trajectory, all other solves with identical counts to 1e-4 (measured 1.3e-5 on a 33-iteration solve);
    and the fused loop kernel (one launch, alg_mpc_solve) reproduces the step-wise launches."""
    ids = np.arange(128, 192)
    T = 200
    pg = alg.scenarios.make_problem("C5", ids)
    po = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
    bg, bo = pg.batch, po.batch
    bg.mpc_totals(reset=True)
    states = [bg.get_x0()]
    n_solves = n_diff = 0
    worst_short = worst_all = worst_first = 0.0
    try:
        for t in range(T):
            if t == 1:                                              # later solves: shift = 1, dual_reset = false
                for p_ in (pg, po):
                    p_.opts.shift, p_.opts.dual_reset = 1, False
                    p_._sync_options()
            z = bg.get_traj(0)
            lam, mu = bg.get_con_duals()
            bo.set_x0(z[:, :bg.n].copy()); bo.set_traj(z, 0); bo.set_con_duals(lam, mu)
            gid = pg.game_id0 + t * 1000003
            sg = bg.newton_solve(init=True, game_id0=gid)
            so = bo.newton_solve(init=True, game_id0=gid)
            same = np.ones(len(ids), dtype=bool)
            for f in ("status", "outer_iters", "newton_iters", "ls_failures", "converged", "records"):
                same &= sg[f] == so[f]
            hard = (sg["newton_iters"] >= 10) & (so["newton_iters"] >= 10)
            assert np.all(same | hard), (t, np.nonzero(~(same | hard))[0], sg["newton_iters"], so["newton_iters"])
            n_solves += len(ids); n_diff += int((~same).sum())
            err = np.abs(bg.get_traj(0) - bo.get_traj(0)).max(axis=1)
            short = same & (sg["newton_iters"] <= 10) & (sg["ls_failures"] == 0) & (sg["converged"] == 1)
            worst_short = max(worst_short, float(err[short].max(initial=0.0)))
            worst_all = max(worst_all, float(err[same].max(initial=0.0)))
            for g in range(len(ids)):
                hg, ho = bg.get_history(g, 1), bo.get_history(g, 1)
                for f in ("res", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
                    d = abs(hg[f][0] - ho[f][0]) / (1e-9 * abs(ho[f][0]) + 1e-12)
                    worst_first = max(worst_first, float(d))
            bg.mpc_advance()
            states.append(bg.get_x0())
        it_step, cv_step = bg.mpc_totals()
    finally:
        for p_ in (pg, po):
            p_.opts.shift, p_.opts.dual_reset = 2 ** 10, True
    assert worst_first <= 1.0, worst_first
    assert n_diff <= 0.002 * n_solves, (n_diff, n_solves)
    assert worst_short <= 1e-8 and worst_all <= 1e-4, (worst_short, worst_all)
    states = np.stack(states)
    assert np.abs(states[-1] - states[0]).max() > 0.5          # the vehicles really travel
    assert it_step.sum() > 64 * T                              # at least one Newton iteration per solve
    # the fused loop kernel against the step-wise launches above (same device arithmetic)
    pf = alg.scenarios.make_problem("C5", ids)
    it_f, cv_f, st_f = alg.mpc_solve(pf, T, record_states=True)
    assert st_f.shape == states.shape == (T + 1, 64, pg.model.n)
    same_f = it_f == it_step
    assert same_f.mean() >= 0.9, (np.nonzero(~same_f)[0], it_f[~same_f], it_step[~same_f])
    assert np.abs(st_f - states)[:, same_f].max() < 1e-6
    assert np.array_equal(cv_f[same_f], cv_step[same_f])
