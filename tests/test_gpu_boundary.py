"""The drop-in boundary as the Julia shim drives it (algames.jl_amd/julia/AlgamesHIP.jl cannot run here: no Julia toolchain).

Every ABI call the shim makes is issued here in the same order through the identical ctypes binding, and the write-back
arithmetic the shim performs on the Julia side is redone in NumPy and checked against the CPU oracle:
  * one handle kept alive across several solves (BatchedGameProblem): create once, newton_solve!, warm-started second solve
    with opts.dual_reset = false after pushing the multipliers the host holds (push_duals! -> alg_set_con_duals);
  * pull_results!: alg_get_traj -> pdtraj, alg_get_con_duals -> conval.λ / conval.μ through the row layout of
    include/algames_hip.h (abi_position), alg_get_history -> prob.stats (statistics.jl:30-57);
  * ibr_newton_solve!(bp; ibr_opts), ibr_newton_solve!(bp, i), mpc_solve!(bp, steps) on the same kind of handle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

UNI = 1


def _abi_positions(p, N, m, n, fin_ctl, nwall=0, ncirc=0, has_sb=False):
    """NumPy twin of AlgamesHIP.abi_position: 0-based positions of the rows of every reference conval in the ABI vector.
    Returns {("col", i, j): (N-1,), ("ctl",): (N-1, n_finite), ("wall", i): (N-1, nwall), ("circ", i): (N-1, ncirc)}."""
    K = N - 1
    out = {}
    col_len = p * (p - 1) * K
    for i in range(p):
        for j in range(p):
            if j != i:
                q = i * (p - 1) + (j if j < i else j - 1)
                out[("col", i, j)] = q * K + (np.arange(2, N + 1) - 2)
    ctl_len = 2 * m * K
    out[("ctl",)] = col_len + (np.arange(1, N)[:, None] - 1) * 2 * m + np.asarray(fin_ctl)[None, :]
    sb_len = p * 2 * n * K if has_sb else 0
    for i in range(p):
        base = col_len + ctl_len + sb_len
        out[("wall", i)] = base + ((i * K + (np.arange(2, N + 1) - 2))[:, None]) * nwall + np.arange(nwall)[None, :]
        out[("circ", i)] = base + p * nwall * K + ((i * K + (np.arange(2, N + 1) - 2))[:, None]) * ncirc + np.arange(ncirc)[None, :]
    return out


def test_shim_call_sequence_and_write_back(alg, orc):
    p, N, B = 3, 12, 4
    ids = np.arange(40, 40 + B)
    pg = alg.scenarios.make_problem("C5", ids, N=N)            # 3-player Unicycle, collision avoidance + control bounds +-1
    po = alg.scenarios.make_problem("C5", ids, N=N, backend=orc.lib())
    bg, bo = pg.batch, po.batch                                # setup!(bp): alg_create, set_options, set_x0, set_lqr, add_* (GameProblem.__init__)
    n, m = bg.n, bg.m
    assert bg.con_len == p * (p - 1) * (N - 1) + 2 * m * (N - 1)      # alg_get_con_len
    # ---- newton_solve!(bp): alg_newton_solve + pull_results!
    sg = bg.newton_solve(init=True, game_id0=int(ids[0]))
    so = bo.newton_solve(init=True, game_id0=int(ids[0]))
    for f in ("status", "outer_iters", "newton_iters", "records", "converged", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), f
    z = bg.get_traj(0)                                         # alg_get_traj: [x_1 | horizontal-order vector]
    X, U, L = bg.split_traj(z)
    assert np.array_equal(X[:, 0], pg.x0) and np.abs(z - bo.get_traj(0)).max() < 1e-8
    lam, mu = bg.get_con_duals()                               # alg_get_con_duals
    lo, mo = bo.get_con_duals()
    assert np.array_equal(mu, mo) and np.abs(lam - lo).max() <= 1e-6 * max(1.0, np.abs(lo).max())
    # conval layout: recompute every constraint value the way the reference's convals hold them and compare with evaluate!'s values
    pos = _abi_positions(p, N, m, n, fin_ctl=np.arange(2 * m))  # all +-1 bounds are finite
    vals = bg.dual_penalty_update()                            # evaluate! (+ dual / penalty update, checked below)
    R = 0.1                                                    # r_i + r_j = 0.05 + 0.05
    for i in range(p):
        for j in range(p):
            if j != i:
                d2 = (X[:, 1:, i] - X[:, 1:, j]) ** 2 + (X[:, 1:, p + i] - X[:, 1:, p + j]) ** 2
                assert np.abs(vals[:, pos[("col", i, j)]] - (R * R - d2)).max() < 1e-12
    cb = np.concatenate([U - 1.0, -1.0 - U], axis=2)           # [u - u_max; u_min - u] in joint control order
    assert np.abs(vals[:, pos[("ctl",)]] - cb).max() < 1e-12
    lam2, mu2 = bg.get_con_duals()
    assert np.array_equal(mu2, np.minimum(10.0 * mu, 1e7))     # penalty_update!: mu <- min(phi mu, mu_max)
    assert np.allclose(lam2, np.clip(lam + mu * vals, 0.0, 1e7), rtol=1e-13, atol=0)    # dual_update! with alpha = 1
    # the final record's per-knot profiles (read_back!: alg_get_violation_profile, before the histories): their maxima are that record's
    vg, vo = bg.violation_profile(), bo.violation_profile()
    for f in ("dyn", "con", "sta"):                              # (the HIP handle's multipliers moved in the dual update above: its optimality rows are no longer the record's)
        assert np.abs(vg[f] - vo[f]).max() <= 1e-9 * (1 + np.abs(vo[f]).max()), f
        assert np.allclose(vg[f].max(axis=1), sg["last"][f + "_vio"], rtol=1e-12, atol=0), f
    assert vg["opt"].shape == vo["opt"].shape == (B, bg.N)
    # prob.stats from alg_get_history (record! per stored record, statistics.jl:30-57)
    for g in range(B):
        hg, ho = bg.get_history(g, int(sg["records"][g])), bo.get_history(g, int(so["records"][g]))
        assert len(hg) == sg["records"][g] == len(ho)
        assert np.array_equal(hg["outer"], ho["outer"])
        for f in ("res", "delta", "dyn_vio", "con_vio", "sta_vio", "opt_vio"):
            assert np.allclose(hg[f], ho[f], rtol=1e-7, atol=1e-12), f
        assert hg["outer"][-1] == sg["outer_iters"][g] and hg["res"][-1] == sg["last"]["res"][g]
    # ---- second solve on the SAME handle, warm-started: push_duals! (alg_set_con_duals) + opts.dual_reset = false + explicit guess
    for b, (l_, m_) in ((bg, (lam, mu)), (bo, (lo, mo))):
        b.set_con_duals(l_, m_)
        b.set_options(dual_reset=0, outer_iter=3)
        b.set_traj(z, 0)                                       # both start from the HIP path's iterate
    s2g = bg.newton_solve(init=False, game_id0=int(ids[0])); s2o = bo.newton_solve(init=False, game_id0=int(ids[0]))
    for f in ("status", "outer_iters", "newton_iters", "records", "converged"):
        assert np.array_equal(s2g[f], s2o[f]), f
    assert np.all(s2g["newton_iters"] <= sg["newton_iters"])   # the warm start does not need more iterations
    assert np.abs(bg.get_traj(0) - bo.get_traj(0)).max() < 1e-8
    l3, m3 = bg.get_con_duals(); l3o, m3o = bo.get_con_duals()
    assert np.array_equal(m3, m3o) and np.abs(l3 - l3o).max() <= 1e-6 * max(1.0, np.abs(l3o).max())
    assert m3.min() >= mu.min()                                # penalties were kept, not reset to rho_0


def test_shim_ibr_and_mpc_on_a_live_handle(alg, orc):
    ids = np.arange(8, 12)
    pg = alg.scenarios.make_problem("C5", ids, N=10)
    po = alg.scenarios.make_problem("C5", ids, N=10, backend=orc.lib())
    # ibr_newton_solve!(bp; ibr_opts): alg_ibr_newton_solve with ordering .- 1
    io = alg.IBROptions(ibr_iter=2, ordering=[3, 1, 2])
    alg.ibr_newton_solve(pg, ibr_opts=io); alg.ibr_newton_solve(po, ibr_opts=io)
    sg, so = pg.stats.summary, po.stats.summary
    for f in ("status", "newton_iters", "records", "outer_iters", "ls_failures"):
        assert np.array_equal(sg[f], so[f]), f
    for g in range(len(ids)):                                  # every accumulated record is kept (history sized for ibr_iter * p solves)
        hg, ho = pg.batch.get_history(g, 4096), po.batch.get_history(g, 4096)
        assert len(hg) == sg["records"][g] == len(ho)
        assert np.allclose(hg["res"], ho["res"], rtol=1e-7, atol=1e-12)
    # ibr_newton_solve!(bp, i): alg_ibr_solve_player(i - 1) on the stored trajectories, statistics keep accumulating
    before = sg["records"].copy()
    alg.ibr_newton_solve(pg, i=2); alg.ibr_newton_solve(po, i=2)
    assert np.array_equal(pg.stats.summary["records"], po.stats.summary["records"]) and np.all(pg.stats.summary["records"] > before)
    hg = pg.batch.get_history(0, 4096)
    assert len(hg) == pg.stats.summary["records"][0]
    assert np.abs(pg.batch.get_traj() - po.batch.get_traj()).max() < 1e-7
    # mpc_solve!(bp, steps): alg_mpc_totals(reset) + alg_mpc_solve + alg_mpc_totals + alg_get_stats, same handle afterwards
    itg, cvg, stg = alg.mpc_solve(pg, 4, record_states=True)
    ito, cvo, sto = alg.mpc_solve(po, 4, record_states=True)
    assert np.array_equal(itg, ito) and np.array_equal(cvg, cvo) and np.abs(stg - sto).max() < 1e-7
    assert np.array_equal(pg.batch.get_x0(), stg[-1])          # the handle's x0 moved with the loop


@pytest.mark.gpu
def test_multi_device_solve_behind_the_boundary_two_handles_on_one_gpu(alg):
    """SURVEY.md 8(e) behind the boundary: `ShardedGameProblem` = contiguous shards, one handle (own stream) per device entry,
    alg_newton_solve_async on all of them, then alg_synchronize.  One GPU here, so both handles sit on device 0 and their
    launches run concurrently on two streams; the result must equal the single-handle batch bit for bit."""
    ids = np.arange(100, 100 + 512)
    one = alg.scenarios.make_problem("C2", ids); one.batch.set_waves_per_game(1)
    alg.newton_solve(one)
    two = alg.scenarios.make_problem("C2", ids, devices=[0, 0])
    for s in two.shards:
        s.batch.set_waves_per_game(1)
    alg.newton_solve(two)
    assert [s.B for s in two.shards] == [256, 256]
    assert np.array_equal(two.get_traj(), one.batch.get_traj())
    assert np.array_equal(two.stats.summary["newton_iters"], one.stats.summary["newton_iters"])
    assert alg.sharding.local_counters(two) == alg.sharding.local_counters(one)
    assert np.array_equal(two.stats.history(300)["res"], one.stats.history(300)["res"])


@pytest.mark.gpu
def test_statistics_t_elap_is_measured_on_the_device(alg):
    """Statistics.t_elap (statistics.jl:8,34; @elapsed at solver_methods.jl:40-42): record r carries the duration of the inner
    iteration that preceded it -- 0 for the first record of a solve, positive afterwards, and the sum over a game's records is
    bounded by the wall time of the launch that produced them."""
    import time
    prob = alg.scenarios.make_problem("C2", np.arange(64))
    alg.newton_solve(prob)                                     # warm-up (module load)
    prob = alg.scenarios.make_problem("C2", np.arange(64))
    t0 = time.perf_counter(); alg.newton_solve(prob); wall = time.perf_counter() - t0
    for game in (0, 17, 63):
        h = prob.stats.history(game)
        assert len(h) == prob.stats.summary["records"][game] >= 3
        assert h["t_elap"][0] == 0.0 and np.all(h["t_elap"][1:] > 0.0)
        assert 1e-6 < h["t_elap"][1:].max() < wall and h["t_elap"].sum() < wall
    assert np.array_equal(prob.stats.t_elap, prob.stats.history(0)["t_elap"])


@pytest.mark.gpu
def test_step_wise_history_grows_by_records(alg):
    """A host-driven loop of alg_newton_step adds one record per call: the device history grows geometrically by RECORDS (it used
    to be re-sized for a whole solve on every call) and keeps every record."""
    prob = alg.scenarios.make_problem("C2", np.arange(8), N=10)
    prob.batch.set_waves_per_game(1)
    prob.batch.init_traj(game_id0=0)
    prob.batch.rollout(0)
    n = 0
    for k in range(1, 4):
        for l in range(1, 60):
            info = prob.batch.newton_step(k, l)
            n += 1
    h = prob.batch.get_history(3)
    assert len(h) == n == 177 and np.all(h["res"] > 0)


@pytest.mark.gpu
def test_jacobian_of_a_game_range_and_scratch_release(alg):
    """alg_residual_jacobian_games builds the dense KKT Jacobians of a sub-range only (what the active-set inspection of ONE game of a
    large batch needs); alg_release_scratch frees the inspection buffer and the next call allocates it again."""
    prob = alg.scenarios.make_problem("C2", np.arange(12), N=8)
    prob.batch.init_traj(game_id0=0); prob.batch.rollout(0)
    full = alg.residual_jacobian(prob, 1e-4)
    part = alg.residual_jacobian(prob, 1e-4, games=(5, 3))
    assert part.shape == (3,) + full.shape[1:] and np.array_equal(part, full[5:8])
    prob.batch.release_scratch(); prob.batch.release_scratch()
    assert np.array_equal(alg.residual_jacobian(prob, 1e-4, games=(11, 1))[0], full[11])
    with pytest.raises(alg.AlgamesError):
        alg.residual_jacobian(prob, 1e-4, games=(11, 2))
    from algames_jl_amd import active_set as A
    core = A.ActiveSetCore(prob.probsize)
    A.residual_jacobian(core, prob, game=7)
    S = prob.probsize.S
    assert np.array_equal(core.jac[:S, :S], alg.residual_jacobian(prob, 0.0, games=(7, 1))[0])


@pytest.mark.gpu
def test_per_record_violation_profiles_through_the_step_wise_entry_points(alg):
    """The fused solver keeps the violation maxima per record; the .vio vectors of every record (violations.jl) are available to a
    caller that drives the inner iterations itself: the profile read after step t is the profile of record t + 1 (record! is made at
    the iterate the previous step left), so its maxima must be that record's *_vio."""
    prob = alg.scenarios.make_problem("C5", np.arange(6), N=12)
    b = prob.batch
    b.set_waves_per_game(1)
    b.init_traj(game_id0=0); b.rollout(0)
    profs = []
    for l in range(1, 6):
        b.newton_step(1, l)
        profs.append(b.violation_profile())
    for g in range(6):
        h = b.get_history(g)
        assert len(h) == 5
        for t in range(4):
            for f in ("dyn", "con", "sta", "opt"):
                # (the record comes from the fused trial pass, the profile from the flat row loops of alg_residual: the same row expressions with
                # their fmas written out wherever a product feeds a sum -- BT_vec, the pair terms -- so the two kernels agree to a few ulp of the
                # maximum, not to the rounding of a row's O(1) terms as in round 5)
                assert np.isclose(profs[t][f][g].max(), h[f + "_vio"][t + 1], rtol=4 * 2.3e-16, atol=1e-300), (g, t, f, profs[t][f][g].max(), h[f + "_vio"][t + 1])
