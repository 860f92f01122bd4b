"""The refinement gate of the Newton direction (alg_set_refinement / alg_get_refinement / alg_get_direction_gate,
alg_game_stats.refinements).  The reference's `lu(core.jac) \\ core.res` (solver_methods.jl:87) needs no such
thing; the structured elimination does where penalties sit at their ceiling.  What is pinned here:
  * the controls round-trip and reject nonsense;
  * a correction is harmless: with the tolerance at zero EVERY direction of every kernel family is corrected `max_steps` times and
    direction, line-search decisions, counts and solutions still agree with the oracle inside the usual tolerances -- and the
    statistics count exactly `max_steps` corrections per Newton direction;
  * a correction helps: at penalties of 1e7 the gate's row-wise backward error of the opt-u rows and the forward error against the
    long-double arbiter both drop when the gate is on;
  * switched off, the gate reports nothing and the solve is the unrefined elimination."""
import numpy as np
import pytest

from test_gpu_parity import DI, UNI, _pair

pytestmark = pytest.mark.gpu

FAMILIES = [(DI, 3, 2, 12), (DI, 2, 3, 8), (UNI, 3, 2, 10), (UNI, 4, 2, 8), (DI, 5, 2, 6), (DI, 3, 3, 6)]   # tile + dense directions


def test_refinement_controls_round_trip(alg, orc):
    g, _ = _pair(alg, orc, DI, 2, 2, 8, B=2)
    ms, tol, mu = g.get_refinement()
    assert (ms, tol, mu) == (2, 2.0 ** -34, 1.6e5)                     # the library's defaults
    gd, _ = _pair(alg, orc, DI, 3, 3, 6, B=2)                          # dense-direction configuration: up to eight corrections (round 6)
    assert gd.get_refinement() == (8, 2.0 ** -34, 1.6e5)
    g.set_refinement(1, 1e-9, 10.0); assert g.get_refinement() == (1, 1e-9, 10.0)
    g.set_refinement(tol=0.0); assert g.get_refinement() == (1, 0.0, 10.0)
    for bad in ((-1, 1e-9, 1.0), (9, 1e-9, 1.0), (1, -1.0, 1.0), (1, float("nan"), 1.0), (1, 1e-9, -2.0)):
        with pytest.raises(Exception):
            g.set_refinement(*bad)
    assert g.get_refinement() == (1, 0.0, 10.0)


@pytest.mark.parametrize("case", FAMILIES)
@pytest.mark.parametrize("waves", [1, 0])                              # one wavefront per game / the automatic team choice
def test_forced_corrections_keep_parity(alg, orc, case, waves):
    g, o = _pair(alg, orc, *case, B=5)
    g.set_waves_per_game(waves)
    z0, con0 = g.get_traj(0), g.get_con_duals()
    g.set_refinement(2, 0.0, 0.0)                                      # tolerance zero: two corrections on every direction
    for reg in (1e-3, 1e-7 * 2 ** 4):
        dg, sg = g.newton_direction(reg); do, so = o.newton_direction(reg)
        assert np.all(sg == 0) and np.all(so == 0)
        assert (np.abs(dg - do) / np.abs(do).max(axis=1, keepdims=True)).max() < 1e-9
        J = o.residual_jacobian(reg); res = o.residual()[0]
        assert np.abs(np.einsum("brc,bc->br", J, dg) + res).max() <= 1e-8 * max(1.0, np.abs(res).max())
        gate = g.get_direction_gate()                                  # [max |rho|, omega, row scale] of the FIRST solve's opt-u rows
        assert np.all(np.isfinite(gate)) and np.all(gate[:, 1] <= 1e-9) and np.all(gate[:, 2] > 0)
        assert np.all(gate[:, 0] <= gate[:, 1] * gate[:, 2] * (1 + 1e-12))
    # full solves: these random problems do not converge inside the iteration limits and amplify rounding differences (1e-7 between
    # the plain elimination and the oracle on some) -- what a correction must not do is add to that distance
    sg, so = g.newton_solve(init=True, game_id0=3), o.newton_solve(init=True, game_id0=3)
    for f in ("status", "outer_iters", "newton_iters", "ls_failures", "converged"):
        assert np.array_equal(sg[f], so[f]), f
    forced = np.abs(g.get_traj(0) - o.get_traj(0)).max()
    assert np.array_equal(sg["refinements"], 2 * sg["newton_iters"]) and sg["newton_iters"].min() > 0
    g.set_traj(z0); g.set_con_duals(*con0); g.set_refinement(0)
    s0 = g.newton_solve(init=True, game_id0=3)
    plain = np.abs(g.get_traj(0) - o.get_traj(0)).max()
    assert np.all(s0["refinements"] == 0) and np.array_equal(s0["newton_iters"], so["newton_iters"])
    assert forced <= 1e-8 + 2.0 * plain, (forced, plain)


def test_forced_corrections_keep_ibr_parity(alg, orc):
    dist = {}
    for name, setting in (("plain", (0, 0.0, 0.0)), ("forced", (2, 0.0, 0.0))):
        g, o = _pair(alg, orc, UNI, 3, 2, 10, B=5)
        g.set_refinement(*setting)
        sg, so = g.ibr_newton_solve(6, init=True, game_id0=1), o.ibr_newton_solve(6, init=True, game_id0=1)
        for f in ("status", "outer_iters", "newton_iters", "ls_failures"):
            assert np.array_equal(sg[f], so[f]), (name, f)
        dist[name] = np.abs(g.get_traj(0) - o.get_traj(0)).max()
        assert np.array_equal(sg["refinements"], (2 if name == "forced" else 0) * sg["newton_iters"]) and sg["newton_iters"].min() > 0
    assert dist["forced"] <= 1e-8 + 2.0 * dist["plain"], dist


def test_gate_lowers_backward_and_forward_error_at_penalty_ceiling(alg, orc):
    """A warm-start-like state of the C5 problem with every penalty at 1e7 (where the lock-step run needed the gate)."""
    import algames_jl_amd as A
    ids = np.arange(128, 136)
    pg, px = A.scenarios.make_problem("C5", ids), A.scenarios.make_problem("C5", ids, backend=orc.lib("x"))
    g, x = pg.batch, px.batch
    g.set_waves_per_game(1)
    A.newton_solve(pg)                                                   # a converged iterate, then the penalties of a late MPC step
    z = g.get_traj(0); lam, mu = g.get_con_duals()
    rng = np.random.default_rng(0)
    z = z + 1e-3 * rng.standard_normal(z.shape); z[:, :g.n] = g.get_x0()
    mu[:] = 1e7; lam = lam + (rng.random(lam.shape) < 0.2) * rng.random(lam.shape)
    for b in (g, x):
        b.set_x0(g.get_x0()); b.set_traj(z); b.set_con_duals(lam, mu)
    reg = 1e-7
    dx, sx = x.newton_direction(reg); assert np.all(sx == 0)
    J, res = x.residual_jacobian(reg), x.residual()[0]
    n, mi, p, N = g.n, g.mi, g.p, g.N
    ur = np.array([i * (N - 1) * (n + mi) + k * (n + mi) + n + j for i in range(p) for k in range(N - 1) for j in range(mi)])
    def omega(d):                                                        # row-wise backward error of the opt-u rows (Oettli-Prager)
        rho = np.abs(np.einsum("brc,bc->br", J[:, ur], d) + res[:, ur])
        return (rho / (np.einsum("brc,bc->br", np.abs(J[:, ur]), np.abs(d)) + np.abs(res[:, ur]) + 1e-300)).max(axis=1)
    scale = np.abs(dx).max(axis=1)
    g.set_refinement(0); d0, s0 = g.newton_direction(reg); gate0 = g.get_direction_gate()
    g.set_refinement(2, 2.0 ** -34, 1.6e5); d1, s1 = g.newton_direction(reg); gate1 = g.get_direction_gate()
    assert np.all(s0 == 0) and np.all(s1 == 0)
    e0, e1 = np.abs(d0 - dx).max(axis=1) / scale, np.abs(d1 - dx).max(axis=1) / scale
    w0, w1 = omega(d0), omega(d1)
    print("forward error vs arbiter: gate off max %.2e median %.2e, gate on max %.2e median %.2e; omega off max %.2e on max %.2e; device omega %.2e"
          % (e0.max(), np.median(e0), e1.max(), np.median(e1), w0.max(), w1.max(), gate1[:, 1].max()))
    assert np.all(gate0 == 0.0)                                          # gate off: nothing evaluated, nothing reported
    # the device's figure (evaluated on the first solve) is the host's figure of the unrefined direction, from ABOVE -- the side that matters
    # for a gate: its row scale |B[:,c]|' |dlambda| keeps the signs of the RK2 coefficients, a lower estimate of the scale (measured on this
    # state: up to 9 x with the round-4 direction, up to 92 x with the split recursion's of round 5, whose largest opt-u residual sits in a row
    # with cancelling signs; 259 x once B' lambda states its fmas, BT_vec).  The upper factor only documents how conservative the estimate can get.
    # Round 6: the device's row scale is the true |J_c| |d| + |r_c| (|B[:,c]|' |dlambda| with the coefficients' magnitudes, BT_vec_abs), so its
    # figure is the host's up to rounding and to the pair it tracks by cross-multiplication: the factor below was 1024 in round 5.
    print("device omega / host omega: max %.3f min %.3f" % ((gate1[:, 1] / w0).max(), (gate1[:, 1] / w0).min()))
    assert np.all(w0 <= gate1[:, 1] * (1 + 1e-6) + 1e-15) and np.all(gate1[:, 1] <= 2 * w0 + 1e-15)
    assert w0.max() > 2.0 ** -34                                         # the state does need the gate ...
    assert w1.max() <= 4 * 2.0 ** -34 and np.all(w1 <= w0 * 1.01 + 1e-16)         # ... and the gate delivers its tolerance
    assert np.all(e1 <= e0 * 1.01 + 1e-13) and np.median(e1) <= np.median(e0) / 50 and e1.max() <= min(1e-8, e0.max() / 50)   # measured: 7.7e-7 -> 1.5e-9


def test_switched_off_is_the_plain_elimination(alg, orc):
    import algames_jl_amd as A
    ids = np.arange(300, 308)
    pg, po = A.scenarios.make_problem("C5", ids), A.scenarios.make_problem("C5", ids, backend=orc.lib())
    pg.batch.set_refinement(0)
    A.newton_solve(pg); A.newton_solve(po)
    sg, so = pg.stats.summary, po.stats.summary
    assert np.all(sg["refinements"] == 0) and np.all(pg.batch.get_direction_gate() == 0.0)
    for f in ("status", "outer_iters", "newton_iters", "ls_failures", "converged"):
        assert np.array_equal(sg[f], so[f]), f
    assert sg["converged"].all() and np.abs(pg.batch.get_traj(0) - po.batch.get_traj(0)).max() <= 1e-8


def test_a_failed_correction_solve_leaves_the_trial_buffer_s_x1_intact(alg):
    """a correction solve that fails on pass > 0 is dropped and the line search runs on the direction of the passes
    before -- but the correction's forward sweep has zeroed x_1 of its output buffer, the TRIAL buffer, and an accepted trial becomes
    pdtraj.  A test build of the C5 kernels (-DALG_TEST_FAIL_CORRECTION: every correction "fails" after its sweeps ran) must still keep
    x_1 = x0 in both buffers, solve like the library with the refinement switched off, and report status OK."""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    var = os.path.join(root, "algames.jl_amd", "lib", "variants", "failcorr.so")
    src = [os.path.join(root, "algames.jl_amd", "csrc", f) for f in ("algames_direction.hpp", "algames_solver.hpp", "algames_assemble.hpp", "algames_device.hpp")]
    if not os.path.exists(var) or any(os.path.getmtime(f) > os.path.getmtime(var) for f in src):
        r = subprocess.run(["bash", os.path.join(root, "tests", "probes", "build_variant.sh"), "failcorr", "-DALG_TEST_FAIL_CORRECTION"],
                           env=dict(os.environ, UNITS="base_7"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    code = r'''
import sys, json; sys.path.insert(0, %r)
import numpy as np, algames_jl_amd as alg
out = {}
for name, rs in (("off", (0, 0.0)), ("forced", (2, 0.0))):
    prob = alg.scenarios.make_problem("C5", np.arange(8)); b = prob.batch
    b.set_waves_per_game(1); b.set_refinement(rs[0], rs[1], 1.6e5)
    b.init_traj(game_id0=0); b.rollout(0)
    d, st = b.newton_direction(1e-7)
    x1_trial = b.get_traj(1)[:, :b.n]
    alg.newton_solve(prob)
    s = prob.stats.summary
    out[name] = dict(status=st.tolist(), x1_trial_err=float(np.abs(x1_trial - b.get_x0()).max()), x1_err=float(np.abs(b.get_traj(0)[:, :b.n] - b.get_x0()).max()),
                     solve_status=s["status"].tolist(), iters=s["newton_iters"].tolist(), refinements=s["refinements"].tolist(), z=b.get_traj(0).tobytes().hex())
print("RESULT " + json.dumps(out))
''' % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ALGAMES_HIP_LIB=var), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    for name in ("off", "forced"):
        assert out[name]["x1_trial_err"] == 0.0 and out[name]["x1_err"] == 0.0, (name, out[name]["x1_trial_err"], out[name]["x1_err"])
        assert all(v == 0 for v in out[name]["status"]) and all(v == 0 for v in out[name]["solve_status"])
    # every "failed" correction was dropped: the solve is the plain elimination's, bit for bit; the attempts are counted
    assert out["forced"]["z"] == out["off"]["z"] and out["forced"]["iters"] == out["off"]["iters"]
    assert all(r_ >= i_ for r_, i_ in zip(out["forced"]["refinements"], out["forced"]["iters"])) and all(v == 0 for v in out["off"]["refinements"])
