// algames_solver.hpp -- solver control flow: line_search, inner_iteration, newton_solve!, dual / penalty update, initialisation, iterated best response
// (part of the device code of libalgames_hip.so; included by algames_device.hpp, which holds the shared declarations and the file-level
// description of the execution model)
#pragma once
#include "algames_device.hpp"

namespace alg {

// ================================================================================================
// Solver control flow (solver_methods.jl:5-125), per game
// ================================================================================================

// record! (statistics.jl:44-57): unregularised residual at pdtraj; also leaves the step records (with the Jacobian
// regularisation jreg folded into R^) for the Newton direction and refreshes G.vals(pr).  The record is pushed to the
// game's Statistics history (lane 0); the two scalars the control flow needs are returned.
struct RecScalars { double res, opt; int nonfinite; };
// Statistics of an accepted line-search trial = what the next record! would recompute (same point, same arithmetic)
// (kept in HBM, G.tc(pr), so that it costs no registers across the Newton direction)
// t_elap of the reference's Statistics (statistics.jl:8,34; @elapsed around inner_iteration, solver_methods.jl:40-42): lane 0 stamps
// the 100 MHz real-time counter into the game's scratch block at the top of an inner iteration and turns it into seconds at its
// end -- through HBM, so that no register is live across the phases of the iteration; the next record! picks it up.
constexpr int TC_TELAP = 8, TC_TSTART = 9;
static_assert(TC_TSTART < TC_LEN, "per-game control slots");
__device__ __forceinline__ void iter_clock_start(CPR pr0, const Game& G0) {
    CPR pr = phase_params(pr0); const Game G = G0.fresh();
    if (phase_lane() == 0) G.tc(pr)[TC_TSTART] = (double)__builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void iter_clock_stop(CPR pr0, const Game& G0) {
    CPR pr = phase_params(pr0); const Game G = G0.fresh();
    if (phase_lane() == 0) G.tc(pr)[TC_TELAP] = ((double)__builtin_amdgcn_s_memrealtime() - G.tc(pr)[TC_TSTART]) * 1e-8;
}
__device__ __forceinline__ void tcache_store(CPR pr0, const Game& G0, const ResOut& ro) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    if (phase_lane() == 0) { G.tc(pr)[0] = ro.l1; G.tc(pr)[1] = ro.opt; G.tc(pr)[2] = ro.dyn; G.tc(pr)[3] = ro.con; G.tc(pr)[4] = ro.sta; G.tc(pr)[5] = (double)ro.nonfinite; }
}
__device__ __forceinline__ void tcache_load(CPR pr0, const Game& G0, ResOut& ro) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    ro.l1 = G.tc(pr)[0]; ro.opt = G.tc(pr)[1]; ro.dyn = G.tc(pr)[2]; ro.con = G.tc(pr)[3]; ro.sta = G.tc(pr)[4]; ro.nonfinite = (int)G.tc(pr)[5]; ro.l1reg = ro.l1;
}

__device__ __forceinline__ RecScalars push_stats(CPR pr0, const Game& G0, const ResOut& ro, double delta, int outer, alg_record* out) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    if (phase_lane() == 0) {
        alg_record rc;
        // (both loads of this block first: one round trip instead of two)
        const double telap = G.tc(pr)[TC_TELAP];      // the previous inner iteration's duration (iter_clock_stop)
        const int idx = G.st(pr)->records;
        rc.outer = outer; rc.ls_j = 0; rc.alpha = 0.0; rc.res = ro.l1 / (double)pr.S; rc.delta = delta;
        rc.dyn_vio = ro.dyn; rc.con_vio = ro.con; rc.sta_vio = ro.sta; rc.opt_vio = ro.opt;
        rc.t_elap = telap;
        if (idx < pr.hist_max) G.hist(pr)[idx] = rc;
        G.st(pr)->records = idx + 1;
        G.st(pr)->last = rc;
        if (out) *out = rc;
    }
    RecScalars r; r.res = uni(ro.l1 / (double)phase_int(pr.S)); r.opt = uni(ro.opt); r.nonfinite = __builtin_amdgcn_readfirstlane(ro.nonfinite);
    return r;
}
// dual = true (fused kernels only, round 6): this record! is the one that follows dual_update! + penalty_update! in newton_solve!
// (solver_methods.jl:57-61, then :73 of the next outer iteration) and performs them on the way (assemble_phase_a, DUAL)
#ifndef ALG_R6_DUALREC
#define ALG_R6_DUALREC 1        // A/B switch (tests/probes/build_variant.sh): 0 = dual_penalty_update as a pass of its own, as until round 5
#endif
template <class C> inline constexpr bool dual_in_record_v = AsmLds<C>::FUSED && ALG_R6_DUALREC != 0;
template <class C>
__device__ __forceinline__ RecScalars make_record(CPR pr, const Game& G, Lds<C>& L, double delta, int outer, double jreg, alg_record* out, bool dual = false) {
    ResOut ro;
    LSP_T0 LSP_COUNT(29)
    if constexpr (AsmLds<C>::FUSED) assemble_fused<C, 1, false, true>(pr, G, L.a, 0.0, false, 0.0, jreg, ro, dual);
    else assemble_pass<C, 1>(pr, G, L.a, 0, -1, 0.0, jreg, ro);
    game_sync();
    LSP(27)
    return push_stats(pr, G, ro, delta, outer, out);
}

// line_search (solver_methods.jl:105-125).  jreg_next >= 0: every trial also leaves the unregularised statistics and
// step records (R^ with jreg_next) so that an accepted trial can serve as the next iteration's record!.
template <class C>
__device__ void line_search(CPR pr, const Game& G, Lds<C>& L, double reg, double res_norm0, double jreg_next,
                            double* alpha_out, int* j_out) {
    int j = 1; double alpha = 1.0;
    bool staged = false;                                  // (LsMulti: [z | dz] of this search are in LDS)
    bool expect_on = false; double expect = 0.0;          // (LsMulti: norm the group pass computed for the step size the next trial evaluates)
    while (j < pr.opt.ls_iter) {
        const auto& o = phase_params(pr).opt;
        LSP_T0 LSP_COUNT(18)
        ResOut ro;
        if constexpr (AsmLds<C>::FUSED) {
            // update_traj! and the residual of the trial in one pass over the trajectory (assemble_fused)
            if (C::TRIAL_REUSE && jreg_next >= 0.0 && o.regularize) assemble_fused<C, 3, true>(pr, G, L.a, alpha, true, reg, jreg_next, ro);
            else assemble_fused<C, 0, true>(pr, G, L.a, alpha, o.regularize != 0, reg, 0.0, ro);
        } else {
        update_traj<C>(pr, G, 1, 0, alpha);
        game_sync();
        LSP(16)
        bool done = false;
        if constexpr (C::TRIAL_REUSE) {
            if (jreg_next >= 0.0 && o.regularize) { assemble_pass<C, 3>(pr, G, L.a, 1, 0, reg, jreg_next, ro); done = true; }
        }
        if (!done) assemble_pass<C, 0>(pr, G, L.a, 1, o.regularize ? 0 : -1, reg, 0.0, ro);
        }
        LSP(17)
        if (jreg_next >= 0.0) tcache_store(pr, G, ro);
        const double rt = uni(ro.l1reg / (double)phase_int(phase_params(pr).S));
        if constexpr (LsMulti<C>::ON) {
            // the group pass below said this step size passes the test: the two passes must have produced the same norm, bit for bit (the step is
            // taken on THIS pass's norm either way; a disagreement would just let the search go on)
            if (expect_on) { expect_on = false; if (ro.l1reg != expect && phase_lane() == 0) G.fresh().st(phase_params(pr))->reserved += 1; }
        }
        if (rt <= (1.0 - alpha * o.beta) * res_norm0) break;
        alpha *= o.alpha_decrease; j += 1;
        if constexpr (LsMulti<C>::ON) {
            // The first step size was rejected: the following ones are tried LsMulti::NA at a time by a pass that leaves only their norms
            // (trial_norms_multi) until one passes the test -- the loop's next trial is then that one, with its outputs -- or none is left.
            // (j, alpha) move exactly as the one-by-one search moves them.
            if (phase_int(phase_params(pr).ls_multi)) {
                constexpr int NA = LsMulti<C>::NA;
                const bool lds_fit = 2 * phase_int(phase_params(pr).traj_len) <= LsMulti<C>::CAP &&
                                     (!LsLds<C>::SC_ON || NA * (phase_int(phase_params(pr).N) - 1) * LsMulti<C>::SW <= LsLds<C>::SCAP);
                while (j < phase_int(phase_params(pr).opt.ls_iter)) {
                    const auto& om = phase_params(pr).opt;
                    double nr[NA];
                    const int left = om.ls_iter - j, nc = left < NA ? left : NA;
                    LSP_T0 LSP_COUNT(18)
                    bool in_lds = false;
                    if constexpr (LsMulti<C>::LDSZ) {
                        if (lds_fit) {
                            if (!staged) { ls_stage_traj<C>(pr, G, L.ls.z); staged = true; }
                            trial_norms_multi<C, NA, true>(pr, G, L.ls.z, L.ls.sc, alpha, om.regularize != 0, reg, nr);
                            in_lds = true;
                        }
                    }
                    if (!in_lds) trial_norms_multi<C, NA, false>(pr, G, nullptr, nullptr, alpha, om.regularize != 0, reg, nr);
                    LSP(30)
                    int hit = -1; double a = alpha, ahit = alpha, last = alpha;
#pragma unroll
                    for (int q = 0; q < NA; q++) {
                        const double rq = uni(nr[q] / (double)phase_int(phase_params(pr).S));
                        if (hit < 0 && q < nc && rq <= (1.0 - a * om.beta) * res_norm0) { hit = q; ahit = a; expect = nr[q]; }
                        if (q < nc) last = a;
                        a *= om.alpha_decrease;
                    }
                    if (hit >= 0) { alpha = ahit; j += hit; expect_on = true; break; }
                    alpha = last * om.alpha_decrease; j += nc;
                }
            }
        }
    }
    *alpha_out = alpha; *j_out = j;
}

// Issue priority of a one-wavefront game.  The SIMD's arbiter serves its wavefronts oldest first: of the four games that share a SIMD at
// the BASELINE batch the first-dispatched one finishes after 3.7 ms and the last after 4.7 ms (tests/probes/finish_times.py), and the
// SIMD runs its last millisecond with three, two, one wavefront.  s_setprio overrides the age order completely (a static priority
// by dispatch round reverses the finishing order exactly), so every inner iteration rotates the priority by one: each game spends
// a quarter of its iterations at each level and the four finish together (mean / max of the per-game durations 0.87 -> 0.96; C2
// 10.2 -> 10.7 M/s, C4 10.6 -> 11.1 M/s in A/B runs; rotating every second iteration, every time step of the backward sweep, or twice per
// iteration all measured worse than once per inner iteration).  Dispatch round = blockIdx / (number of SIMDs: 256 CUs x 4).
template <class C> __device__ __forceinline__ void rotate_priority(int it) {
    if constexpr (C::NW == 1) {
        const int q = ((int)(blockIdx.x >> 10) + it) & 3;
        if (q == 0) __builtin_amdgcn_s_setprio(0); else if (q == 1) __builtin_amdgcn_s_setprio(1); else if (q == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
}
// inner_iteration (solver_methods.jl:67-103).  Returns status (bits 0-7) | control_flow << 8; step details go to
// the history record / *info (lane 0).  `cache` (optional) carries an accepted trial's statistics to the next call.
template <class C>
__device__ __forceinline__ int inner_iteration(CPR pr0, Game& G_, Lds<C>& L, int& LS_count, double& Delta, int k, int l,
                               alg_step_info* info, int* cache_valid) {
    Game& G = G_;
    CPR pr = phase_params(pr0);
    const bool lane0 = phase_lane() == 0;
    const auto& o = pr.opt;
    const double lf = (double)l;
    const double reg = o.reg_0 * (lf * lf * lf * lf);                      // :39  reg_0 * l^4
    if (info && lane0) { alg_step_info z{}; *info = z; }
    rotate_priority<C>(k + l);
    iter_clock_start(pr, G_);                    // @elapsed begins (solver_methods.jl:40); record! below still reads the previous t_elap
    RecScalars rs;                                                         // :73-76 (regularisation term is zero at pdtraj)
    // *cache_valid: 1 = the accepted trial of the previous iteration is this record!; 2 = the dual / penalty update of the previous outer
    // iteration is still due and rides on this record! pass (dual_in_record_v)
    if (cache_valid && *cache_valid == 1) { ResOut cro; tcache_load(pr, G, cro); rs = push_stats(pr, G, cro, Delta, k, info ? &info->rec : nullptr); }
    else rs = make_record<C>(pr, G, L, Delta, k, reg, info ? &info->rec : nullptr, dual_in_record_v<C> && cache_valid && *cache_valid == 2);
    if (cache_valid) *cache_valid = 0;
    Delta = 0.0;                                                           // :79
    auto finish = [&](int status, int flow) { if (info && lane0) { info->status = status; info->control_flow = flow; } iter_clock_stop(pr, G_); return status | (flow << 8); };
    if (rs.nonfinite) return finish(ALG_STATUS_NAN, 1);
    if (rs.opt < o.eps_opt) return finish(ALG_STATUS_OK, 1) | (1 << 16);  // :80-82 (bit 16: pdtraj untouched since this record!)
    double pl1; int st;
    LSP_T0 LSP_COUNT(28)
    st = refined_direction<C>(pr, G, L, reg, -1, &pl1);                                    // :84-88
    LSP(26)
    if (st != ALG_STATUS_OK) return finish(st, 1);
    game_sync();
    double alpha; int j;
    const double lf1 = (double)(l + 1);
    const bool reuse = C::TRIAL_REUSE && cache_valid && l < o.inner_iter && o.regularize;    // the next inner iteration may reuse the trial
    line_search<C>(pr, G, L, reg, rs.res, reuse ? o.reg_0 * (lf1 * lf1 * lf1 * lf1) : -1.0, &alpha, &j);   // :91
    const int failed = (j == o.ls_iter);                                   // :92
    if (failed) LS_count += 1; else LS_count = 0;                          // :93
    game_sync();
    // :94 update_traj!(pdtraj, pdtraj, alpha, delta): the last trial already holds exactly these values unless the search ran
    // out of trials (alpha was halved once more after the last trial) -> exchange the roles of the two buffers
    if (!failed) { const int t = G.zo[0]; G.zo[0] = G.zo[1]; G.zo[1] = t; }
    else update_traj<C>(pr, G, 0, 0, alpha);
    { double sd = pl1; sd *= alpha; sd /= (double)((phase_int(pr.N) - 1) * (C::n + C::m)); Delta = uni(sd); }     // :95 Delta_step
    game_sync();
    if (reuse && !failed) *cache_valid = 1;
    if (lane0) {
        // (measured in round 6: the loads of this block -- the counters, the record index, the clock stamp -- issued together, one round trip instead
        // of four: neutral at C2, profiles/r06_ab_micro_c2.txt; not kept)
        const Game G = G_.fresh();
        G.st(pr)->newton_iters += 1; if (failed) G.st(pr)->ls_failures += 1;
        const int idx = G.st(pr)->records - 1;
        if (idx < pr.hist_max) { G.hist(pr)[idx].alpha = alpha; G.hist(pr)[idx].ls_j = j; }
        G.st(pr)->last.alpha = alpha; G.st(pr)->last.ls_j = j;
        if (info) { info->alpha = alpha; info->ls_j = j; info->ls_failed = failed; info->delta = Delta; info->rec.alpha = alpha; info->rec.ls_j = j; }
    }
    return finish(ALG_STATUS_OK, Delta < o.delta_min ? 1 : 0);             // :96-98
}

// reset!(game_con) (constraints_methods.jl:295-327)
template <int NT = WAVE>
__device__ __forceinline__ void reset_con(CPR pr0, const Game& G0) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    for (int e = phase_lane(); e < pr.con_len; e += NT) { G.lam(pr)[e] = 0.0; G.mu(pr)[e] = pr.opt.rho_0; }
}
// evaluate! + dual_update! + penalty_update! (solver_methods.jl:57-61; constraints_methods.jl:329-379,421-440)
template <class C>
__device__ void dual_penalty_update(CPR pr0, const Game& G0) {
    CPR pr = phase_params(pr0);
    constexpr int n = C::n, m = C::m, P = C::P;
    const Game G = G0.fresh();
    const int N = phase_int(pr.N), tid = phase_lane(); const auto& o = pr.opt; const double* z = G.z(0);
    if (pr.has_colavoid) {
        for (int e = tid; e < pr.col_len; e += C::NT) {
            constexpr int PM1 = P > 1 ? P - 1 : 1;
            const int q = e / (N - 1), k = e % (N - 1) + 1, i = q / PM1, jj = q % PM1, j = jj < i ? jj : jj + 1;
            const double* x = zstate<C>(z, k);
            const double d0 = x[i] - x[j], d1 = x[P + i] - x[P + j], R = pr.ca_pair_r[i * MAXP + j];
            double s2 = pair_dist2(d0, d1);
            if constexpr (C::PD == 3) { const double d2 = pr.ca_dim == 3 ? x[2 * P + i] - x[2 * P + j] : 0.0; s2 = __builtin_fma(d2, d2, s2); }
            const double c = ca_value((double)((pr.ca_mask[i] >> j) & 1u), R, s2);      // (the roundings of assemble_phase_a: its DUAL form performs this very update)
            G.vals(pr)[e] = c;
            G.lam(pr)[e] = dual_ascent(G.lam(pr)[e], o.alphax_dual[i], G.mu(pr)[e], c, o.lambda_max);
        }
    }
    if (pr.has_ctl) {
        for (int e = tid; e < pr.ctl_len; e += C::NT) {
            const int k = e / (2 * m), row = e % (2 * m), c = row % m;
            const double u = z[n + hu<C>(k, 0) + uoff<C>(c)];
            const double cv = row < m ? u - pr.umax[c] : pr.umin[c] - u;
            const int ci = pr.col_len + e;
            G.vals(pr)[ci] = cv;
            if (isfinite(cv)) G.lam(pr)[ci] = dual_ascent(G.lam(pr)[ci], o.alpha_dual, G.mu(pr)[ci], cv, o.lambda_max);
        }
    }
    if constexpr (C::EXT) {
        // state constraints of player i: dual_update! with alphax_dual[i] (constraints_methods.jl:421-440)
        const int e0 = pr.col_len + pr.ctl_len, K = N - 1;
        for (int e = tid; e < pr.sb_len + pr.wall_len + pr.circ_len + pr.wall3_len + pr.cyl_len; e += C::NT) {
            int i, k; double c;
            if (e < pr.sb_len) {
                const int row = e % (2 * n); k = (e / (2 * n)) % K; i = e / (2 * n * K);
                const double* x = zstate<C>(z, k + 1);
                c = row < n ? x[row] - ext_sbmax(pr, pr.extc)[i * n + row] : ext_sbmin(pr, pr.extc)[i * n + row - n] - x[row - n];
            } else if (e < pr.sb_len + pr.wall_len) {
                const int e2 = e - pr.sb_len, w = e2 % pr.nwall; k = (e2 / pr.nwall) % K; i = e2 / (pr.nwall * K);
                const double* x = zstate<C>(z, k + 1); double gx, gy;
                c = (double)((pr.wall_mask[i] >> w) & 1u) * wall_val(ext_walls(pr, pr.extc), w, x[i], x[P + i], &gx, &gy);
            } else if (e < pr.sb_len + pr.wall_len + pr.circ_len) {
                const int e2 = e - pr.sb_len - pr.wall_len, cq = e2 % pr.ncirc; k = (e2 / pr.ncirc) % K; i = e2 / (pr.ncirc * K);
                const double* x = zstate<C>(z, k + 1); double gx, gy;
                c = (double)((pr.circ_mask[i] >> cq) & 1u) * circ_val(ext_circs(pr, pr.extc), cq, x[i], x[P + i], &gx, &gy);
            } else {
                i = 0; k = 0; c = 0.0;
                if constexpr (C::PD == 3) {
                    int e2 = e - pr.sb_len - pr.wall_len - pr.circ_len;
                    const bool w3 = e2 < pr.wall3_len;
                    if (!w3) e2 -= pr.wall3_len;
                    const int cnt = w3 ? pr.nwall3 : pr.ncyl, q = e2 % cnt; k = (e2 / cnt) % K; i = e2 / (cnt * K);
                    const double* x = zstate<C>(z, k + 1);
                    const double pos[3] = {x[i], x[P + i], x[2 * P + i]}; double g[3];
                    const double on = (double)(((w3 ? pr.wall3_mask[i] : pr.cyl_mask[i]) >> q) & 1u);
                    c = on * (w3 ? wall3_val(ext_walls3(pr, pr.extc), q, pos, g) : cyl_val(ext_cyls(pr, pr.extc), q, pos, g));
                }
            }
            const int ci = e0 + e;
            G.vals(pr)[ci] = c;
            if (isfinite(c)) { const double lb = G.lam(pr)[ci] + o.alphax_dual[i] * G.mu(pr)[ci] * c; G.lam(pr)[ci] = fmin(fmax(lb, 0.0), o.lambda_max); }
        }
    }
    // penalty_update! rewrites mu of EVERY row: in a team, another wavefront may still be in the dual-update loops above, which
    // read mu of rows this thread is about to scale (one wavefront alone runs the loops in program order)
    if constexpr (C::NW > 1) game_sync();
    for (int e = tid; e < pr.con_len; e += C::NT) G.mu(pr)[e] = fmin(fmax(G.mu(pr)[e] * o.rho_increase, 0.0), o.rho_max);
}

// rollout!(RK3, model, traj) (solver_methods.jl:17): lanes < P integrate their own player (players are decoupled)
// lds / cap: optional LDS scratch of `cap` doubles (the solver kernels pass their idle Lds union).  The states are written into the very array
// the controls are read from, so the compiler keeps every step's control loads behind the previous step's state stores: the rollout
// of a C2 game waited for 39 global round trips in a row -- 190 K cycles per solve, 3 % of it (profiles/r06_phase_cycles_c2_4096_*.txt).
// With the scratch the controls of all steps are fetched in one go (every thread of the game, all loads in flight) and the serial loop
// reads them at LDS latency.  Same arithmetic on the same numbers.
// (OL: the lane index through an opaque copy -- inside k_mpc_loop the lane predicates of this function were invariants of the receding-horizon
// loop, live across every solve)
// (measured in round 6 and not taken: the roll-out on the player's own compact arrays instead of joint arrays indexed by the lane's player number --
// fewer compare masks in the prologue, but the C3 solve kernel's SGPR spills went 8 -> 13 and the C5 loop kernel's 13 -> 17)
#ifndef ALG_ROLLOUT_OWN
#define ALG_ROLLOUT_OWN 0
#endif
template <class C, bool OL = false>
__device__ __forceinline__ void rollout(CPR pr, double* z, double* lds = nullptr, int cap = 0) {
    constexpr int n = C::n, m = C::m, P = C::P;
    const int lane = OL ? phase_lane() : game_tid();
    const int NU = (pr.N - 1) * m;
    const bool staged = lds != nullptr && NU <= cap;                        // wave-uniform
    if (staged) {
        for (int e = lane; e < NU; e += C::NT) lds[e] = gld(z, n + (e / m) * C::b + n + e % m);      // u_k, player-grouped like the trajectory
        game_sync();
    }
    auto uget = [&](int k, int i, int j) { return staged ? lds[k * m + i * C::mi + j] : z[n + hu<C>(k, i) + j]; };
    if constexpr (C::QUAD) {
        if (lane < P) {
            double xi[12], ui[4], xo[12];
#pragma unroll
            for (int j = 0; j < 12; j++) xi[j] = z[lane + j * P];
            for (int k = 0; k < pr.N - 1; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) ui[j] = uget(k, lane, j);
                quad_rk3(xi, ui, pr.qmass, pr.dt, xo);
#pragma unroll
                for (int j = 0; j < 12; j++) { xi[j] = xo[j]; z[n + hx<C>(k) + lane + j * P] = xo[j]; }
            }
        }
    } else if (ALG_ROLLOUT_OWN && lane < P) {
        double xi[C::ni], ui[C::mi];     // this player's entries
#pragma unroll
        for (int j = 0; j < C::ni; j++) xi[j] = z[lane + j * P];
        for (int k = 0; k < pr.N - 1; k++) {
#pragma unroll
            for (int j = 0; j < C::mi; j++) ui[j] = uget(k, lane, j);
            double xn[C::ni];
            model_player_rk3_own<C>(pr, xi, ui, pr.dt, xn);
#pragma unroll
            for (int j = 0; j < C::ni; j++) { xi[j] = xn[j]; z[n + hx<C>(k) + lane + j * P] = xn[j]; }
        }
    }
    else if (lane < P) {
        double x[n], u[m];     // only this player's entries are used
        for (int j = 0; j < C::ni; j++) x[lane + j * P] = z[lane + j * P];
        for (int k = 0; k < pr.N - 1; k++) {
            for (int j = 0; j < C::mi; j++) u[lane + j * P] = uget(k, lane, j);
            double xn[C::ni];
            model_player_rk3<C>(pr, lane, x, u, pr.dt, xn);
            for (int j = 0; j < C::ni; j++) { x[lane + j * P] = xn[j]; z[n + hx<C>(k) + lane + j * P] = xn[j]; }
        }
    }
    if (staged) game_sync();                                                // the scratch is the caller's LDS union again
}

// init_traj! (primal_dual_traj.jl:29-44) with the counter RNG (same element counters as the oracle)
template <class C, bool OL = false>
__device__ void init_traj(CPR pr, const Game& G, double* z, uint64_t game_id, bool use_shift, int shift = -1) {
    constexpr int n = C::n, m = C::m, P = C::P;
    const int N = pr.N, lane = OL ? phase_lane() : game_tid(); const auto& o = pr.opt;
    const int s = use_shift ? (shift >= 0 ? shift : o.shift) : (1 << 30);
    if (use_shift && s < N) {
        // in-place shift: element e of step k takes element e of step k+s; ascending k is safe within one wave only
        // with a barrier per step, so stage through the trial buffer
        double* tmp = G.z(1);
        for (int e = lane; e < pr.traj_len; e += C::NT) tmp[e] = z[e];
        game_sync();
        z = z; // (same buffer)
        for (int e = lane; e < pr.S; e += C::NT) {
            const int k = e / C::b, a = e % C::b;
            double v;
            if (a < n) {           // x_{k+1}: knot kn = k+1
                const int kn = k + 1;
                v = (kn + s <= N - 1) ? tmp[n + hx<C>(kn + s - 1) + a] : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)kn * (n + m) + a);
            } else if (a < n + m) { // u_k (player-grouped offset a-n -> joint index)
                const int off = a - n, i = off / C::mi, j = off % C::mi, c = i + j * P;
                v = (k + s < N - 1) ? tmp[n + hu<C>(k + s, 0) + off] : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)k * (n + m) + n + c);
            } else {               // lambda_{i,k}
                const int off = a - n - m, i = off / n, r = off % n;
                v = (k + s <= N - 2) ? tmp[n + hl<C>(k + s, i) + r]
                                     : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)N * (n + m) + ((uint64_t)i * (N - 1) + k) * n + r);
            }
            z[n + e] = v;
        }
    } else {
        for (int e = lane; e < pr.S; e += C::NT) {
            const int k = e / C::b, a = e % C::b;
            uint64_t ctr;
            if (a < n) ctr = (uint64_t)(k + 1) * (n + m) + a;
            else if (a < n + m) { const int off = a - n, i = off / C::mi, j = off % C::mi; ctr = (uint64_t)k * (n + m) + n + (i + j * P); }
            else { const int off = a - n - m, i = off / n, r = off % n; ctr = (uint64_t)N * (n + m) + ((uint64_t)i * (N - 1) + k) * n + r; }
            z[n + e] = o.amplitude_init * counter_uniform(o.seed, game_id, ctr);
        }
    }
    if (lane < n) z[lane] = G.x0(pr)[lane];
    game_sync();
}

// After an odd number of buffer exchanges pdtraj lives in the trial buffer: copy it home.  Both buffers then hold pdtraj, which is what the
// reference's pair holds after an accepted step (set_traj!(pdtraj, pdtraj_trial), solver_methods.jl:96: a copy, not an exchange).
// (until round 6 this exchanged the two buffers -- two reads and two writes per element, 2.1 % of the C2 solve's fabric traffic, to leave the
// previous iterate in a buffer nothing reads before the next line search overwrites it)
template <class C>
__device__ __forceinline__ void settle_traj(CPR pr, Game& G) {
    if (G.zo[0] != 0) {
        game_sync();
        const Game H = G.fresh();
        const double* a = H.z(0); double* z_home = H.base;
        // (eight elements per lane in flight: one element per trip exposed 33 global round trips in a row at C2, where the eleven iterations of a
        // solve always leave pdtraj in the trial buffer)
        constexpr int U = 8;
        const int TL = phase_int(pr.traj_len);
        for (int e0 = phase_lane(); e0 < TL; e0 += U * C::NT) {
            double va[U];
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; va[t] = gld(a, e < TL ? e : e0); }
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < TL) gst(z_home, e, va[t]); }
        }
        G.zo[1] = G.zo[0]; G.zo[0] = 0;
        game_sync();
    }
}

// penalty_update! scales mu of EVERY row of the constraint arena, including the rows of constraint kinds the problem does not have
// (no collision avoidance / no control bounds: their rows exist in the layout, nothing reads them).  The record! pass that carries the
// dual update (dual_in_record_v) visits the rows that exist; the others receive the solve's `nup` scalings here, once, in the same
// order of operations -- the arena ends up bit-identical to dual_penalty_update's.
template <class C>
__device__ void penalty_update_unused_rows(CPR pr0, const Game& G0, int nup) {
    CPR pr = phase_params(pr0);
    if (nup <= 0) return;
    const bool col_used = C::P > 1 && pr.has_colavoid, ctl_used = pr.has_ctl != 0;
    if (col_used && ctl_used) return;
    const Game G = G0.fresh();
    const auto& o = pr.opt;
    const int lo = col_used ? pr.col_len : 0, hi = ctl_used ? pr.col_len : pr.col_len + pr.ctl_len;
    constexpr int U = 4;
    for (int e0 = lo + phase_lane(); e0 < hi; e0 += U * C::NT) {
        double v[U];
#pragma unroll
        for (int q = 0; q < U; q++) { const int e = e0 + q * C::NT; v[q] = gld(G.mu(pr), e < hi ? e : e0); }
#pragma unroll
        for (int q = 0; q < U; q++) {
            const int e = e0 + q * C::NT;
            for (int t = 0; t < nup; t++) v[q] = fmin(fmax(v[q] * o.rho_increase, 0.0), o.rho_max);
            if (e < hi) gst(G.mu(pr), e, v[q]);
        }
    }
}

// Straggler hand-off (round 6; no reference counterpart -- the reference solves one game at a time).  A launch lasts as long as its slowest
// game: in a heterogeneous batch (Monte-Carlo scenarios, MPC warm starts) a few games need ten times the iterations of the rest and run
// the tail alone, one wavefront on an otherwise idle chip.  With alg_set_handoff(h, K) the one-wavefront kernel is launched with a budget:
// a game that is about to start its (K+1)-th inner iteration PARKS -- its iterate, multipliers, step records and statistics already live
// in its arena chunk; the solver's loop state (outer / inner index, LS_count, the trial-reuse flag, Delta, which buffer holds pdtraj) goes
// to the game's control slots -- and appends itself to the handle's queue; a second launch then RESUMES the parked games with the team
// kernel (four wavefronts per game), which continues the very same loops.  HO = 0: plain solve; 1: budgeted (may park); 2: resume.
constexpr int TC_HO_K = 18, TC_HO_L = 19, TC_HO_LS = 20, TC_HO_CV = 21, TC_HO_DELTA = 22, TC_HO_ZO = 23;
static_assert(TC_HO_ZO < TC_LEN, "per-game control slots");
template <class C>
__device__ __forceinline__ void handoff_park(CPR pr0, Game& G, int k, int l, int LS_count, int cache_valid, double Delta) {
    game_sync();
    // the scalings of mu that the fused dual updates deferred for the rows of absent constraint kinds: everything behind this point runs
    // dual_penalty_update on ALL rows (the team kernels are not fused kernels)
    if constexpr (dual_in_record_v<C>) penalty_update_unused_rows<C>(pr0, G, (k - 1) - (cache_valid == 2 ? 1 : 0));
    CPR pr = phase_params(pr0);
    if (phase_lane() == 0) {
        const Game H = G.fresh();
        double* tc = H.tc(pr);
        tc[TC_HO_K] = (double)k; tc[TC_HO_L] = (double)l; tc[TC_HO_LS] = (double)LS_count; tc[TC_HO_CV] = (double)cache_valid;
        tc[TC_HO_DELTA] = Delta; tc[TC_HO_ZO] = (double)G.zo[0];
        alg_game_stats* st = H.st(pr);
        st->status = ALG_STATUS_PARKED; st->outer_iters = k;
        int* q = as_global(pr.ho_queue);
        const int at = atomicAdd(q, 1);
        q[1 + at] = H.g;
    }
}

// newton_solve! (solver_methods.jl:5-65)
template <class C, int HO = 0, bool LOOP = false>
__device__ __forceinline__ void newton_solve(CPR pr, Game& G, Lds<C>& L, int init, uint64_t game_id, int shift = -1, int dual_reset = -1, int budget = 0) {
    const auto& o = pr.opt; const int lane = phase_lane();
    int k0 = 1, l0 = 1, ls0 = 0, cv0 = 0; double Delta = 0.0;
#ifdef ALG_PHASE_PROF
    if (lane < 16) G.res(pr)[lane] = 0.0;                                   // scratch instrumentation: per-phase cycle sums (16..: pass-level sums over the handle's lifetime)
    if (game_tid() < 32) lsp_slots()[game_tid()] = 0u;
#endif
    LSP_T0
#ifdef ALG_PHASE_PROF
    const unsigned lsp_solve0 = lsp_now();
#endif
    if constexpr (HO == 2) {
        // resume a parked game: the loop state from its control slots; pdtraj may live in the trial buffer (an odd number of exchanges)
        CPR prs = phase_params(pr); const double* tc = G.fresh().tc(prs);
        k0 = (int)uni(tc[TC_HO_K]); l0 = (int)uni(tc[TC_HO_L]); ls0 = (int)uni(tc[TC_HO_LS]); cv0 = (int)uni(tc[TC_HO_CV]);
        Delta = uni(tc[TC_HO_DELTA]);
        const int z0 = (int)uni(tc[TC_HO_ZO]);
        if (z0 != 0) { G.zo[1] = G.zo[0]; G.zo[0] = z0; }
        game_sync();
        if (cv0 == 2 && !dual_in_record_v<C>) {                              // the dual / penalty update the parked kernel left to its next record!
            dual_penalty_update<C>(pr, G);
            game_sync();
            cv0 = 0;
        }
    } else {
    if constexpr (LOOP) {
        // (inside k_mpc_loop this set-up works on a laundered view of the game and of the parameters like every phase behind it: its addresses and
        // lane predicates were invariants of the receding-horizon loop, i.e. live -- spilled -- across every solve: 25 -> 13 SGPR spills)
        const Game H = G.fresh(); CPR prq = phase_params(pr);
        if (lane == 0) { alg_game_stats z{}; *H.st(prq) = z; H.tc(prq)[TC_TELAP] = 0.0; }
        if (init) init_traj<C, true>(prq, H, H.z(0), game_id, true, shift);
        else { if (lane < C::n) H.z(0)[lane] = H.x0(prq)[lane]; }
        if (lane < C::n) { H.z(1)[lane] = H.x0(prq)[lane]; H.z(2)[lane] = 0.0; }
        game_sync();
        rollout<C, true>(prq, H.z(0), reinterpret_cast<double*>(&L), (int)(sizeof(Lds<C>) / sizeof(double)));
        if (dual_reset >= 0 ? dual_reset : phase_params(pr).opt.dual_reset) reset_con<C::NT>(phase_params(pr), G.fresh());
    } else {
    if (lane == 0) { alg_game_stats z{}; *G.fresh().st(phase_params(pr)) = z; G.fresh().tc(phase_params(pr))[TC_TELAP] = 0.0; } // reset!(prob.stats); t_elap = 0
    if (init) init_traj<C>(pr, G, G.z(0), game_id, true, shift);           // :13
    else { if (lane < C::n) G.z(0)[lane] = G.x0(pr)[lane]; }
    if (lane < C::n) { G.z(1)[lane] = G.x0(pr)[lane]; G.z(2)[lane] = 0.0; }    // :14-15 (only x_1 of the trial matters)
    game_sync();
    rollout<C>(pr, G.z(0), reinterpret_cast<double*>(&L), (int)(sizeof(Lds<C>) / sizeof(double)));     // :17
    if (dual_reset >= 0 ? dual_reset : o.dual_reset) reset_con<C::NT>(pr, G);     // :25
    }
    game_sync();
    }
    LSP(19)
    int out = 0, status = ALG_STATUS_OK, fresh = 0;
    int started = 0;                                                       // (HO == 1) inner iterations this launch has begun
    for (int k = k0; k <= o.outer_iter; k++) {                             // :30
        out = k;
        const bool first = HO == 2 && k == k0;                             // the outer iteration a resumed game re-enters
        int LS_count = first ? ls0 : 0;
        // fused kernels: every outer iteration but the first begins behind a dual / penalty update, which its first record! performs (2)
        int cache_valid = first ? cv0 : ((dual_in_record_v<C> && k > 1) ? 2 : 0);
        for (int l = first ? l0 : 1; l <= o.inner_iter; l++) {             // :38
            if constexpr (HO == 1) {
                if (started >= budget) { handoff_park<C>(pr, G, k, l, LS_count, cache_valid, Delta); return; }
                started += 1;
            }
            const int rcode = inner_iteration<C>(pr, G, L, LS_count, Delta, k, l, nullptr, &cache_valid);
            fresh = (rcode >> 16) & 1;
            if ((rcode & 0xff) != ALG_STATUS_OK) { status = rcode & 0xff; break; }
            if (LS_count >= 1 || ((rcode >> 8) & 0xff) == 1) break;        // :43
        }
        if (status != ALG_STATUS_OK) break;
        game_sync();
        // prob.stats.*_vio[end]: the record made at the top of the last inner iteration (lane 0 wrote it; same wave)
        alg_game_stats* const stk = G.fresh().st(phase_params(pr));
        const alg_record& last = stk->last;
        // (256-register kernels: the tolerances are re-read from the kernel-argument segment here -- as invariants of the outer loop they
        // were spilled across every phase of the solve; the 128-register double-integrator kernels have the scalar registers to keep them)
        const auto& oc = (C::WPE == 4) ? o : phase_params(pr).opt;
        const bool conv = last.dyn_vio < oc.eps_dyn && last.con_vio < oc.eps_con && last.sta_vio < oc.eps_sta && last.opt_vio < oc.eps_opt;
        const int convu = __builtin_amdgcn_readfirstlane((int)conv);
        if (convu && phase_lane() == 0) stk->converged = 1;          // written where it is decided (one loop-carried scalar less)
        if (k == oc.outer_iter || convu) break;                            // :49-55
        if constexpr (!dual_in_record_v<C>) {
            dual_penalty_update<C>(pr, G);                                 // :57-61
            game_sync();
        }                                                                  // (fused kernels: in the first record! of outer iteration k + 1)
    }
    game_sync();
    if constexpr (dual_in_record_v<C>) penalty_update_unused_rows<C>(pr, G, out - 1);
    // :63 record! at the final iterate.  When the solver left its loops at the optimality test of an inner iteration (the usual
    // exit) that iteration's record! was made at this very iterate with these very multipliers: the same numbers, so the
    // assemble pass is not repeated, the record is pushed again (with the Delta and outer index this call passes)
    if (fresh && status == ALG_STATUS_OK) {
        if (phase_lane() == 0) {
            CPR prs = phase_params(pr); const Game Gs = G.fresh();
            alg_game_stats* st = Gs.st(prs);
            st->last.outer = out; st->last.delta = Delta; st->last.alpha = 0.0; st->last.ls_j = 0; st->last.t_elap = Gs.tc(prs)[TC_TELAP];
            const int idx = st->records;
            if (idx < prs.hist_max) {
                alg_record* dst = Gs.hist(prs) + idx; const alg_record* src = &st->last;
                dst->outer = src->outer; dst->ls_j = src->ls_j; dst->alpha = src->alpha; dst->res = src->res; dst->delta = src->delta;
                dst->dyn_vio = src->dyn_vio; dst->con_vio = src->con_vio; dst->sta_vio = src->sta_vio; dst->opt_vio = src->opt_vio; dst->t_elap = src->t_elap;
            }
            st->records = idx + 1;
        }
    } else make_record<C>(pr, G, L, Delta, out, 0.0, nullptr);
    settle_traj<C>(pr, G);
    if (phase_lane() == 0) { alg_game_stats* st = G.fresh().st(phase_params(pr)); st->status = status; st->outer_iters = out; }
#ifdef ALG_PHASE_PROF
    lsp_add(31, lsp_solve0);            // the whole solve
    game_sync();
    if (game_tid() >= 16 && game_tid() < 32) G.fresh().res(phase_params(pr))[game_tid()] += (double)lsp_slots()[game_tid()];
    game_sync();
#endif
}

// ================================================================================================
// Iterated best response (solver_methods.jl:133-289)
// ================================================================================================
// record!(stats, ..., k, i) (statistics.jl:59-73): full residual norm + player-specific violations; also tracks
// maximum(stats.Δ_traj) (G.tc(pr)[6]) for the exit test of ibr_newton_solve! (:157).  Returns the masked norm / opt violation.
template <class C>
__device__ __forceinline__ RecScalars ibr_push_stats(CPR pr, const Game& G, const ResOut& ro, double delta, int outer) {
    const double sm = (double)((pr.N - 1) * (2 * C::n + C::mi));            // length(verti_mask)
    if (game_tid() == 0) {
        alg_record rc;
        rc.outer = outer; rc.ls_j = 0; rc.alpha = 0.0; rc.res = ro.l1full / (double)pr.S; rc.delta = delta;
        rc.dyn_vio = ro.dyn; rc.con_vio = ro.con; rc.sta_vio = ro.sta; rc.opt_vio = ro.opt;
        rc.t_elap = G.tc(pr)[TC_TELAP];
        const int idx = G.st(pr)->records;
        if (idx < pr.hist_max) G.hist(pr)[idx] = rc;
        G.st(pr)->records = idx + 1;
        G.st(pr)->last = rc;
        G.tc(pr)[6] = fmax(G.tc(pr)[6], delta);
    }
    RecScalars r; r.res = uni(ro.l1 / sm); r.opt = uni(ro.opt); r.nonfinite = __builtin_amdgcn_readfirstlane(ro.nonfinite);
    return r;
}
// ibr_inner_iteration (solver_methods.jl:230-268)
template <class C>
__device__ int ibr_inner_iteration(CPR pr, const Game& G, Lds<C>& L, int& LS_count, double& Delta, int k, int l, int ip) {
    const auto& o = pr.opt;
    const double lf = (double)l;
    const double reg = o.reg_0 * (lf * lf * lf * lf);
    const double sm = (double)((pr.N - 1) * (2 * C::n + C::mi));
    ResOut ro;
    iter_clock_start(pr, G);                                               // t_elap = @elapsed ibr_inner_iteration (:151-153)
    assemble_pass<C, 1, true>(pr, G, L.a, 0, -1, 0.0, reg, ro, ip);        // :236-241
    game_sync();
    const RecScalars rs = ibr_push_stats<C>(pr, G, ro, Delta, k);
    Delta = 0.0;
    if (rs.nonfinite) { iter_clock_stop(pr, G); return ALG_STATUS_NAN | (1 << 8); }
    if (rs.opt < o.eps_opt) { iter_clock_stop(pr, G); return ALG_STATUS_OK | (1 << 8); }   // :245-247
    const int st = refined_direction<C, true>(pr, G, L, reg, ip, nullptr);          // :249-252
    if (st != ALG_STATUS_OK) { iter_clock_stop(pr, G); return st | (1 << 8); }
    game_sync();
    int j = 1; double alpha = 1.0;                                                  // ibr_line_search (:270-289)
    while (j < o.ls_iter) {
        update_traj<C>(pr, G, 1, 0, alpha);
        game_sync();
        ResOut rt;
        assemble_pass<C, 0, true>(pr, G, L.a, 1, o.regularize ? 0 : -1, reg, 0.0, rt, ip);
        if (uni(rt.l1 / sm) <= (1.0 - alpha * o.beta) * rs.res) break;
        alpha *= o.alpha_decrease; j += 1;
    }
    const int failed = (j == o.ls_iter);
    if (failed) LS_count += 1; else LS_count = 0;
    game_sync();
    update_traj<C>(pr, G, 0, 0, alpha);                              // :258
    Delta = uni(delta_step<C>(pr, G.z(2), alpha));                                  // :259
    game_sync();
    if (game_tid() == 0) {
        G.st(pr)->newton_iters += 1; if (failed) G.st(pr)->ls_failures += 1;
        const int idx = G.st(pr)->records - 1;
        if (idx < pr.hist_max) { G.hist(pr)[idx].alpha = alpha; G.hist(pr)[idx].ls_j = j; }
        G.st(pr)->last.alpha = alpha; G.st(pr)->last.ls_j = j;
    }
    iter_clock_stop(pr, G);
    return ALG_STATUS_OK | ((Delta < o.delta_min ? 1 : 0) << 8);
}
// ibr_newton_solve!(prob, i) (solver_methods.jl:171-228)
template <class C>
__device__ int ibr_solve_player(CPR pr, const Game& G, Lds<C>& L, int ip) {
    const auto& o = pr.opt; const int lane = game_tid();
    if (o.dual_reset) {                                                            // :181-185
        reset_con(pr, G);
        for (int e = lane; e < (pr.N - 1) * C::P * C::n; e += WAVE) {              // reset_duals!(pdtraj), reset_duals!(pdtraj_trial)
            const int k = e / (C::P * C::n), a = e % (C::P * C::n);
            G.z(0)[C::n + hl<C>(k, 0) + a] *= 0.0; G.z(1)[C::n + hl<C>(k, 0) + a] *= 0.0;
        }
    }
    game_sync();
    int out = 0, status = ALG_STATUS_OK, converged = 0; double Delta = 0.0;
    for (int k = 1; k <= o.outer_iter; k++) {
        out = k; int LS_count = 0;
        for (int l = 1; l <= o.inner_iter; l++) {
            const int rcode = ibr_inner_iteration<C>(pr, G, L, LS_count, Delta, k, l, ip);
            if ((rcode & 0xff) != ALG_STATUS_OK) { status = rcode & 0xff; break; }
            if (LS_count >= 1 || (rcode >> 8) == 1) break;
        }
        if (status != ALG_STATUS_OK) break;
        game_sync();
        const alg_record& last = G.st(pr)->last;
        const bool conv = last.dyn_vio < o.eps_dyn && last.con_vio < o.eps_con && last.sta_vio < o.eps_sta && last.opt_vio < o.eps_opt;
        const int convu = __builtin_amdgcn_readfirstlane((int)conv);
        converged = convu;
        if (k == o.outer_iter || convu) break;
        dual_penalty_update<C>(pr, G);
        game_sync();
    }
    game_sync();
    ResOut ro;
    assemble_pass<C, 1, true>(pr, G, L.a, 0, -1, 0.0, 0.0, ro, ip);          // :226
    game_sync();
    ibr_push_stats<C>(pr, G, ro, Delta, out);
    if (lane == 0) { G.st(pr)->status = status; G.st(pr)->outer_iters = out; G.st(pr)->converged = converged; }
    game_sync();
    return status;
}
struct IbrOrder { int v[MAXP]; };
// ibr_newton_solve!(prob; ibr_opts) (solver_methods.jl:133-169); single = true runs ibr_newton_solve!(prob, player) on the stored
// trajectory instead (one best response, no initialisation).  One call site of ibr_solve_player: it stays inlined.
template <class C>
__device__ void ibr_newton_solve(CPR pr, const Game& G, Lds<C>& L, bool single, int player, int init, uint64_t game_id,
                                 int ibr_iter, const IbrOrder& order, double delta_min) {
    const int lane = game_tid();
    if (!single) {
        if (lane == 0) { alg_game_stats z{}; *G.st(pr) = z; G.tc(pr)[6] = 0.0; G.tc(pr)[TC_TELAP] = 0.0; }             // reset!(prob.stats); the first record carries t_elap = 0
        if (init) init_traj<C>(pr, G, G.z(0), game_id, true);
        else { if (lane < C::n) G.z(0)[lane] = G.x0(pr)[lane]; }
        game_sync();
        for (int e = lane; e < pr.traj_len; e += WAVE) { G.z(1)[e] = G.z(0)[e]; G.z(2)[e] = 0.0; }   // :142-143 (the trial's duals are reset below)
        game_sync();
        rollout<C>(pr, G.z(0));
        game_sync();
    }
    unsigned change = (1u << C::P) - 1u;                                             // Δ_change = trues(p)
    const int rounds = single ? 1 : ibr_iter, nplay = single ? 1 : C::P;
    for (int q = 0; q < rounds; q++) {
        for (int id = 0; id < nplay; id++) {
            const int ip = single ? player : order.v[id];
            const int status = ibr_solve_player<C>(pr, G, L, ip);
            if (single) return;
            const double mx = uni(G.tc(pr)[6]);
            if (!(delta_min > mx)) change |= (1u << ip); else change &= ~(1u << ip);  // :157
            if (status != ALG_STATUS_OK) return;
        }
        if (change == 0u) break;                                                    // :163
    }
}


} // namespace alg
