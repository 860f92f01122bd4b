// algames_assemble.hpp -- assemble pass (residual! / record! / line-search trials), fused trial pass, update_traj!
// (part of the device code of libalgames_hip.so; included by algames_device.hpp, which holds the shared declarations and the file-level
// description of the execution model)
#pragma once
#include "algames_device.hpp"

// Round-6 A/B switches of the fused pass (tests/probes/build_variant.sh; 1 = shipped)
#ifndef ALG_R6_ROWIDX
#define ALG_R6_ROWIDX 1           // fused pass, rows opt_x: (step, entry) of a row carried from trip to trip
#endif
#ifndef ALG_R6_LANEROLE
#define ALG_R6_LANEROLE 1         // fused pass, double integrator: a lane keeps its row of the step for the whole chunk, the trips walk the steps (2: the staging too)
#endif
#ifndef ALG_LSM_DI1W
#define ALG_LSM_DI1W 0            // (documented at LsMulti below)
#endif
#ifndef ALG_R6_LANEROLE_UNI
#define ALG_R6_LANEROLE_UNI 0     // ... the one-wavefront unicycle kernels too (the line search's group pass then deals its rows the same way: bit-identical to the
                                  // fused pass, tests green).  Measured and NOT taken (profiles/r06_ab_lru_*.txt): C5 at 4096 / 1024 games + 0.6 / + 0.8 %, C3 at 4096
                                  // games (P n = 64: the flat dealing wastes no lane there) - 2.5 %, and the receding-horizon loop at 4096 seeds 3.83 -> 3.16 M/s --
                                  // another rounding of the norms, other closed-loop trajectories, a slower straggler
#endif
#ifndef ALG_R6_STAGE
#define ALG_R6_STAGE 1      // every global load of a chunk in flight before the first wait
#endif
#ifndef ALG_R6_STAGE_BATCH
#define ALG_R6_STAGE_BATCH 6    // ... elements of z and dz per lane and batch in the 128-register kernels
#endif
#ifndef ALG_R6_STAGE_BATCH_REC
#define ALG_R6_STAGE_BATCH_REC 6 // ... elements of z per lane and batch in the record pass of the 128-register kernels (no dz; all twelve in one batch measured: neutral, profiles/r06_ab_rec12_c2.txt)
#endif
#ifndef ALG_R6_STAGE_BATCH_LR
#define ALG_R6_STAGE_BATCH_LR 4 // ... blocks per batch of the lane-role staging
#endif
#ifndef ALG_R6_STAGE_BATCH_W2
#define ALG_R6_STAGE_BATCH_W2 10   // ... in the 256-register kernels (the 4-player unicycle's chunk is 20 elements per lane: 40 doubles in flight at once spilled its loop kernel)
#endif
#ifndef ALG_R6_PHASEA_CHUNK
#define ALG_R6_PHASEA_CHUNK 1   // fused pass: phase A per chunk out of the staged blocks, pair-gradient tables straight into the chunk's LDS (no global round trip)
#endif
#ifndef ALG_R6_PHASEA
#define ALG_R6_PHASEA 1     // phase A: multipliers / penalties of an item's pairs requested together with its positions; LDS-only fences between the staged write-outs
#endif

namespace alg {

// ================================================================================================
// Assemble pass: residual! + regularize_residual! + the scalars of record! (+ step records)
//   global_quantities.jl:9-86, statistics.jl:44-57, violations.jl.
//   phase A (parallel, work item = (knot, player)): RK2 Jacobian coefficients, collision cost / collision avoidance
//           (/ wall / circle) terms of the ordered pairs (i, j) -> record [coef | Hh | Hd | gvt], constraint values
//   phase B (parallel, work item = one residual row of one step; three flat row loops):
//           rows opt_i,x_{k+1} | opt_i,u_{i,k} | dyn_k  -> record [rx | ru | rd], R^ (, RQ), statistics
//   MODE 0: statistics only (line-search trials)   MODE 1: + step records (Newton direction input)
//   MODE 2: + residual vector in the reference's vertical order and the constraint values (alg_residual)
//   MODE 3: line-search trial that doubles as the next record!: statistics and step records of the UNREGULARISED
//           residual (what record! sees if the trial is accepted) plus the regularised norm l1reg for the acceptance test
// With zref != nullptr the proximal term reg (x - xref) is added to the rows; out.l1 is the norm of those rows
// (MODE 0/2) or of the unregularised rows (MODE 3, which also returns out.l1reg).
// ================================================================================================
struct ResOut { double l1, opt, dyn, con, sta; int nonfinite; double l1reg; double l1full; };
// Combines the per-wavefront statistics of a team (fixed order: deterministic); every thread leaves with the same values.
template <class C> __device__ __forceinline__ void team_combine(ResOut& o) {
    if constexpr (C::NW > 1) {
        __shared__ double red[C::NW][8];
        const int w = game_tid() >> 6, l = game_tid() & 63;
        if (l == 0) { red[w][0] = o.l1; red[w][1] = o.opt; red[w][2] = o.dyn; red[w][3] = o.con; red[w][4] = o.sta; red[w][5] = (double)o.nonfinite; red[w][6] = o.l1reg; red[w][7] = o.l1full; }
        game_sync();
        ResOut r = {0.0, 0.0, 0.0, 0.0, 0.0, 0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < C::NW; q++) {
            r.l1 += red[q][0]; r.opt = fmax(r.opt, red[q][1]); r.dyn = fmax(r.dyn, red[q][2]); r.con = fmax(r.con, red[q][3]); r.sta = fmax(r.sta, red[q][4]);
            r.nonfinite |= (int)red[q][5]; r.l1reg += red[q][6]; r.l1full += red[q][7];
        }
        game_sync();                         // red[] may be rewritten by the next pass
        o = r;
    }
}

// rows of the fused pass dealt as (step, row of the step) = (trip, lane): see assemble_fused.  The group pass of the line search (trial_norms_multi) on a
// one-wavefront kernel must sum its rows per lane in the order of the fused pass (bit-identical norms): it follows this switch.
template <class C> inline constexpr bool lane_roles_v = ALG_R6_LANEROLE != 0 && AsmLds<C>::FUSED && C::P * C::n <= C::NT &&
    ((C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR && !ALG_LSM_DI1W) || (ALG_R6_LANEROLE_UNI != 0 && C::MODEL == ALG_MODEL_UNICYCLE));
struct AsmAcc { double l1 = 0, l1r = 0, l1f = 0, vopt = 0, vdyn = 0, vcon = 0, vsta = 0; int bad = 0; };
// Roundings of the pair terms, stated once.  Phase A of the assemble pass, phase A of the line search's group pass (trial_norms_multi) and
// dual_penalty_update evaluate the same pair expressions, and the group pass's norms must equal the ordinary pass's bit for bit: wherever a
// product feeds a sum the fma is written out, so that no two copies can be contracted differently (round 6: a restructured phase A came
// out an ulp apart from the group pass's copy in `a a + b b`).
__device__ __forceinline__ double pair_dist2(double d0, double d1) { return __builtin_fma(d0, d0, d1 * d1); }
__device__ __forceinline__ double ca_value(double on, double Rr, double s2) { return on * __builtin_fma(Rr, Rr, -s2); }     // c = on (R^2 - |d|^2), constraints_methods.jl:21-33
__device__ __forceinline__ double dual_ascent(double lam, double a, double mu, double c, double lam_max) {                    // dual_update!, constraints_methods.jl:421-440
    return fmin(fmax(__builtin_fma(a * mu, c, lam), 0.0), lam_max);
}
// tab[f(i)] of a small table in the kernel-argument segment for a per-lane player index i < NP: NP scalar loads and a select chain instead of
// a vector load from constant memory (a global round trip in the middle of a phase-A item's dependency chain)
template <int NP, class T, class F>
__device__ __forceinline__ T sel_player(const ALG_AS4 T* tab, int i, F&& f) {
    T v = tab[f(0)];
#pragma unroll
    for (int q = 1; q < NP; q++) v = (i == q) ? tab[f(q)] : v;
    return v;
}
// The pair / own-position terms of ONE item of phase A -- (knot k + 1, player i): collision cost and collision avoidance of the ordered pairs
// (i, j), the extended set's wall / circle terms -- shared by the pass over all steps (assemble_phase_a) and the per-chunk form of the fused pass
// (round 6).  xp(idx): entry idx of x_{k+1} of the (trial) iterate; lmu(jj, ci, lam, mu): multiplier and penalty of constraint row ci (pair jj);
// rec: the step's record head [.. Hh | Hd ..] (or its staging slot), tab: the step's pair-gradient table.
template <class C, int MODE, bool IBR, bool DUAL, class XP, class LMU>
__device__ __forceinline__ void phase_a_pos_item(CPR pr, const Game& G, int N, int k, int i, int ip, double dt, bool pairs_on, XP&& xp, LMU&& lmu,
                                                 double* __restrict__ rec, double* __restrict__ tab, AsmAcc& acc, bool dual) {
    constexpr int n = C::n, P = C::P, PD = C::PD, NS = C::NS;
    using R = Rec<C>;
    constexpr bool RECS = (MODE == 1 || MODE == 2 || MODE == 3);
    const int kn = k + 1;
    const double w = (kn < N - 1) ? dt : 1.0;
        double xi[PD], ga[PD], dd[NS];
#pragma unroll
        for (int a = 0; a < PD; a++) { xi[a] = xp(a * P + i); ga[a] = 0.0; }
#pragma unroll
        for (int t = 0; t < NS; t++) dd[t] = 0.0;
#if ALG_R6_PHASEA
        // Round 6: everything an item reads from global memory is requested before its arithmetic starts -- the positions of the other
        // players and the multiplier / penalty of every pair (they sat behind the sqrt / division chains of the pair before: one exposed
        // round trip per pair) -- and the per-player constants come from scalar loads (sel_player).
        constexpr int NPR = P > 1 ? P - 1 : 1;
        double xj[NPR][PD], lmq[NPR], muq[NPR];
        if (pairs_on) {
#pragma unroll
            for (int jj = 0; jj < P - 1; jj++) {
                const int j = jj < i ? jj : jj + 1;
#pragma unroll
                for (int a = 0; a < PD; a++) xj[jj][a] = xp(a * P + j);
                lmq[jj] = 0.0; muq[jj] = 0.0;
                if (pr.has_colavoid) lmu(jj, con_col<C>(N, pairq<C>(i, j), kn), lmq[jj], muq[jj]);
            }
        }
        const double cc_mu_i = pr.has_colcost ? sel_player<P>(pr.cc_mu, i, [](int q) { return q; }) : 0.0;
        const double cc_rad_i = pr.has_colcost ? sel_player<P>(pr.cc_radius, i, [](int q) { return q; }) : 0.0;
        const unsigned ca_mask_i = sel_player<P>(pr.ca_mask, i, [](int q) { return q; });
#endif
#pragma unroll
        for (int jj = 0; jj < P - 1; jj++) {
            const int j = jj < i ? jj : jj + 1;
            double gv[PD], H[NS];
#pragma unroll
            for (int a = 0; a < PD; a++) gv[a] = 0.0;
#pragma unroll
            for (int t = 0; t < NS; t++) H[t] = 0.0;
            if (pairs_on) {
                double dl[PD];
#if ALG_R6_PHASEA
#pragma unroll
                for (int a = 0; a < PD; a++) dl[a] = xi[a] - xj[jj][a];
#else
#pragma unroll
                for (int a = 0; a < PD; a++) dl[a] = xi[a] - xp(a * P + j);
#endif
                const double dl0 = dl[0], dl1 = dl[1];
                const double s2 = pair_dist2(dl0, dl1);
                if (pr.has_colcost) {                                    // CollisionCost, objective.jl:134-173 (planar: px[i])
#if ALG_R6_PHASEA
                    const double nrm = sqrt(s2), mu = cc_mu_i, rad = cc_rad_i;
#else
                    const double nrm = sqrt(s2), mu = pr.cc_mu[i], rad = pr.cc_radius[i];
#endif
                    if (fmax(0.0, rad - nrm) > 0.0) {
                        const double eps = 1e-10, eps_norm = eps * sqrt((double)n);
                        const double g0 = mu * (rad * (eps + dl0) / (eps_norm + nrm) - dl0);
                        const double g1 = mu * (rad * (eps + dl1) / (eps_norm + nrm) - dl1);
                        gv[0] += w * (-g0); gv[1] += w * (-g1);
                        const double n3 = nrm * nrm * nrm;
                        H[0] += w * (mu * (1.0 - rad / nrm + rad * (dl0 * dl0) / n3));
                        H[1] += w * (mu * (rad * (dl0 * dl1) / n3));
                        H[2] += w * (mu * (1.0 - rad / nrm + rad * (dl1 * dl1) / n3));
                    }
                }
                if (pr.has_colavoid) {                                   // CollisionConstraint + AL expansion
#if ALG_R6_PHASEA
                    const double Rr = sel_player<P>(pr.ca_pair_r, i, [jj](int q) { return q * MAXP + (jj < q ? jj : jj + 1); });
                    const double on = (double)((ca_mask_i >> j) & 1u);                        // 0: this ordered pair carries no constraint
#else
                    const double Rr = pr.ca_pair_r[i * MAXP + j];
                    const double on = (double)((pr.ca_mask[i] >> j) & 1u);                    // 0: this ordered pair carries no constraint
#endif
                    double s2c = s2;
                    if constexpr (PD == 3) { if (pr.ca_dim != 3) dl[2] = 0.0; s2c = __builtin_fma(dl[2], dl[2], s2c); }   // spherical: pz[i][1:3]
                    const double c = ca_value(on, Rr, s2c);
                    const int ci = con_col<C>(N, pairq<C>(i, j), kn);
#if ALG_R6_PHASEA
                    double lm = lmq[jj], mu_c = muq[jj];
#else
                    double lm = gld(G.lam(pr), ci), mu_c = gld(G.mu(pr), ci);
#endif
                    if (DUAL && dual) {
                        // dual_update! with alphax_dual[i], then penalty_update! (constraints_methods.jl:421-440, 329-379): dual_penalty_update's expressions
                        const auto& od = pr.opt;
#if ALG_R6_PHASEA
                        const double ax = sel_player<P>(od.alphax_dual, i, [](int q) { return q; });
#else
                        const double ax = od.alphax_dual[i];
#endif
                        lm = dual_ascent(lm, ax, mu_c, c, od.lambda_max);
                        mu_c = fmin(fmax(mu_c * od.rho_increase, 0.0), od.rho_max);
                        gst(G.lam(pr), ci, lm); gst(G.mu(pr), ci, mu_c); gst(G.vals(pr), ci, c);
                    }
                    const double am = on * al_active_mu(c, lm, mu_c);
                    const double wl = fma(am, c, on * lm);            // (the contraction the all-pairs form always had: lm + am c)
#pragma unroll
                    for (int a = 0; a < PD; a++) {
                        gv[a] = __builtin_fma(-2.0 * dl[a], wl, gv[a]);
#pragma unroll
                        for (int a2 = 0; a2 <= a; a2++) H[C::sym(a, a2)] += am * 4.0 * dl[a2] * dl[a];
                    }
                    if (MODE == 2) G.vals(pr)[ci] = c; if (!IBR || i == ip) acc.vsta = fmax(acc.vsta, fmax(0.0, c));
                }
            }
#pragma unroll
            for (int a = 0; a < PD; a++) { ga[a] += gv[a]; tab[(i * P + j) * PD + a] = -gv[a]; }   // row opt_i at px(j,.)
#pragma unroll
            for (int t = 0; t < NS; t++) dd[t] += H[t];
            if (RECS) {
                double* hh = rec + R::HH + NS * pairq<C>(i, j);
#pragma unroll
                for (int t = 0; t < NS; t++) hh[t] = H[t];
            }
        }
        if constexpr (C::EXT) {
            // wall / circle constraints of player i on its own position at knot k+1: AL gradient C'(lambda + a mu c)
            // and Gauss-Newton Hessian C' a mu C (constraint_derivatives.jl:10-19,47-58) join the (i,i) position block
            auto al_row = [&](int ci, double c, const double (&g)[PD]) {
                const double lm = gld(G.lam(pr), ci), am = al_active_mu(c, lm, gld(G.mu(pr), ci));
                const double wl = lm + am * c;
#pragma unroll
                for (int a = 0; a < PD; a++) {
                    ga[a] += g[a] * wl;
#pragma unroll
                    for (int a2 = 0; a2 <= a; a2++) dd[C::sym(a, a2)] += am * g[a2] * g[a];
                }
                if (MODE == 2) G.vals(pr)[ci] = c; if (!IBR || i == ip) acc.vsta = fmax(acc.vsta, fmax(0.0, c));
            };
            const double* Wc = ext_walls(pr, pr.extc); const double* Cc = ext_circs(pr, pr.extc);
            const unsigned wmask = pr.wall_mask[i], cmask = pr.circ_mask[i];
            for (int wq = 0; wq < pr.nwall; wq++) {
                double g[PD] = {}; const double on = (double)((wmask >> wq) & 1u);
                const double c = on * wall_val(Wc, wq, xi[0], xi[1], &g[0], &g[1]); g[0] *= on; g[1] *= on;
                al_row(ext_wall_row(pr, i, k, wq), c, g);
            }
            for (int cq = 0; cq < pr.ncirc; cq++) {
                double g[PD] = {}; const double on = (double)((cmask >> cq) & 1u);
                const double c = on * circ_val(Cc, cq, xi[0], xi[1], &g[0], &g[1]); g[0] *= on; g[1] *= on;
                al_row(ext_circ_row(pr, i, k, cq), c, g);
            }
            if constexpr (PD == 3) {
                const double* W3 = ext_walls3(pr, pr.extc); const double* Yc = ext_cyls(pr, pr.extc);
                const unsigned w3mask = pr.wall3_mask[i], cymask = pr.cyl_mask[i];
                for (int wq = 0; wq < pr.nwall3; wq++) {
                    double g[3]; const double on = (double)((w3mask >> wq) & 1u);
                    const double c = on * wall3_val(W3, wq, xi, g); g[0] *= on; g[1] *= on; g[2] *= on;
                    al_row(ext_wall3_row(pr, i, k, wq), c, g);
                }
                for (int cq = 0; cq < pr.ncyl; cq++) {
                    double g[3]; const double on = (double)((cymask >> cq) & 1u);
                    const double c = on * cyl_val(Yc, cq, xi, g); g[0] *= on; g[1] *= on; g[2] *= on;
                    al_row(ext_cyl_row(pr, i, k, cq), c, g);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < PD; a++) tab[(i * P + i) * PD + a] = ga[a];           // row opt_i at px(i,.)
        if (RECS) {
#pragma unroll
            for (int t = 0; t < NS; t++) rec[R::HD + NS * i + t] = dd[t];
        }
}

// Phase A of the assemble pass (see assemble_pass): RK2 Jacobian coefficients and the pair / wall / circle terms of every (knot, player).
// dzp != nullptr: the positions are those of the trial iterate z + alpha dz, formed on the fly (fused trial pass of the double integrator).
// DUAL (round 6): the pass is the record! that follows dual_update! + penalty_update! (solver_methods.jl:57-61 then :73): both stream the same
// iterate and the same multipliers, so the item that evaluates a collision-avoidance row also updates its lambda and mu -- in registers,
// with dual_penalty_update's expressions -- stores them together with the constraint value (evaluate!), and forms the augmented-Lagrangian
// terms with the new values.  One pass instead of two per outer iteration.
// (DUAL is the compile-time capability -- the record! instantiation of the fused pass -- and `dual` the wave-uniform request of this call: the
// solver kernel holds ONE record! instantiation, not two)
template <class C, int MODE, bool IBR, bool DUAL = false>
__device__ __forceinline__ void assemble_phase_a(CPR pr, const Game& G, AsmLds<C>& L, const double* __restrict__ z, const double* __restrict__ dzp, double alpha,
                                                 int N, int lane, double dt, int ip, AsmAcc& acc, bool dual = false) {
    constexpr int n = C::n, P = C::P;
    using R = Rec<C>;
    constexpr bool RECS = (MODE == 1 || MODE == 2 || MODE == 3);
    if (C::NC > 0 || C::POS) {
        const bool pairs_on = P > 1 && (pr.has_colcost || pr.has_colavoid);
        constexpr int HEAD = AsmLds<C>::HEAD, SL = AsmLds<C>::SL, SPP = AsmLds<C>::SPP;
        for (int kA = 0; kA < N - 1; kA += SPP) {
          const int ks = lane / P, i = lane % P, k = kA + ks, kn = k + 1;
          constexpr bool STAGED = AsmLds<C>::STAGED;
          if (lane < SPP * P && k < N - 1) {
            // staged: record head (offsets as in the record) + table at HEAD of this step's LDS slot; else the record itself
            double* __restrict__ rec = STAGED ? L.stage + ks * SL : G.rec(pr) + (size_t)k * R::LEN;
            double* __restrict__ tab = STAGED ? L.stage + ks * SL + HEAD : G.rec(pr) + R::gvt(N, k);       // pair-gradient table of the step
            if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
                const double* sk = zstate<C>(z, k);
                double cf[10];
                bike_coefs<C>(pr, sk[2 * P + i], sk[3 * P + i], z[n + hu<C>(k, i)], z[n + hu<C>(k, i) + 1], dt, cf);
#pragma unroll
                for (int t = 0; t < 10; t++) rec[R::COEF + t * P + i] = cf[t];
            } else if constexpr (C::MODEL == ALG_MODEL_UNICYCLE) {
                // Jacobian coefficients of knot k (A_k, B_k): see the model section (of the trial iterate when dzp is given)
                const double* sk = zstate<C>(z, k); const double* dk = (dzp && k > 0) ? zstate<C>(dzp, k) : nullptr;      // (x_1 does not move)
                const int uo_ = n + hu<C>(k, i);
                auto sv = [&](int idx) { const double q = sk[idx]; return dk ? __builtin_fma(alpha, dk[idx], q) : q; };      // (update_traj!'s fma: the value the trial buffer holds)
                auto uv = [&](int j) { const double q = z[uo_ + j]; return dzp ? __builtin_fma(alpha, dzp[uo_ + j], q) : q; };
                const double th = sv(2 * P + i), v = sv(3 * P + i);
                const double om = uv(0), ac = uv(1);
                const double thm = th + (om * dt) * 0.5, vm = v + (ac * dt) * 0.5;
                double sn, cs; sincos(thm, &sn, &cs);
                rec[R::COEF + 0 * P + i] = -dt * vm * sn; rec[R::COEF + 1 * P + i] = dt * cs;
                rec[R::COEF + 2 * P + i] = dt * vm * cs;  rec[R::COEF + 3 * P + i] = dt * sn;
            }
            if constexpr (C::POS) {
                const double* x1 = z + n + hx<C>(k); const double* d1 = dzp ? dzp + n + hx<C>(k) : nullptr;
                auto xp = [&](int idx) { const double v = gld(x1, idx); return d1 ? __builtin_fma(alpha, gld(d1, idx), v) : v; };    // position of the (trial) iterate: update_traj!'s fma
                auto lmu = [&](int, int ci, double& lm, double& mu_c) { lm = gld(G.lam(pr), ci); mu_c = gld(G.mu(pr), ci); };
                phase_a_pos_item<C, MODE, IBR, DUAL>(pr, G, N, k, i, ip, dt, pairs_on, xp, lmu, rec, tab, acc, dual);
            }
          }
          if constexpr (STAGED) {
              // (round 6: the two fences around the write-out order LDS only -- the staging slots are what they protect; the global stores
              // are waited for once, behind the loop)
              if constexpr (ALG_R6_PHASEA != 0) sweep_sync<C>(); else game_sync();
              // write-out: contiguous [coef | Hh | Hd] and table segments of the staged steps
              const int nst = (N - 1 - kA) < SPP ? (N - 1 - kA) : SPP;
              for (int t = lane; t < nst * SL; t += C::NT) {
                  const int ks2 = t / SL, o = t % SL;
                  const size_t base = (size_t)(kA + ks2) * R::LEN;
                  if (o >= HEAD) G.rec(pr)[R::gvt(N, kA + ks2) + (o - HEAD)] = L.stage[t];
                  else if (RECS || o < C::NC) G.rec(pr)[base + o] = L.stage[t];
              }
              if constexpr (ALG_R6_PHASEA != 0) sweep_sync<C>(); else game_sync();
          }
        }
        if constexpr (!AsmLds<C>::STAGED || ALG_R6_PHASEA != 0) game_sync();
    }
}

// IBR = true: best-response statistics of player ip (solver_methods.jl:230-289): norms over the rows of the vertical mask
// (player ip's opt rows + all dyn rows, newton_core.jl:205-246), player-specific violations (statistics.jl:59-73), the
// proximal term only on player ip's rows (global_quantities.jl:262-280); out.l1full is the full ||res||_1 for record!.
template <class C, int MODE, bool IBR = false>
__device__ void assemble_pass(CPR pr0, const Game& G0, AsmLds<C>& L, int zsel, int zrefsel, double reg, double jreg,
                              ResOut& out, int ip = -1) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    const double* __restrict__ z = G.z(zsel);
    const double* __restrict__ zref = zrefsel >= 0 ? G.z(zrefsel) : nullptr;
    constexpr int n = C::n, m = C::m, P = C::P, mi = C::mi, ni = C::ni, b = C::b;
    using R = Rec<C>;
    const int N = phase_int(pr.N), lane = phase_lane();
    const double dt = phase_f64(pr.dt);
    AsmAcc acc;
    double& l1 = acc.l1; double& l1r = acc.l1r; double& l1f = acc.l1f; double& vopt = acc.vopt; double& vdyn = acc.vdyn; double& vcon = acc.vcon; double& vsta = acc.vsta; int& bad = acc.bad;
    constexpr bool RECS = (MODE == 1 || MODE == 2 || MODE == 3);   // write step records
    LSP_T0 LSP_COUNT(25)
    // ---------------- phase A ------------------------------------------------------------------------------
    if constexpr (C::QUAD) {
        // quadrotor: work item = (knot k, player i, seed direction c of the player's 12 states + 4 rotor commands): column c of
        // [A_i | B_i] = d RK2 / d (x_i, u_i)[c] by forward-mode differentiation along e_c; the item with c = 0 also leaves the
        // RK2 value (the dyn rows of phase B read it)
        const double qmass = phase_f64(pr.qmass);
        for (int e = lane; e < (N - 1) * P * 16; e += C::NT) {
            const int c = e & 15, i = (e >> 4) % P, k = (e >> 4) / P;
            const double* sk = zstate<C>(z, k);
            Jet xj[12], uj[4], xo[12];
#pragma unroll
            for (int j = 0; j < 12; j++) xj[j] = Jet{sk[i + j * P], c == j ? 1.0 : 0.0};
#pragma unroll
            for (int j = 0; j < 4; j++) uj[j] = Jet{gld(z, (int)(n + hu<C>(k, i) + j)), c == 12 + j ? 1.0 : 0.0};
            quad_rk2(xj, uj, qmass, dt, xo);
            double* __restrict__ rc = G.rec(pr) + (size_t)k * R::LEN + R::COEF + i * C::QS;
            const int o = c < 12 ? C::QA + c : C::QB + (c - 12), ld = c < 12 ? 12 : 4;
#pragma unroll
            for (int j = 0; j < 12; j++) rc[o + j * ld] = xo[j].d;
            if (c == 0) {
#pragma unroll
                for (int j = 0; j < 12; j++) rc[C::QX + j] = xo[j].v;
            }
        }
    }
    assemble_phase_a<C, MODE, IBR>(pr, G, L, z, nullptr, 0.0, N, lane, dt, ip, acc);
    LSP(20)
    // ---------------- phase B ------------------------------------------------------------------------------
    // Every residual row of every step is independent once phase A has left the coefficients and the pair-gradient table:
    // three flat row loops (opt_x | opt_u | dyn), work item = one row, operands read straight from the trajectory (the
    // neighbouring lanes read neighbouring addresses; everything is L1/L2 resident after the first touch).
    // Index arithmetic is incremental and all offsets are 32-bit unsigned so that the loads use the scalar-base + vector-
    // offset addressing mode.  Each lane handles ASM_UNROLL rows per pass (rows e, e + 64, ...): their loads are in flight
    // together, which halves the exposed L2 latency.
    typedef unsigned uidx;
    struct Row { double r, dprox; bool mine, ok; uidx rec_off; int vrow; };
    auto finish_row = [&](const Row& q, bool dynrow) {
        if (!q.ok) return;
        double r = q.r;
        // regularize_residual! (global_quantities.jl:67-86): proximal term on the opt rows
        const double rr = zref ? r + reg * q.dprox : r;
        if (MODE == 3) { l1r += fabs(rr); }            // statistics / records of the unregularised rows
        else r = rr;
        bad |= !isfinite(r);
        if (IBR) {
            l1f += fabs(r);
            if (dynrow) { l1 += fabs(r); if (q.mine) vdyn = fmax(vdyn, fabs(r)); }
            else if (q.mine) { l1 += fabs(r); vopt = fmax(vopt, fabs(r)); }
        } else {
            l1 += fabs(r);
            if (dynrow) vdyn = fmax(vdyn, fabs(r)); else vopt = fmax(vopt, fabs(r));
        }
        if (RECS) gst(G.rec(pr), (int)(q.rec_off), r);
        if (MODE == 2) gst(G.res(pr), (int)(q.vrow), r);
    };
    const double* __restrict__ recg = G.rec(pr);
    // advance (k, j) by 64 rows of a row space with LEN rows per step
    constexpr int UR = C::ASM_UNROLL;
    auto run_rows = [&](auto&& row, int LEN, bool dynrow) {
        const int total = (N - 1) * LEN, stepk = (UR * C::NT) / LEN, stepj = (UR * C::NT) % LEN;
        int k[UR], j[UR];
#pragma unroll
        for (int t = 0; t < UR; t++) { const int e0 = lane + t * C::NT; k[t] = e0 / LEN; j[t] = e0 % LEN; }
        for (int e = lane; e < total; e += UR * C::NT) {
            Row q[UR];
#pragma unroll
            for (int t = 0; t < UR; t++) q[t] = row(k[t], j[t], e + t * C::NT < total);
#pragma unroll
            for (int t = 0; t < UR; t++) finish_row(q[t], dynrow);
#pragma unroll
            for (int t = 0; t < UR; t++) { j[t] += stepj; k[t] += stepk; if (j[t] >= LEN) { j[t] -= LEN; k[t] += 1; } }
        }
    };
    // ---- rows opt_i,x_{k+1}[a] = cost grad + pair terms + A_{k+1}' lambda_{i,k+1} - lambda_{i,k} (+ reg (x - xref))
    {
        constexpr int RXN = P * n;
        auto row_x = [&](int k, int ei, bool ok) -> Row {
            Row q; q.ok = ok;
            if (!ok) { k = 0; ei = 0; }
            const int i = ei / n, a = ei % n;
            const uidx zo = (uidx)(n + k * b);                              // block k: x_{k+1} | u_k | lambda_k
            const uidx ro = (uidx)(k * R::LEN);
            const bool has_next = (k + 1 <= N - 2);
            const double w = (k + 1 < N - 1) ? dt : 1.0;
            double r = -gld(z, (int)(zo + (uidx)(n + m + ei)));
            {
                // A_{k+1}' lambda_{i,k+1}: addresses clamped to block k when there is no next block, the term is dropped below
                const uidx lo = zo + (uidx)((has_next ? b : 0) + n + m + i * n), co = ro + (uidx)((has_next ? R::LEN : 0) + R::COEF);
                const double t = AT_vec<C>(recg + co, dt, [&](int rr) { return gld(z, (int)(lo + (uidx)rr)); }, a);
                r += has_next ? t : 0.0;
            }
            const bool own = (a % P == i);
            const double tqv = G.Qd(pr)[i * ni + a / P], txv = G.xf(pr)[i * ni + a / P];
            const double tq = own ? tqv : 0.0, tx = own ? txv : 0.0;
            const double xa = gld(z, (int)(zo + (uidx)a));
            r += w * (tq * (xa - tx));
            if (C::POS) { const double gv = recg[(uidx)R::gvt(N, k) + (uidx)((i * P + a % P) * C::PD + (a < C::PD * P ? a / P : 0))]; r += (a < C::PD * P) ? gv : 0.0; }
            if constexpr (C::EXT) {
                // StateBoundConstraint of player i (state_bound_constraint.jl:85-97): rows (x - x_max)[a], (x_min - x)[a]
                double qsb = 0.0;
                if (pr.has_sb && ok) {
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int ci = ext_sb_row(pr, i, k, half * n + a);
                        const double cv = half == 0 ? xa - ext_sbmax(pr, pr.extc)[ei] : ext_sbmin(pr, pr.extc)[ei] - xa;
                        if (MODE == 2) G.vals(pr)[ci] = cv;
                        if (isfinite(cv)) {
                            const double lm = gld(G.lam(pr), ci), am = al_active_mu(cv, lm, gld(G.mu(pr), ci));
                            const double wl = lm + am * cv;
                            r += (half == 0 ? wl : -wl); qsb += am;
                            if (!IBR || i == ip) vsta = fmax(vsta, fmax(0.0, cv));
                        }
                    }
                }
                if (RECS && ok) gst(G.rec(pr), (int)(ro + (uidx)(R::RQ + ei)), qsb);
            }
            q.mine = IBR ? (i == ip) : true;
            q.dprox = 0.0;
            if (zref) { const double xr = gld(zref, (int)(zo + (uidx)a)); q.dprox = q.mine ? xa - xr : 0.0; }
            q.r = r; q.rec_off = ro + (uidx)(R::RX + ei); q.vrow = MODE == 2 ? vx<C>(N, i, k) + a : 0;
            return q;
        };
        run_rows(row_x, RXN, false);
    }
    LSP(21)
    // ---- rows opt_i,u_{i,k}[c] = dt R (u - uf) + control-bound AL gradient + (B_k' lambda_{i,k})[c] (+ reg (u - uref))
    {
        auto row_u = [&](int k, int c, bool ok) -> Row {
            Row q; q.ok = ok;
            if (!ok) { k = 0; c = 0; }
            const int i = c % P;
            const uidx zo = (uidx)(n + k * b), ro = (uidx)(k * R::LEN);
            const double u = gld(z, (int)(zo + (uidx)(n + uoff<C>(c))));
            const uidx lo = zo + (uidx)(n + m + i * n);
            const double tr = G.Rd(pr)[(c % P) * mi + c / P], tu = G.uf(pr)[(c % P) * mi + c / P];
            double g = 0.0, rhat = dt * tr + jreg;
            if (pr.has_ctl && ok) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int ci = con_ctl<C>(pr, k, half * m + c);
                    const double cv = half == 0 ? u - pr.umax[c] : pr.umin[c] - u;
                    if (MODE == 2) G.vals(pr)[ci] = cv;
                    if (isfinite(cv)) {
                        const double lm = gld(G.lam(pr), ci), am = al_active_mu(cv, lm, gld(G.mu(pr), ci));
                        const double wl = lm + am * cv;
                        g += (half == 0 ? wl : -wl); rhat += am;
                        if (!IBR) vcon = fmax(vcon, fmax(0.0, cv));
                        else if ((pr.ibr_ctl_rows[ip] >> (half * m + c)) & 1ull) vcon = fmax(vcon, fmax(0.0, cv));
                    }
                }
            }
            q.r = dt * (tr * (u - tu)) + g + BT_vec<C>(recg + ro + (uidx)R::COEF, dt, [&](int rr) { return gld(z, (int)(lo + (uidx)rr)); }, c);
            q.mine = IBR ? (i == ip) : true;
            q.dprox = 0.0;
            if (zref) { const double ur = gld(zref, (int)(zo + (uidx)(n + uoff<C>(c)))); q.dprox = q.mine ? u - ur : 0.0; }
            if (RECS && ok) gst(G.rec(pr), (int)(ro + (uidx)(R::RHAT + c)), rhat);
            q.rec_off = ro + (uidx)(R::RU + c); q.vrow = MODE == 2 ? vu<C>(N, i, k) + c / P : 0;
            return q;
        };
        run_rows(row_u, m, false);
    }
    LSP(22)
    // ---- rows dyn_k[a] = RK2(x_k, u_k)[a] - x_{k+1}[a]   (explicit midpoint, RobotDynamics 0.3.1)
    {
        auto row_d = [&](int k, int a, bool ok) -> Row {
            Row q; q.ok = ok;
            if (!ok) { k = 0; a = 0; }
            const uidx zo = (uidx)(n + k * b), ro = (uidx)(k * R::LEN);
            const uidx po = (k == 0) ? 0u : zo - (uidx)b;                   // x_k: x_1 sits in front of block 0
            const double* Ck = recg + ro + (uidx)R::COEF;
            double xn;
            if constexpr (C::QUAD) {
                xn = Ck[(a % P) * C::QS + C::QX + a / P];                    // RK2 value left by phase A
            } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                // position rows: x + (v + dt/2 u) dt ; velocity rows: v + u dt
                const int j = a < m ? a : a - m;
                const double uj = gld(z, (int)(zo + (uidx)(n + uoff<C>(j)))), base = gld(z, (int)(po + (uidx)a)), vel = gld(z, (int)(po + (uidx)(j + m)));
                const double vm = vel + (uj * dt) * 0.5;
                xn = base + (a < m ? vm : uj) * dt;
            } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
                const int blkk = a / P, i = a % P;
                const double ua = gld(z, (int)(zo + (uidx)(n + uoff<C>(i)))), base = gld(z, (int)(po + (uidx)a)), vel = gld(z, (int)(po + (uidx)(2 * P + i)));
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 5 : (blkk == 1 ? 6 : 4)) * P + i];    // dt cos th / dt sin th / dt sin(beta)/lr
                xn = (blkk == 2) ? base + ua * dt : base + vm * cf;
            } else {
                const int blkk = a / P, i = a % P;
                const double ua = gld(z, (int)(zo + (uidx)(n + uoff<C>(P + i)))), base = gld(z, (int)(po + (uidx)a)), vel = gld(z, (int)(po + (uidx)(3 * P + i)));
                const double uo = gld(z, (int)(zo + (uidx)(n + uoff<C>((blkk >= 2 ? blkk - 2 : 0) * P + i))));
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 1 : 3) * P + i];                      // dt cos(thm) / dt sin(thm)
                xn = (blkk <= 1) ? base + vm * cf : base + uo * dt;
            }
            q.r = xn - gld(z, (int)(zo + (uidx)a));
            q.mine = IBR ? (a % P == ip) : true;                           // dynamics_violation(model, pdtraj, i): entries pz[i]
            q.dprox = 0.0; q.rec_off = ro + (uidx)(R::RD + a); q.vrow = MODE == 2 ? vd<C>(N, k) + a : 0;
            return q;
        };
        run_rows(row_d, n, true);
    }
    LSP(23)
    out.l1 = wave_sum(l1); out.opt = wave_max(vopt); out.dyn = wave_max(vdyn);
    out.con = wave_max(vcon); out.sta = wave_max(vsta); out.nonfinite = wave_or(bad);
    out.l1reg = (MODE == 3) ? wave_sum(l1r) : out.l1;
    out.l1full = IBR ? wave_sum(l1f) : out.l1;
    team_combine<C>(out);
    LSP(24)
}

// ================================================================================================
// Fused trial pass (round 4; double integrator, one wavefront per game: the C2 / C4 kernel).  One pass does what update_traj! +
// assemble_pass did in two: the trial iterate z + alpha dz is formed where the source iterate is read, written out once, and every
// residual row is evaluated out of LDS, FT time steps at a time -- the trajectory, the direction and the proximal reference (= the
// source iterate) cross the memory system once instead of the four to five times of the three flat row loops (per-pass counters:
// tests/probes/phase_bytes.sh; r03: 112 KB read per trial against 17 + 17 KB of trajectory and direction).
//   AXPY = false: the rows of the source iterate itself (record!: MODE 1, no proximal term, nothing written but the records)
//   MODE 0 / 3 as in assemble_pass (3 = statistics and records of the unregularised rows + the regularised norm l1reg)
// Same row arithmetic as assemble_pass (same expressions in the same order); the norms are summed in another order.
// ================================================================================================
//   DUAL (record! only): the pass also performs the dual_update! + penalty_update! that precede it in newton_solve! (see assemble_phase_a)
template <class C, int MODE, bool AXPY, bool DUAL = false>
__device__ void assemble_fused(CPR pr0, const Game& G0, AsmLds<C>& L, double alpha, bool prox, double reg, double jreg, ResOut& out, bool dual = false) {
    static_assert(AsmLds<C>::FUSED && (MODE == 0 || MODE == 1 || MODE == 3), "fused trial pass: double integrator / unicycle, statistics / record modes");
    static_assert(!DUAL || (MODE == 1 && !AXPY), "the dual update rides on the record! pass");
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    constexpr int n = C::n, m = C::m, P = C::P, mi = C::mi, ni = C::ni, b = C::b, FT = AsmLds<C>::FT, TAB = AsmLds<C>::TAB, NXU = n + m, NT = C::NT, NC = C::NC;
    // the chunk buffers are shared by the whole team; inside the chunk loop only LDS is exchanged (phase A ends with a full barrier), so the
    // team's barrier orders LDS only -- with barriers that drained vmcnt the fused pass lost to the two passes on teams (C5 loop: 105 vs 152 K/s)
    auto fsync = [&]() { if constexpr (C::NW == 1) sweep_sync<C>(); else team_lds_barrier(); };
    using R = Rec<C>;
    constexpr bool RECS = (MODE == 1 || MODE == 3);
    const int N = phase_int(pr.N), lane = phase_lane();
    const double dt = phase_f64(pr.dt);
    const double* __restrict__ zs = G.z(0);                    // source iterate = proximal reference
    const double* __restrict__ dz = G.z(2);
    double* __restrict__ zo = G.z(1);                          // the trial iterate goes here
    AsmAcc acc;
    LSP_T0 LSP_COUNT(25)
    // ---- phase A over all steps (positions of the trial iterate formed on the fly), heads and tables to the records as in assemble_pass
    // Round 6 (PCH): phase A runs per chunk, out of the staged blocks -- the positions of iterate and direction are not read a second time from
    // global memory, the pair-gradient tables never leave the chunk's LDS, the Jacobian coefficients of the unicycle go straight to the chunk
    // (and to the records for the sweeps), the record heads [Hh | Hd] are stored by the items themselves.  One item per lane: (step, player).
    constexpr bool PCH = ALG_R6_PHASEA_CHUNK != 0 && (C::POS || NC > 0);
    static_assert(!PCH || (FT + 1) * P <= NT, "one phase-A item per lane and chunk");
    if constexpr (!PCH) assemble_phase_a<C, MODE, false, DUAL>(pr, G, L, zs, AXPY ? dz : nullptr, alpha, N, lane, dt, -1, acc, dual);
    LSP(20)
    const bool pairs_on = P > 1 && (pr.has_colcost || pr.has_colavoid);
    auto& Ch = L.ch;
    for (int e = lane; e < AsmLds<C>::NLQR; e += NT) Ch.lqr[e] = G.Qd(pr)[e];          // [Qd | xf | Rd | uf]
    const double* lQd = Ch.lqr; const double* lxf = lQd + P * ni; const double* lRd = lQd + 2 * P * ni; const double* luf = lRd + P * mi;
    double* __restrict__ recg = G.rec(pr);
    auto finish = [&](double r, double dprox, bool dynrow, unsigned rec_off) {
        const double rr = prox ? r + reg * dprox : r;                // regularize_residual! (global_quantities.jl:67-86)
        if (MODE == 3) acc.l1r += fabs(rr); else r = rr;
        acc.bad |= !isfinite(r);
        acc.l1 += fabs(r);
        if (dynrow) acc.vdyn = fmax(acc.vdyn, fabs(r)); else acc.vopt = fmax(acc.vopt, fabs(r));
        if (RECS) gst(recg, (int)rec_off, r);
    };
    for (int k0 = 0; k0 < N - 1; k0 += FT) {
        const int nst = (N - 1 - k0) < FT ? (N - 1 - k0) : FT;            // steps of this chunk
        const int nblk = (k0 + nst < N - 1) ? nst + 1 : nst;             // blocks staged: the chunk's and the next one (A' lambda_{k+1})
        // (PCH) multipliers / penalties of this lane's phase-A item, requested ahead of the chunk's blocks
        constexpr int NPR = P > 1 ? P - 1 : 1;
        const int aks = lane / P, ai = lane % P;                          // phase-A item of this lane: (step k0 + aks, player ai)
        double plm[NPR], pmu[NPR];
        if constexpr (PCH && C::POS) {
#pragma unroll
            for (int jj = 0; jj < NPR; jj++) { plm[jj] = 0.0; pmu[jj] = 0.0; }
            if (pairs_on && pr.has_colavoid && aks < nst) {
#pragma unroll
                for (int jj = 0; jj < P - 1; jj++) {
                    const int j = jj < ai ? jj : jj + 1, ci = con_col<C>(N, pairq<C>(ai, j), k0 + aks + 1);
                    plm[jj] = gld(G.lam(pr), ci); pmu[jj] = gld(G.mu(pr), ci);
                }
            }
        }
        // ---- stage: x_k of the first step, then blocks k0 .. k0 + nblk - 1; the trial blocks of the chunk's own steps go out here
        if (lane < n) {
            const int src = k0 == 0 ? lane : n + (k0 - 1) * b + lane;
            double v = zs[src];
            if (AXPY && k0 > 0) v = __builtin_fma(alpha, dz[src], v);      // (x_1 does not move)
            Ch.xprev[lane] = v;
        }
        {
            const int base = n + k0 * b, cnt = nblk * b, own = nst * b;
#if ALG_R6_STAGE
            // Round 6: EVERY global load of the chunk -- the blocks of z and dz, the pair-gradient tables, the Jacobian coefficients -- is issued
            // before the first one is waited for.  (Round 4's loop fetched four elements per lane and trip and the tables one per trip: a chunk of
            // C2 exposed seven global round trips in a row, a pass twenty-one, with four games per SIMD streaming at the same time.)
            // (batches of SB elements per lane: all of them at once -- 12 x 2 doubles per lane at C2 -- overflows the 128-register budget)
            constexpr int SU = ((FT + 1) * b + NT - 1) / NT;              // elements per lane of the largest chunk (C2: 12)
            // (the record pass loads z alone -- half the registers per element: ALG_R6_STAGE_BATCH_REC elements per batch)
            constexpr int SB4 = AXPY ? ALG_R6_STAGE_BATCH : (ALG_R6_STAGE_BATCH_REC < SU ? ALG_R6_STAGE_BATCH_REC : SU);
            constexpr int SB = C::WPE == 4 ? SB4 : (SU < ALG_R6_STAGE_BATCH_W2 ? SU : ALG_R6_STAGE_BATCH_W2);
            constexpr int TU = C::POS ? (FT * TAB + NT - 1) / NT : 1, CU = NC > 0 ? ((FT + 1) * NC + NT - 1) / NT : 1;
            double gt[TU], cf[CU];
            const int tcnt = nst * TAB, ccnt = nblk * NC;
            if constexpr (C::POS && !PCH) {
#pragma unroll
                for (int t = 0; t < TU; t++) { const int e = lane + t * NT; gt[t] = gld(recg + R::gvt(N, k0), e < tcnt ? e : 0); }            // contiguous behind the records
            }
            if constexpr (NC > 0 && !PCH) {                                // Jacobian coefficients of the staged steps (phase A left them in the records)
#pragma unroll
                for (int t = 0; t < CU; t++) { const int e = lane + t * NT, ec = e < ccnt ? e : 0; cf[t] = gld(recg, (k0 + ec / NC) * R::LEN + R::COEF + ec % NC); }
            }
            // Lanes past the end of the chunk repeat its last element (clamped index) instead of being masked off: they load, form and store the
            // very value the owning lane does -- same address, same bits -- so no per-element exec mask has to be kept in scalar registers.  The
            // trial values of the chunk's extra block (A_{k+1}' lambda_{k+1}) go out here as well as with the next chunk: the same bits twice.
            // Round 6 (lane roles, ALG_R6_LANEROLE >= 2): a lane stages the same entry of every block (NT / b blocks per trip, the trips unrolled): the
            // element of trip T is (T BS b + lane) -- a chunk-invariant lane offset plus an immediate in every address, no clamps, no (block, entry)
            // divided out of a flat index for the [x | u] copy; b of NT lanes work (C2: 54 of 64, 14 trips for 12).
            // Measured and NOT taken (profiles/r06_ab_lr2_c2.txt): bit-identical, but C2 14.08 against 14.55 M/s with the flat staging -- fewer loads in
            // flight per batch (four blocks where the flat loop holds six elements of z and dz; five already spill) and ten idle lanes cost more than
            // the index arithmetic saved.  The default build keeps ALG_R6_LANEROLE 1 (rows only).
            constexpr bool SLR = ALG_R6_LANEROLE >= 2 && C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR && !ALG_LSM_DI1W && b <= NT;
            if constexpr (SLR) {
                constexpr int BS = NT / b, BT = (FT + 1 + BS - 1) / BS, SBL = BT < ALG_R6_STAGE_BATCH_LR ? BT : ALG_R6_STAGE_BATCH_LR;
                int sl = lane; asm volatile("" : "+v"(sl)); __builtin_assume(sl >= 0 && sl < NT);
                const int lb = sl / b, o = sl % b;
                if (lb < BS) {
                    const bool isxu = o < NXU;
                    double* const lxu = Ch.zxu + lb * NXU + o;
#pragma unroll
                    for (int t0 = 0; t0 < BT; t0 += SBL) {
                        double a[SBL], d[SBL];
#pragma unroll
                        for (int t = 0; t < SBL; t++) {
                            const int T = t0 + t, j = BS == 1 ? T : T * BS + lb;
                            if (T < BT && j < nblk) { a[t] = gld(zs + base + T * BS * b, sl); d[t] = AXPY ? gld(dz + base + T * BS * b, sl) : 0.0; }
                        }
#pragma unroll
                        for (int t = 0; t < SBL; t++) {
                            const int T = t0 + t, j = BS == 1 ? T : T * BS + lb;
                            if (T < BT && j < nblk) {
                                const double v = AXPY ? __builtin_fma(alpha, d[t], a[t]) : a[t];
                                Ch.zt[T * BS * b + sl] = v;
                                if (AXPY) gst(zo + base + T * BS * b, sl, v);
                                if (isxu && j < nst) lxu[T * BS * NXU] = a[t];
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int t0 = 0; t0 < SU; t0 += SB) {
                if (t0 * NT >= cnt) break;                                  // (wave-uniform: the last chunk of a horizon is shorter)
                double a[SB], d[SB]; int ecv[SB];
#pragma unroll
                for (int t = 0; t < SB; t++) {
                    const int e = lane + (t0 + t) * NT; ecv[t] = e < cnt ? e : cnt - 1;
                    a[t] = gld(zs + base, ecv[t]); d[t] = AXPY ? gld(dz + base, ecv[t]) : 0.0;
                }
#pragma unroll
                for (int t = 0; t < SB; t++) {
                    if (t0 + t >= SU) break;
                    const int ec = ecv[t];
                    const double v = AXPY ? __builtin_fma(alpha, d[t], a[t]) : a[t];
                    Ch.zt[ec] = v;
                    if (AXPY) gst(zo + base, ec, v);
                    const int j = ec / b, o = ec % b;
                    double* const xu = (o < NXU && j < nst) ? &Ch.zxu[j * NXU + o] : &Ch.dump[0];
                    *xu = a[t];
                }
            }
            if constexpr (C::POS && !PCH) {
#pragma unroll
                for (int t = 0; t < TU; t++) { const int e = lane + t * NT; if (e < tcnt) Ch.gvt[e] = gt[t]; }
            }
            if constexpr (NC > 0 && !PCH) {
#pragma unroll
                for (int t = 0; t < CU; t++) { const int e = lane + t * NT; if (e < ccnt) Ch.coef[e] = cf[t]; }
            }
#else
            for (int e0 = lane; e0 < cnt; e0 += 4 * NT) {
                double a[4], d[4];
#pragma unroll
                for (int t = 0; t < 4; t++) { const int e = e0 + t * NT, ec = e < cnt ? e : e0; a[t] = gld(zs + base, ec); d[t] = AXPY ? gld(dz + base, ec) : 0.0; }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int e = e0 + t * NT;
                    if (e < cnt) {
                        const double v = AXPY ? __builtin_fma(alpha, d[t], a[t]) : a[t];
                        Ch.zt[e] = v;
                        if (AXPY && e < own) gst(zo + base, e, v);
                        const int j = e / b, o = e % b;
                        if (o < NXU && j < nst) Ch.zxu[j * NXU + o] = a[t];
                    }
                }
            }
            if constexpr (C::POS) {
                const int tcnt = nst * TAB;
                for (int e = lane; e < tcnt; e += NT) Ch.gvt[e] = gld(recg + R::gvt(N, k0), e);            // contiguous behind the records
            }
            if constexpr (NC > 0) {                                        // Jacobian coefficients of the staged steps (phase A left them in the records)
                const int ccnt = nblk * NC;
                for (int e = lane; e < ccnt; e += NT) Ch.coef[e] = recg[(size_t)(k0 + e / NC) * R::LEN + R::COEF + e % NC];
            }
#endif
        }
        fsync();
        if constexpr (PCH) {
            const int k = k0 + aks;
            if constexpr (C::MODEL == ALG_MODEL_UNICYCLE) {
                // Jacobian coefficients of step k (A_k, B_k) from the staged (trial) values: x_k is the block before, u_k the step's own block
                if (aks < nblk) {
                    const double* xk = aks == 0 ? Ch.xprev : Ch.zt + (aks - 1) * b; const double* uk = Ch.zt + aks * b + n + ai * mi;
                    const double th = xk[2 * P + ai], v = xk[3 * P + ai], om = uk[0], ac = uk[1];
                    const double thm = th + (om * dt) * 0.5, vm = v + (ac * dt) * 0.5;
                    double sn, cs; sincos(thm, &sn, &cs);
                    const double c0 = -dt * vm * sn, c1 = dt * cs, c2 = dt * vm * cs, c3 = dt * sn;
                    double* cl = Ch.coef + aks * NC;
                    cl[0 * P + ai] = c0; cl[1 * P + ai] = c1; cl[2 * P + ai] = c2; cl[3 * P + ai] = c3;
                    if (RECS && aks < nst) {                                   // (the extra block's step belongs to the next chunk)
                        double* rc = recg + (size_t)k * R::LEN + R::COEF;
                        rc[0 * P + ai] = c0; rc[1 * P + ai] = c1; rc[2 * P + ai] = c2; rc[3 * P + ai] = c3;
                    }
                }
            }
            if constexpr (C::POS) {
                if (aks < nst) {
                    const double* x1 = Ch.zt + aks * b;
                    auto xp = [&](int idx) { return x1[idx]; };
                    auto lmu = [&](int jj, int, double& lm, double& mu_c) {
                        lm = plm[0]; mu_c = pmu[0];
#pragma unroll
                        for (int q = 1; q < NPR; q++) { lm = (jj == q) ? plm[q] : lm; mu_c = (jj == q) ? pmu[q] : mu_c; }
                    };
                    phase_a_pos_item<C, MODE, false, DUAL>(pr, G, N, k, ai, -1, dt, pairs_on, xp, lmu, recg + (size_t)k * R::LEN, Ch.gvt + aks * TAB, acc, dual);
                }
            }
            fsync();
        }
        // ---- rows opt_i,x_{k+1}[a]
        // Round 6 (lane roles, double integrator): the rows of a kind are dealt as (step, row of the step) = (trip, lane) instead of flat over the
        // lanes.  A lane then owns the same row of every step it visits -- player, entry, LQR constants, table slot, control bounds, every LDS
        // offset inside the block are invariants of the chunk, and a trip is its loads, the row's arithmetic and the store.  The flat dealing
        // divided (step, row) out of the flat index on every trip: 65 of the 75 VALU instructions of an opt_x trip at C2 were index arithmetic
        // and selects on it.  opt_x rows: NT / (P n) steps per trip (C2: one, 36 of 64 lanes, 13 trips for 8 -- but a fifth of the instructions
        // each); opt_u rows NT / m steps per trip, dyn rows NT / n (C2: ten and five: the same number of trips as before).  Every row is the
        // same expression as before; a lane sums other rows than it did, so the l1 norms move in the last bits.  The one-wavefront unicycle kernels keep the flat dealing
        // (lane_roles_v: with ALG_R6_LANEROLE_UNI the line search's group pass deals its rows by lane roles too and stays bit-identical -- measured, not taken).
        constexpr int RPS = P * n;
        constexpr bool LR = lane_roles_v<C>;
        // (the roles are worked out per chunk from an opaque copy of the lane id: as invariants of the whole pass they would be hoisted in front
        // of the chunk loop and held in registers through staging and phase A -- spills in a 128-register kernel)
        int rlane = lane;
        if constexpr (LR) { asm volatile("" : "+v"(rlane)); __builtin_assume(rlane >= 0 && rlane < NT); }
        auto xrow = [&](int ks, int ei, int i, int a, double tq, double tx) {
            const int k = k0 + ks;
            const double* blk = Ch.zt + ks * b;
            const bool has_next = (k + 1 <= N - 2);
            const double w = (k + 1 < N - 1) ? dt : 1.0;
            double r = -blk[n + m + ei];
            {
                const double* ln = blk + (has_next ? b : 0) + n + m + (ei - a);                 // (= i n: lambda_i[a] is the row's own entry one block on)
                const double t = AT_vec<C>(Ch.coef + (ks + (has_next ? 1 : 0)) * NC, dt, [&](int rr) { return ln[rr]; }, a);
                r += has_next ? t : 0.0;
            }
            const double xa = blk[a];
            r += w * (tq * (xa - tx));
            if (C::POS) { const double gv = Ch.gvt[ks * TAB + (i * P + a % P) * C::PD + (a < C::PD * P ? a / P : 0)]; r += (a < C::PD * P) ? gv : 0.0; }
            const double dprox = prox ? xa - Ch.zxu[ks * NXU + a] : 0.0;
            finish(r, dprox, false, (unsigned)(k * R::LEN + R::RX + ei));
        };
        auto xconst = [&](int i, int a, double& tq, double& tx) {
            const bool own = (a % P == i);
            const double tqv = lQd[i * ni + a / P], txv = lxf[i * ni + a / P];
            tq = own ? tqv : 0.0; tx = own ? txv : 0.0;
        };
        if constexpr (LR) {
            // (the trips are unrolled: the step of a trip is a compile-time constant -- plus the lane's share where several steps fit a trip --
            // so that every LDS address of a row is one chunk-invariant lane offset and an immediate.  Each trip stays behind its own `ks < nst`:
            // as straight-line code for a full chunk the thirteen trips' loads were scheduled together -- 88 spilled registers at C2)
            constexpr int XS = NT / RPS, XT = (FT + XS - 1) / XS;
            const int ls = rlane / RPS, ei = rlane % RPS, i = ei / n, a = ei % n;
            if (ls < XS) {
                double tq, tx; xconst(i, a, tq, tx);
#pragma unroll
                for (int t = 0; t < XT; t++) { const int ks = XS == 1 ? t : t * XS + ls; if (ks < nst) xrow(ks, ei, i, a, tq, tx); }
            }
        } else {
#if ALG_R6_ROWIDX
        // (row e = lane + NT t of the chunk, as (step, entry) = divmod(e, P n), carried from trip to trip instead of divided out again)
        constexpr int RDQ = NT / RPS, RDR = NT % RPS;
        int rks = lane / RPS, rei = lane % RPS;
#endif
        for (int e = lane; e < nst * P * n; e += NT) {
#if ALG_R6_ROWIDX
            __builtin_assume(rei >= 0 && rei < RPS);
            const int ks = rks, ei = rei, i = ei / n, a = ei % n;
            rks += RDQ; rei += RDR;
            if (rei >= RPS) { rei -= RPS; rks += 1; }
#else
            const int ks = e / (P * n), ei = e % (P * n), i = ei / n, a = ei % n;
#endif
            double tq, tx; xconst(i, a, tq, tx);
            xrow(ks, ei, i, a, tq, tx);
        }
        }
        LSP(21)
        // ---- rows opt_i,u_{i,k}[c]
        auto urow = [&](int ks, int c, double tr, double tu) {
            const int i = c % P, k = k0 + ks;
            const double* blk = Ch.zt + ks * b;
            const double u = blk[n + uoff<C>(c)];
            const double* lo = blk + n + m + i * n;
            double g = 0.0, rhat = dt * tr + jreg;
            if (pr.has_ctl) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int ci = con_ctl<C>(pr, k, half * m + c);
                    const double cv = half == 0 ? u - pr.umax[c] : pr.umin[c] - u;
                    if (DUAL && dual) {
                        // evaluate! + dual_update! (alpha_dual) + penalty_update! of the two control-bound rows of this control (dual_penalty_update's expressions;
                        // penalty_update! scales mu of every row, finite or not)
                        const auto& od = pr.opt;
                        double lm = gld(G.lam(pr), ci), mu_c = gld(G.mu(pr), ci);
                        gst(G.vals(pr), ci, cv);
                        if (isfinite(cv)) { lm = dual_ascent(lm, od.alpha_dual, mu_c, cv, od.lambda_max); gst(G.lam(pr), ci, lm); }
                        mu_c = fmin(fmax(mu_c * od.rho_increase, 0.0), od.rho_max);
                        gst(G.mu(pr), ci, mu_c);
                        if (isfinite(cv)) {
                            const double am = al_active_mu(cv, lm, mu_c);
                            const double wl = lm + am * cv;
                            g += (half == 0 ? wl : -wl); rhat += am;
                            acc.vcon = fmax(acc.vcon, fmax(0.0, cv));
                        }
                    } else
                    if (isfinite(cv)) {
                        const double lm = gld(G.lam(pr), ci), am = al_active_mu(cv, lm, gld(G.mu(pr), ci));
                        const double wl = lm + am * cv;
                        g += (half == 0 ? wl : -wl); rhat += am;
                        acc.vcon = fmax(acc.vcon, fmax(0.0, cv));
                    }
                }
            }
            const double r = dt * (tr * (u - tu)) + g + BT_vec<C>(Ch.coef + ks * NC, dt, [&](int rr) { return lo[rr]; }, c);
            const double dprox = prox ? u - Ch.zxu[ks * NXU + n + uoff<C>(c)] : 0.0;
            if (RECS) gst(recg, k * R::LEN + R::RHAT + c, rhat);
            finish(r, dprox, false, (unsigned)(k * R::LEN + R::RU + c));
        };
        if constexpr (LR) {
            constexpr int US = NT / m, UT = (FT + US - 1) / US;
            const int ls = rlane / m, c = rlane % m;
            if (ls < US) {
                const double tr = lRd[(c % P) * mi + c / P], tu = luf[(c % P) * mi + c / P];
#pragma unroll
                for (int t = 0; t < UT; t++) { const int ks = t * US + ls; if (ks < nst) urow(ks, c, tr, tu); }
            }
        } else {
            for (int e = lane; e < nst * m; e += NT) {
                const int ks = e / m, c = e % m;
                urow(ks, c, lRd[(c % P) * mi + c / P], luf[(c % P) * mi + c / P]);
            }
        }
        LSP(22)
        // ---- rows dyn_k[a] = RK2(x_k, u_k)[a] - x_{k+1}[a]   (explicit midpoint; the expressions of assemble_pass)
        auto drow = [&](int ks, int a) {
            const int k = k0 + ks;
            const double* blk = Ch.zt + ks * b;
            const double* xk = ks == 0 ? Ch.xprev : blk - b;
            double xn;
            if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                // position rows: x + (v + dt/2 u) dt ; velocity rows: v + u dt
                const int j = a < m ? a : a - m;
                const double uj = blk[n + uoff<C>(j)], base = xk[a], vel = xk[j + m];
                const double vm = vel + (uj * dt) * 0.5;
                xn = base + (a < m ? vm : uj) * dt;
            } else {
                const double* Ck = Ch.coef + ks * NC;
                const int blkk = a / P, i = a % P;
                const double ua = blk[n + uoff<C>(P + i)], base = xk[a], vel = xk[3 * P + i];
                const double uo = blk[n + uoff<C>((blkk >= 2 ? blkk - 2 : 0) * P + i)];
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 1 : 3) * P + i];                      // dt cos(thm) / dt sin(thm)
                xn = (blkk <= 1) ? base + vm * cf : base + uo * dt;
            }
            finish(xn - blk[a], 0.0, true, (unsigned)(k * R::LEN + R::RD + a));
        };
        if constexpr (LR) {
            constexpr int DS = NT / n, DT = (FT + DS - 1) / DS;
            const int ls = rlane / n, a = rlane % n;
            if (ls < DS) {
#pragma unroll
                for (int t = 0; t < DT; t++) { const int ks = t * DS + ls; if (ks < nst) drow(ks, a); }
            }
        } else {
            for (int e = lane; e < nst * n; e += NT) drow(e / n, e % n);
        }
        LSP(23)
        fsync();                                           // the next chunk overwrites the buffers
    }
    out.l1 = wave_sum(acc.l1); out.opt = wave_max(acc.vopt); out.dyn = wave_max(acc.vdyn);
    out.con = wave_max(acc.vcon); out.sta = wave_max(acc.vsta); out.nonfinite = wave_or(acc.bad);
    out.l1reg = (MODE == 3) ? wave_sum(acc.l1r) : out.l1;
    out.l1full = out.l1;
    team_combine<C>(out);
    LSP(24)
}

// ================================================================================================
// Line-search trials of several step sizes in ONE pass (team kernels; solver_methods.jl:105-125 evaluates them one after another).
// The trials alpha_j = alpha_decrease^(j-1) of a line search are independent of each other: each is the norm of the regularised residual at
// z + alpha_j dz.  On a team the trial pass is bound by latency (L2 round trips, workgroup barriers), not by arithmetic, and the receding-
// horizon loop spends half of its slowest game's time in it (13 trials per Newton iteration on the game that sets the duration of the C5
// loop, profiles/r05_mpc_prof_c5.txt).  This pass evaluates NA step sizes at once and leaves NOTHING behind but the NA norms: no trial
// iterate, no records.  The search then runs the ordinary pass once, for the accepted step size, which produces the trial iterate, the
// records and the cached statistics exactly as the sequential search's last trial does.  Every row is the expression of assemble_pass
// evaluated on z + alpha dz formed on the fly (the value update_traj! stores and assemble_pass reads back: one fma), rows are dealt to lanes
// and summed in the same order: the norms are bit-identical to the sequential trials', so the same step is accepted.  (line_search checks
// that on every accepted step: alg_game_stats::reserved counts disagreements and stays 0.)
// Phase A's per-step-size output -- Jacobian coefficients and pair-gradient tables -- goes to the gain scratch (dead outside the direction).
//   base constraint set of the double integrator / unicycle only (collision cost, collision avoidance, control bounds)
// ================================================================================================
#ifndef ALG_LSM_DI1W
#define ALG_LSM_DI1W 0        // 1: the group pass on the one-wavefront double-integrator kernels as well.  Measured in round 6 and not
                              // taken: bit-identical, 126 VGPRs, but neutral to -1 % on perturbed C2 / C4 batches with and without the hand-off -- their
                              // stragglers take many iterations, not deep searches (profiles/r06_ab_lsm_di1w_*.txt)
#endif
template <class C> struct LsMulti {
    // team kernels (whose ordinary trial pass is assemble_pass: all rows of a kind in one flat loop) and the one-wavefront unicycle kernels (fused
    // trial pass: the rows chunk by chunk -- the group pass deals and sums its rows in the same order, CHUNK steps at a time).  Norms bit-identical
    // in both cases.  (On the one-wavefront kernels they first came out an ulp apart in one of ten candidates: BT_vec's `a b + c d` had been
    // contracted one way in the fused pass and the other way here; BT_vec now states its fmas.)
    static constexpr bool TEAMS = C::NW > 1 && !AsmLds<C>::FUSED && (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE);
    static constexpr bool ONEW = C::NW == 1 && AsmLds<C>::FUSED && (C::MODEL == ALG_MODEL_UNICYCLE || ALG_LSM_DI1W);
    static constexpr bool ON = (TEAMS || ONEW) && !C::EXT && !C::DENSE && C::POS;
    static constexpr int CHUNK = ONEW ? AsmLds<C>::FT : (1 << 20);
    static constexpr int NA = LS_NA;                          // step sizes per pass
    static constexpr int TAB = C::PD * C::P * C::P, SW = C::NC + TAB;          // scratch doubles per step and step size: [coef | table]
    // source iterate and direction of the search staged in LDS ([z | dz], the kernel's Lds union: nothing else lives there during a search) when
    // they fit: the group passes then read their operands at LDS latency (C5: 2 x 1578 doubles; C3's 2 x 4328 stay in L2)
    static constexpr int CAP = LsLds<C>::CAP;
    static constexpr bool LDSZ = LsLds<C>::ON;
};
template <class C, int NA, bool LZ>
__device__ void trial_norms_multi(CPR pr0, const Game& G0, const double* lz, double* lsc, double alpha0, bool prox, double reg, double (&l1reg)[NA]) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    double al[NA];                                               // alpha0 alpha_decrease^q, formed like the one-by-one search forms them
    { double a = alpha0;
#pragma unroll
      for (int q = 0; q < NA; q++) { al[q] = a; a *= pr.opt.alpha_decrease; } }
    constexpr int n = C::n, m = C::m, P = C::P, mi = C::mi, ni = C::ni, b = C::b, PD = C::PD, NC = C::NC, TAB = LsMulti<C>::TAB, SW = LsMulti<C>::SW, NT = C::NT;
    static_assert(PD == 2, "planar models");
    const int N = phase_int(pr.N), lane = phase_lane();
    const double dt = phase_f64(pr.dt);
    const double* __restrict__ zs = G.z(0); const double* __restrict__ dz = G.z(2);
    const int TL = phase_int(pr.traj_len);
    // entry of the source iterate / of the direction.  The LDS index is opaque like gld's offset: if the compiler saw that the prox term's
    // reference x is the very load inside fma(alpha, dx, x), it would simplify fma(alpha, dx, x) - x (contraction is on) -- the one-by-one
    // search rounds the fma, stores it, and subtracts afterwards; with shared loads 5 of 64 perturbed C5 solves disagreed in the last bit
    auto ldz = [&](int idx) { if constexpr (LZ) { int o = idx; asm("" : "+v"(o)); return lz[o]; } else return gld(zs, idx); };
    auto ldd = [&](int idx) { if constexpr (LZ) { int o = TL + idx; asm("" : "+v"(o)); return lz[o]; } else return gld(dz, idx); };
    // Every trial value goes through an opaque register copy: the ordinary passes read the trial iterate back from memory (or LDS), i.e. they
    // compute on a ROUNDED fma(alpha, dz, z) the compiler knows nothing about.  Left visible, the fma node takes part in the contraction of the
    // expressions around it (fma(alpha, dx, x) - x of the proximal term was simplified outright).
    auto rounded = [](double v) { asm("" : "+v"(v)); return v; };
    // entry idx of the trial iterate of step size q (x_1, the first n entries, does not move: update_traj! never touches it)
    auto tv = [&](int q, int idx) { const double a = ldz(idx), v = __builtin_fma(al[q], ldd(idx), a); return rounded(idx < n ? a : v); };
    constexpr bool LSC = LZ && LsLds<C>::SC_ON;                  // the tables of the step sizes in LDS as well (teams of four)
    double* __restrict__ sc;                                     // [NA][N - 1][SW]
    if constexpr (LSC) sc = lsc; else sc = G.kgain(pr);
    const int SQ = (N - 1) * SW;
    // ---- phase A: work item = (step size q, step k, player i) -- NA x (N - 1) x P items over the team's lanes (with the step sizes inside an
    // item only (N - 1) P of the team's 64 NW lanes had work)
    {
        const bool pairs_on = P > 1 && (pr.has_colcost || pr.has_colavoid);
        const int KP = (N - 1) * P;
        for (int e = lane; e < NA * KP; e += NT) {
            const int q = e / KP, ek = e - q * KP, k = ek / P, i = ek % P, kn = k + 1;
            double aq = al[0];
#pragma unroll
            for (int t = 1; t < NA; t++) aq = (t == q) ? al[t] : aq;
            auto tq_ = [&](int idx) { const double a = ldz(idx), v = __builtin_fma(aq, ldd(idx), a); return rounded(idx < n ? a : v); };
            const double w = (kn < N - 1) ? dt : 1.0;
            const int so = k == 0 ? 0 : n + hx<C>(k - 1), uo_ = n + hu<C>(k, i), x1 = n + hx<C>(k);
            double* __restrict__ cf = sc + q * SQ + k * SW; double* __restrict__ tab = cf + NC;
            if constexpr (C::MODEL == ALG_MODEL_UNICYCLE) {
                const double th = tq_(so + 2 * P + i), v = tq_(so + 3 * P + i);
                const double om = tq_(uo_ + 0), ac = tq_(uo_ + 1);
                const double thm = th + (om * dt) * 0.5, vm = v + (ac * dt) * 0.5;
                double sn, cs; sincos(thm, &sn, &cs);
                cf[0 * P + i] = -dt * vm * sn; cf[1 * P + i] = dt * cs;
                cf[2 * P + i] = dt * vm * cs;  cf[3 * P + i] = dt * sn;
            }
            double xi[PD], ga[PD];
#pragma unroll
            for (int a = 0; a < PD; a++) { xi[a] = tq_(x1 + a * P + i); ga[a] = 0.0; }
#pragma unroll
            for (int jj = 0; jj < P - 1; jj++) {
                const int j = jj < i ? jj : jj + 1;
                double gv[PD];
#pragma unroll
                for (int a = 0; a < PD; a++) gv[a] = 0.0;
                if (pairs_on) {
                    double dl[PD];
#pragma unroll
                    for (int a = 0; a < PD; a++) dl[a] = xi[a] - tq_(x1 + a * P + j);
                    const double dl0 = dl[0], dl1 = dl[1];
                    const double s2 = pair_dist2(dl0, dl1);
                    if (pr.has_colcost) {
                        const double nrm = sqrt(s2), mu = pr.cc_mu[i], rad = pr.cc_radius[i];
                        if (fmax(0.0, rad - nrm) > 0.0) {
                            const double eps = 1e-10, eps_norm = eps * sqrt((double)n);
                            const double g0 = mu * (rad * (eps + dl0) / (eps_norm + nrm) - dl0);
                            const double g1 = mu * (rad * (eps + dl1) / (eps_norm + nrm) - dl1);
                            gv[0] += w * (-g0); gv[1] += w * (-g1);
                        }
                    }
                    if (pr.has_colavoid) {
                        const double Rr = pr.ca_pair_r[i * MAXP + j];
                        const double on = (double)((pr.ca_mask[i] >> j) & 1u);
                        const double c = ca_value(on, Rr, s2);
                        const int ci = con_col<C>(N, pairq<C>(i, j), kn);
                        const double lm = gld(G.lam(pr), ci), am = on * al_active_mu(c, lm, gld(G.mu(pr), ci));
                        const double wl = fma(am, c, on * lm);
#pragma unroll
                        for (int a = 0; a < PD; a++) gv[a] = __builtin_fma(-2.0 * dl[a], wl, gv[a]);
                    }
                }
#pragma unroll
                for (int a = 0; a < PD; a++) { ga[a] += gv[a]; tab[(i * P + j) * PD + a] = -gv[a]; }
            }
#pragma unroll
            for (int a = 0; a < PD; a++) tab[(i * P + i) * PD + a] = ga[a];
        }
        game_sync();
    }
    double l1[NA];
#pragma unroll
    for (int q = 0; q < NA; q++) l1[q] = 0.0;
    auto add = [&](int q, double r, double dprox) { const double rr = prox ? r + reg * dprox : r; l1[q] += fabs(rr); };
    for (int k0 = 0; k0 < N - 1; k0 += LsMulti<C>::CHUNK) {            // (one chunk = the whole horizon for the team kernels)
    const int nst = (N - 1 - k0) < LsMulti<C>::CHUNK ? (N - 1 - k0) : LsMulti<C>::CHUNK;
    // (one-wavefront kernels whose fused pass deals its rows by lane roles: the same dealing here -- per lane the same rows in the same order)
    constexpr bool LRG = LsMulti<C>::ONEW && lane_roles_v<C>;
    // ---- rows opt_i,x_{k+1}[a]
    auto gx = [&](int ks, int ei) {
        const int k = k0 + ks, i = ei / n, a = ei % n;
        const int zo = n + k * b;
        const bool has_next = (k + 1 <= N - 2);
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        const int lo = zo + (has_next ? b : 0) + n + m + i * n;
        const bool own = (a % P == i);
        const double tqv = G.Qd(pr)[i * ni + a / P], txv = G.xf(pr)[i * ni + a / P];
        const double tq = own ? tqv : 0.0, tx = own ? txv : 0.0;
        const double xr = ldz(zo + a);
#pragma unroll
        for (int q = 0; q < NA; q++) {
            const double* cfq = sc + q * SQ + (k + (has_next ? 1 : 0)) * SW; const double* tabq = sc + q * SQ + k * SW + NC;
            double r = -tv(q, zo + n + m + ei);
            {
                const double t = AT_vec<C>(cfq, dt, [&](int rr) { return tv(q, lo + rr); }, a);
                r += has_next ? t : 0.0;
            }
            const double xa = tv(q, zo + a);
            r += w * (tq * (xa - tx));
            { const double gv = tabq[(i * P + a % P) * PD + (a < PD * P ? a / P : 0)]; r += (a < PD * P) ? gv : 0.0; }
            add(q, r, xa - xr);
        }
    };
    if constexpr (LRG) {
        constexpr int RPS = P * n, XS = NT / RPS;
        const int ls = lane / RPS, ei = lane % RPS;
        if (ls < XS) for (int ks = ls; ks < nst; ks += XS) gx(ks, ei);
    } else for (int e = lane; e < nst * P * n; e += NT) gx(e / (P * n), e % (P * n));
    // ---- rows opt_i,u_{i,k}[c]
    auto gu = [&](int ks, int c) {
        const int k = k0 + ks, i = c % P;
        const int zo = n + k * b, lo = zo + n + m + i * n, uo = zo + n + uoff<C>(c);
        const double tr = G.Rd(pr)[(c % P) * mi + c / P], tu = G.uf(pr)[(c % P) * mi + c / P];
        const double ur = ldz(uo);
        double lmc[2] = {0.0, 0.0}, muc[2] = {0.0, 0.0};                 // multipliers of the two control bounds of this control (shared by the step sizes)
        if (pr.has_ctl) {
#pragma unroll
            for (int half = 0; half < 2; half++) { const int ci = con_ctl<C>(pr, k, half * m + c); lmc[half] = gld(G.lam(pr), ci); muc[half] = gld(G.mu(pr), ci); }
        }
#pragma unroll
        for (int q = 0; q < NA; q++) {
            const double u = tv(q, uo);
            double g = 0.0;
            if (pr.has_ctl) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const double cv = half == 0 ? u - pr.umax[c] : pr.umin[c] - u;
                    if (isfinite(cv)) {
                        const double lm = lmc[half], am = al_active_mu(cv, lm, muc[half]);
                        const double wl = lm + am * cv;
                        g += (half == 0 ? wl : -wl);
                    }
                }
            }
            const double r = dt * (tr * (u - tu)) + g + BT_vec<C>(sc + q * SQ + k * SW, dt, [&](int rr) { return tv(q, lo + rr); }, c);
            add(q, r, u - ur);
        }
    };
    if constexpr (LRG) {
        constexpr int US = NT / m;
        const int ls = lane / m, c = lane % m;
        if (ls < US) for (int ks = ls; ks < nst; ks += US) gu(ks, c);
    } else for (int e = lane; e < nst * m; e += NT) gu(e / m, e % m);
    // ---- rows dyn_k[a]
    auto gd = [&](int ks, int a) {
        const int k = k0 + ks;
        const int zo = n + k * b, po = (k == 0) ? 0 : zo - b;
#pragma unroll
        for (int q = 0; q < NA; q++) {
            double xn;
            if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                const int j = a < m ? a : a - m;
                const double uj = tv(q, zo + n + uoff<C>(j)), base = tv(q, po + a), vel = tv(q, po + j + m);
                const double vm = vel + (uj * dt) * 0.5;
                xn = base + (a < m ? vm : uj) * dt;
            } else {
                const double* Ck = sc + q * SQ + k * SW;
                const int blkk = a / P, i = a % P;
                const double ua = tv(q, zo + n + uoff<C>(P + i)), base = tv(q, po + a), vel = tv(q, po + 3 * P + i);
                const double uo = tv(q, zo + n + uoff<C>((blkk >= 2 ? blkk - 2 : 0) * P + i));
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 1 : 3) * P + i];
                xn = (blkk <= 1) ? base + vm * cf : base + uo * dt;
            }
            add(q, xn - tv(q, zo + a), 0.0);
        }
    };
    if constexpr (LRG) {
        constexpr int DS = NT / n;
        const int ls = lane / n, a = lane % n;
        if (ls < DS) for (int ks = ls; ks < nst; ks += DS) gd(ks, a);
    } else for (int e = lane; e < nst * n; e += NT) gd(e / n, e % n);
    }
    // ---- the team's sums, wavefront by wavefront like team_combine
    __shared__ double redm[C::NW][NA];
    {
        double ws[NA];
#pragma unroll
        for (int q = 0; q < NA; q++) ws[q] = wave_sum(l1[q]);
        const int wv = game_tid() >> 6, l = game_tid() & 63;
        if (l == 0) {
#pragma unroll
            for (int q = 0; q < NA; q++) redm[wv][q] = ws[q];
        }
        game_sync();
#pragma unroll
        for (int q = 0; q < NA; q++) {
            double t = 0.0;
#pragma unroll
            for (int x = 0; x < C::NW; x++) t += redm[x][q];
            l1reg[q] = t;
        }
        game_sync();
    }
}

// [z | dz] of the search into LDS (every thread of the team; the caller's barrier is inside)
template <class C>
__device__ __forceinline__ void ls_stage_traj(CPR pr0, const Game& G0, double* lz) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    const double* __restrict__ zs = G.z(0); const double* __restrict__ dz = G.z(2);
    const int TL = phase_int(pr.traj_len), lane = phase_lane();
    constexpr int U = 4;
    for (int e0 = lane; e0 < TL; e0 += U * C::NT) {
        double a[U], d[U];
#pragma unroll
        for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT, ec = e < TL ? e : e0; a[t] = gld(zs, ec); d[t] = gld(dz, ec); }
#pragma unroll
        for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < TL) { lz[e] = a[t]; lz[TL + e] = d[t]; } }
    }
    game_sync();
}

// update_traj!(target, source, alpha, delta) (primal_dual_traj.jl:109-128): coalesced axpy over the S entries
typedef double double2_t __attribute__((ext_vector_type(2)));
template <class C>
__device__ __forceinline__ void update_traj(CPR pr0, const Game& G0, int tsel, int ssel, double alpha) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    double* tgt = G.z(tsel); const double* src = G.z(ssel); const double* dz = G.z(2);
    // pure streaming pass: 16 bytes per lane and four independent load pairs in flight per pass.  Every game's buffers start
    // 16-byte aligned when traj_len is even (n is always even); otherwise the scalar loop runs.
    constexpr int U = 4;
    const int S = phase_int(pr.S), lane = phase_lane();
    if ((pr.traj_len & 1) == 0) {
        const int S2 = S >> 1;                           // pairs; a last odd element is handled below
        const double2_t* __restrict__ s2 = reinterpret_cast<const double2_t*>(src + C::n);
        const double2_t* __restrict__ d2 = reinterpret_cast<const double2_t*>(dz + C::n);
        double2_t* __restrict__ t2 = reinterpret_cast<double2_t*>(tgt + C::n);
        for (int e0 = lane; e0 < S2; e0 += U * C::NT) {
            double2_t a[U], d[U];
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; const int ec = e < S2 ? e : e0; a[t] = gld_t(s2, ec); d[t] = gld_t(d2, ec); }
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < S2) { double2_t v; v.x = __builtin_fma(alpha, d[t].x, a[t].x); v.y = __builtin_fma(alpha, d[t].y, a[t].y); gst_t(t2, e, v); } }
        }
        if ((S & 1) && lane == 0) tgt[C::n + S - 1] = __builtin_fma(alpha, dz[C::n + S - 1], src[C::n + S - 1]);
    } else {
        for (int e0 = lane; e0 < S; e0 += U * C::NT) {
            double a[U], d[U];
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; const int ec = e < S ? e : e0; a[t] = gld(src + C::n, ec); d[t] = gld(dz + C::n, ec); }
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < S) gst(tgt + C::n, e, __builtin_fma(alpha, d[t], a[t])); }
        }
    }
}
// Δ_step (primal_dual_traj.jl:130-147)
template <class C>
__device__ __forceinline__ double delta_step(CPR pr, const double* dz, double alpha) {
    double s = 0;
    for (int e = phase_lane(); e < (pr.N - 1) * (C::n + C::m); e += WAVE) {
        const int k = e / (C::n + C::m), a = e % (C::n + C::m);
        s += fabs(dz[C::n + k * C::b + a]);
    }
    s = wave_sum(s);
    s *= alpha;
    s /= (double)((pr.N - 1) * (C::n + C::m));
    return s;
}


} // namespace alg
