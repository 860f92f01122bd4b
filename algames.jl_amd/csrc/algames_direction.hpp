// algames_direction.hpp -- Newton direction: structured elimination of the KKT system (tile and dense variants) and its refinement gate
// (part of the device code of libalgames_hip.so; included by algames_device.hpp, which holds the shared declarations and the file-level
// description of the execution model)
#pragma once
#include "algames_device.hpp"

#ifndef ALG_DENSE_TOL_FACTOR
#define ALG_DENSE_TOL_FACTOR 0x1p-7      // gate tolerance of the dense-direction configurations relative to Params::refine_tol (refined_direction)
#endif

namespace alg {

// ================================================================================================
// Newton direction: structured elimination of the KKT system (see file header)
// ================================================================================================
// (i,r,c) entry of the position block of Q^_i built from the pair Hessian table Hh (sign pattern [[+H,-H],[-H,+H]])
template <class C>
__device__ __forceinline__ double pairblock(const double* Hh, int i, int r, int c) {
    constexpr int P = C::P;
    if (!C::POS) return 0.0;
    constexpr int NS = C::NS;
    const int jr = r % P, ar = r / P, jc = c % P, ac = c / P, hidx = C::sym(ar, ac);
    double e = 0.0;
    if (jr == i && jc == i) e = Hh[NS * C::NPAIR + NS * i + hidx];          // Hd_i = sum_j Hh(i,j) (+ wall / circle terms)
    else if (jr == i) e = -Hh[pairq<C>(i, jc) * NS + hidx];
    else if (jc == i) e = -Hh[pairq<C>(i, jr) * NS + hidx];
    else if (jr == jc) e = Hh[pairq<C>(i, jr) * NS + hidx];
    return e;
}
// Entry (r,c) of Q^_i = sum_j E[i][j].Q + state-constraint hess + reg I at a knot with stage weight w
template <class C>
__device__ __forceinline__ double qhat_entry(const double* qd, const double* Hh, int i, int r, int c, double w, double reg) {
    double e = 0.0;
    if (r == c) { e = reg; if (r % C::P == i) e += w * qd[i * C::ni + r / C::P]; }
    if (C::POS && r < C::PD * C::P && c < C::PD * C::P) e += pairblock<C>(Hh, i, r, c);
    return e;
}

// the double held by lane ^ 32 (gfx950 v_permlane32_swap: upper half of vdst <-> lower half of src)
__device__ __forceinline__ double xchg32(double v, bool lower_half) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(lower_half ? b[1] : b[0], lower_half ? a[1] : a[0]);
}
// A' X for the double integrator on an MFMA result tile (register r4 = row lq + 4 r4): (A' X)[r] = X[r] + dt X[r - m] for r >= m.
// Row r - m sits in the same lane (m % 4 == 0) or in lane ^ 32 (m % 4 == 2): a few FMAs instead of a second MFMA product
// (on MI355X an f64 MFMA holds the matrix pipe as long as sixteen FMAs hold the VALU).
template <class C>
__device__ __forceinline__ double4_t di_AT_tile(double4_t x, double dt, int lq) {
    constexpr int m = C::m, q = m / 4, sh = m % 4, NR = (C::n + 3) / 4;     // registers r4 < NR hold rows < n
    static_assert(sh == 0 || sh == 2, "double-integrator tile shift");
    double4_t y = x;
    if constexpr (sh == 0) {
#pragma unroll
        for (int r4 = q; r4 < NR; r4++) y[r4] = fma(dt, x[r4 - q], x[r4]);
    } else {
        const bool up = lq >= 2;                 // rows of this lane with r >= m: partner register r4 - q (up) or r4 - q - 1
        double part[4];
#pragma unroll
        for (int r4 = 0; r4 + q < NR; r4++) part[r4] = xchg32(x[r4], !up);
#pragma unroll
        for (int r4 = q; r4 < NR; r4++) {
            const double pu = part[r4 - q], pl = (r4 - q - 1 >= 0) ? part[r4 - q - 1] : 0.0;
            const double src = up ? pu : pl;
            const double dte = (up || r4 - q - 1 >= 0) ? dt : 0.0;
            y[r4] = fma(dte, src, x[r4]);
        }
    }
    return y;
}

// A' X for the 4-player unicycle / bicycle (n = 16): the state blocks have 4 rows, so register r4 of lane group lq holds
// block r4 of player lq -- the sparse A' (AT_vec above) is lane-local on the result tile.
template <class C>
__device__ __forceinline__ double4_t p4_AT_tile(double4_t x, const double* coef, int lq) {
    static_assert(C::P == 4 && C::n == 16, "lane-local A' needs 4-row state blocks");
    constexpr int P = C::P;
    double4_t y = x;
    if constexpr (C::MODEL == ALG_MODEL_UNICYCLE) {
        y[2] = x[2] + coef[0 * P + lq] * x[0] + coef[2 * P + lq] * x[1];
        y[3] = x[3] + coef[1 * P + lq] * x[0] + coef[3 * P + lq] * x[1];
    } else {
        y[2] = x[2] + coef[1 * P + lq] * x[0] + coef[3 * P + lq] * x[1] + coef[4 * P + lq] * x[3];
        y[3] = x[3] + coef[0 * P + lq] * x[0] + coef[2 * P + lq] * x[1];
    }
    return y;
}

// A' X for the 3-player unicycle / bicycle (n = 12) on the result tile: rows 6 + i and 9 + i of player i take c * (row i) +
// c' * (row 3 + i) (the bicycle's speed row 6 + i also c'' * (row 9 + i)).  Rows live in (lane group lq = r % 4, register
// r / 4), so the source rows of a target row are fetched from other 16-lane groups with ds_bpermute (two rounds: targets in
// register 1 -- lane groups 2, 3 -- and in register 2 -- all groups); the per-lane source addresses and coefficient indices
// are the same for every player's tile.
template <class C>
struct P3Gather {
    static constexpr bool BIC = C::MODEL == ALG_MODEL_BICYCLE;
    int aA1, aB1, aC1, aA2, aB2, aC2;    // byte addresses (4 * source lane) of rows i, 3 + i, 9 + i for the register-1 / register-2 target
    int kA1, kB1, kC1, kA2, kB2, kC2;    // coefficient indices (into the step's coef table) of those targets
    bool t1, c2on;                        // register 1 is a target (rows 6, 7); register 2's target takes the third term (row 8)
    __device__ __forceinline__ void init(int lq, int lrow) {
        static_assert(C::MODEL != ALG_MODEL_DOUBLE_INTEGRATOR && C::P == 3, "3-player unicycle / bicycle tile gather");
        constexpr int P = 3;
        auto setup = [&](int rt, int& aA, int& aB, int& aC, int& kA, int& kB, int& kC) {
            const int i = (rt - 6) % 3, sB = 3 + i, sC = 9 + i, hi = rt >= 9 ? 1 : 0;
            aA = 4 * (16 * i + lrow); aB = 4 * (16 * (sB % 4) + lrow); aC = 4 * (16 * (sC % 4) + lrow);
            // unicycle: heading rows (6..8) use coef 0 / 2, speed rows (9..11) coef 1 / 3; bicycle: speed rows (6..8) coef 1 / 3 / 4,
            // heading rows (9..11) coef 0 / 2
            const int ta = BIC ? (hi ? 0 : 1) : hi, tb = BIC ? (hi ? 2 : 3) : 2 + hi;
            kA = ta * P + i; kB = tb * P + i; kC = 4 * P + i;
        };
        t1 = lq >= 2;
        setup(t1 ? 4 + lq : 6, aA1, aB1, aC1, kA1, kB1, kC1);
        setup(8 + lq, aA2, aB2, aC2, kA2, kB2, kC2);
        c2on = (lq == 0);
    }
    __device__ __forceinline__ static double gather(int addr, double v) {
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ double4_t apply(double4_t x, const double* coef, int lq) const {
        const double offB = (lq == 3) ? x[0] : x[1];             // row 3 sits in register 0 of group 3, rows 4, 5 in register 1 of groups 0, 1
        const double gA1 = gather(aA1, x[0]), gB1 = gather(aB1, offB), gA2 = gather(aA2, x[0]), gB2 = gather(aB2, offB);
        double u1 = coef[kA1] * gA1 + coef[kB1] * gB1, u2 = coef[kA2] * gA2 + coef[kB2] * gB2;
        if constexpr (BIC) {                                      // rows 9 + i: register 2 of groups 1, 2, 3
            const double gC1 = gather(aC1, x[2]), gC2 = gather(aC2, x[2]);
            u1 += coef[kC1] * gC1; u2 += c2on ? coef[kC2] * gC2 : 0.0;
        }
        double4_t y = x;
        y[1] = x[1] + (t1 ? u1 : 0.0);
        y[2] = x[2] + u2;
        return y;
    }
};

// wave-uniform broadcast of lane `src`'s double
__device__ __forceinline__ double bcast_lane(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
// Partial-pivot Gauss-Jordan of the m x m control system with its n+1 right-hand sides, column-per-lane:
// lane c < M owns column c of W, lane M + c' owns right-hand-side column c'.  Row operations are lane-local; the
// pivot column is broadcast with v_readlane, so the pivot choice is wave-uniform.  On exit the right-hand-side
// lanes hold the solution columns.  Returns 0 or 1 (singular).
// 1/x for a pivot (normal, non-zero): hardware reciprocal + two Newton steps instead of the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);      // (v_rcp_f64 alone: 4.6e-8 relative; one step 2.2e-15; two steps 1.1e-16 = correctly rounded-ish: tests/probes/rcp_test.hip)
}
// Same elimination with XC columns per lane (lane t: columns t, t + 64, ...; the pivot columns c < M < 64 are first columns): the
// control system of the dense direction (m + n + 1 up to 65 columns) solved by one wavefront without LDS traffic or barriers.
template <int M, int XC>
__device__ __forceinline__ int gj_solve_cols_x(double (&col)[XC][M]) {
    int sing = 0;
#pragma unroll
    for (int c = 0; c < M; c++) {
        double pc[M];
#pragma unroll
        for (int r = 0; r < M; r++) pc[r] = bcast_lane(col[0][r], c);
        double best = fabs(pc[c]), oth = 0.0;
        double rpiv = fast_rcp(pc[c]);
#pragma unroll
        for (int r = c + 1; r < M; r++) oth = fmax(oth, fabs(pc[r]));
        if (__builtin_amdgcn_readfirstlane((int)(oth > best))) {
            int piv = c;
#pragma unroll
            for (int r = c + 1; r < M; r++) { const double v = fabs(pc[r]); if (v > best) { best = v; piv = r; } }
            piv = __builtin_amdgcn_readfirstlane(piv);
#pragma unroll
            for (int r = c + 1; r < M; r++) {
                if (piv == r) {
                    double t = pc[c]; pc[c] = pc[r]; pc[r] = t;
#pragma unroll
                    for (int x = 0; x < XC; x++) { t = col[x][c]; col[x][c] = col[x][r]; col[x][r] = t; }
                }
            }
            rpiv = fast_rcp(pc[c]);
        }
        if (!(best > 0.0) || !isfinite(best)) sing = 1;
#pragma unroll
        for (int x = 0; x < XC; x++) {
            const double prow = col[x][c] * rpiv;
#pragma unroll
            for (int r = 0; r < M; r++) if (r != c) col[x][r] -= pc[r] * prow;
            col[x][c] = prow;
        }
    }
    return sing;
}
template <int M>
__device__ __forceinline__ int gj_solve_cols(double (&col)[M]) {
    int sing = 0;
#pragma unroll
    for (int c = 0; c < M; c++) {
        double pc[M];
#pragma unroll
        for (int r = 0; r < M; r++) pc[r] = bcast_lane(col[r], c);
        // partial pivoting: the diagonal entry is the usual winner -> one max chain + one uniform test, the index
        // search and the row swap only run when another row really has the larger magnitude
        double best = fabs(pc[c]), oth = 0.0;
        double rpiv = fast_rcp(pc[c]);          // started before the pivot test: the two dependency chains overlap
#pragma unroll
        for (int r = c + 1; r < M; r++) oth = fmax(oth, fabs(pc[r]));
        if (__builtin_amdgcn_readfirstlane((int)(oth > best))) {
            int piv = c;
#pragma unroll
            for (int r = c + 1; r < M; r++) { const double v = fabs(pc[r]); if (v > best) { best = v; piv = r; } }
            piv = __builtin_amdgcn_readfirstlane(piv);
#pragma unroll
            for (int r = c + 1; r < M; r++) {
                if (piv == r) { double t = col[c]; col[c] = col[r]; col[r] = t; t = pc[c]; pc[c] = pc[r]; pc[r] = t; }
            }
            rpiv = fast_rcp(pc[c]);
        }
        if (!(best > 0.0) || !isfinite(best)) sing = 1;
        const double prow = col[c] * rpiv;
#pragma unroll
        for (int r = 0; r < M; r++) if (r != c) col[r] -= pc[r] * prow;
        col[c] = prow;
    }
    return sing;
}
// ---- the same elimination with the pivot column broadcast by DPP -----------------------------------------------------------
// gfx950 has v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast:L (lane L of every 16-lane row feeds the whole row): one
// instruction does "broadcast lane L's register and FMA" where the v_readlane form needs two scalar reads per double plus the
// FMA.  The broadcast stays inside a 16-lane row, so every row that holds right-hand-side columns carries its own replica of
// the M columns of W in its lanes 0..M-1 (they are eliminated redundantly: free in SIMD terms).  Lane layout of a row:
// [W col 0..M-1 | right-hand-side columns]; see GjLanes.
// The elimination of one pivot is ONE asm statement (hipcc pads no hazards inside or around it, cdna_hip_programming.md 5.7:
// a VALU write followed by a DPP read of the same VGPR needs two wait states -> s_nop 1 on both ends).
template <int L>
__device__ __forceinline__ double bcast16(double v) {           // lane L of the reader's 16-lane row (v_mov_b64_dpp row_newbcast)
    long long x = __double_as_longlong(v);
    x = __builtin_amdgcn_update_dpp(x, x, 0x150 + L, 0xf, 0xf, false);
    return __longlong_as_double(x);
}
#define ALG_DPPF(d) "v_fmac_f64_dpp %" #d ", %" #d ", -%[np] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\t"
template <int M> struct DppElim;
// col[r] -= bcast_L(col[r]) * np for every r (the caller overwrites col[L] afterwards).  No trailing pad: the next DPP reader of
// these registers is bcast16_asm / the next elimination, which open with their own s_nop 1.
template <> struct DppElim<2> { template <int L> __device__ __forceinline__ static void run(double (&c)[2], double np) {
    asm volatile("s_nop 1\n\t" ALG_DPPF(0) ALG_DPPF(1) "" : "+v"(c[0]), "+v"(c[1]) : [np] "v"(np), [l] "n"(L)); } };
template <> struct DppElim<4> { template <int L> __device__ __forceinline__ static void run(double (&c)[4], double np) {
    asm volatile("s_nop 1\n\t" ALG_DPPF(0) ALG_DPPF(1) ALG_DPPF(2) ALG_DPPF(3) ""
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : [np] "v"(np), [l] "n"(L)); } };
template <> struct DppElim<6> { template <int L> __device__ __forceinline__ static void run(double (&c)[6], double np) {
    asm volatile("s_nop 1\n\t" ALG_DPPF(0) ALG_DPPF(1) ALG_DPPF(2) ALG_DPPF(3) ALG_DPPF(4) ALG_DPPF(5) ""
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : [np] "v"(np), [l] "n"(L)); } };
template <> struct DppElim<8> { template <int L> __device__ __forceinline__ static void run(double (&c)[8], double np) {
    asm volatile("s_nop 1\n\t" ALG_DPPF(0) ALG_DPPF(1) ALG_DPPF(2) ALG_DPPF(3) ALG_DPPF(4) ALG_DPPF(5) ALG_DPPF(6) ALG_DPPF(7) ""
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : [np] "v"(np), [l] "n"(L)); } };
#undef ALG_DPPF
// acc += p * (lane L of the reader's 16-lane row of v): one v_fmac_f64_dpp.  The first term of a chain opens with s_nop 1 (v may come
// straight from a VALU select); acc and p are ordinary operands.
template <int L, bool FIRST>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double v, double p) {
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(p), "n"(L));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(p), "n"(L));
}
// acc + sum_c p[c] * (lane c of the row of v), c = 0 .. NC-1, in that order (the FMA chain `a += p[c] * x[c]` with x spread over a row)
template <int NC, int C0 = 0, class PF>
__device__ __forceinline__ void rowdot_dpp_f(double& acc, double v, PF&& p) {          // p(c): the c-th coefficient, fetched where it is used
    if constexpr (C0 < NC) { fmac_rowbcast<C0, C0 == 0>(acc, v, p(C0)); rowdot_dpp_f<NC, C0 + 1>(acc, v, p); }
}
template <int NC>
__device__ __forceinline__ void rowdot_dpp(double& acc, double v, const double* p) { rowdot_dpp_f<NC>(acc, v, [&](int c) { return p[c]; }); }
// The same chain with its coefficients requested G at a time, one group ahead of the FMAs that use them.  The FMAs are volatile asm
// statements, which the compiler does not move loads across: in rowdot_dpp_f every coefficient's LDS read sits between two FMAs of the
// chain and is waited for on the spot -- NC exposed LDS round trips (a lone wavefront: ~100 cycles each, 1.6 K cycles for the sixteen
// gain columns of a forward-sweep step of the 4-player unicycle).  Here: ceil(NC / G) groups, the first two requested before the first
// FMA.  Same products in the same order (bit-identical); costs 2 G live doubles, so the 128-register kernels take small groups.
template <int C0, int G, int NC, int I = 0, class PF>
__device__ __forceinline__ void rowdot_group_load(double (&buf)[G], PF&& p) {
    if constexpr (I < G && C0 + I < NC) { buf[I] = p(C0 + I); rowdot_group_load<C0, G, NC, I + 1>(buf, p); }
}
template <int C0, int G, int NC, int I = 0>
__device__ __forceinline__ void rowdot_group_fmac(double& acc, double v, const double (&buf)[G]) {
    // (four terms per asm statement -- no s_nop padding at the statement boundaries -- measured neutral on C2 / C3 / C5 in round 4, profiles/r04_ab_asm4_*.txt)
    if constexpr (I < G && C0 + I < NC) { fmac_rowbcast<C0 + I, C0 + I == 0>(acc, v, buf[I]); rowdot_group_fmac<C0, G, NC, I + 1>(acc, v, buf); }
}
template <int NC, int G, int C0, class PF>
__device__ __forceinline__ void rowdot_pipe(double& acc, double v, PF&& p, const double (&cur)[G]) {
    if constexpr (C0 + G < NC) {
        double nxt[G];
        rowdot_group_load<C0 + G, G, NC>(nxt, p);
        rowdot_group_fmac<C0, G, NC>(acc, v, cur);
        rowdot_pipe<NC, G, C0 + G>(acc, v, p, nxt);
    } else rowdot_group_fmac<C0, G, NC>(acc, v, cur);
}
template <int NC, int G, class PF>
__device__ __forceinline__ void rowdot_dpp_g(double& acc, double v, PF&& p) {
    if constexpr (G <= 0) rowdot_dpp_f<NC>(acc, v, p);
    else { double cur[G]; rowdot_group_load<0, G, NC>(cur, p); rowdot_pipe<NC, G, 0>(acc, v, p, cur); }
}
// Two interleaved half-chains (round 6, ALG_R6_ROWDOT_SPLIT): terms 0 .. H-1 accumulate into acc, terms H .. NC-1 into a second accumulator, the two
// are added at the end -- the dependent chain is H + 1 instructions long instead of NC.  A lone wavefront waits out every link of these chains (the
// forward sweep's du = -(Y dx + y0), the backward sweep's y_i = P_i rd + s_i: 12 links each at C2).  Another association of the same sum: results
// move at rounding level against the single chain.  Coefficients are requested in groups of G (two terms of each half per group of four).
template <int NC, int H, int I, int G2, class PF>
__device__ __forceinline__ void rowdot_split_group_load(double (&ca)[G2], double (&cb)[G2], PF&& p) {
#pragma unroll
    for (int t = 0; t < G2; t++) { ca[t] = (I + t < H) ? p(I + t) : 0.0; cb[t] = (H + I + t < NC) ? p(H + I + t) : 0.0; }
}
template <int NC, int H, int I, int G2, int T = 0>
__device__ __forceinline__ void rowdot_split_group_fmac(double& a, double& b2, double v, const double (&ca)[G2], const double (&cb)[G2]) {
    if constexpr (T < G2) {
        if constexpr (I + T < H) fmac_rowbcast<I + T, (I + T == 0)>(a, v, ca[T]);
        if constexpr (H + I + T < NC) fmac_rowbcast<H + I + T, false>(b2, v, cb[T]);
        rowdot_split_group_fmac<NC, H, I, G2, T + 1>(a, b2, v, ca, cb);
    }
}
template <int NC, int H, int I, int G2, class PF>
__device__ __forceinline__ void rowdot_split_pipe(double& a, double& b2, double v, PF&& p, const double (&ca)[G2], const double (&cb)[G2]) {
    if constexpr (I + G2 < H) {
        double na[G2], nb[G2];
        rowdot_split_group_load<NC, H, I + G2, G2>(na, nb, p);
        rowdot_split_group_fmac<NC, H, I, G2>(a, b2, v, ca, cb);
        rowdot_split_pipe<NC, H, I + G2, G2>(a, b2, v, p, na, nb);
    } else rowdot_split_group_fmac<NC, H, I, G2>(a, b2, v, ca, cb);
}
template <int NC, int G, class PF>
__device__ __forceinline__ void rowdot_dpp_split(double& acc, double v, PF&& p) {
    constexpr int H = (NC + 1) / 2, G2 = (G >= 2 ? G / 2 : 1);
    double b2 = 0.0, ca[G2], cb[G2];
    rowdot_split_group_load<NC, H, 0, G2>(ca, cb, p);
    rowdot_split_pipe<NC, H, 0, G2>(acc, b2, v, p, ca, cb);
    acc += b2;
}
// The same with NCH interleaved sub-chains (terms [j H, (j + 1) H) on accumulator j, H = ceil(NC / NCH); accumulators added pairwise at the end):
// dependent length H + ceil(log2 NCH).  NCH = 2 is rowdot_dpp_split's association exactly.
template <int NC, int NCH, int H, int I, int G2, class PF, int... Ts>
__device__ __forceinline__ void rowdot_chains_load(double (&c)[NCH][G2], PF&& p, std::integer_sequence<int, Ts...>) {
    (([&] { constexpr int t = Ts / NCH, j = Ts % NCH, term = j * H + I + t; c[j][t] = (I + t < H && term < NC) ? p(term < NC ? term : 0) : 0.0; }()), ...);
}
template <int NC, int NCH, int H, int I, int G2, int... Ts>
__device__ __forceinline__ void rowdot_chains_fmac(double (&acc)[NCH], double v, const double (&c)[NCH][G2], std::integer_sequence<int, Ts...>) {
    (([&] { constexpr int t = Ts / NCH, j = Ts % NCH, term = j * H + I + t;
            if constexpr (I + t < H && term < NC) fmac_rowbcast<term, (I + t == 0 && j == 0)>(acc[j], v, c[j][t]); }()), ...);
}
template <int NC, int NCH, int H, int I, int G2, class PF>
__device__ __forceinline__ void rowdot_chains_pipe(double (&acc)[NCH], double v, PF&& p, const double (&cur)[NCH][G2]) {
    using Seq = std::make_integer_sequence<int, NCH * G2>;
    if constexpr (I + G2 < H) {
        double nxt[NCH][G2];
        rowdot_chains_load<NC, NCH, H, I + G2, G2>(nxt, p, Seq{});
        rowdot_chains_fmac<NC, NCH, H, I, G2>(acc, v, cur, Seq{});
        rowdot_chains_pipe<NC, NCH, H, I + G2, G2>(acc, v, p, nxt);
    } else rowdot_chains_fmac<NC, NCH, H, I, G2>(acc, v, cur, Seq{});
}
template <int NC, int NCH, int G, class PF>
__device__ __forceinline__ void rowdot_dpp_chains(double& acc0, double v, PF&& p) {
    static_assert(NCH >= 2 && NCH <= 4, "two to four sub-chains");
    constexpr int H = (NC + NCH - 1) / NCH, G2 = (G >= NCH ? G / NCH : 1);
    double acc[NCH], cur[NCH][G2];
    acc[0] = acc0;
#pragma unroll
    for (int j = 1; j < NCH; j++) acc[j] = 0.0;
    rowdot_chains_load<NC, NCH, H, 0, G2>(cur, p, std::make_integer_sequence<int, NCH * G2>{});
    rowdot_chains_pipe<NC, NCH, H, 0, G2>(acc, v, p, cur);
    if constexpr (NCH == 2) acc0 = acc[0] + acc[1];
    else if constexpr (NCH == 3) acc0 = (acc[0] + acc[1]) + acc[2];
    else acc0 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// (measured, profiles/r06_ab_rowdot_chains_c2.txt: three sub-chains neutral, four 2 % slower than two -- the extra accumulators' adds and the
// wider coefficient groups cost what the shorter chain saves; the w_k chain in two halves: neutral.  Two it is.)
#ifndef ALG_R6_ROWDOT_CHAINS
#define ALG_R6_ROWDOT_CHAINS 2
#endif
#ifndef ALG_R6_SADDR
#define ALG_R6_SADDR 1            // per-step base addresses of the sweeps' global accesses forced into scalar registers (uniform_u64): 1 gain stores and
                                  // forward-sweep prefetches (C2 +1.5 ... 2.6 %, profiles/r06_ab_saddr_*.txt); 2 also the backward sweep's record prefetch as
                                  // unconditional clamped loads -- measured 0.7 ... 2 % SLOWER at C2 / C4 (r06_ab_saddr_prefetch_*.txt), not taken
#endif
#ifndef ALG_R6_CURC
#define ALG_R6_CURC 1             // forward sweep: LDS double-buffer index of a step as a compile-time constant of the unrolled loop
#endif
#ifndef ALG_R6_PL1
#define ALG_R6_PL1 1              // forward sweep: |du| + |dx| summed and checked without per-step lane masks
#endif
#ifndef ALG_R6_FWD_LAND
#define ALG_R6_FWD_LAND 0         // where a forward-sweep step lands the next step's prefetched slice in LDS: 0 at its tail, 1 behind its first LDS reads, 2 behind the du chain
#endif                            // (measured at C2, profiles/r06_ab_fwd_land_c2.txt: 1 -0.6 %, 2 neutral)
#ifndef ALG_R6_WCHAIN_SPLIT
#define ALG_R6_WCHAIN_SPLIT 0      // the forward sweep's w_k = rx + Q^ dx chain (double integrator, FWDW) as two half-chains
#endif
// Measured in round 6, same box, alternating (profiles/r06_ab_rowdot_split_*.txt, r06_ab_rdone_*.txt): C2 +0.8 ... 1.5 % on one box, +0.6 ... 2.7 % on two
// others; C3 and C2 at 512 games neutral; the C5 loop (64 seeds x 200 steps, one launch = its slowest seed) 240 -> 202 K/s: no kernel effect -- at 100
// steps both forms take 186 ms -- but another rounding of the closed-loop trajectories, on which another seed meets a long line-search episode.
// So: the double-integrator kernels (every shape of them, so that they stay bit-identical with each other) form these two sums as half-chains,
// the unicycle / bicycle kernels keep the single chain -- nothing to gain there, and their calibrated accuracy tests (tests/test_gpu_refinement.py)
// and the BASELINE C5 trajectories stay what they were.
#ifndef ALG_R6_ROWDOT_SPLIT
#define ALG_R6_ROWDOT_SPLIT 1
#endif
template <class C> inline constexpr bool rowdot_split_v = ALG_R6_ROWDOT_SPLIT != 0 && C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR;
#ifndef ALG_RDG_W2
#define ALG_RDG_W2 16         // coefficient group of the row-broadcast chains, 256-register kernels (0: fetch where used)
#endif
#ifndef ALG_RDG_W4
#define ALG_RDG_W4 4          // ... 128-register kernels
#endif
template <class C> inline constexpr int rowdot_group_v = C::WPE == 2 ? ALG_RDG_W2 : (C::WPE == 4 ? ALG_RDG_W4 : 0);
// Lane roles of the DPP elimination inside one wavefront: row q = lane / 16 holds W's columns in its lanes 0..M-1 and the
// right-hand-side columns q (16 - M) ... in the lanes behind them.
template <int M, int NRHS> struct GjLanes {
    static constexpr int RPR = 16 - M;                               // right-hand-side columns per 16-lane row
    static_assert(RPR > 0 && (NRHS + RPR - 1) / RPR <= 4, "the control system does not fit the four rows of a wavefront");
    __device__ __forceinline__ static bool wlane(int lane) { return (lane & 15) < M; }
    __device__ __forceinline__ static int rhs_col(int lane) { return (lane >> 4) * RPR + (lane & 15) - M; }     // < 0 on W lanes
    __device__ __forceinline__ static bool rhs(int lane) { const int c = rhs_col(lane); return !wlane(lane) && c < NRHS; }
    // column of [W | right-hand sides] this lane builds (idle lanes duplicate the last right-hand side)
    __device__ __forceinline__ static int column(int lane) { return wlane(lane) ? (lane & 15) : (rhs(lane) ? M + rhs_col(lane) : M + NRHS - 1); }
};
// One column per lane across the whole wavefront (lane c < M: column c of W, lane M + c': right-hand side c'): the layout of the
// v_readlane elimination gj_solve_cols, kept for the one configuration whose register budget the DPP form (replicated W columns in
// every row, two extra asm operand sets) does not fit: the 4-player extended bicycle, 8 controls x 17 right-hand sides at 256 VGPRs
template <int M, int NRHS> struct GjFlat {
    static_assert(M + NRHS <= WAVE, "one column per lane");
    __device__ __forceinline__ static bool rhs(int lane) { return lane >= M && lane < M + NRHS; }
    __device__ __forceinline__ static int column(int lane) { return lane < M + NRHS ? lane : M + NRHS - 1; }
};
// max |b[0..NB-1]|: v_max_f64 with |.| modifiers (through fmax() the compiler canonicalises every operand first), ONE asm statement
// (the compiler pads every statement boundary with an s_nop).  A tree of depth <= 3 instead of a chain of depth NB - 1 --
// max is exact and NaN-dropping either way, so the result is the same number; a lone wavefront waits for every link of the chain.
template <int NB>
__device__ __forceinline__ double absmax_below(const double* b) {
    static_assert(NB >= 1 && NB <= 7, "control system larger than the DPP elimination supports");
    double oth;
    if constexpr (NB == 1) oth = fabs(b[0]);
    else if constexpr (NB == 2) asm("v_max_f64 %0, |%1|, |%2|" : "=v"(oth) : "v"(b[0]), "v"(b[1]));
    else if constexpr (NB == 3) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|" : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]));
    else if constexpr (NB == 4) { double t1;
        asm("v_max_f64 %0, |%2|, |%3|\n\tv_max_f64 %1, |%4|, |%5|\n\tv_max_f64 %0, %0, %1"
            : "=&v"(oth), "=&v"(t1) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])); }
    else if constexpr (NB == 5) { double t1;
        asm("v_max_f64 %0, |%2|, |%3|\n\tv_max_f64 %1, |%4|, |%5|\n\tv_max_f64 %0, %0, |%6|\n\tv_max_f64 %0, %0, %1"
            : "=&v"(oth), "=&v"(t1) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4])); }
    else if constexpr (NB == 6) { double t1, t2;
        asm("v_max_f64 %0, |%3|, |%4|\n\tv_max_f64 %1, |%5|, |%6|\n\tv_max_f64 %2, |%7|, |%8|\n\tv_max_f64 %0, %0, %1\n\tv_max_f64 %0, %0, %2"
            : "=&v"(oth), "=&v"(t1), "=&v"(t2) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5])); }
    else { double t1, t2;
        asm("v_max_f64 %0, |%3|, |%4|\n\tv_max_f64 %1, |%5|, |%6|\n\tv_max_f64 %2, |%7|, |%8|\n\tv_max_f64 %0, %0, %1\n\tv_max_f64 %2, %2, |%9|\n\tv_max_f64 %0, %0, %2"
            : "=&v"(oth), "=&v"(t1), "=&v"(t2) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6])); }
    return oth;
}
template <int M, int C>
__device__ __forceinline__ void gj_dpp_pivot(double (&col)[M], int& sing) {
    // lane C of the row owns W's column C: its entries below the diagonal decide the pivot (wave-uniform: every row holds the
    // same replica; bit C of the ballot is row 0's lane C)
    double best = fabs(col[C]);
    // The diagonal entry is the pivot unless the test below says otherwise (it almost never does: W = R^ + V B is
    // dominated by its diagonal), so its broadcast and reciprocal -- rcp + four dependent FMAs -- start BEFORE the pivot search and
    // the two dependency chains overlap; a row exchange repeats them on the exchanged entry.  Same operations on the same numbers.
    double pvt, rpv = 0.0;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(col[C]), "n"(C));
    rpv = fast_rcp(pvt);
    unsigned long long need = 0;
    if constexpr (C + 1 < M) {
        const double oth = absmax_below<M - C - 1>(&col[C + 1]);
        need = __builtin_amdgcn_ballot_w64(oth > best);
    }
    if ((need >> C) & 1ull) {
        int piv = C;
#pragma unroll
        for (int r = C + 1; r < M; r++) { const double v = fabs(col[r]); if (v > best) { best = v; piv = r; } }
        piv = __builtin_amdgcn_readlane(piv, C);
#pragma unroll
        for (int r = C + 1; r < M; r++) {
            if (piv == r) { const double t = col[C]; col[C] = col[r]; col[r] = t; }
        }
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(col[C]), "n"(C));
        rpv = fast_rcp(pvt);
    }
    if (!(fabs(pvt) > 0.0) || !isfinite(pvt)) sing = 1;
    const double prow = col[C] * rpv;
    DppElim<M>::template run<C>(col, prow);
    col[C] = prow;
}
template <int M, int... Cs>
__device__ __forceinline__ void gj_dpp_all(double (&col)[M], int& sing, std::integer_sequence<int, Cs...>) { (gj_dpp_pivot<M, Cs>(col, sing), ...); }
template <int M>
__device__ __forceinline__ int gj_solve_cols_dpp(double (&col)[M]) {
    int sing = 0;
    gj_dpp_all<M>(col, sing, std::make_integer_sequence<int, M>{});
    return sing;
}
// Sparse pattern (<= 3 entries) of column `idx` of the n x (m + n) matrix [B_k | A_k]  (idx < m: B column, else A column)
template <class C>
__device__ __forceinline__ void col_pattern(const double* coef, double dt, int idx, bool useA, int (&rows)[C::NPAT + 1], double (&vals)[C::NPAT + 1]) {
    constexpr int m = C::m, n = C::n, P = C::P;
#pragma unroll
    for (int t = 0; t < C::NPAT + 1; t++) { rows[t] = 0; vals[t] = 0.0; }
    // slot NPAT addresses the extended part of V's rows: lane c < m picks its R^ slot, lane m + n the right-hand side g
    if (idx < m) { rows[C::NPAT] = n + 1 + idx; vals[C::NPAT] = 1.0; }
    else if (idx == m + n) { rows[C::NPAT] = n; vals[C::NPAT] = 1.0; }
    if (idx < m) {
        if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) { rows[0] = idx; vals[0] = 0.5 * dt * dt; rows[1] = idx + m; vals[1] = dt; }
        else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
            const int i = idx % P, kind = idx / P;
            rows[0] = i; rows[1] = P + i; rows[2] = 3 * P + i;
            if (kind == 0) { vals[0] = 0.5 * dt * coef[5 * P + i]; vals[1] = 0.5 * dt * coef[6 * P + i]; vals[2] = 0.5 * dt * coef[4 * P + i]; rows[3] = 2 * P + i; vals[3] = dt; }
            else { vals[0] = coef[7 * P + i]; vals[1] = coef[8 * P + i]; vals[2] = coef[9 * P + i]; }
        } else {
            const int i = idx % P, kind = idx / P;
            rows[0] = i; rows[1] = P + i; rows[2] = (2 + kind) * P + i;
            vals[0] = 0.5 * dt * coef[kind * P + i]; vals[1] = 0.5 * dt * coef[(2 + kind) * P + i]; vals[2] = dt;
        }
    } else if (useA && idx < m + n) {
        const int c = idx - m;
        rows[0] = c; vals[0] = 1.0;
        if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) { if (c >= m) { rows[1] = c - m; vals[1] = dt; } }
        else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
            const int blk = c / P, i = c % P;
            if (blk == 2) { rows[1] = i; vals[1] = coef[1 * P + i]; rows[2] = P + i; vals[2] = coef[3 * P + i]; rows[3] = 3 * P + i; vals[3] = coef[4 * P + i]; }
            else if (blk == 3) { rows[1] = i; vals[1] = coef[0 * P + i]; rows[2] = P + i; vals[2] = coef[2 * P + i]; }
        } else {
            const int blk = c / P, i = c % P;
            if (blk >= 2) { rows[1] = i; vals[1] = coef[(blk - 2) * P + i]; rows[2] = P + i; vals[2] = coef[blk * P + i]; }
        }
    }
}

// Expands [Hh | Hd] of a step record into the table hx[i][jr][jc][3] (entry = block of Q^_i between the positions of
// players jr and jc).  src/sgn are the per-lane loop-invariant source offsets and signs.
template <class C>
struct HxMap {
    static constexpr int SLOTS = (DirLds<C>::NHX + WAVE - 1) / WAVE;
    int src[SLOTS]; double sgn[SLOTS];
    __device__ __forceinline__ void init(int lane) {
        constexpr int P = C::P;
        using R = Rec<C>;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
            const int t = lane + q * WAVE;
            int so = R::HH; double sg = 0.0;
            if (C::POS && t < DirLds<C>::NHX) {
                constexpr int NS = C::NS;
                const int h = t % NS, jc = (t / NS) % P, jr = (t / (NS * P)) % P, i = t / (NS * P * P);
                if (jr == i && jc == i) { so = R::HD + NS * i + h; sg = 1.0; }
                else if (jr == i) { so = R::HH + NS * pairq<C>(i, jc) + h; sg = -1.0; }
                else if (jc == i) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = -1.0; }
                else if (jr == jc) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = 1.0; }
            }
            src[q] = so; sgn[q] = sg;
        }
    }
    __device__ __forceinline__ void expand(int lane, const double* Rc, double* hxt) const {
        if (!C::POS) return;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) { const int t = lane + q * WAVE; if (t < DirLds<C>::NHX) hxt[t] = sgn[q] * Rc[src[q]]; }
    }
};

// Non-zeros of [Q^_i | rx_i] (rx_i only when s_i rides in tile column n, n < 16): after the MFMA products wrote A'(P F) back, every lane adds its entries
//   (r, r): reg + w q_i,r (+ state-bound Hessian) (+ position-block diagonal)   (r, c) r != c < 2P: position block   (r, n): rx_i,r
// into row block i of Pm.  Per lane and pass: packed (dst | src << 11 | qi << 19), sign of the record source, diagonal flag.
// PS > 1 (team of two): the map of the players first, first + PS, ... only.
template <class C, int NTQ = WAVE, int PS = 1>
struct QaddMap {
    static constexpr int NP = C::PD * C::P;                // rows / columns of the position block
    static constexpr int OFF = C::POS ? NP * NP - NP : 0;
    static constexpr bool RXCOL = C::n < 16;             // s_i lives in tile column n (else it is updated on the VALU)
    static constexpr int QE = C::n + OFF + (RXCOL ? C::n : 0), QTOT = (C::P / PS) * QE, PASSES = (QTOT + NTQ - 1) / NTQ;
    unsigned code[PASSES]; float sgn[PASSES], dfl[PASSES];
    __device__ __forceinline__ static void hxsrc(int i, int jr, int jc, int h, int& so, float& sg) {
        using R = Rec<C>;
        so = 0; sg = 0.f;
        constexpr int NS = C::NS;
        if (jr == i && jc == i) { so = R::HD + NS * i + h; sg = 1.f; }
        else if (jr == i) { so = R::HH + NS * pairq<C>(i, jc) + h; sg = -1.f; }
        else if (jc == i) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = -1.f; }
        else if (jr == jc) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = 1.f; }
    }
    __device__ __forceinline__ void init(int lane, int first = 0) {
        constexpr int n = C::n, P = C::P, LDP = n + 1;
        using R = Rec<C>;
        static_assert(C::P * n * LDP < 2048 && R::LEN_SWEEP < 256 && P * n <= 64, "QaddMap packing");
#pragma unroll
        for (int q = 0; q < PASSES; q++) {
            const int e = lane + q * NTQ;
            int dst = 0, so = 0, qi = 0; float sg = 0.f, df = 0.f;
            if (e < QTOT) {
                const int i = (e / QE) * PS + first, t = e % QE;
                if (t < n) {
                    dst = i * n * LDP + t * LDP + t; qi = i * n + t; df = 1.f;
                    if (C::POS && t < NP) hxsrc(i, t % P, t % P, C::sym(t / P, t / P), so, sg);
                } else if (t < n + OFF) {
                    const int u = t - n, r = u / (NP - 1), cc = u % (NP - 1), c = cc < r ? cc : cc + 1;
                    dst = i * n * LDP + r * LDP + c;
                    hxsrc(i, r % P, c % P, C::sym(r / P, c / P), so, sg);
                } else {
                    const int r = t - n - OFF;
                    dst = i * n * LDP + r * LDP + n; so = R::RX + i * n + r; sg = 1.f;
                }
            }
            code[q] = (unsigned)dst | ((unsigned)so << 11) | ((unsigned)qi << 19); sgn[q] = sg; dfl[q] = df;
        }
    }
    __device__ __forceinline__ void apply(int lane, const double* Rc, const double* qdf, double* Pm, double reg, double w, int only_player) const {
        using R = Rec<C>;
        // every pass touches distinct entries of Pm: all loads first, then all stores (the compiler has to assume that a
        // pass's store aliases the next pass's loads and would serialise one LDS round trip per pass)
        double nv[PASSES];
#pragma unroll
        for (int q = 0; q < PASSES; q++) {
            const int e = lane + q * NTQ;
            nv[q] = 0.0;
            if (e < QTOT && (only_player < 0 || e / QE == only_player)) {
                const unsigned u = code[q];
                const int dst = u & 0x7ff, so = (u >> 11) & 0xff, qi = u >> 19;
                double dq = reg + w * qdf[qi];
                if constexpr (C::EXT) dq += Rc[R::RQ + qi];
                nv[q] = Pm[dst] + fma((double)sgn[q], Rc[so], (double)dfl[q] * dq);
            }
        }
#pragma unroll
        for (int q = 0; q < PASSES; q++) {
            const int e = lane + q * NTQ;
            if (e < QTOT && (only_player < 0 || e / QE == only_player)) Pm[code[q] & 0x7ff] = nv[q];
        }
    }
};

// Solves J d = -res for the step records left by assemble_pass<C,1> and writes d into the delta buffer
// (solver_methods.jl:87-88).  Returns ALG_STATUS_*.
// IBR = true: best response of player ip -- only x, u_ip, lambda_ip move (horizontal mask, newton_core.jl:249-294): the other
// players' value recursions are skipped, their rows of the control system become unit rows (du_j = 0), dlambda_j = 0.
// the double held by lane `src` (ds_bpermute; every lane has to execute it: a disabled source lane reads as garbage)
__device__ __forceinline__ double shfl_d(double v, int src) {
    const int a = src << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// v_mov_b32_dpp on both halves of a double: CTRL = 0x100 + k: lane i takes lane i + k of its 16-lane row (row_shl:k), 0x110 + k: lane i - k
// (row_shr:k); lanes whose source falls outside the row read 0.
template <int CTRL>
__device__ __forceinline__ double row_shift(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Row `lane` of A_k dx + B_k du with dx / du held one entry per lane (dx: lanes 0..n-1, du: lanes 0..m-1, joint control order):
// the forward sweep's state update without a trip through LDS.  Branch-free (the shuffles need all lanes); the entries are
// those of A_vec / B_vec.
template <class C>
__device__ __forceinline__ double fwd_next(const double* coef, double dt, double dxr, double duv, int lane) {
    constexpr int n = C::n, m = C::m, P = C::P;
    const int r = lane < n ? lane : 0;
    if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        // position row r takes the velocity dx[r + m] and its control du[r]; velocity row r takes du[r - m]: two shifts inside the 16-lane
        // row that holds dx and du (v_mov_b32_dpp row_shl / row_shr on the two halves of the double; the ds_bpermute form went through
        // the LDS crossbar: four LDS operations per step)
        static_assert(n <= 16 && m < 16, "dx and du live in one 16-lane row");
        const bool lo = r < m;
        // (both shifts are executed by every lane before the selects: a DPP read of a lane that a divergent branch has switched off returns 0)
        const double sx = row_shift<0x100 + m>(dxr), su = row_shift<0x110 + m>(duv);
        const double other = lo ? sx : dxr, uu = lo ? duv : su;
        const double a = lo ? dxr + dt * other : dxr;
        const double b = lo ? 0.5 * dt * dt * uu : dt * uu;
        return a + b;
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int blk = r / P, i = r % P;
        const double ps = shfl_d(dxr, 3 * P + i), vv = shfl_d(dxr, 2 * P + i), w0 = shfl_d(duv, i), w1 = shfl_d(duv, P + i);
        const double c1v = coef[(blk == 1 ? 2 : 0) * P + i], c2v = coef[(blk == 1 ? 3 : (blk == 3 ? 4 : 1)) * P + i];
        const double a1 = blk < 2 ? c1v : 0.0, a2 = blk != 2 ? c2v : 0.0;
        const double a = dxr + a1 * ps + a2 * vv;
        const double c0v = coef[(blk == 0 ? 5 : (blk == 1 ? 6 : 4)) * P + i], c3v = coef[(blk == 0 ? 7 : (blk == 1 ? 8 : 9)) * P + i];
        const double b0 = blk == 2 ? dt : 0.5 * dt * c0v, b1 = blk == 2 ? 0.0 : c3v;
        return a + (b0 * w0 + b1 * w1);
    } else {
        const int blk = r / P, i = r % P;
        // heading / speed of the row's player and its two controls, by shifts inside the 16-lane row that holds dx and du (executed by
        // every lane, selected afterwards): rows i and P + i read dx[2P + i], dx[3P + i], du[i], du[P + i]; rows 2P + i / 3P + i read du[i] / du[P + i]
        static_assert(n <= 16, "dx and du live in one 16-lane row");
        const double x1 = row_shift<0x100 + P>(dxr), x2 = row_shift<0x100 + 2 * P>(dxr), x3 = row_shift<0x100 + 3 * P>(dxr);
        const double u1 = row_shift<0x100 + P>(duv), d1 = row_shift<0x110 + P>(duv), d2 = row_shift<0x110 + 2 * P>(duv);
        const bool b0 = blk == 0;
        const double th = b0 ? x2 : x1, vv = b0 ? x3 : x2;
        const double w0 = blk >= 2 ? d2 : (b0 ? duv : d1), w1 = blk >= 2 ? d2 : (b0 ? u1 : duv);
        const bool pos = blk < 2;
        const double ca = coef[(pos ? 2 * blk : 0) * P + i], cb = coef[(pos ? 2 * blk + 1 : 1) * P + i];
        const double a = pos ? dxr + ca * th + cb * vv : dxr;
        const double b = pos ? 0.5 * dt * (ca * w0 + cb * w1) : dt * (blk == 2 ? w0 : w1);
        return a + b;
    }
}

// Scratch instrumentation (-DALG_PHASE_PROF, tests/probes/phase_prof.sh): shader-clock cycles per phase of the sweeps, accumulated
// into G.res(pr)[0..] (unused by the fused solver).  Never defined in the product build.
#ifdef ALG_PHASE_PROF
// (32-bit differences: the upper half of the 64-bit counter read is not dependable across s_memtime reads on this part)
#define ALG_PROF_DECL unsigned prof_t_ = (unsigned)__builtin_readcyclecounter(), prof_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define ALG_PROF(j) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); const unsigned d_ = t_ - prof_t_; prof_acc_[j] += d_ < (1u << 28) ? d_ : 0u; prof_t_ = t_; }
#define ALG_PROF_FLUSH if (game_tid() == 0) { for (int j_ = 0; j_ < 12; j_++) G.res(pr)[j_] += (double)prof_acc_[j_]; }
#elif defined(ALG_ISA_MARK)
// static accounting (tests/probes/isa_phases.py): the phase boundaries show up as comments in the -S output
#define ALG_PROF_DECL
#define ALG_PROF(j) asm volatile("; ALGMARK " #j ::: "memory");
#define ALG_PROF_FLUSH
#else
#define ALG_PROF_DECL
#define ALG_PROF(j)
#define ALG_PROF_FLUSH
#endif
// Flat parallel loop of the dense direction: entries e = tid, tid + BT, ... < total; U entries per thread and trip are evaluated
// together (independent dependency chains in flight; out-of-range slots re-evaluate the trip's first entry) and stored afterwards.
template <int U, class F, class S>
__device__ __forceinline__ void flat_loop(int tid, int BT, int total, F&& f, S&& st) {
    for (int e0 = tid; e0 < total; e0 += U * BT) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u * BT; v[u] = f(e < total ? e : e0); }
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u * BT; if (e < total) st(e, v[u]); }
    }
}

// ---- dense variant (Cfg::DENSE: QuadrotorGame, n = 12 p up to 48, dense 12 x 12 / 12 x 4 Jacobian blocks per player) ------------
// The same structured elimination with everything of one backward step LDS-resident and all threads of the workgroup on every
// phase: [P_i | s_i] [[F f],[0 1]] as ceil(n/16) x ceil((n+1)/16) tiles of v_mfma_f64_16x16x4_f64 chains (the tiles of a player are
// spread over the wavefronts), the block-diagonal A_{k+1}' applied from LDS, the m x (m + n + 1) control system solved by a
// partially pivoted Gauss-Jordan in LDS (one column per thread); records and gains are read from / written to HBM (L2) directly.
template <class C, bool IBR>
__device__ int newton_direction_dense(CPR pr0, const Game& G0, DirLds<C>& L, double reg, int ip, double* primal_l1) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, WC = DirLds<C>::WC, NK = m * (n + 1);
    constexpr int BT = C::NT, NWV = BT / WAVE;
    constexpr int FU = 2;                                     // entries per thread and trip of the flat loops
    constexpr int TR = (n + 15) / 16, TC = (n + 1 + 15) / 16, KBN = (n + 1 + 3) / 4;
    constexpr bool TIGHT = DirLds<C>::TIGHT;                  // ten players: see DirLds<C, true>
    constexpr int SKIP0 = DirLds<C>::SKIP0, SKIP1 = DirLds<C>::SKIP1;
    static_assert(NWV <= 4, "cross-wavefront reduction slots");
    static_assert(C::NW == 1 || C::NW >= 4, "every wavefront of the team runs this function (inner_iteration sends teams of two through wavefront 0 only)");
    using R = Rec<C>;
    const int N = phase_int(pr.N), tid = phase_lane(), lane = tid & 63, wv = tid >> 6, lrow = lane & 15, lq = lane >> 4;
    const double dt = phase_f64(pr.dt);
    const double* __restrict__ recs = G.rec(pr);
    const double* __restrict__ Qd = G.Qd(pr);
    double* __restrict__ kg = G.kgain(pr);
    auto& B = L.bw;
    for (int e = tid; e < (n + 1) * LDP; e += BT) B.Fx[e] = (e == n * LDP + n) ? 1.0 : 0.0;     // last row e_n: passes s_i through the product
    for (int e = tid; e < P * n * LDP; e += BT) B.Pm[e] = 0.0;
    int sing = 0;
    // Step records travel HBM -> registers -> LDS ahead of their use: step k - 1's record is requested at the tail of step k (after
    // the gains went out) and landed in the single LDS copy right after the value recursion of step k - 1 -- the last reader of the
    // previous coefficient block -- so the load latency hides behind the tail of one step and the recursion of the next
    // (PREF: only while a thread's share of a record is small -- the largest shapes, e.g. four quadrotors on one wavefront, would
    // run out of registers; they copy the record at the landing point instead)
    constexpr int RPT = (R::LEN_SWEEP + BT - 1) / BT;
    constexpr bool PREF = RPT <= 12;
    double pre[PREF ? RPT : 1];
    auto rec_load = [&](int kk) {
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < RPT; q++) { const int e = tid + q * BT; pre[q] = recs[(size_t)kk * R::LEN + (e < R::LEN_SWEEP ? e : 0)]; }
        }
    };
    auto rec_store = [&](int kk) {
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < RPT; q++) {
                const int e = tid + q * BT;
                if (e < C::NC) B.cf[e] = pre[q];
                else if (e < SKIP0) B.rs[e - C::NC] = pre[q];
                else if (e >= SKIP1 && e < R::LEN_SWEEP) B.rs[e - C::NC - (SKIP1 - SKIP0)] = pre[q];
            }
        } else {
            const double* Rk = recs + (size_t)kk * R::LEN;
            for (int e = tid; e < C::NC; e += BT) B.cf[e] = Rk[e];
            for (int e = tid; e < SKIP0 - C::NC; e += BT) B.rs[e] = Rk[C::NC + e];
            for (int e = tid; e < R::LEN_SWEEP - SKIP1; e += BT) B.rs[SKIP0 - C::NC + e] = Rk[SKIP1 + e];
        }
    };
    rec_load(N - 2);
    game_sync();
    ALG_PROF_DECL
    // ------------------------------------------------------------------ backward sweep
    for (int k = N - 2; k >= 0; k--) {
        const double* Rl = B.rs - C::NC;                                     // record offsets >= NC address the staged copy ...
        const double* Rt = B.rs - C::NC - (SKIP1 - SKIP0);                   // ... [R^ | ru | rd] behind the blocks a TIGHT layout does not stage
        const double* Rq = TIGHT ? recs + (size_t)k * R::LEN : Rl;           // [RQ | rx]: read once, by the Q-add
        const double* coefk = B.cf;                                        // step k's block -- after the landing point below
        const double* coefn = B.cf;                                        // A_{k+1}: what the buffer holds during the value recursion
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        // ---- value recursion: [P_i | s_i] <- A_{k+1}' ([P_i | s_i] [[F f],[0 1]])
        if (k < N - 2) {
            for (int i = 0; i < P; i++) {
                if (IBR && i != ip) continue;
                double* Pi = &B.Pm[i * n * LDP];
                // work item = column tile; its TR row tiles advance together (independent accumulator chains, one B operand load
                // per k-block for all of them)
                for (int tc = wv; tc < TC; tc += NWV) {
                    const int bcol = 16 * tc + lrow;
                    double4_t acc[TR];
#pragma unroll
                    for (int tr = 0; tr < TR; tr++) acc[tr] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < KBN; kb++) {
                        const int kk = 4 * kb + lq;
                        const bool kok = kk <= n, bok = kok && bcol <= n;
                        const double bv = B.Fx[bok ? kk * LDP + bcol : 0];
#pragma unroll
                        for (int tr = 0; tr < TR; tr++) {
                            const int arow = 16 * tr + lrow;
                            const bool aok = kok && arow < n;
                            const double av = Pi[aok ? arow * LDP + kk : 0];
                            acc[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aok ? av : 0.0, bok ? bv : 0.0, acc[tr], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int tr = 0; tr < TR; tr++)
#pragma unroll
                        for (int r4 = 0; r4 < 4; r4++) { const int row = 16 * tr + lq + 4 * r4; if (row < n && bcol <= n) B.Tm[row * LDP + bcol] = acc[tr][r4]; }
                }
                game_sync();
                ALG_PROF(0)
                if constexpr (C::QUAD) {
                    // block-diagonal A' (dense 12 x 12 block per player j) as MFMA products too: rows (., j) of the result =
                    // A_j' x rows (., j) of the product; work item = (block j, column tile)
                    for (int t = wv; t < P * TC; t += NWV) {
                        const int j = t / TC, tc = t % TC, bcol = 16 * tc + lrow;
                        const bool aok = lrow < 12, bok = bcol <= n;
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kb = 0; kb < 3; kb++) {
                            const int a2 = 4 * kb + lq;                    // A'[a][a2] = A_j[a2][a]
                            const double av = coefn[j * C::QS + C::QA + a2 * 12 + (aok ? lrow : 0)];
                            const double bv = B.Tm[(a2 * P + j) * LDP + (bok ? bcol : 0)];
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aok ? av : 0.0, bok ? bv : 0.0, acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int r4 = 0; r4 < 3; r4++) { const int a = lq + 4 * r4; if (bok) Pi[(a * P + j) * LDP + bcol] = acc[r4]; }
                    }
                } else {
                    flat_loop<FU>(tid, BT, n * LDP, [&](int e) {
                        const int r = e / LDP, c = e % LDP;
                        return AT_vec<C>(coefn, dt, [&](int rr) { return B.Tm[rr * LDP + c]; }, r);
                    }, [&](int e, double v) { Pi[e] = v; });
                }
                game_sync();
                ALG_PROF(1)
            }
        }
        // ---- landing point of step k's record (requested at the tail of step k + 1): the recursion above was the last reader of
        // step k + 1's coefficients
        rec_store(k);
        game_sync();
        // ---- + [Q^_i | rx_i]: diagonal, position block, column n
        for (int e = tid; e < P * n; e += BT) {
            const int i = e / n, r = e % n;
            if (IBR && i != ip) continue;
            double* row = &B.Pm[i * n * LDP + r * LDP];
            double qd = reg + ((r % P == i) ? w * Qd[i * C::ni + r / P] : 0.0);
            if constexpr (C::EXT) qd += Rq[R::RQ + e];
            row[r] += qd;
            row[n] += Rq[R::RX + e];
            if (C::POS && r < C::PD * P) {
                for (int c = 0; c < C::PD * P; c++) row[c] += pairblock<C>(Rl + R::HH, i, r, c);
            }
        }
        game_sync();
        ALG_PROF(2)
        // ---- V[c][:] = B[:,c]' P_i(c),  y_i = P_i rd + s_i
        flat_loop<FU>(tid, BT, m * n, [&](int e) {
            const int c = e / n, col = e % n; const double* Pi = &B.Pm[(c % P) * n * LDP];
            return BT_vec<C>(coefk, dt, [&](int rr) { return Pi[rr * LDP + col]; }, c);
        }, [&](int e, double v) { B.V()[e] = v; });
        flat_loop<FU>(tid, BT, P * n, [&](int e) {
            const double* Pr = &B.Pm[e * LDP];
            double a = Pr[n];
            for (int c = 0; c < n; c++) a += Pr[c] * Rt[R::RD + c];
            return a;
        }, [&](int e, double v) { B.y()[e] = v; });
        game_sync();
        ALG_PROF(3)
        // ---- [ W | V A_k | g ],  W = diag(R^) + V B,  g_c = ru_c + B[:,c]' y_i(c): three uniform loops (no divergent entry kinds)
        auto ibr_mask = [&](int c, int t, double v) {
            if (IBR) {
                if (c % P != ip) v = (t == c) ? 1.0 : 0.0;                   // unit row: du_c = 0
                else if (t < m && t % P != ip) v = 0.0;                      // fixed controls of the other players
            }
            return v;
        };
        flat_loop<FU>(tid, BT, m * m, [&](int e) {
            const int c = e / m, t = e % m; const double* Vc = &B.V()[c * n];
            return ibr_mask(c, t, BT_vec<C>(coefk, dt, [&](int rr) { return Vc[rr]; }, t) + (t == c ? Rt[R::RHAT + c] : 0.0));
        }, [&](int e, double v) { B.sv.Wm[(e / m) * WC + e % m] = v; });
        flat_loop<FU>(tid, BT, m * n, [&](int e) {
            const int c = e / n, col = e % n; const double* Vc = &B.V()[c * n];
            return ibr_mask(c, m + col, (k >= 1) ? AT_vec<C>(coefk, dt, [&](int rr) { return Vc[rr]; }, col) : 0.0);    // dx_1 = 0: A_0 never acts
        }, [&](int e, double v) { B.sv.Wm[(e / n) * WC + m + e % n] = v; });
        flat_loop<1>(tid, BT, m, [&](int c) {
            const double* yi = &B.y()[(c % P) * n];
            return ibr_mask(c, m + n, Rt[R::RU + c] + BT_vec<C>(coefk, dt, [&](int rr) { return yi[rr]; }, c));
        }, [&](int c, double v) { B.sv.Wm[c * WC + m + n] = v; });
        game_sync();
        ALG_PROF(4)
        // ---- partially pivoted Gauss-Jordan (pivot rule and row operations of the tile path's gj_solve_cols).  m <= 8: wavefront 0
        // alone, columns in registers, the pivot column by v_readlane (no LDS traffic, no barrier).  m > 8 (the 2 m scalar registers
        // of a readlane broadcast inside the fully unrolled elimination push those kernels into scratch): all threads, columns in
        // registers, the pivot column through LDS, one barrier per pivot.
        if constexpr (m <= 8) {
            constexpr int XC = (WC + WAVE - 1) / WAVE;
            static_assert(m < WAVE && XC <= 2, "control system of the dense direction: at most 128 columns");
            int sg = 0;
            if (wv == 0) {
                double col[XC][m];
#pragma unroll
                for (int x = 0; x < XC; x++) {
                    const int t = lane + x * WAVE;
#pragma unroll
                    for (int r = 0; r < m; r++) col[x][r] = B.sv.Wm[r * WC + (t < WC ? t : 0)];
                }
                sg = gj_solve_cols_x<m, XC>(col);
                // solved right-hand-side columns back to LDS (the gains and the closed-loop rows read them)
#pragma unroll
                for (int x = 0; x < XC; x++) {
                    const int t = lane + x * WAVE;
                    if (t >= m && t < WC) {
#pragma unroll
                        for (int r = 0; r < m; r++) B.sv.Wm[r * WC + t] = col[x][r];
                    }
                }
                if (lane == 0) L.red[7] = (double)sg;
            }
            game_sync();
            if constexpr (NWV > 1) sg = (int)L.red[7];
            sing |= __builtin_amdgcn_readfirstlane(sg);
        } else {
            constexpr int XC = (WC + BT - 1) / BT;
            double col[XC][m];
#pragma unroll
            for (int x = 0; x < XC; x++) {
                const int t = tid + x * BT;
#pragma unroll
                for (int r = 0; r < m; r++) col[x][r] = B.sv.Wm[r * WC + (t < WC ? t : 0)];
            }
#pragma unroll
            for (int c = 0; c < m; c++) {
                if (tid == c) {
#pragma unroll
                    for (int r = 0; r < m; r++) B.pcol(c & 1)[r] = col[0][r];
                }
                game_sync();
                double pc[m];
#pragma unroll
                for (int r = 0; r < m; r++) pc[r] = B.pcol(c & 1)[r];
                double best = fabs(pc[c]); int piv = c;
#pragma unroll
                for (int r = c + 1; r < m; r++) { const double v = fabs(pc[r]); if (v > best) { best = v; piv = r; } }
                if (!(best > 0.0) || !isfinite(best)) sing = 1;
                piv = __builtin_amdgcn_readfirstlane(piv);
                if (piv != c) {
#pragma unroll
                    for (int r = c + 1; r < m; r++) {
                        if (piv == r) {
                            double t2 = pc[c]; pc[c] = pc[r]; pc[r] = t2;
#pragma unroll
                            for (int x = 0; x < XC; x++) { t2 = col[x][c]; col[x][c] = col[x][r]; col[x][r] = t2; }
                        }
                    }
                }
                const double rpiv = fast_rcp(pc[c]);
#pragma unroll
                for (int x = 0; x < XC; x++) {
                    const double prow = col[x][c] * rpiv;
#pragma unroll
                    for (int r = 0; r < m; r++) if (r != c) col[x][r] -= pc[r] * prow;
                    col[x][c] = prow;
                }
            }
            // solved right-hand-side columns back to LDS (the gains and the closed-loop rows read them)
#pragma unroll
            for (int x = 0; x < XC; x++) {
                const int t = tid + x * BT;
                if (t >= m && t < WC) {
#pragma unroll
                    for (int r = 0; r < m; r++) B.sv.Wm[r * WC + t] = col[x][r];
                }
            }
            game_sync();
        }
        ALG_PROF(5)
        // ---- [F | f] = [A_k | rd] + B [K | kappa] ; K = -Y -> HBM (column-major m x (n+1))
        if (k > 0) {
            flat_loop<FU>(tid, BT, n * LDP, [&](int e) {
                const int r = e / LDP, col = e % LDP;
                const double base = col < n ? A_entry<C>(coefk, dt, r, col) : Rt[R::RD + r];
                return base + B_vec<C>(coefk, dt, [&](int c2) { return -B.sv.Wm[c2 * WC + m + col]; }, r);
            }, [&](int e, double v) { B.Fx[e] = v; });
        }
        for (int e = tid; e < NK; e += BT) { const int col = e / m, c = e % m; kg[(size_t)k * NK + e] = -B.sv.Wm[c * WC + m + col]; }   // gains out
        if (k > 0) rec_load(k - 1);                                          // then request step k - 1 (landed after its value recursion)
        game_sync();
        ALG_PROF(6)
    }
    if (sing) return ALG_STATUS_SINGULAR;                  // uniform: every thread saw the same pivots
    // ------------------------------------------------------------------ forward sweep: dx, du
    double* __restrict__ dz = G.z(2);
    auto& F = L.fw;
    for (int e = tid; e < n; e += BT) { F.dx[e] = 0.0; dz[e] = 0.0; }
    // forward sweep slice of a record: [coef | rd], one step ahead through registers like above
    constexpr int FSL = C::NC + n, FPT = (FSL + BT - 1) / BT;
    double fpre[PREF ? FPT : 1];
    int fw_req = 0;                                          // step whose slice is in flight / due at the landing point
    auto fw_load = [&](int kk) {
        fw_req = kk;
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < FPT; q++) { const int e = tid + q * BT; fpre[q] = recs[(size_t)kk * R::LEN + (e < C::NC ? e : (e < FSL ? R::RD + (e - C::NC) : 0))]; }
        }
    };
    auto fw_store = [&]() {
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < FPT; q++) {
                const int e = tid + q * BT;
                if (e < C::NC) F.cf[0][e] = fpre[q];
                else if (e < FSL) F.rs[0][R::RD - C::NC + (e - C::NC)] = fpre[q];
            }
        } else {
            const double* Rk = recs + (size_t)fw_req * R::LEN;
            for (int e = tid; e < C::NC; e += BT) F.cf[0][e] = Rk[e];
            for (int e = tid; e < n; e += BT) F.rs[0][R::RD - C::NC + e] = Rk[R::RD + e];
        }
    };
    fw_load(0); fw_store();
    if (1 < N - 1) fw_load(1);
    game_sync();
    double pl1 = 0.0; int bad = 0;
    constexpr int XPT = (n + BT - 1) / BT;
    for (int k = 0; k < N - 1; k++) {
        const double* coefk = F.cf[0];
        const double* Kg = kg + (size_t)k * NK;
        constexpr int UPT = (m + BT - 1) / BT;
        double duv[UPT];
#pragma unroll
        for (int q0 = 0; q0 < UPT; q0++) {
            const int c = tid + q0 * BT; duv[q0] = 0.0;
            if (c < m) {
                double a = Kg[n * m + c];
                for (int q = 0; q < n; q++) a += Kg[q * m + c] * F.dx[q];
                F.du[c] = a; duv[q0] = a;
                pl1 += fabs(a); bad |= !isfinite(a);
            }
        }
        game_sync();
        double nx[XPT];
#pragma unroll
        for (int q = 0; q < XPT; q++) {
            const int r = tid + q * BT; nx[q] = 0.0;
            if (r < n) nx[q] = (A_vec<C>(coefk, dt, [&](int rr) { return F.dx[rr]; }, r) + B_vec<C>(coefk, dt, [&](int cc) { return F.du[cc]; }, r)) + F.rs[0][R::RD - C::NC + r];
        }
        game_sync();
#pragma unroll
        for (int q = 0; q < XPT; q++) {
            const int r = tid + q * BT;
            if (r < n) { F.dx[r] = nx[q]; pl1 += fabs(nx[q]); bad |= !isfinite(nx[q]); }
        }
        if (k + 1 < N - 1) fw_store();                                       // (1) land step k + 1's slice
#pragma unroll
        for (int q0 = 0; q0 < UPT; q0++) { const int c = tid + q0 * BT; if (c < m) dz[n + hu<C>(k, 0) + uoff<C>(c)] = duv[q0]; }   // (2) results out
#pragma unroll
        for (int q = 0; q < XPT; q++) { const int r = tid + q * BT; if (r < n) dz[n + hx<C>(k) + r] = nx[q]; }
        if (k + 2 < N - 1) fw_load(k + 2);                                   // (3) request step k + 2
        game_sync();
    }
    ALG_PROF(7)
    // ------------------------------------------------------------------ costate sweep:
    //   dlambda_{i,k} = Q^_{i,k+1} dx_{k+1} + A_{k+1}' dlambda_{i,k+1} + rx_{i,k+1}
    // costate slice: this step's coefficient block (applied as A_{k+1}' one step later) and the NEXT (earlier) step's
    // [Hh | Hd | RQ | rx], both requested at the top of a step and parked in LDS at its end
    // plus dx_k (the direction's state block the step multiplies with): cs_load(kk) requests the coefficients of step kk + 1 and
    // [Hh | Hd | RQ | rx], dx of step kk; cs_store(kk) parks them in the slots step kk reads
    constexpr int CSL = R::LEN_COSTATE, CPT = (CSL + n + BT - 1) / BT;
    double cpre[PREF ? CPT : 1];
    auto cs_load = [&](int kk) {
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < CPT; q++) {
                const int e = tid + q * BT;
                const double* src = e < C::NC ? recs + (size_t)(kk + 1 < N - 1 ? kk + 1 : kk) * R::LEN + e
                                  : e < CSL ? recs + (size_t)kk * R::LEN + e
                                  : dz + n + hx<C>(kk) + (e < CSL + n ? e - CSL : 0);
                cpre[q] = *src;
            }
        }
    };
    auto cs_store = [&](int kk) {
        if constexpr (PREF) {
#pragma unroll
            for (int q = 0; q < CPT; q++) {
                const int e = tid + q * BT;
                if (e < C::NC) F.cf[(kk + 1) & 1][e] = cpre[q];
                else if (e < CSL) F.rs[kk & 1][e - C::NC] = cpre[q];
                else if (e < CSL + n) F.dxb[kk & 1][e - CSL] = cpre[q];
            }
        } else {
            const double* Rn = recs + (size_t)(kk + 1 < N - 1 ? kk + 1 : kk) * R::LEN; const double* Rk = recs + (size_t)kk * R::LEN;
            for (int e = tid; e < C::NC; e += BT) F.cf[(kk + 1) & 1][e] = Rn[e];
            for (int e = tid; e < CSL - C::NC; e += BT) F.rs[kk & 1][e] = Rk[C::NC + e];
            for (int e = tid; e < n; e += BT) F.dxb[kk & 1][e] = dz[n + hx<C>(kk) + e];
        }
    };
    cs_load(N - 2); cs_store(N - 2);
    if (N - 3 >= 0) cs_load(N - 3);
    game_sync();
    for (int k = N - 2; k >= 0; k--) {
        const int cur = k & 1;
        const double* Rl = F.rs[cur] - C::NC;
        const double* coefn = F.cf[cur ^ 1];
        const double* dxk = F.dxb[cur];
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        constexpr int LPT = (P * n + BT - 1) / BT;
        double lv[LPT];
#pragma unroll
        for (int q0 = 0; q0 < LPT; q0++) {
            const int e = tid + q0 * BT; lv[q0] = 0.0;
            if (e < P * n) {
                const int i = e / n, r = e % n;
                double acc = 0.0;
                if (!IBR || i == ip) {
                    double qd = reg + ((r % P == i) ? w * Qd[i * C::ni + r / P] : 0.0);
                    if constexpr (C::EXT) qd += Rl[R::RQ + e];
                    acc = Rl[R::RX + e] + qd * dxk[r];
                    if (C::POS && r < C::PD * P) {
                        for (int c = 0; c < C::PD * P; c++) acc += pairblock<C>(Rl + R::HH, i, r, c) * dxk[c];
                    }
                    if (k < N - 2) { const double* dli = &F.dl[cur ^ 1][i * n]; acc += AT_vec<C>(coefn, dt, [&](int rr) { return dli[rr]; }, r); }
                }
                F.dl[cur][e] = acc; lv[q0] = acc; bad |= !isfinite(acc);
            }
        }
        if (k > 0) cs_store(k - 1);                                          // (1) land step k - 1's slices (other slots than the ones read above)
#pragma unroll
        for (int q0 = 0; q0 < LPT; q0++) { const int e = tid + q0 * BT; if (e < P * n) dz[n + hl<C>(k, 0) + e] = lv[q0]; }   // (2) results out
        if (k > 1) cs_load(k - 2);                                           // (3) request step k - 2
        game_sync();
    }
    ALG_PROF(8)
    ALG_PROF_FLUSH
    pl1 = wave_sum(pl1); bad = wave_or(bad);
    if constexpr (NWV > 1) {
        if (lane == 0) { L.red[wv] = pl1; L.red[4 + wv] = (double)bad; }
        game_sync();
        pl1 = 0.0; bad = 0;
#pragma unroll
        for (int q = 0; q < NWV; q++) { pl1 += L.red[q]; bad |= (int)L.red[4 + q]; }
    }
    if (primal_l1) *primal_l1 = pl1;
    return __builtin_amdgcn_readfirstlane(bad) ? ALG_STATUS_SINGULAR : ALG_STATUS_OK;
}

template <class C, bool IBR>
__device__ int newton_direction_tile(CPR pr0, const Game& G0, DirLds<C>& L, double reg, int ip, double* primal_l1);
template <class C, bool IBR = false>
__device__ __forceinline__ int newton_direction(CPR pr0, const Game& G0, DirLds<C>& L, double reg, int ip = -1, double* primal_l1 = nullptr) {
    if constexpr (C::DENSE) return newton_direction_dense<C, IBR>(pr0, G0, L, reg, ip, primal_l1);
    else return newton_direction_tile<C, IBR>(pr0, G0, L, reg, ip, primal_l1);
}
// Row c of the augmented control system [W | V A_k] from row c of V = B[:,c]' P_{i(c)} (one 16-lane row of lanes per control, col = the
// lane's column): the SYSROW form of the backward sweep's V phase (described there).  Reads P_{c % P} and the step record, writes row c of L.bw.V.
template <class C>
__device__ __forceinline__ void v_sysrow(DirLds<C>& L, const double* Rc, int k, double dt, int c, int col) {
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, VW = DirLds<C>::VW;
    using R = Rec<C>;
    const double* coefk = Rc + R::COEF;
    const bool cok = c < m; const int cq = cok ? c : 0, colr = col < n ? col : n - 1;
    const double* Pi = &L.bw.Pm[(cq % P) * n * LDP];
    const double vcol = BT_vec<C>(coefk, dt, [&](int rr) { return Pi[rr * LDP + colr]; }, cq);
    const double rh = (col == cq) ? Rc[R::RHAT + cq] : 0.0;
    double ea, eb;
    if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        const double below = row_shift<0x110 + m>(vcol), above = row_shift<0x100 + m>(vcol);
        ea = (k >= 1) ? (col >= m ? fma(dt, below, vcol) : vcol) : 0.0;
        eb = (0.5 * dt * dt) * vcol; eb = fma(dt, above, eb);
    } else {
        // unicycle (col_pattern): column idx = kind P + i of B touches rows i, P + i (coefficients) and (2 + kind) P + i (dt); column
        // (2 + kind') P + i of A touches rows i, P + i besides its own
        const double d1 = row_shift<0x110 + P>(vcol), d2 = row_shift<0x110 + 2 * P>(vcol), d3 = row_shift<0x110 + 3 * P>(vcol);
        const double u1 = row_shift<0x100 + P>(vcol), u2 = row_shift<0x100 + 2 * P>(vcol);
        const int blk = colr / P, pi = colr % P;
        const double ca = coefk[(blk >= 2 ? blk - 2 : 0) * P + pi], cb = coefk[(blk >= 2 ? blk : 2) * P + pi];
        double va = vcol;
        if (blk >= 2) { va = fma(ca, blk == 3 ? d3 : d2, va); va = fma(cb, blk == 3 ? d2 : d1, va); }
        ea = (k >= 1) ? va : 0.0;
        const int kind = blk & 1;                                  // col < m: blk = kind
        const double wa = 0.5 * dt * coefk[kind * P + pi], wb = 0.5 * dt * coefk[(2 + kind) * P + pi];
        eb = wa * (kind ? d1 : vcol); eb = fma(wb, kind ? vcol : u1, eb); eb = fma(dt, u2, eb);
    }
    eb = fma(1.0, rh, eb);
    if (cok && col < n) L.bw.V[c * VW + m + col] = ea;
    if (cok && col < m) L.bw.V[c * VW + col] = eb;
}

// Team of two, per-player part of a backward step that follows the Q-add, for the players first and first + 2:
//   s_i <- rx_i + A_{k+1}' t_i                     (n == 16: s_i does not ride in the MFMA tile)
//   y_i = P_i rd + s_i, g_c = ru_c + B[:,c]' y_i   with TWO lanes per row of P_i (lanes 0..31: columns 0..7, lanes 32..63: columns 8..15; the
//   halves meet through v_permlane32_swap), i.e. FMA chains of eight instead of sixteen -- the sums associate differently from the
//   one-wavefront kernel's (rounding-level differences, like the team's norms).
// s_i <- rx_i + A_{k+1}' t_i for the players first and first + 2 (n == 16: s_i lives in column n of the players' LDS rows, outside the MFMA tiles)
template <class C>
__device__ __forceinline__ void s_half(DirLds<C>& L, const double* Rc, double dt, bool rec, int first, int lane) {
    constexpr int n = C::n, LDP = DirLds<C>::LDP;
    using R = Rec<C>;
    if (lane < 2 * n) {
        const int i = first + 2 * (lane / n), r = lane % n; const double* ti = &L.bw.t[i * n];
        double v = Rc[R::RX + i * n + r];
        if (rec) v += AT_vec<C>(L.coefn, dt, [&](int rr) { return ti[rr]; }, r);
        L.bw.Pm[i * n * LDP + r * LDP + n] = v;
    }
}
template <class C>
__device__ __forceinline__ void player_tail_half(DirLds<C>& L, const double* Rc, double dt, int k, int N, int first, int lane) {
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, VW = DirLds<C>::VW;
    static_assert(n == 16 && P == 4 && C::MODEL == ALG_MODEL_UNICYCLE, "lane layout of the team-of-two tail (4-player unicycle)");
    using R = Rec<C>;
    const double* coefk = Rc + R::COEF;
    if constexpr (!DirLds<C>::AUGS) {
        s_half<C>(L, Rc, dt, k < N - 2, first, lane);
        sweep_sync<C>();
    }
    const int h = lane >> 5, yp = first + 2 * ((lane >> 4) & 1), yr = lane & 15;
    const double* Pr = &L.bw.Pm[yp * n * LDP + yr * LDP];
    const double rdl = Rc[R::RD + ((yr + 8 * h) & 15)];          // lane j of an upper-half row holds rd[j + 8]
    const double* Ph = Pr + 8 * h;
    double a = h ? 0.0 : Pr[n];
    rowdot_dpp_g<8, (rowdot_group_v<C> < 8 ? rowdot_group_v<C> : 8)>(a, rdl, [&](int c) { return Ph[c]; });
    a += xchg32(a, lane < 32);
    const double dn = row_shift<0x110 + P>(a), up = row_shift<0x100 + P>(a), up2 = row_shift<0x100 + 2 * P>(a);
    const int kind = yr < m ? yr / P : 0;
    const double vi = kind ? dn : a, vpi = kind ? a : up;
    const double gb = 0.5 * dt * (coefk[kind * P + yp] * vi + coefk[(2 + kind) * P + yp] * vpi) + dt * up2;
    if (h == 0 && yr < m && yr % P == yp) L.bw.V[yr * VW + m + n] = Rc[R::RU + yr] + gb;
}

// Team of two: t_i = P_i f + s_i (n == 16) for the players first and first + 2 from the value functions of the step before, two lanes per
// row like player_tail_half; runs before the same wavefront's value recursion overwrites those P_i.
template <class C>
__device__ __forceinline__ void t_half(DirLds<C>& L, int first, int lane) {
    constexpr int n = C::n, LDP = DirLds<C>::LDP;
    if constexpr (!DirLds<C>::AUGS) {
        const int h = lane >> 5, i = first + 2 * ((lane >> 4) & 1), r = lane & 15;
        const double* Pr = &L.bw.Pm[i * n * LDP + r * LDP];
        double a = h ? 0.0 : Pr[n];
#pragma unroll
        for (int c = 0; c < 8; c++) a += Pr[c + 8 * h] * L.bw.fv[c + 8 * h];
        a += xchg32(a, lane < 32);
        if (h == 0) L.bw.t[i * n + r] = a;
    }
}

// (measured and not kept, round 4: t_i / s_i of a wavefront's players held in registers across the value recursion -- two LDS round trips and
// a fence less per step, +0.3 % at C3; and the same work issued behind the MFMA chains, -0.7 %: profiles/r04_ab_help5_c3.txt, r04_ab_help6_c3.txt)

// Team of two (round 4): splitting the whole backward step over the two wavefronts costs more in barriers than it gains (measured on C3
// at 1024 games: 2.30 against 2.35 M/s), but the value recursion alone -- a quarter of a step, independent per player, LDS in / LDS out
// -- is worth two LDS-only barriers: wavefront 1 takes the odd players' MFMA chains of every step and does nothing else in the direction.
template <class C, bool IBR>
inline constexpr bool help2_v = C::NW == 2 && !IBR && !C::DENSE && C::WPE == 2 && C::MODEL == ALG_MODEL_UNICYCLE && C::P == 4;
// [P_i | s_i] <- A' ([P_i | s_i] [[F f],[0 1]]) for the players first, first + 2, ...: operands of all of them read first, their MFMA
// chains interleaved, results written back to the players' own LDS row blocks (nobody else touches those between the two barriers).
template <class C>
__device__ __forceinline__ void value_recursion_half(DirLds<C>& L, int first, int lrow, int lq, double dt) {
    constexpr int n = C::n, P = C::P, PH = P / 2, LDP = DirLds<C>::LDP, KB1 = DirLds<C>::KB1;
    constexpr bool AUGS = DirLds<C>::AUGS;
    constexpr int oPm = (int)(offsetof(typename DirLds<C>::Bwd, Pm) / 8), oPad = (int)(offsetof(typename DirLds<C>::Bwd, pad) / 8);
    double* const bwb = reinterpret_cast<double*>(&L.bw);
    const bool colP = lrow < n;
    double bF[KB1], pv[PH][KB1];
#pragma unroll
    for (int kb = 0; kb < KB1; kb++) bF[kb] = L.bw.Fx[(4 * kb + lq) * 16 + lrow];
#pragma unroll
    for (int ii = 0; ii < PH; ii++)
#pragma unroll
        for (int kb = 0; kb < KB1; kb++) {
            const double v = L.bw.Pm[(first + 2 * ii) * n * LDP + (colP ? lrow : 0) * LDP + 4 * kb + lq];
            pv[ii][kb] = colP ? v : 0.0;
        }
    double4_t c1[PH], c2[PH];
#pragma unroll
    for (int ii = 0; ii < PH; ii++) c1[ii] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < KB1; kb++)
#pragma unroll
        for (int ii = 0; ii < PH; ii++) c1[ii] = __builtin_amdgcn_mfma_f64_16x16x4f64(pv[ii][kb], bF[kb], c1[ii], 0, 0, 0);
#pragma unroll
    for (int ii = 0; ii < PH; ii++) {
        if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) c2[ii] = di_AT_tile<C>(c1[ii], dt, lq);
        else c2[ii] = p4_AT_tile<C>(c1[ii], L.coefn, lq);
    }
    sweep_sync<C>();
#pragma unroll
    for (int ii = 0; ii < PH; ii++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const int row = lq + 4 * r4;
            const int slot = (row < n && lrow < n + (AUGS ? 1 : 0)) ? oPm + (first + 2 * ii) * n * LDP + row * LDP + lrow : oPad;
            bwb[slot] = c2[ii][r4];
        }
}

// Split value recursion of one backward step (round 5; described at SPLITF in newton_direction_tile): for every player
//   [P_i | s_i] <- A_{k+1}' ( [P_i A_k | y_i] + (P_i B_k) [K | kappa] )
// with the accumulator tile started at [P_i A_k | y_i] and ceil(m / 4) f64 MFMAs per player for the product.  TEAM: player i on wavefront i.
template <class C, bool TEAM>
__device__ __forceinline__ void split_value_recursion(DirLds<C>& L, double dt, int lrow, int lq, int tw) {
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, KBS = (m + 3) / 4;
    double kt[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) kt[kb] = L.bw.Fx[(4 * kb + lq) * 16 + lrow];       // [K | kappa] of step k + 1, rows >= m zero
    // the products and the write-back, given a model's operand / write-back helpers
    auto run = [&](auto&& operands, auto&& write_back) {
        if constexpr (TEAM) {
            // team: player i on wavefront i (the same operations on the same numbers as below: bit-identical to one wavefront per game)
            static_assert(!TEAM || P <= C::NW, "one player per wavefront of the team");
            if (tw < P) {
                double4_t acc; double pb[KBS];
                operands(tw, acc, pb);
    #pragma unroll
                for (int kb = 0; kb < KBS; kb++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[kb], kt[kb], acc, 0, 0, 0);
                sweep_sync<C>();
                write_back(tw, acc);
            }
        } else {
            // the players' chains are independent: all operands first, the products interleaved in the matrix pipe, then the write-backs
            double4_t acc[P];
            double pb[P][KBS];
    #pragma unroll
            for (int i = 0; i < P; i++) operands(i, acc[i], pb[i]);
    #pragma unroll
            for (int kb = 0; kb < KBS; kb++)
    #pragma unroll
                for (int i = 0; i < P; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[i][kb], kt[kb], acc[i], 0, 0, 0);
            sweep_sync<C>();                                           // every read of [P_i | y_i] is done
    #pragma unroll
            for (int i = 0; i < P; i++) write_back(i, acc[i]);
        }
    };
    if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        // Row placement in the MFMA tile: the product's row at tile position (lane group l, register r4) is whatever row of P_i the
        // A operand's lane 4 r4 + l feeds, so the rows are PLACED such that A' costs no lane exchange afterwards: A' X adds dt x row
        // r to row r + m, and the pair (r, r + m) sits in ONE lane group, r = 4 j + l in register 2 j and r + m in register 2 j + 1
        // (the natural order r = l + 4 r4 has row r - m in lane ^ 32 when m = 2 mod 4: four v_permlane32_swap + selects per tile).
        // Lane roles: the tile entry (row, column lrow) of P_i A takes dt x column lrow - m of the same row (columns m..n-1); the
        // operand entry (row, k = 4 kb + lq) of P_i B is dt^2/2 P[row][k] + dt P[row][k + m] (entries with k >= m meet zero rows of
        // [K | kappa]: whatever finite value the two loads return there is multiplied by zero).
        constexpr int NR = 2 * ((m + 3) / 4);                           // accumulator registers in use
        static_assert(NR <= 4, "row pairs of the split recursion");
        // (rows as affine functions of the lane coordinates, so that every LDS address below is one of four lane-dependent bases plus
        // an immediate: register r4 of lane group l holds row l + RC(r4); lane groups whose row does not exist -- l >= m - 4 j in the
        // last pair of registers -- read rows of the next block, harmlessly, and do not write)
        auto rconst = [](int r4) { return 4 * (r4 >> 1) + ((r4 & 1) ? m : 0); };
        const int opl = lrow & 3, opr4 = lrow >> 2;
        const int oprow = (opr4 < NR && 4 * (opr4 >> 1) + opl < m) ? opl + 4 * (opr4 >> 1) + ((opr4 & 1) ? m : 0) : 0;      // the row this lane feeds as A operand
        const bool shc = lrow >= m && lrow < n;
        const int lsh = shc ? lrow - m : lrow;
        const double dtc = shc ? dt : 0.0, hdt2 = 0.5 * dt * dt;
        const double* const tbase = &L.bw.Pm[lq * LDP + lrow], * const sbase = &L.bw.Pm[lq * LDP + lsh], * const obase = &L.bw.Pm[oprow * LDP + lq];
        double* const wbase = &L.bw.Pm[lq * LDP + lrow];
        auto operands = [&](int i, double4_t& acc, double (&pb)[KBS]) {
    #pragma unroll
            for (int r4 = 0; r4 < 4; r4++) acc[r4] = r4 < NR ? fma(dtc, sbase[i * n * LDP + rconst(r4) * LDP], tbase[i * n * LDP + rconst(r4) * LDP]) : 0.0;
    #pragma unroll
            for (int kb = 0; kb < KBS; kb++) pb[kb] = fma(-dt, obase[i * n * LDP + 4 * kb + m], -hdt2 * obase[i * n * LDP + 4 * kb]);      // -(P_i B): the B operand holds -[K | kappa]
        };
        auto write_back = [&](int i, double4_t acc) {
    #pragma unroll
            for (int j = 0; 2 * j + 1 < NR; j++) acc[2 * j + 1] = fma(dt, acc[2 * j], acc[2 * j + 1]);      // A': row r + m += dt x row r
            if (lrow < n + 1) {
    #pragma unroll
                for (int r4 = 0; r4 < NR; r4++) {
                    if (4 * (r4 >> 1) + 4 <= m) wbase[i * n * LDP + rconst(r4) * LDP] = acc[r4];             // rows of every lane group
                }
                if constexpr ((m & 3) != 0) {
                    if (lq < (m & 3)) {
    #pragma unroll
                        for (int r4 = NR - 2; r4 < NR; r4++) wbase[i * n * LDP + rconst(r4) * LDP] = acc[r4];  // last pair: lane groups l < m mod 4
                    }
                }
            }
        };
        run(operands, write_back);
    } else {
        // 3-player unicycle: state rows a P + i (a = 0: x_i, 1: y_i, 2: theta_i, 3: v_i).  Tile position (lane group l = i, register
        // a) holds row a P + i: the A operand's lane 4 a + i feeds that row, A' then touches one lane group only.
        //   (P A)[r][c]  = P[r][c] + ca P[r][c - 2P or c - 3P] + cb P[r][c - P or c - 2P]   (c a heading / speed column, else P[r][c])
        //   (P B)[r][k]  = dt/2 (cx P[r][i'] + cy P[r][P + i']) + dt P[r][(2 + kind) P + i'],   k = kind P + i'
        //   (A' X)[theta_i] = X[theta_i] + c0 X[x_i] + c2 X[y_i],   (A' X)[v_i] = X[v_i] + c1 X[x_i] + c3 X[y_i]
        // with the step's coefficients c0 = A[x][theta], c1 = A[x][v], c2 = A[y][theta], c3 = A[y][v] of the player (L.coefn: step k + 1).
        const int ti = lq < P ? lq : 0;                                     // player of this lane group's rows
        const int oa = lrow >> 2, oi = lrow & 3;
        const int oprow = (oi < P) ? oa * P + oi : 0;                          // the row this lane feeds as A operand
        const int cblk = lrow / P, ci = lrow % P;                          // column block / player of tile column lrow
        const bool con = lrow >= 2 * P && lrow < n;
        const int colA = con ? ci : lrow, colB = con ? P + ci : lrow;
        const double ca = con ? L.coefn[(cblk - 2) * P + ci] : 0.0, cb = con ? L.coefn[cblk * P + ci] : 0.0;
        const double* const tbase = &L.bw.Pm[ti * LDP + lrow], * const abase = &L.bw.Pm[ti * LDP + colA], * const bbase = &L.bw.Pm[ti * LDP + colB];
        double* const wbase = &L.bw.Pm[ti * LDP + lrow];
        const double* ob[KBS][3]; double cx[KBS], cy[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) {
            const int kk = 4 * kb + lq; const bool kok = kk < m; const int ki = kok ? kk % P : 0, kind = kok ? kk / P : 0;
            ob[kb][0] = &L.bw.Pm[oprow * LDP + ki]; ob[kb][1] = &L.bw.Pm[oprow * LDP + P + ki]; ob[kb][2] = &L.bw.Pm[oprow * LDP + (2 + kind) * P + ki];
            cx[kb] = kok ? L.coefn[kind * P + ki] : 0.0; cy[kb] = kok ? L.coefn[(2 + kind) * P + ki] : 0.0;
        }
        const double c0 = L.coefn[0 * P + ti], c1 = L.coefn[1 * P + ti], c2 = L.coefn[2 * P + ti], c3 = L.coefn[3 * P + ti];
        auto operands = [&](int i, double4_t& acc, double (&pb)[KBS]) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const int o = i * n * LDP + r4 * P * LDP;
                acc[r4] = fma(cb, bbase[o], fma(ca, abase[o], tbase[o]));
            }
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
                const int o = i * n * LDP;
                const double h = fma(cy[kb], ob[kb][1][o], cx[kb] * ob[kb][0][o]);
                pb[kb] = fma(-dt, ob[kb][2][o], (-0.5 * dt) * h);       // -(P_i B): the B operand holds -[K | kappa]
            }
        };
        auto write_back = [&](int i, double4_t acc) {
            acc[2] = fma(c2, acc[1], fma(c0, acc[0], acc[2]));          // A': heading row
            acc[3] = fma(c3, acc[1], fma(c1, acc[0], acc[3]));          // A': speed row
            if (lrow < n + 1 && lq < P) {
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) wbase[i * n * LDP + r4 * P * LDP] = acc[r4];
            }
        };
        run(operands, write_back);
    }
}

// Team of two: the helper wavefront's whole backward sweep (newton_direction_tile sends wavefront 1 here; split off in round 6 for
// readability).  Step by step between wavefront 0's two barriers: value recursion, Q-add, V rows and tail of the odd players, then -- while
// wavefront 0 runs its half of the serial tail -- the fetch of the next step record, the same pivoted solve, the odd rows of [F | f], the gains.
template <class C, class QAM>
__device__ __forceinline__ void direction_helper_wavefront(CPR pr, const Game& G, DirLds<C>& L, const QAM& qam, int N, double dt, int tid, int lrow, int lq, double reg) {
    constexpr int n = C::n, m = C::m, NK = m * (n + 1), VW = DirLds<C>::VW, BT = WAVE;
    using R = Rec<C>;
    constexpr int RPL = (R::LEN_SWEEP + BT - 1) / BT;
    // the helper wavefront: value recursion and Q-add of the odd players, step by step between wavefront 0's two barriers
    double* const bwh = reinterpret_cast<double*>(&L.bw) + (int)(offsetof(typename DirLds<C>::Bwd, Pm) / 8);
    int curh = 0;
    for (int k = N - 2; k >= 0; k--, curh ^= 1) {
        team_lds_barrier();
        if (k < N - 2) {
            t_half<C>(L, 1, tid);
            value_recursion_half<C>(L, 1, lrow, lq, dt);
        }
        sweep_sync<C>();
        qam.apply(tid, L.rec[curh], L.qdf, bwh, reg, (k + 1 < N - 1) ? dt : 1.0, -1);
        sweep_sync<C>();
        v_sysrow<C>(L, L.rec[curh], k, dt, 2 * (tid >> 4) + 1, tid & 15);       // V rows of the odd players' controls
        player_tail_half<C>(L, L.rec[curh], dt, k, N, 1, tid);                 // their s_i, y_i, g_c
        team_lds_barrier();
        // while wavefront 0 runs the serial tail of step k: the record of step k - 1 from global memory into the other LDS slot
        // (wavefront 0 never waits for a load inside the sweep)
        double nx[RPL];
        if (k > 0) {
#pragma unroll
            for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; nx[q] = e < R::LEN_SWEEP ? gld(G.rec(pr) + (size_t)(k - 1) * R::LEN, e) : 0.0; }
        }
        // ... and its share of that tail, with the record's loads in flight: the same columns of [W | V A_k | g], the same pivoted
        // solve (every wavefront needs the solved columns for its rows), the odd rows of [F | f] = [A_k | rd] + B [K | kappa],
        // and the gain stores -- wavefront 0 forms the even rows and issues no store at all in the sweep
        {
            using GLh = GjLanes<m, n + 1>;
            const double* Rh = L.rec[curh]; const double* coefh = Rh + R::COEF;
            const int cidx = GLh::column(tid); const bool rhsl = GLh::rhs(tid);
            double col[m];
#pragma unroll
            for (int c = 0; c < m; c++) col[c] = L.bw.V[c * VW + cidx];
            gj_solve_cols_dpp<m>(col);
            if (rhsl) {
                const int cc = cidx - m;
#pragma unroll
                for (int c = 0; c < m; c++) col[c] = -col[c];
                const double* acol = (cc < n) ? &L.bw.T[cc * n] : Rh + R::RD;
                double fxv[n];
#pragma unroll
                for (int r = 1; r < n; r += 2) fxv[r] = B_vec<C>(coefh, dt, [&](int c2) { return col[c2]; }, r) + acol[r];
#pragma unroll
                for (int r = 1; r < n; r += 2) {
                    if constexpr (n < 16) L.bw.Fx[r * 16 + cc] = fxv[r];
                    else { double* dst = cc < n ? &L.bw.Fx[r * 16 + cc] : &L.bw.fv[r]; *dst = fxv[r]; }
                }
            }
            if (k > 0) {
#pragma unroll
                for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; if (e < R::LEN_SWEEP) L.rec[curh ^ 1][e] = nx[q]; }
            }
            asm volatile("" ::: "memory");
            if (rhsl) {
                double* __restrict__ Kg = G.kgain(pr) + (size_t)k * NK;
#pragma unroll
                for (int c = 0; c < m; c++) gst(Kg, (cidx - m) * m + c, col[c]);
            }
        }
    }
    game_sync();                 // the gains are this wavefront's stores: wavefront 0's forward sweep reads them behind a full barrier
}

// Forward sweep for (dx, du) and costate sweep for dlambda of the tile path (newton_direction_tile's second half; split off in round 6 for
// readability -- __forceinline__, same code).  Runs on one wavefront (wavefront 0 of a team); the gains and the step records are in
// global memory behind the caller's backward sweep.
template <class C, bool IBR>
__device__ __forceinline__ int direction_forward_costate(CPR pr, const Game& G0, DirLds<C>& L, int N, double dt, int lane, double reg, int ip, double* primal_l1) {
    constexpr int n = C::n, m = C::m, P = C::P, NK = m * (n + 1);
    using R = Rec<C>;
    constexpr int KPL = (NK + WAVE - 1) / WAVE;
    constexpr bool TEAM = C::NW >= 4 && !IBR;
    constexpr bool AUGS = DirLds<C>::AUGS;
    constexpr bool SPLITU = C::MODEL == ALG_MODEL_UNICYCLE && P == 3;
    constexpr bool SPLITF = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || SPLITU) && AUGS && !IBR && (C::NW == 1 || TEAM);      // (the gains in HBM are -[K | kappa])
    HxMap<C> hxm;
    ALG_PROF_DECL
    // ------------------------------------------------------------------ forward sweep: dx, du
    if constexpr (C::NW == 1) game_sync(); else dir_sync<C>();   // the gains are in global memory (the sweeps' own syncs order LDS only)
    Game G = G0.fresh();
    double* __restrict__ dz = G.z(2);
    if (lane < n) dz[lane] = 0.0;
    constexpr bool DIROW = P * 16 <= WAVE;
    constexpr int NPOS = C::POS ? C::PD * P : 1;
    // FWDW: the forward sweep also forms the part of dlambda_k that does not depend on dlambda_{k+1} -- w_k = rx + Q^ dx_{k+1} -- right after
    // dx_{k+1} exists, in the bubbles of its own dependency chain (every 16-lane row runs the forward recursion redundantly, so row i has
    // dx for player i's products), and parks it in dlambda's slot; the costate sweep is left with dlambda_k = w_k + A' dlambda_{k+1}: one
    // load, a few shifts, no LDS, no fence.  Bit-identical: w_k is exactly the intermediate value the one-sweep form holds in a register.
    // (double integrator only: measured +1.1 % at C2; for the unicycle the recursion's coefficient loads cost more than the shorter chain
    // saves: C3 -0.8 %, C5 loop -0.6 %)
    constexpr bool FWDW = DIROW && C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR;
    // unconditional loads from clamped addresses (a conditional load into a zeroed register costs a vmcnt drain, see the forward sweep)
    const int cdxo = DIROW ? ((lane & 15) < n ? (lane & 15) : 0) : (lane < n ? lane : 0);
    const int ri_ = lane >> 4, rr_ = lane & 15;
    const bool rok = DIROW && ri_ < P && rr_ < n;
    const int re_ = rok ? ri_ * n + rr_ : 0;                           // entry of rx / qdf / dlambda this lane owns
    const double qdfv = DIROW ? L.qdf[re_] : 0.0;
    int hso[NPOS]; float hsg[NPOS];                                     // record offset and sign of Q^_i's position-block entry (row rr_, column c)
    if constexpr (DIROW && C::POS) {
        constexpr int NS = C::NS;
        const int i = ri_ < P ? ri_ : 0, jr = rr_ % P, ar = rr_ / P;
#pragma unroll
        for (int c = 0; c < NPOS; c++) {
            const int jc = c % P, h = C::sym(ar < C::PD ? ar : 0, c / P);
            int so = R::HH; float sg = 0.f;
            if (rr_ < C::PD * P) {
                if (jr == i && jc == i) { so = R::HD + NS * i + h; sg = 1.f; }
                else if (jr == i) { so = R::HH + NS * pairq<C>(i, jc) + h; sg = -1.f; }
                else if (jc == i) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = -1.f; }
                else if (jr == jc) { so = R::HH + NS * pairq<C>(i, jr) + h; sg = 1.f; }
            }
            hso[c] = so; hsg[c] = sg;
        }
    }
    double lamp = 0.0;                                                  // dlambda of the previous (later) step, entry re_
    // state-dependent models: this lane's entries of A_{k+1}' (AT_vec), taken from step k + 1's record one iteration earlier
    const int cblk = rr_ / P, cpi = rr_ % P;
    const bool c_hi = C::MODEL == ALG_MODEL_BICYCLE ? cblk == 2 : cblk == 3, c_on = cblk >= 2 && rr_ < n;
    const int cia = (c_hi ? 1 : 0) * P + cpi, cib = (c_hi ? 3 : 2) * P + cpi, cic = 4 * P + cpi;
    double can = 0.0, cbn = 0.0, ccn = 0.0;
    // the forward sweep reads only [coef | rd] of a record: one load per lane
    static_assert(C::NC + n <= WAVE, "forward sweep record slice");
    const int fro = lane < C::NC ? R::COEF + lane : R::RD + (lane - C::NC);      // record offset of this lane's slice entry
    const bool frok = lane < C::NC + n;
    // FWDW: the slice is [coef | Hh | Hd | RQ | rx] (the costate's) followed by rd, FPL entries per lane (entries past the end duplicate rd[0])
    constexpr int FSL2 = R::LEN_COSTATE + n, FPL = FWDW ? (FSL2 + WAVE - 1) / WAVE : 1;
    int fso[FPL];
#pragma unroll
    for (int q = 0; q < FPL; q++) { const int e = lane + q * WAVE; fso[q] = FWDW ? (e < R::LEN_COSTATE ? e : (e < FSL2 ? R::RD + (e - R::LEN_COSTATE) : R::RD)) : (frok ? fro : R::RD); }
    if constexpr (FWDW) {
#pragma unroll
        for (int q = 0; q < FPL; q++) L.rec[0][fso[q]] = gld(G.rec(pr), fso[q]);
    } else if (frok) L.rec[0][fro] = gld(G.rec(pr), fro);
    for (int e = lane; e < NK; e += WAVE) L.fw.kg[0][e] = gld(G.kgain(pr), e);
    // Global-memory schedule of a step.  gfx9 counts loads and stores in one vmcnt, so a wait for loaded data also waits for the
    // write acknowledgement of every store in flight; and a conditional load into a zero-initialised register makes the compiler
    // drain vmcnt at the top of the loop (write-after-write on the register).  Hence: unconditional loads from clamped
    // addresses, and everything at the tail of the step in the order (1) land the data of step k+1 (requested one step ago) in
    // LDS, (2) issue this step's result stores, (3) request the data of step k+2 -- the single wait of a step meets loads and
    // stores that have been in flight for a whole step.
    auto fwd_load = [&](int kk, double (&rf)[FPL], double (&rk)[KPL]) {
#if ALG_R6_SADDR
        const int kc = __builtin_amdgcn_readfirstlane(kk < N - 1 ? kk : N - 2);
        // (the step's two base addresses as scalar pairs, see the gain stores of the backward sweep)
        const double* const Rb = as_global(reinterpret_cast<const double*>(uniform_u64(reinterpret_cast<unsigned long long>(G.rec(pr) + (size_t)kc * R::LEN))));
        const double* const Kb = as_global(reinterpret_cast<const double*>(uniform_u64(reinterpret_cast<unsigned long long>(G.kgain(pr) + (size_t)kc * NK))));
#else
        const int kc = kk < N - 1 ? kk : N - 2;
        const double* const Rb = G.rec(pr) + (size_t)kc * R::LEN; const double* const Kb = G.kgain(pr) + (size_t)kc * NK;
#endif
#pragma unroll
        for (int q = 0; q < FPL; q++) rf[q] = gld(Rb, fso[q]);
#pragma unroll
        for (int q = 0; q < KPL; q++) { const int e = lane + q * WAVE; rk[q] = gld(Kb, e < NK ? e : NK - 1); }
    };
    // Prefetch ring: the slices of steps k + 1 .. k + SD are in flight in registers while step k computes.  With four games per
    // SIMD all streaming, a fetch takes about two microseconds -- longer than a step of this sweep -- so with one step in flight
    // (rounds 1-2) the sweep ran at memory latency: 4.7 K cycles per step at 4096 games against 1.0 K for a lone wavefront
    // (tests/probes/phase_prof.py).  The loop is unrolled by SD so that every ring slot is a fixed register (a rotating copy
    // would read the newest load and wait for it).
    constexpr int SD = C::SWEEP_DEPTH;
    double pref[SD][FPL], prek[SD][KPL];
#pragma unroll
    for (int u = 0; u < SD; u++) fwd_load(1 + u, pref[(1 + u) % SD], prek[(1 + u) % SD]);
    sweep_sync<C>();
    int curv = 0;
    double pl1 = 0.0;                               // sum |dx| + |du| of this lane's entries (Delta_step, primal_dual_traj.jl:130-147)
    // non-finite direction entries (checked where they are produced): lane mask in scalar registers.  Neither the sum nor the check is masked per step:
    // the entries of the lanes outside the vectors are exact zeros (finite, and x + 0 = x), the 16-lane rows of the FWDW form hold replicas of row 0 --
    // the sum drops them once, after the sweep (bit-identical; 8 VALU instructions per step less)
    unsigned long long badm = 0;
    // dx_k lives one entry per lane (lanes 0..n-1) and is broadcast with v_readlane; du and dx_{k+1} never pass through LDS:
    // one LDS round trip (gain rows, record slice) per step instead of three.
    double dxr = 0.0;
    for (int k0 = 0; k0 < N - 1; k0 += SD) {
#pragma unroll
      for (int u = 0; u < SD; u++) {
        const int k = k0 + u;
        if (k >= N - 1) break;
#if ALG_R6_CURC
        // (the LDS double buffer's index as a compile-time constant of the unrolled step -- SD is even, the sweep starts at slot 0 -- so that every LDS
        // address of the step is a loop-invariant lane offset plus an immediate: with a run-time index each of the step's ~13 addresses was re-formed)
        static_assert(SD % 2 == 0 || SD == 1, "sweep depth");
        const int cur = (SD % 2 == 0) ? (u & 1) : curv;
#else
        const int cur = curv;
#endif
        const double* Rc = L.rec[cur]; const double* Kl = L.fw.kg[cur];
        const int fl = FWDW ? (lane & 15) : lane;     // FWDW: every 16-lane row runs the recursion (same LDS addresses, same instructions)
        const int cl = fl < m ? fl : 0;
        double acc = Kl[n * m + cl];
        auto land_next = [&]() {
            // (1) data of step k+1 (requested SD steps ago) -> LDS (clamped duplicates at the last steps are never read)
#pragma unroll
            for (int q = 0; q < FPL; q++) L.rec[cur ^ 1][fso[q]] = pref[(u + 1) % SD][q];
#pragma unroll
            for (int q = 0; q < KPL; q++) { const int e = lane + q * WAVE; L.fw.kg[cur ^ 1][e < NK ? e : NK - 1] = prek[(u + 1) % SD][q]; }
        };
        if constexpr (ALG_R6_FWD_LAND == 1) land_next();
        if constexpr (rowdot_split_v<C> && ALG_R6_ROWDOT_CHAINS > 2) rowdot_dpp_chains<n, ALG_R6_ROWDOT_CHAINS, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(acc, dxr, [&](int q) { return Kl[q * m + cl]; });
        else if constexpr (rowdot_split_v<C>) rowdot_dpp_split<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(acc, dxr, [&](int q) { return Kl[q * m + cl]; });
        else rowdot_dpp_g<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(acc, dxr, [&](int q) { return Kl[q * m + cl]; });                  // dx_k sits in lanes 0..n-1 of the row, the control rows in its lanes 0..m-1 (same FMA order as the v_readlane form)
        if constexpr (ALG_R6_FWD_LAND == 2) land_next();
        const double duv = fl < m ? (SPLITF ? -acc : acc) : 0.0;     // (split recursion: the gains in HBM are -[K | kappa])
        const double rdv = Rc[R::RD + (fl < n ? fl : 0)];
        double dxn = fwd_next<C>(Rc + R::COEF, dt, dxr, duv, fl) + rdv;
        dxn = fl < n ? dxn : 0.0;
#if ALG_R6_PL1
        pl1 += fabs(duv); pl1 += fabs(dxn);
        badm |= __builtin_amdgcn_ballot_w64(!isfinite(duv)) | __builtin_amdgcn_ballot_w64(!isfinite(dxn));
#else
        if (lane < m) { pl1 += fabs(duv); badm |= __builtin_amdgcn_ballot_w64(!isfinite(duv)); }
        if (lane < n) { pl1 += fabs(dxn); badm |= __builtin_amdgcn_ballot_w64(!isfinite(dxn)); }
#endif
        dxr = dxn;
        if constexpr (FWDW) {
            // w_k = rx_{i,k+1} + Q^_{i,k+1} dx_{k+1} for (player, row) = (ri_, rr_): the head of the costate sweep's FMA sequence
            const double wq = (k + 1 < N - 1) ? dt : 1.0;
            double qd = reg + wq * qdfv;
            if constexpr (C::EXT) qd += Rc[R::RQ + re_];
            double wk = Rc[R::RX + re_] + qd * dxn;
            if constexpr (C::POS) {
                double hv[NPOS];
#pragma unroll
                for (int c = 0; c < NPOS; c++) hv[c] = (double)hsg[c] * Rc[hso[c]];
                double t = wk;
                if constexpr (ALG_R6_WCHAIN_SPLIT != 0 && rowdot_split_v<C>) rowdot_dpp_split<NPOS, NPOS>(t, dxn, [&](int c) { return hv[c]; });
                else rowdot_dpp<NPOS>(t, dxn, hv);
                wk = rr_ < C::PD * P ? t : wk;
            }
            if (rok) gst(dz + n + hl<C>(k, 0), re_, wk);
        }
        if constexpr (ALG_R6_FWD_LAND == 0) land_next();
        asm volatile("" ::: "memory");
        // (2) results out
        if (lane < m) gst(dz + n + hu<C>(k, 0), uoff<C>(lane), duv);
        if (lane < n) gst(dz + n + hx<C>(k), lane, dxn);
        // (3) request step k+1+SD into the slot that was just emptied
        fwd_load(k + 1 + SD, pref[(u + 1) % SD], prek[(u + 1) % SD]);
        sweep_sync<C>();
        curv ^= 1;
      }
    }
#if defined(ALG_DIR_STOP) && ALG_DIR_STOP == 2
    return ALG_STATUS_OK;
#endif
    ALG_PROF(7)
    // ------------------------------------------------------------------ costate sweep:
    //   dlambda_{i,k} = Q^_{i,k+1} dx_{k+1} + A_{k+1}' dlambda_{i,k+1} + rx_{i,k+1}
    // Lane 16 i + r = (player i, row r), one 16-lane row per player.  dx_{k+1} is replicated in every row and reaches the position-block
    // products through the DPP row broadcast, A' dlambda is a few shifts inside the row (double integrator: velocity row r takes dt
    // times position row r - m; unicycle / bicycle: the heading / speed rows take the coefficient-weighted position rows r - P .. r - 3P),
    // the pair-Hessian entries are read straight from the record with per-lane offsets: no dx / dlambda / table round trips through
    // LDS and one fence per step instead of three.  Same products in the same order as the general form below.
    if constexpr (!DIROW) hxm.init(phase_lane());
    if constexpr (C::NW == 1) game_sync(); else dir_sync<C>();   // dx of every step is in global memory
    G = G0.fresh();
    dz = G.z(2);
    if constexpr (FWDW) {
        // dlambda_k = w_k + A_{k+1}' dlambda_{k+1}: w_k comes back from dlambda's own slot (this lane wrote it in the forward sweep), the
        // coefficients of A_{k+1} (state-dependent models) from step k + 1's record; SD steps in flight, no LDS, no fence
        constexpr int NCF = C::NC > 0 ? (C::MODEL == ALG_MODEL_BICYCLE ? 3 : 2) : 0;
#ifndef ALG_FWDW_DEPTH
#define ALG_FWDW_DEPTH 8
#endif
        constexpr int CD = ALG_FWDW_DEPTH;           // steps in flight: three doubles per slot, and nothing but these loads feeds the recursion
        double wkr[CD], cfr[CD][NCF > 0 ? NCF : 1];
        auto cw_load = [&](int kk, double& wv, double (&cf)[NCF > 0 ? NCF : 1]) {
            const int kc = kk > 0 ? kk : 0, kn = kc + 1 < N - 1 ? kc + 1 : N - 2;
            wv = gld(dz + n + hl<C>(kc, 0), re_);
            if constexpr (NCF > 0) {
                const double* Rn = G.rec(pr) + (size_t)kn * R::LEN + R::COEF;
                cf[0] = Rn[cia]; cf[1] = Rn[cib];
                if constexpr (NCF > 2) cf[2] = Rn[cic];
            }
        };
#pragma unroll
        for (int u = 0; u < CD; u++) cw_load(N - 2 - u, wkr[u], cfr[u]);
        for (int k0 = N - 2; k0 >= 0; k0 -= CD) {
#pragma unroll
          for (int u = 0; u < CD; u++) {
            const int k = k0 - u;
            if (k < 0) break;
            double acc = wkr[u];
            if (k < N - 2) {
                if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                    const bool hi = rr_ >= m; const double sh = row_shift<0x110 + m>(lamp); acc += lamp + (hi ? dt : 0.0) * (hi ? sh : lamp);
                } else {
                    const double ca_ = cfr[u][0], cb_ = cfr[u][1];
                    const double s1 = row_shift<0x110 + P>(lamp), s2 = row_shift<0x110 + 2 * P>(lamp), s3 = row_shift<0x110 + 3 * P>(lamp);
                    const double vi = cblk == 3 ? s3 : s2, vpi = cblk == 3 ? s2 : s1;
                    if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
                        const double u1 = row_shift<0x100 + P>(lamp), cc_ = cfr[u][NCF > 2 ? 2 : 0];
                        acc += lamp + (c_on ? ca_ : 0.0) * (c_on ? vi : lamp) + (c_on ? cb_ : 0.0) * (c_on ? vpi : lamp) + (cblk == 2 ? cc_ : 0.0) * (cblk == 2 ? u1 : lamp);
                    } else acc += lamp + (c_on ? ca_ : 0.0) * (c_on ? vi : lamp) + (c_on ? cb_ : 0.0) * (c_on ? vpi : lamp);
                }
            }
            acc = (rok && (!IBR || ri_ == ip)) ? acc : 0.0;
            lamp = acc;
            if (rok) gst(dz + n + hl<C>(k, 0), re_, acc);
            badm |= __builtin_amdgcn_ballot_w64(!isfinite(acc));
            cw_load(k - CD, wkr[u], cfr[u]);
          }
        }
    } else {
    constexpr int RPLC = (R::LEN_COSTATE + WAVE - 1) / WAVE;
    for (int e = lane; e < R::LEN_COSTATE; e += WAVE) L.rec[0][e] = gld(G.rec(pr) + (size_t)(N - 2) * R::LEN, e);
    const int ci_ = lane < P * n ? lane / n : 0, cr_ = lane < P * n ? lane % n : 0;        // (player, row) of this lane
    const bool cpos = C::POS && cr_ < C::PD * P;
    double dxk = lane < n ? gld(dz + n + hx<C>(N - 2), lane) : 0.0;      // dx_{k+1}, fetched ahead like the records
    if constexpr (DIROW) dxk = gld(dz + n + hx<C>(N - 2), cdxo);
    auto cs_load = [&](int kk, double& rdx, double (&rr)[RPLC]) {
        const int kc = kk > 0 ? kk : 0;
#pragma unroll
        for (int q = 0; q < RPLC; q++) { const int e = lane + q * WAVE; rr[q] = gld(G.rec(pr) + (size_t)kc * R::LEN, (e < R::LEN_COSTATE ? e : R::LEN_COSTATE - 1)); }
        rdx = gld(dz + n + hx<C>(kc), cdxo);
    };
    // register ring like the forward sweep's: [record slice | dx] of steps k - 1 .. k - SD are in flight while step k computes
    double pre[SD][RPLC], pdx[SD];
#pragma unroll
    for (int u = 0; u < SD; u++) cs_load(N - 3 - u, pdx[(1 + u) % SD], pre[(1 + u) % SD]);
    sweep_sync<C>();
    int curc = 0;
    for (int k0 = N - 2; k0 >= 0; k0 -= SD) {
#pragma unroll
      for (int u = 0; u < SD; u++) {
        const int k = k0 - u;
        if (k < 0) break;
#if ALG_R6_CURC
        const int cur = (SD % 2 == 0) ? (u & 1) : curc;      // (compile-time slot index, like the forward sweep)
#else
        const int cur = curc;
#endif
        const double* Rc = L.rec[cur];
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        if constexpr (DIROW) {
            double qd = reg + w * qdfv;
            if constexpr (C::EXT) qd += Rc[R::RQ + re_];
            double acc = Rc[R::RX + re_] + qd * dxk;
            if constexpr (C::POS) {
                double hv[NPOS];
#pragma unroll
                for (int c = 0; c < NPOS; c++) hv[c] = (double)hsg[c] * Rc[hso[c]];
                double t = acc;
                rowdot_dpp<NPOS>(t, dxk, hv);
                acc = rr_ < C::PD * P ? t : acc;
            }
            if (k < N - 2) {
                if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                    const bool hi = rr_ >= m; const double sh = row_shift<0x110 + m>(lamp); acc += lamp + (hi ? dt : 0.0) * (hi ? sh : lamp);
                } else {
                    // AT_vec: v(r) + ca v(i) + cb v(P + i) (+ cc v(3P + i), bicycle heading rows), i = r % P: rows 2P + i take lanes r - 2P, r - P
                    // (, r + P), rows 3P + i take lanes r - 3P, r - 2P
                    const double s1 = row_shift<0x110 + P>(lamp), s2 = row_shift<0x110 + 2 * P>(lamp), s3 = row_shift<0x110 + 3 * P>(lamp);
                    const double vi = cblk == 3 ? s3 : s2, vpi = cblk == 3 ? s2 : s1;
                    if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
                        const double u1 = row_shift<0x100 + P>(lamp);
                        acc += lamp + (c_on ? can : 0.0) * (c_on ? vi : lamp) + (c_on ? cbn : 0.0) * (c_on ? vpi : lamp) + (cblk == 2 ? ccn : 0.0) * (cblk == 2 ? u1 : lamp);
                    } else acc += lamp + (c_on ? can : 0.0) * (c_on ? vi : lamp) + (c_on ? cbn : 0.0) * (c_on ? vpi : lamp);
                }
            }
            if constexpr (C::NC > 0) { can = Rc[R::COEF + cia]; cbn = Rc[R::COEF + cib]; if constexpr (C::MODEL == ALG_MODEL_BICYCLE) ccn = Rc[R::COEF + cic]; }
            acc = (rok && (!IBR || ri_ == ip)) ? acc : 0.0;
            lamp = acc;
            if (rok) gst(dz + n + hl<C>(k, 0), re_, acc);
            badm |= __builtin_amdgcn_ballot_w64(!isfinite(acc));
        } else {
        if (lane < n) L.fw.dx[lane] = dxk;
        hxm.expand(lane, Rc, L.fw.hx);
        sweep_sync<C>();
        double acc = 0.0;
        if (lane < P * n && (!IBR || ci_ == ip)) {
            double qd = reg + w * L.qdf[lane];
            if constexpr (C::EXT) qd += Rc[R::RQ + lane];
            acc = Rc[R::RX + lane] + qd * L.fw.dx[cr_];
            if (cpos) {
                const double* hrow = &L.fw.hx[(ci_ * P + cr_ % P) * P * C::NS];
                const int ar = cr_ / P;
#pragma unroll
                for (int c = 0; c < C::PD * P; c++) acc += hrow[(c % P) * C::NS + C::sym(ar, c / P)] * L.fw.dx[c];
            }
            if (k < N - 2) { const double* dli = &L.fw.dl[ci_ * n]; acc += AT_vec<C>(L.coefn, dt, [&](int rr) { return dli[rr]; }, cr_); }
        }
        sweep_sync<C>();
        if (lane < P * n) { L.fw.dl[lane] = acc; gst(dz + n + hl<C>(k, 0), lane, acc); }
        badm |= __builtin_amdgcn_ballot_w64(lane < P * n && !isfinite(acc));
        if (C::NC > 0 && lane < C::NC) L.coefn[lane] = Rc[R::COEF + lane];
        }
        // land step k - 1 (requested SD steps ago), then request step k - 1 - SD into the emptied slot
        dxk = (DIROW || lane < n) ? pdx[(u + 1) % SD] : 0.0;
        if (k > 0) {
#pragma unroll
            for (int q = 0; q < RPLC; q++) { const int e = lane + q * WAVE; if (e < R::LEN_COSTATE) L.rec[cur ^ 1][e] = pre[(u + 1) % SD][q]; }
        }
        cs_load(k - 1 - SD, pdx[(u + 1) % SD], pre[(u + 1) % SD]);
        sweep_sync<C>();
        curc ^= 1;
      }
    }
    }
    ALG_PROF(8)
    ALG_PROF_FLUSH
    // non-finite direction -> singular (the reference would throw / propagate NaN)
    if (primal_l1) *primal_l1 = wave_sum((FWDW && lane >= 16) ? 0.0 : pl1);
    return badm != 0 ? ALG_STATUS_SINGULAR : ALG_STATUS_OK;     // scalar: the solver's control flow stays on the SALU
}

template <class C, bool IBR>
__device__ int newton_direction_tile(CPR pr0, const Game& G0, DirLds<C>& L, double reg, int ip, double* primal_l1) {
#if defined(ALG_DIR_STOP) && ALG_DIR_STOP == 0       // per-sweep byte / time accounts (tests/probes/dir_split.sh): nothing at all
    return ALG_STATUS_OK;
#endif
    CPR pr = phase_params(pr0);
    Game G = G0.fresh();
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, KB = DirLds<C>::KB, NK = m * (n + 1);
    using R = Rec<C>;
    // Team kernels (Cfg::NW wavefronts per game): the backward sweep runs on the whole team -- player i's value recursion on
    // wavefront i % NW, the table-driven phases strided over all threads, the column build and the pivoted solve redundantly in
    // every wavefront (their result is needed everywhere for the closed-loop rows), real workgroup barriers between the phases;
    // the forward and costate sweeps then run on wavefront 0 alone.
    // (a team of two gains less from the split than its barriers cost -- measured on C3 at 1024 games: 2.35 M/s with the whole
    // direction on wavefront 0, 2.30 M/s with the split -- so only teams of four or more split the backward sweep)
    constexpr bool TEAM = C::NW >= 4 && !IBR;
    constexpr bool HELP2 = help2_v<C, IBR>;              // team of two: wavefront 1 runs the odd players' value recursion (above)
    constexpr int BT = TEAM ? C::NT : WAVE;               // threads of the backward sweep
    const int N = phase_int(pr.N);
    const int hw = HELP2 ? team_wave<C>() : 0;            // 1: the helper wavefront of a team of two
    const int tid = HELP2 ? (phase_lane() & 63) : phase_lane();
    const int lane = TEAM ? (tid & 63) : tid;             // lane inside the wavefront
    const int tw = TEAM ? team_wave<C>() : 0;
    // (team: the wavefronts exchange LDS only inside the backward sweep -- a barrier that also drained vmcnt made every phase boundary wait
    // for the record prefetch and the gain stores in flight, like the single-wavefront sweeps before round 3)
    auto bsync = [&]() { if constexpr (TEAM) team_lds_barrier(); else sweep_sync<C>(); };
    const int lrow = lane & 15, lq = lane >> 4;          // MFMA lane coordinates
    const double dt = phase_f64(pr.dt);
    constexpr int RPL = (R::LEN_SWEEP + BT - 1) / BT;          // record doubles per thread
    constexpr int KPL = (NK + WAVE - 1) / WAVE;
    constexpr bool AUGS = DirLds<C>::AUGS;               // s_i rides through the first MFMA product (n < 16)
    constexpr int KB1 = DirLds<C>::KB1, VW = DirLds<C>::VW;
    constexpr bool GFUSE = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE);
    constexpr bool SYSROW = GFUSE;      // the V phase forms the system's rows (needs g from the y lanes)
    // Split value recursion (round 5; double integrator): with F = A_k + B K and f = rd + B kappa,
    //   [P_i | s_i] [[F f],[0 1]] = [P_i A_k | y_i] + (P_i B) [K | kappa],      y_i = P_i rd + s_i  (already formed for g_c),
    // and for the double integrator both P_i A_k (P + dt x its columns shifted by m) and P_i B (dt^2/2 x columns 0..m-1 + dt x columns
    // m..n-1) are two-term combinations of entries of P_i.  The accumulator tile starts at [P_i A_k | y_i] and the product has inner
    // dimension m instead of n + 1: ceil(m / 4) f64 MFMAs per player instead of (n + 4) / 4 (C2: 6 per step instead of 12 -- the matrix
    // pipe of a SIMD is shared by its four resident games and was busy 3 250 of the ~10 500 cycles of a backward step), and the
    // closed-loop phase ([F | f] = [A_k | rd] + B [K | kappa] -> LDS) disappears: the solved columns [K | kappa] go to LDS as they are
    // (the B operand of the product) and y_i replaces s_i in column n of the player's rows.
    // (round 5, later: the 3-player unicycle -- C5 -- as well.  Its A_k and B_k carry the step's four Jacobian coefficients per player, so
    // P_i A_k and P_i B_k take three entries of P_i each; with player i's four state rows placed in lane group i of the tile, A' is
    // lane-local too and the ds_bpermute gathers of the unsplit form (P3Gather) go away.)
    constexpr bool SPLITU = C::MODEL == ALG_MODEL_UNICYCLE && P == 3;
    constexpr bool SPLITF = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || SPLITU) && AUGS && !IBR && (C::NW == 1 || TEAM);
    constexpr int KBS = (m + 3) / 4;                    // k-blocks of the split product (inner dimension m)
    QaddMap<C, BT, (HELP2 ? 2 : 1)> qam; qam.init(tid, hw);
    if constexpr (HELP2) {
        if (hw == 1) { direction_helper_wavefront<C>(pr, G, L, qam, N, dt, tid, lrow, lq, reg); return ALG_STATUS_OK; }
    }
    struct NoGather { __device__ void init(int, int) {} };
    typename std::conditional<(C::P == 3 && C::MODEL != ALG_MODEL_DOUBLE_INTEGRATOR), P3Gather<C>, NoGather>::type p3g;
    p3g.init(lq, lrow);
    for (int e = tid; e < P * n; e += BT) { const int i = e / n, r = e % n; L.qdf[e] = (r % P == i) ? G.Qd(pr)[i * C::ni + r / P] : 0.0; }
    // (split recursion: rows 0..m-1 of Fx hold [K | kappa] of the step before, rows m.. and columns n+1.. stay zero)
    for (int e = tid; e < 16 * 16; e += BT) L.bw.Fx[e] = (!SPLITF && AUGS && e == n * 16 + n) ? 1.0 : 0.0;   // row n = e_n: passes s_i through
    for (int e = tid; e < P * n * LDP; e += BT) L.bw.Pm[e] = 0.0;
    for (int e = tid; e < m * VW; e += BT) L.bw.V[e] = 0.0;
    for (int e = tid; e < (SPLITF ? 0 : n * n); e += BT) {      // constant part of A' (the coefficient entries follow per step)
        const int c = e / n, r = e % n;
        double v = (r == c) ? 1.0 : 0.0;
        if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) { if (r < m && c == r + m) v = dt; }
        L.bw.T[e] = v;
    }
    if (tid == 0) L.bw.pad[0] = 0.0;
    for (int e = tid; e < R::LEN_SWEEP; e += BT) L.rec[0][e] = gld(G.rec(pr) + (size_t)(N - 2) * R::LEN, e);
    // ---- loop-invariant lane roles of the MFMA tiles: register r4 holds (row = lq + 4 r4, col = lrow)
    const bool colP = lrow < n;
    bool rowok[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) rowok[r4] = (lq + 4 * r4) < n;
    bsync();
    double* const bwb = reinterpret_cast<double*>(&L.bw);
    constexpr int oPm = (int)(offsetof(typename DirLds<C>::Bwd, Pm) / 8), oPad = (int)(offsetof(typename DirLds<C>::Bwd, pad) / 8);
    // ------------------------------------------------------------------ backward sweep
    // P_i (n x n) and s_i (column n of the same LDS rows): P_i <- Q^_i + A_{k+1}' P_i F,  s_i <- rx_i + A_{k+1}' (P_i f + s_i)
    int cur = 0, sing = 0;
    ALG_PROF_DECL
    ALG_PROF(11)
    for (int k = N - 2; k >= 0; k--, cur ^= 1) {
        const double* Rc = L.rec[cur];
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        const double* coefk = Rc + R::COEF;
        // ---- value recursion.  n < 16: [P_i | s_i] [[F f],[0 1]] = [P_i F | P_i f + s_i], then A' x that (on the result tile in
        // registers where the model's row layout allows it, else a second MFMA product) -- f64 MFMA chains per player (f and s_i
        // ride in the spare tile column / k-block).  n == 16: the products cover P_i only and
        // s_i <- rx_i + A'(P_i f + s_i) runs on the VALU.  Accumulators start at zero; the sparse Q^_i (and rx_i) are added
        // afterwards (Q-add phase).  Player i's chain reads only row block i of Pm, so its result is written back before the
        // next player starts: one accumulator tile live.
        if constexpr (SPLITF) {
          if (k < N - 2) {
            split_value_recursion<C, TEAM>(L, dt, lrow, lq, tw);
            bsync();
          }
        } else
        if (k < N - 2) {
            if constexpr (!AUGS && !HELP2) {            // (team of two: per player, between the barriers below)
                for (int e = tid; e < P * n; e += BT) {                     // t_i = P_i f + s_i (one (i,r) per thread)
                    const int i = e / n, r = e % n; double a = L.bw.Pm[i * n * LDP + r * LDP + n];
                    for (int c = 0; c < n; c++) a += L.bw.Pm[i * n * LDP + r * LDP + c] * L.bw.fv[c];
                    L.bw.t[e] = a;
                }
                bsync();
            }
            double bF[KB1], aA[KB];
#pragma unroll
            for (int kb = 0; kb < KB1; kb++) bF[kb] = L.bw.Fx[(4 * kb + lq) * 16 + lrow];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) aA[kb] = colP ? A_entry<C>(L.coefn, dt, 4 * kb + lq, lrow) : 0.0;    // (A')[lrow][k] = A[k][lrow]
            // (measured for the 128-VGPR configurations as well in round 3: 126 VGPRs, no spills, C2 10.64 vs 10.67 M/s -- neutral, not enabled)
            if constexpr (HELP2) {
                // (below, outside this branch: the first step has no recursion but the same two barriers)
            } else if constexpr (C::WPE == 2 && !IBR && !TEAM && C::MODEL != ALG_MODEL_BICYCLE) {
                // 256-VGPR configurations (one game per SIMD at their batch sizes): all players' operands are read first,
                // the P independent MFMA chains overlap in the matrix pipeline, then all results are written back
                double pv[P][KB1];
#pragma unroll
                for (int i = 0; i < P; i++)
#pragma unroll
                    for (int kb = 0; kb < KB1; kb++) {
                        const double v = L.bw.Pm[i * n * LDP + (colP ? lrow : 0) * LDP + 4 * kb + lq];
                        pv[i][kb] = colP ? v : 0.0;
                    }
                double4_t c1[P], c2[P];
#pragma unroll
                for (int i = 0; i < P; i++) { c1[i] = double4_t{0.0, 0.0, 0.0, 0.0}; c2[i] = double4_t{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
                for (int kb = 0; kb < KB1; kb++)
#pragma unroll
                    for (int i = 0; i < P; i++) c1[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(pv[i][kb], bF[kb], c1[i], 0, 0, 0);
                if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
#pragma unroll
                    for (int i = 0; i < P; i++) c2[i] = di_AT_tile<C>(c1[i], dt, lq);
                } else if constexpr (C::P == 4) {
#pragma unroll
                    for (int i = 0; i < P; i++) c2[i] = p4_AT_tile<C>(c1[i], L.coefn, lq);
                } else if constexpr (C::P == 3) {
#pragma unroll
                    for (int i = 0; i < P; i++) c2[i] = p3g.apply(c1[i], L.coefn, lq);
                } else {
#pragma unroll
                    for (int kb = 0; kb < KB; kb++)
#pragma unroll
                        for (int i = 0; i < P; i++) c2[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[kb], c1[i][kb], c2[i], 0, 0, 0);
                }
                sweep_sync<C>();
#pragma unroll
                for (int i = 0; i < P; i++)
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const int row = lq + 4 * r4;
                        const int slot = (rowok[r4] && lrow < n + (AUGS ? 1 : 0)) ? oPm + i * n * LDP + row * LDP + lrow : oPad;
                        bwb[slot] = c2[i][r4];
                    }
            } else {
#pragma unroll
                for (int i = 0; i < P; i++) {
                    if (IBR && i != ip) continue;
                    if (TEAM && (i % C::NW) != tw) continue;        // this player belongs to another wavefront of the team
                    double4_t c1 = {0.0, 0.0, 0.0, 0.0}, c2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < KB1; kb++) {
                        // n < 16: columns n+1.. of the last k-block read past the row (finite values) and meet zero rows of Fx
                        const double pv = L.bw.Pm[i * n * LDP + (colP ? lrow : 0) * LDP + 4 * kb + lq];
                        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(colP ? pv : 0.0, bF[kb], c1, 0, 0, 0);
                    }
                    if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) c2 = di_AT_tile<C>(c1, dt, lq);
                    else if constexpr (C::P == 4) c2 = p4_AT_tile<C>(c1, L.coefn, lq);
                    else if constexpr (C::P == 3) c2 = p3g.apply(c1, L.coefn, lq);
                    else {
#pragma unroll
                        for (int kb = 0; kb < KB; kb++) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aA[kb], c1[kb], c2, 0, 0, 0);
                    }
                    sweep_sync<C>();            // this player's reads of P_i / s_i are done (single wave: a wait + compiler fence)
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        const int row = lq + 4 * r4;
                        const int slot = (rowok[r4] && lrow < n + (AUGS ? 1 : 0)) ? oPm + i * n * LDP + row * LDP + lrow : oPad;
                        bwb[slot] = c2[r4];
                    }
                }
            }
            bsync();
        }
        ALG_PROF(0)
        // ---- Q-add: the non-zeros of [Q^_i | rx_i] (diagonal, position block, column n), one entry per lane and pass
        if constexpr (HELP2) {
            team_lds_barrier();                           // [F f], the coefficients, the record and every P_i of the step before are in LDS for both wavefronts
            if (k < N - 2) {
                t_half<C>(L, 0, tid);
                value_recursion_half<C>(L, 0, lrow, lq, dt);                // even players here, odd players on wavefront 1
            }
            sweep_sync<C>();
            qam.apply(tid, Rc, L.qdf, bwb + oPm, reg, w, -1);               // Q-add of the even players
            sweep_sync<C>();
            static_assert(!HELP2 || (SYSROW && m == 2 * P && m / 2 <= WAVE / 16), "V rows of one wavefront's players in one pass");
            v_sysrow<C>(L, Rc, k, dt, 2 * (tid >> 4) + 0, tid & 15);       // V rows of the even players' controls (c % P = player)
            player_tail_half<C>(L, Rc, dt, k, N, 0, tid);                   // their s_i, y_i, g_c
            // coefficient entries of A_k' for both wavefronts' closed-loop rows (the table's last readers finished before the first barrier)
            if (tid < 4 * P) { const int kind = tid / P, i = tid % P; L.bw.T[(((kind & 1) ? 3 : 2) * P + i) * n + ((kind >> 1) ? P + i : i)] = coefk[tid]; }
            team_lds_barrier();                           // all players' P_i, V rows and g_c are back
        } else
        qam.apply(tid, Rc, L.qdf, bwb + oPm, reg, w, IBR ? ip : -1);
        constexpr bool TAIL2 = HELP2;                      // team of two: s_i, y_i, g_c were formed per player above
        if constexpr (!AUGS && !TAIL2) {
            for (int e = tid; e < P * n; e += BT) {                         // s_i <- rx_i + A_{k+1}' t_i
                const int i = e / n, r = e % n; const double* ti = &L.bw.t[i * n];
                double v = Rc[R::RX + e];
                if (k < N - 2) v += AT_vec<C>(L.coefn, dt, [&](int rr) { return ti[rr]; }, r);
                L.bw.Pm[i * n * LDP + r * LDP + n] = v;
            }
        }
        bsync();
        ALG_PROF(1)
        // prefetch of the next step's record: issued after the register-hungry MFMA phase, landed by the end of the step
        double pre[RPL];
        // (one-wavefront kernels only: in the C5 loop kernel -- team of four -- the same change raised the SGPR spills from 13 to 34)
        constexpr bool PRE2 = ALG_R6_SADDR >= 2 && C::NW == 1;
        if constexpr (PRE2) {
            // unconditional loads from clamped addresses off a scalar base (step 0 re-requests its own record, nothing lands it): the conditional
            // form -- zeroed registers, two exec-masked branches, each re-reading the record offset from the kernel arguments -- drained vmcnt
            // and exposed two scalar round trips per step
            const int kp = __builtin_amdgcn_readfirstlane(k > 0 ? k - 1 : 0);
            const double* const Rb = as_global(reinterpret_cast<const double*>(uniform_u64(reinterpret_cast<unsigned long long>(G.rec(pr) + (size_t)kp * R::LEN))));
#pragma unroll
            for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; pre[q] = gld(Rb, e < R::LEN_SWEEP ? e : R::LEN_SWEEP - 1); }
        }
        else if (!HELP2 && k > 0) {                                  // (team of two: the helper wavefront fetches the record)
#pragma unroll
            for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; pre[q] = e < R::LEN_SWEEP ? gld(G.rec(pr) + (size_t)(k - 1) * R::LEN, e) : 0.0; }
        }
        // ---- V[c][0..n) = B[:,c]' P_{i(c)},  V[c][n+1+c] = R^_c,  y_i = P_i rd + s_i   (lane = 16 c + col: shifts, no divisions)
        constexpr int CPP = BT / 16;                                   // control rows of V per pass
#pragma unroll
        for (int q = 0; q < (HELP2 ? 0 : (m + CPP - 1) / CPP); q++) {       // (team of two: done per player between the two barriers above)
            const int c = CPP * q + (tid >> 4), col = tid & 15;
            if constexpr (SYSROW) {
                // Row c of the augmented system (double integrator shown; unicycle: coefficient-weighted shifts by P, 2P, 3P) [W | V A_k | g] is a combination of row c of V with itself shifted by m
                // ((V A)[c][j] = V[c][j] + dt V[c][j - m], W[c][j] = dt^2/2 V[c][j] + dt V[c][j + m] + R^ slot), and a 16-lane row of this
                // phase IS row c of V: the lanes form the system's entries from their own V entry and two row shifts, so the column
                // build reads its m entries instead of 3 m entries of V (same FMA sequences as the pattern form: bit-identical)
                v_sysrow<C>(L, Rc, k, dt, c, col);
            } else if (c < m && col < n) {
                const double* Pi = &L.bw.Pm[(c % P) * n * LDP];
                L.bw.V[c * VW + col] = BT_vec<C>(coefk, dt, [&](int rr) { return Pi[rr * LDP + col]; }, c);
            }
        }
        static_assert(P * 16 <= WAVE, "one (player, row) per lane");
        // Team of four: the V rows above occupy wavefront 0 (and one or two rows of wavefront 1); y_i / g_c below and the A' table are
        // independent of them inside this phase, so they run on wavefronts 2 and 3 at the same time instead of behind the V rows on
        // wavefront 0 (same lanes of a 16-lane row, same instructions: bit-identical)
        constexpr int YOFF = (TEAM && C::NT >= 256) ? 128 : 0, TOFF = (TEAM && C::NT >= 256) ? 192 : 0;
        const int ty = tid - YOFF, tT = tid - TOFF;
        if (!TAIL2 && ty >= 0 && (ty >> 4) < P) {
            // rd sits one entry per lane in every 16-lane row and reaches the FMA chain through the DPP row broadcast: one LDS read of
            // rd per lane instead of n (same products, same order: bit-identical to `a += Pr[c] * rd[c]`)
            const int yp = ty >> 4, yr = (ty & 15) < n ? (ty & 15) : n - 1;
            const double* Pr = &L.bw.Pm[yp * n * LDP + yr * LDP];
            const double rdl = Rc[R::RD + yr];
            double a = Pr[n];
            if constexpr (rowdot_split_v<C> && ALG_R6_ROWDOT_CHAINS > 2) rowdot_dpp_chains<n, ALG_R6_ROWDOT_CHAINS, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(a, rdl, [&](int c) { return Pr[c]; });
            else if constexpr (rowdot_split_v<C>) rowdot_dpp_split<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(a, rdl, [&](int c) { return Pr[c]; });
            else rowdot_dpp_g<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(a, rdl, [&](int c) { return Pr[c]; });
            if (!GFUSE && (ty & 15) < n) L.bw.t[yp * n + yr] = a;
            // split recursion: y_i takes the place of s_i (this lane was its only reader): column n of [P_i A_k | y_i] in the next step
            if constexpr (SPLITF) { if ((ty & 15) < n) L.bw.Pm[yp * n * LDP + yr * LDP + n] = a; }
            // g_c = ru_c + B[:,c]' y_i for the controls c of this row's player (c % P == i): the rows of y_i that column c of B touches are
            // shifts away inside the row, so lane 16 i + c finishes g_c here -- no second phase, no trip of y through LDS (BT_vec's expression)
            if constexpr (GFUSE) {
                const int rl = ty & 15;
                double gb;
                if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
                    const double up = row_shift<0x100 + m>(a);
                    gb = 0.5 * dt * dt * a + dt * up;
                } else {
                    const double dn = row_shift<0x110 + P>(a), up = row_shift<0x100 + P>(a), up2 = row_shift<0x100 + 2 * P>(a);
                    const int kind = rl < m ? rl / P : 0;
                    const double vi = kind ? dn : a, vpi = kind ? a : up;
                    gb = 0.5 * dt * (coefk[kind * P + yp] * vi + coefk[(2 + kind) * P + yp] * vpi) + dt * up2;
                }
                if (rl < m && rl % P == yp) L.bw.V[rl * VW + (SYSROW ? m + n : n)] = Rc[R::RU + rl] + gb;
            }
        }
        // coefficient entries of A_k' (state-dependent models)
        if constexpr (C::MODEL == ALG_MODEL_UNICYCLE) {
            if (!HELP2 && tT >= 0 && tT < 4 * P) { const int kind = tT / P, i = tT % P; L.bw.T[(((kind & 1) ? 3 : 2) * P + i) * n + ((kind >> 1) ? P + i : i)] = coefk[tT]; }
        } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
            if (tT >= 0 && tT < 5 * P) {
                const int kind = tT / P, i = tT % P;          // (x,psi) (x,v) (y,psi) (y,v) (psi,v)
                const int colb = (kind == 0 || kind == 2) ? 3 : 2, row = kind < 2 ? i : (kind < 4 ? P + i : 3 * P + i);
                L.bw.T[(colb * P + i) * n + row] = coefk[tT];
            }
        }
        if (!SYSROW && tid < m) L.bw.V[tid * VW + n + 1 + tid] = Rc[R::RHAT + tid];
        bsync();
        ALG_PROF(2)
        // ---- V[c][n] = g_c = ru_c + B[:,c]' (P rd + s)   (double integrator / unicycle: done in the y_i lanes above)
        if constexpr (!GFUSE) {
            if (tid < m) {
                const double* yi = &L.bw.t[(tid % P) * n];
                L.bw.V[tid * VW + n] = Rc[R::RU + tid] + BT_vec<C>(coefk, dt, [&](int rr) { return yi[rr]; }, tid);
            }
            bsync();
        }
        ALG_PROF(3)
        // ---- column-per-lane augmented system [ W | V A_k | g ],  W = diag(R^) + V B: every lane forms its column as the same
        // short sparse combination of row c of the extended V (lane < m: B column + R^ slot; lane < m+n: A column; lane m+n: g slot)
        // (lane layout of the DPP elimination: every 16-lane row carries W's columns in its lanes 0..m-1 and its share of the
        // n + 1 right-hand-side columns behind them, GjLanes)
        constexpr bool FLATGJ = C::MODEL == ALG_MODEL_BICYCLE && P == 4 && C::EXT;
        using GL = typename std::conditional<FLATGJ, GjFlat<m, n + 1>, GjLanes<m, n + 1>>::type;
        const int cidx = GL::column(lane);                 // column of [W | V A_k | g] this lane builds
        const bool rhsl = GL::rhs(lane);                   // ... and whether it is a right-hand side (its solution column is used)
        double col[m];
        {
            int rows[C::NPAT + 1]; double vals[C::NPAT + 1];
            if constexpr (!SYSROW) col_pattern<C>(coefk, dt, cidx, k >= 1, rows, vals);
#pragma unroll
            for (int c = 0; c < m; c++) {
                const double* Vc = &L.bw.V[c * VW];
                double v;
                if constexpr (SYSROW) v = Vc[cidx];
                else {
                    v = vals[0] * Vc[rows[0]];
#pragma unroll
                    for (int t = 1; t < C::NPAT + 1; t++) v = fma(vals[t], Vc[rows[t]], v);
                }
                if (IBR) {
                    if (c % P != ip) v = (cidx == c) ? 1.0 : 0.0;               // unit row: du_c = 0
                    else if (cidx < m && cidx % P != ip) v = 0.0;               // fixed controls of the other players
                }
                col[c] = v;
            }
        }
        ALG_PROF(4)
        if constexpr (FLATGJ) sing |= gj_solve_cols<m>(col); else sing |= gj_solve_cols_dpp<m>(col);
        ALG_PROF(5)
        // ---- K = -Y -> HBM (column-major m x (n+1)) ; [F | f] = [A_k | rd] + B [K | kappa]
        if (rhsl) {
            const int cc = cidx - m;
            // (split recursion: the solved columns Y = -[K | kappa] are used as they are -- the sign goes into the constants that form the
            // A operand P_i B of the next step's product and into the forward sweep's du = -(Y dx + y_0): exact negations, bit-identical)
            if constexpr (!SPLITF) {
#pragma unroll
                for (int c = 0; c < m; c++) col[c] = -col[c];
            }
            if constexpr (SPLITF) {
                // split recursion: column cc of -[K | kappa] is the B operand of the next step's product as it is
#pragma unroll
                for (int c = 0; c < m; c++) { if (!TEAM || (c % C::NW) == tw) L.bw.Fx[c * 16 + cc] = col[c]; }     // (team: every wavefront holds the columns)
            } else {
            // column cc of [A_k | rd]: contiguous in LDS (T row cc, or the record's rd); A_0 is never used (dx_1 = 0)
            const double* acol = (cc < n) ? &L.bw.T[cc * n] : Rc + R::RD;
            // all LDS reads first, then all writes: the compiler cannot prove that the Fx stores do not alias the T / record
            // loads and would otherwise serialise one LDS round trip per row
            // (team: wavefront tw forms the rows r = tw (mod NW); every wavefront holds the solved columns)
            // (team of two: the helper wavefront forms the odd rows)
            constexpr int RS = TEAM ? C::NW : (HELP2 ? 2 : 1);
            double fxv[n];
#pragma unroll
            for (int r = 0; r < n; r++) { if (RS == 1 || (r % RS) == tw) fxv[r] = B_vec<C>(coefk, dt, [&](int c2) { return col[c2]; }, r) + acol[r]; }
#pragma unroll
            for (int r = 0; r < n; r++) {
                if (RS != 1 && (r % RS) != tw) continue;
                if constexpr (n < 16) L.bw.Fx[r * 16 + cc] = fxv[r];         // f rides in tile column n
                else { double* dst = cc < n ? &L.bw.Fx[r * 16 + cc] : &L.bw.fv[r]; *dst = fxv[r]; }   // one store, selected address (no exec-mask flip per row)
            }
            }
        }
        ALG_PROF(9)
        if (C::NC > 0 && tid < C::NC) L.coefn[tid] = coefk[tid];
        if (!HELP2 && k > 0) {
#pragma unroll
            for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; if (e < R::LEN_SWEEP) L.rec[cur ^ 1][e] = pre[q]; }
        }
        ALG_PROF(10)
        // the gains go out last (gfx9 counts loads and stores in one vmcnt: the wait for the prefetched record above should not
        // meet stores that were just issued; measured neutral, the phase profile shows no exposed wait either way)
        asm volatile("" ::: "memory");
        if (!HELP2 && tw == 0 && rhsl) {
#if ALG_R6_SADDR
            // (the step's base address as a scalar pair, ONE lane offset, the m entries of the column at immediate offsets: the compiler keeps the game's
            // base pointers in VGPRs once the scalar file is full, and every store then paid two 64-bit vector adds, a copy and the offset's reload)
            double* const Kg = reinterpret_cast<double*>(uniform_u64(reinterpret_cast<unsigned long long>(G.kgain(pr) + (size_t)k * NK)));
            const unsigned ko = goff((cidx - m) * m);
#pragma unroll
            for (int c = 0; c < m; c++) *reinterpret_cast<double*>(reinterpret_cast<char*>(as_global(Kg) + c) + ko) = col[c];
#else
            double* __restrict__ Kg = G.kgain(pr) + (size_t)k * NK;
#pragma unroll
            for (int c = 0; c < m; c++) gst(Kg, (cidx - m) * m + c, col[c]);
#endif
        }
        bsync();
        ALG_PROF(6)
    }
    if constexpr (HELP2) game_sync();                        // the helper wavefront's gain stores
    if (__builtin_amdgcn_readfirstlane(sing)) return ALG_STATUS_SINGULAR;      // wave-uniform (every lane factors the same matrix)
    if (TEAM && tw != 0) return ALG_STATUS_OK;         // the serial sweeps below belong to wavefront 0 (the caller holds a barrier)
#if defined(ALG_DIR_STOP) && ALG_DIR_STOP == 1
    return ALG_STATUS_OK;
#endif
    // the serial forward and costate sweeps (wavefront 0 of a team): direction_forward_costate below
    ALG_PROF_FLUSH                                          // (profile builds: the backward sweep's phase sums; the sweeps below flush their own)
    return direction_forward_costate<C, IBR>(pr, G0, L, N, dt, lane, reg, ip, primal_l1);
}

// ================================================================================================
// Iterative refinement of the Newton direction (round 4; replaces the backward stability of `lu(core.jac)`, solver_methods.jl:87).
//
// The structured elimination is a block LU without pivoting across blocks: stable only up to the conditioning of its pivot blocks
// R^ + B' P B (controls acting through two integrators, penalties at their ceiling), where UMFPACK's partial pivoting is backward
// stable regardless.  Which rows of J d = -res can carry a residual is known, though: the forward sweep evaluates the dynamics rows
// (dx_{k+1} = A dx_k + B du_k + rd) and the costate sweep the opt-x rows (dlambda_k = Q^ dx_{k+1} + A' dlambda_{k+1} + rx) on the final
// numbers, so both hold to rounding whatever happened to the gains; every error of the elimination surfaces in the opt-u rows
//     rho_{c,k} = R^_c du_{c,k} + B_k[:,c]' dlambda_{i(c),k} + ru_{c,k}.
// dir_urow_residual evaluates them (one flat pass over (step, control)) together with lower bounds of |J|_inf and |d|_inf; when the
// normwise backward error  max |rho| / (|J| |d|)  exceeds Params::refine_tol the direction is corrected by e from  J e = -(0, rho, 0)
// -- the same elimination on the step records with ru <- rho, rx <- 0, rd <- 0 -- written to the (dead) trial buffer and added.  One
// step of this fixed-precision refinement makes the solve backward stable (Skeel) as long as the elimination has any accuracy at
// all; at most Params::refine_max steps are taken.  Well-conditioned solves pay the gate (one pass over du, part of dlambda and
// 2 m + a few record entries per step), nothing else.
// ================================================================================================
struct DirGate { double rho, omega, smax; };      // max |rho|, row-wise max |rho_c| / (|J_c| |d| + |ru_c|), max row scale
template <class C, int K> __device__ __forceinline__ void team_max(double (&v)[K]) {
#pragma unroll
    for (int q = 0; q < K; q++) v[q] = wave_max(v[q]);
    if constexpr (C::NW > 1) {
        __shared__ double tmx[C::NW][K];
        const int w = game_tid() >> 6, l = game_tid() & 63;
        if (l == 0) {
#pragma unroll
            for (int q = 0; q < K; q++) tmx[w][q] = v[q];
        }
        game_sync();
#pragma unroll
        for (int q = 0; q < K; q++) { double r = tmx[0][q]; for (int x = 1; x < C::NW; x++) r = fmax(r, tmx[x][q]); v[q] = r; }
        game_sync();
    }
#pragma unroll
    for (int q = 0; q < K; q++) v[q] = uni(v[q]);
}
// WRITE = false: the gate statistics of the direction in G.z(2) against the step records: one flat pass over the (step, control) pairs,
// five loads each.  WRITE = true: also turns the records into the right-hand side of the correction system (ru <- rho, rx <- 0, rd <- 0).
template <class C, bool IBR, bool WRITE>
__device__ DirGate dir_urow_residual(CPR pr0, const Game& G0, int ip) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    constexpr int n = C::n, m = C::m, P = C::P;
    using R = Rec<C>;
    const int N = phase_int(pr.N), tid = phase_lane();
    const double dt = phase_f64(pr.dt);
    const double* __restrict__ dz = G.z(2);
    double* __restrict__ recs = G.rec(pr);
    // the pair with the largest ratio |rho| / scale is tracked by cross-multiplication: one division per lane at the end
    double rho_m = 0.0, s_m = 0.0, wr = 0.0, ws = 1.0;
    // (measured again in round 6: four items per lane with their loads in flight together -- neutral at C2, like round 4's attempt:
    // profiles/r06_ab_micro_c2.txt; the pass keeps its one-item trips)
    for (int e = tid; e < (N - 1) * m; e += C::NT) {
        const int k = e / m, c = e % m, i = c % P;
        double* Rk = recs + (size_t)k * R::LEN;
        const int ro = k * R::LEN, dlo = n + hl<C>(k, i);                 // (32-bit offsets from the game's buffers: gld / gst)
        const double du = gld(dz, n + hu<C>(k, 0) + uoff<C>(c));
        const double rh = gld(recs, ro + R::RHAT + c), ru = gld(recs, ro + R::RU + c);
        const double bl = BT_vec<C>(Rk + R::COEF, dt, [&](int rr) { return gld(dz, dlo + rr); }, c);
        // |B[:,c]|' |dlambda|: the true row scale (round 6; until round 5 the coefficients kept their signs -- a lower estimate that made the
        // gate's figure up to 259 x conservative on rows whose terms cancel)
        const double bla = BT_vec_abs<C>(Rk + R::COEF, dt, [&](int rr) { return fabs(gld(dz, dlo + rr)); }, c);
        double rho = fma(rh, du, ru) + bl;
        if (IBR && i != ip) rho = 0.0;                                  // unit rows of the other players (du_c = 0)
        const double sc = fabs(rh * du) + fabs(ru) + fabs(bla);         // row scale |J_c| |d| + |ru_c|
        const double ar = fabs(rho);
        if (ar * ws > wr * sc) { wr = ar; ws = sc; }
        rho_m = fmax(rho_m, ar); s_m = fmax(s_m, sc);
        if constexpr (WRITE) gst(recs, ro + R::RU + c, rho);
    }
    if constexpr (WRITE) {
        for (int e = tid; e < (N - 1) * P * n; e += C::NT) gst(recs, (e / (P * n)) * R::LEN + R::RX + e % (P * n), 0.0);
        for (int e = tid; e < (N - 1) * n; e += C::NT) gst(recs, (e / n) * R::LEN + R::RD + e % n, 0.0);
    }
    double v[3] = {rho_m, wr / fmax(ws, 1e-300), s_m};
    team_max<C, 3>(v);
    return DirGate{v[0], v[1], v[2]};
}
// d <- d + e (e in the trial buffer); returns sum |d_primal| (Delta_step), max |d| and the non-finite flag of the corrected direction;
// restores x_1 of the trial buffer, which the correction's forward sweep zeroed.
template <class C>
__device__ void dir_add_correction(CPR pr0, const Game& G0, double& pl1, double& dn, int& bad) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    constexpr int n = C::n, m = C::m;
    const int S = phase_int(pr.S), tid = phase_lane();
    double* __restrict__ dz = G.z(2); double* __restrict__ ez = G.z(1); const double* __restrict__ z0 = G.z(0);
    double s = 0.0, mx = 0.0; int nf = 0;
    for (int e = tid; e < S; e += C::NT) {
        const double v = gld(dz, n + e) + gld(ez, n + e);
        gst(dz, n + e, v);
        if (e % C::b < n + m) s += fabs(v);
        mx = fmax(mx, fabs(v)); nf |= !isfinite(v);
    }
    if (tid < n) ez[tid] = z0[tid];
    s = wave_sum(s); nf = wave_or(nf);
    if constexpr (C::NW > 1) {
        __shared__ double tad[C::NW][2];
        const int w = game_tid() >> 6, l = game_tid() & 63;
        if (l == 0) { tad[w][0] = s; tad[w][1] = (double)nf; }
        game_sync();
        s = 0.0; nf = 0;
        for (int x = 0; x < C::NW; x++) { s += tad[x][0]; nf |= (int)tad[x][1]; }
        game_sync();
    }
    double v[1] = {mx};
    team_max<C, 1>(v);
    pl1 = uni(s); dn = v[0]; bad = __builtin_amdgcn_readfirstlane(nf);
}
// Largest penalty of the game's constraint rows (ALConVal mu): the scale by which the augmented-Lagrangian terms can worsen the
// conditioning of the KKT system.  Only evaluated for directions whose backward error falls between the two tolerances.
template <class C>
__device__ double con_mu_max(CPR pr0, const Game& G0) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    const double* __restrict__ mu = G.mu(pr);
    double mx = 0.0;
    // rows of constraint kinds the problem has (the layout also holds rows for absent collision avoidance / control bounds, whose penalties
    // condition nothing -- and which the fused kernels scale once per solve, not once per outer iteration: penalty_update_unused_rows)
    const int c0 = (C::P > 1 && pr.has_colavoid) ? 0 : pr.col_len, c1 = pr.col_len, c2 = pr.has_ctl ? pr.col_len : pr.col_len + pr.ctl_len;
    for (int e = phase_lane(); e < pr.con_len; e += C::NT) { const bool used = (e >= c0 && e < c1) || e >= c2; mx = fmax(mx, used ? mu[e] : 0.0); }
    double v[1] = {mx};
    team_max<C, 1>(v);
    return v[0];
}
// Newton direction with the refinement gate: what the solver calls.  Every thread of the game's workgroup calls it (team kernels
// included) and leaves with the same status / sum |d_primal|.
// COUNT = false (the step-wise inspection entry alg_newton_direction): the game's `refinements` statistic belongs to the solver paths.
template <class C, bool IBR = false, bool COUNT = true>
__device__ __forceinline__ int refined_direction(CPR pr0, const Game& G0, Lds<C>& L, double reg, int ip, double* primal_l1) {
    // Nothing but the pass counter is live across the sweeps: sum |d_primal| and the two norms of the gate wait in the game's control
    // slots (HBM), the output view is rebuilt from the pass counter.
#ifdef ALG_NO_REFINE          // A/B builds (tests/probes/build_variant.sh): the bare elimination, gate compiled out
    if constexpr (C::NW == 1) return newton_direction<C, IBR>(pr0, G0, L.d, reg, ip, primal_l1);
    else {
        __shared__ double dir_out0[2];
        int st0 = ALG_STATUS_OK; double pl0 = 0.0;
        game_sync();
        if (C::NW >= 4 || help2_v<C, IBR> || team_wave<C>() == 0) st0 = newton_direction<C, IBR>(pr0, G0, L.d, reg, ip, &pl0);
        if (game_tid() == 0) { dir_out0[0] = (double)st0; dir_out0[1] = pl0; }
        game_sync();
        if (primal_l1) *primal_l1 = uni(dir_out0[1]);
        return __builtin_amdgcn_readfirstlane((int)dir_out0[0]);
    }
#endif
    constexpr int TC_PL1 = 10, TC_RHO = 13, TC_OMEGA = 14, TC_SMAX = 15;       // 13 .. 15: alg_get_direction_gate
    constexpr int TC_OMCUR = 16, TC_RHOCUR = 17;                                                      // gate state between correction solves
    static_assert(TC_RHOCUR < TC_LEN, "per-game control slots");
    int st = ALG_STATUS_OK;
    for (int pass = 0;; pass++) {
        CPR pr = phase_params(pr0);
        Game Gd = G0.fresh();                          // view whose delta slot is the sweeps' output buffer:
        if (pass > 0) Gd.zo[2] = Gd.zo[1];             // a correction goes to the trial buffer (dead until the line search rewrites it)
        double pl1s = 0.0;
        if constexpr (C::NW == 1) st = newton_direction<C, IBR>(pr, Gd, L.d, reg, ip, &pl1s);          // solver_methods.jl:84-88
        else {
            // team: the serial sweeps run on wavefront 0; status and sum |d_primal| reach the other wavefronts through LDS
            __shared__ double dir_out[2];
            game_sync();
            // teams of >= 4: backward sweep on the whole team, forward / costate on wavefront 0; team of 2: wavefront 0 does it all
            // (team of two: wavefront 1 enters as the helper of the value recursion and returns with the backward sweep)
            if (C::NW >= 4 || help2_v<C, IBR> || team_wave<C>() == 0) st = newton_direction<C, IBR>(pr, Gd, L.d, reg, ip, &pl1s);
            if (game_tid() == 0) { dir_out[0] = (double)st; dir_out[1] = pl1s; }
            game_sync();
            st = __builtin_amdgcn_readfirstlane((int)dir_out[0]); pl1s = uni(dir_out[1]);
        }
#ifdef ALG_TEST_FAIL_CORRECTION         // test build (tests/test_gpu_refinement.py): every correction solve "fails" after its sweeps have run
        if (pass > 0) st = ALG_STATUS_SINGULAR;
#endif
        const int rmax = phase_int(pr.refine_max);
        if (pass == 0 && rmax <= 0) {                                                     // gate and refinement off: nothing to report
            if (phase_lane() == 0) { double* t0 = G0.fresh().tc(pr); t0[TC_RHO] = 0.0; t0[TC_OMEGA] = 0.0; t0[TC_SMAX] = 0.0; }
            if (primal_l1) *primal_l1 = pl1s;
            return st;
        }
        double* tc = G0.fresh().tc(pr);
        if (pass == 0 && phase_lane() == 0) tc[TC_PL1] = pl1s;       // read back only after a correction (below)
        // a correction solve that fails (singular pivot block or non-finite output on the right-hand side (0, rho, 0)) is dropped: nothing of
        // it has been added yet, the direction of the passes before is valid and is what the solver continues with
        if (st != ALG_STATUS_OK) {
            if (pass == 0) { if (primal_l1) *primal_l1 = pl1s; return st; }
            // The correction's forward sweep zeroed x_1 of its output buffer -- the TRIAL buffer -- and only dir_add_correction, which this path
            // skips, puts it back: the line search reads x_1 of the trial buffer (the fused pass stages it, assemble_pass reads it at k = 0),
            // and an accepted trial BECOMES pdtraj.  Restore it here.
            { const Game Gr = G0.fresh(); const int tr = phase_lane(); if (tr < C::n) Gr.z(1)[tr] = Gr.z(0)[tr]; }
            game_sync();
            st = ALG_STATUS_OK; break;
        }
        game_sync();                                   // the direction is in global memory
        const DirGate gt = dir_urow_residual<C, IBR, false>(pr, Gd, ip);
        // backward error of the whole direction: the row-wise omega of the first solve; after a correction, the residual of the correction
        // system IS the new residual of the whole system, so omega contracts like max |rho| did
        double omega;
        bool stalled = false;                          // a correction that did not at least halve max |rho|: the next one would not either
        if (pass == 0) {
            omega = gt.omega;
            if (phase_lane() == 0) { tc[TC_RHO] = gt.rho; tc[TC_OMEGA] = gt.omega; tc[TC_SMAX] = gt.smax; tc[TC_OMCUR] = gt.omega; tc[TC_RHOCUR] = gt.rho; }
        } else {
            int bad = 0; double pl1, dn;
            dir_add_correction<C>(pr, G0, pl1, dn, bad);
            const double rho_prev = tc[TC_RHOCUR];
            omega = tc[TC_OMCUR] * (gt.rho / fmax(rho_prev, 1e-300));
            stalled = !(uni(gt.rho) < 0.5 * uni(rho_prev));
            game_sync();                               // every lane has read the slots
            if (phase_lane() == 0) { tc[TC_PL1] = pl1; tc[TC_OMCUR] = omega; tc[TC_RHOCUR] = gt.rho; }
            game_sync();
            if (bad) { st = ALG_STATUS_SINGULAR; break; }
        }
        // The tolerance follows the conditioning the penalties bring: a forward error target delta needs a backward error of delta / cond(J), and
        // cond(J) grows with the largest penalty.  tol applies from mu_max >= refine_mu on; below, it is relaxed in proportion, at most 256 x.
        // (Most directions are far below tol: the penalties are only looked at inside the band.  In a homogeneous batch a correction
        // solve delays its game by a whole direction and the launch with it: 68 corrections in 45 056 directions of C2 cost 3 %.)
        // The dense elimination (quadrotor: dense 12 x 12 blocks per player, controls acting through two integrators, rotor costs down to 1e-4;
        // n up to 48) needs a tighter gate and no relaxation: its directions miss the LU's backward error (1e-18) by four orders at row-wise
        // errors of 1e-11 already (tests/test_gpu_fuzz.py::test_direction_backward_error_against_the_arbiter passes from tol / 64 on).
        // (round 6: the gate's row scale is the true |J_c| |d| + |r_c| -- until round 5 an estimate from below made omega up to 259 x conservative
        // and tol / 64 was calibrated on that; with the true scale the same directions need tol / 128 for the 1e-15 normwise bound.  The dense
        // configurations take up to six corrections while each one at least halves the residual: an ill-conditioned quadrotor system (fuzz seed
        // 400051: forward error 5.6e-4 from the bare elimination) contracts by 30 ... 3000 x per correction, tests/probes/r06_dense_gap.py.)
        const double tol = phase_f64(pr.refine_tol) * (C::DENSE ? ALG_DENSE_TOL_FACTOR : 1.0);
        // a stalled correction ends the refinement only near the tolerance (within 2^10 of it: the rounding floor of the rows' own evaluation keeps
        // some directions above tol for good); far above it the sequence is not monotone -- on fuzz seed 400051 a correction that gains nothing is
        // followed by ones that gain orders (tests/probes/r06_seed_solve.py: 8 forced corrections 2e-8 from the arbiter, stop-at-first-stall 1e-4)
        bool done = !(uni(omega) > tol) || pass >= rmax || (stalled && tol > 0.0 && !(uni(omega) > 1024.0 * tol));     // (tol = 0 forces max_steps corrections: the tests' way to count them)
        if (!done && !C::DENSE && !(uni(omega) > 256.0 * tol)) {
            const double mumax = con_mu_max<C>(pr, G0);
            const double relax = fmin(fmax(phase_f64(pr.refine_mu) / fmax(mumax, 1e-300), 1.0), 256.0);
            done = !(uni(omega) > relax * tol);
        }
        if (done) {
            // the common case leaves from the first pass with sum |d_primal| still in a register: no round trip through the control
            // slots (the slot stores above are not waited for; every wavefront of a team has passed the gate's barriers)
            if (pass == 0) { if (primal_l1) *primal_l1 = pl1s; return st; }
            break;
        }
        // rhs of the correction system from the buffer that holds the latest solve (d itself, or the previous correction)
        dir_urow_residual<C, IBR, true>(pr, Gd, ip);
        if (COUNT && phase_lane() == 0) G0.fresh().st(pr)->refinements += 1;
        game_sync();
    }
    game_sync();
    if (primal_l1) *primal_l1 = uni(G0.fresh().tc(phase_params(pr0))[TC_PL1]);
    return st;
}

// residual_jacobian! + regularize_residual_jacobian! into a dense S x S column-major matrix (global_quantities.jl:109-193).
// Parity / inspection entry point; built from the same step records and block functions the solver uses.
template <class C>
__device__ void jacobian_dense(CPR pr, const Game& G, double reg, double* J) {
    constexpr int n = C::n, m = C::m, P = C::P;
    using R = Rec<C>;
    const int N = pr.N, lane = game_tid(); const size_t S = pr.S; const double dt = pr.dt;
    for (size_t e = lane; e < S * S; e += WAVE) J[e] = 0.0;
    game_sync();
    auto at = [&](int r, int c) -> double& { return J[(size_t)c * S + r]; };
    for (int k = 0; k < N - 1; k++) {
        const double* Rc = G.rec(pr) + (size_t)k * R::LEN;
        const double* coefk = Rc + R::COEF;
        const double w = (k + 1 < N - 1) ? dt : 1.0;
        for (int e = lane; e < P * n * n; e += WAVE) {
            const int i = e / (n * n), r = (e / n) % n, c = e % n;
            double qv = qhat_entry<C>(G.Qd(pr), Rc + R::HH, i, r, c, w, reg);
            if constexpr (C::EXT) { if (r == c) qv += Rc[R::RQ + i * n + r]; }
            at(vx<C>(N, i, k) + r, hx<C>(k) + c) = qv;
        }
        for (int c = lane; c < m; c += WAVE) { const int i = c % P, j = c / P; at(vu<C>(N, i, k) + j, hu<C>(k, i) + j) = Rc[R::RHAT + c]; }
        for (int e = lane; e < n * n; e += WAVE) {
            const int r = e / n, c = e % n; const double a = A_entry<C>(coefk, dt, r, c);
            if (k >= 1) {
                at(vd<C>(N, k) + r, hx<C>(k - 1) + c) = a;
                for (int i = 0; i < P; i++) at(vx<C>(N, i, k - 1) + c, hl<C>(k, i) + r) = a;
            }
        }
        for (int e = lane; e < n * m; e += WAVE) {
            const int r = e / m, c = e % m, i = c % P, j = c / P; const double bv = B_entry<C>(coefk, dt, r, c);
            at(vd<C>(N, k) + r, hu<C>(k, i) + j) = bv;
            at(vu<C>(N, i, k) + j, hl<C>(k, i) + r) = bv;
        }
        for (int r = lane; r < n; r += WAVE) {
            at(vd<C>(N, k) + r, hx<C>(k) + r) = -1.0;
            for (int i = 0; i < P; i++) at(vx<C>(N, i, k) + r, hl<C>(k, i) + r) = -1.0;
        }
    }
}


} // namespace alg
