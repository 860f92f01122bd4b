// algames_qt.hpp -- "quad team" kernels: FOUR games per 256-thread workgroup (gfx950).
//
// Why.  With one game per wavefront (algames_device.hpp) the batch of BASELINE config 2 -- 4096 games -- is exactly four
// wavefronts per SIMD, and the launch lasts as long as one game's dependency chain: the serial phases of the Newton direction
// (value recursion -> control system -> pivoted solve -> closed loop, 39 times per sweep) run at 15-20 cycles per instruction on
// a third of the lanes, and removing instructions from them does not shorten the chain (round 3: the DPP elimination cut the
// pivoted solve from 240 to 150 instructions per step -- same throughput).  What shortens it is a different decomposition:
//
//   * a workgroup of four wavefronts owns four games.  Outside the Newton direction every wavefront runs its own game exactly
//     as before (assemble pass, line search, dual update: 64 lanes per game, code of algames_device.hpp, wave-local syncs);
//   * the Newton direction (solver_methods.jl:87, `newton_direction`) is a COLLECTIVE of the four wavefronts over the four
//     games.  Lane layout inside the collective: 16-lane row q = game q of the workgroup, lane l of the row = one state row
//     (positions in lanes 0..5, velocities in lanes 8..13: the partner of a row under A' / B' is lane ^ 8, one DPP row_ror:8).
//     Wavefront i < P carries player i's value matrix [P_i | s_i] for all four games IN REGISTERS, one matrix row per lane, and
//     advances it with v_fmac_f64_dpp row_newbcast products (the gain row K[c][:] sits in lane c of the row and is broadcast
//     inside the FMA: 13 independent accumulators per broadcast, no LDS round trip, no MFMA tile padding); the three players
//     advance concurrently on three SIMDs.  Wavefront P solves the four 6 x 6 control systems with their 13 right-hand sides
//     column-per-lane (DPP elimination, per-game pivoting), writes the gains, and prepares the next step's tables while the
//     player wavefronts work.  Two workgroup barriers per time step.
//   * the forward sweep is one wavefront over four games (gain row per lane, dx broadcast by DPP), the costate sweep the three
//     player wavefronts again.
//
// The arithmetic is the structured elimination of algames_device.hpp (same recursion, same pivot rule, same gains -- stored
// row-major here); sums are formed in another order, so results agree with the one-wavefront kernel to rounding.
//
// Only instantiated for the 3-player planar double integrator without the extended constraint set (BASELINE configs 2 and 4),
// batches that are a multiple of four games.  This header must be compiled with ALG_QT defined (algames_qt.hip): the per-game
// code then takes its lane index from threadIdx.x & 63 and synchronises wave-locally.
#pragma once
#ifndef ALG_QT
#error "algames_qt.hpp needs ALG_QT (quad-team translation unit)"
#endif
#include "algames_device.hpp"

namespace alg {

// ---- DPP helpers ----------------------------------------------------------------------------------------------------------------
// 64-bit value moved by a 32-bit DPP pattern (two v_mov_b32_dpp; the compiler pads the hazards of the builtin itself)
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROR8 = 0x128, DPP_ROR4 = 0x124, DPP_ROR2 = 0x122, DPP_ROR1 = 0x121;
// lane ^ 8 inside the 16-lane row: ds_swizzle SWAP,8 -- on the LDS crossbar, not on the VALU (every VALU instruction, 32-bit DPP
// moves included, costs the SIMD 4.4 cycles: tests/probes/valu_rate.hip; the vector pipe is what the collective is bound by)
__device__ __forceinline__ double partner8(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x201f), hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x201f);
    return __hiloint2double(hi, lo);
}
// sum / or over the 16 lanes of a row (every lane leaves with the result)
__device__ __forceinline__ double row_sum(double v) {
    v += dpp64<DPP_ROR8>(v); v += dpp64<DPP_ROR4>(v); v += dpp64<DPP_ROR2>(v); v += dpp64<DPP_ROR1>(v);
    return v;
}
__device__ __forceinline__ int row_or(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROR8, 0xf, 0xf, false); v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROR4, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROR2, 0xf, 0xf, false); v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROR1, 0xf, 0xf, false);
    return v;
}

// Workgroup barrier that orders LDS only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains vmcnt, which would
// expose the latency of every global load / store in flight (the record prefetch, the gain stores) at each of the two barriers of
// a time step; the data handed over at those barriers lives in LDS.
__device__ __forceinline__ void qt_barrier_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// ordering of one wavefront's own LDS writes and reads (other lanes' entries): compiler fence, LDS executes in order
__device__ __forceinline__ void qt_wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// acc += src(lane L of the row) * mul  -- one instruction line; statements built from these open with s_nop 1 (a VALU write
// followed by a DPP read of the same VGPR needs two wait states and hipcc pads nothing inside an asm statement)
#define QT_FL(acc, src, mul, L) "v_fmac_f64_dpp %[" #acc "], %[" #src "], %[" #mul "] row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
#define QT_FN(acc, src, mul, L) "v_fmac_f64_dpp %[" #acc "], %[" #src "], -%[" #mul "] row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
#define QT_FI(acc, src, mul) "v_fmac_f64_dpp %[" #acc "], %[" #src "], %[" #mul "] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\t"
#define QT_FIN(acc, src, mul) "v_fmac_f64_dpp %[" #acc "], %[" #src "], -%[" #mul "] row_newbcast:%[l] row_mask:0xf bank_mask:0xf\n\t"

// x[j] += k[j](lane L) * mul, j = 0..12: one gain row against one column of P B
template <int L>
__device__ __forceinline__ void qt_fmac13(double (&x)[13], const double (&k)[13], double mul) {
    asm volatile("s_nop 1\n\t"
                 QT_FI(x0, k0, m) QT_FI(x1, k1, m) QT_FI(x2, k2, m) QT_FI(x3, k3, m) QT_FI(x4, k4, m) QT_FI(x5, k5, m) QT_FI(x6, k6, m)
                 QT_FI(x7, k7, m) QT_FI(x8, k8, m) QT_FI(x9, k9, m) QT_FI(x10, k10, m) QT_FI(x11, k11, m) QT_FI(x12, k12, m)
                 : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]),
                   [x7] "+v"(x[7]), [x8] "+v"(x[8]), [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11]), [x12] "+v"(x[12])
                 : [k0] "v"(k[0]), [k1] "v"(k[1]), [k2] "v"(k[2]), [k3] "v"(k[3]), [k4] "v"(k[4]), [k5] "v"(k[5]), [k6] "v"(k[6]),
                   [k7] "v"(k[7]), [k8] "v"(k[8]), [k9] "v"(k[9]), [k10] "v"(k[10]), [k11] "v"(k[11]), [k12] "v"(k[12]),
                   [m] "v"(mul), [l] "n"(L));
}
// a0 + a1 + a2 += sum_j v[j] * s(lane of state row j): the dot product of this lane's matrix row with a vector that lives one
// entry per lane (state rows 0..5 in lanes 0..5, 6..11 in lanes 8..13); three partial sums keep the FMA chains short
__device__ __forceinline__ double qt_rowdot12(const double (&v)[13], double s, double init) {
    double a0 = init, a1 = 0.0, a2 = 0.0;
    asm volatile("s_nop 1\n\t"
                 QT_FL(a0, s, v0, 0) QT_FL(a1, s, v1, 1) QT_FL(a2, s, v2, 2) QT_FL(a0, s, v3, 3) QT_FL(a1, s, v4, 4) QT_FL(a2, s, v5, 5)
                 QT_FL(a0, s, v6, 8) QT_FL(a1, s, v7, 9) QT_FL(a2, s, v8, 10) QT_FL(a0, s, v9, 11) QT_FL(a1, s, v10, 12) QT_FL(a2, s, v11, 13)
                 : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2)
                 : [s] "v"(s), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]),
                   [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]));
    return (a0 + a1) + a2;
}
// a0 + a1 += sum_{j<6} e[j] * s(lane j): position block of Q^_i against dx
__device__ __forceinline__ double qt_rowdot6(const double (&e)[6], double s, double init) {
    double a0 = init, a1 = 0.0;
    asm volatile("s_nop 1\n\t"
                 QT_FL(a0, s, e0, 0) QT_FL(a1, s, e1, 1) QT_FL(a0, s, e2, 2) QT_FL(a1, s, e3, 3) QT_FL(a0, s, e4, 4) QT_FL(a1, s, e5, 5)
                 : [a0] "+v"(a0), [a1] "+v"(a1)
                 : [s] "v"(s), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]));
    return a0 + a1;
}
// one pivot of the two-columns-per-lane elimination: cr[r] -= cw[r](lane L) * pr ; cw[r] -= cw[r](lane L) * pw   (r = 0..5; the
// right-hand-side columns first: they need lane L's W column as it was before this pivot)
template <int L>
__device__ __forceinline__ void qt_elim6(double (&cw)[6], double (&cr)[6], double pw, double pr) {
    asm volatile("s_nop 1\n\t"
                 QT_FIN(r0, w0, pr) QT_FIN(r1, w1, pr) QT_FIN(r2, w2, pr) QT_FIN(r3, w3, pr) QT_FIN(r4, w4, pr) QT_FIN(r5, w5, pr)
                 QT_FIN(w0, w0, pw) QT_FIN(w1, w1, pw) QT_FIN(w2, w2, pw) QT_FIN(w3, w3, pw) QT_FIN(w4, w4, pw) QT_FIN(w5, w5, pw)
                 : [r0] "+v"(cr[0]), [r1] "+v"(cr[1]), [r2] "+v"(cr[2]), [r3] "+v"(cr[3]), [r4] "+v"(cr[4]), [r5] "+v"(cr[5]),
                   [w0] "+v"(cw[0]), [w1] "+v"(cw[1]), [w2] "+v"(cw[2]), [w3] "+v"(cw[3]), [w4] "+v"(cw[4]), [w5] "+v"(cw[5])
                 : [pw] "v"(pw), [pr] "v"(pr), [l] "n"(L));
}

// Scratch instrumentation (-DALG_PHASE_PROF, tests/probes/qt_prof.sh): shader-clock cycles per phase and role, accumulated into the
// res buffer of the row's game (slots 16 + 8 wavefront + j).  Never defined in the product build.
#ifdef ALG_PHASE_PROF
#define QT_PROF_DECL unsigned qp_t_ = (unsigned)__builtin_readcyclecounter(), qp_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QT_PROF(j) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); qp_acc_[j] += t_ - qp_t_; qp_t_ = t_; }
#define QT_PROF_FLUSH if (l == 0) { for (int j_ = 0; j_ < 8; j_++) (base + pr.o_res)[16 + 8 * wv + j_] += (double)qp_acc_[j_]; }
#else
#define QT_PROF_DECL
#define QT_PROF(j)
#define QT_PROF_FLUSH
#endif
// x[6 + c] += v in lane 8 + c of every 16-lane row (c = 0..5) in one statement: the diagonal entries of the velocity rows of Q^_i
// fall into a different register in every lane.  As selects that is a compare + two v_cndmask + an add per register; under a
// literal EXEC mask it is one VALU instruction (the SALU moves are free for the vector pipe).  All lanes are active around the
// statement (wave-uniform control flow only); EXEC is saved and restored.
#define QT_OH(x, v, L) "s_mov_b32 exec_lo, " #L "\n\ts_mov_b32 exec_hi, " #L "\n\tv_add_f64 %[" #x "], %[" #x "], %[" #v "]\n\t"
__device__ __forceinline__ void qt_onehot_vel(double (&Pn)[13], double dp) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %[keep], exec\n\t"
                 QT_OH(p6, dp, 0x01000100) QT_OH(p7, dp, 0x02000200) QT_OH(p8, dp, 0x04000400)
                 QT_OH(p9, dp, 0x08000800) QT_OH(p10, dp, 0x10001000) QT_OH(p11, dp, 0x20002000)
                 "s_mov_b64 exec, %[keep]\n\ts_nop 4"
                 : [keep] "=&s"(keep), [p6] "+v"(Pn[6]), [p7] "+v"(Pn[7]), [p8] "+v"(Pn[8]), [p9] "+v"(Pn[9]), [p10] "+v"(Pn[10]), [p11] "+v"(Pn[11])
                 : [dp] "v"(dp));
}
// c[k] += v in lane k of every 16-lane row (k = 0..5): R^ on the diagonal of W, column-per-lane
__device__ __forceinline__ void qt_onehot_col(double (&c)[6], double v) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %[keep], exec\n\t"
                 QT_OH(c0, v, 0x00010001) QT_OH(c1, v, 0x00020002) QT_OH(c2, v, 0x00040004)
                 QT_OH(c3, v, 0x00080008) QT_OH(c4, v, 0x00100010) QT_OH(c5, v, 0x00200020)
                 "s_mov_b64 exec, %[keep]\n\ts_nop 4"
                 : [keep] "=&s"(keep), [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5])
                 : [v] "v"(v));
}
#undef QT_OH
// ---- shared memory of a quad-team workgroup -----------------------------------------------------------------------------------
template <class C> struct QtLds {                      // live only inside the collective direction
    static constexpr int RS = 96;                      // doubles of a staged record slice (>= Rec::LEN_SWEEP)
    static constexpr int TROW = 19;                    // 18 rows (player, position row) of the Q^ table + one zero row
    // everything of one game behind ONE per-lane base address (the arrays are reached with immediate offsets)
    struct Game {
        double rec[2][RS];                             // [buffer][record entry]
        double tab[2][TROW * 8];                       // position-block rows of Q^_i (diagonal included): [(i 6 + r) 8 + j]
        double Zp[6 * 16], Zv[6 * 16];                 // rows c and 6 + c of [P_i | s_i | y] of the control's player i = c mod 3 (position / velocity row)
        double Kt[6 * 16];                             // gains of the step, row-major [c 16 + j], column 12 = kappa
        double qd[16];                                 // LQR diagonal of the game [i 4 + j], slot 15 = 0
    };
    Game g[4];
};
struct QtShake {                                       // hand-shake words; never aliased by per-game LDS
    int want[4];
    double reg[4];
    int sing[4], badf[4], badc[3][4];
    double pl1[4];
};
template <class C> struct QtBlock {
    union { Lds<C> per[4]; QtLds<C> q; };
    QtShake sh;
};
// the workgroup's block from a wavefront's own Lds (per[] sits at offset 0 of the block)
template <class C> __device__ __forceinline__ QtBlock<C>* qt_block(Lds<C>& mine) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    return reinterpret_cast<QtBlock<C>*>(reinterpret_cast<char*>(&mine) - (size_t)wv * sizeof(Lds<C>));
}

// ================================================================================================================================
// The collective.  Every thread of the workgroup calls it; `wv` = wavefront, role: wv < 3 player wv, wv == 3 control solve.
// Solves J d = -res for the four games of the workgroup from the step records left by their assemble passes (record! with the
// Jacobian regularisation folded in) and writes d into each game's delta buffer.
// ================================================================================================================================
// (not inlined: the collective is a self-contained register-allocation unit -- inlined into the solver, the per-game state that
// is live across the call pushed it into 80 VGPR / 106 SGPR spills inside its loops; as a call, that state is saved once per direction)
template <class C>
__device__ __attribute__((noinline)) void qt_direction(CPR pr0, QtBlock<C>& B) {
    static_assert(C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR && C::P == 3 && C::D == 2 && !C::EXT, "quad-team direction: 3-player planar double integrator");
    constexpr int n = 12, m = 6, P = 3, bb = C::b;
    using R = Rec<C>;
    using Q = QtLds<C>;
    static_assert(R::LEN_SWEEP <= Q::RS && C::NC == 0, "record slice");
    CPR pr = phase_params(pr0);
    QtLds<C>& L = B.q;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    // Role of this wavefront: 0..2 = player, 3 = control solve.  Rotated with the workgroup index: a workgroup's wavefronts land on
    // the CU's four SIMDs in a fixed order, so with fixed roles every workgroup of a CU would put its control wavefront on the
    // same SIMD (four serial control solves sharing one vector pipe while the other SIMDs wait).
    const int wv = __builtin_amdgcn_readfirstlane(((tid >> 6) + (int)blockIdx.x) & 3);
    const int q = lane >> 4, l = lane & 15;                               // game of the workgroup, lane inside its row
    typename QtLds<C>::Game& Lg = L.g[q];                                 // this row's game in LDS
    const int N = phase_int(pr.N);
    const double dt = phase_f64(pr.dt);
    const double hdt2 = 0.5 * dt * dt;
    const int gq = (int)blockIdx.x * 4 + q;
    double* const base = pr.arena + (size_t)gq * pr.stride;               // this row's game
    const double* const recg = base + pr.o_rec;
    double* const kg = base + pr.o_kgain;
    double* const dz = base + pr.o_z2;
    const bool pos = l < 6, vel = l >= 8 && l < 14, valid = pos || vel;
    const int r = pos ? l : (vel ? l - 2 : 0);                            // state row of this lane (0 on idle lanes)
    const double regq = B.sh.reg[q];
    const double dtv = vel ? dt : 0.0;
    // The two roles run through the same sequence of workgroup barriers -- S1, S2, (B2, B1) per backward step, G, F, one per costate
    // step, Z -- in two separate code paths, so that neither role's registers are live in the other's loops.
    if (wv < 3) {
        // ======================================================================================================== player wavefront i
        const int i = wv;
        // LQR weight of this lane's row for player i (diagonal of Q^_i outside the position block)
        const double qown = (valid && (r % P) == i) ? (pr.lqr + (size_t)gq * pr.lqr_stride)[i * 4 + r / P] : 0.0;
        const int lc = l < 6 ? l : 0;
        QT_PROF_DECL
        __syncthreads();                                                       // S1
        __syncthreads();                                                       // S2
        // row r of [P_i | s_i] (13 registers), y = P_i rd + s_i of the step just assembled
        double Pc[13], yv = 0.0;
#pragma unroll
        for (int j = 0; j < 13; j++) Pc[j] = 0.0;
        QT_PROF(6)
        for (int k = N - 2; k >= 0; k--) {
            const int buf = k & 1;
            const double w = (k + 1 < N - 1) ? dt : 1.0;
            const double* Rc = Lg.rec[buf];
            // ---- value recursion in the split form  [P | s] [[F f],[0 1]] = [P A | y] + (P B) [K | kappa],  F = A + B K, f = rd + B kappa,
            // y = P rd + s (left by the previous step), then A' on the rows (partner lane) -- all on this lane's row.  The gain rows
            // arrive scaled by dt (Kt = dt K), so the column of P B is formed without its factor dt: one FMA each.
            if (k < N - 2) {
                double Kr[13];                                         // gain row c = l (lanes 0..5 are the broadcast sources)
                { const double* Kq = &Lg.Kt[lc * 16];
#pragma unroll
                  for (int j = 0; j < 13; j++) Kr[j] = Kq[j]; }
                double PB[6];
#pragma unroll
                for (int c = 0; c < 6; c++) PB[c] = fma(0.5 * dt, Pc[c], Pc[6 + c]);         // (P B)[r][c] / dt
#pragma unroll
                for (int c = 0; c < 6; c++) Pc[6 + c] = fma(dt, Pc[c], Pc[6 + c]);          // P A: velocity columns take dt x position columns
                Pc[12] = yv;
                qt_fmac13<0>(Pc, Kr, PB[0]); qt_fmac13<1>(Pc, Kr, PB[1]); qt_fmac13<2>(Pc, Kr, PB[2]);
                qt_fmac13<3>(Pc, Kr, PB[3]); qt_fmac13<4>(Pc, Kr, PB[4]); qt_fmac13<5>(Pc, Kr, PB[5]);
            }
            QT_PROF(0)
            // X = Pc (zero at the terminal step).  Per column: the partner row's entry (lane ^ 8), then
            //   P_new[pos row] = X + E,  P_new[vel row] = X + dt Xp + dq e_r   (A' on the velocity rows, + Q^_i; column 12: + rx_i)
            double E[6];
            { const double* Tq = &Lg.tab[buf][(pos ? (i * 6 + l) : 18) * 8];
#pragma unroll
              for (int j = 0; j < 6; j++) E[j] = Tq[j]; }
            const double rxv = Rc[R::RX + i * n + r];
#pragma unroll
            for (int j = 0; j < 13; j++) {
                const double xp = partner8(Pc[j]);
                double pn = fma(dtv, xp, Pc[j]);
                if (j < 6) pn += E[j];
                else if (j == 12) pn += rxv;
                Pc[j] = pn;
            }
            qt_onehot_vel(Pc, regq + w * qown);                       // diagonal of the velocity rows (lane 8 + c, column 6 + c)
            // y = P_new rd + s_new
            const double rdv = Rc[R::RD + r];
            yv = qt_rowdot12(Pc, rdv, Pc[12]);
            // ---- the rows the control system is built from: rows c and 6 + c of the player that owns control c = l mod 3
            // ([P_new | s_new | y], the velocity row's y with ru_c / dt folded in: g_c / dt = ru_c / dt + dt/2 y_c + y_{6+c})
            if (valid && (r % P) == i) {
                double* Zq = pos ? &Lg.Zp[l * 16] : &Lg.Zv[(l - 8) * 16];
#pragma unroll
                for (int j = 0; j < 13; j++) Zq[j] = Pc[j];
                Zq[13] = pos ? yv : fma(1.0 / dt, Rc[R::RU + (l - 8)], yv);
            }
            QT_PROF(1)
            qt_barrier_lds();                                                  // B2: the control systems are in LDS
            qt_barrier_lds();                                                  // B1: gains of step k, record / table of step k - 1
            QT_PROF(2)
        }
        __syncthreads();                                                       // G: the gains are in HBM / L2
        // ---- forward sweep on wavefront 0: du = kappa + K dx (gain row c in lane c, dx broadcast by DPP), dx+ = A dx + B du + rd
        if (wv == 0) {
            if (valid) dz[r] = 0.0;
            double dxv = 0.0, pl1 = 0.0; int bad = 0;
            auto fload = [&](int kk, double (&Kr)[13], double& rdv) {
                const int kc = kk < N - 1 ? kk : N - 2;
#pragma unroll
                for (int j = 0; j < 13; j++) Kr[j] = kg[(size_t)kc * 96 + lc * 16 + j];
                rdv = recg[(size_t)kc * R::LEN + R::RD + r];
            };
            auto fstep = [&](int k, double (&Kr)[13], double& rdv) {          // consumes (Kr, rdv) of step k, refills them for step k + 2
                const double duv = qt_rowdot12(Kr, dxv, Kr[12]);             // lanes 0..5: du_c
                const double rdk = rdv;
                fload(k + 2, Kr, rdv);
                const double dxp = partner8(dxv), dup = partner8(duv);
                double dxn = pos ? fma(dt, dxp, dxv) + hdt2 * duv : fma(dt, dup, dxv);
                dxn += rdk;
                if (pos) { pl1 += fabs(duv); bad |= !isfinite(duv); dz[n + k * bb + n + (l % P) * 2 + l / P] = duv; }
                if (valid) { pl1 += fabs(dxn); bad |= !isfinite(dxn); dz[n + k * bb + r] = dxn; }
                dxv = valid ? dxn : 0.0;
            };
            // gains and rd two steps ahead (a load issued one step ahead returned only after the step: the sweep ran at memory latency)
            double Ka[13], Kb[13], rda, rdb;
            fload(0, Ka, rda); fload(1, Kb, rdb);
            for (int k = 0; k < N - 1; k += 2) {
                fstep(k, Ka, rda);
                if (k + 1 < N - 1) fstep(k + 1, Kb, rdb);
            }
            const double sm = row_sum(valid ? pl1 : 0.0); const int bq = row_or(valid ? bad : 0);
            if (l == 0) { B.sh.pl1[q] = sm; B.sh.badf[q] = bq; }
        }
        __syncthreads();                                                       // F
        QT_PROF(3)
        // ---- costate sweep: dlambda_{i,k} = rx_{i,k+1} + Q^_{i,k+1} dx_{k+1} + A' dlambda_{i,k+1}
        {
            double dlv = 0.0; int bad = 0;
            auto dxload = [&](int kk) { return dz[n + (kk > 0 ? kk : 0) * bb + r]; };        // dx_{kk+1}
            auto cstep = [&](int k, double& dxq) {                                          // consumes dx_{k+1}, refills for step k - 2
                const int buf = k & 1;
                const double w = (k + 1 < N - 1) ? dt : 1.0;
                const double dxv = valid ? dxq : 0.0;
                dxq = dxload(k - 2);
                const double* Rc = Lg.rec[buf];
                double E[6];
                { const double* Tq = &Lg.tab[buf][(pos ? (i * 6 + l) : 18) * 8];
#pragma unroll
                  for (int j = 0; j < 6; j++) E[j] = Tq[j]; }
                double acc = Rc[R::RX + i * n + r];
                if (vel) acc = fma(regq + w * qown, dxv, acc);
                acc = qt_rowdot6(E, dxv, acc);
                const double dlp = partner8(dlv);
                if (k < N - 2) acc += fma(dtv, dlp, dlv);                  // A' dlambda_{k+1}: own row + dt x position partner
                dlv = valid ? acc : 0.0;
                if (valid) { dz[n + k * bb + n + m + i * n + r] = acc; bad |= !isfinite(acc); }
                QT_PROF(4)
                qt_barrier_lds();                                              // one per costate step
                QT_PROF(5)
            };
            double dxa = dxload(N - 2), dxb = dxload(N - 3);
            for (int k = N - 2; k >= 0; k -= 2) {
                cstep(k, dxa);
                if (k - 1 >= 0) cstep(k - 1, dxb);
            }
            const int bq = row_or(valid ? bad : 0);
            if (l == 0) B.sh.badc[i][q] = bq;
        }
        QT_PROF_FLUSH
    } else {
        // ======================================================================================================== control wavefront
        // ---- staging of a record slice: HBM -> registers -> LDS, 16 lanes per game, two steps ahead (one step is shorter than the
        // latency of a global load while every CU streams)
        constexpr int RPL = Q::RS / 16;
        // (addresses are re-derived from an opaque copy of the lane index at every use: hoisted out of the step loop they would
        // occupy two dozen registers and push the control solve into scratch)
        auto olane = [&]() { int v = l; asm volatile("" : "+v"(v)); return v; };
        auto rec_load = [&](int k, double (&v)[RPL]) {
            static_assert(Q::RS <= R::LEN, "a staged slice never reads past its record");
            const double* src = recg + (size_t)k * R::LEN + olane();
#pragma unroll
            for (int t = 0; t < RPL; t++) v[t] = src[16 * t];
        };
        auto rec_land = [&](int buf, const double (&v)[RPL]) {
            double* dst = &Lg.rec[buf][olane()];
#pragma unroll
            for (int t = 0; t < RPL; t++) dst[16 * t] = v[t];
        };
        // ---- table of the position-block rows of Q^_i (qhat_entry of algames_device.hpp; entry (i, r, j), r, j < 6), built one step
        // ahead.  Work item = (player i, players jr, jc of the row / column): its four entries (row jr + 3 ar, column jc + 3 ac) are
        // +- the three numbers H[base + ar + ac] of one block of the record (pairblock's sign pattern), the diagonal ones (jr == jc,
        // ar == ac) also take reg + w q.  27 items per game = two passes of the row's 16 lanes; per pass five constants.
        constexpr int TP = 2;
        int tsrc[TP], tdst[TP]; double tsg[TP], tq0[TP], tq1[TP];     // record source, table destination, sign, own LQR weights of the two diagonal entries (< 0: not diagonal)
#pragma unroll
        for (int t = 0; t < TP; t++) {
            const int e = l + 16 * t;
            int so = R::HH, dst = (Q::TROW - 1) * 8; double sg = 0.0, q0 = -1.0, q1 = -1.0;
            if (e < 27) {
                const int i = e / 9, jr = (e / 3) % 3, jc = e % 3;
                if (jr == i && jc == i) { so = R::HD + 3 * i; sg = 1.0; }
                else if (jr == i) { so = R::HH + 3 * pairq<C>(i, jc); sg = -1.0; }
                else if (jc == i) { so = R::HH + 3 * pairq<C>(i, jr); sg = -1.0; }
                else if (jr == jc) { so = R::HH + 3 * pairq<C>(i, jr); sg = 1.0; }
                if (jr == jc) {
                    const double* qdg = pr.lqr + (size_t)gq * pr.lqr_stride;
                    q0 = (jr == i) ? qdg[i * 4 + 0] : 0.0; q1 = (jr == i) ? qdg[i * 4 + 1] : 0.0;
                }
                dst = (i * 6 + jr) * 8 + jc;
            }
            tsrc[t] = so; tdst[t] = dst; tsg[t] = sg; tq0[t] = q0; tq1[t] = q1;
        }
        auto tab_build = [&](int buf, double w) {
#pragma unroll
            for (int t = 0; t < TP; t++) {
                const double* hs = &Lg.rec[buf][tsrc[t]];
                const double h0 = tsg[t] * hs[0], h1 = tsg[t] * hs[1], h2 = tsg[t] * hs[2];
                const double d0 = tq0[t] >= 0.0 ? fma(w, tq0[t], regq) : 0.0, d1 = tq1[t] >= 0.0 ? fma(w, tq1[t], regq) : 0.0;
                double* td = &Lg.tab[buf][tdst[t]];
                if (l + 16 * t < 27) { td[0] = h0 + d0; td[3] = h1; td[24] = h1; td[27] = h2 + d1; }
            }
        };
        { const double* qdg = pr.lqr + (size_t)gq * pr.lqr_stride; Lg.qd[l] = l < P * 4 ? qdg[l] : 0.0; }
        if (l < 8) { Lg.tab[0][18 * 8 + l] = 0.0; Lg.tab[1][18 * 8 + l] = 0.0; }
        QT_PROF_DECL
        __syncthreads();                                                       // S1
        double pa[RPL], pb[RPL];
        rec_load(N - 2, pa);
        rec_land((N - 2) & 1, pa);
        qt_wave_lds_fence();
        tab_build((N - 2) & 1, 1.0);                                           // knot N: terminal stage weight 1
        rec_load(N >= 3 ? N - 3 : 0, pa); rec_load(N >= 4 ? N - 4 : 0, pb);
        int sing = 0;
        __syncthreads();                                                       // S2
        QT_PROF(7)
        // one backward step of the control wavefront; `pre` holds record k - 1 and is refilled with record k - 3
        auto bstep = [&](int k, double (&pre)[RPL]) {
            const int buf = k & 1;
            // during the players' phase: land record k - 1, build its table, request record k - 3
            if (k > 0) {
                rec_land(buf ^ 1, pre);
                qt_wave_lds_fence();
                tab_build(buf ^ 1, dt);
            }
            rec_load(k >= 3 ? k - 3 : 0, pre);
            QT_PROF(0)
            qt_barrier_lds();                                                  // B2
            QT_PROF(1)
            // ---- the four control systems / dt, column-per-lane: lane j < 13 right-hand side j ([U A | g]), lane j < 6 also W's column j,
            // W = diag(R^) + U B, U = B' P_new / dt = dt/2 (row c) + (row 6 + c) of the player owning control c.  Lane j reads column
            // j of the six row pairs and the column of the other half (j +- 6); lane 12 reads the y column, which yields g
            double cw[6], cr[6];
            { const int lo = olane();
              const int jj = lo < 12 ? lo : 13, jo = lo < 6 ? lo + 6 : (lo < 12 ? lo - 6 : 13);
              const double dto = (lo >= 6 && lo < 12) ? dt : 0.0;
              const double* zp = &Lg.Zp[jj]; const double* zv = &Lg.Zv[jj]; const double* zpo = &Lg.Zp[jo]; const double* zvo = &Lg.Zv[jo];
#pragma unroll
              for (int c = 0; c < 6; c++) {
                  const double uo = fma(0.5 * dt, zp[c * 16], zv[c * 16]), ut = fma(0.5 * dt, zpo[c * 16], zvo[c * 16]);
                  cr[c] = fma(dto, ut, uo);
                  cw[c] = lo < 6 ? fma(dt, ut, hdt2 * uo) : 0.0;
              }
              qt_onehot_col(cw, Lg.rec[buf][R::RHAT + (lo < 6 ? lo : 0)] * (1.0 / dt)); }
            auto pivot = [&](auto Ctag) {
                constexpr int Cc = decltype(Ctag)::value;
                if constexpr (Cc + 1 < m) {
                    double oth;
                    constexpr int NB = m - Cc - 1;
                    const double* b = &cw[Cc + 1];
                    if constexpr (NB == 1) oth = fabs(b[0]);
                    else if constexpr (NB == 2) asm("v_max_f64 %0, |%1|, |%2|" : "=v"(oth) : "v"(b[0]), "v"(b[1]));
                    else if constexpr (NB == 3) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|" : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]));
                    else if constexpr (NB == 4) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|" : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
                    else asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|\n\tv_max_f64 %0, %0, |%5|"
                             : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]));
                    const unsigned long long need = __builtin_amdgcn_ballot_w64(oth > fabs(cw[Cc]));
                    if (need & (0x0001000100010001ull << Cc)) {
                        // some game needs a row exchange: lane Cc of its row decides, the row follows (per-lane selects)
                        int piv = Cc; double best = fabs(cw[Cc]);
#pragma unroll
                        for (int rr = Cc + 1; rr < m; rr++) { const double v = fabs(cw[rr]); if (v > best) { best = v; piv = rr; } }
                        piv = __builtin_amdgcn_update_dpp(0, piv, 0x150 + Cc, 0xf, 0xf, false);
#pragma unroll
                        for (int rr = Cc + 1; rr < m; rr++) {
                            const bool sw = piv == rr;
                            const double a = cw[Cc], b2 = cw[rr], c3 = cr[Cc], d4 = cr[rr];
                            cw[Cc] = sw ? b2 : a; cw[rr] = sw ? a : b2; cr[Cc] = sw ? d4 : c3; cr[rr] = sw ? c3 : d4;
                        }
                    }
                }
                double pvt;
                asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(cw[Cc]), "n"(Cc));
                if (!(fabs(pvt) > 0.0) || !isfinite(pvt)) sing = 1;
                const double rp = fast_rcp(pvt);
                const double pw = cw[Cc] * rp, prr = cr[Cc] * rp;
                qt_elim6<Cc>(cw, cr, pw, prr);
                cw[Cc] = pw; cr[Cc] = prr;
            };
            pivot(std::integral_constant<int, 0>{}); pivot(std::integral_constant<int, 1>{}); pivot(std::integral_constant<int, 2>{});
            pivot(std::integral_constant<int, 3>{}); pivot(std::integral_constant<int, 4>{}); pivot(std::integral_constant<int, 5>{});
            // gains K = -Y: rows scaled by dt to LDS for the players' next product, plain rows to HBM for the forward sweep
            { const int lo = olane();
              if (lo < 13) {
                double* kq = &Lg.Kt[lo]; double* kh = kg + (size_t)k * 96 + lo;
#pragma unroll
                for (int c = 0; c < 6; c++) { kq[c * 16] = -dt * cr[c]; kh[c * 16] = -cr[c]; }
              } }
            QT_PROF(2)
            qt_barrier_lds();                                                  // B1
            QT_PROF(3)
        };
        for (int k = N - 2; k >= 0; k -= 2) {
            bstep(k, pa);
            if (k - 1 >= 0) bstep(k - 1, pb);
        }
        __syncthreads();                                                       // G
        { const int sg = row_or(sing); if (l == 0) B.sh.sing[q] = sg; }
        // while wavefront 0 runs the forward sweep: first record / table of the costate sweep
        rec_load(N - 2, pa);
        rec_land((N - 2) & 1, pa);
        qt_wave_lds_fence();
        tab_build((N - 2) & 1, 1.0);
        rec_load(N >= 3 ? N - 3 : 0, pa); rec_load(N >= 4 ? N - 4 : 0, pb);
        __syncthreads();                                                       // F
        QT_PROF(4)
        auto cstep = [&](int k, double (&pre)[RPL]) {
            const int buf = k & 1;
            if (k > 0) {
                rec_land(buf ^ 1, pre);
                qt_wave_lds_fence();
                tab_build(buf ^ 1, dt);
            }
            rec_load(k >= 3 ? k - 3 : 0, pre);
            QT_PROF(5)
            qt_barrier_lds();                                                  // one per costate step
            QT_PROF(6)
        };
        for (int k = N - 2; k >= 0; k -= 2) {
            cstep(k, pa);
            if (k - 1 >= 0) cstep(k - 1, pb);
        }
        QT_PROF_FLUSH
    }
    __syncthreads();                                                           // Z
}

// Newton direction of ONE game as seen by its wavefront: posts the request, joins the collective, returns the game's status and
// sum |d_primal|.  want = 0: the wavefront only helps (its game needs no direction); returns -1 when no game of the workgroup
// wanted one (everybody is draining: time to leave).
template <class C>
__device__ int qt_direction_call(CPR pr, Lds<C>& mine, int want, double reg, double* primal_l1) {
    QtBlock<C>& B = *qt_block<C>(mine);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((threadIdx.x & 63) == 0) { B.sh.want[wv] = want; B.sh.reg[wv] = reg; }
    __syncthreads();
    const int any = B.sh.want[0] | B.sh.want[1] | B.sh.want[2] | B.sh.want[3];
    if (!__builtin_amdgcn_readfirstlane(any)) return -1;
    qt_direction<C>(pr, B);
    const int bad = B.sh.sing[wv] | B.sh.badf[wv] | B.sh.badc[0][wv] | B.sh.badc[1][wv] | B.sh.badc[2][wv];
    if (primal_l1) *primal_l1 = uni(B.sh.pl1[wv]);
    const int st = __builtin_amdgcn_readfirstlane(bad) ? ALG_STATUS_SINGULAR : ALG_STATUS_OK;
    __syncthreads();                                    // the hand-shake words may be rewritten by the next call
    return st;
}
// A wavefront whose game is finished keeps helping until every game of the workgroup is
template <class C>
__device__ __forceinline__ void qt_drain(CPR pr, Lds<C>& mine) {
    while (qt_direction_call<C>(pr, mine, 0, 0.0, nullptr) >= 0) {}
}

}  // namespace alg
