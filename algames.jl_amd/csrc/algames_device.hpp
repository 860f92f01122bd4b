// algames_device.hpp -- gfx950 device code of the batched ALGAMES Newton / augmented-Lagrangian path.
//
// Execution model: ONE GAME PER WAVEFRONT (workgroup = 64 threads = 1 wave).  Games are independent
// (SURVEY.md 8(e)), so every wave runs its own solver state machine; no inter-workgroup traffic.
//
//  * assemble pass (residual / statistics / line-search trials): phase A, work item = (knot k, player i): Jacobian
//    coefficients and pair / wall / circle terms; phase B, work item = one residual row of one step (flat loops over all
//    rows, operands straight from HBM/L2); wave shuffles give ||.||_1 and the violation maxima, and (for the Newton
//    direction) one compact "step record" per knot is left in HBM: Jacobian coefficients, pair Hessian blocks, R^, and
//    the residual rows.
//  * Newton direction: the KKT system of solver_methods.jl:87 is never materialised.  Its block-tridiagonal
//    structure is eliminated by a structured block LU in the order (u_k via R^, lambda_k via -I,
//    x_{k+1} via an m x m pivoted solve) -- a game-theoretic Riccati sweep.  Backward over k: the per-player value
//    matrices [P_i | s_i] (n x (n+1), LDS) are advanced with a chain of v_mfma_f64_16x16x4_f64 per player
//    ([P_i F | P_i f + s_i]; the sparse A' is then applied on the result tile in registers, or by a second MFMA product) followed by a table-driven sparse add of [Q^_i | rx_i]; the m x m control
//    system with its n + 1 right-hand sides is solved by a column-per-lane Gauss-Jordan with partial pivoting (lane c
//    owns column c, the pivot column is broadcast with v_readlane: wave-uniform pivots); gains go to HBM (m (n+1)
//    doubles per step instead of the b^2 + b p n of a dense block LU).  Then a forward sweep for (dx, du) and a
//    backward costate sweep for dlambda.  Records and gains are prefetched one step ahead into double-buffered LDS slots.
//
// Reference citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
#include <type_traits>
#include <utility>
#include "../../include/algames_hip.h"

// Build switches that remain (tests/probes/build_variant.sh builds A/B variants of the library with them): the numeric tunables
// ALG_SWEEP_DEPTH / ALG_SWEEP_DEPTH_W2 / ALG_FWDW_DEPTH (prefetch rings of the sweeps) and ALG_RDG_W2 / ALG_RDG_W4 (coefficient groups of the
// row-broadcast chains), and the instrumentation builds ALG_PHASE_PROF (per-phase cycle sums), ALG_ISA_MARK (phase markers in the ISA),
// ALG_DIR_STOP (direction cut after a sweep, for per-sweep counters) and ALG_NO_REFINE (refinement gate compiled out).  The A/B switches
// of rounds 3 - 4 whose outcome is recorded in DESIGN.md section 8 were resolved to their shipped side in round 5.

namespace alg {

// Tunables of round 5 (measured: DESIGN.md section 4, "Chunks of the fused trial pass" / "step sizes in groups")
constexpr int FT_DI3 = 13;     // time steps per chunk of the fused trial pass, 3-player double integrator (the chunk buffers fill the LDS that sixteen games per CU leave)
constexpr int FT_UNI3 = 15;    // 3-player unicycle: twelve games per CU (3 x 152 VGPRs per SIMD) leave 13.3 KB each; N = 30 is two chunks
constexpr int FT_UNI4 = 13;    // 4-player unicycle: the chunk buffers stay under the direction's 18.2 KB
constexpr int LS_CAP = 3200;   // doubles of LDS for [z | dz] of a line search (LsLds)
constexpr int LS_SCAP = 3520;  // doubles of LDS for the group pass's Jacobian coefficients and pair-gradient tables (teams of four)
constexpr int LS_NA = 4;       // step sizes per group pass of the team kernels' line search (LsMulti, algames_assemble.hpp)

constexpr int WAVE = 64;
constexpr int MAXP = 10;    // alphax_dual caps p <= 10 (options.jl:68)
constexpr int MAXM = 32;
constexpr int HIST_MAX = ALG_HIST_MAX;

// Everything that is shared by the games of a handle; passed to kernels by value.
constexpr int TC_LEN = 32;       // per-game control slots (G.tc): solver scalars 0..7, t_elap accumulator and start stamp 8, 9, refinement gate 10..17
struct Params {
    int model, p, d, N, n, m, mi, ni, S, b, traj_len, npair, col_len, ctl_len, con_len, B;
    double dt;
    alg_options opt;
    int has_colcost, has_colavoid, has_ctl, lqr_per_game;
    double cc_radius[MAXP], cc_mu[MAXP];
    // collision avoidance of the ordered pair (i, j): radius of its CollisionConstraint and presence (bit j of ca_mask[i]);
    // add_collision_avoidance!(game_con, i, j, radius) adds single pairs, the vector form all of them with r_i + r_j
    // (constraints_methods.jl:5-43).  A pair whose bit is clear evaluates to c = 0 with a zero Jacobian: inert everywhere.
    double ca_pair_r[MAXP * MAXP];
    unsigned ca_mask[MAXP];
    double umax[MAXM], umin[MAXM];
    int hist_max;
    int refine_max;         // iterative refinement of the Newton direction: correction solves allowed per direction (0 = off)
    double refine_tol;      // ... taken while the row-wise backward error of the opt-u rows exceeds this (alg_set_refinement)
    double refine_mu;       // ... relaxed up to 256 x in proportion while the game's largest penalty stays below this
    int kscratch_len;       // per game doubles of gain scratch
    int rec_len;            // per game doubles of step records
    int ls_multi;           // line search of the team kernels: further step sizes are tried LsMulti::NA at a time once the first one was rejected (0 = one after another)
    unsigned long long ibr_ctl_rows[MAXP];   // control-bound rows counted by control_violation(game_con, pdtraj, i) (violations.jl:69-82)
    // extended ingredient set (Cfg::EXT instantiations only)
    int ext, has_sb, nwall, ncirc, sb_len, wall_len, circ_len;
    int nwall3, ncyl, wall3_len, cyl_len, ca_dim;   // 3-D half (Cfg::PD == 3 only): Wall3D, Cylinder, spherical collision avoidance (ca_dim = 3)
    double lf, lr;          // BicycleGame(lf, lr), bicycle.jl:15
    double qmass;           // QuadrotorGame(; mass), quadrotor.jl:20 (0.5 unless alg_set_quadrotor)
    // per-player wall / circle sets (add_wall_constraint!(game_con, i, walls), add_circle_constraint!(game_con, i, ...),
    // constraints_methods.jl:121-139, 161-187): bit w of wall_mask[i] = table entry w constrains player i.  A row whose bit is
    // clear evaluates to c = 0 with a zero Jacobian -- exactly inert in the AL terms, the violations and the dual update.
    unsigned wall_mask[MAXP], circ_mask[MAXP];
    // the same for the 3-D sets: add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) / (..., i, ::Vector{CylinderWall}) (constraints_methods.jl:208-247, 256-299)
    unsigned wall3_mask[MAXP], cyl_mask[MAXP];
    // ---- device memory of the handle (filled in by the host; see the "Per-game data view" section) ----------------------
    // main arena: B x stride doubles; one contiguous, 128-byte aligned chunk per game holding every per-game array at the
    // offsets below (doubles, multiples of 16): [pdtraj | trial | delta | x0 | res | rec | kgain | tcache | stats | mpc totals]
    double* arena;
    int stride, o_z1, o_z2, o_x0, o_res, o_rec, o_kgain, o_tc, o_st, o_mpc;      // o_tc: TC_LEN control slots per game
    // constraint arena: B x con_stride doubles, per game [lam | mu | vals] (con_pad doubles each; re-created when extended
    // constraints are added)
    double* con;
    int con_stride, con_pad;
    // LQR block [Qd (p ni) | xf (p ni) | Rd (p mi) | uf (p mi)] (compact, own indices): per game (lqr_stride > 0) or shared (0)
    const double* lqr;
    int lqr_stride;
    // constants of the extended constraints, shared by all games:
    // [x_max (p n) | x_min (p n) | walls x1 y1 x2 y2 xv yv (6 ALG_MAX_WALLS) | circles xc yc r (3 ALG_MAX_CIRCLES)
    //  | 3-D walls p1 p2 p3 v (12 per wall, ALG_MAX_WALLS) | cylinders p (3) axis l r (6 per cylinder, ALG_MAX_CIRCLES)]
    const double* extc;
    alg_record* hist;       // B x hist_max Statistics records
    int* ho_queue;          // straggler hand-off (alg_set_handoff): [count | game indices of the parked games] (B + 1 ints; nullptr while the feature is off)
};

// The handle's parameters are read straight from the kernel-argument segment (constant address space, scalar loads): the
// kernels take `Params` by value as their FIRST argument and the device code refers to it through this reference type, never
// through the by-value copy (whose address, once taken, would force a 1.3 KB scratch copy per lane).
#define ALG_AS4 __attribute__((address_space(4)))
typedef const ALG_AS4 Params& CPR;
__device__ __forceinline__ CPR kernel_params() {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const ALG_AS4 Params*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return *(const ALG_AS4 Params*)(unsigned long)64;      // host pass: never executed
#endif
}
// Device buffers are global memory, and the code has to SAY so: every buffer pointer of this file is either read out of the
// kernel-argument segment through `CPR` (a pointer-typed field of a constant-address-space struct) or laundered through an integer
// register (uniform_u64, Game::fresh), and in both cases the compiler's address-space inference loses track -- the accesses became FLAT
// instructions (two thirds of the memory instructions of k_newton_solve until round 5: 191 flat loads and 125 flat stores against 100 / 40
// global ones).  A flat access needs a 64-bit VGPR address, and -- worse for this code -- counts in lgkmcnt as well as vmcnt: every
// `s_waitcnt lgkmcnt(0)` of an LDS phase boundary also waited for the record prefetches and result stores in flight.  as_global() casts
// states it at the roots; everything derived from its result is selected as global_load / global_store.
template <class T> __device__ __forceinline__ T* as_global(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (generic -> address space 1 -> generic as two pointer casts is folded away before the inference runs; through an integer the first
    // cast is an inttoptr INTO address space 1, which the inference keeps and propagates to every access derived from the result)
    return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
#else
    return p;
#endif
}
// Opaque copy of the reference for one phase of the solver (see phase_int below): nothing that is derived from the
// parameters inside the phase can be hoisted in front of the solver's outer loops.
// (readfirstlane first: inside a function the inliner left out of line, arguments arrive in VGPRs; on a value that already
// lives in SGPRs the compiler folds it away)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    v = ((unsigned long long)hi << 32) | lo;
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ CPR phase_params(CPR pr) { return *(const ALG_AS4 Params*)uniform_u64((unsigned long long)&pr); }

// EXT_ = 1 instantiations carry the extended ingredient set of examples/intro_example.jl (state bounds, walls, circles;
// the bicycle model is always EXT); the EXT_ = 0 instantiations (the BASELINE configurations) pay nothing for it.
// NW_ > 1: NW_ wavefronts work on one game (workgroup = NW_ x 64 threads; small batches that leave most SIMDs empty): the
// streaming phases (assemble pass, trajectory updates, dual updates) are spread over all of them, the serial Newton-direction
// sweeps run on wavefront 0.  NW_ = 1 is the one-game-per-wavefront kernel of the large batches.
// LS_ = 0: a team kernel WITHOUT the line search's LDS staging (LsLds: 54 KB per team, two teams per CU).  The kernels that resume parked
// games (straggler hand-off) take it: a second launch wants as many teams resident as the registers allow -- four per CU -- and its line
// searches read their operands from L2 (same arithmetic, same results).
template <int MODEL_, int P_, int D_, int EXT_ = 0, int NW_ = 1, int LS_ = 1>
struct Cfg {
    static constexpr int MODEL = MODEL_, P = P_, D = D_;
    static constexpr bool LS_STAGE = LS_ != 0;
    static constexpr int NW = NW_, NT = NW_ * 64;          // wavefronts / threads per game
    static constexpr bool EXT = EXT_ != 0;
    static constexpr bool POS = (P_ > 1) || EXT;     // position blocks (pair / wall / circle terms) present in Q^_i
    // QuadrotorGame (quadrotor.jl:20-46): dense 12 x 12 / 12 x 4 Jacobian blocks per player, n up to 48
    static constexpr bool QUAD = MODEL_ == ALG_MODEL_QUADROTOR;
    static constexpr int n = (MODEL_ == ALG_MODEL_DOUBLE_INTEGRATOR) ? 2 * D_ * P_ : QUAD ? 12 * P_ : 4 * P_;
    static constexpr int m = (MODEL_ == ALG_MODEL_DOUBLE_INTEGRATOR) ? D_ * P_ : QUAD ? 4 * P_ : 2 * P_;
    // Newton direction variant: the single 16 x 16 tile path needs n <= 16 and n % 4 == 0; everything else (the quadrotor, the
    // double integrator in three dimensions with p = 1, 3, 4) takes the LDS-resident dense variant (newton_direction_dense)
    static constexpr bool DENSE = QUAD || n > 16 || (n % 4) != 0;
    static constexpr int mi = m / P_;
    static constexpr int ni = n / P_;
    static constexpr int b = n + m + P_ * n;
    static constexpr int NPAIR = P_ * (P_ - 1);
    // position dimensions that carry pair / wall terms: px[i] = (x, y) everywhere (double_integrator.jl:19, unicycle.jl, bicycle.jl);
    // the 3-D ingredients (spherical collision avoidance, Wall3D, Cylinder) act on pz[i][1:3] = (x, y, z) of DoubleIntegrator d = 3
    static constexpr int PD = (EXT_ != 0 && ((MODEL_ == ALG_MODEL_DOUBLE_INTEGRATOR && D_ == 3) || QUAD)) ? 3 : 2;
    static constexpr int NS = PD * (PD + 1) / 2;          // entries of a symmetric PD x PD block: (0,0) (0,1) (1,1) [(0,2) (1,2) (2,2)]
    __host__ __device__ static constexpr int sym(int a, int c) { return PD == 2 ? a + c : (a > c ? a * (a + 1) / 2 + c : c * (c + 1) / 2 + a); }
    // quadrotor: per player [A_i (12 x 12, row-major) | B_i (12 x 4) | RK2(x_k, u_k) entries of the player (12)]
    static constexpr int QA = 0, QB = 144, QX = 192, QS = 204;
    static constexpr int NC = (MODEL_ == ALG_MODEL_UNICYCLE) ? 4 * P_ : (MODEL_ == ALG_MODEL_BICYCLE) ? 10 * P_ : QUAD ? QS * P_ : 0;   // state-dependent RK2 Jacobian coefficients per knot
    static constexpr int NPAT = (MODEL_ == ALG_MODEL_BICYCLE) ? 4 : (MODEL_ == ALG_MODEL_UNICYCLE) ? 3 : 2;   // max non-zeros of a column of [B_k | A_k]
    static constexpr int WC = m + n + 1;         // augmented width of the control system
    // register budget of the solver kernels: waves per SIMD the compiler must leave room for (512 / WPE VGPRs per lane)
    // (the 3-D EXT instantiation carries 3 x 3 position blocks and does not fit 128 VGPRs without scratch)
    // (the LDS-resident dense direction of the larger configurations leaves room for less than one wavefront per SIMD)
    // (team kernels run small batches -- at most two wavefronts per SIMD are resident -- so they take the 256-register budget as well)
    static constexpr int WPE = (DENSE && n >= 24) ? 1 : (DENSE || n >= 16 || MODEL_ != ALG_MODEL_DOUBLE_INTEGRATOR || (EXT_ != 0 && D_ == 3) || NW_ > 1) ? 2 : 4;
    // reuse the accepted line-search trial as the next record! (one assemble pass less per Newton iteration)
    static constexpr bool TRIAL_REUSE = true;
    // forward / costate sweeps of the tile path: time steps whose record slice / gains / dx are in flight (register ring, loop unrolled by it)
#ifndef ALG_SWEEP_DEPTH
#define ALG_SWEEP_DEPTH 6
#endif
    // (measured in round 3, depth 1 / 2 / 4 / 8: C2 10.22 / 10.28 / 10.37 / 10.33 M/s, C3 2.38 / 2.43 / 2.43 / 2.44 M/s, C5 loop 154 / 158 / 157 / 156 K/s; again on the
    // round-6 kernel, whose sweep steps are shorter -- depth 2 / 4 / 6: 3.095 / 3.056 / 3.019 ms per C2 launch, bit-identical, eight spills the C2 kernel:
    // profiles/r06_ab_sweep_depth_c2.txt)
#ifndef ALG_SWEEP_DEPTH_W2
#define ALG_SWEEP_DEPTH_W2 4       // 256-register kernels (2 until late in round 6; 4: bit-identical, C3 +1.6 %, C5 loop +1.5 %, C5 / C3 at 4096 games +2.8 / +3.9 %: profiles/r06_ab_sdw4_*.txt)
#endif
    static constexpr int SWEEP_DEPTH = WPE == 4 ? ALG_SWEEP_DEPTH : (ALG_SWEEP_DEPTH < ALG_SWEEP_DEPTH_W2 ? ALG_SWEEP_DEPTH : ALG_SWEEP_DEPTH_W2);
    // rows per lane and pass of the assemble row loops (memory-level parallelism against the L2 / store-ack latency)
#ifndef ALG_ASM_UNROLL_W2
#define ALG_ASM_UNROLL_W2 2       // ... in the 256-register kernels (3 / 4 measured on the team kernels: bit-identical, C3 -1 %, C5 loop -2 %, C2 at 512 games -3 %; profiles/r06_ab_ur_*.txt)
#endif
    static constexpr int ASM_UNROLL = WPE <= 2 ? ALG_ASM_UNROLL_W2 : 2;
};

// newton_solve / rollout are __forceinline__: they have two callers per instantiation (k_newton_solve, k_mpc_loop) and the
// inliner would outline the largest instantiations -- a kernel whose callee is outlined gets its by-value Params copied to
// scratch (1.3 KB per lane) and loses a third of its speed.  (Forcing the other solver-level functions changes the inlining
// order and costs registers: they are left to the inliner, tests/probes/isa_stats.py + `grep s_swappc` guard against outlining.)

// ---- index maps (newton_core.jl:40-89), 0-based --------------------------------------------------
template <class C> __device__ __forceinline__ int hx(int k) { return k * C::b; }
template <class C> __device__ __forceinline__ int hu(int k, int i) { return k * C::b + C::n + i * C::mi; }
template <class C> __device__ __forceinline__ int hl(int k, int i) { return k * C::b + C::n + C::m + i * C::n; }
template <class C> __device__ __forceinline__ int vx(int N, int i, int k) { return i * (N - 1) * (C::n + C::mi) + k * (C::n + C::mi); }
template <class C> __device__ __forceinline__ int vu(int N, int i, int k) { return vx<C>(N, i, k) + C::n; }
template <class C> __device__ __forceinline__ int vd(int N, int k) { return C::P * (N - 1) * (C::n + C::mi) + k * C::n; }
// joint control index c -> offset inside the u block of the horizontal order (player-grouped)
template <class C> __device__ __forceinline__ int uoff(int c) { return (c % C::P) * C::mi + c / C::P; }
template <class C> __device__ __forceinline__ int pairq(int i, int j) { return i * (C::P - 1) + (j < i ? j : j - 1); }
template <class C> __device__ __forceinline__ int con_col(int N, int q, int k /*knot 1..N-1*/) { return q * (N - 1) + (k - 1); }
template <class C> __device__ __forceinline__ int con_ctl(CPR pr, int k, int row) { return pr.col_len + k * 2 * C::m + row; }

// state of knot k (0-based) inside a traj buffer
template <class C> __device__ __forceinline__ const double* zstate(const double* z, int k) { return k == 0 ? z : z + C::n + hx<C>(k - 1); }

// ---- thread index / synchronisation of ONE game -------------------------------------------------------------------------------
// Every kernel runs one game per workgroup (one wavefront, or a team of Cfg::NW), so the game's thread index is the workgroup's
// and its barrier is the workgroup barrier.
__device__ __forceinline__ int game_tid() { return (int)threadIdx.x; }
// (measured in round 6: 63 of the C2 kernel's 194 full 32-bit multiplies become 24-bit ones, the rate does not move -- profiles/r06_ab_lane_range_c2.txt --
// and the 4-player DoubleIntegrator d = 1 kernel starts to spill: off)
#ifndef ALG_R6_LANE_RANGE
#define ALG_R6_LANE_RANGE 0
#endif
__device__ __forceinline__ void game_sync() { __syncthreads(); }
// ---- wave reductions ---------------------------------------------------------------------------
// Opaque copy of the lane id: keeps per-lane role / address computations of a phase from being hoisted out of the
// solver's outer loops (where every phase's invariants would be live at once).
// (ALG_R6_LANE_RANGE: the copy keeps the id's value range -- a workgroup has at most 256 threads -- so that products of lane-derived indices can take
// the 24-bit multiply)
__device__ __forceinline__ int phase_lane() {
    int l = game_tid(); asm volatile("" : "+v"(l));
#if ALG_R6_LANE_RANGE
    __builtin_assume((unsigned)l < 256u);
#endif
    return l;
}
// Opaque copies of wave-uniform loop invariants (problem sizes, dt, base pointers), taken at the start of a phase: whatever is
// derived from them (row counts, address vectors, dt^2 / 2, (double)S ...) is recomputed inside the phase with a handful of
// scalar instructions instead of being hoisted in front of the solver's outer loops and kept alive -- or spilled -- there.
__device__ __forceinline__ int phase_int(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ double phase_f64(double v) { return __longlong_as_double((long long)uniform_u64((unsigned long long)__double_as_longlong(v))); }

// Butterfly over the 64 lanes without the LDS crossbar (the __shfl_xor form costs two ds_bpermute per level and double, ~100 cycles of
// latency each).  Levels 32 and 16: v_permlane32_swap / v_permlane16_swap return (own | partner's) and (partner's | own) in their two
// results, so result0 (op) result1 is own (op) partner in every lane, without a lane select.  Levels 8, 4, 2, 1: row rotations
// (v_mov_b32_dpp row_ror): after level 8 the row holds every value twice 8 lanes apart, so lane i + 4 carries what lane i ^ 4 does, and
// so on down -- the same operands meet at every level as in the xor butterfly: bit-identical results for commutative ops.
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) { return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v))); }
template <class Op>
__device__ __forceinline__ double wave_butterfly(double v, Op op) {
    {
        const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = op(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    {
        const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = op(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    v = op(v, dpp_f64<0x128>(v)); v = op(v, dpp_f64<0x124>(v)); v = op(v, dpp_f64<0x122>(v)); v = op(v, dpp_f64<0x121>(v));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_butterfly(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wave_max(double v) { return wave_butterfly(v, [](double a, double b) { return fmax(a, b); }); }
// fmax drops NaNs; carry a separate finite flag where NaN detection matters
__device__ __forceinline__ int wave_or(int v) {
    { const auto a = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = a[0] | a[1]; }
    { const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = a[0] | a[1]; }
    v |= dpp_i32<0x128>(v); v |= dpp_i32<0x124>(v); v |= dpp_i32<0x122>(v); v |= dpp_i32<0x121>(v);
    return v;
}

// wave-uniform scalars live in SGPRs
__device__ __forceinline__ double uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// ---- wavefront team of one game (Cfg::NW) ----------------------------------------------------------------------------------
// Synchronisation inside the Newton-direction sweeps, which only wavefront 0 of a team executes: a wave-local fence (the same
// fences game_sync() carries, without the workgroup barrier).  NW == 1: the workgroup is the wavefront, plain game_sync().
template <class C> __device__ __forceinline__ void dir_sync() {
    if constexpr (C::NW == 1) game_sync();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
}
// Synchronisation between the LDS phases of the tile-path sweeps when one wavefront runs them: orders LDS only.  game_sync() /
// __syncthreads() also drain vmcnt, i.e. every phase boundary of a time step (eight in the backward sweep) waited for the record
// prefetch and the gain stores that had just been issued -- the sweeps ran at global-memory latency.  What the sweeps exchange
// through global memory (gains, dx) crosses a full game_sync() between the sweeps.
// (team kernels: the serial sweeps run on wavefront 0 alone, so the same holds; until round 4 they used the full fence of dir_sync())
template <class C> __device__ __forceinline__ void sweep_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Workgroup barrier that orders LDS only (no vmcnt drain: the record prefetch and the gain stores of a sweep step stay in flight)
__device__ __forceinline__ void team_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <class C> __device__ __forceinline__ void rotate_priority(int it);
template <class C> __device__ __forceinline__ int team_wave() { return C::NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(game_tid() >> 6)); }

// ---- global memory access as (uniform 64-bit base) + (unsigned 32-bit byte offset) ------------------------------------------------
// `p[i]` with an int index sign-extends i and forms a 64-bit address in VGPRs for every access (v_ashrrev_i32 + v_lshl_add_u64: 1.4 K
// 64-bit integer instructions per game-Newton-iteration at C2, SQ_INSTS_VALU_INT64 in profiles/r05_pmc_c2_final_summary.txt).  With
// the base uniform (an SGPR pair: the game's chunk plus a step offset) and the lane's part an unsigned byte offset, the access is
// `global_load_dwordx2 v, v_off, s[base]`: no address arithmetic beyond the (usually loop-invariant) 32-bit offset.  Offsets stay far
// below 2^32 bytes: they address one game's chunk.
// (the empty asm keeps the 32-bit offset a 32-bit value INSIDE the basic block of the access: hoisted out of a loop it becomes a
// zero-extended 64-bit pair, instruction selection -- per block -- no longer sees the zero extension and falls back to a 64-bit add)
__device__ __forceinline__ unsigned goff(int i) { unsigned o = (unsigned)i << 3; asm("" : "+v"(o)); return o; }
__device__ __forceinline__ double gld(const double* p, int i) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(p) + goff(i)); }
__device__ __forceinline__ void gst(double* p, int i, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(p) + goff(i)) = v; }
// ... and for any element type (16-byte pairs of the streaming axpy)
template <class T> __device__ __forceinline__ T gld_t(const T* p, int i) { unsigned o = (unsigned)i * (unsigned)sizeof(T); asm("" : "+v"(o)); return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + o); }
template <class T> __device__ __forceinline__ void gst_t(T* p, int i, T v) { unsigned o = (unsigned)i * (unsigned)sizeof(T); asm("" : "+v"(o)); *reinterpret_cast<T*>(reinterpret_cast<char*>(p) + o) = v; }

// ---- counter RNG shared bit-for-bit with the oracle ------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double counter_uniform(uint64_t seed, uint64_t game, uint64_t counter) {
    uint64_t h = splitmix64(seed ^ splitmix64(game * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull));
    h = splitmix64(h + counter * 0x9E3779B97F4A7C15ull);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ================================================================================================
// Models.  Both in-scope models are player-decoupled: player i owns state entries pz(i,j)=i+j*P and
// control entries pu(i,j)=i+j*P (double_integrator.jl:18-20, unicycle.jl:18-20).  The RK2 (explicit
// midpoint) Jacobian [A B] (local_quantities.jl:20-27) is I + a handful of entries per player; those
// entries are the "Jacobian coefficients" coef[4*P]:
//   unicycle (state [x(P) y(P) th(P) v(P)], control [om(P) a(P)]), with thm = th + dt/2 om, vm = v + dt/2 a:
//     coef[0*P+i] = d x+/d th = -dt vm sin thm      coef[1*P+i] = d x+/d v = dt cos thm
//     coef[2*P+i] = d y+/d th =  dt vm cos thm      coef[3*P+i] = d y+/d v = dt sin thm
//     d x+/d om = dt/2 coef0, d x+/d a = dt/2 coef1, d y+/d om = dt/2 coef2, d y+/d a = dt/2 coef3,
//     d th+/d om = dt, d v+/d a = dt
//   double integrator: A = [[I, dt I],[0, I]], B = [[dt^2/2 I],[dt I]] (no state dependence; coef unused)
//   bicycle (bicycle.jl:28-41; state [x(P) y(P) v(P) psi(P)], control [a(P) delta(P)]), beta = atan(lr tan delta, lr + lf),
//     sg = sin(beta)/lr, vm = v + dt/2 a, psm = psi + dt/2 v sg, th = beta + psm, beta' = d beta/d delta, sg' = cos(beta) beta'/lr:
//     coef[0] = d x+/d psi = -dt vm sin th          coef[1] = d x+/d v = dt cos th - dt vm sin th (dt/2 sg)
//     coef[2] = d y+/d psi =  dt vm cos th          coef[3] = d y+/d v = dt sin th + dt vm cos th (dt/2 sg)
//     coef[4] = d psi+/d v = dt sg                  coef[5] = dt cos th      coef[6] = dt sin th
//     coef[7] = d x+/d delta = -dt vm sin th (beta' + dt/2 v sg')   coef[8] = d y+/d delta = dt vm cos th (beta' + dt/2 v sg')
//     coef[9] = d psi+/d delta = dt vm sg'
//     d x+/d a = dt/2 coef5, d y+/d a = dt/2 coef6, d v+/d a = dt, d psi+/d a = dt/2 coef4        (each coef[t] at [t*P + i])
// ================================================================================================
// ---- QuadrotorGame (quadrotor.jl:49-121).  Player-local state [r (0..2) | MRP g (3..5) | v (6..8) | omega (9..11)], rotor commands
// w1..w4.  Rotations.jl 1.0 MRP: rotation matrix I + (4 (1 - s) [g x] + 8 [g x]^2) / (1 + s)^2, s = |g|^2 (only its third column is
// needed: the rotor force acts along body z); kinematics(g, w) = 1/4 ((1 - s) w + 2 g x w + 2 (g . w) g).  Written once for a
// scalar type T: double for the value, Jet (value + one directional derivative) for a column of the RK2 Jacobian -- the
// forward-mode AD of discrete_jacobian! (local_quantities.jl:20-27) with one seed direction per lane.
struct Jet { double v, d; };
__device__ __forceinline__ Jet operator+(const Jet& a, const Jet& b) { return Jet{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Jet operator-(const Jet& a, const Jet& b) { return Jet{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Jet operator*(const Jet& a, const Jet& b) { return Jet{a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Jet operator/(const Jet& a, const Jet& b) { const double q = a.v / b.v; return Jet{q, (a.d - q * b.d) * (1.0 / b.v)}; }
__device__ __forceinline__ Jet operator*(const Jet& a, double s) { return Jet{a.v * s, a.d * s}; }
__device__ __forceinline__ Jet operator+(const Jet& a, double s) { return Jet{a.v + s, a.d}; }
__device__ __forceinline__ Jet max0(const Jet& a) { return a.v > 0.0 ? a : Jet{0.0, 0.0}; }      // ForwardDiff: derivative of the selected branch
__device__ __forceinline__ double max0(double a) { return a > 0.0 ? a : 0.0; }
template <class T>
__device__ __forceinline__ void quad_f(const T (&x)[12], const T (&u)[4], double mass, T (&xd)[12]) {
    const double J0 = 0.0023, J1 = 0.0023, J2 = 0.004, grav = -9.81, L = 0.1750, kf = 1.245, km = 1.0;
    const T g0 = x[3], g1 = x[4], g2 = x[5], w0 = x[9], w1 = x[10], w2 = x[11];
    const T F1 = max0(u[0] * kf), F2 = max0(u[1] * kf), F3 = max0(u[2] * kf), F4 = max0(u[3] * kf);
    const T Ft = F1 + F2 + F3 + F4;
    const T s = g0 * g0 + g1 * g1 + g2 * g2;
    const T den = (s + 1.0) * (s + 1.0);
    const T c4 = ((s * (-1.0)) + 1.0) * 4.0;
    const T r02 = (c4 * g1 + (g0 * g2) * 8.0) / den;
    const T r12 = ((c4 * g0) * (-1.0) + (g1 * g2) * 8.0) / den;
    const T r22 = (((g0 * g0 + g1 * g1) * (-8.0)) / den) + 1.0;
    const T t0 = (F2 - F4) * L, t1 = (F3 - F1) * L, t2 = (u[0] - u[1] + u[2] - u[3]) * km;
    xd[0] = x[6]; xd[1] = x[7]; xd[2] = x[8];
    const T gw = g0 * w0 + g1 * w1 + g2 * w2, oms = (s * (-1.0)) + 1.0;
    xd[3] = (oms * w0 + (g1 * w2 - g2 * w1) * 2.0 + (gw * g0) * 2.0) * 0.25;
    xd[4] = (oms * w1 + (g2 * w0 - g0 * w2) * 2.0 + (gw * g1) * 2.0) * 0.25;
    xd[5] = (oms * w2 + (g0 * w1 - g1 * w0) * 2.0 + (gw * g2) * 2.0) * 0.25;
    xd[6] = (r02 * Ft) * (1.0 / mass);
    xd[7] = (r12 * Ft) * (1.0 / mass);
    xd[8] = ((r22 * Ft) * (1.0 / mass)) + grav;
    xd[9] = (t0 - (w1 * w2) * (J2 - J1)) * (1.0 / J0);
    xd[10] = (t1 - (w2 * w0) * (J0 - J2)) * (1.0 / J1);
    xd[11] = (t2 - (w0 * w1) * (J1 - J0)) * (1.0 / J2);
}
// RobotDynamics 0.3.1 RK2 (explicit midpoint) of one quadrotor
template <class T>
__device__ __forceinline__ void quad_rk2(const T (&x)[12], const T (&u)[4], double mass, double dt, T (&xn)[12]) {
    T k[12], xm[12];
    quad_f(x, u, mass, k);
#pragma unroll
    for (int j = 0; j < 12; j++) xm[j] = x[j] + k[j] * (dt * 0.5);
    quad_f(xm, u, mass, k);
#pragma unroll
    for (int j = 0; j < 12; j++) xn[j] = x[j] + k[j] * dt;
}

// RobotDynamics 0.3.1 RK3 of one quadrotor (rollout!, solver_methods.jl:17)
__device__ __forceinline__ void quad_rk3(const double (&x)[12], const double (&u)[4], double mass, double dt, double (&xn)[12]) {
    double k1[12], k2[12], k3[12], t[12];
    quad_f(x, u, mass, k1);
#pragma unroll
    for (int j = 0; j < 12; j++) { k1[j] *= dt; t[j] = x[j] + k1[j] / 2; }
    quad_f(t, u, mass, k2);
#pragma unroll
    for (int j = 0; j < 12; j++) { k2[j] *= dt; t[j] = x[j] - k1[j] + 2 * k2[j]; }
    quad_f(t, u, mass, k3);
#pragma unroll
    for (int j = 0; j < 12; j++) { k3[j] *= dt; xn[j] = x[j] + (k1[j] + 4 * k2[j] + k3[j]) / 6; }
}

struct BikeGeom { double beta, sg, dbeta, dsg; };
__device__ __forceinline__ BikeGeom bike_geom(double delta, double lf, double lr) {
    const double L = lr + lf, td = tan(delta), y = lr * td;
    BikeGeom g;
    g.beta = atan2(y, L);
    double sb, cb; sincos(g.beta, &sb, &cb);
    g.sg = sb / lr;
    g.dbeta = (lr * L) * (1.0 + td * td) / (L * L + y * y);
    g.dsg = cb * g.dbeta / lr;
    return g;
}
// Jacobian coefficients of player i at a knot with own state (v, psi) and controls (a, delta)
template <class C>
__device__ __forceinline__ void bike_coefs(CPR pr, double v, double psi, double a, double delta, double dt, double (&cf)[10]) {
    const BikeGeom g = bike_geom(delta, pr.lf, pr.lr);
    const double vm = v + (a * dt) * 0.5, psm = psi + (v * g.sg * dt) * 0.5;
    double sn, cs; sincos(g.beta + psm, &sn, &cs);
    const double dth = g.dbeta + 0.5 * dt * v * g.dsg;
    cf[0] = -dt * vm * sn; cf[1] = dt * cs - dt * vm * sn * (0.5 * dt * g.sg);
    cf[2] = dt * vm * cs;  cf[3] = dt * sn + dt * vm * cs * (0.5 * dt * g.sg);
    cf[4] = dt * g.sg; cf[5] = dt * cs; cf[6] = dt * sn;
    cf[7] = -dt * vm * sn * dth; cf[8] = dt * vm * cs * dth; cf[9] = dt * vm * g.dsg;
}
template <class C>
__device__ __forceinline__ void model_player(CPR pr, int i, const double* x, const double* u, double dt,
                                             double* xn /*ni: entries pz(i,j)*/, double* coef /*4: entries j*P+i (unicycle only)*/) {
    if constexpr (C::QUAD) {
        double xi[12], ui[4], xo[12];
#pragma unroll
        for (int j = 0; j < 12; j++) xi[j] = x[i + j * C::P];
#pragma unroll
        for (int j = 0; j < 4; j++) ui[j] = u[i + j * C::P];
        quad_rk2(xi, ui, pr.qmass, dt, xo);
#pragma unroll
        for (int j = 0; j < 12; j++) xn[j] = xo[j];
        coef[0] = coef[1] = coef[2] = coef[3] = 0.0;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
#pragma unroll
        for (int j = 0; j < C::D; j++) {
            const int ip = i + j * C::P, iv = C::m + i + j * C::P;
            // k1 = f(x,u) dt ; xm = x + k1/2 ; k2 = f(xm,u) dt ; x + k2   (RobotDynamics 0.3.1 RK2)
            const double vm = x[iv] + (u[ip] * dt) * 0.5;
            xn[j] = x[ip] + vm * dt;
            xn[C::D + j] = x[iv] + u[ip] * dt;
        }
        coef[0] = coef[1] = coef[2] = coef[3] = 0.0;
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P;
        const double v = x[2 * P + i], psi = x[3 * P + i], a = u[i];
        const BikeGeom g = bike_geom(u[P + i], pr.lf, pr.lr);
        const double vm = v + (a * dt) * 0.5, psm = psi + (v * g.sg * dt) * 0.5;
        double s, c; sincos(g.beta + psm, &s, &c);
        xn[0] = x[i] + (vm * c) * dt;
        xn[1] = x[P + i] + (vm * s) * dt;
        xn[2] = v + a * dt;
        xn[3] = psi + (vm * g.sg) * dt;
        coef[0] = coef[1] = coef[2] = coef[3] = 0.0;
    } else {
        const int P = C::P;
        const double th = x[2 * P + i], v = x[3 * P + i], om = u[i], a = u[P + i];
        const double thm = th + (om * dt) * 0.5, vm = v + (a * dt) * 0.5;
        double s, c;
        sincos(thm, &s, &c);
        xn[0] = x[i] + (c * vm) * dt;
        xn[1] = x[P + i] + (s * vm) * dt;
        xn[2] = th + om * dt;
        xn[3] = v + a * dt;
        coef[0] = -dt * vm * s; coef[1] = dt * c; coef[2] = dt * vm * c; coef[3] = dt * s;
    }
}
// RK3 step of player i (rollout!, solver_methods.jl:17; RobotDynamics 0.3.1 RK3)
// (the player's own entries as compact arrays xi[ni], ui[mi]: the roll-out keeps its state in this form -- a joint array indexed by the lane's
// player number is a register array with a run-time index, i.e. a select chain over n compare masks per access)
template <class C>
__device__ __forceinline__ void model_player_rk3_own(CPR pr, const double (&xi)[C::ni], const double (&ui)[C::mi], double dt, double* xn) {
    static_assert(!C::QUAD, "the quadrotor integrates through quad_rk3 (rollout)");
    double k1[C::ni], k2[C::ni], k3[C::ni], t[C::ni];
    auto f = [&](const double* s, double* o) {
        if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
#pragma unroll
            for (int j = 0; j < C::D; j++) { o[j] = s[C::D + j]; o[C::D + j] = ui[j]; }
        } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
            const BikeGeom g = bike_geom(ui[1], pr.lf, pr.lr);
            double sn, cs; sincos(g.beta + s[3], &sn, &cs);
            o[0] = s[2] * cs; o[1] = s[2] * sn; o[2] = ui[0]; o[3] = s[2] * g.sg;
        } else {
            double sn, cs; sincos(s[2], &sn, &cs);
            o[0] = cs * s[3]; o[1] = sn * s[3]; o[2] = ui[0]; o[3] = ui[1];
        }
    };
    f(xi, k1);
#pragma unroll
    for (int j = 0; j < C::ni; j++) { k1[j] *= dt; t[j] = xi[j] + k1[j] / 2; }
    f(t, k2);
#pragma unroll
    for (int j = 0; j < C::ni; j++) { k2[j] *= dt; t[j] = xi[j] - k1[j] + 2 * k2[j]; }
    f(t, k3);
#pragma unroll
    for (int j = 0; j < C::ni; j++) { k3[j] *= dt; xn[j] = xi[j] + (k1[j] + 4 * k2[j] + k3[j]) / 6; }
}
template <class C>
__device__ __forceinline__ void model_player_rk3(CPR pr, int i, const double* x, const double* u, double dt, double* xn) {
    double xi[C::ni], ui[C::mi];
#pragma unroll
    for (int j = 0; j < C::ni; j++) xi[j] = x[i + j * C::P];
#pragma unroll
    for (int j = 0; j < C::mi; j++) ui[j] = u[i + j * C::P];
    model_player_rk3_own<C>(pr, xi, ui, dt, xn);
}

// (A^T v)[r] for a vector accessor v(r'): A = I + E.  Branch-free: every lane issues the same loads (clamped indices) and
// masks the coefficients, so that divergent rows do not serialise their memory latencies.
template <class C, class V>
__device__ __forceinline__ double AT_vec(const double* coef, double dt, V v, int r) {
    if constexpr (C::QUAD) {
        const int P = C::P, i = r % P, a = r / P; const double* Ai = coef + i * C::QS + C::QA;
        double acc = 0.0;
#pragma unroll
        for (int a2 = 0; a2 < 12; a2++) acc += Ai[a2 * 12 + a] * v(a2 * P + i);
        return acc;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        const bool hi = r >= C::m;
        return v(r) + (hi ? dt : 0.0) * v(hi ? r - C::m : r);
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, blk = r / P, i = r % P;
        const bool b2 = blk == 2, on = blk >= 2;
        const double ca = coef[(b2 ? 1 : 0) * P + i], cb = coef[(b2 ? 3 : 2) * P + i], cc = coef[4 * P + i];
        return v(r) + (on ? ca : 0.0) * v(i) + (on ? cb : 0.0) * v(P + i) + (b2 ? cc : 0.0) * v(3 * P + i);
    } else {
        const int P = C::P, blk = r / P, i = r % P;
        const bool b3 = blk == 3, on = blk >= 2;
        const double ca = coef[(b3 ? 1 : 0) * P + i], cb = coef[(b3 ? 3 : 2) * P + i];
        return v(r) + (on ? ca : 0.0) * v(i) + (on ? cb : 0.0) * v(P + i);
    }
}
// (A v)[r]
template <class C, class V>
__device__ __forceinline__ double A_vec(const double* coef, double dt, V v, int r) {
    if constexpr (C::QUAD) {
        const int P = C::P, i = r % P, a = r / P; const double* Ai = coef + i * C::QS + C::QA;
        double acc = 0.0;
#pragma unroll
        for (int a2 = 0; a2 < 12; a2++) acc += Ai[a * 12 + a2] * v(a2 * P + i);
        return acc;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        return r < C::m ? v(r) + dt * v(r + C::m) : v(r);
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, blk = r / P, i = r % P;
        if (blk == 0) return v(r) + coef[0 * P + i] * v(3 * P + i) + coef[1 * P + i] * v(2 * P + i);
        if (blk == 1) return v(r) + coef[2 * P + i] * v(3 * P + i) + coef[3 * P + i] * v(2 * P + i);
        if (blk == 3) return v(r) + coef[4 * P + i] * v(2 * P + i);
        return v(r);
    } else {
        const int P = C::P, blk = r / P, i = r % P;
        if (blk == 0) return v(r) + coef[0 * P + i] * v(2 * P + i) + coef[1 * P + i] * v(3 * P + i);
        if (blk == 1) return v(r) + coef[2 * P + i] * v(2 * P + i) + coef[3 * P + i] * v(3 * P + i);
        return v(r);
    }
}
// (X A)[c] for a row accessor X(r'): column op
template <class C, class V>
__device__ __forceinline__ double XA_vec(const double* coef, double dt, V X, int c) { return AT_vec<C>(coef, dt, X, c); }
// A[r][c]
template <class C>
__device__ __forceinline__ double A_entry(const double* coef, double dt, int r, int c) {
    double e = (r == c) ? 1.0 : 0.0;
    if constexpr (C::QUAD) {
        e = (r % C::P == c % C::P) ? coef[(r % C::P) * C::QS + C::QA + (r / C::P) * 12 + c / C::P] : 0.0;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        if (r < C::m && c == r + C::m) e = dt;
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, br = r / P, i = r % P;
        if (br == 0) { if (c == 3 * P + i) e = coef[0 * P + i]; else if (c == 2 * P + i) e = coef[1 * P + i]; }
        else if (br == 1) { if (c == 3 * P + i) e = coef[2 * P + i]; else if (c == 2 * P + i) e = coef[3 * P + i]; }
        else if (br == 3) { if (c == 2 * P + i) e = coef[4 * P + i]; }
    } else {
        const int P = C::P, br = r / P, i = r % P;
        if (br == 0) { if (c == 2 * P + i) e = coef[0 * P + i]; else if (c == 3 * P + i) e = coef[1 * P + i]; }
        else if (br == 1) { if (c == 2 * P + i) e = coef[2 * P + i]; else if (c == 3 * P + i) e = coef[3 * P + i]; }
    }
    return e;
}
// B[r][c]  (c: joint control index)
template <class C>
__device__ __forceinline__ double B_entry(const double* coef, double dt, int r, int c) {
    if constexpr (C::QUAD) {
        return (r % C::P == c % C::P) ? coef[(r % C::P) * C::QS + C::QB + (r / C::P) * 4 + c / C::P] : 0.0;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        if (r == c) return 0.5 * dt * dt;
        if (r == c + C::m) return dt;
        return 0.0;
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, i = c % P, kind = c / P;   // kind 0: a_i, 1: delta_i
        if (kind == 0) {
            if (r == i) return 0.5 * dt * coef[5 * P + i];
            if (r == P + i) return 0.5 * dt * coef[6 * P + i];
            if (r == 2 * P + i) return dt;
            if (r == 3 * P + i) return 0.5 * dt * coef[4 * P + i];
        } else {
            if (r == i) return coef[7 * P + i];
            if (r == P + i) return coef[8 * P + i];
            if (r == 3 * P + i) return coef[9 * P + i];
        }
        return 0.0;
    } else {
        const int P = C::P, i = c % P, kind = c / P;   // kind 0: omega_i, 1: a_i
        if (kind == 0) {
            if (r == i) return 0.5 * dt * coef[0 * P + i];
            if (r == P + i) return 0.5 * dt * coef[2 * P + i];
            if (r == 2 * P + i) return dt;
        } else {
            if (r == i) return 0.5 * dt * coef[1 * P + i];
            if (r == P + i) return 0.5 * dt * coef[3 * P + i];
            if (r == 3 * P + i) return dt;
        }
        return 0.0;
    }
}
// (B^T v)[c] : column c of B has <= 4 non-zeros (branch-free, see AT_vec)
template <class C, class V>
__device__ __forceinline__ double BT_vec(const double* coef, double dt, V v, int c) {
    if constexpr (C::QUAD) {
        const int P = C::P, i = c % P, j = c / P; const double* Bi = coef + i * C::QS + C::QB;
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 12; a++) acc += Bi[a * 4 + j] * v(a * P + i);
        return acc;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        // (explicit fmas: `a b + c d` can be contracted either way, and the compiler did choose differently from one pass to another --
        // the fused trial pass and the line search's group pass came out an ulp apart in these rows)
        return __builtin_fma(0.5 * dt * dt, v(c), dt * v(c + C::m));
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, i = c % P; const bool k0 = (c / P) == 0;
        const double ca = coef[(k0 ? 5 : 7) * P + i], cb = coef[(k0 ? 6 : 8) * P + i], cc = coef[(k0 ? 4 : 9) * P + i];
        return (k0 ? 0.5 * dt : 1.0) * (ca * v(i) + cb * v(P + i) + cc * v(3 * P + i)) + (k0 ? dt : 0.0) * v(2 * P + i);
    } else {
        const int P = C::P, i = c % P, kind = c / P;
        const double inner = __builtin_fma(coef[kind * P + i], v(i), coef[(2 + kind) * P + i] * v(P + i));
        return __builtin_fma(0.5 * dt, inner, dt * v((2 + kind) * P + i));
    }
}
// (|B|^T v)[c] for v >= 0: BT_vec with the magnitudes of the coefficients -- the row scale |B[:,c]|' |dlambda| of the refinement gate's
// backward error (dir_urow_residual), in BT_vec's association
template <class C, class V>
__device__ __forceinline__ double BT_vec_abs(const double* coef, double dt, V v, int c) {
    if constexpr (C::QUAD) {
        const int P = C::P, i = c % P, j = c / P; const double* Bi = coef + i * C::QS + C::QB;
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 12; a++) acc += fabs(Bi[a * 4 + j]) * v(a * P + i);
        return acc;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        return __builtin_fma(0.5 * dt * dt, v(c), dt * v(c + C::m));                 // B is non-negative
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, i = c % P; const bool k0 = (c / P) == 0;
        const double ca = fabs(coef[(k0 ? 5 : 7) * P + i]), cb = fabs(coef[(k0 ? 6 : 8) * P + i]), cc = fabs(coef[(k0 ? 4 : 9) * P + i]);
        return (k0 ? 0.5 * dt : 1.0) * (ca * v(i) + cb * v(P + i) + cc * v(3 * P + i)) + (k0 ? dt : 0.0) * v(2 * P + i);
    } else {
        const int P = C::P, i = c % P, kind = c / P;
        const double inner = __builtin_fma(fabs(coef[kind * P + i]), v(i), fabs(coef[(2 + kind) * P + i]) * v(P + i));
        return __builtin_fma(0.5 * dt, inner, dt * v((2 + kind) * P + i));
    }
}
// (B w)[r] for a control-vector accessor w(c): row r of B has <= 2 non-zeros
template <class C, class V>
__device__ __forceinline__ double B_vec(const double* coef, double dt, V w, int r) {
    if constexpr (C::QUAD) {
        const int P = C::P, i = r % P, a = r / P; const double* Bi = coef + i * C::QS + C::QB;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++) acc += Bi[a * 4 + j] * w(j * P + i);
        return acc;
    } else if constexpr (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR) {
        return r < C::m ? 0.5 * dt * dt * w(r) : dt * w(r - C::m);
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, br = r / P, i = r % P;
        if (br == 0) return 0.5 * dt * coef[5 * P + i] * w(i) + coef[7 * P + i] * w(P + i);
        if (br == 1) return 0.5 * dt * coef[6 * P + i] * w(i) + coef[8 * P + i] * w(P + i);
        if (br == 2) return dt * w(i);
        return 0.5 * dt * coef[4 * P + i] * w(i) + coef[9 * P + i] * w(P + i);
    } else {
        const int P = C::P, br = r / P, i = r % P;
        if (br == 0) return 0.5 * dt * (coef[0 * P + i] * w(i) + coef[1 * P + i] * w(P + i));
        if (br == 1) return 0.5 * dt * (coef[2 * P + i] * w(i) + coef[3 * P + i] * w(P + i));
        if (br == 2) return dt * w(i);
        return dt * w(P + i);
    }
}

// ================================================================================================
// Per-game data view
// ================================================================================================
// One base pointer per game (the game's chunk of the main arena) plus 32-bit offsets: every other address is derived from
// the kernel arguments where it is used, so that a solver kernel does not carry eighteen 64-bit base pointers in SGPRs
// through all of its phases.
struct Game {
    double* base;           // this game's chunk of the main arena
    int g;                  // game index inside the handle's batch
    int zo[3];              // offsets of pdtraj / trial / delta inside the chunk (the line search exchanges the first two)
    // Opaque copy for one phase of the solver: addresses derived from it cannot be hoisted out of the solver's outer loops
    // (where the base pointers of every phase would be live -- and spilled -- at once); they are recomputed per phase with a
    // few scalar instructions instead.
    __device__ __forceinline__ Game fresh() const {
        Game H = *this;
        H.base = as_global(reinterpret_cast<double*>(uniform_u64(reinterpret_cast<unsigned long long>(base))));
        H.g = __builtin_amdgcn_readfirstlane(g); asm volatile("" : "+s"(H.g));
        return H;
    }
    __device__ __forceinline__ double* z(int t) const { return base + zo[t]; }
    __device__ __forceinline__ const double* x0(CPR pr) const { return base + pr.o_x0; }
    __device__ __forceinline__ double* x0w(CPR pr) const { return base + pr.o_x0; }
    __device__ __forceinline__ double* res(CPR pr) const { return base + pr.o_res; }
    __device__ __forceinline__ double* rec(CPR pr) const { return base + pr.o_rec; }
    __device__ __forceinline__ double* kgain(CPR pr) const { return base + pr.o_kgain; }
    __device__ __forceinline__ double* tc(CPR pr) const { return base + pr.o_tc; }
    __device__ __forceinline__ alg_game_stats* st(CPR pr) const { return reinterpret_cast<alg_game_stats*>(base + pr.o_st); }
    __device__ __forceinline__ long long* mpc(CPR pr) const { return reinterpret_cast<long long*>(base + pr.o_mpc); }
    __device__ __forceinline__ double* lam(CPR pr) const { return as_global(pr.con) + (size_t)g * pr.con_stride; }
    __device__ __forceinline__ double* mu(CPR pr) const { return lam(pr) + pr.con_pad; }
    __device__ __forceinline__ double* vals(CPR pr) const { return lam(pr) + 2 * pr.con_pad; }
    __device__ __forceinline__ const double* Qd(CPR pr) const { return as_global(pr.lqr) + (size_t)g * pr.lqr_stride; }
    __device__ __forceinline__ const double* xf(CPR pr) const { return Qd(pr) + pr.p * pr.ni; }
    __device__ __forceinline__ const double* Rd(CPR pr) const { return Qd(pr) + 2 * pr.p * pr.ni; }
    __device__ __forceinline__ const double* uf(CPR pr) const { return Qd(pr) + 2 * pr.p * pr.ni + pr.p * pr.mi; }
    __device__ __forceinline__ alg_record* hist(CPR pr) const { return as_global(pr.hist) + (size_t)g * pr.hist_max; }
};
__device__ __forceinline__ Game game_view(CPR pr, int g) {
    Game G;
    G.base = as_global(pr.arena) + (size_t)g * pr.stride; G.g = g;
    G.zo[0] = 0; G.zo[1] = pr.o_z1; G.zo[2] = pr.o_z2;
    return G;
}

// Altro 0.3.0 cost_expansion!: a = (c >= 0) | (lambda > 0)  [PINNED test/constraints/constraint_derivatives.jl:28-34]
__device__ __forceinline__ double al_active_mu(double c, double lam, double mu) { return ((c >= 0.0) || (lam > 0.0)) ? mu : 0.0; }

// ---- extended constraints (all on knots 2..N; `k` below is the 0-based step, i.e. knot k+2 of the reference) ----------
__device__ __forceinline__ int ext_sb_row(CPR pr, int i, int k, int row) { return pr.col_len + pr.ctl_len + (i * (pr.N - 1) + k) * 2 * pr.n + row; }
__device__ __forceinline__ int ext_wall_row(CPR pr, int i, int k, int w) { return pr.col_len + pr.ctl_len + pr.sb_len + (i * (pr.N - 1) + k) * pr.nwall + w; }
__device__ __forceinline__ int ext_circ_row(CPR pr, int i, int k, int c) { return pr.col_len + pr.ctl_len + pr.sb_len + pr.wall_len + (i * (pr.N - 1) + k) * pr.ncirc + c; }
__device__ __forceinline__ const double* ext_sbmax(CPR pr, const double* ec) { return as_global(ec); }
__device__ __forceinline__ const double* ext_sbmin(CPR pr, const double* ec) { return as_global(ec) + pr.p * pr.n; }
__device__ __forceinline__ const double* ext_walls(CPR pr, const double* ec) { return as_global(ec) + 2 * pr.p * pr.n; }
__device__ __forceinline__ const double* ext_circs(CPR pr, const double* ec) { return as_global(ec) + 2 * pr.p * pr.n + 6 * ALG_MAX_WALLS; }
__device__ __forceinline__ const double* ext_walls3(CPR pr, const double* ec) { return as_global(ec) + 2 * pr.p * pr.n + 6 * ALG_MAX_WALLS + 3 * ALG_MAX_CIRCLES; }
__device__ __forceinline__ const double* ext_cyls(CPR pr, const double* ec) { return ext_walls3(pr, ec) + 12 * ALG_MAX_WALLS; }
__device__ __forceinline__ int ext_wall3_row(CPR pr, int i, int k, int w) { return pr.col_len + pr.ctl_len + pr.sb_len + pr.wall_len + pr.circ_len + (i * (pr.N - 1) + k) * pr.nwall3 + w; }
__device__ __forceinline__ int ext_cyl_row(CPR pr, int i, int k, int c) { return pr.col_len + pr.ctl_len + pr.sb_len + pr.wall_len + pr.circ_len + pr.wall3_len + (i * (pr.N - 1) + k) * pr.ncyl + c; }
// WallConstraint evaluate / jacobian! (wall_constraint.jl:57-96): c = ((x-x1) xv + (y-y1) yv) left right
__device__ __forceinline__ double wall_val(const double* W, int w, double x, double y, double* gx, double* gy) {
    const double x1 = W[w], y1 = W[ALG_MAX_WALLS + w], x2 = W[2 * ALG_MAX_WALLS + w], y2 = W[3 * ALG_MAX_WALLS + w];
    const double xv = W[4 * ALG_MAX_WALLS + w], yv = W[5 * ALG_MAX_WALLS + w];
    const double left = ((x - x1) * (x2 - x1) + (y - y1) * (y2 - y1) > 0.0) ? 1.0 : 0.0;
    const double right = ((x - x2) * (x1 - x2) + (y - y2) * (y1 - y2) > 0.0) ? 1.0 : 0.0;
    *gx = left * right * xv; *gy = left * right * yv;
    return ((x - x1) * xv + (y - y1) * yv) * left * right;
}
// TrajectoryOptimization 0.4.1 CircleConstraint: c = r^2 - (x-xc)^2 - (y-yc)^2
__device__ __forceinline__ double circ_val(const double* Cc, int c, double x, double y, double* gx, double* gy) {
    const double dx = x - Cc[c], dy = y - Cc[ALG_MAX_CIRCLES + c], r = Cc[2 * ALG_MAX_CIRCLES + c];
    *gx = -2.0 * dx; *gy = -2.0 * dy;
    return -(dx * dx) - (dy * dy) + r * r;
}
// Wall3DConstraint evaluate / jacobian! (wall_constraint.jl:186-236): c = (q - p1).v inside the slab spanned by (p1,p2), (p2,p3)
__device__ __forceinline__ double wall3_val(const double* W, int w, const double (&q)[3], double (&g)[3]) {
    const double* p1 = W + 12 * w; const double* p2 = p1 + 3; const double* p3 = p1 + 6; const double* v = p1 + 9;
    auto dot = [&](const double* a, const double* e, const double* c) { return (q[0] - a[0]) * (e[0] - c[0]) + (q[1] - a[1]) * (e[1] - c[1]) + (q[2] - a[2]) * (e[2] - c[2]); };
    const double left = dot(p1, p2, p1) > 0.0 ? 1.0 : 0.0, right = dot(p2, p1, p2) > 0.0 ? 1.0 : 0.0;
    const double bottom = dot(p3, p2, p3) > 0.0 ? 1.0 : 0.0, top = dot(p2, p3, p2) > 0.0 ? 1.0 : 0.0;
    const double in = left * right * bottom * top;
    g[0] = in * v[0]; g[1] = in * v[1]; g[2] = in * v[2];
    return ((q[0] - p1[0]) * v[0] + (q[1] - p1[1]) * v[1] + (q[2] - p1[2]) * v[2]) * in;
}
// CylinderConstraint evaluate / jacobian! (cylinder_constraint.jl:68-127): axis-aligned, c = r^2 - (distance to the axis)^2
// while 0 < (q - p)[axis] < l, else 0
__device__ __forceinline__ double cyl_val(const double* Y, int c, const double (&q)[3], double (&g)[3]) {
    const double* p = Y + 6 * c; const int ax = (int)p[3]; const double l = p[4], r = p[5];
    const double t0[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
    const double ta = ax == 0 ? t0[0] : (ax == 1 ? t0[1] : t0[2]);
    const double valid = (ta > 0.0 && ta < l) ? 1.0 : 0.0;
    const double out = r * r - t0[0] * t0[0] - t0[1] * t0[1] - t0[2] * t0[2] + ta * ta;
#pragma unroll
    for (int a = 0; a < 3; a++) g[a] = (a == ax) ? 0.0 : -valid * 2 * t0[a];
    return out * valid;
}

// ================================================================================================
// Step records.  The assemble pass leaves one compact record per time step k in HBM; the serial sweeps of the Newton
// direction read nothing else (plus the gains they spill themselves).
//   [coefk (NC)] [Hh (NS NPAIR): pair Hessian blocks at knot k+1] [Hd (NS P): sum_j Hh(i,j)]      (NS = 3, or 6 with 3-D positions)
//   (EXT only: [RQ (P n): diagonal state-bound Hessian of player i at knot k+1])
//   [rx (P n): rows opt_i,x_{k+1}]                                                                  <- LEN_COSTATE
//   [Rhat (m): R^ of knot k incl. reg] [ru (m): rows opt_i,u_{i,k}, joint order] [rd (n): dyn_k]      <- LEN_SWEEP
//   the pair gradient tables (PD P^2 per step, only used inside the assemble pass) follow the N - 1 records (Rec::gvt)
// ================================================================================================
template <class C> struct Rec {
    static constexpr int COEF = 0;
    static constexpr int HH = COEF + C::NC;
    static constexpr int HD = HH + C::NS * C::NPAIR;
    static constexpr int RQ = HD + C::NS * C::P;
    static constexpr int RX = RQ + (C::EXT ? C::P * C::n : 0);
    static constexpr int LEN_COSTATE = RX + C::P * C::n;          // the costate sweep reads [coef | Hh | Hd | RQ | rx] only
    static constexpr int RHAT = LEN_COSTATE;
    static constexpr int RU = RHAT + C::m;
    static constexpr int RD = RU + C::m;
    static constexpr int LEN_SWEEP = RD + C::n;
    static constexpr int LEN = LEN_SWEEP;                         // record stride: the sweeps stream whole records
    // pair-gradient tables (only used inside the assemble pass): behind the N - 1 records, TAB doubles per step
    static constexpr int TAB = C::PD * C::P * C::P;
    __device__ static constexpr int gvt(int N, int k) { return (N - 1) * LEN + k * TAB; }
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// LDS of the Newton-direction sweeps
template <class C, bool DENSE = C::DENSE> struct DirLds;
// dense variant (newton_direction_dense): everything of one backward step LDS-resident, records and gains read from HBM / L2
template <class C>
struct DirLds<C, true> {
    static constexpr int LDP = C::n + 1;                 // row stride of [P_i | s_i] and of [F | f] (odd: conflict-free column reads)
    static constexpr int WC = C::m + C::n + 1;           // [W | V A_k | g]
    static constexpr int CFL = C::NC > 0 ? C::NC : 1;
    static constexpr int RS_FULL = Rec<C>::LEN_SWEEP - C::NC;
    static constexpr int SYS_V = C::m * C::n, SYS_Y = C::P * C::n, SYS_W = C::m * WC, SYS_PC = 2 * C::m;
    static constexpr int TM = C::n * LDP, FX = (C::n + 1) * LDP;
    // TIGHT (ten players: [P_i | s_i] alone are 131 KB of the CU's 160): the regular layout would need 171 KB.  Two things change:
    // V, y and the pivot column live in rows 0 .. n-1 of Fx -- [F | f] is dead between the value recursion at the top of a step and the
    // closed-loop build at its end, which reads the solved system only (row n, the constant e_n, is never touched) -- and the blocks of
    // the step record that the Q-add reads exactly once ([RQ | rx]) are not staged: it takes them from the record in HBM / L2.
    static constexpr bool TIGHT = 8L * (C::P * C::n * LDP + FX + (TM > SYS_V + SYS_Y + SYS_W + SYS_PC ? TM : SYS_V + SYS_Y + SYS_W + SYS_PC) + CFL + RS_FULL + 8) > 160L * 1024;
    static_assert(!TIGHT || SYS_V + SYS_Y + SYS_PC <= TM, "TIGHT layout: V, y and the pivot column share rows 0 .. n-1 of Fx");
    // record offsets [SKIP0, SKIP1) are not staged (TIGHT: [RQ | rx]); offsets below address rs - NC, offsets above rs - NC - (SKIP1 - SKIP0)
    static constexpr int SKIP0 = TIGHT ? Rec<C>::RQ : Rec<C>::LEN_SWEEP, SKIP1 = TIGHT ? Rec<C>::RHAT : Rec<C>::LEN_SWEEP;
    static constexpr int RS_LEN = RS_FULL - (SKIP1 - SKIP0);
    struct Bwd {
        double Pm[C::P * C::n * LDP];                    // [P_i | s_i], row-major
        double Fx[FX];                                   // [[F f],[0 1]]
        struct Sys {
            double V[TIGHT ? 1 : SYS_V];                 // V[c][:] = B[:,c]' P_i(c)
            double y[TIGHT ? 1 : SYS_Y];                 // y_i = P_i rd + s_i
            double Wm[SYS_W];                            // augmented control system, row-major
            double pcol[TIGHT ? 1 : SYS_PC];             // pivot column of the Gauss-Jordan (double-buffered)
        };
        union {                                          // the value recursion's product and the control system are never live together
            double Tm[TM];                               // [P_i F | P_i f + s_i] of the player being advanced
            Sys sv;
        };
        // one step record: during the value recursion of step k the coefficient block still is step k + 1's (A_{k+1}'), then step
        // k's record is landed from the registers that prefetched it
        double cf[CFL];                                  // Jacobian coefficient block
        double rs[RS_LEN];                               // the record behind it ([Hh | Hd | RQ | rx | R^ | ru | rd]; TIGHT: [Hh | Hd | R^ | ru | rd])
        __device__ __forceinline__ double* V() { return TIGHT ? Fx : sv.V; }
        __device__ __forceinline__ double* y() { return TIGHT ? Fx + SYS_V : sv.y; }
        __device__ __forceinline__ double* pcol(int which) { return (TIGHT ? Fx + SYS_V + SYS_Y : sv.pcol) + which * C::m; }
    };
    struct Fwd { double dx[C::n], du[C::m], dl[2][C::P * C::n], cf[2][CFL], rs[2][RS_FULL], dxb[2][C::n]; };
    union { Bwd bw; Fwd fw; };
    double red[8];
};
template <class C>
struct DirLds<C, false> {
    static constexpr int LDP = C::n + 1;                 // padded row stride of P_i
    static constexpr int KB = C::n / 4;                  // k-blocks of the 16x16x4 f64 MFMA
    static constexpr int NHX = C::P * C::P * C::P * C::NS;   // expanded pair-Hessian table [i][jr][jc][NS]
    static_assert(C::n % 4 == 0 && C::n <= 16, "MFMA tile path needs n % 4 == 0 and n <= 16");
    static constexpr bool AUGS = C::n < 16;              // spare tile column: f and s_i ride through the MFMA products
    static constexpr int KB1 = AUGS ? (C::n + 4) / 4 : KB;   // k-blocks of the first product ([P_i | s_i]: n + 1 columns)
    static constexpr int VW = C::n + 1 + C::m;           // row of the extended V: [B' P (n) | g (1) | diag R^ slots (m)]
    struct Bwd {                       // live only during the backward sweep
        double Pm[C::P * C::n * LDP];  // [P_i | s_i], row-major (s_i in the pad column n)
        double Fx[16 * 16];            // [F | f | 0] (f in column n when n < 16); row n = e_n (n < 16), other rows >= n zero
        double fv[C::n];
        double t[C::P * C::n];         // t_i = P_i f + s_i (n == 16 path) / y_i = P_i rd + s_i
        double V[C::m * VW];
        double T[C::n * C::n];         // A_k transposed: T[c][r] = A_k[r][c] (column c of A_k contiguous, read by lane m + c)
        double pad[1];                 // dump slot for the masked-off lanes of the MFMA result write-back
    };
    struct Fwd {                       // live only during the forward / costate sweeps
        double kg[2][C::m * (C::n + 1)];
        double dx[C::n], du[C::m];
        double dl[C::P * C::n];
        double hx[NHX];                // expanded pair-Hessian table of the costate sweep
    };
    union { Bwd bw; Fwd fw; };
    double rec[2][Rec<C>::LEN_SWEEP];
    double coefn[C::NC > 0 ? C::NC : 1];
    double qdf[C::P * C::n];           // LQR diagonal of player i padded to joint dims (zero off pz[i])
};
// The assemble pass works out of registers and HBM/L2 (no LDS staging); the type is kept for the kernels' LDS union.
template <class C>
struct AsmLds {
    // phase A stages the record heads [coef | Hh | Hd] and gradient tables of WAVE / P steps here and writes them out as
    // contiguous segments (the (step, player) items would otherwise scatter 8..24-byte fragments over 64 records per store)
    // (only where four games share a SIMD and write traffic matters: the 256-VGPR configurations are latency-bound at their
    // batch sizes and write directly)
    static constexpr bool STAGED = (C::WPE == 4 && C::NW == 1);
    static constexpr int HEAD = Rec<C>::RQ, SL = HEAD + C::PD * C::P * C::P, SPP = C::NT / C::P;
    // Fused trial pass (assemble_fused; double integrator and unicycle, base constraint set): the rows of FT time steps are evaluated out of LDS.
    // A chunk holds x_k of its first step, the FT + 1 blocks [x_{k+1} | u_k | lambda_k] the rows touch (the last one only for
    // A_{k+1}' lambda_{k+1}), the [x | u] parts of the proximal reference, the pair-gradient tables and the LQR constants.
    // (one wavefront per game only: on a team the chunks' workgroup barriers cost more than the pass saves -- C5 loop, team of four: 105 vs 152 K/s)
    static constexpr bool FUSED = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE) && !C::EXT && !C::DENSE && (C::NW == 1 || 0);
    static constexpr int FT = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR && C::P == 3 && C::D == 2) ? FT_DI3 : (C::MODEL == ALG_MODEL_UNICYCLE && C::P == 3) ? FT_UNI3 : (C::MODEL == ALG_MODEL_UNICYCLE && C::P == 4) ? FT_UNI4 : 8, TAB = C::PD * C::P * C::P, NLQR = 2 * C::P * (C::ni + C::mi), NCF = C::NC > 0 ? C::NC : 1;
    struct Chunk { double xprev[C::n], zt[(FT + 1) * C::b], zxu[FT * (C::n + C::m)], gvt[FT * TAB], coef[C::NC > 0 ? (FT + 1) * NCF : 1], lqr[NLQR], dump[1]; };   // dump: target of the staging loop's masked-off stores
    struct NoChunk {};
    union {
        double stage[STAGED ? SPP * SL : 1];
        typename std::conditional<FUSED, Chunk, NoChunk>::type ch;
    };
};
// (team kernels of the base double integrator / unicycle: the line search stages [z | dz] of a search here, LsMulti in algames_assemble.hpp)
template <class C> struct LsLds {
    // (kernels of up to three players: a 4-player unicycle trajectory, b = 88 doubles per step, outgrows the buffer from N = 19 on)
    static constexpr bool ON = C::LS_STAGE && C::NW > 1 && C::P <= 3 && !AsmLds<C>::FUSED && !C::EXT && !C::DENSE && C::POS && (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE);
    static constexpr int CAP = LS_CAP;
    // teams of four run at most two per CU (team_width: B x 4 <= 2048): room for the group pass's per-step-size tables as well
    static constexpr bool SC_ON = ON && C::NW >= 4;
    static constexpr int SCAP = LS_SCAP;
    double z[ON ? CAP : 1];
    double sc[SC_ON ? SCAP : 1];
};
template <class C> union Lds { DirLds<C> d; AsmLds<C> a; LsLds<C> ls; };

// Pass-level instrumentation of the solver (same build flag): shader-clock cycles of the axpy / assemble phases / Newton direction as
// seen by thread 0 of the game, accumulated in LDS and flushed into G.res(pr)[16..] at the end of every newton_solve (slots:
// 16 axpy + barrier, 17 trial assemble, 18 #trials, 20 phase A, 21 rows x, 22 rows u, 23 rows d, 24 reductions, 25 #passes,
// 26 direction, 27 record pass, 28 #directions, 29 #record passes, 19 init_traj + rollout, 30 group passes of the line search (counted in 18), 31 whole solve)
#ifdef ALG_PHASE_PROF
__device__ __forceinline__ unsigned* lsp_slots() { __shared__ unsigned slots[32]; return slots; }
__device__ __forceinline__ unsigned lsp_now() { return (unsigned)__builtin_readcyclecounter(); }
__device__ __forceinline__ void lsp_add(int slot, unsigned t0) { if (game_tid() == 0) { const unsigned d = lsp_now() - t0; lsp_slots()[slot] += d < (1u << 28) ? d : 0u; } }
__device__ __forceinline__ void lsp_count(int slot) { if (game_tid() == 0) lsp_slots()[slot] += 1u; }
#define LSP_T0 unsigned lsp_t0_ = lsp_now();
#define LSP(slot) { lsp_add(slot, lsp_t0_); lsp_t0_ = lsp_now(); }
#define LSP_COUNT(slot) lsp_count(slot);
#else
#define LSP_T0
#define LSP(slot)
#define LSP_COUNT(slot)
#endif
} // namespace alg

// the phases of the path, one header each (this header is their common base: parameters, configurations, models, per-game view,
// step records, LDS layouts)
#include "algames_assemble.hpp"
#include "algames_direction.hpp"
#include "algames_solver.hpp"
