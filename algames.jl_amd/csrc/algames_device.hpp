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

// tunables (tests/probes/build_variant.sh builds A/B variants of the library with other values)
#ifndef ALG_ASM_UNROLL
#define ALG_ASM_UNROLL 2
#endif
// round-3 A/B switches of the tile path's sparse phases (all on; 0 restores the round-2 form of the phase, results are bit-identical either way):
//   ALG_DIROW  costate sweep with one 16-lane row per player (1: double integrator only, 2: all tile models)
//   ALG_GFUSE  g_c finished in the y_i lanes        ALG_SYSROW  rows of the augmented control system formed in the V phase
#ifndef ALG_DIROW
#define ALG_DIROW 2
#endif
#ifndef ALG_AXPY_U
#define ALG_AXPY_U 4
#endif


namespace alg {

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
template <int MODEL_, int P_, int D_, int EXT_ = 0, int NW_ = 1>
struct Cfg {
    static constexpr int MODEL = MODEL_, P = P_, D = D_;
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
#define ALG_SWEEP_DEPTH 4
#endif
    // (measured, depth 1 / 2 / 4 / 8: C2 10.22 / 10.28 / 10.37 / 10.33 M/s, C3 2.38 / 2.43 / 2.43 / 2.44 M/s, C5 loop 154 / 158 / 157 / 156 K/s)
#ifndef ALG_SWEEP_DEPTH_W2
#define ALG_SWEEP_DEPTH_W2 2       // 256-register kernels
#endif
    static constexpr int SWEEP_DEPTH = WPE == 4 ? ALG_SWEEP_DEPTH : (ALG_SWEEP_DEPTH < ALG_SWEEP_DEPTH_W2 ? ALG_SWEEP_DEPTH : ALG_SWEEP_DEPTH_W2);
    // rows per lane and pass of the assemble row loops (memory-level parallelism against the L2 / store-ack latency)
    static constexpr int ASM_UNROLL = ALG_ASM_UNROLL;
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
__device__ __forceinline__ void game_sync() { __syncthreads(); }
// ---- wave reductions ---------------------------------------------------------------------------
// Opaque copy of the lane id: keeps per-lane role / address computations of a phase from being hoisted out of the
// solver's outer loops (where every phase's invariants would be live at once).
__device__ __forceinline__ int phase_lane() { int l = game_tid(); asm volatile("" : "+v"(l)); return l; }
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
#ifdef ALG_TEAM_FULL_FENCE        // A/B builds
    if constexpr (C::NW > 1) { dir_sync<C>(); return; }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Workgroup barrier that orders LDS only (no vmcnt drain: the record prefetch and the gain stores of a sweep step stay in flight)
__device__ __forceinline__ void team_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <class C> __device__ __forceinline__ void rotate_priority(int it);
template <class C> __device__ __forceinline__ int team_wave() { return C::NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(game_tid() >> 6)); }

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
template <class C>
__device__ __forceinline__ void model_player_rk3(CPR pr, int i, const double* x, const double* u, double dt, double* xn) {
    static_assert(!C::QUAD, "the quadrotor integrates through quad_rk3 (rollout)");
    double xi[C::ni], k1[C::ni], k2[C::ni], k3[C::ni], t[C::ni], ui[C::mi];
#pragma unroll
    for (int j = 0; j < C::ni; j++) xi[j] = x[i + j * C::P];
#pragma unroll
    for (int j = 0; j < C::mi; j++) ui[j] = u[i + j * C::P];
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
        return 0.5 * dt * dt * v(c) + dt * v(c + C::m);
    } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
        const int P = C::P, i = c % P; const bool k0 = (c / P) == 0;
        const double ca = coef[(k0 ? 5 : 7) * P + i], cb = coef[(k0 ? 6 : 8) * P + i], cc = coef[(k0 ? 4 : 9) * P + i];
        return (k0 ? 0.5 * dt : 1.0) * (ca * v(i) + cb * v(P + i) + cc * v(3 * P + i)) + (k0 ? dt : 0.0) * v(2 * P + i);
    } else {
        const int P = C::P, i = c % P, kind = c / P;
        return 0.5 * dt * (coef[kind * P + i] * v(i) + coef[(2 + kind) * P + i] * v(P + i)) + dt * v((2 + kind) * P + i);
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
        H.base = reinterpret_cast<double*>(uniform_u64(reinterpret_cast<unsigned long long>(base)));
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
    __device__ __forceinline__ double* lam(CPR pr) const { return pr.con + (size_t)g * pr.con_stride; }
    __device__ __forceinline__ double* mu(CPR pr) const { return lam(pr) + pr.con_pad; }
    __device__ __forceinline__ double* vals(CPR pr) const { return lam(pr) + 2 * pr.con_pad; }
    __device__ __forceinline__ const double* Qd(CPR pr) const { return pr.lqr + (size_t)g * pr.lqr_stride; }
    __device__ __forceinline__ const double* xf(CPR pr) const { return Qd(pr) + pr.p * pr.ni; }
    __device__ __forceinline__ const double* Rd(CPR pr) const { return Qd(pr) + 2 * pr.p * pr.ni; }
    __device__ __forceinline__ const double* uf(CPR pr) const { return Qd(pr) + 2 * pr.p * pr.ni + pr.p * pr.mi; }
    __device__ __forceinline__ alg_record* hist(CPR pr) const { return pr.hist + (size_t)g * pr.hist_max; }
};
__device__ __forceinline__ Game game_view(CPR pr, int g) {
    Game G;
    G.base = pr.arena + (size_t)g * pr.stride; G.g = g;
    G.zo[0] = 0; G.zo[1] = pr.o_z1; G.zo[2] = pr.o_z2;
    return G;
}

// Altro 0.3.0 cost_expansion!: a = (c >= 0) | (lambda > 0)  [PINNED test/constraints/constraint_derivatives.jl:28-34]
__device__ __forceinline__ double al_active_mu(double c, double lam, double mu) { return ((c >= 0.0) || (lam > 0.0)) ? mu : 0.0; }

// ---- extended constraints (all on knots 2..N; `k` below is the 0-based step, i.e. knot k+2 of the reference) ----------
__device__ __forceinline__ int ext_sb_row(CPR pr, int i, int k, int row) { return pr.col_len + pr.ctl_len + (i * (pr.N - 1) + k) * 2 * pr.n + row; }
__device__ __forceinline__ int ext_wall_row(CPR pr, int i, int k, int w) { return pr.col_len + pr.ctl_len + pr.sb_len + (i * (pr.N - 1) + k) * pr.nwall + w; }
__device__ __forceinline__ int ext_circ_row(CPR pr, int i, int k, int c) { return pr.col_len + pr.ctl_len + pr.sb_len + pr.wall_len + (i * (pr.N - 1) + k) * pr.ncirc + c; }
__device__ __forceinline__ const double* ext_sbmax(CPR pr, const double* ec) { return ec; }
__device__ __forceinline__ const double* ext_sbmin(CPR pr, const double* ec) { return ec + pr.p * pr.n; }
__device__ __forceinline__ const double* ext_walls(CPR pr, const double* ec) { return ec + 2 * pr.p * pr.n; }
__device__ __forceinline__ const double* ext_circs(CPR pr, const double* ec) { return ec + 2 * pr.p * pr.n + 6 * ALG_MAX_WALLS; }
__device__ __forceinline__ const double* ext_walls3(CPR pr, const double* ec) { return ec + 2 * pr.p * pr.n + 6 * ALG_MAX_WALLS + 3 * ALG_MAX_CIRCLES; }
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
    struct Bwd {
        double Pm[C::P * C::n * LDP];                    // [P_i | s_i], row-major
        double Fx[(C::n + 1) * LDP];                     // [[F f],[0 1]]
        struct Sys {
            double V[C::m * C::n];                       // V[c][:] = B[:,c]' P_i(c)
            double y[C::P * C::n];                       // y_i = P_i rd + s_i
            double Wm[C::m * WC];                        // augmented control system, row-major
            double pcol[2][C::m];                        // pivot column of the Gauss-Jordan (double-buffered)
        };
        union {                                          // the value recursion's product and the control system are never live together
            double Tm[C::n * LDP];                       // [P_i F | P_i f + s_i] of the player being advanced
            Sys sv;
        };
        // one step record: during the value recursion of step k the coefficient block still is step k + 1's (A_{k+1}'), then step
        // k's record is landed from the registers that prefetched it
        double cf[CFL];                                  // Jacobian coefficient block
        double rs[Rec<C>::LEN_SWEEP - C::NC];            // the record behind it ([Hh | Hd | RQ | rx | R^ | ru | rd])
    };
    struct Fwd { double dx[C::n], du[C::m], dl[2][C::P * C::n], cf[2][CFL], rs[2][Rec<C>::LEN_SWEEP - C::NC], dxb[2][C::n]; };
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
#ifndef ALG_FUSED_TEAMS
#define ALG_FUSED_TEAMS 0     // 1 (A/B builds): the fused pass on teams as well -- loses with LDS-only barriers too: C3 2.72 vs 2.79 M/s, C5 loop 122 vs 152 K/s
#endif
    static constexpr bool FUSED = (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE) && !C::EXT && !C::DENSE && (C::NW == 1 || ALG_FUSED_TEAMS);
    static constexpr int FT = 8, TAB = C::PD * C::P * C::P, NLQR = 2 * C::P * (C::ni + C::mi), NCF = C::NC > 0 ? C::NC : 1;
    struct Chunk { double xprev[C::n], zt[(FT + 1) * C::b], zxu[FT * (C::n + C::m)], gvt[FT * TAB], coef[(FT + 1) * NCF], lqr[NLQR]; };
    struct NoChunk {};
    union {
        double stage[STAGED ? SPP * SL : 1];
        typename std::conditional<FUSED, Chunk, NoChunk>::type ch;
    };
};
template <class C> union Lds { DirLds<C> d; AsmLds<C> a; };

// Pass-level instrumentation of the solver (same build flag): shader-clock cycles of the axpy / assemble phases / Newton direction as
// seen by thread 0 of the game, accumulated in LDS and flushed into G.res(pr)[16..] at the end of every newton_solve (slots:
// 16 axpy + barrier, 17 trial assemble, 18 #trials, 20 phase A, 21 rows x, 22 rows u, 23 rows d, 24 reductions, 25 #passes,
// 26 direction, 27 record pass, 28 #directions, 29 #record passes)
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

struct AsmAcc { double l1 = 0, l1r = 0, l1f = 0, vopt = 0, vdyn = 0, vcon = 0, vsta = 0; int bad = 0; };
// Phase A of the assemble pass (see assemble_pass): RK2 Jacobian coefficients and the pair / wall / circle terms of every (knot, player).
// dzp != nullptr: the positions are those of the trial iterate z + alpha dz, formed on the fly (fused trial pass of the double integrator).
template <class C, int MODE, bool IBR>
__device__ __forceinline__ void assemble_phase_a(CPR pr, const Game& G, AsmLds<C>& L, const double* __restrict__ z, const double* __restrict__ dzp, double alpha,
                                                 int N, int lane, double dt, int ip, AsmAcc& acc) {
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
                auto sv = [&](int idx) { const double q = sk[idx]; return dk ? q + alpha * dk[idx] : q; };
                auto uv = [&](int j) { const double q = z[uo_ + j]; return dzp ? q + alpha * dzp[uo_ + j] : q; };
                const double th = sv(2 * P + i), v = sv(3 * P + i);
                const double om = uv(0), ac = uv(1);
                const double thm = th + (om * dt) * 0.5, vm = v + (ac * dt) * 0.5;
                double sn, cs; sincos(thm, &sn, &cs);
                rec[R::COEF + 0 * P + i] = -dt * vm * sn; rec[R::COEF + 1 * P + i] = dt * cs;
                rec[R::COEF + 2 * P + i] = dt * vm * cs;  rec[R::COEF + 3 * P + i] = dt * sn;
            }
            if constexpr (C::POS) {
                constexpr int PD = C::PD, NS = C::NS;
                const double w = (kn < N - 1) ? dt : 1.0;
                const double* x1 = z + n + hx<C>(k); const double* d1 = dzp ? dzp + n + hx<C>(k) : nullptr;
                auto xp = [&](int idx) { const double v = x1[idx]; return d1 ? v + alpha * d1[idx] : v; };    // position of the (trial) iterate
                double xi[PD], ga[PD], dd[NS];
#pragma unroll
                for (int a = 0; a < PD; a++) { xi[a] = xp(a * P + i); ga[a] = 0.0; }
#pragma unroll
                for (int t = 0; t < NS; t++) dd[t] = 0.0;
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
#pragma unroll
                        for (int a = 0; a < PD; a++) dl[a] = xi[a] - xp(a * P + j);
                        const double dl0 = dl[0], dl1 = dl[1];
                        const double s2 = dl0 * dl0 + dl1 * dl1;
                        if (pr.has_colcost) {                                    // CollisionCost, objective.jl:134-173 (planar: px[i])
                            const double nrm = sqrt(s2), mu = pr.cc_mu[i], rad = pr.cc_radius[i];
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
                            const double Rr = pr.ca_pair_r[i * MAXP + j];
                            const double on = (double)((pr.ca_mask[i] >> j) & 1u);                    // 0: this ordered pair carries no constraint
                            double s2c = s2;
                            if constexpr (PD == 3) { if (pr.ca_dim != 3) dl[2] = 0.0; s2c += dl[2] * dl[2]; }   // spherical: pz[i][1:3]
                            const double c = on * (Rr * Rr - s2c);
                            const int ci = con_col<C>(N, pairq<C>(i, j), kn);
                            const double lm = G.lam(pr)[ci], am = on * al_active_mu(c, lm, G.mu(pr)[ci]);
                            const double wl = fma(am, c, on * lm);            // (the contraction the all-pairs form always had: lm + am c)
#pragma unroll
                            for (int a = 0; a < PD; a++) {
                                gv[a] += -2.0 * dl[a] * wl;
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
                        const double lm = G.lam(pr)[ci], am = al_active_mu(c, lm, G.mu(pr)[ci]);
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
          }
          if constexpr (STAGED) {
              game_sync();
              // write-out: contiguous [coef | Hh | Hd] and table segments of the staged steps
              const int nst = (N - 1 - kA) < SPP ? (N - 1 - kA) : SPP;
              for (int t = lane; t < nst * SL; t += C::NT) {
                  const int ks2 = t / SL, o = t % SL;
                  const size_t base = (size_t)(kA + ks2) * R::LEN;
                  if (o >= HEAD) G.rec(pr)[R::gvt(N, kA + ks2) + (o - HEAD)] = L.stage[t];
                  else if (RECS || o < C::NC) G.rec(pr)[base + o] = L.stage[t];
              }
              game_sync();
          }
        }
        if constexpr (!AsmLds<C>::STAGED) game_sync();
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
            for (int j = 0; j < 4; j++) uj[j] = Jet{z[n + hu<C>(k, i) + j], c == 12 + j ? 1.0 : 0.0};
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
        if (RECS) G.rec(pr)[q.rec_off] = r;
        if (MODE == 2) G.res(pr)[q.vrow] = r;
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
            double r = -z[zo + (uidx)(n + m + ei)];
            {
                // A_{k+1}' lambda_{i,k+1}: addresses clamped to block k when there is no next block, the term is dropped below
                const uidx lo = zo + (uidx)((has_next ? b : 0) + n + m + i * n), co = ro + (uidx)((has_next ? R::LEN : 0) + R::COEF);
                const double t = AT_vec<C>(recg + co, dt, [&](int rr) { return z[lo + (uidx)rr]; }, a);
                r += has_next ? t : 0.0;
            }
            const bool own = (a % P == i);
            const double tqv = G.Qd(pr)[i * ni + a / P], txv = G.xf(pr)[i * ni + a / P];
            const double tq = own ? tqv : 0.0, tx = own ? txv : 0.0;
            const double xa = z[zo + (uidx)a];
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
                            const double lm = G.lam(pr)[ci], am = al_active_mu(cv, lm, G.mu(pr)[ci]);
                            const double wl = lm + am * cv;
                            r += (half == 0 ? wl : -wl); qsb += am;
                            if (!IBR || i == ip) vsta = fmax(vsta, fmax(0.0, cv));
                        }
                    }
                }
                if (RECS && ok) G.rec(pr)[ro + (uidx)(R::RQ + ei)] = qsb;
            }
            q.mine = IBR ? (i == ip) : true;
            q.dprox = 0.0;
            if (zref) { const double xr = zref[zo + (uidx)a]; q.dprox = q.mine ? xa - xr : 0.0; }
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
            const double u = z[zo + (uidx)(n + uoff<C>(c))];
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
                        const double lm = G.lam(pr)[ci], am = al_active_mu(cv, lm, G.mu(pr)[ci]);
                        const double wl = lm + am * cv;
                        g += (half == 0 ? wl : -wl); rhat += am;
                        if (!IBR) vcon = fmax(vcon, fmax(0.0, cv));
                        else if ((pr.ibr_ctl_rows[ip] >> (half * m + c)) & 1ull) vcon = fmax(vcon, fmax(0.0, cv));
                    }
                }
            }
            q.r = dt * (tr * (u - tu)) + g + BT_vec<C>(recg + ro + (uidx)R::COEF, dt, [&](int rr) { return z[lo + (uidx)rr]; }, c);
            q.mine = IBR ? (i == ip) : true;
            q.dprox = 0.0;
            if (zref) { const double ur = zref[zo + (uidx)(n + uoff<C>(c))]; q.dprox = q.mine ? u - ur : 0.0; }
            if (RECS && ok) G.rec(pr)[ro + (uidx)(R::RHAT + c)] = rhat;
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
                const double uj = z[zo + (uidx)(n + uoff<C>(j))], base = z[po + (uidx)a], vel = z[po + (uidx)(j + m)];
                const double vm = vel + (uj * dt) * 0.5;
                xn = base + (a < m ? vm : uj) * dt;
            } else if constexpr (C::MODEL == ALG_MODEL_BICYCLE) {
                const int blkk = a / P, i = a % P;
                const double ua = z[zo + (uidx)(n + uoff<C>(i))], base = z[po + (uidx)a], vel = z[po + (uidx)(2 * P + i)];
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 5 : (blkk == 1 ? 6 : 4)) * P + i];    // dt cos th / dt sin th / dt sin(beta)/lr
                xn = (blkk == 2) ? base + ua * dt : base + vm * cf;
            } else {
                const int blkk = a / P, i = a % P;
                const double ua = z[zo + (uidx)(n + uoff<C>(P + i))], base = z[po + (uidx)a], vel = z[po + (uidx)(3 * P + i)];
                const double uo = z[zo + (uidx)(n + uoff<C>((blkk >= 2 ? blkk - 2 : 0) * P + i))];
                const double vm = vel + (ua * dt) * 0.5;
                const double cf = Ck[(blkk == 0 ? 1 : 3) * P + i];                      // dt cos(thm) / dt sin(thm)
                xn = (blkk <= 1) ? base + vm * cf : base + uo * dt;
            }
            q.r = xn - z[zo + (uidx)a];
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
template <class C, int MODE, bool AXPY>
__device__ void assemble_fused(CPR pr0, const Game& G0, AsmLds<C>& L, double alpha, bool prox, double reg, double jreg, ResOut& out) {
    static_assert(AsmLds<C>::FUSED && (MODE == 0 || MODE == 1 || MODE == 3), "fused trial pass: double integrator / unicycle, statistics / record modes");
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
    assemble_phase_a<C, MODE, false>(pr, G, L, zs, AXPY ? dz : nullptr, alpha, N, lane, dt, -1, acc);
    LSP(20)
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
        if (RECS) recg[rec_off] = r;
    };
    for (int k0 = 0; k0 < N - 1; k0 += FT) {
        const int nst = (N - 1 - k0) < FT ? (N - 1 - k0) : FT;            // steps of this chunk
        const int nblk = (k0 + nst < N - 1) ? nst + 1 : nst;             // blocks staged: the chunk's and the next one (A' lambda_{k+1})
        // ---- stage: x_k of the first step, then blocks k0 .. k0 + nblk - 1; the trial blocks of the chunk's own steps go out here
        if (lane < n) {
            const int src = k0 == 0 ? lane : n + (k0 - 1) * b + lane;
            double v = zs[src];
            if (AXPY && k0 > 0) v = v + alpha * dz[src];                  // (x_1 does not move)
            Ch.xprev[lane] = v;
        }
        {
            const int base = n + k0 * b, cnt = nblk * b, own = nst * b;
            for (int e0 = lane; e0 < cnt; e0 += 4 * NT) {
                double a[4], d[4];
#pragma unroll
                for (int t = 0; t < 4; t++) { const int e = e0 + t * NT, ec = e < cnt ? e : e0; a[t] = zs[base + ec]; d[t] = AXPY ? dz[base + ec] : 0.0; }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int e = e0 + t * NT;
                    if (e < cnt) {
                        const double v = AXPY ? a[t] + alpha * d[t] : a[t];
                        Ch.zt[e] = v;
                        if (AXPY && e < own) zo[base + e] = v;
                        const int j = e / b, o = e % b;
                        if (o < NXU && j < nst) Ch.zxu[j * NXU + o] = a[t];
                    }
                }
            }
            if constexpr (C::POS) {
                const int tcnt = nst * TAB;
                for (int e = lane; e < tcnt; e += NT) Ch.gvt[e] = recg[R::gvt(N, k0) + e];            // contiguous behind the records
            }
            if constexpr (NC > 0) {                                        // Jacobian coefficients of the staged steps (phase A left them in the records)
                const int ccnt = nblk * NC;
                for (int e = lane; e < ccnt; e += NT) Ch.coef[e] = recg[(size_t)(k0 + e / NC) * R::LEN + R::COEF + e % NC];
            }
        }
        fsync();
        // ---- rows opt_i,x_{k+1}[a]
        for (int e = lane; e < nst * P * n; e += NT) {
            const int ks = e / (P * n), ei = e % (P * n), i = ei / n, a = ei % n, k = k0 + ks;
            const double* blk = Ch.zt + ks * b;
            const bool has_next = (k + 1 <= N - 2);
            const double w = (k + 1 < N - 1) ? dt : 1.0;
            double r = -blk[n + m + ei];
            {
                const double* ln = blk + (has_next ? b : 0) + n + m + i * n;
                const double t = AT_vec<C>(Ch.coef + (ks + (has_next ? 1 : 0)) * NC, dt, [&](int rr) { return ln[rr]; }, a);
                r += has_next ? t : 0.0;
            }
            const bool own = (a % P == i);
            const double tqv = lQd[i * ni + a / P], txv = lxf[i * ni + a / P];
            const double tq = own ? tqv : 0.0, tx = own ? txv : 0.0;
            const double xa = blk[a];
            r += w * (tq * (xa - tx));
            if (C::POS) { const double gv = Ch.gvt[ks * TAB + (i * P + a % P) * C::PD + (a < C::PD * P ? a / P : 0)]; r += (a < C::PD * P) ? gv : 0.0; }
            const double dprox = prox ? xa - Ch.zxu[ks * NXU + a] : 0.0;
            finish(r, dprox, false, (unsigned)(k * R::LEN + R::RX + ei));
        }
        LSP(21)
        // ---- rows opt_i,u_{i,k}[c]
        for (int e = lane; e < nst * m; e += NT) {
            const int ks = e / m, c = e % m, i = c % P, k = k0 + ks;
            const double* blk = Ch.zt + ks * b;
            const double u = blk[n + uoff<C>(c)];
            const double* lo = blk + n + m + i * n;
            const double tr = lRd[(c % P) * mi + c / P], tu = luf[(c % P) * mi + c / P];
            double g = 0.0, rhat = dt * tr + jreg;
            if (pr.has_ctl) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int ci = con_ctl<C>(pr, k, half * m + c);
                    const double cv = half == 0 ? u - pr.umax[c] : pr.umin[c] - u;
                    if (isfinite(cv)) {
                        const double lm = G.lam(pr)[ci], am = al_active_mu(cv, lm, G.mu(pr)[ci]);
                        const double wl = lm + am * cv;
                        g += (half == 0 ? wl : -wl); rhat += am;
                        acc.vcon = fmax(acc.vcon, fmax(0.0, cv));
                    }
                }
            }
            const double r = dt * (tr * (u - tu)) + g + BT_vec<C>(Ch.coef + ks * NC, dt, [&](int rr) { return lo[rr]; }, c);
            const double dprox = prox ? u - Ch.zxu[ks * NXU + n + uoff<C>(c)] : 0.0;
            if (RECS) recg[(size_t)k * R::LEN + R::RHAT + c] = rhat;
            finish(r, dprox, false, (unsigned)(k * R::LEN + R::RU + c));
        }
        LSP(22)
        // ---- rows dyn_k[a] = RK2(x_k, u_k)[a] - x_{k+1}[a]   (explicit midpoint; the expressions of assemble_pass)
        for (int e = lane; e < nst * n; e += NT) {
            const int ks = e / n, a = e % n, k = k0 + ks;
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

// update_traj!(target, source, alpha, delta) (primal_dual_traj.jl:109-128): coalesced axpy over the S entries
typedef double double2_t __attribute__((ext_vector_type(2)));
template <class C>
__device__ __forceinline__ void update_traj(CPR pr0, const Game& G0, int tsel, int ssel, double alpha) {
    CPR pr = phase_params(pr0);
    const Game G = G0.fresh();
    double* tgt = G.z(tsel); const double* src = G.z(ssel); const double* dz = G.z(2);
    // pure streaming pass: 16 bytes per lane and four independent load pairs in flight per pass.  Every game's buffers start
    // 16-byte aligned when traj_len is even (n is always even); otherwise the scalar loop runs.
    constexpr int U = ALG_AXPY_U;
    const int S = phase_int(pr.S), lane = phase_lane();
    if ((pr.traj_len & 1) == 0) {
        const int S2 = S >> 1;                           // pairs; a last odd element is handled below
        const double2_t* __restrict__ s2 = reinterpret_cast<const double2_t*>(src + C::n);
        const double2_t* __restrict__ d2 = reinterpret_cast<const double2_t*>(dz + C::n);
        double2_t* __restrict__ t2 = reinterpret_cast<double2_t*>(tgt + C::n);
        for (int e0 = lane; e0 < S2; e0 += U * C::NT) {
            double2_t a[U], d[U];
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; const int ec = e < S2 ? e : e0; a[t] = s2[ec]; d[t] = d2[ec]; }
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < S2) { double2_t v; v.x = a[t].x + alpha * d[t].x; v.y = a[t].y + alpha * d[t].y; t2[e] = v; } }
        }
        if ((S & 1) && lane == 0) tgt[C::n + S - 1] = src[C::n + S - 1] + alpha * dz[C::n + S - 1];
    } else {
        for (int e0 = lane; e0 < S; e0 += U * C::NT) {
            double a[U], d[U];
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; const int ec = e < S ? e : e0; a[t] = src[C::n + ec]; d[t] = dz[C::n + ec]; }
#pragma unroll
            for (int t = 0; t < U; t++) { const int e = e0 + t * C::NT; if (e < S) tgt[C::n + e] = a[t] + alpha * d[t]; }
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
// 1/x for a pivot (normal, non-zero): hardware reciprocal + one Newton step instead of the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
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
#ifndef ALG_RDG_ASM4
#define ALG_RDG_ASM4 0        // (measured neutral on C2 / C3 / C5, profiles/r04_ab_asm4_*.txt: off) four terms of a chain per asm statement (the compiler pads every statement boundary of inline asm with an s_nop)
#endif
template <int L0, bool FIRST>
__device__ __forceinline__ void fmac_rowbcast4(double& acc, double v, double p0, double p1, double p2, double p3) {
#define ALG_F4 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t" \
               "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
    if constexpr (FIRST) asm volatile("s_nop 1\n\t" ALG_F4 : "+v"(acc) : "v"(v), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "n"(L0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
    else asm volatile(ALG_F4 : "+v"(acc) : "v"(v), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "n"(L0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
#undef ALG_F4
}
template <int C0, int G, int NC, int I = 0>
__device__ __forceinline__ void rowdot_group_fmac(double& acc, double v, const double (&buf)[G]) {
    if constexpr (ALG_RDG_ASM4 && I + 3 < G && C0 + I + 3 < NC) {
        fmac_rowbcast4<C0 + I, C0 + I == 0>(acc, v, buf[I], buf[I + 1], buf[I + 2], buf[I + 3]); rowdot_group_fmac<C0, G, NC, I + 4>(acc, v, buf);
    } else if constexpr (I < G && C0 + I < NC) { fmac_rowbcast<C0 + I, C0 + I == 0>(acc, v, buf[I]); rowdot_group_fmac<C0, G, NC, I + 1>(acc, v, buf); }
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
#ifndef ALG_GJSPEC
#define ALG_GJSPEC 1          // 1: reciprocal of the diagonal candidate started before the pivot test, pivot search as a max tree (bit-identical)
#endif
// max |b[0..NB-1]|: v_max_f64 with |.| modifiers (through fmax() the compiler canonicalises every operand first), ONE asm statement
// (the compiler pads every statement boundary with an s_nop).  ALG_GJSPEC: a tree of depth <= 3 instead of a chain of depth NB - 1 --
// max is exact and NaN-dropping either way, so the result is the same number; a lone wavefront waits for every link of the chain.
template <int NB>
__device__ __forceinline__ double absmax_below(const double* b) {
    static_assert(NB >= 1 && NB <= 7, "control system larger than the DPP elimination supports");
    double oth;
    if constexpr (NB == 1) oth = fabs(b[0]);
    else if constexpr (NB == 2) asm("v_max_f64 %0, |%1|, |%2|" : "=v"(oth) : "v"(b[0]), "v"(b[1]));
    else if constexpr (NB == 3) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|" : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]));
#if ALG_GJSPEC
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
#else
    else if constexpr (NB == 4) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|" : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    else if constexpr (NB == 5) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|\n\tv_max_f64 %0, %0, |%5|"
                                    : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]));
    else if constexpr (NB == 6) asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|\n\tv_max_f64 %0, %0, |%5|\n\tv_max_f64 %0, %0, |%6|"
                                    : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]));
    else asm("v_max_f64 %0, |%1|, |%2|\n\tv_max_f64 %0, %0, |%3|\n\tv_max_f64 %0, %0, |%4|\n\tv_max_f64 %0, %0, |%5|\n\tv_max_f64 %0, %0, |%6|\n\tv_max_f64 %0, %0, |%7|"
             : "=&v"(oth) : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]));
#endif
    return oth;
}
template <int M, int C>
__device__ __forceinline__ void gj_dpp_pivot(double (&col)[M], int& sing) {
    // lane C of the row owns W's column C: its entries below the diagonal decide the pivot (wave-uniform: every row holds the
    // same replica; bit C of the ballot is row 0's lane C)
    double best = fabs(col[C]);
    // ALG_GJSPEC: the diagonal entry is the pivot unless the test below says otherwise (it almost never does: W = R^ + V B is
    // dominated by its diagonal), so its broadcast and reciprocal -- rcp + four dependent FMAs -- start BEFORE the pivot search and
    // the two dependency chains overlap; a row exchange repeats them on the exchanged entry.  Same operations on the same numbers.
    double pvt, rpv = 0.0;
#if ALG_GJSPEC
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(col[C]), "n"(C));
    rpv = fast_rcp(pvt);
#endif
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
#if ALG_GJSPEC
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(col[C]), "n"(C));
        rpv = fast_rcp(pvt);
#endif
    }
#if !ALG_GJSPEC
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pvt) : "v"(col[C]), "n"(C));
    rpv = fast_rcp(pvt);
#endif
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
                else if (e < R::LEN_SWEEP) B.rs[e - C::NC] = pre[q];
            }
        } else {
            const double* Rk = recs + (size_t)kk * R::LEN;
            for (int e = tid; e < C::NC; e += BT) B.cf[e] = Rk[e];
            for (int e = tid; e < R::LEN_SWEEP - C::NC; e += BT) B.rs[e] = Rk[C::NC + e];
        }
    };
    rec_load(N - 2);
    game_sync();
    ALG_PROF_DECL
    // ------------------------------------------------------------------ backward sweep
    for (int k = N - 2; k >= 0; k--) {
        const double* Rl = B.rs - C::NC;                                     // record offsets >= NC address the staged copy
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
            if constexpr (C::EXT) qd += Rl[R::RQ + e];
            row[r] += qd;
            row[n] += Rl[R::RX + e];
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
        }, [&](int e, double v) { B.sv.V[e] = v; });
        flat_loop<FU>(tid, BT, P * n, [&](int e) {
            const double* Pr = &B.Pm[e * LDP];
            double a = Pr[n];
            for (int c = 0; c < n; c++) a += Pr[c] * Rl[R::RD + c];
            return a;
        }, [&](int e, double v) { B.sv.y[e] = v; });
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
            const int c = e / m, t = e % m; const double* Vc = &B.sv.V[c * n];
            return ibr_mask(c, t, BT_vec<C>(coefk, dt, [&](int rr) { return Vc[rr]; }, t) + (t == c ? Rl[R::RHAT + c] : 0.0));
        }, [&](int e, double v) { B.sv.Wm[(e / m) * WC + e % m] = v; });
        flat_loop<FU>(tid, BT, m * n, [&](int e) {
            const int c = e / n, col = e % n; const double* Vc = &B.sv.V[c * n];
            return ibr_mask(c, m + col, (k >= 1) ? AT_vec<C>(coefk, dt, [&](int rr) { return Vc[rr]; }, col) : 0.0);    // dx_1 = 0: A_0 never acts
        }, [&](int e, double v) { B.sv.Wm[(e / n) * WC + m + e % n] = v; });
        flat_loop<1>(tid, BT, m, [&](int c) {
            const double* yi = &B.sv.y[(c % P) * n];
            return ibr_mask(c, m + n, Rl[R::RU + c] + BT_vec<C>(coefk, dt, [&](int rr) { return yi[rr]; }, c));
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
                    for (int r = 0; r < m; r++) B.sv.pcol[c & 1][r] = col[0][r];
                }
                game_sync();
                double pc[m];
#pragma unroll
                for (int r = 0; r < m; r++) pc[r] = B.sv.pcol[c & 1][r];
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
                const double base = col < n ? A_entry<C>(coefk, dt, r, col) : Rl[R::RD + r];
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
template <class C, bool SKIP_S = false, bool SREG = false>
__device__ __forceinline__ void player_tail_half(DirLds<C>& L, const double* Rc, double dt, int k, int N, int first, int lane, double sreg = 0.0) {
    constexpr int n = C::n, m = C::m, P = C::P, LDP = DirLds<C>::LDP, VW = DirLds<C>::VW;
    static_assert(n == 16 && P == 4 && C::MODEL == ALG_MODEL_UNICYCLE, "lane layout of the team-of-two tail (4-player unicycle)");
    using R = Rec<C>;
    const double* coefk = Rc + R::COEF;
    if constexpr (!DirLds<C>::AUGS && !SKIP_S) {
        s_half<C>(L, Rc, dt, k < N - 2, first, lane);
        sweep_sync<C>();
    }
    const int h = lane >> 5, yp = first + 2 * ((lane >> 4) & 1), yr = lane & 15;
    const double* Pr = &L.bw.Pm[yp * n * LDP + yr * LDP];
    const double rdl = Rc[R::RD + ((yr + 8 * h) & 15)];          // lane j of an upper-half row holds rd[j + 8]
    const double* Ph = Pr + 8 * h;
    double a = h ? 0.0 : (SREG ? sreg : Pr[n]);          // (SREG: s_i[yr] is still in this lane's register, ts_half)
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

// t_i and s_i of the players first and first + 2 in one go (ALG_HELP2 >= 6): t_i stays in registers -- lane (row = player slot, r) of either
// half holds t_i[r] after the exchange -- and the rows of t_i that A_{k+1}' t_i needs are shifts away inside the 16-lane row (the costate
// sweep's form of AT_vec), so t_i makes no trip through LDS; s_i goes to its LDS column for the next step AND stays in the lane that
// starts y_i = P_i rd + s_i from it (player_tail_half<.., SREG>): two LDS round trips and a fence less per step.  Same operations on the
// same numbers as t_half + s_half (the lambda hands AT_vec exactly the entries it would have read).
template <class C>
__device__ __forceinline__ double ts_half(DirLds<C>& L, const double* Rc, double dt, bool rec, int first, int lane) {
    constexpr int n = C::n, P = C::P, LDP = DirLds<C>::LDP;
    static_assert(n == 16 && P == 4 && C::MODEL == ALG_MODEL_UNICYCLE && !DirLds<C>::AUGS, "lane layout of the team-of-two tail (4-player unicycle)");
    using R = Rec<C>;
    const int h = lane >> 5, i = first + 2 * ((lane >> 4) & 1), r = lane & 15;
    double v = Rc[R::RX + i * n + r];
    if (rec) {
        const double* Pr = &L.bw.Pm[i * n * LDP + r * LDP];
        double a = h ? 0.0 : Pr[n];
#pragma unroll
        for (int c = 0; c < 8; c++) a += Pr[c + 8 * h] * L.bw.fv[c + 8 * h];
        a += xchg32(a, lane < 32);
        const double tr = a;
        const double s1 = row_shift<0x110 + P>(tr), s2 = row_shift<0x110 + 2 * P>(tr), s3 = row_shift<0x110 + 3 * P>(tr), u1 = row_shift<0x100 + P>(tr);
        const int blk = r / P, ii = r % P;
        const double vi = blk == 0 ? tr : (blk == 1 ? s1 : (blk == 2 ? s2 : s3)), vpi = blk == 0 ? u1 : (blk == 1 ? tr : (blk == 2 ? s1 : s2));
        v += AT_vec<C>(L.coefn, dt, [&](int rr) { return rr == r ? tr : (rr == ii ? vi : vpi); }, r);
    }
    if (lane < 2 * n) L.bw.Pm[i * n * LDP + r * LDP + n] = v;
    return v;
}

// Team of two (round 4): splitting the whole backward step over the two wavefronts costs more in barriers than it gains (measured on C3
// at 1024 games: 2.30 against 2.35 M/s), but the value recursion alone -- a quarter of a step, independent per player, LDS in / LDS out
// -- is worth two LDS-only barriers: wavefront 1 takes the odd players' MFMA chains of every step and does nothing else in the direction.
#ifndef ALG_HELP2
#define ALG_HELP2 4            // 0: off; 1: value recursion, Q-add, V rows, record fetch (bit-identical to the unsplit direction); 2: + s_i, y_i, g_c;
#endif                         // 3: + the helper solves the control system as well and forms the odd rows of [F | f]; 4: + it writes the gains; 5: t_i and s_i behind the MFMA chains (measured: loses); 6: t_i, s_i in registers (ts_half; measured +0.3 %, not shipped) (3 - 6: bit-identical to 2)
template <class C, bool IBR>
inline constexpr bool help2_v = ALG_HELP2 && C::NW == 2 && !IBR && !C::DENSE && C::WPE == 2 && C::MODEL == ALG_MODEL_UNICYCLE && C::P == 4;
// [P_i | s_i] <- A' ([P_i | s_i] [[F f],[0 1]]) for the players first, first + 2, ...: operands of all of them read first, their MFMA
// chains interleaved, results written back to the players' own LDS row blocks (nobody else touches those between the two barriers).
// SHADOW (ALG_HELP2 >= 5, n == 16): t_i = P_i f + s_i and s_i <- rx_i + A' t_i of the same players -- VALU / LDS work that needs the OLD
// P_i only -- are issued behind the MFMA chains and run while the matrix pipeline works (the wavefront would otherwise wait for
// the accumulators); same instructions on the same numbers as in their own phases.
template <class C, bool SHADOW = false>
__device__ __forceinline__ void value_recursion_half(DirLds<C>& L, int first, int lrow, int lq, double dt, const double* Rc = nullptr, int lane = 0) {
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
    if constexpr (SHADOW) {
        __builtin_amdgcn_sched_barrier(0);
        t_half<C>(L, first, lane);
        sweep_sync<C>();
        s_half<C>(L, Rc, dt, true, first, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
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
#ifdef ALG_TEAM_FULL_FENCE
    auto bsync = [&]() { if constexpr (TEAM) game_sync(); else sweep_sync<C>(); };
#else
    auto bsync = [&]() { if constexpr (TEAM) team_lds_barrier(); else sweep_sync<C>(); };
#endif
    const int lrow = lane & 15, lq = lane >> 4;          // MFMA lane coordinates
    const double dt = phase_f64(pr.dt);
    constexpr int RPL = (R::LEN_SWEEP + BT - 1) / BT;          // record doubles per thread
    constexpr int KPL = (NK + WAVE - 1) / WAVE;
    constexpr bool AUGS = DirLds<C>::AUGS;               // s_i rides through the first MFMA product (n < 16)
    constexpr int KB1 = DirLds<C>::KB1, VW = DirLds<C>::VW;
#ifndef ALG_GFUSE
#define ALG_GFUSE 1
#endif
    constexpr bool GFUSE = ALG_GFUSE && (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || C::MODEL == ALG_MODEL_UNICYCLE);
#ifndef ALG_SYSROW
#define ALG_SYSROW 1
#endif
    constexpr bool SYSROW = ALG_SYSROW && GFUSE;      // the V phase forms the system's rows (needs g from the y lanes)
    // Split value recursion (round 5; double integrator): with F = A_k + B K and f = rd + B kappa,
    //   [P_i | s_i] [[F f],[0 1]] = [P_i A_k | y_i] + (P_i B) [K | kappa],      y_i = P_i rd + s_i  (already formed for g_c),
    // and for the double integrator both P_i A_k (P + dt x its columns shifted by m) and P_i B (dt^2/2 x columns 0..m-1 + dt x columns
    // m..n-1) are two-term combinations of entries of P_i.  The accumulator tile starts at [P_i A_k | y_i] and the product has inner
    // dimension m instead of n + 1: ceil(m / 4) f64 MFMAs per player instead of (n + 4) / 4 (C2: 6 per step instead of 12 -- the matrix
    // pipe of a SIMD is shared by its four resident games and was busy 3 250 of the ~10 500 cycles of a backward step), and the
    // closed-loop phase ([F | f] = [A_k | rd] + B [K | kappa] -> LDS) disappears: the solved columns [K | kappa] go to LDS as they are
    // (the B operand of the product) and y_i replaces s_i in column n of the player's rows.
#ifndef ALG_SPLITF
#define ALG_SPLITF 1
#endif
    constexpr bool SPLITF = ALG_SPLITF && C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR && AUGS && !IBR && (C::NW == 1 || TEAM);
    constexpr int KBS = (m + 3) / 4;                    // k-blocks of the split product (inner dimension m)
    QaddMap<C, BT, (HELP2 ? 2 : 1)> qam; qam.init(tid, hw);
    if constexpr (HELP2) {
        if (hw == 1) {
            // the helper wavefront: value recursion and Q-add of the odd players, step by step between wavefront 0's two barriers
            double* const bwh = reinterpret_cast<double*>(&L.bw) + (int)(offsetof(typename DirLds<C>::Bwd, Pm) / 8);
            int curh = 0;
            for (int k = N - 2; k >= 0; k--, curh ^= 1) {
                team_lds_barrier();
#if ALG_HELP2 >= 6
                const double sregh = ts_half<C>(L, L.rec[curh], dt, k < N - 2, 1, tid);
                if (k < N - 2) value_recursion_half<C>(L, 1, lrow, lq, dt);
#elif ALG_HELP2 >= 5
                if (k < N - 2) value_recursion_half<C, true>(L, 1, lrow, lq, dt, L.rec[curh], tid);
                else s_half<C>(L, L.rec[curh], dt, false, 1, tid);
#else
                if (k < N - 2) {
#if ALG_HELP2 >= 2
                    t_half<C>(L, 1, tid);
#endif
                    value_recursion_half<C>(L, 1, lrow, lq, dt);
                }
#endif
                sweep_sync<C>();
                qam.apply(tid, L.rec[curh], L.qdf, bwh, reg, (k + 1 < N - 1) ? dt : 1.0, -1);
                sweep_sync<C>();
                v_sysrow<C>(L, L.rec[curh], k, dt, 2 * (tid >> 4) + 1, tid & 15);       // V rows of the odd players' controls
#if ALG_HELP2 >= 2
#if ALG_HELP2 >= 6
                player_tail_half<C, true, true>(L, L.rec[curh], dt, k, N, 1, tid, sregh);              // their y_i, g_c
#else
                player_tail_half<C, (ALG_HELP2 >= 5)>(L, L.rec[curh], dt, k, N, 1, tid);                 // their s_i, y_i, g_c
#endif
#endif
                team_lds_barrier();
                // while wavefront 0 runs the serial tail of step k: the record of step k - 1 from global memory into the other LDS slot
                // (wavefront 0 never waits for a load inside the sweep)
                double nx[RPL];
                if (k > 0) {
#pragma unroll
                    for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; nx[q] = e < R::LEN_SWEEP ? G.rec(pr)[(size_t)(k - 1) * R::LEN + e] : 0.0; }
                }
#if ALG_HELP2 >= 3
                // ... and its share of that tail, with the record's loads in flight: the same columns of [W | V A_k | g], the same pivoted
                // solve (every wavefront needs the solved columns for its rows), the odd rows of [F | f] = [A_k | rd] + B [K | kappa],
                // and (ALG_HELP2 >= 4) the gain stores -- wavefront 0 forms the even rows and issues no store at all in the sweep
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
#if ALG_HELP2 >= 4
                    asm volatile("" ::: "memory");
                    if (rhsl) {
                        double* __restrict__ Kg = G.kgain(pr) + (size_t)k * NK + (cidx - m) * m;
#pragma unroll
                        for (int c = 0; c < m; c++) Kg[c] = col[c];
                    }
#endif
                }
#else
                if (k > 0) {
#pragma unroll
                    for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; if (e < R::LEN_SWEEP) L.rec[curh ^ 1][e] = nx[q]; }
                }
#endif
            }
#if ALG_HELP2 >= 4
            game_sync();                 // the gains are this wavefront's stores: wavefront 0's forward sweep reads them behind a full barrier
#endif
            return ALG_STATUS_OK;
        }
    }
    HxMap<C> hxm;
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
    for (int e = tid; e < R::LEN_SWEEP; e += BT) L.rec[0][e] = G.rec(pr)[(size_t)(N - 2) * R::LEN + e];
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
            double kt[KBS];
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) kt[kb] = L.bw.Fx[(4 * kb + lq) * 16 + lrow];       // [K | kappa] of step k + 1, rows >= m zero
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
                for (int kb = 0; kb < KBS; kb++) pb[kb] = fma(dt, obase[i * n * LDP + 4 * kb + m], hdt2 * obase[i * n * LDP + 4 * kb]);
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
            bsync();
          }
        } else
        if (k < N - 2) {
            if constexpr (!AUGS && !(HELP2 && ALG_HELP2 >= 2)) {            // (team of two: per player, between the barriers below)
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
#if ALG_HELP2 >= 6
            const double sreg0 = ts_half<C>(L, Rc, dt, k < N - 2, 0, tid);
            if (k < N - 2) value_recursion_half<C>(L, 0, lrow, lq, dt);                // even players here, odd players on wavefront 1
#elif ALG_HELP2 >= 5
            if (k < N - 2) value_recursion_half<C, true>(L, 0, lrow, lq, dt, Rc, tid);     // even players here, odd players on wavefront 1
            else s_half<C>(L, Rc, dt, false, 0, tid);
#else
            if (k < N - 2) {
#if ALG_HELP2 >= 2
                t_half<C>(L, 0, tid);
#endif
                value_recursion_half<C>(L, 0, lrow, lq, dt);                // even players here, odd players on wavefront 1
            }
#endif
            sweep_sync<C>();
            qam.apply(tid, Rc, L.qdf, bwb + oPm, reg, w, -1);               // Q-add of the even players
            sweep_sync<C>();
            static_assert(!HELP2 || (SYSROW && m == 2 * P && m / 2 <= WAVE / 16), "V rows of one wavefront's players in one pass");
            v_sysrow<C>(L, Rc, k, dt, 2 * (tid >> 4) + 0, tid & 15);       // V rows of the even players' controls (c % P = player)
#if ALG_HELP2 >= 2
#if ALG_HELP2 >= 6
            player_tail_half<C, true, true>(L, Rc, dt, k, N, 0, tid, sreg0);                // their y_i, g_c
#else
            player_tail_half<C, (ALG_HELP2 >= 5)>(L, Rc, dt, k, N, 0, tid);                   // their s_i, y_i, g_c
#endif
#endif
#if ALG_HELP2 >= 3
            // coefficient entries of A_k' for both wavefronts' closed-loop rows (the table's last readers finished before the first barrier)
            if (tid < 4 * P) { const int kind = tid / P, i = tid % P; L.bw.T[(((kind & 1) ? 3 : 2) * P + i) * n + ((kind >> 1) ? P + i : i)] = coefk[tid]; }
#endif
            team_lds_barrier();                           // all players' P_i, V rows and g_c are back
        } else
        qam.apply(tid, Rc, L.qdf, bwb + oPm, reg, w, IBR ? ip : -1);
        constexpr bool TAIL2 = HELP2 && ALG_HELP2 >= 2;                      // team of two: s_i, y_i, g_c were formed per player above
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
        if (!HELP2 && k > 0) {                                       // (team of two: the helper wavefront fetches the record)
#pragma unroll
            for (int q = 0; q < RPL; q++) { const int e = tid + q * BT; pre[q] = e < R::LEN_SWEEP ? G.rec(pr)[(size_t)(k - 1) * R::LEN + e] : 0.0; }
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
#ifndef ALG_TEAM_YSPLIT
#define ALG_TEAM_YSPLIT 1
#endif
        constexpr int YOFF = (TEAM && ALG_TEAM_YSPLIT && C::NT >= 256) ? 128 : 0, TOFF = (TEAM && ALG_TEAM_YSPLIT && C::NT >= 256) ? 192 : 0;
        const int ty = tid - YOFF, tT = tid - TOFF;
        if (!TAIL2 && ty >= 0 && (ty >> 4) < P) {
            // rd sits one entry per lane in every 16-lane row and reaches the FMA chain through the DPP row broadcast: one LDS read of
            // rd per lane instead of n (same products, same order: bit-identical to `a += Pr[c] * rd[c]`)
            const int yp = ty >> 4, yr = (ty & 15) < n ? (ty & 15) : n - 1;
            const double* Pr = &L.bw.Pm[yp * n * LDP + yr * LDP];
            const double rdl = Rc[R::RD + yr];
            double a = Pr[n];
            rowdot_dpp_g<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(a, rdl, [&](int c) { return Pr[c]; });
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
            if (!(HELP2 && ALG_HELP2 >= 3) && tT >= 0 && tT < 4 * P) { const int kind = tT / P, i = tT % P; L.bw.T[(((kind & 1) ? 3 : 2) * P + i) * n + ((kind >> 1) ? P + i : i)] = coefk[tT]; }
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
#ifndef ALG_NO_GJ
        if constexpr (FLATGJ) sing |= gj_solve_cols<m>(col); else sing |= gj_solve_cols_dpp<m>(col);
#endif
        ALG_PROF(5)
        // ---- K = -Y -> HBM (column-major m x (n+1)) ; [F | f] = [A_k | rd] + B [K | kappa]
        if (rhsl) {
            const int cc = cidx - m;
#pragma unroll
            for (int c = 0; c < m; c++) col[c] = -col[c];
            if constexpr (SPLITF) {
                // split recursion: column cc of [K | kappa] is the B operand of the next step's product as it is
#pragma unroll
                for (int c = 0; c < m; c++) { if (!TEAM || (c % C::NW) == tw) L.bw.Fx[c * 16 + cc] = col[c]; }     // (team: every wavefront holds the columns)
            } else {
            // column cc of [A_k | rd]: contiguous in LDS (T row cc, or the record's rd); A_0 is never used (dx_1 = 0)
            const double* acol = (cc < n) ? &L.bw.T[cc * n] : Rc + R::RD;
            // all LDS reads first, then all writes: the compiler cannot prove that the Fx stores do not alias the T / record
            // loads and would otherwise serialise one LDS round trip per row
            // (team: wavefront tw forms the rows r = tw (mod NW); every wavefront holds the solved columns)
            // (team of two, ALG_HELP2 >= 3: the helper wavefront forms the odd rows)
            constexpr int RS = TEAM ? C::NW : ((HELP2 && ALG_HELP2 >= 3) ? 2 : 1);
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
        if (!(HELP2 && ALG_HELP2 >= 4) && tw == 0 && rhsl) {
            double* __restrict__ Kg = G.kgain(pr) + (size_t)k * NK + (cidx - m) * m;
#pragma unroll
            for (int c = 0; c < m; c++) Kg[c] = col[c];
        }
        bsync();
        ALG_PROF(6)
    }
    if constexpr (HELP2 && ALG_HELP2 >= 4) game_sync();                        // the helper wavefront's gain stores
    if (__builtin_amdgcn_readfirstlane(sing)) return ALG_STATUS_SINGULAR;      // wave-uniform (every lane factors the same matrix)
    if (TEAM && tw != 0) return ALG_STATUS_OK;         // the serial sweeps below belong to wavefront 0 (the caller holds a barrier)
#if defined(ALG_DIR_STOP) && ALG_DIR_STOP == 1
    return ALG_STATUS_OK;
#endif
    // ------------------------------------------------------------------ forward sweep: dx, du
    if constexpr (C::NW == 1) game_sync(); else dir_sync<C>();   // the gains are in global memory (the sweeps' own syncs order LDS only)
    G = G0.fresh();
    double* __restrict__ dz = G.z(2);
    if (lane < n) dz[lane] = 0.0;
    constexpr bool DIROW = ALG_DIROW && (C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR || ALG_DIROW >= 2) && P * 16 <= WAVE;
    constexpr int NPOS = C::POS ? C::PD * P : 1;
#ifndef ALG_FWDW
#define ALG_FWDW 1
#endif
    // FWDW: the forward sweep also forms the part of dlambda_k that does not depend on dlambda_{k+1} -- w_k = rx + Q^ dx_{k+1} -- right after
    // dx_{k+1} exists, in the bubbles of its own dependency chain (every 16-lane row runs the forward recursion redundantly, so row i has
    // dx for player i's products), and parks it in dlambda's slot; the costate sweep is left with dlambda_k = w_k + A' dlambda_{k+1}: one
    // load, a few shifts, no LDS, no fence.  Bit-identical: w_k is exactly the intermediate value the one-sweep form holds in a register.
    // (double integrator only: measured +1.1 % at C2; for the unicycle the recursion's coefficient loads cost more than the shorter chain
    // saves: C3 -0.8 %, C5 loop -0.6 %)
    constexpr bool FWDW = ALG_FWDW && DIROW && C::MODEL == ALG_MODEL_DOUBLE_INTEGRATOR;
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
        for (int q = 0; q < FPL; q++) L.rec[0][fso[q]] = G.rec(pr)[fso[q]];
    } else if (frok) L.rec[0][fro] = G.rec(pr)[fro];
    for (int e = lane; e < NK; e += WAVE) L.fw.kg[0][e] = G.kgain(pr)[e];
    // Global-memory schedule of a step.  gfx9 counts loads and stores in one vmcnt, so a wait for loaded data also waits for the
    // write acknowledgement of every store in flight; and a conditional load into a zero-initialised register makes the compiler
    // drain vmcnt at the top of the loop (write-after-write on the register).  Hence: unconditional loads from clamped
    // addresses, and everything at the tail of the step in the order (1) land the data of step k+1 (requested one step ago) in
    // LDS, (2) issue this step's result stores, (3) request the data of step k+2 -- the single wait of a step meets loads and
    // stores that have been in flight for a whole step.
    auto fwd_load = [&](int kk, double (&rf)[FPL], double (&rk)[KPL]) {
        const int kc = kk < N - 1 ? kk : N - 2;
#pragma unroll
        for (int q = 0; q < FPL; q++) rf[q] = G.rec(pr)[(size_t)kc * R::LEN + fso[q]];
#pragma unroll
        for (int q = 0; q < KPL; q++) { const int e = lane + q * WAVE; rk[q] = G.kgain(pr)[(size_t)kc * NK + (e < NK ? e : NK - 1)]; }
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
    cur = 0;
    double pl1 = 0.0;                               // sum |dx| + |du| of this lane's entries (Delta_step, primal_dual_traj.jl:130-147)
    int bad = 0;                                    // non-finite direction entries (checked where they are produced)
    // dx_k lives one entry per lane (lanes 0..n-1) and is broadcast with v_readlane; du and dx_{k+1} never pass through LDS:
    // one LDS round trip (gain rows, record slice) per step instead of three.
    double dxr = 0.0;
    for (int k0 = 0; k0 < N - 1; k0 += SD) {
#pragma unroll
      for (int u = 0; u < SD; u++) {
        const int k = k0 + u;
        if (k >= N - 1) break;
        const double* Rc = L.rec[cur]; const double* Kl = L.fw.kg[cur];
        const int fl = FWDW ? (lane & 15) : lane;     // FWDW: every 16-lane row runs the recursion (same LDS addresses, same instructions)
        const int cl = fl < m ? fl : 0;
        double acc = Kl[n * m + cl];
        rowdot_dpp_g<n, (rowdot_group_v<C> < n ? rowdot_group_v<C> : n)>(acc, dxr, [&](int q) { return Kl[q * m + cl]; });                  // dx_k sits in lanes 0..n-1 of the row, the control rows in its lanes 0..m-1 (same FMA order as the v_readlane form)
        const double duv = fl < m ? acc : 0.0;
        const double rdv = Rc[R::RD + (fl < n ? fl : 0)];
        double dxn = fwd_next<C>(Rc + R::COEF, dt, dxr, duv, fl) + rdv;
        dxn = fl < n ? dxn : 0.0;
        if (lane < m) { pl1 += fabs(duv); bad |= !isfinite(duv); }
        if (lane < n) { pl1 += fabs(dxn); bad |= !isfinite(dxn); }
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
                rowdot_dpp<NPOS>(t, dxn, hv);
                wk = rr_ < C::PD * P ? t : wk;
            }
            if (rok) dz[n + hl<C>(k, 0) + re_] = wk;
        }
        // (1) data of step k+1 (requested SD steps ago) -> LDS (clamped duplicates at the last steps are never read)
#pragma unroll
        for (int q = 0; q < FPL; q++) L.rec[cur ^ 1][fso[q]] = pref[(u + 1) % SD][q];
#pragma unroll
        for (int q = 0; q < KPL; q++) { const int e = lane + q * WAVE; L.fw.kg[cur ^ 1][e < NK ? e : NK - 1] = prek[(u + 1) % SD][q]; }
        asm volatile("" ::: "memory");
        // (2) results out
        if (lane < m) dz[n + hu<C>(k, 0) + uoff<C>(lane)] = duv;
        if (lane < n) dz[n + hx<C>(k) + lane] = dxn;
        // (3) request step k+1+SD into the slot that was just emptied
        fwd_load(k + 1 + SD, pref[(u + 1) % SD], prek[(u + 1) % SD]);
        sweep_sync<C>();
        cur ^= 1;
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
            wv = dz[n + hl<C>(kc, 0) + re_];
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
            if (rok) { dz[n + hl<C>(k, 0) + re_] = acc; bad |= !isfinite(acc); }
            cw_load(k - CD, wkr[u], cfr[u]);
          }
        }
    } else {
    constexpr int RPLC = (R::LEN_COSTATE + WAVE - 1) / WAVE;
    for (int e = lane; e < R::LEN_COSTATE; e += WAVE) L.rec[0][e] = G.rec(pr)[(size_t)(N - 2) * R::LEN + e];
    const int ci_ = lane < P * n ? lane / n : 0, cr_ = lane < P * n ? lane % n : 0;        // (player, row) of this lane
    const bool cpos = C::POS && cr_ < C::PD * P;
    double dxk = lane < n ? dz[n + hx<C>(N - 2) + lane] : 0.0;      // dx_{k+1}, fetched ahead like the records
    if constexpr (DIROW) dxk = dz[n + hx<C>(N - 2) + cdxo];
    auto cs_load = [&](int kk, double& rdx, double (&rr)[RPLC]) {
        const int kc = kk > 0 ? kk : 0;
#pragma unroll
        for (int q = 0; q < RPLC; q++) { const int e = lane + q * WAVE; rr[q] = G.rec(pr)[(size_t)kc * R::LEN + (e < R::LEN_COSTATE ? e : R::LEN_COSTATE - 1)]; }
        rdx = dz[n + hx<C>(kc) + cdxo];
    };
    // register ring like the forward sweep's: [record slice | dx] of steps k - 1 .. k - SD are in flight while step k computes
    double pre[SD][RPLC], pdx[SD];
#pragma unroll
    for (int u = 0; u < SD; u++) cs_load(N - 3 - u, pdx[(1 + u) % SD], pre[(1 + u) % SD]);
    sweep_sync<C>();
    cur = 0;
    for (int k0 = N - 2; k0 >= 0; k0 -= SD) {
#pragma unroll
      for (int u = 0; u < SD; u++) {
        const int k = k0 - u;
        if (k < 0) break;
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
            if (rok) { dz[n + hl<C>(k, 0) + re_] = acc; bad |= !isfinite(acc); }
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
        if (lane < P * n) { L.fw.dl[lane] = acc; dz[n + hl<C>(k, 0) + lane] = acc; bad |= !isfinite(acc); }
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
        cur ^= 1;
      }
    }
    }
    ALG_PROF(8)
    ALG_PROF_FLUSH
    // non-finite direction -> singular (the reference would throw / propagate NaN)
    if (primal_l1) *primal_l1 = wave_sum(pl1);
    return __builtin_amdgcn_readfirstlane(wave_or(bad)) ? ALG_STATUS_SINGULAR : ALG_STATUS_OK;     // scalar: the solver's control flow stays on the SALU
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
#ifndef ALG_GATE_UNROLL
#define ALG_GATE_UNROLL 0      // > 1: trips of the gate's flat loop whose loads are in flight together (a trip is one exposed memory round trip)
#endif
#if ALG_GATE_UNROLL > 1
#pragma unroll ALG_GATE_UNROLL
#endif
    for (int e = tid; e < (N - 1) * m; e += C::NT) {
        const int k = e / m, c = e % m, i = c % P;
        double* Rk = recs + (size_t)k * R::LEN;
        const double* dl = dz + n + hl<C>(k, i);
        const double du = dz[n + hu<C>(k, 0) + uoff<C>(c)];
        const double rh = Rk[R::RHAT + c], ru = Rk[R::RU + c];
        const double bl = BT_vec<C>(Rk + R::COEF, dt, [&](int rr) { return dl[rr]; }, c);
        // |B[:,c]|' |dlambda| from below: the coefficients keep their signs (exact for the double integrator, whose B is non-negative)
        const double bla = BT_vec<C>(Rk + R::COEF, dt, [&](int rr) { return fabs(dl[rr]); }, c);
        double rho = fma(rh, du, ru) + bl;
        if (IBR && i != ip) rho = 0.0;                                  // unit rows of the other players (du_c = 0)
        const double sc = fabs(rh * du) + fabs(ru) + fabs(bla);         // row scale |J_c| |d| + |ru_c|
        const double ar = fabs(rho);
        if (ar * ws > wr * sc) { wr = ar; ws = sc; }
        rho_m = fmax(rho_m, ar); s_m = fmax(s_m, sc);
        if constexpr (WRITE) Rk[R::RU + c] = rho;
    }
    if constexpr (WRITE) {
        for (int e = tid; e < (N - 1) * P * n; e += C::NT) recs[(size_t)(e / (P * n)) * R::LEN + R::RX + e % (P * n)] = 0.0;
        for (int e = tid; e < (N - 1) * n; e += C::NT) recs[(size_t)(e / n) * R::LEN + R::RD + e % n] = 0.0;
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
        const double v = dz[n + e] + ez[n + e];
        dz[n + e] = v;
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
    for (int e = phase_lane(); e < pr.con_len; e += C::NT) mx = fmax(mx, mu[e]);
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
        if (st != ALG_STATUS_OK) { if (pass == 0) { if (primal_l1) *primal_l1 = pl1s; return st; } st = ALG_STATUS_OK; break; }
        game_sync();                                   // the direction is in global memory
        const DirGate gt = dir_urow_residual<C, IBR, false>(pr, Gd, ip);
        // backward error of the whole direction: the row-wise omega of the first solve; after a correction, the residual of the correction
        // system IS the new residual of the whole system, so omega contracts like max |rho| did
        double omega;
        if (pass == 0) {
            omega = gt.omega;
            if (phase_lane() == 0) { tc[TC_RHO] = gt.rho; tc[TC_OMEGA] = gt.omega; tc[TC_SMAX] = gt.smax; tc[TC_OMCUR] = gt.omega; tc[TC_RHOCUR] = gt.rho; }
        } else {
            int bad = 0; double pl1, dn;
            dir_add_correction<C>(pr, G0, pl1, dn, bad);
            const double rho_prev = tc[TC_RHOCUR];
            omega = tc[TC_OMCUR] * (gt.rho / fmax(rho_prev, 1e-300));
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
        const double tol = phase_f64(pr.refine_tol) * (C::DENSE ? 0x1p-6 : 1.0);
        bool done = !(uni(omega) > tol) || pass >= rmax;
        if (!done && !C::DENSE && !(uni(omega) > 256.0 * tol)) {
            const double mumax = con_mu_max<C>(pr, G0);
            const double relax = fmin(fmax(phase_f64(pr.refine_mu) / fmax(mumax, 1e-300), 1.0), 256.0);
            done = !(uni(omega) > relax * tol);
        }
        if (done) {
            // the common case leaves from the first pass with sum |d_primal| still in a register: no round trip through the control
            // slots (the slot stores above are not waited for; every wavefront of a team has passed the gate's barriers)
#ifndef ALG_GATE_NO_EARLY     // A/B builds (tests/probes/build_variant.sh)
            if (pass == 0) { if (primal_l1) *primal_l1 = pl1s; return st; }
#endif
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
        rc.outer = outer; rc.ls_j = 0; rc.alpha = 0.0; rc.res = ro.l1 / (double)pr.S; rc.delta = delta;
        rc.dyn_vio = ro.dyn; rc.con_vio = ro.con; rc.sta_vio = ro.sta; rc.opt_vio = ro.opt;
        rc.t_elap = G.tc(pr)[TC_TELAP];               // the previous inner iteration's duration (iter_clock_stop)
        const int idx = G.st(pr)->records;
        if (idx < pr.hist_max) G.hist(pr)[idx] = rc;
        G.st(pr)->records = idx + 1;
        G.st(pr)->last = rc;
        if (out) *out = rc;
    }
    RecScalars r; r.res = uni(ro.l1 / (double)phase_int(pr.S)); r.opt = uni(ro.opt); r.nonfinite = __builtin_amdgcn_readfirstlane(ro.nonfinite);
    return r;
}
template <class C>
__device__ __forceinline__ RecScalars make_record(CPR pr, const Game& G, Lds<C>& L, double delta, int outer, double jreg, alg_record* out) {
    ResOut ro;
    LSP_T0 LSP_COUNT(29)
    if constexpr (AsmLds<C>::FUSED) assemble_fused<C, 1, false>(pr, G, L.a, 0.0, false, 0.0, jreg, ro);
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
        if (rt <= (1.0 - alpha * o.beta) * res_norm0) break;
        alpha *= o.alpha_decrease; j += 1;
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
    if (cache_valid && *cache_valid) { ResOut cro; tcache_load(pr, G, cro); rs = push_stats(pr, G, cro, Delta, k, info ? &info->rec : nullptr); }
    else rs = make_record<C>(pr, G, L, Delta, k, reg, info ? &info->rec : nullptr);
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
            double s2 = d0 * d0 + d1 * d1;
            if constexpr (C::PD == 3) { const double d2 = pr.ca_dim == 3 ? x[2 * P + i] - x[2 * P + j] : 0.0; s2 += d2 * d2; }
            const double c = (double)((pr.ca_mask[i] >> j) & 1u) * (R * R - s2);
            G.vals(pr)[e] = c;
            const double lb = G.lam(pr)[e] + o.alphax_dual[i] * G.mu(pr)[e] * c;
            G.lam(pr)[e] = fmin(fmax(lb, 0.0), o.lambda_max);
        }
    }
    if (pr.has_ctl) {
        for (int e = tid; e < pr.ctl_len; e += C::NT) {
            const int k = e / (2 * m), row = e % (2 * m), c = row % m;
            const double u = z[n + hu<C>(k, 0) + uoff<C>(c)];
            const double cv = row < m ? u - pr.umax[c] : pr.umin[c] - u;
            const int ci = pr.col_len + e;
            G.vals(pr)[ci] = cv;
            if (isfinite(cv)) { const double lb = G.lam(pr)[ci] + o.alpha_dual * G.mu(pr)[ci] * cv; G.lam(pr)[ci] = fmin(fmax(lb, 0.0), o.lambda_max); }
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
template <class C>
__device__ __forceinline__ void rollout(CPR pr, double* z) {
    constexpr int n = C::n, m = C::m, P = C::P;
    const int lane = game_tid();
    if constexpr (C::QUAD) {
        if (lane < P) {
            double xi[12], ui[4], xo[12];
#pragma unroll
            for (int j = 0; j < 12; j++) xi[j] = z[lane + j * P];
            for (int k = 0; k < pr.N - 1; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) ui[j] = z[n + hu<C>(k, lane) + j];
                quad_rk3(xi, ui, pr.qmass, pr.dt, xo);
#pragma unroll
                for (int j = 0; j < 12; j++) { xi[j] = xo[j]; z[n + hx<C>(k) + lane + j * P] = xo[j]; }
            }
        }
    } else if (lane < P) {
        double x[n], u[m];     // only this player's entries are used
        for (int j = 0; j < C::ni; j++) x[lane + j * P] = z[lane + j * P];
        for (int k = 0; k < pr.N - 1; k++) {
            for (int j = 0; j < C::mi; j++) u[lane + j * P] = z[n + hu<C>(k, lane) + j];
            double xn[C::ni];
            model_player_rk3<C>(pr, lane, x, u, pr.dt, xn);
            for (int j = 0; j < C::ni; j++) { x[lane + j * P] = xn[j]; z[n + hx<C>(k) + lane + j * P] = xn[j]; }
        }
    }
}

// init_traj! (primal_dual_traj.jl:29-44) with the counter RNG (same element counters as the oracle)
template <class C>
__device__ void init_traj(CPR pr, const Game& G, double* z, uint64_t game_id, bool use_shift, int shift = -1) {
    constexpr int n = C::n, m = C::m, P = C::P;
    const int N = pr.N, lane = game_tid(); const auto& o = pr.opt;
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

// After an odd number of buffer exchanges pdtraj lives in the trial buffer: move it home (and leave the trial buffer with
// the previous iterate, as update_traj! would have)
template <class C>
__device__ __forceinline__ void settle_traj(CPR pr, Game& G) {
    if (G.zo[0] != 0) {
        game_sync();
        const Game H = G.fresh();
        double* a = H.z(0); double* z_home = H.base;
        for (int e = phase_lane(); e < pr.traj_len; e += C::NT) { const double v = a[e]; a[e] = z_home[e]; z_home[e] = v; }
        G.zo[1] = G.zo[0]; G.zo[0] = 0;
        game_sync();
    }
}

// newton_solve! (solver_methods.jl:5-65)
template <class C>
__device__ __forceinline__ void newton_solve(CPR pr, Game& G, Lds<C>& L, int init, uint64_t game_id, int shift = -1, int dual_reset = -1) {
    const auto& o = pr.opt; const int lane = phase_lane();
    if (lane == 0) { alg_game_stats z{}; *G.fresh().st(phase_params(pr)) = z; G.fresh().tc(phase_params(pr))[TC_TELAP] = 0.0; } // reset!(prob.stats); t_elap = 0
#ifdef ALG_PHASE_PROF
    if (lane < 16) G.res(pr)[lane] = 0.0;                                   // scratch instrumentation: per-phase cycle sums (16..: pass-level sums over the handle's lifetime)
    if (game_tid() < 32) lsp_slots()[game_tid()] = 0u;
#endif
#ifndef ALG_TEST_NOINIT
    if (init) init_traj<C>(pr, G, G.z(0), game_id, true, shift);           // :13
    else { if (lane < C::n) G.z(0)[lane] = G.x0(pr)[lane]; }
    if (lane < C::n) { G.z(1)[lane] = G.x0(pr)[lane]; G.z(2)[lane] = 0.0; }    // :14-15 (only x_1 of the trial matters)
    game_sync();
    rollout<C>(pr, G.z(0));                                                // :17
#endif
    if (dual_reset >= 0 ? dual_reset : o.dual_reset) reset_con<C::NT>(pr, G);     // :25
    game_sync();
    int out = 0, status = ALG_STATUS_OK, fresh = 0; double Delta = 0.0;
    for (int k = 1; k <= o.outer_iter; k++) {                              // :30
        out = k;
        int LS_count = 0;
        int cache_valid = 0;
        for (int l = 1; l <= o.inner_iter; l++) {                          // :38
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
        dual_penalty_update<C>(pr, G);                                     // :57-61
        game_sync();
    }
    game_sync();
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
