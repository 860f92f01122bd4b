// algames_kernels.hpp -- the __global__ entry points (one workgroup = one wavefront = one game) and the list of compiled
// (model, p, d, ext) instantiations.  The base instantiations live in algames_hip.hip; the EXT ones (bicycle model, state
// bounds, walls, circles) are explicitly instantiated in algames_ext_*.hip so that the translation units build in parallel.
#pragma once
#include "algames_device.hpp"

using namespace alg;

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(C::NT, C::WPE) k_newton_solve(Params pr_arg, int init, uint64_t game_id0) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    newton_solve<C>(pr, G, L, init, game_id0 + (uint64_t)g);
}

// Straggler hand-off (alg_set_handoff; newton_solve<C, HO>, algames_solver.hpp): the budgeted one-wavefront solve, whose games park after
// `budget` inner iterations, and the team kernel that resumes the parked games (block b takes the b-th entry of the handle's queue; the
// launch covers the whole batch, blocks past the queue's count leave at once).
template <class C>
__global__ void __launch_bounds__(C::NT, C::WPE) k_newton_solve_ho(Params pr_arg, int init, uint64_t game_id0, int budget) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    newton_solve<C, 1>(pr, G, L, init, game_id0 + (uint64_t)g, -1, -1, budget);
}
template <class C>
__global__ void __launch_bounds__(C::NT, C::WPE) k_newton_resume(Params pr_arg) {        // (C = Cfg<..., NW, 0>: no line-search staging in LDS)
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int* q = as_global(pr.ho_queue);
    if ((int)blockIdx.x >= __builtin_amdgcn_readfirstlane(q[0])) return;
    const int g = __builtin_amdgcn_readfirstlane(q[1 + blockIdx.x]);
    Game G = game_view(pr, g);
    newton_solve<C, 2>(pr, G, L, 0, 0);
}

template <class C>
// (no occupancy bound: the host-driven stepping entry point is not throughput code, and inner_iteration's live ranges at the four-waves-per-SIMD
// budget are sized for the fused kernel, where the surrounding loops are in the same function)
__global__ void __launch_bounds__(WAVE) k_newton_step(Params pr_arg, int k, int l, const double* delta_in, alg_step_info* out) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    int ls = 0; double dl = delta_in ? delta_in[g] : 0.0;          // the caller's Δ goes into record! (solver_methods.jl:75)
    inner_iteration<C>(pr, G, L, ls, dl, k, l, out + g, nullptr);
    settle_traj<C>(pr, G);
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_residual(Params pr_arg, int which, double reg, double* rn_out) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    ResOut ro;
    // the proximal term is taken w.r.t. pdtraj (regularize_residual!, global_quantities.jl:67-86)
    assemble_pass<C, 2>(pr, G, L.a, which, reg != 0.0 ? 0 : -1, reg, 0.0, ro);
    __syncthreads();
    if (rn_out && threadIdx.x == 0) rn_out[g] = ro.l1 / (double)pr.S;
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_jacobian(Params pr_arg, double reg, double* J, int g0) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x + g0;        // games g0 .. g0 + gridDim.x - 1; J holds gridDim.x blocks
    Game G = game_view(pr, g);
    ResOut ro;
    assemble_pass<C, 1>(pr, G, L.a, 0, -1, 0.0, reg, ro);
    __syncthreads();
    jacobian_dense<C>(pr, G, reg, J + (size_t)blockIdx.x * pr.S * pr.S);
}

template <class C>
__global__ void __launch_bounds__(WAVE, C::WPE) k_direction(Params pr_arg, double reg, int* status) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    ResOut ro;
    assemble_pass<C, 1>(pr, G, L.a, 0, -1, 0.0, reg, ro);
    __syncthreads();
    const int st = refined_direction<C, false, false>(pr, G, L, reg, -1, nullptr);
    if (status && threadIdx.x == 0) status[g] = st;
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_line_search(Params pr_arg, double reg, const double* rn, double* alpha, int* j) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    double a; int jj;
    line_search<C>(pr, G, L, reg, rn[g], -1.0, &a, &jj);
    if (threadIdx.x == 0) { alpha[g] = a; j[g] = jj; }
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_update(Params pr_arg, int tgt, int src, const double* alpha) {
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    update_traj<C>(pr, G, tgt, src, alpha[g]);
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_record(Params pr_arg, alg_record* out) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    make_record<C>(pr, G, L, 0.0, 0, 0.0, out + g);
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_dual_update(Params pr_arg) {
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    dual_penalty_update<C>(pr, G);
}

template <class C>
__global__ void __launch_bounds__(WAVE) k_init(Params pr_arg, uint64_t game_id0, int use_shift, int do_init, int which) {
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    if (do_init) {
        init_traj<C>(pr, G, G.z(0), game_id0 + (uint64_t)g, use_shift != 0);
        if (threadIdx.x < C::n) G.z(1)[threadIdx.x] = G.x0(pr)[threadIdx.x];
        __syncthreads();
        rollout<C>(pr, G.z(0));
    } else {
        rollout<C>(pr, G.z(which));
    }
}

// mode 0: ibr_newton_solve!(prob, player) on the stored trajectory ; mode 1: ibr_newton_solve!(prob; ibr_opts)
template <class C>
__global__ void __launch_bounds__(WAVE, (C::WPE < 2 ? C::WPE : 2)) k_ibr(Params pr_arg, int mode, int player, int init, uint64_t game_id0,
                                                      int ibr_iter, IbrOrder order, double delta_min) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    ibr_newton_solve<C>(pr, G, L, mode == 0, player, init, game_id0 + (uint64_t)g, ibr_iter, order, delta_min);
}

// builder-defined MPC advance: x0 <- RK2(x_1, u_1) per game (lanes < P own a player), totals += solve
template <class C>
__device__ __forceinline__ void mpc_advance(CPR pr, const Game& G) {
    const int lane = phase_lane();
    if constexpr (C::QUAD) {
        if (lane < C::P) {
            double xi[12], ui[4], xo[12];
#pragma unroll
            for (int j = 0; j < 12; j++) xi[j] = G.z(0)[lane + j * C::P];
#pragma unroll
            for (int j = 0; j < 4; j++) ui[j] = G.z(0)[C::n + hu<C>(0, lane) + j];
            quad_rk2(xi, ui, pr.qmass, pr.dt, xo);
#pragma unroll
            for (int j = 0; j < 12; j++) { const int a = lane + j * C::P; G.x0w(pr)[a] = xo[j]; G.z(0)[a] = xo[j]; G.z(1)[a] = xo[j]; }
        }
    } else if (lane < C::P) {
        double x[C::n], u[C::m], xo[C::ni], co[4];
        for (int j = 0; j < C::ni; j++) x[lane + j * C::P] = G.z(0)[lane + j * C::P];
        for (int j = 0; j < C::mi; j++) u[lane + j * C::P] = G.z(0)[C::n + hu<C>(0, lane) + j];
        model_player<C>(pr, lane, x, u, pr.dt, xo, co);
        for (int j = 0; j < C::ni; j++) {
            const int a = lane + j * C::P;
            G.x0w(pr)[a] = xo[j]; G.z(0)[a] = xo[j]; G.z(1)[a] = xo[j];
        }
    }
    if (lane == 0) { G.mpc(pr)[0] += G.st(pr)->newton_iters; G.mpc(pr)[1] += G.st(pr)->converged; }
}
template <class C>
__global__ void __launch_bounds__(WAVE) k_mpc_advance(Params pr_arg) {
    CPR pr = kernel_params();
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    mpc_advance<C>(pr, G);
}

// The whole receding-horizon loop of one game in one wave (BASELINE config 5): `steps` x (newton_solve! from the shifted
// warm start, advance x0 by one RK2 step).  Games never wait for each other between MPC steps, so a game that needs many
// Newton iterations at one step only delays itself.  Step 0 uses the handle's shift / dual_reset, later steps shift = 1
// and dual_reset = false (the reference's warm-start hooks, options.jl / primal_dual_traj.jl:29-44).
// Register budget of the 256-VGPR class for every configuration: the loop carries more live state (step counter, state log,
// totals) than a single solve; at 128 VGPRs the DoubleIntegrator instantiations spilled SGPRs so heavily that the 2-player
// one faulted on a null base pointer (tests/test_gpu_parity_ext.py::test_no_kernel_writes_outside_its_buffers runs this
// kernel for every instantiation).
// (the loop's own arguments are re-read from the kernel-argument segment where they are used, like `Params`: as by-value arguments they
// were live -- in SGPRs, i.e. spilled -- across every phase of every solve)
// The loop kernels of the teams of four (BASELINE C5 runs k_mpc_loop<Cfg<UNICYCLE, 3, 2, 0, 4>>) also run the solve's set-up on laundered views
// (newton_solve<C, 0, LOOP = true>): 20 -> 13 SGPR spills in that kernel.  Only there: the solver-level functions are left to the inliner, and a loop
// kernel that instantiates another newton_solve than the configuration's solve kernel changes the inliner's decisions for BOTH kernels of the
// configuration -- with the switch on everywhere the C3 solve kernel (team of two) went from 8 to 10 spills (round 6, measured per combination).
#ifndef ALG_LOOP_LAUNDER
#define ALG_LOOP_LAUNDER 1
#endif
template <class C> inline constexpr bool mpc_loop_launder_v = ALG_LOOP_LAUNDER != 0 && C::NW == 4;
struct MpcLoopArgs { Params pr; int steps; uint64_t game_id0; double* states; };
// (the 4-player bicycle with the extended constraint set -- 8 controls x 17 right-hand sides in registers -- sits at the 256-register
// ceiling with the loop's own state on top: it takes the one-wavefront-per-SIMD budget, where the allocator parks the overflow in the
// accumulation registers instead of scratch; the loop runs small batches, never two wavefronts per SIMD)
template <class C> inline constexpr int mpc_loop_wpe = (C::WPE < 2 || (C::MODEL == ALG_MODEL_BICYCLE && C::P == 4)) ? 1 : 2;
template <class C>
__global__ void __launch_bounds__(C::NT, mpc_loop_wpe<C>) k_mpc_loop(Params pr_arg, int steps_arg, uint64_t game_id0_arg, double* states_arg) {
    __shared__ Lds<C> L;
    CPR pr = kernel_params();
#if defined(__HIP_DEVICE_COMPILE__)
    const ALG_AS4 MpcLoopArgs& ka = *(const ALG_AS4 MpcLoopArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const MpcLoopArgs& ka = *(const MpcLoopArgs*)nullptr;      // host pass: never executed
#endif
    const int g = blockIdx.x;
    Game G = game_view(pr, g);
    // (the loop's arguments are read through a laundered copy of the segment pointer wherever they are used -- `kq()` -- so that no load is shared
    // between the prologue, the loop test and the loop body: a shared load is a value live, i.e. spilled, across every phase of every solve)
    auto kq = [&]() -> const ALG_AS4 MpcLoopArgs& { return *(const ALG_AS4 MpcLoopArgs*)uniform_u64((unsigned long long)&ka); };
    {
        double* const st0 = kq().states; const int l0 = phase_lane();
        if (st0 && l0 < C::n) { const Game H = G.fresh(); CPR pr0 = phase_params(pr); st0[(size_t)phase_int(g) * C::n + l0] = H.x0(pr0)[l0]; }
    }
    if (kq().steps < 1) return;
    for (int t = 0; ; t++) {
        // (everything the step needs besides `t` is re-derived from opaque roots inside the loop -- the game's index, the lane's predicates, the
        // arguments: as invariants of this loop they were live, i.e. spilled, across every phase of every solve)
        const int gq = phase_int(g);
        newton_solve<C, 0, mpc_loop_launder_v<C>>(pr, G, L, 1, kq().game_id0 + (uint64_t)t * 1000003ull + (uint64_t)gq, t == 0 ? -1 : 1, t == 0 ? -1 : 0);
        __syncthreads();
        mpc_advance<C>(phase_params(pr), G.fresh());
        __syncthreads();
        double* const states = kq().states;
        const int ln = phase_lane();
        if (states && ln < C::n) { CPR prs = phase_params(pr); states[((size_t)(t + 1) * prs.B + phase_int(g)) * C::n + ln] = G.fresh().z(0)[ln]; }
        __syncthreads();
        if (t + 1 >= kq().steps) break;
    }
}

// ------------------------------------------------------------------------------------------------
// Instantiation lists: X(model, p, d, ext)
// ------------------------------------------------------------------------------------------------
#define ALG_CFGS_BASE(X)                                    \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 3, 0)                  \
    X(ALG_MODEL_UNICYCLE, 1, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, 2, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, 3, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, 4, 2, 0)
#define ALG_CFGS_EXT_DI(X)                                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 2, 1)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 2, 1)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 1)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 2, 1)
#define ALG_CFGS_EXT_UNI(X)                                 \
    X(ALG_MODEL_UNICYCLE, 1, 2, 1)                           \
    X(ALG_MODEL_UNICYCLE, 2, 2, 1)                           \
    X(ALG_MODEL_UNICYCLE, 3, 2, 1)                           \
    X(ALG_MODEL_UNICYCLE, 4, 2, 1)
#define ALG_CFGS_EXT_BIC(X)                                 \
    X(ALG_MODEL_BICYCLE, 1, 2, 1)                            \
    X(ALG_MODEL_BICYCLE, 2, 2, 1)                            \
    X(ALG_MODEL_BICYCLE, 3, 2, 1)                            \
    X(ALG_MODEL_BICYCLE, 4, 2, 1)
#define ALG_CFGS_EXT_DI3(X)                                 \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 3, 1)
// QuadrotorGame (quadrotor.jl:22: p <= 4), dense Newton direction (algames_quad.hip)
#define ALG_CFGS_QUAD(X)                                    \
    X(ALG_MODEL_QUADROTOR, 1, 3, 0)                          \
    X(ALG_MODEL_QUADROTOR, 2, 3, 0)                          \
    X(ALG_MODEL_QUADROTOR, 3, 3, 0)                          \
    X(ALG_MODEL_QUADROTOR, 4, 3, 0)
#define ALG_CFGS_QUAD_EXT(X)                                \
    X(ALG_MODEL_QUADROTOR, 1, 3, 1)                          \
    X(ALG_MODEL_QUADROTOR, 2, 3, 1)                          \
    X(ALG_MODEL_QUADROTOR, 3, 3, 1)                          \
    X(ALG_MODEL_QUADROTOR, 4, 3, 1)
// DoubleIntegrator in three dimensions with p = 1, 3, 4 (n = 6, 18, 24: dense Newton direction; algames_di3.hip)
#define ALG_CFGS_DI3D(X)                                    \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 3, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 3, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 3, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 3, 1)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 3, 1)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 3, 1)
// DoubleIntegratorGame(p, d = 1) (double_integrator.jl:13-25 takes any d; round 6): n = 2 p, m = p.  px[i] = (i, i + p) is (position, velocity) of
// player i here -- the reference's index sets do not depend on d -- and the collision terms act on that pair as they do in the reference.
// p = 2, 4 take the tile path, p = 1, 3 (n = 2, 6) the dense direction; base constraint set (algames_di1.hip)
#define ALG_CFGS_DI1(X)                                     \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 1, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 1, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 1, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 1, 0)
// Five and six players (n = 20 / 24: dense Newton direction; algames_p5.hip, algames_p6.hip).  The reference itself caps p at 10 (options.jl:68)
#define ALG_CFGS_P5(X)                                      \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 5, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 5, 2, 1)                  \
    X(ALG_MODEL_UNICYCLE, 5, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, 5, 2, 1)                           \
    X(ALG_MODEL_BICYCLE, 5, 2, 1)
#define ALG_CFGS_P6(X)                                      \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 6, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 6, 2, 1)                  \
    X(ALG_MODEL_UNICYCLE, 6, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, 6, 2, 1)                           \
    X(ALG_MODEL_BICYCLE, 6, 2, 1)
#define ALG_CFGS_P56(X) ALG_CFGS_P5(X) ALG_CFGS_P6(X)
// Seven to nine players (round 6; n = 28 ... 36, m = 14 ... 18: the dense direction with the value matrices of all players LDS-resident --
// 65 / 93 / 128 KB of the CU's 160, one game per CU; algames_p7.hip ... algames_p9.hip).  Ten players (the reference's cap, options.jl:68;
// algames_p10.hip): 131 KB of value matrices, the step's workspace in the TIGHT layout of DirLds<C, true> -- 161 KB.
#define ALG_CFGS_PN(X, N)                                   \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, N, 2, 0)                  \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, N, 2, 1)                  \
    X(ALG_MODEL_UNICYCLE, N, 2, 0)                           \
    X(ALG_MODEL_UNICYCLE, N, 2, 1)                           \
    X(ALG_MODEL_BICYCLE, N, 2, 1)
#define ALG_CFGS_P7(X) ALG_CFGS_PN(X, 7)
#define ALG_CFGS_P8(X) ALG_CFGS_PN(X, 8)
#define ALG_CFGS_P9(X) ALG_CFGS_PN(X, 9)
#define ALG_CFGS_P10(X) ALG_CFGS_PN(X, 10)
#define ALG_CFGS_P789(X) ALG_CFGS_P7(X) ALG_CFGS_P8(X) ALG_CFGS_P9(X) ALG_CFGS_P10(X)
#define ALG_CFGS_DENSE(X) ALG_CFGS_QUAD(X) ALG_CFGS_QUAD_EXT(X) ALG_CFGS_DI3D(X) ALG_CFGS_P56(X) ALG_CFGS_P789(X)
#define ALG_CFGS_EXT(X) ALG_CFGS_EXT_DI(X) ALG_CFGS_EXT_UNI(X) ALG_CFGS_EXT_BIC(X) ALG_CFGS_EXT_DI3(X)

// every kernel of one instantiation; PREFIX is `template` (definition) or `extern template` (declaration)
#define ALG_INSTANTIATE_KERNELS(PREFIX, M, P, D, E)                                                                        \
    PREFIX __global__ void k_newton_solve<Cfg<M, P, D, E>>(Params, int, uint64_t);                                \
    PREFIX __global__ void k_newton_step<Cfg<M, P, D, E>>(Params, int, int, const double*, alg_step_info*);                      \
    PREFIX __global__ void k_residual<Cfg<M, P, D, E>>(Params, int, double, double*);                    \
    PREFIX __global__ void k_jacobian<Cfg<M, P, D, E>>(Params, double, double*, int);                             \
    PREFIX __global__ void k_direction<Cfg<M, P, D, E>>(Params, double, int*);                                    \
    PREFIX __global__ void k_line_search<Cfg<M, P, D, E>>(Params, double, const double*, double*, int*);          \
    PREFIX __global__ void k_update<Cfg<M, P, D, E>>(Params, int, int, const double*);                            \
    PREFIX __global__ void k_record<Cfg<M, P, D, E>>(Params, alg_record*);                                        \
    PREFIX __global__ void k_dual_update<Cfg<M, P, D, E>>(Params);                                                \
    PREFIX __global__ void k_init<Cfg<M, P, D, E>>(Params, uint64_t, int, int, int);                              \
    PREFIX __global__ void k_ibr<Cfg<M, P, D, E>>(Params, int, int, int, uint64_t, int, IbrOrder, double);        \
    PREFIX __global__ void k_mpc_advance<Cfg<M, P, D, E>>(Params);                                                \
    PREFIX __global__ void k_mpc_loop<Cfg<M, P, D, E>>(Params, int, uint64_t, double*);
// Team kernels (Cfg::NW wavefronts per game, small batches): X(model, p, d, ext, nw).  Only the fused solver and the fused
// receding-horizon loop exist in this shape; the step-wise entry points always use one wavefront per game.
#define ALG_CFGS_MW(X)                                      \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0, 4)               \
    X(ALG_MODEL_UNICYCLE, 3, 2, 0, 4)                        \
    X(ALG_MODEL_UNICYCLE, 4, 2, 0, 2)                        \
    X(ALG_MODEL_UNICYCLE, 4, 2, 0, 4)
// Team kernels of the dense-direction configurations (algames_mw_dense.hip): where the LDS footprint leaves room for few
// workgroups per CU at any batch size the automatic choice is the team of four regardless of the batch (team_width, algames_hip.hip)
#define ALG_CFGS_MW_DENSE(X)                                \
    X(ALG_MODEL_QUADROTOR, 2, 3, 0, 4)                       \
    X(ALG_MODEL_QUADROTOR, 3, 3, 0, 4)                       \
    X(ALG_MODEL_QUADROTOR, 4, 3, 0, 4)                       \
    X(ALG_MODEL_QUADROTOR, 2, 3, 1, 4)                       \
    X(ALG_MODEL_QUADROTOR, 3, 3, 1, 4)                       \
    X(ALG_MODEL_QUADROTOR, 4, 3, 1, 4)                       \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 3, 0, 4)               \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 3, 0, 4)               \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 3, 1, 4)               \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 3, 1, 4)
#define ALG_INSTANTIATE_MW(PREFIX, M, P, D, E, W)                                                                         \
    PREFIX __global__ void k_newton_solve<Cfg<M, P, D, E, W>>(Params, int, uint64_t);                                     \
    PREFIX __global__ void k_mpc_loop<Cfg<M, P, D, E, W>>(Params, int, uint64_t, double*);
// Hand-off pairs: X(model, p, d, ext, w) = the budgeted one-wavefront kernel of (model, p, d, ext) parks, the team kernel of width w resumes
// (base configurations that have a team kernel in ALG_CFGS_MW)
#define ALG_CFGS_HANDOFF(X)                                 \
    X(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0, 4)               \
    X(ALG_MODEL_UNICYCLE, 3, 2, 0, 4)                        \
    X(ALG_MODEL_UNICYCLE, 4, 2, 0, 4)
#define ALG_INSTANTIATE_HO_PARK(PREFIX, M, P, D, E, W) PREFIX __global__ void k_newton_solve_ho<Cfg<M, P, D, E>>(Params, int, uint64_t, int);
#define ALG_INSTANTIATE_HO_RESUME(PREFIX, M, P, D, E, W) PREFIX __global__ void k_newton_resume<Cfg<M, P, D, E, W, 0>>(Params);
#define ALG_DEFINE_HO_RESUME(M, P, D, E, W) ALG_INSTANTIATE_HO_RESUME(template, M, P, D, E, W)
#define ALG_DECLARE_HO(M, P, D, E, W) ALG_INSTANTIATE_HO_PARK(extern template, M, P, D, E, W) ALG_INSTANTIATE_HO_RESUME(extern template, M, P, D, E, W)
#define ALG_DEFINE_MW(M, P, D, E, W) ALG_INSTANTIATE_MW(template, M, P, D, E, W)
#define ALG_DECLARE_MW(M, P, D, E, W) ALG_INSTANTIATE_MW(extern template, M, P, D, E, W)
#define ALG_DEFINE_KERNELS(M, P, D, E) ALG_INSTANTIATE_KERNELS(template, M, P, D, E)
#define ALG_DECLARE_KERNELS(M, P, D, E) ALG_INSTANTIATE_KERNELS(extern template, M, P, D, E)
