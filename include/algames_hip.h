/*
 * algames_hip.h -- C ABI of the MI355X-native batched ALGAMES Newton / augmented-Lagrangian
 * hot path (libalgames_hip.so).
 *
 * The reference (RoboticExplorationLab/Algames.jl v0.1.6) has no FFI: its boundary is the
 * Julia API itself.  Every entry point below names the reference function (file:line under
 * /root/reference) that it replaces for a *batch* of B independent games that share
 * (model, p, N, dt, options, constraint structure) and differ in data (x0, targets,
 * multipliers, iterate).  A Julia host binds these with `ccall`.
 *
 * Conventions
 *   - all floating point data is IEEE fp64, all buffers are caller-owned host memory unless a
 *     function says "device"; the library owns device memory and its HIP stream.
 *   - batched buffers are game-major and contiguous: buf[g*len + e].
 *   - return value: 0 = ALG_OK, <0 = error, message in alg_last_error() (thread local).
 *     Numerical failure of individual games is reported per game (alg_game_stats.status),
 *     never as a call error.
 *   - a handle is not thread-safe; distinct handles are independent.
 *
 * Layouts (src/core/newton_core.jl:40-89)
 *   traj  (len n+S):  [ x_1 (n) | for k=1..N-1: x_{k+1} (n), u_{1,k} (mi) .. u_{p,k} (mi),
 *                                               lambda_{1,k} (n) .. lambda_{p,k} (n) ]
 *          i.e. x_1 followed by the reference's "horizontal" order = order of `Δtraj`
 *          (src/struct/primal_dual_traj.jl:46-107).  u_{i,k} holds joint-control entries
 *          pu[i] = {i + (j-1)p} in increasing order.
 *   res   (len S):    reference "vertical" order: for i=1..p, for k=1..N-1:
 *          [opt_i,x_{k+1} (n) | opt_i,u_{i,k} (mi)], then for k=1..N-1: dyn_k (n).
 *   jac   (S x S, column-major): rows vertical order, columns horizontal order.
 *   con duals / penalties / values (len alg_con_len()):
 *          [ collision avoidance: for pair q=(i,j) in the order of
 *            add_collision_avoidance! (constraints_methods.jl:21-33: for i, for j != i),
 *            for knot k=2..N: 1 row ]  then
 *          [ control bound: for knot k=1..N-1: rows (u-u_max)(m) then (u_min-u)(m) ].
 *          Rows of infinite bounds are kept (the reference drops them,
 *          control_bound_constraint.jl:35-38); they are never active and report value -inf.
 *          Extended constraints append, in this order and only when added:
 *          [ state bound: for player i, knot k=2..N: rows (x-x_max)(n) then (x_min-x)(n) ]
 *          [ wall: for player i, knot k=2..N: one row per wall ] [ circle: for player i, knot k=2..N: one row per circle ]
 *          [ 3-D wall: for player i, knot k=2..N: one row per Wall3D ] [ cylinder: for player i, knot k=2..N: one row per cylinder ]
 *          (spherical collision avoidance uses the collision-avoidance rows)
 *          alg_get_con_len() returns the current length.
 */
#ifndef ALGAMES_HIP_H
#define ALGAMES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALG_OK 0
#define ALG_ERR_ARG (-1)        /* bad argument / unsupported configuration            */
#define ALG_ERR_DEVICE (-2)     /* HIP runtime error                                   */
#define ALG_ERR_STATE (-3)      /* call sequence error (e.g. data not set)             */

/* src/dynamics/double_integrator.jl:13-31, src/dynamics/unicycle.jl:14-32 */
#define ALG_MODEL_DOUBLE_INTEGRATOR 0
#define ALG_MODEL_UNICYCLE 1
#define ALG_MODEL_BICYCLE 2             /* src/dynamics/bicycle.jl:2-41 (lf = lr = 0.05 unless alg_set_bicycle) */
#define ALG_MODEL_QUADROTOR 3           /* src/dynamics/quadrotor.jl:3-206: n = 12 p, m = 4 p; per player [x(3) | MRP attitude (3) | v(3) | omega(3)],
                                           rotor commands w1..w4; mass 0.5 (alg_set_quadrotor), J = diag(0.0023, 0.0023, 0.004), g = (0, 0, -9.81),
                                           motor_dist 0.175, kf 1.245, km 1.0 (the constructor's values) */
#define ALG_MAX_WALLS 8
#define ALG_MAX_CIRCLES 8

/* which trajectory buffer of the problem (problem.jl:27-29) */
#define ALG_TRAJ_PD 0      /* prob.pdtraj        */
#define ALG_TRAJ_TRIAL 1   /* prob.pdtraj_trial  */
#define ALG_TRAJ_DELTA 2   /* prob.Δpdtraj (x_1 slot is zero) */

/* per-game status */
#define ALG_STATUS_OK 0
#define ALG_STATUS_SINGULAR 1   /* zero / non-finite pivot in the Newton solve (reference: SingularException) */
#define ALG_STATUS_NAN 2        /* non-finite residual */
#define ALG_STATUS_PARKED 3     /* transient: the game used up the hand-off budget of the first launch and waits for the second (alg_set_handoff);
                                   never visible after alg_newton_solve / alg_synchronize */

typedef struct alg_handle alg_handle;

/* ProblemSize + model + dt (src/struct/problem_size.jl:18-35, problem.jl:35-53) */
typedef struct alg_desc {
    int32_t model;   /* ALG_MODEL_*                                           */
    int32_t p;       /* number of players                                     */
    int32_t d;       /* integrator dimension (DoubleIntegrator; unicycle: 2)  */
    int32_t N;       /* knot points                                           */
    double  dt;      /* step                                                  */
    int32_t batch;   /* number of games B owned by this handle                */
    int32_t device;  /* HIP device ordinal (ignored by the CPU oracle)        */
} alg_desc;

/* POD mirror of the `Options` fields read on the hot path (src/struct/options.jl:5-116). */
typedef struct alg_options {
    double  amplitude_init;  /* 1e-8  */
    int32_t shift;           /* 2^10  */
    int32_t regularize;      /* true  */
    double  reg_0;           /* 1e-3  */
    double  alpha_decrease;  /* 0.5   */
    double  beta;            /* 0.01  */
    int32_t ls_iter;         /* 25    */
    int32_t dual_reset;      /* true  */
    double  delta_min;       /* 1e-9  */
    double  rho_0;           /* 1.0   */
    double  rho_increase;    /* 10.0  */
    double  rho_max;         /* 1e7   */
    double  lambda_max;      /* 1e7   */
    double  alpha_dual;      /* 1.0   */
    double  alphax_dual[10]; /* ones  */
    double  eps_dyn, eps_sta, eps_con, eps_opt; /* 1e-3 */
    int32_t outer_iter;      /* 7     */
    int32_t inner_iter;      /* 20    */
    int64_t seed;            /* 100   */
} alg_options;

/* One `Statistics` record (src/struct/statistics.jl:5-15,44-57): what record! pushes. */
typedef struct alg_record {
    int32_t outer;     /* stats.outer_iter[end]        */
    int32_t ls_j;      /* line-search count j of the step taken after this record (0 if none) */
    double  alpha;     /* step taken after this record (0 if none)                            */
    double  res;       /* ||res||_1 / S                */
    double  delta;     /* Δ_traj passed to record!     */
    double  dyn_vio, con_vio, sta_vio, opt_vio; /* the four .max scalars */
    double  t_elap;    /* stats.t_elap[end] (statistics.jl:8,34): seconds the previous inner_iteration of this solve took
                          (@elapsed at solver_methods.jl:40-42; 0 for the first record), measured on the device with the
                          100 MHz real-time counter from the first to the last instruction of the game's iteration */
} alg_record;

/* Result of newton_solve! for one game (src/problem/solver_methods.jl:5-65). */
typedef struct alg_game_stats {
    int32_t status;        /* ALG_STATUS_*                                              */
    int32_t outer_iters;   /* `out`                                                     */
    int32_t newton_iters;  /* inner iterations that performed a linear solve            */
    int32_t records;       /* stats.iter                                                */
    int32_t converged;     /* exit test solver_methods.jl:49-53 met (not the k==outer_iter arm) */
    int32_t ls_failures;   /* failed line searches                                      */
    int32_t refinements;   /* correction solves of the Newton direction's iterative refinement (alg_set_refinement; 0 for the CPU oracle, whose pivoted LU needs none) */
    int32_t reserved;      /* 0.  (Team kernels: counts line-search steps whose norm from the group pass differed from the ordinary pass's -- a self-check that must stay 0, tests/test_gpu_line_search_batch.py) */
    alg_record last;       /* final record! (solver_methods.jl:63)                      */
} alg_game_stats;

/* Result of one inner_iteration (src/problem/solver_methods.jl:67-103). */
typedef struct alg_step_info {
    int32_t status;
    int32_t control_flow;  /* 0 = :continue, 1 = :break                                 */
    int32_t ls_j;          /* j returned by line_search (0 if the step was skipped)     */
    int32_t ls_failed;     /* j == ls_iter                                              */
    double  alpha;
    double  delta;         /* Δ_step                                                    */
    alg_record rec;        /* the record! made at the top of the iteration              */
} alg_step_info;

const char* alg_last_error(void);
void alg_default_options(alg_options* o);                 /* Options() defaults, options.jl:5-116 */

/* sizes derived from a descriptor (problem_size.jl:18-35) */
int alg_dims(const alg_desc* d, int32_t* n, int32_t* m, int32_t* mi, int32_t* S,
             int32_t* traj_len, int32_t* con_len);

/* GameProblem(N, dt, x0, model, opts, game_obj, game_con)  (problem.jl:35-53) */
int  alg_create(const alg_desc* d, alg_handle** out);
void alg_destroy(alg_handle* h);
int  alg_set_options(alg_handle* h, const alg_options* o);   /* also set_constraint_params!, game_constraints.jl:33-53 */
int  alg_get_options(alg_handle* h, alg_options* o);
/* Kernel shape of the fused solver entry points (alg_newton_solve*, alg_mpc_solve): wavefronts that work on one game.
 * 1 = one game per wavefront (large batches: every SIMD holds several games); 2 / 4 = a team of wavefronts per game (small
 * batches that would leave most of the 1024 SIMDs empty: the streaming phases of the solver are spread over the team, the
 * serial Newton-direction sweeps stay on one wavefront); 0 (default) = automatic: a team kernel when one is compiled for the
 * configuration and batch x width <= 2048 wavefronts.  Results agree with the one-wavefront kernel to rounding (the norms are
 * summed in a different order).  alg_get_waves_per_game returns the width the next solve will use. */
int  alg_set_waves_per_game(alg_handle* h, int32_t waves);
int  alg_get_waves_per_game(alg_handle* h, int32_t* waves);
/* Line search (solver_methods.jl:105-125) of the team kernels and the one-wavefront unicycle kernels: once the first step size has been
 * rejected the following ones are evaluated four at a time by a norm-only pass (bit-identical norms, the same step is accepted).
 * on = 0 selects the one-by-one search in the same binary (default 1).  Nothing else changes at the boundary. */
int  alg_set_line_search_groups(alg_handle* h, int32_t on);
int  alg_get_line_search_groups(alg_handle* h, int32_t* on);
/* Straggler hand-off for heterogeneous batches (no reference counterpart: the reference solves one game at a time).  A launch lasts as
 * long as its slowest game.  With iters = K > 0 the one-wavefront solver kernel behind alg_newton_solve* gives every game a budget of
 * K inner iterations BEGUN (solver_methods.jl:38: each makes a record!, whether or not it reaches the linear solve -- a game that converges
 * in 11 Newton iterations over 4 outer iterations begins 15); a game that needs more parks -- its whole state lives in its arena chunk -- and a second launch on the same
 * stream finishes the parked games with the team kernel (four wavefronts per game), which continues the same outer / inner loops.
 * Results per game: the iterations before the hand-off are the one-wavefront kernel's, the ones after it the team kernel's (the two
 * agree to rounding: the norms are summed in a different order, see alg_set_waves_per_game).  0 (default) = off.  Only configurations
 * with a team kernel accept K > 0 (3-player DoubleIntegrator d = 2, 3- / 4-player Unicycle, base constraint set; ALG_ERR_ARG
 * otherwise); the setting is ignored while the handle runs a team kernel itself and by alg_mpc_solve.  alg_get_handoff also returns
 * the number of games the most recent solve handed over (parked_last may be NULL). */
int  alg_set_handoff(alg_handle* h, int32_t iters);
int  alg_get_handoff(alg_handle* h, int32_t* iters, int32_t* parked_last);
/* Iterative refinement of the Newton direction (replaces the backward stability of `lu(core.jac) \\ core.res`, solver_methods.jl:87).
 * The structured elimination behind alg_newton_direction / alg_newton_step / alg_newton_solve* is a block LU without pivoting across
 * blocks; the forward and costate sweeps satisfy the dynamics and opt-x rows of J d = -res by construction, so all of the elimination's
 * error surfaces in the opt-u rows, which are evaluated after every solve (the gate).  While their row-wise backward error
 * max_c |rho_c| / (|J_c| |d| + |res_c|) exceeds `tol`, the direction is corrected by one more elimination on the residual (at most
 * `max_steps` correction solves per direction).  `tol` is the tolerance for games whose largest constraint penalty (ALConVal mu) has
 * reached `mu_tight`; below that it is relaxed in proportion mu_tight / mu_max, at most 256 x (a forward-error target needs a backward
 * error of target / cond(J), and cond(J) grows with the penalties).  The dense-direction configurations (Quadrotor, n > 16) use
 * tol / 128 without relaxation.  Within 2^10 of the tolerance a correction that does not at least halve max |rho| ends the refinement.
 * Defaults: max_steps = 2 (dense-direction configurations: 8), tol = 2^-34, mu_tight = 1.6e5;
 * max_steps = 0 switches gate and refinement off (the round-3 arithmetic).  alg_game_stats.refinements counts the correction
 * solves of a newton_solve!.  A correction solve uses the trial trajectory (ALG_TRAJ_TRIAL) as its output buffer: after
 * alg_newton_direction that buffer holds the last correction (x_1 restored), until the next line search rewrites it -- as in
 * the reference, pdtraj_trial is only meaningful between a line search and the update that follows it. */
int  alg_set_refinement(alg_handle* h, int32_t max_steps, double tol, double mu_tight);
int  alg_get_refinement(alg_handle* h, int32_t* max_steps, double* tol, double* mu_tight);
/* Inspection: the gate statistics of the most recent Newton direction of every game, before any correction; out is B x 3:
 * [ max |rho| over the opt-u rows, row-wise backward error max |rho_c| / (|J_c| |d| + |ru_c|), largest row scale ].
 * Not refreshed while the gate is off (max_steps = 0); zeros for the CPU oracle. */
int  alg_get_direction_gate(alg_handle* h, double* out);
/* Launch on a caller-provided hipStream_t (e.g. torch's current stream); NULL = library stream. */
int  alg_set_stream(alg_handle* h, void* hip_stream);

/* x0: B x n */
int alg_set_x0(alg_handle* h, const double* x0);
/* GameObjective(Q,R,xf,uf,N,model) (objective.jl:12-35): diagonals on the player's own indices.
 * Qdiag, xf: [B x] p x ni ; Rdiag, uf: [B x] p x mi.  per_game=0: one set shared by all games. */
int alg_set_lqr(alg_handle* h, const double* Qdiag, const double* Rdiag, const double* xf,
                const double* uf, int32_t per_game);
/* add_collision_cost!(game_obj, radius, mu) (objective.jl:84-100); NULL,NULL removes it. */
int alg_add_collision_cost(alg_handle* h, const double* radius /*p*/, const double* mu /*p*/);
/* add_collision_avoidance!(game_con, radius::Vector) (constraints_methods.jl:21-33) */
int alg_add_collision_avoidance(alg_handle* h, const double* radius /*p*/);
/* add_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:5-19) and
 * add_spherical_collision_avoidance!(game_con, i, j, radius) (:45-64): ONE CollisionConstraint of player i against player j
 * (0-based) with its own radius -- asymmetric radii or a subset of the ordered pairs.  The multiplier / value rows keep the
 * all-pairs layout [pair q = i (p-1) + (j < i ? j : j-1)][knot]; a pair that was never added is inert (c = 0, zero Jacobian,
 * multiplier stays 0).  The first pair call on a handle starts from "no pair"; the vector forms (alg_add_collision_avoidance,
 * alg_add_spherical_collision_avoidance) set every ordered pair to r_i + r_j.  One constraint per ordered pair. */
int alg_add_collision_avoidance_pair(alg_handle* h, int32_t i, int32_t j, double radius);
int alg_add_spherical_collision_avoidance_pair(alg_handle* h, int32_t i, int32_t j, double radius);
/* add_control_bound!(game_con, u_max, u_min) (constraints_methods.jl:104-115); +-inf allowed */
int alg_add_control_bound(alg_handle* h, const double* u_max /*m*/, const double* u_min /*m*/);

/* Every adder of the extended set below (state bounds, walls, circles, and the 3-D set incl. spherical collision avoidance)
 * re-creates the handle's multiplier storage: all lambda = 0, all mu = the current opts.rho_0 -- like freshly built ALConVals
 * (it matters only when the first solve runs with dual_reset = false). */
/* BicycleGame(p; lf, lr) parameters (bicycle.jl:15); only for ALG_MODEL_BICYCLE */
int alg_set_bicycle(alg_handle* h, double lf, double lr);
/* QuadrotorGame(; p, mass) (quadrotor.jl:20-46): the constructor's one parameter (default 0.5 kg; inertia, gravity, motor_dist, kf, km are
 * fixed there); only for ALG_MODEL_QUADROTOR */
int alg_set_quadrotor(alg_handle* h, double mass);
/* add_state_bound!(game_con, i, x_max, x_min) (constraints_methods.jl:86-98; state_bound_constraint.jl): bounds on the joint
 * state (n each, +-inf allowed) attached to player `player` (0-based), knots 2..N */
int alg_add_state_bound(alg_handle* h, int32_t player, const double* x_max /*n*/, const double* x_min /*n*/);
/* add_wall_constraint!(game_con, walls) (constraints_methods.jl:152-195; wall_constraint.jl:30-96): every player, knots 2..N;
 * wall w: segment (x1,y1)-(x2,y2), normal (xv,yv) pointing into the forbidden half space */
int alg_add_wall_constraint(alg_handle* h, int32_t n_wall, const double* x1, const double* y1, const double* x2,
                            const double* y2, const double* xv, const double* yv);
/* add_circle_constraint!(game_con, xc, yc, radius) (constraints_methods.jl:120-148; TrajOpt CircleConstraint): every player */
int alg_add_circle_constraint(alg_handle* h, int32_t n_circle, const double* xc, const double* yc, const double* radius);
/* Per-player variants: add_wall_constraint!(game_con, i, walls) (constraints_methods.jl:161-187) and
 * add_circle_constraint!(game_con, i, xc, yc, radius) (:121-139).  The walls / circles join the handle's table (at most
 * ALG_MAX_WALLS / ALG_MAX_CIRCLES distinct entries; an entry that is already there is shared) and constrain player `player`
 * (0-based) only.  The constraint rows keep the layout [player][knot][table entry]: the rows of entries that do not constrain
 * a player exist but are inert (value 0, multiplier 0).  The all-player calls above replace the table. */
int alg_add_wall_constraint_player(alg_handle* h, int32_t player, int32_t n_wall, const double* x1, const double* y1,
                                   const double* x2, const double* y2, const double* xv, const double* yv);
int alg_add_circle_constraint_player(alg_handle* h, int32_t player, int32_t n_circle, const double* xc, const double* yc,
                                     const double* radius);
/* ---- 3-D half of the constraint set.  The reference addresses pz[i][1:3] (constraints_methods.jl:52-53,231-236,275): the
 * first three state entries of player i, which are its x, y, z positions for DoubleIntegratorGame(d = 3) -- the only model
 * these three calls accept (ALG_ERR_ARG otherwise). */
/* add_spherical_collision_avoidance!(game_con, radius) (constraints_methods.jl:45-81): as alg_add_collision_avoidance with the
 * 3-D distance; replaces a previously added planar collision avoidance (they share the constraint rows) */
int alg_add_spherical_collision_avoidance(alg_handle* h, const double* radius /*p*/);
/* add_wall_constraint!(game_con, walls::Vector{Wall3D}) (constraints_methods.jl:201-242; wall_constraint.jl:127-236): every
 * player, knots 2..N; wall w = parallelogram corner points p1, p2, p3 and normal v into the forbidden half space (n_wall x 3 each) */
int alg_add_wall3d_constraint(alg_handle* h, int32_t n_wall, const double* p1, const double* p2, const double* p3, const double* v);
/* add_wall_constraint!(game_con, walls::Vector{CylinderWall}) (constraints_methods.jl:249-284; cylinder_constraint.jl:35-127):
 * axis-aligned cylinders, origin p (n_cyl x 3), axis 0/1/2 = :x/:y/:z, length l, radius r; every player, knots 2..N */
int alg_add_cylinder_constraint(alg_handle* h, int32_t n_cyl, const double* p, const int32_t* axis, const double* l, const double* r);
/* add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) (constraints_methods.jl:208-247) and
 * add_wall_constraint!(game_con, i, walls::Vector{CylinderWall}) (:256-299): the constraint joins state_conlist[i] of ONE player
 * (0-based here).  Same mechanism as the planar *_player adders: the entries join the handle's table of distinct 3-D walls /
 * cylinders (at most ALG_MAX_WALLS / ALG_MAX_CIRCLES; an identical entry is shared, row w of the ABI's block is table entry w for
 * every player) and a per-player mask says whom an entry constrains; a row whose bit is clear is inert (value 0, zero Jacobian,
 * multiplier untouched).  Can be called repeatedly; after an all-player set the masks become explicit first. */
int alg_add_wall3d_constraint_player(alg_handle* h, int32_t player, int32_t n_wall, const double* p1, const double* p2, const double* p3, const double* v);
int alg_add_cylinder_constraint_player(alg_handle* h, int32_t player, int32_t n_cyl, const double* p, const int32_t* axis, const double* l, const double* r);
/* current length of the constraint dual / penalty / value vectors of one game */
int alg_get_con_len(alg_handle* h, int32_t* con_len);

/* set_traj!/get_traj! (primal_dual_traj.jl:46-107) over the batch: B x traj_len.
 * (ALG_TRAJ_TRIAL after a solve: x_1 = x0; the rest is the library's scratch -- the line search accepts a trial by exchanging buffer
 * offsets, a solve that ends on the exchanged side copies pdtraj home: the trial buffer then holds pdtraj as in the reference, otherwise the iterate before it) */
int alg_set_traj(alg_handle* h, int32_t which, const double* z);
int alg_get_traj(alg_handle* h, int32_t which, double* z);
/* ALConVal lambda / mu (Altro 0.3.0): B x con_len each; NULL pointers are skipped */
int alg_set_con_duals(alg_handle* h, const double* lambda, const double* mu);
int alg_get_con_duals(alg_handle* h, double* lambda, double* mu);

/* init_traj! + rollout!(RK3) (solver_methods.jl:12-18, primal_dual_traj.jl:29-44).
 * f_init = rand is replaced by a counter-based generator (SplitMix64 keyed by seed, global game id
 * game_id0+g, element counter) because Julia's MersenneTwister stream cannot be reproduced
 * (SURVEY.md section 7 "hard parts").  use_shift!=0 applies the `shift` warm start to the stored pdtraj.
 * Any other `opts.f_init` (options.jl:11: zeros, randn, a closure) is a host-side matter: the caller makes the
 * guess (Julia shim: the reference's own init_traj!; Python: host.init_traj_host), stores it with alg_set_traj and
 * solves with init = 0 -- the states are rolled out either way (solver_methods.jl:17). */
int alg_init_traj(alg_handle* h, int64_t game_id0, int32_t use_shift);
/* rollout!(RK3, model, pdtraj.pr) only (keeps controls/duals as set by alg_set_traj). */
int alg_rollout(alg_handle* h, int32_t which);

/* residual! + regularize_residual! (global_quantities.jl:9-86).  `which` selects the iterate;
 * the proximal term is taken w.r.t. pdtraj with reg = reg_x = reg_u (0 disables).
 * res (B x S, vertical order) and res_norm (B, ||res||_1/S) may each be NULL. */
int alg_residual(alg_handle* h, int32_t which, double reg, double* res, double* res_norm);
/* residual_jacobian! + regularize_residual_jacobian! (global_quantities.jl:109-193) at pdtraj:
 * jac is B x S x S dense column-major (parity / inspection entry point; the solver itself never
 * materialises it). */
int alg_residual_jacobian(alg_handle* h, double reg, double* jac);
/* The same for games first_game .. first_game + n_games - 1 only: jac is n_games x S x S.  The active-set inspection
 * (active_set_methods.jl:127-170 works on ONE problem) uses this so that looking at one game of a large batch does not
 * materialise B dense Jacobians. */
int alg_residual_jacobian_games(alg_handle* h, double reg, int32_t first_game, int32_t n_games, double* jac);
/* Frees the device scratch the inspection entry points (dense Jacobians, MPC state logs) grow on demand.  The solver never
 * uses that buffer; the next inspection call allocates it again. */
int alg_release_scratch(alg_handle* h);
/* Per-knot violation profiles at pdtraj: the .vio vectors the reference's violation objects carry next to .max
 * (violations.jl:5-26 dynamics, :41-67 control, :86-114 state, :140-168 optimality).  dyn, con: B x (N-1) (steps 1..N-1);
 * sta, opt: B x N (knots 1..N; sta[0] is 0: no state constraint acts on x_1).  Any pointer may be NULL.  Evaluated on the device
 * from the residual and the constraint values at the current pdtraj (one assemble pass + one reduction kernel). */
int alg_get_violation_profile(alg_handle* h, double* dyn, double* con, double* sta, double* opt);
/* Δtraj = -lu(jac) \ res ; set_traj!(Δpdtraj, Δtraj) (solver_methods.jl:87-88).  delta: B x S or NULL. */
int alg_newton_direction(alg_handle* h, double reg, double* delta, int32_t* status /*B or NULL*/);
/* line_search (solver_methods.jl:105-125) on the stored Δpdtraj. */
int alg_line_search(alg_handle* h, double reg, const double* res_norm /*B*/, double* alpha /*B*/,
                    int32_t* j /*B*/);
/* update_traj!(target, source, alpha, Δpdtraj) (primal_dual_traj.jl:109-128); alpha: B */
int alg_update_traj(alg_handle* h, int32_t target, int32_t source, const double* alpha);
/* record!'s scalars at pdtraj (statistics.jl:44-57, violations.jl) */
int alg_record_stats(alg_handle* h, alg_record* rec /*B*/);
/* reset!(game_con) (constraints_methods.jl:295-327) */
int alg_reset_con(alg_handle* h);
/* evaluate!(game_con, pdtraj.pr); dual_update!(game_con); penalty_update!(game_con)
 * (solver_methods.jl:57-61, constraints_methods.jl:329-379,421-440).  vals: B x con_len or NULL. */
int alg_dual_penalty_update(alg_handle* h, double* vals);

/* inner_iteration(prob, LS_count, t_elap, Δ, k, l) (solver_methods.jl:67-103) for every game;
 * reg is set to reg_0*l^4 as in solver_methods.jl:39.  delta_in (B or NULL = zeros) is the caller's Δ, which record! stores
 * in the Statistics record made at the top of the iteration (solver_methods.jl:75). */
int alg_newton_step(alg_handle* h, int32_t k_outer, int32_t l_inner, const double* delta_in /*B or NULL*/,
                    alg_step_info* info /*B*/);
/* newton_solve!(prob) (solver_methods.jl:5-65) for every game.  init!=0: run alg_init_traj first
 * (game ids game_id0+g); init==0: keep the stored controls/duals as the initial guess, still
 * rolling out the states (solver_methods.jl:17).  stats: B or NULL. */
int alg_newton_solve(alg_handle* h, int32_t init, int64_t game_id0, alg_game_stats* stats);
/* Same, asynchronous on the handle's stream: no host synchronisation, results stay on the device
 * until alg_get_stats().  Used by the benchmark so that HIP events bracket only device work. */
int alg_newton_solve_async(alg_handle* h, int32_t init, int64_t game_id0);
int alg_get_stats(alg_handle* h, alg_game_stats* stats /*B*/);
/* Statistics history of one game (statistics.jl:5-15): up to max_records; returns count in *n_out.
 * The history buffer is sized from the options (alg_set_options: outer_iter * inner_iter + 1 records per newton_solve!, never
 * fewer than ALG_HIST_MAX; the IBR entry points size it for ibr_iter * p solves) so that every record! of a solve is kept.
 * Only a request beyond 1 GiB of history for the whole batch is capped: alg_game_stats.records then exceeds *n_out (records
 * past the capacity are dropped, .last is always the final record) -- callers must treat n_out < records as truncation. */
#define ALG_HIST_MAX 192
int alg_get_history(alg_handle* h, int32_t game, int32_t max_records, alg_record* out, int32_t* n_out);
int alg_synchronize(alg_handle* h);
/* Diagnostic (tests): every device allocation is followed by a 4 KiB guard zone and the per-game segments of the arenas are
 * padded to 128-byte lines; returns the number of guard zones / paddings a kernel has written to (0 = clean, <0 = error). */
int alg_debug_check_guards(alg_handle* h);

/* Iterated best response.
 * alg_ibr_solve_player: ibr_newton_solve!(prob, i) (solver_methods.jl:171-228) for every game, on the stored trajectory:
 *   player `player` (0-based) best-responds -- only x, u_player, lambda_player move (masks of newton_core.jl:205-294),
 *   residual norm over the player's rows + dynamics rows, player-specific violations (statistics.jl:59-73).
 *   Statistics accumulate (the reference does not reset them between players).
 * alg_ibr_newton_solve: ibr_newton_solve!(prob; ibr_opts) (solver_methods.jl:133-169): reset!(stats), init_traj! + rollout!,
 *   then up to ibr_iter sweeps over `ordering` (p 0-based player ids), leaving when no player changed
 *   (delta_min > maximum(stats.Δ_traj), literally as in the reference). init has the meaning of alg_newton_solve. */
int alg_ibr_solve_player(alg_handle* h, int32_t player, alg_game_stats* stats /*B or NULL*/);
int alg_ibr_newton_solve(alg_handle* h, int32_t init, int64_t game_id0, int32_t ibr_iter, const int32_t* ordering,
                         double delta_min, alg_game_stats* stats /*B or NULL*/);

/* Receding-horizon (MPC) support for BASELINE config 5.  The reference has no MPC loop, only the warm-start hooks
 * `opts.shift` (init_traj!, primal_dual_traj.jl:29-44) and `opts.dual_reset` (solver_methods.jl:25); the loop is
 * builder-defined: after a solve, x0 <- RK2(x_1, u_1) (the discretisation of
 * local_quantities.jl:13), then the next newton_solve! runs with shift = 1 and dual_reset = false.
 * alg_mpc_advance performs the x0 update for every game (asynchronously on the handle's stream) and adds the
 * finished solve's newton_iters / converged flag to per-game running totals. */
int alg_mpc_advance(alg_handle* h);
int alg_mpc_totals(alg_handle* h, int64_t* newton_iters /*B or NULL*/, int64_t* converged /*B or NULL*/, int32_t reset);
/* The whole loop in one launch: `steps` x (newton_solve! with init = 1 and game ids game_id0 + t*1000003 + g, then the
 * advance above); step 0 uses the handle's shift / dual_reset, later steps shift = 1 and dual_reset = false.  Every game
 * runs its own loop (one wavefront per game): games do not wait for each other between MPC steps.  Totals accumulate as
 * with alg_mpc_advance.  states: NULL (asynchronous launch) or host array (steps+1) x B x n receiving x0 before the loop
 * and after every step (synchronous). */
int alg_mpc_solve(alg_handle* h, int32_t steps, int64_t game_id0, double* states);

#ifdef __cplusplus
}
#endif
#endif /* ALGAMES_HIP_H */
