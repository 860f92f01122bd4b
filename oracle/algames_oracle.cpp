// algames_oracle.cpp -- CPU ORACLE (test infrastructure, NOT the product).
//
// A literal fp64 restatement of the hot path of RoboticExplorationLab/Algames.jl v0.1.6
// (newton_solve! -> residual!/residual_jacobian! -> lu \ -> line_search), one game at a time,
// batched with an OpenMP loop over games.  It deliberately keeps the reference's structure:
// a global S-vector residual in "vertical" order, a global S x S Jacobian in
// (vertical, horizontal) order, a general partial-pivot LU of that matrix (stand-in for
// UMFPACK `lu`, src/problem/solver_methods.jl:87), >= 3 residual evaluations per Newton
// iteration (solver_methods.jl:73, statistics.jl:50, solver_methods.jl:113), forward-mode
// dual numbers for the RK2 Jacobian (stand-in for ForwardDiff inside
// RobotDynamics.discrete_jacobian!, src/problem/local_quantities.jl:20-27).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY PINNING.  The reference is Julia with un-vendored dependencies and cannot be executed
// in this environment (no julia binary).  This oracle is pinned against every
// literal known-answer value the reference's own tests hold for this path (SURVEY.md Appendix B;
// tests/test_oracle_kat.py cites each test file:line) and the five end-to-end convergence
// thresholds of test/problem/solver_methods.jl.  The following third-party formulas are
// restated from the published source of the pinned dependency versions and are NOT fixed by
// any literal value in the reference tree ("parity unpinned" for these items): the RK2 / RK3
// formulas of RobotDynamics 0.3.1, TrajectoryOptimization 0.4.1's CollisionConstraint value and
// Jacobian, the terminal-knot dt convention.  They are covered only by the end-to-end
// thresholds.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).

#include "../include/algames_hip.h"

#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#if defined(ORC_REAL_KIND) && ORC_REAL_KIND == 2
#include <quadmath.h>
#endif

namespace {
// Scalar type of the restated arithmetic.  `double` is the oracle proper (the reference's Float64).  The same source compiled with
// -DORC_REAL_KIND=1 (long double, 64-bit mantissa: liboracle_x.so) or 2 (__float128: liboracle_q.so) is the ARBITER of the parity
// tests: where the HIP path and the double oracle disagree beyond the tolerances (ill-conditioned, diverging problems), both are
// compared with the extended-precision run of the very same algorithm on the very same double inputs.  The C ABI stays double.
#if !defined(ORC_REAL_KIND) || ORC_REAL_KIND == 0
typedef double real;
#elif ORC_REAL_KIND == 1
typedef long double real;
#else
typedef __float128 real;
#endif
#if defined(ORC_REAL_KIND) && ORC_REAL_KIND == 2
inline real r_sqrt(real x) { return sqrtq(x); }
inline real r_sin(real x) { return sinq(x); }
inline real r_cos(real x) { return cosq(x); }
inline real r_tan(real x) { return tanq(x); }
inline real r_atan2(real y, real x) { return atan2q(y, x); }
inline real r_fabs(real x) { return fabsq(x); }
inline bool r_isfinite(real x) { return finiteq(x) != 0; }
#else
inline real r_sqrt(real x) { return std::sqrt(x); }
inline real r_sin(real x) { return std::sin(x); }
inline real r_cos(real x) { return std::cos(x); }
inline real r_tan(real x) { return std::tan(x); }
inline real r_atan2(real y, real x) { return std::atan2(y, x); }
inline real r_fabs(real x) { return std::fabs(x); }
inline bool r_isfinite(real x) { return std::isfinite(x); }
#endif

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

// ------------------------------------------------------------------------------------------
// Sizes and index maps
// ------------------------------------------------------------------------------------------
struct Dims {
    int model = 0, p = 0, d = 0, N = 0;
    int n = 0, m = 0, mi = 0, ni = 0, S = 0, b = 0;
    int traj_len = 0, npair = 0, col_len = 0, ctl_len = 0, con_len = 0;
    int nwall = 0, ncirc = 0, has_sb = 0, sb_len = 0, wall_len = 0, circ_len = 0;      // extended constraints
    int nwall3 = 0, ncyl = 0, wall3_len = 0, cyl_len = 0, ca_dim = 2;                  // 3-D half: Wall3D, Cylinder, spherical collision avoidance
    real dt = 0, lf = 0.05, lr = 0.05;                                              // BicycleGame(lf, lr), bicycle.jl:15
    real qmass = 0.5;                                                               // QuadrotorGame(; mass), quadrotor.jl:20
    // src/struct/problem_size.jl:18-35 ; src/dynamics/double_integrator.jl:13-25 ; unicycle.jl:14-25
    bool init(const alg_desc& a) {
        model = a.model; p = a.p; N = a.N; dt = a.dt;
        if (p < 1 || p > 10 || N < 2) return false;
        if (model == ALG_MODEL_DOUBLE_INTEGRATOR) {
            d = a.d; if (d < 1 || d > 3) return false;
            n = 2 * d * p; m = d * p; mi = d; ni = 2 * d;
        } else if (model == ALG_MODEL_UNICYCLE || model == ALG_MODEL_BICYCLE) {
            d = 2; n = 4 * p; m = 2 * p; mi = 2; ni = 4;
        } else if (model == ALG_MODEL_QUADROTOR) {           // quadrotor.jl:20-46
            if (p > 4) return false;
            d = 3; n = 12 * p; m = 4 * p; mi = 4; ni = 12;
        } else return false;
        S = n * p * (N - 1) + m * (N - 1) + n * (N - 1);   // problem_size.jl:22
        b = n + m + p * n;
        traj_len = n + S;
        npair = p * (p - 1);
        col_len = npair * (N - 1);
        ctl_len = 2 * m * (N - 1);
        recount();
        return true;
    }
    void recount() {
        sb_len = has_sb ? p * 2 * n * (N - 1) : 0; wall_len = p * nwall * (N - 1); circ_len = p * ncirc * (N - 1);
        wall3_len = p * nwall3 * (N - 1); cyl_len = p * ncyl * (N - 1);
        con_len = col_len + ctl_len + sb_len + wall_len + circ_len + wall3_len + cyl_len;
    }
    int o_sb(int i, int k /*knot 1..N-1*/, int row) const { return col_len + ctl_len + (i * (N - 1) + (k - 1)) * 2 * n + row; }
    int o_wall(int i, int k, int w) const { return col_len + ctl_len + sb_len + (i * (N - 1) + (k - 1)) * nwall + w; }
    int o_circ(int i, int k, int c) const { return col_len + ctl_len + sb_len + wall_len + (i * (N - 1) + (k - 1)) * ncirc + c; }
    int o_wall3(int i, int k, int w) const { return col_len + ctl_len + sb_len + wall_len + circ_len + (i * (N - 1) + (k - 1)) * nwall3 + w; }
    int o_cyl(int i, int k, int c) const { return col_len + ctl_len + sb_len + wall_len + circ_len + wall3_len + (i * (N - 1) + (k - 1)) * ncyl + c; }
    // index sets pu/px/pz = {i + (j-1)p} (double_integrator.jl:18-20), 0-based
    int pu(int i, int j) const { return i + j * p; }
    int pz(int i, int j) const { return i + j * p; }
    int px(int i, int j) const { return i + j * p; }
    // horizontal order (newton_core.jl:65-89), 0-based knot k = 0..N-2 <-> reference k = 1..N-1
    int hx(int k) const { return k * b; }                       // x_{k+1}
    int hu(int k, int i) const { return k * b + n + i * mi; }   // u_{i,k}
    int hl(int k, int i) const { return k * b + n + m + i * n; } // lambda_{i,k}
    // vertical order (newton_core.jl:40-63)
    int vx(int i, int k) const { return i * (N - 1) * (n + mi) + k * (n + mi); }      // opt_i, x_{k+1}
    int vu(int i, int k) const { return i * (N - 1) * (n + mi) + k * (n + mi) + n; }  // opt_i, u_{i,k}
    int vd(int k) const { return p * (N - 1) * (n + mi) + k * n; }                     // dyn_k
    // ordered pair index, order of add_collision_avoidance! (constraints_methods.jl:21-33)
    int pair(int i, int j) const { return i * (p - 1) + (j < i ? j : j - 1); }
};

// ------------------------------------------------------------------------------------------
// Forward-mode dual numbers (stand-in for ForwardDiff 0.10 used by discrete_jacobian!)
// ------------------------------------------------------------------------------------------
constexpr int MAXD = 96;
// Per-thread grow-only scratch (keeps the allocator out of the OpenMP loop over games: the band matrix alone is
// several MB per Newton iteration).  Contents are unspecified on entry; every user writes before it reads.
#define TL_VEC(T, name, count)                                                  \
    thread_local std::vector<T> name##_tl;                                      \
    if (name##_tl.size() < (size_t)(count)) name##_tl.resize((size_t)(count)); \
    std::vector<T>& name = name##_tl
struct Dual {
    real v = 0;
    std::array<real, MAXD> e;            // only e[0 .. nd) is ever written or read
    int nd = 0;
};
inline Dual dconst(real v, int nd) { Dual r; r.v = v; r.nd = nd; for (int i = 0; i < nd; i++) r.e[i] = 0.0; return r; }
inline Dual operator+(const Dual& a, const Dual& b) { Dual r; r.nd = a.nd; r.v = a.v + b.v; for (int i = 0; i < a.nd; i++) r.e[i] = a.e[i] + b.e[i]; return r; }
inline Dual operator*(const Dual& a, const Dual& b) { Dual r; r.nd = a.nd; r.v = a.v * b.v; for (int i = 0; i < a.nd; i++) r.e[i] = a.e[i] * b.v + a.v * b.e[i]; return r; }
inline Dual operator-(const Dual& a, const Dual& b) { Dual r; r.nd = a.nd; r.v = a.v - b.v; for (int i = 0; i < a.nd; i++) r.e[i] = a.e[i] - b.e[i]; return r; }
inline Dual operator+(const Dual& a, real s) { Dual r = a; r.v = a.v + s; return r; }
inline Dual operator*(real s, const Dual& a);
inline Dual operator/(const Dual& a, const Dual& b) { Dual r; r.nd = a.nd; r.v = a.v / b.v; const real ib = 1.0 / b.v; for (int i = 0; i < a.nd; i++) r.e[i] = (a.e[i] - r.v * b.e[i]) * ib; return r; }
// max(0, a) as ForwardDiff differentiates it (derivative of the selected branch)
inline Dual dmax0(const Dual& a) { if (a.v > 0.0) return a; Dual r; r.nd = a.nd; r.v = 0.0; for (int i = 0; i < a.nd; i++) r.e[i] = 0.0; return r; }
inline real dmax0(real a) { return a > 0.0 ? a : 0.0; }
inline Dual operator*(const Dual& a, real s) { Dual r; r.nd = a.nd; r.v = a.v * s; for (int i = 0; i < a.nd; i++) r.e[i] = a.e[i] * s; return r; }
inline Dual dcos(const Dual& a) { Dual r; r.nd = a.nd; r.v = r_cos(a.v); real s = -r_sin(a.v); for (int i = 0; i < a.nd; i++) r.e[i] = s * a.e[i]; return r; }
inline Dual dsin(const Dual& a) { Dual r; r.nd = a.nd; r.v = r_sin(a.v); real c = r_cos(a.v); for (int i = 0; i < a.nd; i++) r.e[i] = c * a.e[i]; return r; }
inline Dual dtan(const Dual& a) { Dual r; r.nd = a.nd; r.v = r_tan(a.v); real s = 1.0 + r.v * r.v; for (int i = 0; i < a.nd; i++) r.e[i] = s * a.e[i]; return r; }
inline Dual datan2(const Dual& y, real x) { Dual r; r.nd = y.nd; r.v = r_atan2(y.v, x); real s = x / (x * x + y.v * y.v); for (int i = 0; i < y.nd; i++) r.e[i] = s * y.e[i]; return r; }
inline Dual operator*(real s, const Dual& a) { return a * s; }
inline real dtan(real a) { return r_tan(a); }
inline real datan2(real y, real x) { return r_atan2(y, x); }
inline real dcos(real a) { return r_cos(a); }
inline real dsin(real a) { return r_sin(a); }

// continuous dynamics.  DoubleIntegrator: xdot = [x[m+1:n]; u] (double_integrator.jl:27-31).
// Unicycle: xdot_i = cos(th_i) v_i, ydot_i = sin(th_i) v_i, thdot = u[1:p], vdot = u[p+1:2p]
// (unicycle.jl:27-32; state = [x(1..p), y(1..p), th(1..p), v(1..p)]).
template <class T>
void dynamics(const Dims& D, const T* x, const T* u, T* xd) {
    if (D.model == ALG_MODEL_DOUBLE_INTEGRATOR) {
        for (int i = 0; i < D.m; i++) xd[i] = x[D.m + i];
        for (int i = 0; i < D.m; i++) xd[D.m + i] = u[i];
    } else if (D.model == ALG_MODEL_UNICYCLE) {
        const int P = D.p, M = D.m;
        for (int i = 0; i < P; i++) xd[i] = dcos(x[M + i]) * x[M + i + P];
        for (int i = 0; i < P; i++) xd[P + i] = dsin(x[M + i]) * x[M + i + P];
        for (int i = 0; i < M; i++) xd[M + i] = u[i];
    } else if (D.model == ALG_MODEL_QUADROTOR) {
        // QuadrotorGame (quadrotor.jl:49-121), player i: r = x[(0..2)P+i], MRP g = x[(3..5)P+i], v = x[(6..8)P+i], w = x[(9..11)P+i];
        // Rotations.jl 1.0 MRP [restated from the published source; parity unpinned]: rotation matrix of g (via the unit quaternion
        // ((1 - |g|^2), 2 g) / (1 + |g|^2)), kinematics(g, w) = 1/4 ((1 - |g|^2) w + 2 g x w + 2 (g . w) g)
        const int P = D.p;
        const real mass = D.qmass, Jd[3] = {0.0023, 0.0023, 0.004}, grav = -9.81, L = 0.1750, kf = 1.245, km = 1.0;
        for (int i = 0; i < P; i++) {
            const T g0 = x[3 * P + i], g1 = x[4 * P + i], g2 = x[5 * P + i];
            const T w0 = x[9 * P + i], w1 = x[10 * P + i], w2 = x[11 * P + i];
            const T F1 = dmax0(u[0 * P + i] * kf), F2 = dmax0(u[1 * P + i] * kf), F3 = dmax0(u[2 * P + i] * kf), F4 = dmax0(u[3 * P + i] * kf);
            const T Ft = F1 + F2 + F3 + F4;                                   // total rotor force along body z (forces, :51-69)
            const T s = g0 * g0 + g1 * g1 + g2 * g2;
            const T den = (s + 1.0) * (s + 1.0);
            // third column of R = I + (4 (1 - s) [g x] + 8 [g x]^2) / (1 + s)^2
            const T c4 = ((s * (-1.0)) + 1.0) * 4.0;
            const T r02 = (c4 * g1 + (g0 * g2) * 8.0) / den;
            const T r12 = ((c4 * g0) * (-1.0) + (g1 * g2) * 8.0) / den;
            const T r22 = (((g0 * g0 + g1 * g1) * (-8.0)) / den) + 1.0;
            // moments (:71-94): tau = [L (F2 - F4), L (F3 - F1), km (w1 - w2 + w3 - w4)]
            const T t0 = (F2 - F4) * L, t1 = (F3 - F1) * L, t2 = (u[0 * P + i] - u[1 * P + i] + u[2 * P + i] - u[3 * P + i]) * km;
            // xdot = v
            for (int a = 0; a < 3; a++) xd[a * P + i] = x[(6 + a) * P + i];
            // qdot = kinematics(MRP, w)
            const T gw = g0 * w0 + g1 * w1 + g2 * w2, oms = (s * (-1.0)) + 1.0;
            xd[3 * P + i] = (oms * w0 + (g1 * w2 - g2 * w1) * 2.0 + (gw * g0) * 2.0) * 0.25;
            xd[4 * P + i] = (oms * w1 + (g2 * w0 - g0 * w2) * 2.0 + (gw * g1) * 2.0) * 0.25;
            xd[5 * P + i] = (oms * w2 + (g0 * w1 - g1 * w0) * 2.0 + (gw * g2) * 2.0) * 0.25;
            // vdot = (m g + R F) / m
            xd[6 * P + i] = (r02 * Ft) * (1.0 / mass);
            xd[7 * P + i] = (r12 * Ft) * (1.0 / mass);
            xd[8 * P + i] = ((r22 * Ft) * (1.0 / mass)) + grav;
            // wdot = Jinv (tau - w x (J w))
            xd[9 * P + i] = (t0 - (w1 * w2) * (Jd[2] - Jd[1])) * (1.0 / Jd[0]);
            xd[10 * P + i] = (t1 - (w2 * w0) * (Jd[0] - Jd[2])) * (1.0 / Jd[1]);
            xd[11 * P + i] = (t2 - (w0 * w1) * (Jd[1] - Jd[0])) * (1.0 / Jd[2]);
        }
    } else {
        // BicycleGame (bicycle.jl:28-41): X = [x, y, v, psi] (each block of P), U = [a, delta];
        // beta = atan(lr tan(delta), lr + lf); Xdot = [v cos(beta+psi), v sin(beta+psi), a, v sin(beta)/lr]
        const int P = D.p; const real L = D.lr + D.lf;
        for (int i = 0; i < P; i++) {
            const T beta = datan2(dtan(u[P + i]) * D.lr, L);
            xd[i] = x[2 * P + i] * dcos(beta + x[3 * P + i]);
            xd[P + i] = x[2 * P + i] * dsin(beta + x[3 * P + i]);
            xd[2 * P + i] = u[i];
            xd[3 * P + i] = (x[2 * P + i] * dsin(beta)) * (1.0 / D.lr);
        }
    }
}

// RobotDynamics 0.3.1 discrete_dynamics(RK2,...): k1 = f(x,u) dt; k2 = f(x + k1/2, u) dt; x + k2
// [restated from the published source; parity unpinned, see header]
template <class T>
void rk2(const Dims& D, const T* x, const T* u, T* xn) {
    TL_VEC(T, k1, D.n); TL_VEC(T, xm, D.n); TL_VEC(T, k2, D.n);
    dynamics(D, x, u, k1.data());
    for (int i = 0; i < D.n; i++) xm[i] = x[i] + k1[i] * (D.dt * 0.5);
    dynamics(D, xm.data(), u, k2.data());
    for (int i = 0; i < D.n; i++) xn[i] = x[i] + k2[i] * D.dt;
}
// RobotDynamics 0.3.1 discrete_dynamics(RK3,...) used by rollout! (solver_methods.jl:17)
void rk3(const Dims& D, const real* x, const real* u, real* xn) {
    TL_VEC(real, k1, D.n); TL_VEC(real, k2, D.n); TL_VEC(real, k3, D.n); TL_VEC(real, t, D.n);
    dynamics(D, x, u, k1.data());
    for (int i = 0; i < D.n; i++) { k1[i] *= D.dt; t[i] = x[i] + k1[i] / 2; }
    dynamics(D, t.data(), u, k2.data());
    for (int i = 0; i < D.n; i++) { k2[i] *= D.dt; t[i] = x[i] - k1[i] + 2 * k2[i]; }
    dynamics(D, t.data(), u, k3.data());
    for (int i = 0; i < D.n; i++) { k3[i] *= D.dt; xn[i] = x[i] + (k1[i] + 4 * k2[i] + k3[i]) / 6; }
}
// ∇dynamics! (local_quantities.jl:20-27): n x (n+m) Jacobian [A B] of the RK2 map, row-major J[r*(n+m)+c]
void rk2_jacobian(const Dims& D, const real* x, const real* u, real* J) {
    const int nd = D.n + D.m;
    TL_VEC(Dual, xd, D.n); TL_VEC(Dual, ud, D.m); TL_VEC(Dual, xn, D.n);
    for (int i = 0; i < D.n; i++) { xd[i] = dconst(x[i], nd); xd[i].e[i] = 1.0; }
    for (int i = 0; i < D.m; i++) { ud[i] = dconst(u[i], nd); ud[i].e[D.n + i] = 1.0; }
    rk2(D, xd.data(), ud.data(), xn.data());
    for (int r = 0; r < D.n; r++) for (int c = 0; c < nd; c++) J[r * nd + c] = xn[r].e[c];
}

// SplitMix64-based counter RNG shared bit-for-bit with the device:
// value(seed, game, counter) in [0,1)
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline real counter_uniform(uint64_t seed, uint64_t game, uint64_t counter) {
    uint64_t h = splitmix64(seed ^ splitmix64(game * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull));
    h = splitmix64(h + counter * 0x9E3779B97F4A7C15ull);
    return (real)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------
// One game
// ------------------------------------------------------------------------------------------
struct Shared {
    Dims D;
    alg_options opt;
    bool has_colcost = false, has_colavoid = false, has_ctl = false;
    std::vector<real> cc_radius, cc_mu;   // collision cost (objective.jl:84-100)
    std::vector<real> ca_radius;          // collision avoidance radii per player (vector form of the adder)
    // per ordered pair: radius of its CollisionConstraint, < 0 = this pair carries none (add_collision_avoidance!(game_con, i, j, radius),
    // constraints_methods.jl:5-19; the vector form fills every pair with r_i + r_j, :21-33)
    std::vector<real> pair_r;
    bool pair_on(int i, int j) const { return pair_r[(size_t)i * D.p + j] >= 0.0; }
    std::vector<real> umax, umin;         // control bound
    std::vector<real> sbmax, sbmin;       // state bounds [p][n] (+-inf where absent)
    std::vector<real> wx1, wy1, wx2, wy2, wxv, wyv;   // walls
    std::vector<real> cxc, cyc, crad;     // circles
    // per-player sets (add_wall_constraint!(game_con, i, walls) / add_circle_constraint!(game_con, i, ...), constraints_methods.jl:121-187):
    // the reference pushes the constraint object to state_conval[i] only; here the table is shared and bit w of wall_mask[i] says
    // whether entry w constrains player i -- a row whose bit is clear evaluates to c = 0 with a zero Jacobian (inert)
    unsigned wall_mask[10], circ_mask[10];
    // the same for add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) / (game_con, i, walls::Vector{CylinderWall}) (constraints_methods.jl:208-247, 256-299)
    unsigned wall3_mask[10], cyl_mask[10];
    Shared() { for (int i = 0; i < 10; i++) wall_mask[i] = circ_mask[i] = wall3_mask[i] = cyl_mask[i] = 0xffffffffu; }
    std::vector<real> w3p1, w3p2, w3p3, w3v;   // Wall3D(p1, p2, p3, v): nwall3 x 3 each (constraints_methods.jl:201-206)
    std::vector<real> cyp, cyl, cyr; std::vector<int> cyax;   // CylinderWall(p, v, l, r): ncyl x 3, axis 0/1/2 = :x/:y/:z (:249-254)
};

struct Game {
    // joint-dimension zero-padded LQR data per player (objective.jl:24-28)
    std::vector<real> Q, R, xf, uf;     // p*n, p*m, p*n, p*m
    std::vector<real> x0;
    std::vector<real> z[3];             // pdtraj, trial, delta (traj_len each)
    std::vector<real> lam, mu, vals;    // con_len
    std::vector<alg_record> hist;
    alg_game_stats st{};
    int64_t mpc_iters = 0, mpc_conv = 0;
    real max_delta = 0.0;       // maximum(prob.stats.Δ_traj) over the Statistics history
    double t_elap = 0.0;        // @elapsed of the previous inner iteration (statistics.jl:8; 0 before the first)
    // scratch
    std::vector<real> res, jac;
};

struct Handle {
    Shared sh;
    std::vector<Game> g;
    bool x0_set = false, lqr_set = false;
};

inline const real* state(const Dims& D, const std::vector<real>& z, int k) {  // knot k = 0..N-1
    return k == 0 ? z.data() : z.data() + D.n + D.hx(k - 1);
}
inline real* state(const Dims& D, std::vector<real>& z, int k) {
    return k == 0 ? z.data() : z.data() + D.n + D.hx(k - 1);
}
inline void get_control(const Dims& D, const std::vector<real>& z, int k, real* u) {  // joint order
    for (int i = 0; i < D.p; i++) for (int j = 0; j < D.mi; j++) u[D.pu(i, j)] = z[D.n + D.hu(k, i) + j];
}
inline void set_control(const Dims& D, std::vector<real>& z, int k, const real* u) {
    for (int i = 0; i < D.p; i++) for (int j = 0; j < D.mi; j++) z[D.n + D.hu(k, i) + j] = u[D.pu(i, j)];
}
inline const real* dual(const Dims& D, const std::vector<real>& z, int i, int k) {
    return z.data() + D.n + D.hl(k, i);
}

// ---- objective (src/objective/objective.jl) ------------------------------------------------
// cost_gradient! / TrajectoryOptimization.cost_gradient!(E,obj,traj,true): q scaled by dt for
// k<N and by 1 at the terminal knot, r scaled by dt for k<N and 0 at the terminal knot
// [PINNED test/objective/objective.jl:52-64].
// q (n) of player i at knot k for all objectives j (LQR + collision costs), summed as in
// global_quantities.jl:26-31.
void cost_grad_x(const Shared& sh, const Game& g, int i, int k, const real* x, real* q) {
    const Dims& D = sh.D;
    const real w = (k < D.N - 1) ? D.dt : 1.0;
    for (int r = 0; r < D.n; r++) q[r] = w * (g.Q[i * D.n + r] * (x[r] - g.xf[i * D.n + r]));   // LQRCost: Q(x-xf)
    if (sh.has_colcost) {
        // CollisionCost gradient (objective.jl:134-149)
        const real eps = 1e-10, eps_norm = eps * r_sqrt((real)D.n);
        for (int j = 0; j < D.p; j++) if (j != i) {
            real dl[2], nrm = 0;
            for (int a = 0; a < 2; a++) { dl[a] = x[D.px(i, a)] - x[D.px(j, a)]; nrm += dl[a] * dl[a]; }
            nrm = r_sqrt(nrm);
            const real mu = sh.cc_mu[i], rad = sh.cc_radius[i];
            if (std::max<real>(0.0, rad - nrm) > 0.0) {
                for (int a = 0; a < 2; a++) {
                    real gg = mu * (rad * (eps + dl[a]) / (eps_norm + nrm) - dl[a]);
                    q[D.px(i, a)] += w * (-gg);
                    q[D.px(j, a)] += w * (gg);
                }
            }
        }
    }
}
// r[pu[i]] of player i at knot k < N-1 (global_quantities.jl:34-40); only the LQR term is non-zero
void cost_grad_u(const Shared& sh, const Game& g, int i, const real* u, real* r /*mi*/) {
    const Dims& D = sh.D;
    for (int j = 0; j < D.mi; j++) { int c = D.pu(i, j); r[j] = D.dt * (g.R[i * D.m + c] * (u[c] - g.uf[i * D.m + c])); }
}
// cost_hessian!: Q (n x n, row-major) of player i at knot k, all objectives (global_quantities.jl:128-136)
void cost_hess_x(const Shared& sh, const Game& g, int i, int k, const real* x, real* Qm) {
    const Dims& D = sh.D;
    const real w = (k < D.N - 1) ? D.dt : 1.0;
    std::fill(Qm, Qm + D.n * D.n, 0.0);
    for (int r = 0; r < D.n; r++) Qm[r * D.n + r] += w * g.Q[i * D.n + r];
    if (sh.has_colcost) {
        // CollisionCost Hessian (objective.jl:157-173)
        for (int j = 0; j < D.p; j++) if (j != i) {
            real dl[2], nrm = 0;
            for (int a = 0; a < 2; a++) { dl[a] = x[D.px(i, a)] - x[D.px(j, a)]; nrm += dl[a] * dl[a]; }
            nrm = r_sqrt(nrm);
            const real mu = sh.cc_mu[i], rad = sh.cc_radius[i];
            if (std::max<real>(0.0, rad - nrm) > 0.0) {
                for (int a = 0; a < 2; a++) for (int c = 0; c < 2; c++) {
                    real h = mu * ((a == c ? 1.0 : 0.0) - (a == c ? rad / nrm : 0.0) + rad * (dl[a] * dl[c]) / (nrm * nrm * nrm));
                    Qm[D.px(i, a) * D.n + D.px(i, c)] += w * h;
                    Qm[D.px(i, a) * D.n + D.px(j, c)] += -w * h;
                    Qm[D.px(j, a) * D.n + D.px(i, c)] += -w * h;
                    Qm[D.px(j, a) * D.n + D.px(j, c)] += w * h;
                }
            }
        }
    }
}

// ---- constraints ----------------------------------------------------------------------------
// con buffer offsets
inline int con_col(const Dims& D, int pairq, int k /*knot 1..N-1 (0-based)*/) { return pairq * (D.N - 1) + (k - 1); }
inline int con_ctl(const Dims& D, int k /*0..N-2*/, int row) { return D.col_len + k * 2 * D.m + row; }

// TrajectoryOptimization 0.4.1 CollisionConstraint: c = radius^2 - |x[x1]-x[x2]|^2, d c/d x1 = -2 d,
// d c/d x2 = 2 d  [restated; parity unpinned].  Pair radius = r_i + r_j (constraints_methods.jl:27-29).
inline real colavoid_val(const Shared& sh, int i, int j, const real* x, real* dl) {
    const Dims& D = sh.D;
    real R = sh.pair_r[(size_t)i * D.p + j], s = 0;
    // add_collision_avoidance!: px[i] (2 positions, constraints_methods.jl:13); add_spherical_collision_avoidance!: pz[i][1:3] (:52-54)
    for (int a = 0; a < D.ca_dim; a++) { dl[a] = x[D.pz(i, a)] - x[D.pz(j, a)]; s += dl[a] * dl[a]; }
    return R * R - s;
}
// ControlBoundConstraint evaluate (control_bound_constraint.jl:94-96): [u - u_max; u_min - u]
inline real ctl_val(const Shared& sh, const real* u, int row) {
    const int m = sh.D.m;
    return row < m ? u[row] - sh.umax[row] : sh.umin[row - m] - u[row - m];
}
inline real al_active_mu(real c, real lam, real mu);
// WallConstraint evaluate / jacobian (wall_constraint.jl:68-96): c = ((x-x1) xv + (y-y1) yv) left right, grad = [xv, yv] left right
inline real wall_val(const Shared& sh, int w, real x, real y, real* gx, real* gy) {
    const real x1 = sh.wx1[w], y1 = sh.wy1[w], x2 = sh.wx2[w], y2 = sh.wy2[w], xv = sh.wxv[w], yv = sh.wyv[w];
    const real left = ((x - x1) * (x2 - x1) + (y - y1) * (y2 - y1) > 0) ? 1.0 : 0.0;
    const real right = ((x - x2) * (x1 - x2) + (y - y2) * (y1 - y2) > 0) ? 1.0 : 0.0;
    *gx = left * right * xv; *gy = left * right * yv;
    return ((x - x1) * xv + (y - y1) * yv) * left * right;
}
// TrajectoryOptimization 0.4.1 CircleConstraint: c = r^2 - (x-xc)^2 - (y-yc)^2, grad = [-2(x-xc), -2(y-yc)]  [restated; parity unpinned]
inline real circ_val(const Shared& sh, int c, real x, real y, real* gx, real* gy) {
    const real dx = x - sh.cxc[c], dy = y - sh.cyc[c];
    *gx = -2.0 * dx; *gy = -2.0 * dy;
    return -(dx * dx) - (dy * dy) + sh.crad[c] * sh.crad[c];
}
// Wall3DConstraint evaluate / jacobian! (wall_constraint.jl:186-236): c = (x - p1).v inside the slab spanned by (p1,p2) and (p2,p3)
inline real wall3_val(const Shared& sh, int w, const real* q /*x y z*/, real* gv /*3*/) {
    const real* p1 = &sh.w3p1[3 * w]; const real* p2 = &sh.w3p2[3 * w]; const real* p3 = &sh.w3p3[3 * w]; const real* v = &sh.w3v[3 * w];
    auto dot = [&](const real* a, const real* b2, const real* c) { return (q[0] - a[0]) * (b2[0] - c[0]) + (q[1] - a[1]) * (b2[1] - c[1]) + (q[2] - a[2]) * (b2[2] - c[2]); };
    const real left = dot(p1, p2, p1) > 0 ? 1.0 : 0.0, right = dot(p2, p1, p2) > 0 ? 1.0 : 0.0;
    const real bottom = dot(p3, p2, p3) > 0 ? 1.0 : 0.0, top = dot(p2, p3, p2) > 0 ? 1.0 : 0.0;
    const real in = left * right * bottom * top;
    for (int a = 0; a < 3; a++) gv[a] = in * v[a];
    return ((q[0] - p1[0]) * v[0] + (q[1] - p1[1]) * v[1] + (q[2] - p1[2]) * v[2]) * in;
}
// CylinderConstraint evaluate / jacobian! (cylinder_constraint.jl:68-127): axis-aligned cylinder of radius r starting at p,
// length l along axis v; c = r^2 - (squared distance to the axis) while 0 < (q - p)[v] < l, else 0
inline real cyl_val(const Shared& sh, int c, const real* q, real* gv /*3*/) {
    const real* p = &sh.cyp[3 * c]; const int ax = sh.cyax[c]; const real l = sh.cyl[c], r = sh.cyr[c];
    const real t0[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
    const real valid = (t0[ax] > 0.0 && t0[ax] < l) ? 1.0 : 0.0;
    real out = r * r - t0[0] * t0[0] - t0[1] * t0[1] - t0[2] * t0[2] + t0[ax] * t0[ax];
    for (int a = 0; a < 3; a++) gv[a] = (a == ax) ? 0.0 : -valid * 2 * t0[a];
    return out * valid;
}
// StateBoundConstraint evaluate (state_bound_constraint.jl:85-87): [x - x_max; x_min - x]
inline real sb_val(const Shared& sh, int i, const real* x, int row) {
    const int n = sh.D.n;
    return row < n ? x[row] - sh.sbmax[i * n + row] : sh.sbmin[i * n + row - n] - x[row - n];
}
// Adds the AL gradient (into res rows of player i at knot k) and/or the AL Hessian (through add) of the extended
// state constraints of player i at knot k (constraint_derivatives.jl:10-19,47-58): all are per-row scalar constraints.
template <class Add>
void ext_state_con(const Shared& sh, Game& g, const std::vector<real>& z, int i, int k, real* grad /*n or null*/, Add add) {
    const Dims& D = sh.D; const real* x = state(D, z, k);
    auto row = [&](int ci, real c, const int* idx, const real* gv, int cnt) {
        g.vals[ci] = c;
        if (!r_isfinite(c)) return;
        const real am = al_active_mu(c, g.lam[ci], g.mu[ci]);
        const real w = g.lam[ci] + am * c;
        for (int a = 0; a < cnt; a++) { if (grad) grad[idx[a]] += gv[a] * w; for (int b2 = 0; b2 < cnt; b2++) if (am != 0.0) add(idx[a], idx[b2], am * gv[a] * gv[b2]); }
    };
    if (D.has_sb) for (int r = 0; r < 2 * D.n; r++) { int idx[1] = {r % D.n}; real gv[1] = {r < D.n ? 1.0 : -1.0}; row(D.o_sb(i, k, r), sb_val(sh, i, x, r), idx, gv, 1); }
    const int idx2[2] = {D.px(i, 0), D.px(i, 1)};
    for (int w = 0; w < D.nwall; w++) {
        real gv[2]; const real on = (real)((sh.wall_mask[i] >> w) & 1u);
        const real c = on * wall_val(sh, w, x[idx2[0]], x[idx2[1]], &gv[0], &gv[1]); gv[0] *= on; gv[1] *= on;
        row(D.o_wall(i, k, w), c, idx2, gv, 2);
    }
    for (int c2 = 0; c2 < D.ncirc; c2++) {
        real gv[2]; const real on = (real)((sh.circ_mask[i] >> c2) & 1u);
        const real c = on * circ_val(sh, c2, x[idx2[0]], x[idx2[1]], &gv[0], &gv[1]); gv[0] *= on; gv[1] *= on;
        row(D.o_circ(i, k, c2), c, idx2, gv, 2);
    }
    // Wall3D / Cylinder act on pz[i][1..3] (constraints_methods.jl:231-236,275)
    const int idx3[3] = {D.pz(i, 0), D.pz(i, 1), D.pz(i, 2)};
    const real q3[3] = {x[idx3[0]], x[idx3[1]], x[idx3[2]]};
    for (int w = 0; w < D.nwall3; w++) {
        real gv[3]; const real on = (real)((sh.wall3_mask[i] >> w) & 1u);
        const real c = on * wall3_val(sh, w, q3, gv); for (int a = 0; a < 3; a++) gv[a] *= on;
        row(D.o_wall3(i, k, w), c, idx3, gv, 3);
    }
    for (int c2 = 0; c2 < D.ncyl; c2++) {
        real gv[3]; const real on = (real)((sh.cyl_mask[i] >> c2) & 1u);
        const real c = on * cyl_val(sh, c2, q3, gv); for (int a = 0; a < 3; a++) gv[a] *= on;
        row(D.o_cyl(i, k, c2), c, idx3, gv, 3);
    }
}

// evaluate!(game_con, traj) (constraints_methods.jl:367-379)
void evaluate_con(const Shared& sh, Game& g, const std::vector<real>& z) {
    const Dims& D = sh.D;
    std::vector<real> u(D.m);
    if (sh.has_colavoid)
        for (int i = 0; i < D.p; i++) for (int j = 0; j < D.p; j++) if (j != i && sh.pair_on(i, j))
            for (int k = 1; k < D.N; k++) { real dl[3]; g.vals[con_col(D, D.pair(i, j), k)] = colavoid_val(sh, i, j, state(D, z, k), dl); }
    if (sh.has_ctl)
        for (int k = 0; k < D.N - 1; k++) { get_control(D, z, k, u.data()); for (int r = 0; r < 2 * D.m; r++) g.vals[con_ctl(D, k, r)] = ctl_val(sh, u.data(), r); }
    for (int i = 0; i < D.p; i++) for (int k = 1; k < D.N; k++) {
        const real* x = state(D, z, k); real gx, gy;
        if (D.has_sb) for (int r = 0; r < 2 * D.n; r++) g.vals[D.o_sb(i, k, r)] = sb_val(sh, i, x, r);
        for (int w = 0; w < D.nwall; w++) g.vals[D.o_wall(i, k, w)] = (real)((sh.wall_mask[i] >> w) & 1u) * wall_val(sh, w, x[D.px(i, 0)], x[D.px(i, 1)], &gx, &gy);
        for (int c = 0; c < D.ncirc; c++) g.vals[D.o_circ(i, k, c)] = (real)((sh.circ_mask[i] >> c) & 1u) * circ_val(sh, c, x[D.px(i, 0)], x[D.px(i, 1)], &gx, &gy);
        if (D.nwall3 + D.ncyl > 0) {
            const real q3[3] = {x[D.pz(i, 0)], x[D.pz(i, 1)], x[D.pz(i, 2)]}; real gv[3];
            for (int w = 0; w < D.nwall3; w++) g.vals[D.o_wall3(i, k, w)] = (real)((sh.wall3_mask[i] >> w) & 1u) * wall3_val(sh, w, q3, gv);
            for (int c = 0; c < D.ncyl; c++) g.vals[D.o_cyl(i, k, c)] = (real)((sh.cyl_mask[i] >> c) & 1u) * cyl_val(sh, c, q3, gv);
        }
    }
}
// Altro 0.3.0 / TrajOpt cost_expansion!(conval): a = (c >= 0) | (lambda > 0); I_mu = diag(a*mu);
// grad = C'(lambda + I_mu c); hess = C' I_mu C  [PINNED test/constraints/constraint_derivatives.jl:28-34]
inline real al_active_mu(real c, real lam, real mu) { return ((c >= 0) || (lam > 0)) ? mu : 0.0; }

// ---- residual! (global_quantities.jl:9-65) + regularize_residual! (:67-86) -------------------
void residual(const Shared& sh, Game& g, const std::vector<real>& z, real reg, const std::vector<real>* zref) {
    const Dims& D = sh.D;
    const int n = D.n, m = D.m, p = D.p, N = D.N, nd = n + m;
    std::vector<real>& res = g.res;
    res.assign(D.S, 0.0);                                                            // :18
    TL_VEC(real, q, n); TL_VEC(real, r, D.mi); TL_VEC(real, u, m); TL_VEC(real, J, n * nd); TL_VEC(real, xn, n); TL_VEC(real, uref, m);
    // Cost (:23-41).  stamp (opt,i,x,k) is invalid for the first knot (stamp.jl:203).
    for (int i = 0; i < p; i++) {
        for (int k = 1; k < N; k++) {
            cost_grad_x(sh, g, i, k, state(D, z, k), q.data());
            for (int a = 0; a < n; a++) res[D.vx(i, k - 1) + a] += q[a];
        }
        for (int k = 0; k < N - 1; k++) {
            get_control(D, z, k, u.data());
            cost_grad_u(sh, g, i, u.data(), r.data());
            for (int a = 0; a < D.mi; a++) res[D.vu(i, k) + a] += r[a];
        }
    }
    // Dynamics penalty (:43-54)
    for (int k = 0; k < N - 1; k++) {
        get_control(D, z, k, u.data());
        rk2_jacobian(D, state(D, z, k), u.data(), J.data());
        for (int i = 0; i < p; i++) {
            const real* lam = dual(D, z, i, k);
            if (k >= 1)                                                              // (opt,i,x,k) valid only for knots 2..N
                for (int c = 0; c < n; c++) { real s = 0; for (int rr = 0; rr < n; rr++) s += J[rr * nd + c] * lam[rr]; res[D.vx(i, k - 1) + c] += s; }
            for (int j = 0; j < D.mi; j++) { int c = n + D.pu(i, j); real s = 0; for (int rr = 0; rr < n; rr++) s += J[rr * nd + c] * lam[rr]; res[D.vu(i, k) + j] += s; }
            for (int c = 0; c < n; c++) res[D.vx(i, k) + c] += -lam[c];
        }
    }
    // Constraints: constraint_residual! (constraint_derivatives.jl:39-74)
    if (sh.has_colavoid) {
        for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) if (j != i && sh.pair_on(i, j)) {
            const int qd = D.pair(i, j);
            for (int k = 1; k < N; k++) {
                real dl[3];
                const real c = colavoid_val(sh, i, j, state(D, z, k), dl);
                const int ci = con_col(D, qd, k);
                g.vals[ci] = c;
                const real w = g.lam[ci] + al_active_mu(c, g.lam[ci], g.mu[ci]) * c;
                for (int a = 0; a < D.ca_dim; a++) {                                 // grad = C' w, C = [-2d' at px[i], +2d' at px[j]]
                    res[D.vx(i, k - 1) + D.pz(i, a)] += -2 * dl[a] * w;
                    res[D.vx(i, k - 1) + D.pz(j, a)] += 2 * dl[a] * w;
                }
            }
        }
    }
    if (D.con_len > D.col_len + D.ctl_len)
        for (int i = 0; i < p; i++) for (int k = 1; k < N; k++) ext_state_con(sh, g, z, i, k, &res[D.vx(i, k - 1)], [](int, int, real) {});
    if (sh.has_ctl) {
        for (int k = 0; k < N - 1; k++) {
            get_control(D, z, k, u.data());
            for (int i = 0; i < p; i++) for (int j = 0; j < D.mi; j++) {
                const int c = D.pu(i, j);
                real gsum = 0;
                for (int half = 0; half < 2; half++) {
                    const int row = half * m + c, ci = con_ctl(D, k, row);
                    const real cv = ctl_val(sh, u.data(), row);
                    g.vals[ci] = cv;
                    if (!r_isfinite(cv)) continue;                                // infinite bound: row absent in the reference
                    const real w = g.lam[ci] + al_active_mu(cv, g.lam[ci], g.mu[ci]) * cv;
                    gsum += (half == 0 ? 1.0 : -1.0) * w;
                }
                res[D.vu(i, k) + j] += gsum;
            }
        }
    }
    // Dynamics (:60-63, local_quantities.jl:5-14)
    for (int k = 0; k < N - 1; k++) {
        get_control(D, z, k, u.data());
        rk2(D, state(D, z, k), u.data(), xn.data());
        const real* x1 = state(D, z, k + 1);
        for (int a = 0; a < n; a++) res[D.vd(k) + a] += xn[a] - x1[a];
    }
    // regularize_residual! (:67-86)
    if (zref && reg != 0.0) {
        for (int k = 0; k < N - 1; k++) {
            const real* x = state(D, z, k + 1); const real* xr = state(D, *zref, k + 1);
            get_control(D, z, k, u.data()); get_control(D, *zref, k, uref.data());
            for (int i = 0; i < p; i++) {
                for (int a = 0; a < n; a++) res[D.vx(i, k) + a] += reg * (x[a] - xr[a]);
                for (int j = 0; j < D.mi; j++) { int c = D.pu(i, j); res[D.vu(i, k) + j] += reg * (u[c] - uref[c]); }
            }
        }
    }
}

real res_norm(const Shared& sh, const Game& g) {   // norm(core.res,1)/length(core.res)  (solver_methods.jl:76)
    real s = 0; for (real v : g.res) s += r_fabs(v); return s / sh.D.S;
}

// ---- residual_jacobian! (:109-174) + regularize_residual_jacobian! (:176-193) ----------------
// add(row_vertical, col_horizontal, value)
template <class Add>
void jacobian(const Shared& sh, Game& g, const std::vector<real>& z, real reg, Add add) {
    const Dims& D = sh.D;
    const int n = D.n, m = D.m, p = D.p, N = D.N, nd = n + m;
    TL_VEC(real, Qm, n * n); TL_VEC(real, u, m); TL_VEC(real, J, n * nd);
    // Cost (:128-145)
    for (int i = 0; i < p; i++) {
        for (int k = 1; k < N; k++) {
            cost_hess_x(sh, g, i, k, state(D, z, k), Qm.data());
            for (int a = 0; a < n; a++) for (int c = 0; c < n; c++) if (Qm[a * n + c] != 0.0) add(D.vx(i, k - 1) + a, D.hx(k - 1) + c, Qm[a * n + c]);
        }
        for (int k = 0; k < N - 1; k++)
            for (int j = 0; j < D.mi; j++) { int c = D.pu(i, j); add(D.vu(i, k) + j, D.hu(k, i) + j, D.dt * g.R[i * m + c]); }   // R[pu[i],pu[i]] diagonal
    }
    // Constraints: constraint_jacobian_residual! (constraint_derivatives.jl:1-36)
    if (sh.has_colavoid) {
        for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) if (j != i && sh.pair_on(i, j)) {
            const int qd = D.pair(i, j);
            for (int k = 1; k < N; k++) {
                real dl[3];
                const real c = colavoid_val(sh, i, j, state(D, z, k), dl);
                const int ci = con_col(D, qd, k);
                const real am = al_active_mu(c, g.lam[ci], g.mu[ci]);
                if (am == 0.0) continue;
                // hess = C' I_mu C with C = [-2d at px[i], 2d at px[j]]
                const int cd = D.ca_dim;
                int idx[6]; real cv[6];
                for (int a = 0; a < cd; a++) { idx[a] = D.pz(i, a); idx[cd + a] = D.pz(j, a); cv[a] = -2 * dl[a]; cv[cd + a] = 2 * dl[a]; }
                for (int a = 0; a < 2 * cd; a++) for (int c2 = 0; c2 < 2 * cd; c2++)
                    add(D.vx(i, k - 1) + idx[a], D.hx(k - 1) + idx[c2], am * cv[a] * cv[c2]);
            }
        }
    }
    if (D.con_len > D.col_len + D.ctl_len)
        for (int i = 0; i < p; i++) for (int k = 1; k < N; k++)
            ext_state_con(sh, g, z, i, k, nullptr, [&](int a, int c, real v) { add(D.vx(i, k - 1) + a, D.hx(k - 1) + c, v); });
    if (sh.has_ctl) {
        for (int k = 0; k < N - 1; k++) {
            get_control(D, z, k, u.data());
            for (int i = 0; i < p; i++) for (int j = 0; j < D.mi; j++) {
                const int c = D.pu(i, j);
                real h = 0;
                for (int half = 0; half < 2; half++) {
                    const int row = half * m + c, ci = con_ctl(D, k, row);
                    const real cv = ctl_val(sh, u.data(), row);
                    if (!r_isfinite(cv)) continue;
                    h += al_active_mu(cv, g.lam[ci], g.mu[ci]);
                }
                if (h != 0.0) add(D.vu(i, k) + j, D.hu(k, i) + j, h);
            }
        }
    }
    // Dynamics (:151-172)
    for (int k = 0; k < N - 1; k++) {
        get_control(D, z, k, u.data());
        rk2_jacobian(D, state(D, z, k), u.data(), J.data());
        if (k >= 1) for (int a = 0; a < n; a++) for (int c = 0; c < n; c++) if (J[a * nd + c] != 0.0) add(D.vd(k) + a, D.hx(k - 1) + c, J[a * nd + c]);
        for (int i = 0; i < p; i++) for (int j = 0; j < D.mi; j++) for (int a = 0; a < n; a++) {
            real v = J[a * nd + n + D.pu(i, j)]; if (v != 0.0) add(D.vd(k) + a, D.hu(k, i) + j, v);
        }
        for (int a = 0; a < n; a++) add(D.vd(k) + a, D.hx(k) + a, -1.0);
        for (int i = 0; i < p; i++) {
            if (k >= 1) for (int a = 0; a < n; a++) for (int c = 0; c < n; c++) if (J[a * nd + c] != 0.0) add(D.vx(i, k - 1) + c, D.hl(k, i) + a, J[a * nd + c]);
            for (int j = 0; j < D.mi; j++) for (int a = 0; a < n; a++) { real v = J[a * nd + n + D.pu(i, j)]; if (v != 0.0) add(D.vu(i, k) + j, D.hl(k, i) + a, v); }
            for (int a = 0; a < n; a++) add(D.vx(i, k) + a, D.hl(k, i) + a, -1.0);
        }
    }
    // regularize_residual_jacobian! (:176-193)
    if (reg != 0.0)
        for (int k = 0; k < N - 1; k++) for (int i = 0; i < p; i++) {
            for (int a = 0; a < n; a++) add(D.vx(i, k) + a, D.hx(k) + a, reg);
            for (int j = 0; j < D.mi; j++) add(D.vu(i, k) + j, D.hu(k, i) + j, reg);
        }
}

// ---- linear solve: general partial-pivot LU (stand-in for UMFPACK lu, solver_methods.jl:87) ---
// Rows are permuted to time-major order so the matrix is banded; the LU then does *full* partial
// pivoting over each column inside the band (LAPACK dgbtf2 algorithm), i.e. exactly the pivots a
// dense partial-pivot LU of the row-permuted matrix would take.
struct Banded {
    int S = 0, kl = 0, ku = 0, ld = 0;
    std::vector<real> ab;    // (2kl+ku+1) x S, column-major, LAPACK band storage
    std::vector<int> ipiv;
    void init(int S_, int kl_, int ku_) { S = S_; kl = kl_; ku = ku_; ld = 2 * kl + ku + 1; ab.assign((size_t)ld * S, 0.0); ipiv.assign(S, 0); }
    real& at(int r, int c) { return ab[(size_t)c * ld + (kl + ku + r - c)]; }
    // returns 0 ok, >0 singular at column
    int factor() {
        for (int j = 0; j < S; j++) {
            const int km = std::min(kl, S - 1 - j);
            int jp = 0; real best = r_fabs(at(j, j));
            for (int i = 1; i <= km; i++) { real v = r_fabs(at(j + i, j)); if (v > best) { best = v; jp = i; } }
            ipiv[j] = j + jp;
            if (best == 0.0 || !r_isfinite(best)) return j + 1;
            const int ju = std::min(j + ku + kl, S - 1);   // last column affected (U fill-in bound)
            if (jp != 0) for (int c = j; c <= ju; c++) std::swap(at(j, c), at(j + jp, c));
            const real inv = 1.0 / at(j, j);
            for (int i = 1; i <= km; i++) at(j + i, j) *= inv;
            for (int c = j + 1; c <= ju; c++) {
                const real v = at(j, c);
                if (v != 0.0) for (int i = 1; i <= km; i++) at(j + i, c) -= at(j + i, j) * v;
            }
        }
        return 0;
    }
    void solve(std::vector<real>& x) {
        for (int j = 0; j < S; j++) {
            const int km = std::min(kl, S - 1 - j);
            if (ipiv[j] != j) std::swap(x[j], x[ipiv[j]]);
            const real v = x[j];
            if (v != 0.0) for (int i = 1; i <= km; i++) x[j + i] -= at(j + i, j) * v;
        }
        for (int j = S - 1; j >= 0; j--) {
            x[j] /= at(j, j);
            const real v = x[j];
            const int i0 = std::max(0, j - ku - kl);
            if (v != 0.0) for (int i = i0; i < j; i++) x[i] -= at(i, j) * v;
        }
    }
};

// Row / column permutations that make the KKT matrix narrow-banded: per time step k the rows are ordered
// (dyn_k, opt_u_k, opt_x_{k+1}) and the columns (lambda_k, u_k, x_{k+1}).  A permutation changes neither the
// solution nor the fact that every column is searched in full for its pivot.
void build_perms(const Dims& D, std::vector<int>& rpos, std::vector<int>& cpos) {
    rpos.assign(D.S, 0); cpos.assign(D.S, 0);
    int off = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int a = 0; a < D.n; a++) rpos[D.vd(k) + a] = off++;
        for (int i = 0; i < D.p; i++) for (int j = 0; j < D.mi; j++) rpos[D.vu(i, k) + j] = off++;
        for (int i = 0; i < D.p; i++) for (int a = 0; a < D.n; a++) rpos[D.vx(i, k) + a] = off++;
    }
    off = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int i = 0; i < D.p; i++) for (int a = 0; a < D.n; a++) cpos[D.hl(k, i) + a] = off++;
        for (int i = 0; i < D.p; i++) for (int j = 0; j < D.mi; j++) cpos[D.hu(k, i) + j] = off++;
        for (int a = 0; a < D.n; a++) cpos[D.hx(k) + a] = off++;
    }
}

// Δtraj = - lu(jac) \ res ; set_traj!(core, Δpdtraj, Δtraj)  (solver_methods.jl:87-88)
int newton_direction(const Shared& sh, Game& g, real reg) {
    const Dims& D = sh.D;
    thread_local std::vector<int> rpos, cpos; build_perms(D, rpos, cpos);
    int kl = 0, ku = 0;
    jacobian(sh, g, g.z[0], reg, [&](int r, int c, real) { int dlt = rpos[r] - cpos[c]; kl = std::max(kl, dlt); ku = std::max(ku, -dlt); });
    thread_local Banded B; B.init(D.S, kl, ku);           // per-thread storage, zero-filled by init (no allocation after the first call)
    jacobian(sh, g, g.z[0], reg, [&](int r, int c, real v) { B.at(rpos[r], cpos[c]) += v; });
    TL_VEC(real, rhs, D.S);
    for (int r = 0; r < D.S; r++) rhs[rpos[r]] = g.res[r];
    if (B.factor() != 0) return ALG_STATUS_SINGULAR;
    B.solve(rhs);
    std::vector<real>& dz = g.z[2];
    for (int a = 0; a < D.n; a++) dz[a] = 0.0;
    for (int c = 0; c < D.S; c++) dz[D.n + c] = -rhs[cpos[c]];
    for (int c = 0; c < D.S; c++) if (!r_isfinite(dz[D.n + c])) return ALG_STATUS_SINGULAR;
    return ALG_STATUS_OK;
}

// update_traj!(target, source, alpha, Δ) (primal_dual_traj.jl:109-128): x_{2..N}, u_{1..N-1}, duals; x_1 untouched
void update_traj(const Shared& sh, std::vector<real>& tgt, const std::vector<real>& src, real alpha, const std::vector<real>& dz) {
    const Dims& D = sh.D;
    for (int c = 0; c < D.S; c++) tgt[D.n + c] = src[D.n + c] + alpha * dz[D.n + c];
}
// Δ_step (primal_dual_traj.jl:130-147)
real delta_step(const Shared& sh, const std::vector<real>& dz, real alpha) {
    const Dims& D = sh.D;
    real s = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int a = 0; a < D.n; a++) s += r_fabs(dz[D.n + D.hx(k) + a]);
        for (int a = 0; a < D.m; a++) s += r_fabs(dz[D.n + D.hu(k, 0) + a]);
    }
    s *= alpha;
    s /= (real)((D.N - 1) * (D.n + D.m));
    return s;
}

// record! (statistics.jl:44-57): residual_norm (recomputes residual!, unregularised) + four violations
// @elapsed of the reference (solver_methods.jl:40-42, :151-153): wall time of the previous (ibr_)inner_iteration of this game
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
alg_record record(const Shared& sh, Game& g, real delta, int outer) {
    const Dims& D = sh.D;
    alg_record rc{};
    rc.outer = outer; rc.delta = delta; rc.t_elap = g.t_elap;
    residual(sh, g, g.z[0], 0.0, nullptr);                       // residual_norm(prob, pdtraj) (global_quantities.jl:92-97)
    rc.res = res_norm(sh, g);
    // dynamics_violation (violations.jl:18-26): max_k max|dyn_k|
    real dv = 0; for (int k = 0; k < D.N - 1; k++) for (int a = 0; a < D.n; a++) dv = std::max<real>(dv, r_fabs(g.res[D.vd(k) + a]));
    rc.dyn_vio = dv;
    // control_violation / state_violation (violations.jl:57-67,101-114): max(0, max c) [PINNED test/struct/violations.jl:27-49]
    real cv = 0, sv = 0;
    if (sh.has_ctl) for (int k = 0; k < D.N - 1; k++) for (int r = 0; r < 2 * D.m; r++) cv = std::max<real>(cv, std::max<real>(0.0, g.vals[con_ctl(D, k, r)]));
    if (sh.has_colavoid) for (int q = 0; q < D.npair; q++) for (int k = 1; k < D.N; k++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[con_col(D, q, k)]));
    for (int e = D.col_len + D.ctl_len; e < D.con_len; e++) if (r_isfinite(g.vals[e])) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[e]));
    rc.con_vio = cv; rc.sta_vio = sv;
    // optimality_violation (violations.jl:153-168): max |res| over opt rows
    real ov = 0; const int nopt = D.p * (D.N - 1) * (D.n + D.mi);
    for (int r = 0; r < nopt; r++) ov = std::max<real>(ov, r_fabs(g.res[r]));
    rc.opt_vio = ov;
    return rc;
}

// line_search (solver_methods.jl:105-125)
void line_search(const Shared& sh, Game& g, real reg, real res_norm0, real* alpha_out, int* j_out) {
    const alg_options& o = sh.opt;
    int j = 1; real alpha = 1.0;
    while (j < o.ls_iter) {
        update_traj(sh, g.z[1], g.z[0], alpha, g.z[2]);
        residual(sh, g, g.z[1], o.regularize ? reg : 0.0, &g.z[0]);
        const real rt = res_norm(sh, g);
        if (rt <= (1.0 - alpha * o.beta) * res_norm0) break;
        alpha *= o.alpha_decrease; j += 1;
    }
    *alpha_out = alpha; *j_out = j;
}

// inner_iteration (solver_methods.jl:67-103)
alg_step_info inner_iteration(const Shared& sh, Game& g, int& LS_count, real& Delta, int k, int l) {
    const alg_options& o = sh.opt;
    alg_step_info info{};
    const real reg = o.reg_0 * ((real)l * (real)l * (real)l * (real)l);                  // solver_methods.jl:39
    residual(sh, g, g.z[0], o.regularize ? reg : 0.0, &g.z[0]);           // :73-74 (adds zero)
    alg_record rc = record(sh, g, Delta, k);                              // :75
    const real rn = res_norm(sh, g);                                    // :76
    info.rec = rc;
    Delta = 0.0;                                                          // :79
    if (!r_isfinite(rn)) { info.status = ALG_STATUS_NAN; info.control_flow = 1; g.hist.push_back(rc); g.st.records++; return info; }
    if (rc.opt_vio < o.eps_opt) { info.control_flow = 1; g.hist.push_back(rc); g.st.records++; return info; }   // :80-82
    int st = newton_direction(sh, g, reg);                                // :84-88
    if (st != ALG_STATUS_OK) { info.status = st; info.control_flow = 1; g.hist.push_back(rc); g.st.records++; return info; }
    g.st.newton_iters++;
    real alpha; int j;
    line_search(sh, g, reg, rn, &alpha, &j);                              // :91
    const bool failed = (j == o.ls_iter);                                 // :92
    if (failed) { LS_count += 1; g.st.ls_failures++; } else LS_count = 0; // :93
    update_traj(sh, g.z[0], g.z[0], alpha, g.z[2]);                       // :94
    Delta = delta_step(sh, g.z[2], alpha);                                // :95
    info.alpha = alpha; info.ls_j = j; info.ls_failed = failed; info.delta = Delta;
    rc.alpha = alpha; rc.ls_j = j; info.rec = rc;
    g.hist.push_back(rc); g.st.records++;
    if (Delta < o.delta_min) info.control_flow = 1;                       // :96-98
    return info;
}

// reset!(game_con) (constraints_methods.jl:295-327): lambda <- 0, mu <- mu0 = rho_0 [PINNED test/constraints/constraints_methods.jl:176-229]
void reset_con(const Shared& sh, Game& g) {
    std::fill(g.lam.begin(), g.lam.end(), 0.0);
    std::fill(g.mu.begin(), g.mu.end(), sh.opt.rho_0);
}
// evaluate! + dual_update! + penalty_update! (solver_methods.jl:57-61; constraints_methods.jl:349-365,421-440;
// Altro.penalty_update!: mu <- min(phi mu, mu_max) [PINNED test/constraints/constraints_methods.jl:180-193])
void dual_penalty_update(const Shared& sh, Game& g) {
    const Dims& D = sh.D; const alg_options& o = sh.opt;
    evaluate_con(sh, g, g.z[0]);
    if (sh.has_colavoid)
        for (int i = 0; i < D.p; i++) for (int j = 0; j < D.p; j++) if (j != i && sh.pair_on(i, j)) for (int k = 1; k < D.N; k++) {
            const int ci = con_col(D, D.pair(i, j), k);
            const real lb = g.lam[ci] + o.alphax_dual[i] * g.mu[ci] * g.vals[ci];
            g.lam[ci] = std::min<real>(std::max<real>(lb, 0.0), o.lambda_max);
        }
    if (sh.has_ctl)
        for (int k = 0; k < D.N - 1; k++) for (int r = 0; r < 2 * D.m; r++) {
            const int ci = con_ctl(D, k, r);
            if (!r_isfinite(g.vals[ci])) continue;
            const real lb = g.lam[ci] + o.alpha_dual * g.mu[ci] * g.vals[ci];
            g.lam[ci] = std::min<real>(std::max<real>(lb, 0.0), o.lambda_max);
        }
    for (int e = D.col_len + D.ctl_len; e < D.con_len; e++) {
        if (!r_isfinite(g.vals[e])) continue;
        const int e2 = e - D.col_len - D.ctl_len;
        int i, e3 = e2;
        if (e3 < D.sb_len) i = e3 / ((D.N - 1) * 2 * D.n);
        else if ((e3 -= D.sb_len) < D.wall_len) i = e3 / ((D.N - 1) * D.nwall);
        else if ((e3 -= D.wall_len) < D.circ_len) i = e3 / ((D.N - 1) * D.ncirc);
        else if ((e3 -= D.circ_len) < D.wall3_len) i = e3 / ((D.N - 1) * D.nwall3);
        else i = (e3 - D.wall3_len) / ((D.N - 1) * D.ncyl);
        const real lb = g.lam[e] + o.alphax_dual[i] * g.mu[e] * g.vals[e];
        g.lam[e] = std::min<real>(std::max<real>(lb, 0.0), o.lambda_max);
    }
    for (real& v : g.mu) v = std::min<real>(std::max<real>(v * o.rho_increase, 0.0), o.rho_max);
}

// rollout!(RK3, model, traj) (solver_methods.jl:17)
void rollout(const Shared& sh, std::vector<real>& z) {
    const Dims& D = sh.D;
    std::vector<real> u(D.m), xn(D.n);
    for (int k = 0; k < D.N - 1; k++) {
        get_control(D, z, k, u.data());
        rk3(D, state(D, z, k), u.data(), xn.data());
        std::copy(xn.begin(), xn.end(), state(D, z, k + 1));
    }
}

// init_traj! (primal_dual_traj.jl:29-44) with f = counter RNG.  Element counters: knot k (0-based),
// entry e of z_k=[x_k;u_k] (joint order) -> k*(n+m)+e ; dual (i,k,r) -> N*(n+m) + (i*(N-1)+k)*n + r.
// The terminal knot's control is drawn by the reference but never used.
void init_traj(const Shared& sh, Game& g, std::vector<real>& z, uint64_t game_id, bool use_shift, bool zero) {
    const Dims& D = sh.D; const alg_options& o = sh.opt;
    const int s = use_shift ? o.shift : (1 << 30);
    std::vector<real> old = z, u(D.m);
    for (int k = 0; k < D.N; k++) {
        const bool sh_ok = (k + s <= D.N - 1);
        // states are overwritten by the rollout for k >= 1, kept here for literalness
        if (k >= 1) for (int a = 0; a < D.n; a++)
            state(D, z, k)[a] = sh_ok ? state(D, old, k + s)[a] : (zero ? 0.0 : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)k * (D.n + D.m) + a));
        if (k < D.N - 1) {
            if (sh_ok && k + s < D.N - 1) get_control(D, old, k + s, u.data());
            else for (int a = 0; a < D.m; a++) u[a] = zero ? 0.0 : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)k * (D.n + D.m) + D.n + a);
            set_control(D, z, k, u.data());
        }
    }
    for (int i = 0; i < D.p; i++) for (int k = 0; k < D.N - 1; k++) for (int r = 0; r < D.n; r++)
        z[D.n + D.hl(k, i) + r] = (k + s <= D.N - 2) ? old[D.n + D.hl(k + s, i) + r]
                                 : (zero ? 0.0 : o.amplitude_init * counter_uniform(o.seed, game_id, (uint64_t)D.N * (D.n + D.m) + ((uint64_t)i * (D.N - 1) + k) * D.n + r));
    for (int a = 0; a < D.n; a++) z[a] = g.x0[a];                          // set_state!(pdtraj.pr[1], x0)
}

// newton_solve! (solver_methods.jl:5-65)
void newton_solve(const Shared& sh, Game& g, bool init, uint64_t game_id) {
    const alg_options& o = sh.opt;
    g.st = alg_game_stats{}; g.hist.clear(); g.t_elap = 0.0;                // reset!(prob.stats)
    if (init) init_traj(sh, g, g.z[0], game_id, true, false);             // :13
    else for (int a = 0; a < sh.D.n; a++) g.z[0][a] = g.x0[a];
    g.z[1] = g.z[0];                                                       // :14 (trial is overwritten before use; x_1 = x0 matters)
    std::fill(g.z[2].begin(), g.z[2].end(), 0.0);                          // :15
    rollout(sh, g.z[0]);                                                   // :17
    if (o.dual_reset) reset_con(sh, g);                                    // :25
    int out = 0; real Delta = 0.0;
    for (int k = 1; k <= o.outer_iter; k++) {                              // :30
        out = k;
        int LS_count = 0;                                                  // :35
        alg_record last{};
        bool any = false;
        for (int l = 1; l <= o.inner_iter; l++) {                          // :38
            const double t0 = now_s();
            alg_step_info info = inner_iteration(sh, g, LS_count, Delta, k, l);
            g.t_elap = now_s() - t0;
            last = info.rec; any = true;
            if (info.status != ALG_STATUS_OK) { g.st.status = info.status; break; }
            if (LS_count >= 1 || info.control_flow == 1) break;            // :43
        }
        if (g.st.status != ALG_STATUS_OK) break;
        const bool conv = any && last.dyn_vio < o.eps_dyn && last.con_vio < o.eps_con && last.sta_vio < o.eps_sta && last.opt_vio < o.eps_opt;
        if (conv) g.st.converged = 1;
        if (k == o.outer_iter || conv) break;                              // :49-55
        dual_penalty_update(sh, g);                                        // :57-61
    }
    alg_record fin = record(sh, g, Delta, out);                            // :63
    g.hist.push_back(fin); g.st.records++;
    g.st.outer_iters = out; g.st.last = fin;
}


// ==========================================================================================================
// Iterated best response (solver_methods.jl:133-289, global_quantities.jl:199-365, newton_core.jl:205-294)
// ==========================================================================================================
// vertical_mask / horizontal_mask (newton_core.jl:205-294, splitted_state = false): positions of the player's
// rows (opt_i x, opt_i u_i, dyn) and columns (x, u_i, lambda_i) in the masked system; -1 elsewhere.
void ibr_masks(const Dims& D, int i, std::vector<int>& rmask, std::vector<int>& cmask, int& Sm) {
    rmask.assign(D.S, -1); cmask.assign(D.S, -1);
    // time-major order inside the masked system (dyn, opt_u, opt_x | lambda, u, x) keeps it narrow-banded
    int off = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int a = 0; a < D.n; a++) rmask[D.vd(k) + a] = off++;
        for (int j = 0; j < D.mi; j++) rmask[D.vu(i, k) + j] = off++;
        for (int a = 0; a < D.n; a++) rmask[D.vx(i, k) + a] = off++;
    }
    Sm = off; off = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int a = 0; a < D.n; a++) cmask[D.hl(k, i) + a] = off++;
        for (int j = 0; j < D.mi; j++) cmask[D.hu(k, i) + j] = off++;
        for (int a = 0; a < D.n; a++) cmask[D.hx(k) + a] = off++;
    }
}
// norm(core.res[verti_mask], 1) / length(verti_mask)  (solver_methods.jl:241)
real ibr_res_norm(const Shared& sh, const Game& g, int i) {
    const Dims& D = sh.D; real s = 0;
    for (int k = 0; k < D.N - 1; k++) {
        for (int a = 0; a < D.n; a++) s += r_fabs(g.res[D.vx(i, k) + a]) + r_fabs(g.res[D.vd(k) + a]);
        for (int j = 0; j < D.mi; j++) s += r_fabs(g.res[D.vu(i, k) + j]);
    }
    return s / (real)((D.N - 1) * (2 * D.n + D.mi));
}
// ibr_residual! + regularize_ibr_residual! restricted to the mask == the full residual! on the player's rows; the
// proximal term only touches the player's rows (global_quantities.jl:262-280).  Rows outside the mask are not used.
void ibr_residual(const Shared& sh, Game& g, const std::vector<real>& z, int i, real reg, const std::vector<real>* zref) {
    residual(sh, g, z, 0.0, nullptr);
    if (zref && reg != 0.0) {
        const Dims& D = sh.D; std::vector<real> u(D.m), ur(D.m);
        for (int k = 0; k < D.N - 1; k++) {
            const real* x = state(D, z, k + 1); const real* xr = state(D, *zref, k + 1);
            get_control(D, z, k, u.data()); get_control(D, *zref, k, ur.data());
            for (int a = 0; a < D.n; a++) g.res[D.vx(i, k) + a] += reg * (x[a] - xr[a]);
            for (int j = 0; j < D.mi; j++) { int c = D.pu(i, j); g.res[D.vu(i, k) + j] += reg * (u[c] - ur[c]); }
        }
    }
}
// record!(stats, prob, model, game_con, pdtraj, t_elap, Δ, k, i) (statistics.jl:59-73): full residual norm, player-specific violations
alg_record ibr_record(const Shared& sh, Game& g, real delta, int outer, int i) {
    const Dims& D = sh.D;
    alg_record rc{}; rc.outer = outer; rc.delta = delta; rc.t_elap = g.t_elap;
    residual(sh, g, g.z[0], 0.0, nullptr);
    rc.res = res_norm(sh, g);
    real dv = 0;                                               // dynamics_violation(model, pdtraj, i): entries pz[i]
    for (int k = 0; k < D.N - 1; k++) for (int j = 0; j < D.ni; j++) dv = std::max<real>(dv, r_fabs(g.res[D.vd(k) + D.pz(i, j)]));
    rc.dyn_vio = dv;
    // control_violation(game_con, pdtraj, i) (violations.jl:69-82): c_max = max(0, maximum(v[pu[i]])) -- v is the vector of
    // FINITE bound rows [u - u_max; u_min - u][inds] and is indexed by the control indices pu[i] (literal restatement)
    real cv = 0;
    if (sh.has_ctl) for (int k = 0; k < D.N - 1; k++) {
        std::vector<real> fin;
        for (int r = 0; r < 2 * D.m; r++) if (r_isfinite(g.vals[con_ctl(D, k, r)])) fin.push_back(g.vals[con_ctl(D, k, r)]);
        real mx = -std::numeric_limits<double>::infinity();
        for (int j = 0; j < D.mi; j++) { int pos = D.pu(i, j); if (pos < (int)fin.size()) mx = std::max<real>(mx, fin[pos]); }
        cv = std::max<real>(cv, std::max<real>(0.0, mx));
    }
    rc.con_vio = cv;
    real sv = 0;                                               // state_violation(game_con, pdtraj, i): player i's convals
    if (sh.has_colavoid) for (int j = 0; j < D.p; j++) if (j != i && sh.pair_on(i, j)) for (int k = 1; k < D.N; k++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[con_col(D, D.pair(i, j), k)]));
    for (int k = 1; k < D.N; k++) {
        if (D.has_sb) for (int r = 0; r < 2 * D.n; r++) { real v = g.vals[D.o_sb(i, k, r)]; if (r_isfinite(v)) sv = std::max<real>(sv, std::max<real>(0.0, v)); }
        for (int w = 0; w < D.nwall; w++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[D.o_wall(i, k, w)]));
        for (int c = 0; c < D.ncirc; c++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[D.o_circ(i, k, c)]));
        for (int w = 0; w < D.nwall3; w++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[D.o_wall3(i, k, w)]));
        for (int c = 0; c < D.ncyl; c++) sv = std::max<real>(sv, std::max<real>(0.0, g.vals[D.o_cyl(i, k, c)]));
    }
    rc.sta_vio = sv;
    real ov = 0;                                               // optimality_violation(core, i)
    for (int k = 0; k < D.N - 1; k++) { for (int a = 0; a < D.n; a++) ov = std::max<real>(ov, r_fabs(g.res[D.vx(i, k) + a])); for (int j = 0; j < D.mi; j++) ov = std::max<real>(ov, r_fabs(g.res[D.vu(i, k) + j])); }
    rc.opt_vio = ov;
    g.max_delta = std::max<real>(g.max_delta, delta);
    return rc;
}
// Δtraj[horiz_mask] = - lu(jac[verti_mask, horiz_mask]) \ res[verti_mask]  (solver_methods.jl:249-251)
int ibr_direction(const Shared& sh, Game& g, int i, real reg) {
    const Dims& D = sh.D;
    thread_local std::vector<int> rm, cm; int Sm; ibr_masks(D, i, rm, cm, Sm);
    int kl = 0, ku = 0;
    jacobian(sh, g, g.z[0], reg, [&](int r, int c, real) { if (rm[r] >= 0 && cm[c] >= 0) { int dlt = rm[r] - cm[c]; kl = std::max(kl, dlt); ku = std::max(ku, -dlt); } });
    thread_local Banded B; B.init(Sm, kl, ku);
    jacobian(sh, g, g.z[0], reg, [&](int r, int c, real v) { if (rm[r] >= 0 && cm[c] >= 0) B.at(rm[r], cm[c]) += v; });
    TL_VEC(real, rhs, Sm);
    for (int r = 0; r < D.S; r++) if (rm[r] >= 0) rhs[rm[r]] = g.res[r];
    if (B.factor() != 0) return ALG_STATUS_SINGULAR;
    B.solve(rhs);
    std::vector<real>& dz = g.z[2];
    std::fill(dz.begin(), dz.end(), 0.0);
    for (int c = 0; c < D.S; c++) if (cm[c] >= 0) { dz[D.n + c] = -rhs[cm[c]]; if (!r_isfinite(dz[D.n + c])) return ALG_STATUS_SINGULAR; }
    return ALG_STATUS_OK;
}
// ibr_line_search (solver_methods.jl:270-289)
void ibr_line_search(const Shared& sh, Game& g, int i, real reg, real res_norm0, real* alpha_out, int* j_out) {
    const alg_options& o = sh.opt; int j = 1; real alpha = 1.0;
    while (j < o.ls_iter) {
        update_traj(sh, g.z[1], g.z[0], alpha, g.z[2]);
        ibr_residual(sh, g, g.z[1], i, o.regularize ? reg : 0.0, &g.z[0]);
        if (ibr_res_norm(sh, g, i) <= (1.0 - alpha * o.beta) * res_norm0) break;
        alpha *= o.alpha_decrease; j += 1;
    }
    *alpha_out = alpha; *j_out = j;
}
// ibr_inner_iteration (solver_methods.jl:230-268)
alg_step_info ibr_inner_iteration(const Shared& sh, Game& g, int& LS_count, real& Delta, int k, int l, int i) {
    const alg_options& o = sh.opt; alg_step_info info{};
    const real reg = o.reg_0 * ((real)l * (real)l * (real)l * (real)l);
    alg_record rc = ibr_record(sh, g, Delta, k, i);                        // :238-240 (leaves the full residual in core.res)
    const real rn = ibr_res_norm(sh, g, i);                              // :241
    info.rec = rc; Delta = 0.0;
    auto done = [&](int status, int flow) { info.status = status; info.control_flow = flow; g.hist.push_back(info.rec); g.st.records++; return info; };
    if (!r_isfinite(rn)) return done(ALG_STATUS_NAN, 1);
    if (rc.opt_vio < o.eps_opt) return done(ALG_STATUS_OK, 1);            // :245-247
    int st = ibr_direction(sh, g, i, reg);                                 // :249-252
    if (st != ALG_STATUS_OK) return done(st, 1);
    g.st.newton_iters++;
    real alpha; int j; ibr_line_search(sh, g, i, reg, rn, &alpha, &j);   // :255
    const bool failed = (j == o.ls_iter);
    if (failed) { LS_count += 1; g.st.ls_failures++; } else LS_count = 0;
    update_traj(sh, g.z[0], g.z[0], alpha, g.z[2]);                        // :258
    Delta = delta_step(sh, g.z[2], alpha);                                 // :259
    info.alpha = alpha; info.ls_j = j; info.ls_failed = failed; info.delta = Delta; info.rec.alpha = alpha; info.rec.ls_j = j;
    return done(ALG_STATUS_OK, Delta < o.delta_min ? 1 : 0);
}
// reset_duals!(pdtraj) (primal_dual_traj.jl:149-158)
void reset_traj_duals(const Dims& D, std::vector<real>& z) {
    for (int k = 0; k < D.N - 1; k++) for (int i = 0; i < D.p; i++) for (int a = 0; a < D.n; a++) z[D.n + D.hl(k, i) + a] *= 0.0;
}
// ibr_newton_solve!(prob, i) (solver_methods.jl:171-228)
void ibr_solve_player(const Shared& sh, Game& g, int i) {
    const alg_options& o = sh.opt;
    if (o.dual_reset) { reset_con(sh, g); reset_traj_duals(sh.D, g.z[0]); reset_traj_duals(sh.D, g.z[1]); }   // :181-185
    int out = 0; real Delta = 0.0; g.st.status = ALG_STATUS_OK; g.st.converged = 0;
    for (int k = 1; k <= o.outer_iter; k++) {
        out = k; int LS_count = 0; alg_record last{}; bool any = false;
        for (int l = 1; l <= o.inner_iter; l++) {
            const double t0 = now_s();
            alg_step_info info = ibr_inner_iteration(sh, g, LS_count, Delta, k, l, i);
            g.t_elap = now_s() - t0;
            last = info.rec; any = true;
            if (info.status != ALG_STATUS_OK) { g.st.status = info.status; break; }
            if (LS_count >= 1 || info.control_flow == 1) break;
        }
        if (g.st.status != ALG_STATUS_OK) break;
        const bool conv = any && last.dyn_vio < o.eps_dyn && last.con_vio < o.eps_con && last.sta_vio < o.eps_sta && last.opt_vio < o.eps_opt;
        if (conv) g.st.converged = 1;
        if (k == o.outer_iter || conv) break;
        dual_penalty_update(sh, g);
    }
    alg_record fin = ibr_record(sh, g, Delta, out, i);                      // :226
    g.hist.push_back(fin); g.st.records++; g.st.outer_iters = out; g.st.last = fin;
}
// ibr_newton_solve!(prob; ibr_opts) (solver_methods.jl:133-169)
void ibr_newton_solve(const Shared& sh, Game& g, bool init, uint64_t game_id, int ibr_iter, const int* ordering, real delta_min) {
    const Dims& D = sh.D;
    g.st = alg_game_stats{}; g.hist.clear(); g.max_delta = 0.0; g.t_elap = 0.0;   // reset!(prob.stats)
    if (init) init_traj(sh, g, g.z[0], game_id, true, false); else for (int a = 0; a < D.n; a++) g.z[0][a] = g.x0[a];
    g.z[1] = g.z[0]; std::fill(g.z[2].begin(), g.z[2].end(), 0.0);
    rollout(sh, g.z[0]);
    std::vector<char> change(D.p, 1);
    for (int q = 0; q < ibr_iter; q++) {
        for (int id = 0; id < D.p; id++) {
            const int i = ordering[id];
            ibr_solve_player(sh, g, i);
            change[i] = !(delta_min > g.max_delta);                         // :157 (maximum over the whole Statistics history)
            if (g.st.status != ALG_STATUS_OK) return;
        }
        bool any = false; for (char c : change) any |= (c != 0);
        if (!any) break;                                                    // :163
    }
}

} // namespace

// ------------------------------------------------------------------------------------------
// C ABI (same signatures as include/algames_hip.h with the orc_ prefix)
// ------------------------------------------------------------------------------------------
#define H ((Handle*)h)
// double <-> real at the ABI (the known-answer hooks pass plain double arrays)
static std::vector<real> rin(const double* p, size_t n) { return p ? std::vector<real>(p, p + n) : std::vector<real>(n, 0); }
struct ROut {
    double* dst; std::vector<real> v;
    ROut(double* d, size_t n) : dst(d), v(n, 0) {}
    real* ptr() { return dst ? v.data() : nullptr; }
    ~ROut() { if (dst) for (size_t i = 0; i < v.size(); i++) dst[i] = (double)v[i]; }
};
extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

void orc_default_options(alg_options* o) {   // options.jl:5-116
    std::memset(o, 0, sizeof(*o));
    o->amplitude_init = 1e-8; o->shift = 1 << 10; o->regularize = 1; o->reg_0 = 1e-3;
    o->alpha_decrease = 0.5; o->beta = 0.01; o->ls_iter = 25; o->dual_reset = 1; o->delta_min = 1e-9;
    o->rho_0 = 1.0; o->rho_increase = 10.0; o->rho_max = 1e7; o->lambda_max = 1e7; o->alpha_dual = 1.0;
    for (int i = 0; i < 10; i++) o->alphax_dual[i] = 1.0;
    o->eps_dyn = o->eps_sta = o->eps_con = o->eps_opt = 1e-3;
    o->outer_iter = 7; o->inner_iter = 20; o->seed = 100;
}

int orc_dims(const alg_desc* d, int32_t* n, int32_t* m, int32_t* mi, int32_t* S, int32_t* traj_len, int32_t* con_len) {
    Dims D; if (!d || !D.init(*d)) return fail(ALG_ERR_ARG, "orc_dims: unsupported descriptor");
    if (n) *n = D.n; if (m) *m = D.m; if (mi) *mi = D.mi; if (S) *S = D.S; if (traj_len) *traj_len = D.traj_len; if (con_len) *con_len = D.con_len;
    return ALG_OK;
}

int orc_create(const alg_desc* d, alg_handle** out) {
    if (!d || !out) return fail(ALG_ERR_ARG, "orc_create: null argument");
    Handle* hd = new Handle();
    if (!hd->sh.D.init(*d) || d->batch < 1) { delete hd; return fail(ALG_ERR_ARG, "orc_create: unsupported descriptor"); }
    if (hd->sh.D.n + hd->sh.D.m > MAXD) { delete hd; return fail(ALG_ERR_ARG, "orc_create: n+m too large for the dual-number width"); }
    orc_default_options(&hd->sh.opt);
    const Dims& D = hd->sh.D;
    hd->g.resize(d->batch);
    for (Game& g : hd->g) {
        g.Q.assign(D.p * D.n, 0.0); g.R.assign(D.p * D.m, 0.0); g.xf.assign(D.p * D.n, 0.0); g.uf.assign(D.p * D.m, 0.0);
        g.x0.assign(D.n, 0.0);
        for (auto& z : g.z) z.assign(D.traj_len, 0.0);
        g.lam.assign(D.con_len, 0.0); g.mu.assign(D.con_len, hd->sh.opt.rho_0); g.vals.assign(D.con_len, 0.0);
    }
    *out = (alg_handle*)hd;
    return ALG_OK;
}
void orc_destroy(alg_handle* h) { delete H; }

int orc_set_options(alg_handle* h, const alg_options* o) {
    if (!h || !o) return fail(ALG_ERR_ARG, "orc_set_options: null argument");
    if (o->ls_iter < 1 || o->outer_iter < 1 || o->inner_iter < 1) return fail(ALG_ERR_ARG, "orc_set_options: iteration counts must be >= 1");
    H->sh.opt = *o; return ALG_OK;
}
int orc_get_options(alg_handle* h, alg_options* o) { *o = H->sh.opt; return ALG_OK; }
int orc_set_stream(alg_handle*, void*) { return ALG_OK; }

int orc_set_x0(alg_handle* h, const double* x0) {
    const Dims& D = H->sh.D;
    for (size_t gi = 0; gi < H->g.size(); gi++) { Game& g = H->g[gi]; std::copy(x0 + gi * D.n, x0 + (gi + 1) * D.n, g.x0.begin()); for (auto& z : g.z) std::copy(g.x0.begin(), g.x0.end(), z.begin()); std::fill(g.z[2].begin(), g.z[2].begin() + D.n, 0.0); }
    H->x0_set = true; return ALG_OK;
}
// GameObjective constructor (objective.jl:12-35): expand_vector onto pz[i] / pu[i]
int orc_set_lqr(alg_handle* h, const double* Qd, const double* Rd, const double* xf, const double* uf, int32_t per_game) {
    const Dims& D = H->sh.D;
    for (size_t gi = 0; gi < H->g.size(); gi++) {
        Game& g = H->g[gi];
        const size_t ox = per_game ? gi * D.p * D.ni : 0, ou = per_game ? gi * D.p * D.mi : 0;
        std::fill(g.Q.begin(), g.Q.end(), 0.0); std::fill(g.R.begin(), g.R.end(), 0.0);
        std::fill(g.xf.begin(), g.xf.end(), 0.0); std::fill(g.uf.begin(), g.uf.end(), 0.0);
        for (int i = 0; i < D.p; i++) {
            for (int j = 0; j < D.ni; j++) { g.Q[i * D.n + D.pz(i, j)] = Qd[ox + i * D.ni + j]; g.xf[i * D.n + D.pz(i, j)] = xf[ox + i * D.ni + j]; }
            for (int j = 0; j < D.mi; j++) { g.R[i * D.m + D.pu(i, j)] = Rd[ou + i * D.mi + j]; g.uf[i * D.m + D.pu(i, j)] = uf[ou + i * D.mi + j]; }
        }
    }
    H->lqr_set = true; return ALG_OK;
}
int orc_add_collision_cost(alg_handle* h, const double* radius, const double* mu) {
    Shared& s = H->sh;
    if (!radius || !mu) { s.has_colcost = false; return ALG_OK; }
    s.cc_radius.assign(radius, radius + s.D.p); s.cc_mu.assign(mu, mu + s.D.p); s.has_colcost = true; return ALG_OK;
}
// vector form: add_collision_avoidance!(game_con, radius::Vector) loops over i, j != i with r_i + r_j (constraints_methods.jl:21-33)
static void set_all_pairs(Shared& s, const double* radius) {
    const int p = s.D.p;
    s.ca_radius.assign(radius, radius + p);
    s.pair_r.assign((size_t)p * p, -1.0);
    for (int i = 0; i < p; i++) for (int j = 0; j < p; j++) if (j != i) s.pair_r[(size_t)i * p + j] = radius[i] + radius[j];
}
int orc_add_collision_avoidance(alg_handle* h, const double* radius) {
    Shared& s = H->sh;
    if (!radius) { s.has_colavoid = false; return ALG_OK; }
    set_all_pairs(s, radius); s.has_colavoid = true; s.D.ca_dim = 2; return ALG_OK;
}
// add_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:5-19): one CollisionConstraint(n, px[i], px[j], radius)
static int add_pair(Handle* hd, const char* who, int i, int j, double radius, int dim) {
    Shared& s = hd->sh; const int p = s.D.p;
    if (i < 0 || j < 0 || i >= p || j >= p || i == j) return fail(ALG_ERR_ARG, std::string(who) + ": players i != j in 0..p-1");
    if (!(radius > 0.0)) return fail(ALG_ERR_ARG, std::string(who) + ": radius must be positive");
    if (s.has_colavoid && s.D.ca_dim != dim) return fail(ALG_ERR_ARG, std::string(who) + ": planar and spherical collision avoidance cannot be mixed on one handle");
    if (!s.has_colavoid) s.pair_r.assign((size_t)p * p, -1.0);
    if (s.pair_on(i, j)) return fail(ALG_ERR_ARG, std::string(who) + ": this ordered pair already carries a collision-avoidance constraint (one per pair)");
    s.pair_r[(size_t)i * p + j] = radius; s.has_colavoid = true; s.D.ca_dim = dim;
    return ALG_OK;
}
int orc_add_collision_avoidance_pair(alg_handle* h, int32_t i, int32_t j, double radius) { return add_pair(H, "orc_add_collision_avoidance_pair", i, j, radius, 2); }
int orc_add_control_bound(alg_handle* h, const double* umax, const double* umin) {
    Shared& s = H->sh;
    if (!umax || !umin) { s.has_ctl = false; return ALG_OK; }
    for (int i = 0; i < s.D.m; i++) if (!(umax[i] >= umin[i])) return fail(ALG_ERR_ARG, "Upper bounds must be greater than or equal to lower bounds");  // control_bound_constraint.jl:69-75
    s.umax.assign(umax, umax + s.D.m); s.umin.assign(umin, umin + s.D.m); s.has_ctl = true; return ALG_OK;
}
static void orc_resize_con(Handle* hd) {
    hd->sh.D.recount();
    for (Game& g : hd->g) { g.lam.assign(hd->sh.D.con_len, 0.0); g.mu.assign(hd->sh.D.con_len, hd->sh.opt.rho_0); g.vals.assign(hd->sh.D.con_len, 0.0); }
}
int orc_set_quadrotor(alg_handle* h, double mass) {
    if (H->sh.D.model != ALG_MODEL_QUADROTOR) return fail(ALG_ERR_ARG, "orc_set_quadrotor: not a quadrotor model");
    if (!(mass > 0)) return fail(ALG_ERR_ARG, "orc_set_quadrotor: mass must be positive");
    H->sh.D.qmass = mass; return ALG_OK;
}
int orc_set_bicycle(alg_handle* h, double lf, double lr) {
    if (H->sh.D.model != ALG_MODEL_BICYCLE) return fail(ALG_ERR_ARG, "orc_set_bicycle: not a bicycle model");
    if (!(lr > 0) || !(lf >= 0)) return fail(ALG_ERR_ARG, "orc_set_bicycle: bad lengths");
    H->sh.D.lf = lf; H->sh.D.lr = lr; return ALG_OK;
}
int orc_add_state_bound(alg_handle* h, int32_t player, const double* xmax, const double* xmin) {
    Shared& s = H->sh; const int n = s.D.n, p = s.D.p;
    if (player < 0 || player >= p || !xmax || !xmin) return fail(ALG_ERR_ARG, "orc_add_state_bound: bad argument");
    for (int a = 0; a < n; a++) if (!(xmax[a] >= xmin[a])) return fail(ALG_ERR_ARG, "Upper bounds must be greater than or equal to lower bounds");
    if (!s.D.has_sb) { s.sbmax.assign(p * n, std::numeric_limits<double>::infinity()); s.sbmin.assign(p * n, -std::numeric_limits<double>::infinity()); }
    for (int a = 0; a < n; a++) { s.sbmax[player * n + a] = xmax[a]; s.sbmin[player * n + a] = xmin[a]; }
    s.D.has_sb = 1; orc_resize_con(H); return ALG_OK;
}
int orc_add_wall_constraint(alg_handle* h, int32_t nw, const double* x1, const double* y1, const double* x2, const double* y2, const double* xv, const double* yv) {
    Shared& s = H->sh;
    if (nw < 0 || nw > ALG_MAX_WALLS) return fail(ALG_ERR_ARG, "orc_add_wall_constraint: too many walls");
    s.wx1.assign(x1, x1 + nw); s.wy1.assign(y1, y1 + nw); s.wx2.assign(x2, x2 + nw); s.wy2.assign(y2, y2 + nw); s.wxv.assign(xv, xv + nw); s.wyv.assign(yv, yv + nw);
    s.D.nwall = nw; for (int i = 0; i < 10; i++) s.wall_mask[i] = 0xffffffffu; orc_resize_con(H); return ALG_OK;
}
int orc_add_wall_constraint_player(alg_handle* h, int32_t player, int32_t nw, const double* x1, const double* y1, const double* x2, const double* y2, const double* xv, const double* yv) {
    Shared& s = H->sh;
    if (player < 0 || player >= s.D.p || nw < 0) return fail(ALG_ERR_ARG, "orc_add_wall_constraint_player: bad argument");
    if (s.D.nwall == 0) for (int i = 0; i < 10; i++) s.wall_mask[i] = 0u;
    else for (int i = 0; i < 10; i++) if (s.wall_mask[i] == 0xffffffffu) s.wall_mask[i] = (1u << s.D.nwall) - 1u;   // after an all-player set: explicit bits (as the HIP library)
    std::vector<real>* tab[6] = {&s.wx1, &s.wy1, &s.wx2, &s.wy2, &s.wxv, &s.wyv}; const double* src[6] = {x1, y1, x2, y2, xv, yv};
    for (int f = 0; f < 6; f++) tab[f]->resize(s.D.nwall);
    for (int w = 0; w < nw; w++) {
        int at = -1;
        for (int e = 0; e < s.D.nwall && at < 0; e++) { bool same = true; for (int f = 0; f < 6; f++) same &= ((*tab[f])[e] == src[f][w]); if (same) at = e; }
        if (at < 0) {
            if (s.D.nwall >= ALG_MAX_WALLS) return fail(ALG_ERR_ARG, "orc_add_wall_constraint_player: too many walls");
            at = s.D.nwall++;
            for (int f = 0; f < 6; f++) tab[f]->push_back(src[f][w]);
        }
        s.wall_mask[player] |= 1u << at;
    }
    orc_resize_con(H); return ALG_OK;
}
int orc_add_circle_constraint(alg_handle* h, int32_t nc, const double* xc, const double* yc, const double* rad) {
    Shared& s = H->sh;
    if (nc < 0 || nc > ALG_MAX_CIRCLES) return fail(ALG_ERR_ARG, "orc_add_circle_constraint: too many circles");
    s.cxc.assign(xc, xc + nc); s.cyc.assign(yc, yc + nc); s.crad.assign(rad, rad + nc);
    s.D.ncirc = nc; for (int i = 0; i < 10; i++) s.circ_mask[i] = 0xffffffffu; orc_resize_con(H); return ALG_OK;
}
int orc_add_circle_constraint_player(alg_handle* h, int32_t player, int32_t nc, const double* xc, const double* yc, const double* rad) {
    Shared& s = H->sh;
    if (player < 0 || player >= s.D.p || nc < 0) return fail(ALG_ERR_ARG, "orc_add_circle_constraint_player: bad argument");
    if (s.D.ncirc == 0) for (int i = 0; i < 10; i++) s.circ_mask[i] = 0u;
    else for (int i = 0; i < 10; i++) if (s.circ_mask[i] == 0xffffffffu) s.circ_mask[i] = (1u << s.D.ncirc) - 1u;
    std::vector<real>* tab[3] = {&s.cxc, &s.cyc, &s.crad}; const double* src[3] = {xc, yc, rad};
    for (int f = 0; f < 3; f++) tab[f]->resize(s.D.ncirc);
    for (int c = 0; c < nc; c++) {
        int at = -1;
        for (int e = 0; e < s.D.ncirc && at < 0; e++) { bool same = true; for (int f = 0; f < 3; f++) same &= ((*tab[f])[e] == src[f][c]); if (same) at = e; }
        if (at < 0) {
            if (s.D.ncirc >= ALG_MAX_CIRCLES) return fail(ALG_ERR_ARG, "orc_add_circle_constraint_player: too many circles");
            at = s.D.ncirc++;
            for (int f = 0; f < 3; f++) tab[f]->push_back(src[f][c]);
        }
        s.circ_mask[player] |= 1u << at;
    }
    orc_resize_con(H); return ALG_OK;
}
// 3-D ingredients: the reference indexes pz[i][1:3]; meaningful (positions) for DoubleIntegratorGame(d = 3) only
static int need_3d(Handle* hd, const char* who) {
    if (!(hd->sh.D.model == ALG_MODEL_QUADROTOR || (hd->sh.D.model == ALG_MODEL_DOUBLE_INTEGRATOR && hd->sh.D.d == 3))) return fail(ALG_ERR_ARG, std::string(who) + ": needs a model with three position dimensions (DoubleIntegrator d = 3, Quadrotor)");
    return ALG_OK;
}
int orc_add_spherical_collision_avoidance(alg_handle* h, const double* radius) {
    Shared& s = H->sh;
    if (!radius) { s.has_colavoid = false; s.D.ca_dim = 2; return ALG_OK; }
    if (int rc = need_3d(H, "orc_add_spherical_collision_avoidance")) return rc;
    // (like every adder of the extended set: the multipliers are re-created, lambda = 0, mu = rho_0 -- include/algames_hip.h)
    set_all_pairs(s, radius); s.has_colavoid = true; s.D.ca_dim = 3; orc_resize_con(H); return ALG_OK;
}
// add_spherical_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:45-64): CollisionConstraint(n, pz[i][1:3], pz[j][1:3], radius)
int orc_add_spherical_collision_avoidance_pair(alg_handle* h, int32_t i, int32_t j, double radius) {
    if (int rc = need_3d(H, "orc_add_spherical_collision_avoidance_pair")) return rc;
    if (int rc = add_pair(H, "orc_add_spherical_collision_avoidance_pair", i, j, radius, 3)) return rc;
    orc_resize_con(H); return ALG_OK;
}
int orc_add_wall3d_constraint(alg_handle* h, int32_t nw, const double* p1, const double* p2, const double* p3, const double* v) {
    Shared& s = H->sh;
    if (nw < 0 || nw > ALG_MAX_WALLS) return fail(ALG_ERR_ARG, "orc_add_wall3d_constraint: too many walls");
    if (int rc = need_3d(H, "orc_add_wall3d_constraint")) return rc;
    s.w3p1.assign(p1, p1 + 3 * nw); s.w3p2.assign(p2, p2 + 3 * nw); s.w3p3.assign(p3, p3 + 3 * nw); s.w3v.assign(v, v + 3 * nw);
    s.D.nwall3 = nw; for (int i = 0; i < 10; i++) s.wall3_mask[i] = 0xffffffffu; orc_resize_con(H); return ALG_OK;
}
// add_wall_constraint!(game_con, i, walls::Vector{Wall3D}) (constraints_methods.jl:208-247): the Wall3DConstraint joins state_conlist[i] only
int orc_add_wall3d_constraint_player(alg_handle* h, int32_t player, int32_t nw, const double* p1, const double* p2, const double* p3, const double* v) {
    Shared& s = H->sh;
    if (player < 0 || player >= s.D.p || nw < 0) return fail(ALG_ERR_ARG, "orc_add_wall3d_constraint_player: bad argument");
    if (int rc = need_3d(H, "orc_add_wall3d_constraint_player")) return rc;
    if (s.D.nwall3 == 0) for (int i = 0; i < 10; i++) s.wall3_mask[i] = 0u;
    else for (int i = 0; i < 10; i++) if (s.wall3_mask[i] == 0xffffffffu) s.wall3_mask[i] = (1u << s.D.nwall3) - 1u;
    std::vector<real>* tab[4] = {&s.w3p1, &s.w3p2, &s.w3p3, &s.w3v}; const double* src[4] = {p1, p2, p3, v};
    for (int f = 0; f < 4; f++) tab[f]->resize(3 * s.D.nwall3);
    for (int w = 0; w < nw; w++) {
        int at = -1;
        for (int e = 0; e < s.D.nwall3 && at < 0; e++) { bool same = true; for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) same &= ((*tab[f])[3 * e + a] == src[f][3 * w + a]); if (same) at = e; }
        if (at < 0) {
            if (s.D.nwall3 >= ALG_MAX_WALLS) return fail(ALG_ERR_ARG, "orc_add_wall3d_constraint_player: too many walls");
            at = s.D.nwall3++;
            for (int f = 0; f < 4; f++) for (int a = 0; a < 3; a++) tab[f]->push_back(src[f][3 * w + a]);
        }
        s.wall3_mask[player] |= 1u << at;
    }
    orc_resize_con(H); return ALG_OK;
}
int orc_add_cylinder_constraint(alg_handle* h, int32_t nc, const double* p, const int32_t* axis, const double* l, const double* r) {
    Shared& s = H->sh;
    if (nc < 0 || nc > ALG_MAX_CIRCLES) return fail(ALG_ERR_ARG, "orc_add_cylinder_constraint: too many cylinders");
    if (int rc = need_3d(H, "orc_add_cylinder_constraint")) return rc;
    for (int c = 0; c < nc; c++) if (axis[c] < 0 || axis[c] > 2) return fail(ALG_ERR_ARG, "orc_add_cylinder_constraint: axis must be 0 (:x), 1 (:y) or 2 (:z)");
    s.cyp.assign(p, p + 3 * nc); s.cyax.assign(axis, axis + nc); s.cyl.assign(l, l + nc); s.cyr.assign(r, r + nc);
    s.D.ncyl = nc; for (int i = 0; i < 10; i++) s.cyl_mask[i] = 0xffffffffu; orc_resize_con(H); return ALG_OK;
}
// add_wall_constraint!(game_con, i, walls::Vector{CylinderWall}) (constraints_methods.jl:256-299): the CylinderConstraint joins state_conlist[i] only
int orc_add_cylinder_constraint_player(alg_handle* h, int32_t player, int32_t nc, const double* p, const int32_t* axis, const double* l, const double* r) {
    Shared& s = H->sh;
    if (player < 0 || player >= s.D.p || nc < 0) return fail(ALG_ERR_ARG, "orc_add_cylinder_constraint_player: bad argument");
    if (int rc = need_3d(H, "orc_add_cylinder_constraint_player")) return rc;
    for (int c = 0; c < nc; c++) if (axis[c] < 0 || axis[c] > 2) return fail(ALG_ERR_ARG, "orc_add_cylinder_constraint_player: axis must be 0 (:x), 1 (:y) or 2 (:z)");
    if (s.D.ncyl == 0) for (int i = 0; i < 10; i++) s.cyl_mask[i] = 0u;
    else for (int i = 0; i < 10; i++) if (s.cyl_mask[i] == 0xffffffffu) s.cyl_mask[i] = (1u << s.D.ncyl) - 1u;
    s.cyp.resize(3 * s.D.ncyl); s.cyax.resize(s.D.ncyl); s.cyl.resize(s.D.ncyl); s.cyr.resize(s.D.ncyl);
    for (int c = 0; c < nc; c++) {
        int at = -1;
        for (int e = 0; e < s.D.ncyl && at < 0; e++) {
            bool same = s.cyax[e] == axis[c] && s.cyl[e] == l[c] && s.cyr[e] == r[c];
            for (int a = 0; a < 3; a++) same &= (s.cyp[3 * e + a] == p[3 * c + a]);
            if (same) at = e;
        }
        if (at < 0) {
            if (s.D.ncyl >= ALG_MAX_CIRCLES) return fail(ALG_ERR_ARG, "orc_add_cylinder_constraint_player: too many cylinders");
            at = s.D.ncyl++;
            for (int a = 0; a < 3; a++) s.cyp.push_back(p[3 * c + a]);
            s.cyax.push_back(axis[c]); s.cyl.push_back(l[c]); s.cyr.push_back(r[c]);
        }
        s.cyl_mask[player] |= 1u << at;
    }
    orc_resize_con(H); return ALG_OK;
}
int orc_get_con_len(alg_handle* h, int32_t* n) { *n = H->sh.D.con_len; return ALG_OK; }
int orc_set_traj(alg_handle* h, int32_t which, const double* z) {
    if (which < 0 || which > 2) return fail(ALG_ERR_ARG, "bad traj selector");
    const int L = H->sh.D.traj_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) std::copy(z + gi * L, z + (gi + 1) * L, H->g[gi].z[which].begin());
    return ALG_OK;
}
int orc_get_traj(alg_handle* h, int32_t which, double* z) {
    if (which < 0 || which > 2) return fail(ALG_ERR_ARG, "bad traj selector");
    const int L = H->sh.D.traj_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) std::copy(H->g[gi].z[which].begin(), H->g[gi].z[which].end(), z + gi * L);
    return ALG_OK;
}
int orc_set_con_duals(alg_handle* h, const double* lam, const double* mu) {
    const int L = H->sh.D.con_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) { if (lam) std::copy(lam + gi * L, lam + (gi + 1) * L, H->g[gi].lam.begin()); if (mu) std::copy(mu + gi * L, mu + (gi + 1) * L, H->g[gi].mu.begin()); }
    return ALG_OK;
}
int orc_get_con_duals(alg_handle* h, double* lam, double* mu) {
    const int L = H->sh.D.con_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) { if (lam) std::copy(H->g[gi].lam.begin(), H->g[gi].lam.end(), lam + gi * L); if (mu) std::copy(H->g[gi].mu.begin(), H->g[gi].mu.end(), mu + gi * L); }
    return ALG_OK;
}
int orc_init_traj(alg_handle* h, int64_t game_id0, int32_t use_shift) {
    const int B = (int)H->g.size();
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { Game& g = H->g[gi]; init_traj(H->sh, g, g.z[0], (uint64_t)(game_id0 + gi), use_shift != 0, false); g.z[1] = g.z[0]; rollout(H->sh, g.z[0]); }
    return ALG_OK;
}
int orc_rollout(alg_handle* h, int32_t which) {
    if (which < 0 || which > 1) return fail(ALG_ERR_ARG, "bad traj selector");
    for (Game& g : H->g) rollout(H->sh, g.z[which]);
    return ALG_OK;
}
int orc_residual(alg_handle* h, int32_t which, double reg, double* res, double* rn) {
    if (which < 0 || which > 1) return fail(ALG_ERR_ARG, "bad traj selector");
    const int B = (int)H->g.size(), S = H->sh.D.S;
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) {
        Game& g = H->g[gi];
        residual(H->sh, g, g.z[which], reg, &g.z[0]);
        if (res) std::copy(g.res.begin(), g.res.end(), res + (size_t)gi * S);
        if (rn) rn[gi] = res_norm(H->sh, g);
    }
    return ALG_OK;
}
int orc_residual_jacobian_games(alg_handle* h, double reg, int32_t first_game, int32_t n_games, double* jac) {
    const int B = (int)H->g.size(); const size_t S = H->sh.D.S;
    if (first_game < 0 || n_games < 1 || (long long)first_game + n_games > B) return fail(ALG_ERR_ARG, "orc_residual_jacobian_games: game range outside the batch");
    for (int gi = first_game; gi < first_game + n_games; gi++) {
        double* Jm = jac + (size_t)(gi - first_game) * S * S; std::fill(Jm, Jm + S * S, 0.0);
        jacobian(H->sh, H->g[gi], H->g[gi].z[0], reg, [&](int r, int c, double v) { Jm[(size_t)c * S + r] += v; });
    }
    return ALG_OK;
}
int orc_residual_jacobian(alg_handle* h, double reg, double* jac) { return orc_residual_jacobian_games(h, reg, 0, (int)H->g.size(), jac); }
int orc_release_scratch(alg_handle*) { return ALG_OK; }
// Per-knot violation profiles at pdtraj: the .vio vectors of dynamics_violation (violations.jl:18-26), control_violation (:57-67),
// state_violation (:101-114) and optimality_violation (:153-168).  dyn, con: B x (N-1); sta, opt: B x N (knot 1 first).
int orc_get_violation_profile(alg_handle* h, double* dyn, double* con, double* sta, double* opt) {
    const Shared& sh = H->sh; const Dims& D = sh.D; const int N = D.N, K = N - 1;
    for (size_t gi = 0; gi < H->g.size(); gi++) {
        Game& g = H->g[gi];
        residual(sh, g, g.z[0], 0.0, nullptr);
        auto pos = [](real c) { return (r_isfinite(c) && c > 0.0) ? c : (real)0.0; };
        for (int j = 0; j < N; j++) {
            real vo = 0, vs = 0;
            if (j < K) {
                real vd = 0, vc = 0;
                for (int a = 0; a < D.n; a++) vd = std::max<real>(vd, r_fabs(g.res[D.vd(j) + a]));
                if (sh.has_ctl) for (int r = 0; r < 2 * D.m; r++) vc = std::max<real>(vc, pos(g.vals[con_ctl(D, j, r)]));
                if (dyn) dyn[gi * K + j] = (double)vd;
                if (con) con[gi * K + j] = (double)vc;
                for (int i = 0; i < D.p; i++) for (int c = 0; c < D.mi; c++) vo = std::max<real>(vo, r_fabs(g.res[D.vu(i, j) + c]));
            }
            if (j >= 1) {
                const int k = j - 1;
                for (int i = 0; i < D.p; i++) for (int a = 0; a < D.n; a++) vo = std::max<real>(vo, r_fabs(g.res[D.vx(i, k) + a]));
                if (sh.has_colavoid) for (int q = 0; q < D.npair; q++) vs = std::max<real>(vs, pos(g.vals[con_col(D, q, k + 1)]));
                for (int i = 0; i < D.p; i++) {
                    if (D.has_sb) for (int r = 0; r < 2 * D.n; r++) vs = std::max<real>(vs, pos(g.vals[D.o_sb(i, k + 1, r)]));
                    for (int w = 0; w < D.nwall; w++) vs = std::max<real>(vs, pos(g.vals[D.o_wall(i, k + 1, w)]));
                    for (int c = 0; c < D.ncirc; c++) vs = std::max<real>(vs, pos(g.vals[D.o_circ(i, k + 1, c)]));
                    for (int w = 0; w < D.nwall3; w++) vs = std::max<real>(vs, pos(g.vals[D.o_wall3(i, k + 1, w)]));
                    for (int c = 0; c < D.ncyl; c++) vs = std::max<real>(vs, pos(g.vals[D.o_cyl(i, k + 1, c)]));
                }
            }
            if (sta) sta[gi * N + j] = (double)vs;
            if (opt) opt[gi * N + j] = (double)vo;
        }
    }
    return ALG_OK;
}
int orc_newton_direction(alg_handle* h, double reg, double* delta, int32_t* status) {
    const int B = (int)H->g.size(), S = H->sh.D.S, n = H->sh.D.n;
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) {
        Game& g = H->g[gi];
        residual(H->sh, g, g.z[0], 0.0, nullptr);
        int st = newton_direction(H->sh, g, reg);
        if (status) status[gi] = st;
        if (delta) std::copy(g.z[2].begin() + n, g.z[2].end(), delta + (size_t)gi * S);
    }
    return ALG_OK;
}
int orc_line_search(alg_handle* h, double reg, const double* rn, double* alpha, int32_t* j) {
    const int B = (int)H->g.size();
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { int jj; real a; line_search(H->sh, H->g[gi], reg, rn[gi], &a, &jj); alpha[gi] = (double)a; j[gi] = jj; }
    return ALG_OK;
}
int orc_update_traj(alg_handle* h, int32_t target, int32_t source, const double* alpha) {
    if (target < 0 || target > 1 || source < 0 || source > 1) return fail(ALG_ERR_ARG, "bad traj selector");
    for (size_t gi = 0; gi < H->g.size(); gi++) { Game& g = H->g[gi]; update_traj(H->sh, g.z[target], g.z[source], alpha[gi], g.z[2]); }
    return ALG_OK;
}
int orc_record_stats(alg_handle* h, alg_record* rec) {
    for (size_t gi = 0; gi < H->g.size(); gi++) rec[gi] = record(H->sh, H->g[gi], 0.0, 0);
    return ALG_OK;
}
int orc_reset_con(alg_handle* h) { for (Game& g : H->g) reset_con(H->sh, g); return ALG_OK; }
int orc_dual_penalty_update(alg_handle* h, double* vals) {
    const int L = H->sh.D.con_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) { Game& g = H->g[gi]; dual_penalty_update(H->sh, g); if (vals) std::copy(g.vals.begin(), g.vals.end(), vals + gi * L); }
    return ALG_OK;
}
int orc_newton_step(alg_handle* h, int32_t k_outer, int32_t l_inner, const double* delta_in, alg_step_info* info) {
    const int B = (int)H->g.size();
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { int ls = 0; real dl = delta_in ? delta_in[gi] : 0.0; alg_step_info si = inner_iteration(H->sh, H->g[gi], ls, dl, k_outer, l_inner); if (info) info[gi] = si; }
    return ALG_OK;
}
int orc_newton_solve(alg_handle* h, int32_t init, int64_t game_id0, alg_game_stats* stats) {
    const int B = (int)H->g.size();
    if (!H->x0_set || !H->lqr_set) return fail(ALG_ERR_STATE, "orc_newton_solve: x0 / LQR data not set");
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { newton_solve(H->sh, H->g[gi], init != 0, (uint64_t)(game_id0 + gi)); if (stats) stats[gi] = H->g[gi].st; }
    return ALG_OK;
}
int orc_debug_check_guards(alg_handle*) { return 0; }
int orc_set_waves_per_game(alg_handle*, int32_t nw) { return (nw == 0 || nw == 1 || nw == 2 || nw == 4) ? ALG_OK : fail(ALG_ERR_ARG, "bad width"); }   // kernel shape: no meaning on the CPU (ABI mirror)
int orc_set_refinement(alg_handle*, int32_t max_steps, double tol, double mu_tight) { return (max_steps >= 0 && max_steps <= 8 && tol >= 0.0 && mu_tight >= 0.0) ? ALG_OK : fail(ALG_ERR_ARG, "bad refinement setting"); }   // the pivoted LU needs none (ABI mirror)
int orc_get_direction_gate(alg_handle* h, double* out) { if (out) for (size_t e = 0; e < 3 * H->g.size(); e++) out[e] = 0.0; return ALG_OK; }
int orc_get_refinement(alg_handle*, int32_t* max_steps, double* tol, double* mu_tight) { if (max_steps) *max_steps = 0; if (tol) *tol = 0.0; if (mu_tight) *mu_tight = 0.0; return ALG_OK; }
int orc_set_line_search_groups(alg_handle*, int32_t) { return ALG_OK; }     // the oracle's line search is the reference's one-by-one loop (ABI mirror)
int orc_get_line_search_groups(alg_handle*, int32_t* on) { if (on) *on = 0; return ALG_OK; }
int orc_set_handoff(alg_handle*, int32_t iters) { return iters >= 0 ? ALG_OK : fail(ALG_ERR_ARG, "bad hand-off budget"); }   // kernel scheduling: no meaning on the CPU (ABI mirror)
int orc_get_handoff(alg_handle*, int32_t* iters, int32_t* parked_last) { if (iters) *iters = 0; if (parked_last) *parked_last = 0; return ALG_OK; }
int orc_get_waves_per_game(alg_handle*, int32_t* nw) { if (nw) *nw = 1; return ALG_OK; }      // host vectors: nothing to check (ABI mirror)
int orc_newton_solve_async(alg_handle* h, int32_t init, int64_t game_id0) { return orc_newton_solve(h, init, game_id0, nullptr); }
int orc_get_stats(alg_handle* h, alg_game_stats* stats) { for (size_t gi = 0; gi < H->g.size(); gi++) stats[gi] = H->g[gi].st; return ALG_OK; }
int orc_get_history(alg_handle* h, int32_t game, int32_t max_records, alg_record* out, int32_t* n_out) {
    if (game < 0 || game >= (int)H->g.size()) return fail(ALG_ERR_ARG, "bad game index");
    const auto& hs = H->g[game].hist; int c = std::min<int>(max_records, (int)hs.size());
    for (int i = 0; i < c; i++) out[i] = hs[i];
    if (n_out) *n_out = c; return ALG_OK;
}
int orc_synchronize(alg_handle*) { return ALG_OK; }
int orc_ibr_solve_player(alg_handle* h, int32_t player, alg_game_stats* stats) {
    if (player < 0 || player >= H->sh.D.p) return fail(ALG_ERR_ARG, "orc_ibr_solve_player: bad player index");
    const int B = (int)H->g.size();
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { ibr_solve_player(H->sh, H->g[gi], player); if (stats) stats[gi] = H->g[gi].st; }
    return ALG_OK;
}
int orc_ibr_newton_solve(alg_handle* h, int32_t init, int64_t game_id0, int32_t ibr_iter, const int32_t* ordering, double delta_min, alg_game_stats* stats) {
    const int B = (int)H->g.size(), p = H->sh.D.p;
    if (!H->x0_set || !H->lqr_set) return fail(ALG_ERR_STATE, "orc_ibr_newton_solve: x0 / LQR data not set");
    if (!ordering || ibr_iter < 1) return fail(ALG_ERR_ARG, "orc_ibr_newton_solve: bad arguments");
    for (int i = 0; i < p; i++) if (ordering[i] < 0 || ordering[i] >= p) return fail(ALG_ERR_ARG, "orc_ibr_newton_solve: ordering entries must be player ids");
#pragma omp parallel for schedule(dynamic)
    for (int gi = 0; gi < B; gi++) { ibr_newton_solve(H->sh, H->g[gi], init != 0, (uint64_t)(game_id0 + gi), ibr_iter, ordering, delta_min); if (stats) stats[gi] = H->g[gi].st; }
    return ALG_OK;
}
// builder-defined MPC advance: x0 <- RK2(x_1, u_1); totals += finished solve
int orc_mpc_advance(alg_handle* h) {
    const Dims& D = H->sh.D;
    std::vector<real> u(D.m), xn(D.n);
    for (Game& g : H->g) {
        get_control(D, g.z[0], 0, u.data());
        rk2(D, state(D, g.z[0], 0), u.data(), xn.data());
        g.x0 = xn;
        for (int t = 0; t < 2; t++) std::copy(xn.begin(), xn.end(), g.z[t].begin());
        g.mpc_iters += g.st.newton_iters; g.mpc_conv += g.st.converged;
    }
    return ALG_OK;
}
// the loop of alg_mpc_solve, literally: per step newton_solve! (shift / dual_reset of the handle at step 0, then 1 / false) + advance
int orc_mpc_solve(alg_handle* h, int32_t steps, int64_t game_id0, double* states) {
    if (steps < 1) return fail(ALG_ERR_ARG, "orc_mpc_solve: bad argument");
    const Dims& D = H->sh.D; const int B = (int)H->g.size();
    const alg_options saved = H->sh.opt;
    auto snap = [&](int t) { if (states) for (int gi = 0; gi < B; gi++) std::copy(H->g[gi].x0.begin(), H->g[gi].x0.end(), states + ((size_t)t * B + gi) * D.n); };
    snap(0);
    int rc = ALG_OK;
    for (int t = 0; t < steps && rc == ALG_OK; t++) {
        if (t >= 1) { H->sh.opt.shift = 1; H->sh.opt.dual_reset = 0; }
        rc = orc_newton_solve(h, 1, game_id0 + (int64_t)t * 1000003, nullptr);
        if (rc == ALG_OK) rc = orc_mpc_advance(h);
        snap(t + 1);
    }
    H->sh.opt = saved;
    return rc;
}
int orc_mpc_totals(alg_handle* h, int64_t* it, int64_t* cv, int32_t reset) {
    for (size_t gi = 0; gi < H->g.size(); gi++) {
        if (it) it[gi] = H->g[gi].mpc_iters; if (cv) cv[gi] = H->g[gi].mpc_conv;
        if (reset) { H->g[gi].mpc_iters = 0; H->g[gi].mpc_conv = 0; }
    }
    return ALG_OK;
}

// number of OpenMP threads used for the loop over games (cpu_baseline: all cores / one core); returns the previous max
int orc_set_threads(int nthreads) {
#ifdef _OPENMP
    int prev = omp_get_max_threads();
    if (nthreads > 0) omp_set_num_threads(nthreads);
    return prev;
#else
    (void)nthreads; return 1;
#endif
}

// ---- fine-grained pieces exposed for the known-answer tests (Appendix B) ----------------------
int orc_kat_dynamics(const alg_desc* d, const double* x, const double* u, double* xdot, double* x_rk2, double* x_rk3, double* jac_rk2) {
    Dims D; if (!D.init(*d)) return fail(ALG_ERR_ARG, "bad descriptor");
    const std::vector<real> xr = rin(x, D.n), ur = rin(u, D.m);
    ROut o1(xdot, D.n), o2(x_rk2, D.n), o3(x_rk3, D.n), o4(jac_rk2, (size_t)D.n * (D.n + D.m));
    if (xdot) dynamics(D, xr.data(), ur.data(), o1.ptr());
    if (x_rk2) rk2(D, xr.data(), ur.data(), o2.ptr());
    if (x_rk3) rk3(D, xr.data(), ur.data(), o3.ptr());
    if (jac_rk2) rk2_jacobian(D, xr.data(), ur.data(), o4.ptr());
    return ALG_OK;
}
// the same on a handle's model (its parameters: bicycle lengths, quadrotor mass)
int orc_kat_dynamics_h(alg_handle* h, const double* x, const double* u, double* xdot, double* x_rk2, double* x_rk3, double* jac_rk2) {
    const Dims& D = H->sh.D;
    const std::vector<real> xr = rin(x, D.n), ur = rin(u, D.m);
    ROut o1(xdot, D.n), o2(x_rk2, D.n), o3(x_rk3, D.n), o4(jac_rk2, (size_t)D.n * (D.n + D.m));
    if (xdot) dynamics(D, xr.data(), ur.data(), o1.ptr());
    if (x_rk2) rk2(D, xr.data(), ur.data(), o2.ptr());
    if (x_rk3) rk3(D, xr.data(), ur.data(), o3.ptr());
    if (jac_rk2) rk2_jacobian(D, xr.data(), ur.data(), o4.ptr());
    return ALG_OK;
}
// cost gradient/Hessian of player i at knot k (0-based) for state x / control u: q (n), r (mi), Q (n x n row-major)
int orc_kat_cost(alg_handle* h, int32_t game, int32_t i, int32_t k, const double* x, const double* u, double* q, double* r, double* Qm) {
    Game& g = H->g[game]; const Dims& D = H->sh.D;
    const std::vector<real> xr = rin(x, D.n), ur = rin(u, D.m);
    ROut oq(q, D.n), orr(r, D.mi), oQ(Qm, (size_t)D.n * D.n);
    if (q) cost_grad_x(H->sh, g, i, k, xr.data(), oq.ptr());
    if (r) cost_grad_u(H->sh, g, i, u ? ur.data() : nullptr, orr.ptr());
    if (Qm) cost_hess_x(H->sh, g, i, k, xr.data(), oQ.ptr());
    return ALG_OK;
}
// stage_cost of CollisionCost (objective.jl:122-126), for the 0.05 KAT
double orc_kat_collision_cost(double mu, double r, const double* xi, const double* xj) {
    double s = 0; for (int a = 0; a < 2; a++) s += (xi[a] - xj[a]) * (xi[a] - xj[a]);
    double v = std::max(0.0, r - std::sqrt(s)); return 0.5 * mu * v * v;
}
int orc_kat_evaluate_con(alg_handle* h, double* vals) {
    const int L = H->sh.D.con_len;
    for (size_t gi = 0; gi < H->g.size(); gi++) { Game& g = H->g[gi]; evaluate_con(H->sh, g, g.z[0]); std::copy(g.vals.begin(), g.vals.end(), vals + gi * L); }
    return ALG_OK;
}
double orc_kat_delta_step(alg_handle* h, int32_t game, double alpha) { return delta_step(H->sh, H->g[game].z[2], alpha); }
double orc_counter_uniform(uint64_t seed, uint64_t game, uint64_t counter) { return counter_uniform(seed, game, counter); }
// dense partial-pivot LU solve of the literal S x S system in the reference's own (vertical, horizontal)
// ordering -- cross-check of the banded solver on small problems.  delta: S
int orc_kat_dense_direction(alg_handle* h, int32_t game, double reg, double* delta) {
    Game& g = H->g[game]; const int S = H->sh.D.S;
    std::vector<double> A((size_t)S * S, 0.0), rhs(S);
    residual(H->sh, g, g.z[0], 0.0, nullptr);
    jacobian(H->sh, g, g.z[0], reg, [&](int r, int c, double v) { A[(size_t)r * S + c] += v; });
    for (int r = 0; r < S; r++) rhs[r] = g.res[r];
    for (int j = 0; j < S; j++) {
        int pv = j; double best = std::fabs(A[(size_t)j * S + j]);
        for (int r = j + 1; r < S; r++) { double v = std::fabs(A[(size_t)r * S + j]); if (v > best) { best = v; pv = r; } }
        if (best == 0.0) return fail(ALG_ERR_STATE, "singular");
        if (pv != j) { for (int c = 0; c < S; c++) std::swap(A[(size_t)j * S + c], A[(size_t)pv * S + c]); std::swap(rhs[j], rhs[pv]); }
        for (int r = j + 1; r < S; r++) {
            double f = A[(size_t)r * S + j] / A[(size_t)j * S + j];
            if (f == 0.0) continue;
            for (int c = j; c < S; c++) A[(size_t)r * S + c] -= f * A[(size_t)j * S + c];
            rhs[r] -= f * rhs[j];
        }
    }
    for (int j = S - 1; j >= 0; j--) { double s = rhs[j]; for (int c = j + 1; c < S; c++) s -= A[(size_t)j * S + c] * delta[c]; delta[j] = s / A[(size_t)j * S + j]; }
    for (int j = 0; j < S; j++) delta[j] = -delta[j];
    return ALG_OK;
}

} // extern "C"
