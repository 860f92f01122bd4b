"""Active-set / null-space analysis of a solved game: host-side post-processing, NumPy only.

Mirror of src/active_set/active_set_stamp.jl, active_set_core.jl, active_set_methods.jl.  The reference augments the KKT
system of newton_solve! with the collision-avoidance constraints that are active at the solution -- one extra row per
unordered pair (i < j) and knot, one extra column (multiplier) per ordered pair (i, j != i) and knot -- and takes the null
space of the rows / columns selected by the active set: the directions along which the generalized Nash equilibrium is not
isolated.  It consumes what the batched solver leaves behind: `residual` / `residual_jacobian` of the device path (through the
C ABI, dense, reg = 0) and the trajectory + multipliers of one game of the batch.

Literal behaviours of the reference that are kept (and said so where they happen):
  * `residual_jacobian!(ascore, ...)` (active_set_methods.jl:127-170) tests `valid(hs, N, p)` on an HStamp that is only
    assigned INSIDE that branch; the default HStamp(:x, 0, 0) is invalid, so the branch that would write the constraint rows
    d c / d x never runs and only the columns C' (opt_i,x_k rows x multiplier columns) are filled.
  * `update_nullspace!` calls `nullspace(djac, atol=1e-20)`: with atol far below the floating-point noise of an SVD, every one of
    the min(rows, cols) singular values counts as non-zero -- also those of the identically-zero constraint rows -- which is
    what makes the reference's own test see (N-1) p null vectors (test/active_set/active_set_methods.jl:112-116).  `nullspace`
    below reproduces that count deterministically (see its docstring).
Indices are 1-based like the reference's, so its tests' literals can be quoted unchanged."""
import dataclasses
import types

import numpy as np


# ---- CStamp (active_set_stamp.jl:13-81) -----------------------------------------------------------------------------------
@dataclasses.dataclass(eq=True, frozen=False, unsafe_hash=True)
class CStamp:
    dim: str = "x"      # dimension: "v" (constraint row, unordered pair i < j) or "h" (multiplier column, ordered pair i != j)
    con: str = "x"      # name of the constraint ("col")
    i: int = 0
    j: int = 0
    k: int = 0


def stampify_c(dim, con, i, j, k):
    return CStamp(dim, con, i, j, k)


def valid_c(s, N, p):
    """valid(s::CStamp, N, p), active_set_stamp.jl:64-81."""
    if s.dim == "v":
        return s.i < s.j and 1 <= s.i <= p and 1 <= s.j <= p and 2 <= s.k <= N
    if s.dim == "h":
        return 1 <= s.i <= p and 1 <= s.j <= p and 2 <= s.k <= N and s.i != s.j
    return False


# ---- NullSpace / ActiveSetCore (active_set_core.jl:5-92) --------------------------------------------------------------------
class NullSpace:
    def __init__(self, probsize):
        self.probsize = probsize
        self.reset()

    def reset(self):
        self.mat = np.zeros((0, 0))
        self.vec, self.Δtraj, self.Δλ = [], [], []

    def add_matrix(self, mat, hmask):
        """add_matrix!(null, mat, hmask), active_set_core.jl:28-48: scatter every null vector to the full horizontal space, scale it
        to unit mean absolute value, split it into the trajectory part and the constraint-multiplier part."""
        ps = self.probsize
        S, Sh = ps.S, ps.S + ps.p * (ps.p - 1) * (ps.N - 1)
        m1, m2 = mat.shape
        assert S <= m1 <= Sh
        self.mat = mat
        idx = np.asarray(hmask) - 1
        for l in range(m2):
            vec = np.zeros(Sh)
            vec[idx] = mat[:, l]
            vec /= np.mean(np.abs(vec))
            self.vec.append(vec); self.Δtraj.append(vec[:S].copy()); self.Δλ.append(vec[S:].copy())


def complete_vertical_indices(probsize):
    """active_set_core.jl:97-124: the rows of newton_core.jl:40-63 followed by one row per (k = 2..N, i, j > i)."""
    from .host import vertical_indices
    out = dict(vertical_indices(probsize))
    off = probsize.S
    for k in range(2, probsize.N + 1):
        for i in range(1, probsize.p + 1):
            for j in range(i + 1, probsize.p + 1):
                off += 1
                out[CStamp("v", "col", i, j, k)] = [off]
    return out


def complete_horizontal_indices(probsize):
    """active_set_core.jl:126-155: the columns of newton_core.jl:65-89 followed by one column per (k = 2..N, i, j != i)."""
    from .host import horizontal_indices
    out = dict(horizontal_indices(probsize))
    off = probsize.S
    for k in range(2, probsize.N + 1):
        for i in range(1, probsize.p + 1):
            for j in range(1, probsize.p + 1):
                if j != i:
                    off += 1
                    out[CStamp("h", "col", i, j, k)] = [off]
    return out


class ActiveSetCore:
    """ActiveSetCore(probsize), active_set_core.jl:54-92.  res (Sv), jac (Sv x Sh, dense here), index maps, masks, NullSpace."""

    def __init__(self, probsize):
        N, p = probsize.N, probsize.p
        self.probsize = probsize
        self.Sv = probsize.S + p * (p - 1) * (N - 1) // 2
        self.Sh = probsize.S + p * (p - 1) * (N - 1)
        self.res = np.zeros(self.Sv)
        self.res_tmp = self.res.copy()
        self.jac = np.zeros((self.Sv, self.Sh))
        self.verti_inds = complete_vertical_indices(probsize)
        self.horiz_inds = complete_horizontal_indices(probsize)
        self.vmask = list(range(1, self.Sv + 1))
        self.hmask = list(range(1, self.Sh + 1))
        self.null = NullSpace(probsize)


def vertical_idx(core, stamp):
    return core.verti_inds[stamp]


def horizontal_idx(core, stamp):
    return core.horiz_inds[stamp]


# ---- collision-avoidance constraint values of one game (the reference's state_conval[i] entries of CollisionConstraint) -------
def collision_convals(game_con):
    """{(i, j): conval} for j != i (1-based), each with inds = 2..N and per-knot arrays vals, λ, active, jac (N-1, n) -- the
    fields of Altro.ALConVal that the active-set code reads.  Created on first use, kept on game_con (tests poke them like the
    reference's tests poke `game_con.state_conval[1][1].λ[1][1]`)."""
    if getattr(game_con, "_col_conval", None) is None:
        ps = game_con.probsize
        cv = {}
        K = ps.N - 1
        mk = lambda radius: types.SimpleNamespace(inds=list(range(2, ps.N + 1)), vals=np.zeros(K), λ=np.zeros(K),
                                                  active=np.zeros(K, dtype=bool), jac=np.zeros((K, ps.n)), radius=float(radius))
        if game_con.collision_radius is not None:
            for i in range(1, ps.p + 1):
                for j in range(1, ps.p + 1):
                    if j != i:
                        cv[(i, j)] = mk(game_con.collision_radius[i - 1] + game_con.collision_radius[j - 1])
        # add_collision_avoidance!(game_con, i, j, radius) (constraints_methods.jl:5-19): single ordered pairs with their own radius
        for (i, j), radius in sorted(getattr(game_con, "collision_pairs", {}).items()):
            cv[(i, j)] = mk(radius)
        game_con._col_conval = cv
    return game_con._col_conval


def _col_inds(game_con, i):
    """State entries a CollisionConstraint of player i reads: px[i] (x, y), or pz[i][1:3] (x, y, z) for the spherical form
    (constraints_methods.jl:13, 52)."""
    ps = game_con.probsize
    idx = ps.pz[i - 1][:3] if getattr(game_con, "spherical", False) else ps.px[i - 1]
    return [a - 1 for a in idx]


def evaluate(game_con, states):
    """evaluate!(game_con, traj) + jacobian! for the collision constraints (constraints_methods.jl:367-393):
    c = R^2 - |x[px_i] - x[px_j]|^2 at knots 2..N, d c / d x = -2 Δ on px_i, +2 Δ on px_j.  states: (N, n)."""
    for (i, j), cv in collision_convals(game_con).items():
        pi, pj = _col_inds(game_con, i), _col_inds(game_con, j)
        d = states[1:, pi] - states[1:, pj]
        cv.vals[:] = cv.radius ** 2 - (d * d).sum(axis=1)
        cv.jac[:] = 0.0
        cv.jac[:, pi] = -2.0 * d
        cv.jac[:, pj] = 2.0 * d


def update_active_set(game_con, states=None, tol=None):
    """update_active_set!(game_con[, traj]) (constraints_methods.jl:396-415) with Altro's rule a = (c >= -tol) | (λ > 0)
    [pinned: test/active_set/active_set_methods.jl:16-34 with tol = 0]."""
    if states is not None:
        evaluate(game_con, states)
    tol = getattr(game_con, "active_set_tolerance", 0.0) if tol is None else tol
    for cv in collision_convals(game_con).values():
        cv.active[:] = (cv.vals >= -tol) | (cv.λ > 0)


def active(game_con, stamp):
    """active(game_con, stamp), active_set_methods.jl:5-27: [conval(i, j).active at knot k] for a valid collision stamp, else [0]."""
    ps = game_con.probsize
    if stamp.con == "col" and valid_c(stamp, ps.N, ps.p):
        cv = collision_convals(game_con).get((stamp.i, stamp.j))
        if cv is not None:
            return [int(cv.active[cv.inds.index(stamp.k)])]
    return [0]


def active_vertical_mask(ascore, game_con):
    """active_vertical_mask!, active_set_methods.jl:29-52: 1..S plus the rows of the active (i < j, k) constraints."""
    ps = game_con.probsize
    ascore.vmask = list(range(1, ps.S + 1))
    for k in range(2, ps.N + 1):
        for i in range(1, ps.p + 1):
            for j in range(i + 1, ps.p + 1):
                s = CStamp("v", "col", i, j, k)
                if active(game_con, s)[0] == 1:
                    ascore.vmask += vertical_idx(ascore, s)


def active_horizontal_mask(ascore, game_con):
    """active_horizontal_mask!, active_set_methods.jl:54-77: 1..S plus the multiplier columns of the active (i, j != i, k)."""
    ps = game_con.probsize
    ascore.hmask = list(range(1, ps.S + 1))
    for k in range(2, ps.N + 1):
        for i in range(1, ps.p + 1):
            for j in range(1, ps.p + 1):
                if j != i:
                    s = CStamp("h", "col", i, j, k)
                    if active(game_con, s)[0] == 1:
                        ascore.hmask += horizontal_idx(ascore, s)


def _game_state(prob, game):
    X = prob.pdtraj.states[game]
    lam, _ = prob.batch.get_con_duals()
    ps = prob.probsize
    K = ps.N - 1
    cvs = collision_convals(prob.game_con)
    for (i, j), cv in cvs.items():                  # ABI layout: pair q = (i, j) in add_collision_avoidance! order, knots 2..N
        q = (i - 1) * (ps.p - 1) + ((j - 1) if j < i else (j - 2))
        cv.λ[:] = lam[game, q * K:(q + 1) * K]
    return X


def residual(ascore, prob, game=0):
    """residual!(ascore, prob, pdtraj), active_set_methods.jl:99-125: core.res followed by the values of the (i < j) constraints."""
    from . import host
    ps = ascore.probsize
    ascore.res[:] = 0.0
    ascore.res[:ps.S] = host.residual(prob)[game]
    X = _game_state(prob, game)
    evaluate(prob.game_con, X)
    for (i, j), cv in collision_convals(prob.game_con).items():
        if i < j:
            for l, k in enumerate(cv.inds):
                s = CStamp("v", "col", i, j, k)
                if valid_c(s, ps.N, ps.p):
                    ascore.res[vertical_idx(ascore, s)[0] - 1] += cv.vals[l]


def residual_jacobian(ascore, prob, game=0, constraint_rows=False):
    """residual_jacobian!(ascore, prob, pdtraj), active_set_methods.jl:127-170.  The S x S block is the solver's KKT Jacobian
    (alg_residual_jacobian, reg = 0); column (h, col, i, j, k) receives d c_ij / d x on the rows opt_i,x_k.
    constraint_rows=False is the reference as it executes (see the module docstring); True also writes d c_ij / d x into the
    row (v, col, i, j, k), i < j, on the columns x_k -- what the dead branch was written to do."""
    from . import host
    ps = ascore.probsize
    N, p = ps.N, ps.p
    ascore.jac[:] = 0.0
    ascore.jac[:ps.S, :ps.S] = host.residual_jacobian(prob, 0.0, games=(game, 1))[0]      # this game only, and give the device copy back
    prob.batch.release_scratch()
    X = _game_state(prob, game)
    evaluate(prob.game_con, X)
    for (i, j), cv in collision_convals(prob.game_con).items():
        for l, k in enumerate(cv.inds):
            vs = ("opt", i, "x", 1, k)
            cs = CStamp("h", "col", i, j, k)
            if valid_c(cs, N, p) and host.valid(vs, N, p):
                rows = np.asarray(ascore.verti_inds[vs]) - 1
                ascore.jac[rows, horizontal_idx(ascore, cs)[0] - 1] += cv.jac[l]
            if constraint_rows and i < j:
                cols = np.asarray(ascore.horiz_inds[("x", 1, k)]) - 1
                ascore.jac[vertical_idx(ascore, CStamp("v", "col", i, j, k))[0] - 1, cols] += cv.jac[l]


def nullspace(A, atol=1e-20):
    """LinearAlgebra.nullspace(A; atol) as update_nullspace! uses it (active_set_methods.jl:181): the right singular vectors past
    the first r, r = number of singular values > atol.  The reference's atol = 1e-20 is below what LAPACK leaves in the singular
    values of identically-zero rows (~1e-17 sigma_max), so in the reference those count as non-zero; NumPy may return them as
    exact zeros.  To give the same answer on every platform, singular values that belong to identically-zero rows of A are
    counted as non-zero whenever atol is below eps * sigma_max.

    This is an APPROXIMATION of the reference's call, exact in the case the reference meets (rank deficiency from identically-zero
    rows: inactive constraint rows of the active-set Jacobian): the rank used is (numerical rank at eps * sigma_max * max(m, n))
    + (number of zero rows).  A matrix that is rank deficient for another reason (dependent non-zero rows) has singular values of
    ~1e-17 sigma_max there, which LinearAlgebra.nullspace(atol = 1e-20) counts as non-zero and this function does not: its null
    space is then larger than the reference's.  Pass atol >= eps * sigma_max to get the plain `s > atol` count."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    if m == 0 or n == 0:
        return np.eye(n)
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    r = int((s > atol).sum())
    if atol < np.finfo(float).eps * (s[0] if s.size else 0.0):
        zero_rows = int((~A.any(axis=1)).sum())
        r = min(min(m, n), max(r, int((s > np.finfo(float).eps * s[0] * max(m, n)).sum()) + zero_rows))
    return Vt[r:].T.copy()


def update_nullspace(ascore, prob, game=0, atol=1e-20, constraint_rows=False):
    """update_nullspace!(ascore, prob, pdtraj), active_set_methods.jl:173-184."""
    X = _game_state(prob, game)
    update_active_set(prob.game_con, X, tol=getattr(prob.opts, "active_set_tolerance", 0.0))
    active_vertical_mask(ascore, prob.game_con)
    active_horizontal_mask(ascore, prob.game_con)
    residual_jacobian(ascore, prob, game, constraint_rows=constraint_rows)
    djac = ascore.jac[np.ix_(np.asarray(ascore.vmask) - 1, np.asarray(ascore.hmask) - 1)]
    ascore.null.reset()
    ascore.null.add_matrix(nullspace(djac, atol=atol), ascore.hmask)
