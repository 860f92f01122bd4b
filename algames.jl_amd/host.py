"""Host-side mirror of the Algames.jl API surface for the Newton / augmented-Lagrangian hot path.

Same names, argument meaning and error behaviour as the reference (Julia `f!` -> Python `f`), but a
`GameProblem` owns a *batch* of B games (x0 of shape (B, n)) that are solved together on one MI355X
through the C ABI of include/algames_hip.h.  Index sets / stamps are kept 1-based like the reference
so that its tests' literal values can be quoted unchanged.

Reference: src/Algames.jl:19-165 (exports), src/problem/problem.jl, src/problem/solver_methods.jl,
src/struct/*.jl, src/dynamics/{double_integrator,unicycle}.jl, src/objective/objective.jl,
src/constraints/{game_constraints,constraints_methods}.jl.
"""
import dataclasses
import os

import numpy as np

from . import _abi
from ._abi import (ALG_MODEL_BICYCLE, ALG_MODEL_DOUBLE_INTEGRATOR, ALG_MODEL_QUADROTOR, ALG_MODEL_UNICYCLE, ALG_TRAJ_PD, ALG_TRAJ_TRIAL,
                   ALG_TRAJ_DELTA, AlgamesError, Batch, CLib)

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("ALGAMES_HIP_LIB", os.path.join(_HERE, "lib", "libalgames_hip.so"))
_hip = None


def hip_lib():
    """The product backend.  Fails loudly when the HIP extension is missing -- there is no CPU fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise AlgamesError(
                f"{HIP_LIB_PATH} is not built. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        _hip = CLib(HIP_LIB_PATH, "alg_")
    return _hip


# --------------------------------------------------------------------------------------------------
# Models (src/dynamics/double_integrator.jl:2-25, src/dynamics/unicycle.jl:2-25)
# --------------------------------------------------------------------------------------------------
class AbstractGameModel:
    pass


class DoubleIntegratorGame(AbstractGameModel):
    model_id = ALG_MODEL_DOUBLE_INTEGRATOR

    def __init__(self, p=2, d=2):
        self.p, self.d = p, d
        self.n, self.m = 2 * d * p, d * p
        self.pu = [[i + (j - 1) * p for j in range(1, d + 1)] for i in range(1, p + 1)]
        self.px = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.pz = [[i + (j - 1) * p for j in range(1, 2 * d + 1)] for i in range(1, p + 1)]
        self.ni = [2 * d] * p
        self.mi = [d] * p


class UnicycleGame(AbstractGameModel):
    model_id = ALG_MODEL_UNICYCLE

    def __init__(self, p=2):
        self.p, self.d = p, 2
        self.n, self.m = 4 * p, 2 * p
        self.pu = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.px = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.pz = [[i + (j - 1) * p for j in range(1, 5)] for i in range(1, p + 1)]
        self.ni = [4] * p
        self.mi = [2] * p


class BicycleGame(AbstractGameModel):
    """BicycleGame(p; lf, lr), src/dynamics/bicycle.jl:2-27: X = [x, y, v, psi], U = [a, delta]."""
    model_id = ALG_MODEL_BICYCLE

    def __init__(self, p=2, lf=0.05, lr=0.05):
        self.p, self.d = p, 2
        self.lf, self.lr = float(lf), float(lr)
        self.n, self.m = 4 * p, 2 * p
        self.pu = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.px = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.pz = [[i + (j - 1) * p for j in range(1, 5)] for i in range(1, p + 1)]
        self.ni = [4] * p
        self.mi = [2] * p


class QuadrotorGame(AbstractGameModel):
    """QuadrotorGame(; p, mass), src/dynamics/quadrotor.jl:3-46: per player 12 states [x, y, z | MRP q1 q2 q3 | vx vy vz | wx wy wz]
    and 4 rotor commands; the constructor's constants (J, gravity, motor_dist, kf, km) are fixed in the reference."""
    model_id = ALG_MODEL_QUADROTOR

    def __init__(self, p=2, mass=0.5):
        assert p <= 4                      # quadrotor.jl:22
        if not mass > 0:
            raise AlgamesError("QuadrotorGame: mass must be positive")
        self.p, self.d, self.mass = p, 3, float(mass)
        self.n, self.m = 12 * p, 4 * p
        self.pu = [[i + (j - 1) * p for j in range(1, 5)] for i in range(1, p + 1)]
        self.px = [[i + (j - 1) * p for j in range(1, 3)] for i in range(1, p + 1)]
        self.pz = [[i + (j - 1) * p for j in range(1, 13)] for i in range(1, p + 1)]
        self.ni = [12] * p
        self.mi = [4] * p


def dim(model):
    if isinstance(model, QuadrotorGame):
        return 3                           # quadrotor.jl:208
    return model.mi[0] if isinstance(model, DoubleIntegratorGame) else 2


# --------------------------------------------------------------------------------------------------
# ProblemSize (src/struct/problem_size.jl:5-35) and index maps (src/core/newton_core.jl:40-89)
# --------------------------------------------------------------------------------------------------
class ProblemSize:
    def __init__(self, N, model):
        self.N, self.n, self.m, self.p = N, model.n, model.m, model.p
        self.ni, self.mi = list(model.ni), list(model.mi)
        self.pu, self.px, self.pz = model.pu, model.px, model.pz
        self.S = self.n * self.p * (N - 1) + self.m * (N - 1) + self.n * (N - 1)

    def __eq__(self, o):
        return all(getattr(self, f) == getattr(o, f) for f in ("N", "n", "m", "p", "ni", "mi", "pu", "px", "pz", "S"))


def stampify(*a):
    """Stamps are plain tuples: VStamp (prob,i0,n1,i1,v1), HStamp (n2,i2,v2), Stamp = V + H."""
    return tuple(a)


def valid(stamp, N, p):
    """Stamp validity rules, src/core/stamp.jl:167-229."""
    def v1(prob, i0, n1, i1, k):
        if prob == "opt" and 1 <= i0 <= p:
            if n1 == "u" and i1 == i0 and 1 <= k <= N - 1:
                return True
            if n1 == "x" and i1 == 1 and 2 <= k <= N:
                return True
        if prob == "dyn" and i0 == 1:
            if n1 == "x" and i1 == 1 and 1 <= k <= N - 1:
                return True
        return False

    def v2(n2, i2, k, prob=None, i0=None):
        if n2 == "u" and 1 <= i2 <= p and 1 <= k <= N - 1:
            return True
        if n2 == "λ" and 1 <= k <= N - 1 and ((prob is None and 1 <= i2 <= p) or (prob == "opt" and i2 == i0)):
            return True
        if n2 == "x" and i2 == 1 and 2 <= k <= N:
            return True
        return False

    if len(stamp) == 5:
        return v1(*stamp)
    if len(stamp) == 3:
        return v2(*stamp)
    prob, i0 = stamp[0], stamp[1]
    if prob == "opt" and not (1 <= i0 <= p):
        return False
    if prob == "dyn" and i0 != 1:
        return False
    if prob not in ("opt", "dyn"):
        return False
    return v1(*stamp[:5]) and v2(*stamp[5:], prob=prob, i0=i0)


def vertical_indices(probsize):
    """Row ("vertical") 1-based index ranges of `core.res`, src/core/newton_core.jl:40-63."""
    N, n, p, mi = probsize.N, probsize.n, probsize.p, probsize.mi
    out, off = {}, 0
    for i in range(1, p + 1):
        for k in range(1, N):
            out[stampify("opt", i, "x", 1, k + 1)] = list(range(off + 1, off + n + 1)); off += n
            out[stampify("opt", i, "u", i, k)] = list(range(off + 1, off + mi[i - 1] + 1)); off += mi[i - 1]
    for k in range(1, N):
        out[stampify("dyn", 1, "x", 1, k)] = list(range(off + 1, off + n + 1)); off += n
    return out


def horizontal_indices(probsize):
    """Column ("horizontal") 1-based index ranges of `Δtraj`, src/core/newton_core.jl:65-89."""
    N, n, p, mi = probsize.N, probsize.n, probsize.p, probsize.mi
    out, off = {}, 0
    for k in range(1, N):
        out[stampify("x", 1, k + 1)] = list(range(off + 1, off + n + 1)); off += n
        for i in range(1, p + 1):
            out[stampify("u", i, k)] = list(range(off + 1, off + mi[i - 1] + 1)); off += mi[i - 1]
        for i in range(1, p + 1):
            out[stampify("λ", i, k)] = list(range(off + 1, off + n + 1)); off += n
    return out


# --------------------------------------------------------------------------------------------------
# Options / Regularizer (src/struct/options.jl:5-116, src/struct/regularizer.jl:5-35)
# --------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Regularizer:
    x: float = 1e-3
    u: float = 1e-3
    λ: float = 1e-3


def _ones10():
    return [1.0] * 10


@dataclasses.dataclass
class Options:
    θ: float = 1e-2
    f_init: object = "rand"          # "rand": counter-based generator on the device (alg_init_traj); a callable f(size): init_traj_host
    amplitude_init: float = 1e-8
    shift: int = 2 ** 10
    regularize: bool = True
    reg: Regularizer = dataclasses.field(default_factory=Regularizer)
    reg_0: float = 1e-3
    α_0: float = 1.0
    α_increase: float = 1.2
    α_decrease: float = 0.5
    β: float = 0.01
    ls_iter: int = 25
    Δ_min: float = 1e-9
    ρ_0: float = 1.0
    ρ_trial: float = 1.0
    ρ_increase: float = 10.0
    ρ_max: float = 1e7
    λ_max: float = 1e7
    α_dual: float = 1.0
    αx_dual: list = dataclasses.field(default_factory=_ones10)
    active_set_tolerance: float = 1e-4
    ϵ_dyn: float = 1e-3
    ϵ_sta: float = 1e-3
    ϵ_con: float = 1e-3
    ϵ_opt: float = 1e-3
    outer_iter: int = 7
    inner_iter: int = 20
    γ: float = 1.0
    mpc_horizon: int = 20
    upsampling: int = 2
    inner_print: bool = True
    outer_print: bool = True
    seed: int = 100
    dual_reset: bool = True

    def to_abi(self):
        return dict(amplitude_init=self.amplitude_init, shift=int(min(self.shift, 2 ** 30)),
                    regularize=int(self.regularize), reg_0=self.reg_0, alpha_decrease=self.α_decrease,
                    beta=self.β, ls_iter=self.ls_iter, dual_reset=int(self.dual_reset),
                    delta_min=self.Δ_min, rho_0=self.ρ_0, rho_increase=self.ρ_increase,
                    rho_max=self.ρ_max, lambda_max=self.λ_max, alpha_dual=self.α_dual,
                    alphax_dual=list(self.αx_dual)[:10], eps_dyn=self.ϵ_dyn, eps_sta=self.ϵ_sta,
                    eps_con=self.ϵ_con, eps_opt=self.ϵ_opt, outer_iter=self.outer_iter,
                    inner_iter=self.inner_iter, seed=self.seed)


# --------------------------------------------------------------------------------------------------
# GameObjective (src/objective/objective.jl:6-100)
# --------------------------------------------------------------------------------------------------
def _diag(M):
    M = np.asarray(M, dtype=np.float64)
    return np.diag(M).copy() if M.ndim == 2 else M.copy()


def expand_vector(v, inds, n):
    V = np.zeros(n)
    V[np.asarray(inds) - 1] = v
    return V


class GameObjective:
    def __init__(self, Q, R, xf, uf, N, model):
        p = model.p
        assert len(Q) == len(R) == len(xf) == len(uf) == p
        self.probsize = ProblemSize(N, model)
        self.Qdiag = np.stack([_diag(Q[i]) for i in range(p)])          # (p, ni)
        self.Rdiag = np.stack([_diag(R[i]) for i in range(p)])          # (p, mi)
        self.xf = np.stack([np.asarray(xf[i], dtype=np.float64) for i in range(p)])
        self.uf = np.stack([np.asarray(uf[i], dtype=np.float64) for i in range(p)])
        self.collision_radius = None
        self.collision_μ = None


def add_collision_cost(game_obj, radius, μ):
    """add_collision_cost!(game_obj, radius, μ), objective.jl:84-100 (one set per objective)."""
    p = game_obj.probsize.p
    assert p == len(radius) == len(μ)
    if game_obj.collision_radius is not None:
        raise AlgamesError("only one collision-cost set per GameObjective is supported")
    game_obj.collision_radius = np.asarray(radius, dtype=np.float64)
    game_obj.collision_μ = np.asarray(μ, dtype=np.float64)


# --------------------------------------------------------------------------------------------------
# GameConstraintValues (src/constraints/game_constraints.jl, constraints_methods.jl)
# --------------------------------------------------------------------------------------------------
class GameConstraintValues:
    def __init__(self, probsize):
        self.probsize = probsize
        self.α_dual = 1.0
        self.αx_dual = [1.0] * probsize.p
        self.active_set_tolerance = 0.0  # game_constraints.jl:23; set_constraint_params! copies opts.active_set_tolerance (:37)
        self.collision_radius = None     # per player, pair radius = r_i + r_j
        self.collision_pairs = {}        # (i, j) 1-based ordered pair -> radius of its own CollisionConstraint (the per-pair adders)
        self.u_max = None
        self.u_min = None
        self.state_bounds = {}           # player (1-based) -> merged (x_max, x_min) on the joint state
        self.state_conval = [[] for _ in range(probsize.p)]   # per player: the state-bound sets in the order they were added
        self.walls = None
        self.circles = None
        self.player_walls = {}           # player (1-based) -> list of Wall (add_wall_constraint!(game_con, i, walls))
        self.player_circles = {}         # player (1-based) -> (xc, yc, radius) (add_circle_constraint!(game_con, i, ...))
        self.spherical = False           # collision_radius applies to the 3-D distance (add_spherical_collision_avoidance!)
        self.walls3d = None
        self.cylinders = None
        self.player_walls3d = {}         # player (1-based) -> list of Wall3D (add_wall_constraint!(game_con, i, walls::Vector{Wall3D}))
        self.player_cylinders = {}       # player (1-based) -> list of CylinderWall


def add_collision_avoidance(game_con, *args):
    """add_collision_avoidance!(game_con, radius) (constraints_methods.jl:21-39: every ordered pair, r_i + r_j) or
    add_collision_avoidance!(game_con, i, j, radius) (:5-19: ONE CollisionConstraint of player i against player j, 1-based, with
    its own radius)."""
    p = game_con.probsize.p
    if len(args) == 3:
        i, j, radius = int(args[0]), int(args[1]), float(args[2])
        if not (1 <= i <= p and 1 <= j <= p and i != j):
            raise AlgamesError("add_collision_avoidance!(game_con, i, j, radius): players i != j in 1..p")
        if game_con.collision_radius is not None or (i, j) in game_con.collision_pairs:
            raise AlgamesError("one collision-avoidance constraint per ordered pair is supported")
        game_con.collision_pairs[(i, j)] = radius
        return
    (radius,) = args
    r = np.asarray(radius, dtype=np.float64)
    if r.ndim == 0:
        r = r * np.ones(p)
    assert p == len(r)
    if game_con.collision_radius is not None or game_con.collision_pairs:
        raise AlgamesError("only one collision-avoidance set per GameConstraintValues is supported")
    game_con.collision_radius = r


def add_spherical_collision_avoidance(game_con, *args):
    """add_spherical_collision_avoidance!(game_con, radius) / (game_con, i, j, radius), constraints_methods.jl:45-81:
    CollisionConstraint on pz[i][1:3] (the x, y, z positions of a DoubleIntegratorGame with d = 3 or a Quadrotor; other models
    are rejected when the problem is built)."""
    if (game_con.collision_radius is not None or game_con.collision_pairs) and not game_con.spherical:
        raise AlgamesError("planar and spherical collision avoidance cannot be mixed")
    add_collision_avoidance(game_con, *args)
    game_con.spherical = True


def add_control_bound(game_con, u_max, u_min):
    """add_control_bound!(game_con, u_max, u_min), constraints_methods.jl:104-115."""
    u_max = np.asarray(u_max, dtype=np.float64); u_min = np.asarray(u_min, dtype=np.float64)
    if not np.all(u_max >= u_min):
        raise ValueError("Upper bounds must be greater than or equal to lower bounds")   # control_bound_constraint.jl:69-75
    if game_con.u_max is not None:
        raise AlgamesError("only one control-bound set per GameConstraintValues is supported")
    game_con.u_max, game_con.u_min = u_max, u_min


def add_state_bound(game_con, i, x_max, x_min):
    """add_state_bound!(game_con, i, x_max, x_min), constraints_methods.jl:87-98 (player i is 1-based; bounds on the
    joint state, +-inf allowed).  The reference appends one StateBoundConstraint per call; the rows of several calls for
    the same player are independent scalar constraints, so they are kept as one merged (x_max, x_min) pair -- which
    requires that no state entry is bounded on the same side by two calls."""
    n, p = game_con.probsize.n, game_con.probsize.p
    x_max = np.array(x_max, dtype=np.float64); x_min = np.array(x_min, dtype=np.float64)
    if x_max.shape != (n,) or x_min.shape != (n,):
        raise ValueError("state bounds must have length n")
    if not (1 <= i <= p):
        raise ValueError("player index out of range")
    if not np.all(x_max >= x_min):
        raise ValueError("Upper bounds must be greater than or equal to lower bounds")   # checkBounds
    game_con.state_conval[i - 1].append((x_max.copy(), x_min.copy()))
    if i in game_con.state_bounds:
        mx, mn = game_con.state_bounds[i]
        if np.any(np.isfinite(mx) & np.isfinite(x_max)) or np.any(np.isfinite(mn) & np.isfinite(x_min)):
            raise AlgamesError("a state entry of one player can carry only one upper and one lower bound")
        x_max, x_min = np.minimum(mx, x_max), np.maximum(mn, x_min)
    game_con.state_bounds[i] = (x_max, x_min)


def velocity_index(model, i):
    """velocity_index(model, i), velocity_constraint.jl:30-43 (1-based)."""
    assert 1 <= i <= model.p
    if isinstance(model, UnicycleGame):
        return model.pz[i - 1][3]
    if isinstance(model, BicycleGame):
        return model.pz[i - 1][2]
    raise AlgamesError("Velocity Index is not implemented for DoubleIntegratorGame.")


def add_velocity_bound(model, game_con, v_max, v_min):
    """add_velocity_bound!(model, game_con, v_max, v_min), velocity_constraint.jl:1-28: the bound on player i's speed is a
    state bound added to every player's constraint list."""
    n, p = model.n, model.p
    v_max, v_min = np.asarray(v_max, dtype=np.float64), np.asarray(v_min, dtype=np.float64)
    assert len(v_max) == len(v_min) == p
    for i in range(1, p + 1):
        if v_max[i - 1] != np.inf or v_min[i - 1] != -np.inf:
            x_max = np.full(n, np.inf); x_min = np.full(n, -np.inf)
            x_max[velocity_index(model, i) - 1] = v_max[i - 1]
            x_min[velocity_index(model, i) - 1] = v_min[i - 1]
            for j in range(1, p + 1):
                add_state_bound(game_con, j, x_max, x_min)


class Wall:
    """Wall(p1, p2, v), constraints_methods.jl:155-159: segment p1-p2, v orthogonal to it pointing into the forbidden half space."""

    def __init__(self, p1, p2, v):
        self.p1, self.p2, self.v = (np.asarray(a, dtype=np.float64) for a in (p1, p2, v))


class Wall3D:
    """Wall3D(p1, p2, p3, v), constraints_methods.jl:201-206: parallelogram corners p1, p2, p3 and the normal v of its plane
    pointing into the forbidden half space."""

    def __init__(self, p1, p2, p3, v):
        self.p1, self.p2, self.p3, self.v = (np.asarray(a, dtype=np.float64) for a in (p1, p2, p3, v))


class CylinderWall:
    """CylinderWall(p, v, l, r), constraints_methods.jl:249-254: axis-aligned cylinder with origin p, axis v in (:x, :y, :z)
    (also accepted: "x"/"y"/"z" or 0/1/2), length l, radius r."""

    def __init__(self, p, v, l, r):
        self.p = np.asarray(p, dtype=np.float64)
        self.v = {"x": 0, "y": 1, "z": 2, ":x": 0, ":y": 1, ":z": 2, 0: 0, 1: 1, 2: 2}[v]
        self.l, self.r = float(l), float(r)


def add_wall_constraint(game_con, *args):
    """add_wall_constraint!(game_con, walls) (constraints_methods.jl:189-195, every player) or add_wall_constraint!(game_con, i, walls)
    (:161-187, player i only, 1-based); dispatches on the wall type like the reference's methods for Vector{Wall} (:161),
    Vector{Wall3D} (:208) and Vector{CylinderWall} (:256)."""
    if len(args) == 2:
        i, walls = int(args[0]), list(args[1])
        if not 1 <= i <= game_con.probsize.p:
            raise ValueError("add_wall_constraint: player index out of range")
        kinds = {type(w) for w in walls}
        if len(kinds) != 1 or not kinds <= {Wall, Wall3D, CylinderWall}:
            raise TypeError("add_wall_constraint(game_con, i, walls): walls must all be Wall, all Wall3D or all CylinderWall")
        kind = kinds.pop()                         # the reference's three methods: Vector{Wall} (:161), Vector{Wall3D} (:208), Vector{CylinderWall} (:256)
        shared, own = {Wall: ("walls", "player_walls"), Wall3D: ("walls3d", "player_walls3d"), CylinderWall: ("cylinders", "player_cylinders")}[kind]
        if getattr(game_con, shared) is not None:
            raise AlgamesError("per-player walls cannot be combined with an all-player wall set")
        getattr(game_con, own).setdefault(i, []).extend(walls)
        return
    (walls,) = args
    walls = list(walls)
    kinds = {type(w) for w in walls}
    if len(kinds) != 1:
        raise TypeError("add_wall_constraint: walls must all be Wall, all Wall3D or all CylinderWall")
    slot = {Wall: "walls", Wall3D: "walls3d", CylinderWall: "cylinders"}[kinds.pop()]
    if getattr(game_con, slot) is not None or getattr(game_con, {"walls": "player_walls", "walls3d": "player_walls3d", "cylinders": "player_cylinders"}[slot]):
        raise AlgamesError("only one wall set of each kind per GameConstraintValues is supported")
    setattr(game_con, slot, walls)


def add_circle_constraint(game_con, *args):
    """add_circle_constraint!(game_con, xc, yc, radius) (constraints_methods.jl:141-148, every player) or
    add_circle_constraint!(game_con, i, xc, yc, radius) (:121-139, player i only, 1-based)."""
    player = None
    if len(args) == 4:
        player, args = int(args[0]), args[1:]
        if not 1 <= player <= game_con.probsize.p:
            raise ValueError("add_circle_constraint: player index out of range")
    xc, yc, radius = (np.asarray(a, dtype=np.float64) for a in args)
    if not (xc.shape == yc.shape == radius.shape and xc.ndim == 1):
        raise ValueError("xc, yc, radius must be vectors of equal length")
    if player is not None:
        if game_con.circles is not None:
            raise AlgamesError("per-player circles cannot be combined with an all-player circle set")
        old = game_con.player_circles.get(player)
        game_con.player_circles[player] = (xc, yc, radius) if old is None else tuple(np.concatenate([o, a]) for o, a in zip(old, (xc, yc, radius)))
        return
    if game_con.circles is not None or game_con.player_circles:
        raise AlgamesError("only one circle set per GameConstraintValues is supported")
    game_con.circles = (xc, yc, radius)


# --------------------------------------------------------------------------------------------------
# PrimalDualTraj / Statistics views
# --------------------------------------------------------------------------------------------------
class PrimalDualTraj:
    """Batched view of a primal-dual trajectory (src/struct/primal_dual_traj.jl:5-23).
    states (B,N,n), controls (B,N-1,m) in joint order, duals (B,p,N-1,n)."""

    def __init__(self, states, controls, duals):
        self.states, self.controls, self.duals = states, controls, duals


class Statistics:
    """Per-game `Statistics` history (src/struct/statistics.jl:5-15) as numpy record arrays; the reference's field names
    (`outer_iter`, `res`, `Δ_traj`, `dyn_vio`, `con_vio`, `sta_vio`, `opt_vio`) are views of game 0's history, the `*_vio`
    entries being the `.max` of the reference's violation objects.  `t_elap[r]` is the duration (seconds) of the inner iteration
    that preceded record r, measured on the device with the 100 MHz real-time counter (0 for the first record of a solve)."""

    def __init__(self, summary, history_fn):
        self.summary = summary
        self._history_fn = history_fn

    def history(self, game=0):
        return self._history_fn(game)

    @property
    def iter(self):
        return self.summary["records"]

    def _col(self, name, game=0):
        return np.array(self.history(game)[name])

    outer_iter = property(lambda self: self._col("outer"))
    res = property(lambda self: self._col("res"))
    Δ_traj = property(lambda self: self._col("delta"))
    dyn_vio = property(lambda self: self._col("dyn_vio"))
    con_vio = property(lambda self: self._col("con_vio"))
    sta_vio = property(lambda self: self._col("sta_vio"))
    opt_vio = property(lambda self: self._col("opt_vio"))
    t_elap = property(lambda self: self._col("t_elap"))


# --------------------------------------------------------------------------------------------------
# Printers (src/utils.jl:37-84).  The solver runs on the device, so the per-iteration table of `opts.inner_print` is printed
# from the recorded history after the solve (game 0), one line per Newton iteration as solver_methods.jl:100 prints them.
# --------------------------------------------------------------------------------------------------
def scn(a, digits=1):
    """scn(a; digits), utils.jl:63-84: mantissa/exponent string such as ' 1.2e+3'."""
    assert digits >= 0
    if a == 0:
        e, m = 0, 0.0
    else:
        e = int(np.floor(np.log(abs(a)) / np.log(10)))
        m = a * np.exp(-e * np.log(10))
    m = round(m, digits)
    if digits == 0:
        strm = str(int(np.floor(m)))
    else:
        strm = str(float(m))
        strm = strm + "0" * abs(2 + digits + (m < 0) - len(strm))
    return f"{' ' if a >= 0 else ''}{strm}e{'+' if e >= 0 else ''}{e}"


def display_solver_header():
    print("%-3s %-2s %-2s %-6s %-6s %-6s " % ("out", "in", "α", "Δ", "res", "reg"))


def display_solver_data(k, l, j, Δ, res_norm, reg):
    reg_x = reg.x if hasattr(reg, "x") else reg
    print("%-3s %-2s %-2s %-6s %-6s %-6s " % (k, l, j, "%.0e" % Δ, "%.0e" % res_norm, "%.0e" % reg_x))


# Plot recipes (src/plots/solver_plots.jl:18-37,83-125) as plain data: what the recipes hand to Plots.jl, no drawing here.
def recipe_traj(model, states):
    """recipe_traj(model, traj): per player the x and y series (state entries pz[i][1], pz[i][2]) of a (N, n) state array;
    returned twice like the recipe (scatter + path series)."""
    states = np.asarray(states)
    x = [states[:, i] for i in range(model.p)]
    y = [states[:, model.p + i] for i in range(model.p)]
    return [x, x], [y, y]


def recipe_violation(stats, game=0):
    """recipe_violation(stats): log10 of the four violation histories clipped at 1e-10 and the outer-iteration epochs; returns
    (series_x, series_y, labels) with one shaded band per outer iteration followed by dyn / con / sta / opt."""
    h = stats.history(game)
    it = len(h)
    ser = {k: np.log10(np.maximum(1e-10, h[k + "_vio"])) for k in ("dyn", "con", "sta", "opt")}
    y_max = max(v.max() for v in ser.values())
    epochs = np.asarray(h["outer"])
    xs, ys, labels, i_start = [], [], [], 1
    for k in range(1, int(epochs[-1]) + 1):
        i_end = int(np.nonzero(epochs == k)[0][-1]) + 1
        xs.append(np.linspace(i_start, i_end, it)); ys.append(np.full(it, y_max)); labels.append("")
        i_start = i_end + 1
    x = np.arange(1, it + 1)
    for k in ("dyn", "con", "sta", "opt"):
        xs.append(x); ys.append(ser[k]); labels.append(k)
    return xs, ys, labels


def _print_history(prob):
    """What `opts.inner_print` shows (solver_methods.jl:36,100), replayed from game 0's history: a record is written at the
    top of every inner iteration; the line of an iteration carries that iteration's step (Δ is the next record's Δ_traj)."""
    h = prob.stats.history(0)
    display_solver_header()
    l = 0
    for idx in range(len(h) - 1):
        l = l + 1 if idx > 0 and h["outer"][idx] == h["outer"][idx - 1] else 1
        if h["ls_j"][idx] == 0:
            continue                                       # a record without a Newton step (gate / final record)
        display_solver_data(int(h["outer"][idx]), l, int(h["ls_j"][idx]), h["delta"][idx + 1], h["res"][idx], prob.opts.reg_0 * l ** 4)


# --------------------------------------------------------------------------------------------------
# GameProblem (src/problem/problem.jl:19-53)
# --------------------------------------------------------------------------------------------------
class GameProblem:
    def __init__(self, N, dt, x0, model, opts, game_obj, game_con, backend=None, device=0, game_id0=0):
        self.probsize = ProblemSize(N, model)
        self.model, self.opts, self.game_obj, self.game_con = model, opts, game_obj, game_con
        self.dt = dt
        x0 = np.asarray(x0, dtype=np.float64)
        self.single = x0.ndim == 1
        self.x0 = np.ascontiguousarray(x0.reshape(-1, model.n))
        self.B = self.x0.shape[0]
        self.game_id0 = game_id0
        lib = backend if backend is not None else hip_lib()
        self.batch = Batch(lib, model.model_id, model.p, N, dt, self.B, d=model.d, device=device)
        if isinstance(model, BicycleGame):
            self.batch.set_bicycle(model.lf, model.lr)
        if isinstance(model, QuadrotorGame) and model.mass != 0.5:
            self.batch.set_quadrotor(model.mass)
        self.batch.set_x0(self.x0)
        self.batch.set_lqr(game_obj.Qdiag, game_obj.Rdiag, game_obj.xf, game_obj.uf)
        if game_obj.collision_radius is not None:
            self.batch.add_collision_cost(game_obj.collision_radius, game_obj.collision_μ)
        if game_con.collision_radius is not None:
            if game_con.spherical:
                self.batch.add_spherical_collision_avoidance(game_con.collision_radius)
            else:
                self.batch.add_collision_avoidance(game_con.collision_radius)
        for (i, j) in sorted(game_con.collision_pairs):
            self.batch.add_collision_avoidance_pair(i - 1, j - 1, game_con.collision_pairs[(i, j)], spherical=game_con.spherical)
        if game_con.u_max is not None:
            self.batch.add_control_bound(game_con.u_max, game_con.u_min)
        for i in sorted(game_con.state_bounds):
            self.batch.add_state_bound(i - 1, *game_con.state_bounds[i])
        if game_con.walls:
            w = game_con.walls
            self.batch.add_wall_constraint([a.p1[0] for a in w], [a.p1[1] for a in w], [a.p2[0] for a in w],
                                           [a.p2[1] for a in w], [a.v[0] for a in w], [a.v[1] for a in w])
        if game_con.circles is not None:
            self.batch.add_circle_constraint(*game_con.circles)
        for i in sorted(game_con.player_walls):
            w = game_con.player_walls[i]
            self.batch.add_wall_constraint_player(i - 1, [a.p1[0] for a in w], [a.p1[1] for a in w], [a.p2[0] for a in w],
                                                  [a.p2[1] for a in w], [a.v[0] for a in w], [a.v[1] for a in w])
        for i in sorted(game_con.player_circles):
            self.batch.add_circle_constraint_player(i - 1, *game_con.player_circles[i])
        if game_con.walls3d:
            w = game_con.walls3d
            self.batch.add_wall3d_constraint([a.p1 for a in w], [a.p2 for a in w], [a.p3 for a in w], [a.v for a in w])
        if game_con.cylinders:
            c = game_con.cylinders
            self.batch.add_cylinder_constraint([a.p for a in c], [a.v for a in c], [a.l for a in c], [a.r for a in c])
        for i in sorted(game_con.player_walls3d):
            w = game_con.player_walls3d[i]
            self.batch.add_wall3d_constraint_player(i - 1, [a.p1 for a in w], [a.p2 for a in w], [a.p3 for a in w], [a.v for a in w])
        for i in sorted(game_con.player_cylinders):
            c = game_con.player_cylinders[i]
            self.batch.add_cylinder_constraint_player(i - 1, [a.p for a in c], [a.v for a in c], [a.l for a in c], [a.r for a in c])
        self.stats = None
        game_con.active_set_tolerance = opts.active_set_tolerance      # set_constraint_params!, game_constraints.jl:37
        self._sync_options()       # set_constraint_params!(game_con, opts), problem.jl:49

    def _sync_options(self):
        # `opts` is shared by reference in Julia and read at solve time (tests mutate it after construction)
        self.batch.set_options(**self.opts.to_abi())

    def _traj(self, which):
        X, U, L = self.batch.split_traj(self.batch.get_traj(which))
        return PrimalDualTraj(X, U, L)

    @property
    def pdtraj(self):
        return self._traj(ALG_TRAJ_PD)

    @property
    def pdtraj_trial(self):
        return self._traj(ALG_TRAJ_TRIAL)

    @property
    def Δpdtraj(self):
        return self._traj(ALG_TRAJ_DELTA)

    def set_pdtraj(self, states=None, controls=None, duals=None):
        cur = self.pdtraj
        X = cur.states if states is None else np.broadcast_to(np.asarray(states, dtype=np.float64), cur.states.shape)
        U = cur.controls if controls is None else np.broadcast_to(np.asarray(controls, dtype=np.float64), cur.controls.shape)
        L = cur.duals if duals is None else np.broadcast_to(np.asarray(duals, dtype=np.float64), cur.duals.shape)
        self.batch.set_traj(self.batch.join_traj(np.array(X), np.array(U), np.array(L)))


# --------------------------------------------------------------------------------------------------
# Solver methods (src/problem/solver_methods.jl:5-125, src/problem/global_quantities.jl:1-193)
# --------------------------------------------------------------------------------------------------
def init_traj_host(prob):
    """init_traj!(prob.pdtraj; x0, f = opts.f_init, amplitude = opts.amplitude_init, s = opts.shift) (primal_dual_traj.jl:29-44,
    called at solver_methods.jl:13) for a CALLER-SUPPLIED generator: `opts.f_init` is a callable `f(size) -> array of that size`
    (the reference calls `f(SVector{n+m,T})` per knot and `f(SVector{n,T})` per player and step; `zeros`, `ones`, `randn` are the
    usual choices).  Calls are made game by game in the reference's order: knots k = 1..N (those with k + s > N), then players i,
    steps k = 1..N-1 (those with k + s > N-1).  The default `"rand"` never comes here: the device generates it (alg_init_traj).
    The result is stored as the handle's pdtraj; the solve then runs with init = 0, which still rolls the states out
    (solver_methods.jl:17), so only x_1 = x0, the controls and the duals of this guess matter.  The layout keeps no control at knot
    N (the reference's `pr[N]` carries one that nothing reads): when the shift copies knot N, its control is set to zero -- no draw
    is consumed for it, so a stateful generator stays aligned with the reference's call sequence.
    Every game is one `newton_solve!` of the reference, which calls `Random.seed!(opts.seed)` first (solver_methods.jl:9): NumPy's
    global generator is seeded the same way before each game's draws, so `f_init = np.random.randn` gives every game the stream the
    Julia shim gives it (generators that carry their own state, e.g. a `default_rng(...)` method, are not touched by this).  The
    caller's global NumPy generator state is saved before the loop and restored after it: the seeding is a detail of this call, not a
    side effect on the caller's stream.  (Deliberate deviation, stated: where the shift copies knot N the reference copies the STALE
    control `pr[N]` carries, primal_dual_traj.jl:35; the layout here has no such slot and stores zero.)"""
    b, o = prob.batch, prob.opts
    f, a, s = o.f_init, float(o.amplitude_init), int(min(o.shift, 2 ** 30))
    X, U, L = b.split_traj(b.get_traj(0))
    B, N, n, m, p = X.shape[0], b.N, b.n, b.m, b.p
    draw = lambda size: a * np.asarray(f(size), dtype=np.float64).reshape(size)
    rng_state = np.random.get_state()
    try:
        _init_traj_host_fill(o, X, U, L, B, N, n, m, p, s, draw)
    finally:
        np.random.set_state(rng_state)
    X[:, 0] = b.get_x0()
    b.set_traj(b.join_traj(X, U, L))


def _init_traj_host_fill(o, X, U, L, B, N, n, m, p, s, draw):
    for g in range(B):
        np.random.seed(int(o.seed) % (2 ** 32))                     # Random.seed!(opts.seed), solver_methods.jl:9
        for k in range(1, N + 1):                                   # 1-based like the reference
            if k + s <= N:
                X[g, k - 1] = X[g, k + s - 1]
                if k <= N - 1:
                    U[g, k - 1] = U[g, k + s - 1] if k + s <= N - 1 else 0.0
            else:
                z = draw(n + m)
                X[g, k - 1] = z[:n]
                if k <= N - 1:
                    U[g, k - 1] = z[n:]
        for i in range(p):
            for k in range(1, N):
                L[g, i, k - 1] = L[g, i, k + s - 1] if k + s <= N - 1 else draw(n)


def newton_solve(prob, init=True):
    """newton_solve!(prob) for every game of the batch (solver_methods.jl:5-65).
    init=False keeps the stored controls/duals as the initial guess (explicit warm start).
    A `sharding.ShardedGameProblem` (batch split over several devices) is solved on all of its devices concurrently."""
    if hasattr(prob, "shards"):
        from . import sharding
        if init and any(callable(q.opts.f_init) for q in prob.shards):
            for q in prob.shards:
                init_traj_host(q)
            init = False
        return sharding.newton_solve_sharded(prob, init=init)
    prob._sync_options()
    if init and callable(prob.opts.f_init):
        init_traj_host(prob)
        init = False
    summary = prob.batch.newton_solve(init=init, game_id0=prob.game_id0)
    prob.stats = Statistics(summary, prob.batch.get_history)
    if prob.opts.inner_print:
        _print_history(prob)
    return None


def residual(prob, which=ALG_TRAJ_PD, reg=0.0):
    """residual!(prob, pdtraj) (+ regularize_residual!): returns core.res, shape (B, S), vertical order."""
    prob._sync_options()
    return prob.batch.residual(which, reg)[0]


def residual_norm(prob, which=ALG_TRAJ_PD):
    """residual_norm(prob, pdtraj), global_quantities.jl:88-97."""
    prob._sync_options()
    return prob.batch.residual(which, 0.0, want_res=False)[1]


def residual_jacobian(prob, reg=0.0, games=None):
    """residual_jacobian! + regularize_residual_jacobian!: dense (B, S, S), [row vertical, col horizontal]; games = (first, count)
    builds only that range of the batch ((count, S, S))."""
    prob._sync_options()
    return prob.batch.residual_jacobian(reg, games)


def inner_iteration(prob, LS_count, t_elap, Δ, k, l):
    """inner_iteration(prob, LS_count, t_elap, Δ, k, l) (solver_methods.jl:67-103).
    Returns (LS_count (B,), control_flow (B,) of 'continue'/'break', Δ (B,), info)."""
    prob._sync_options()
    info = prob.batch.newton_step(k, l, delta=Δ)
    LS = np.where(info["ls_failed"] == 1, np.asarray(LS_count) + 1, 0)
    LS = np.where(info["ls_j"] == 0, np.asarray(LS_count), LS)
    flow = np.where(info["control_flow"] == 1, "break", "continue")
    return LS, flow, info["delta"], info


def line_search(prob, res_norm, reg=0.0):
    """line_search(prob, res_norm) (solver_methods.jl:105-125) -> (α, j)."""
    prob._sync_options()
    return prob.batch.line_search(res_norm, reg)


def violation_profile(prob):
    """The `.vio` vectors of dynamics_violation / control_violation / state_violation / optimality_violation at pdtraj
    (violations.jl:5-26, 41-67, 86-114, 140-168): dict(dyn (B, N-1), con (B, N-1), sta (B, N), opt (B, N)); their maxima over the knots
    are the `*_vio` entries record! stores."""
    prob._sync_options()
    return prob.batch.violation_profile()


def dynamics_violation(prob):
    return prob.batch.record()["dyn_vio"]


def control_violation(prob):
    return prob.batch.record()["con_vio"]


def state_violation(prob):
    return prob.batch.record()["sta_vio"]


def optimality_violation(prob):
    return prob.batch.record()["opt_vio"]


# --------------------------------------------------------------------------------------------------
# Iterated best response (src/struct/options.jl:123-136, src/problem/solver_methods.jl:133-289)
# --------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class IBROptions:
    ibr_iter: int = 100
    ordering: list = dataclasses.field(default_factory=lambda: list(range(1, 101)))   # 1-based player ids, like the reference
    Δ_min: float = 1e-9
    live_plotting: bool = False


def ibr_newton_solve(prob, i=None, ibr_opts=None, init=True):
    """ibr_newton_solve!(prob; ibr_opts) (solver_methods.jl:133-169), or with a 1-based player index `i`
    ibr_newton_solve!(prob, i) (:171-228) on the stored trajectory."""
    prob._sync_options()
    if i is not None:
        summary = prob.batch.ibr_solve_player(i - 1)
    else:
        o = ibr_opts if ibr_opts is not None else IBROptions()
        order = [j - 1 for j in list(o.ordering)[:prob.probsize.p]]
        if init and callable(prob.opts.f_init):
            init_traj_host(prob)
            init = False
        summary = prob.batch.ibr_newton_solve(o.ibr_iter, order, o.Δ_min, init=init, game_id0=prob.game_id0)
    prob.stats = Statistics(summary, prob.batch.get_history)
    return None


# --------------------------------------------------------------------------------------------------
# Receding-horizon loop (BASELINE config 5).  Not in the reference: Algames.jl v0.1.6 only has the warm-start hooks
# `opts.shift` / `opts.dual_reset` (options.jl:16-17, primal_dual_traj.jl:35-39, solver_methods.jl:25).  Builder-defined
# (SURVEY.md 8(d) C5): solve; x0 <- RK2(x_1, u_1); next solve warm-started with shift = 1 and dual_reset = false.
# --------------------------------------------------------------------------------------------------
def mpc_solve(prob, steps, record_states=False, fused=True):
    """Runs `steps` receding-horizon solves for every game of the batch.  Returns (newton_iters (B,), converged (B,),
    states (steps+1, B, n) or None).  fused=True: one launch, every game runs its own loop (alg_mpc_solve); fused=False:
    one newton_solve! launch + one advance launch per MPC step (the batch waits for its slowest game at every step).
    No host synchronisation happens inside the loop unless record_states is set."""
    b = prob.batch
    b.mpc_totals(reset=True)
    prob._sync_options()
    if fused:
        states = b.mpc_solve(steps, prob.game_id0, record_states)
        it, cv = b.mpc_totals()
        return it, cv, states
    shift0, reset0 = prob.opts.shift, prob.opts.dual_reset
    states = [b.get_x0()] if record_states else None
    try:
        for t in range(steps):
            if t == 1:
                prob.opts.shift, prob.opts.dual_reset = 1, False
                prob._sync_options()
            b.newton_solve_async(init=True, game_id0=prob.game_id0 + t * 1000003)
            b.mpc_advance()
            if record_states:
                states.append(b.get_x0())
        it, cv = b.mpc_totals()
    finally:
        prob.opts.shift, prob.opts.dual_reset = shift0, reset0
        prob._sync_options()
    return it, cv, (np.stack(states) if record_states else None)
