"""ctypes mirror of include/algames_hip.h.

`CLib(path, prefix)` binds every entry point the header declares.  The product binds
libalgames_hip.so with prefix ``alg_``; the CPU oracle (oracle/, test infrastructure) exports the
same signatures with prefix ``orc_`` and is bound by oracle/oracle.py through this same class.
"""
import ctypes as C
import numpy as np

ALG_OK = 0
ALG_MODEL_DOUBLE_INTEGRATOR = 0
ALG_MODEL_UNICYCLE = 1
ALG_MODEL_BICYCLE = 2
ALG_MODEL_QUADROTOR = 3
ALG_TRAJ_PD, ALG_TRAJ_TRIAL, ALG_TRAJ_DELTA = 0, 1, 2
ALG_STATUS_OK, ALG_STATUS_SINGULAR, ALG_STATUS_NAN = 0, 1, 2


class alg_desc(C.Structure):
    _fields_ = [("model", C.c_int32), ("p", C.c_int32), ("d", C.c_int32), ("N", C.c_int32),
                ("dt", C.c_double), ("batch", C.c_int32), ("device", C.c_int32)]


class alg_options(C.Structure):
    _fields_ = [("amplitude_init", C.c_double), ("shift", C.c_int32), ("regularize", C.c_int32),
                ("reg_0", C.c_double), ("alpha_decrease", C.c_double), ("beta", C.c_double),
                ("ls_iter", C.c_int32), ("dual_reset", C.c_int32), ("delta_min", C.c_double),
                ("rho_0", C.c_double), ("rho_increase", C.c_double), ("rho_max", C.c_double),
                ("lambda_max", C.c_double), ("alpha_dual", C.c_double),
                ("alphax_dual", C.c_double * 10),
                ("eps_dyn", C.c_double), ("eps_sta", C.c_double), ("eps_con", C.c_double),
                ("eps_opt", C.c_double), ("outer_iter", C.c_int32), ("inner_iter", C.c_int32),
                ("seed", C.c_int64)]


class alg_record(C.Structure):
    _fields_ = [("outer", C.c_int32), ("ls_j", C.c_int32), ("alpha", C.c_double),
                ("res", C.c_double), ("delta", C.c_double), ("dyn_vio", C.c_double),
                ("con_vio", C.c_double), ("sta_vio", C.c_double), ("opt_vio", C.c_double), ("t_elap", C.c_double)]


class alg_game_stats(C.Structure):
    _fields_ = [("status", C.c_int32), ("outer_iters", C.c_int32), ("newton_iters", C.c_int32),
                ("records", C.c_int32), ("converged", C.c_int32), ("ls_failures", C.c_int32),
                ("refinements", C.c_int32), ("reserved", C.c_int32), ("last", alg_record)]


class alg_step_info(C.Structure):
    _fields_ = [("status", C.c_int32), ("control_flow", C.c_int32), ("ls_j", C.c_int32),
                ("ls_failed", C.c_int32), ("alpha", C.c_double), ("delta", C.c_double),
                ("rec", alg_record)]


record_dtype = np.dtype([("outer", "<i4"), ("ls_j", "<i4"), ("alpha", "<f8"), ("res", "<f8"),
                         ("delta", "<f8"), ("dyn_vio", "<f8"), ("con_vio", "<f8"),
                         ("sta_vio", "<f8"), ("opt_vio", "<f8"), ("t_elap", "<f8")])
game_stats_dtype = np.dtype([("status", "<i4"), ("outer_iters", "<i4"), ("newton_iters", "<i4"),
                             ("records", "<i4"), ("converged", "<i4"), ("ls_failures", "<i4"),
                             ("refinements", "<i4"), ("reserved", "<i4"), ("last", record_dtype)])
step_info_dtype = np.dtype([("status", "<i4"), ("control_flow", "<i4"), ("ls_j", "<i4"),
                            ("ls_failed", "<i4"), ("alpha", "<f8"), ("delta", "<f8"),
                            ("rec", record_dtype)])
assert record_dtype.itemsize == C.sizeof(alg_record)
assert game_stats_dtype.itemsize == C.sizeof(alg_game_stats)
assert step_info_dtype.itemsize == C.sizeof(alg_step_info)

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)

# name -> (restype, argtypes); exactly the functions declared in include/algames_hip.h
SIGNATURES = {
    "last_error": (C.c_char_p, []),
    "default_options": (None, [C.POINTER(alg_options)]),
    "dims": (C.c_int, [C.POINTER(alg_desc), _I, _I, _I, _I, _I, _I]),
    "create": (C.c_int, [C.POINTER(alg_desc), C.POINTER(_P)]),
    "destroy": (None, [_P]),
    "set_options": (C.c_int, [_P, C.POINTER(alg_options)]),
    "get_options": (C.c_int, [_P, C.POINTER(alg_options)]),
    "set_stream": (C.c_int, [_P, _P]),
    "set_waves_per_game": (C.c_int, [_P, C.c_int32]),
    "get_waves_per_game": (C.c_int, [_P, _I]),
    "set_refinement": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double]),
    "get_refinement": (C.c_int, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "get_direction_gate": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "set_line_search_groups": (C.c_int, [_P, C.c_int32]),
    "get_line_search_groups": (C.c_int, [_P, _I]),
    "set_handoff": (C.c_int, [_P, C.c_int32]),
    "get_handoff": (C.c_int, [_P, _I, _I]),
    "set_x0": (C.c_int, [_P, _D]),
    "set_lqr": (C.c_int, [_P, _D, _D, _D, _D, C.c_int32]),
    "add_collision_cost": (C.c_int, [_P, _D, _D]),
    "add_collision_avoidance": (C.c_int, [_P, _D]),
    "add_collision_avoidance_pair": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double]),
    "add_spherical_collision_avoidance_pair": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double]),
    "add_control_bound": (C.c_int, [_P, _D, _D]),
    "set_bicycle": (C.c_int, [_P, C.c_double, C.c_double]),
    "set_quadrotor": (C.c_int, [_P, C.c_double]),
    "add_state_bound": (C.c_int, [_P, C.c_int32, _D, _D]),
    "add_wall_constraint": (C.c_int, [_P, C.c_int32, _D, _D, _D, _D, _D, _D]),
    "add_circle_constraint": (C.c_int, [_P, C.c_int32, _D, _D, _D]),
    "add_wall_constraint_player": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D, _D, _D, _D, _D]),
    "add_circle_constraint_player": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D, _D]),
    "add_spherical_collision_avoidance": (C.c_int, [_P, _D]),
    "add_wall3d_constraint": (C.c_int, [_P, C.c_int32, _D, _D, _D, _D]),
    "add_cylinder_constraint": (C.c_int, [_P, C.c_int32, _D, _I, _D, _D]),
    "add_wall3d_constraint_player": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D, _D, _D]),
    "add_cylinder_constraint_player": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _I, _D, _D]),
    "get_con_len": (C.c_int, [_P, _I]),
    "set_traj": (C.c_int, [_P, C.c_int32, _D]),
    "get_traj": (C.c_int, [_P, C.c_int32, _D]),
    "set_con_duals": (C.c_int, [_P, _D, _D]),
    "get_con_duals": (C.c_int, [_P, _D, _D]),
    "init_traj": (C.c_int, [_P, C.c_int64, C.c_int32]),
    "rollout": (C.c_int, [_P, C.c_int32]),
    "residual": (C.c_int, [_P, C.c_int32, C.c_double, _D, _D]),
    "residual_jacobian": (C.c_int, [_P, C.c_double, _D]),
    "residual_jacobian_games": (C.c_int, [_P, C.c_double, C.c_int32, C.c_int32, _D]),
    "release_scratch": (C.c_int, [_P]),
    "get_violation_profile": (C.c_int, [_P, _D, _D, _D, _D]),
    "newton_direction": (C.c_int, [_P, C.c_double, _D, _I]),
    "line_search": (C.c_int, [_P, C.c_double, _D, _D, _I]),
    "update_traj": (C.c_int, [_P, C.c_int32, C.c_int32, _D]),
    "record_stats": (C.c_int, [_P, _P]),
    "reset_con": (C.c_int, [_P]),
    "dual_penalty_update": (C.c_int, [_P, _D]),
    "newton_step": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "newton_solve": (C.c_int, [_P, C.c_int32, C.c_int64, _P]),
    "newton_solve_async": (C.c_int, [_P, C.c_int32, C.c_int64]),
    "get_stats": (C.c_int, [_P, _P]),
    "get_history": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _I]),
    "synchronize": (C.c_int, [_P]),
    "debug_check_guards": (C.c_int, [_P]),
    "ibr_solve_player": (C.c_int, [_P, C.c_int32, _P]),
    "ibr_newton_solve": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, _I, C.c_double, _P]),
    "mpc_advance": (C.c_int, [_P]),
    "mpc_totals": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    "mpc_solve": (C.c_int, [_P, C.c_int32, C.c_int64, _D]),
}


class AlgamesError(RuntimeError):
    pass


def _dptr(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_D)


def _iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_I)


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


class CLib:
    """A loaded shared library exporting the ABI of include/algames_hip.h under `prefix`."""

    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        self.missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(self.dll, prefix + name)
            except AttributeError:
                self.missing.append(prefix + name)
                continue
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        if self.missing:
            raise AlgamesError(f"{path}: missing ABI symbols {self.missing}")

    def check(self, rc):
        if rc != ALG_OK:
            msg = self.last_error()
            raise AlgamesError((msg or b"").decode() + f" (code {rc})")

    def default_opts(self):
        o = alg_options()
        self.default_options(C.byref(o))
        return o

    def sizes(self, desc):
        v = [C.c_int32() for _ in range(6)]
        self.check(self.dims(C.byref(desc), *[C.byref(x) for x in v]))
        n, m, mi, S, traj_len, con_len = [x.value for x in v]
        return dict(n=n, m=m, mi=mi, S=S, traj_len=traj_len, con_len=con_len)


class Batch:
    """Thin object wrapper over one alg_handle: a batch of B games sharing structure.

    This is the level the Julia shim binds; the reference-shaped host API
    (GameProblem, newton_solve, ...) in algames_jl_amd/__init__.py is built on it.
    """

    def __init__(self, lib, model, p, N, dt, batch, d=2, device=0):
        self.lib = lib
        self.desc = alg_desc(model, p, d, N, dt, batch, device)
        sz = lib.sizes(self.desc)
        self.n, self.m, self.mi, self.S = sz["n"], sz["m"], sz["mi"], sz["S"]
        self.traj_len, self.con_len = sz["traj_len"], sz["con_len"]
        self.ni = self.n // p
        self.p, self.N, self.dt, self.B, self.d = p, N, dt, batch, d
        self.b = self.n + self.m + p * self.n
        h = _P()
        lib.check(lib.create(C.byref(self.desc), C.byref(h)))
        self.h = h
        self.opts = lib.default_opts()

    def close(self):
        if getattr(self, "h", None):
            self.lib.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------------------------
    def set_options(self, **kw):
        for k, v in kw.items():
            if k == "alphax_dual":
                for i, x in enumerate(v):
                    self.opts.alphax_dual[i] = x
            else:
                if not hasattr(self.opts, k):
                    raise AttributeError(k)
                setattr(self.opts, k, v)
        self.lib.check(self.lib.set_options(self.h, C.byref(self.opts)))

    def set_waves_per_game(self, nw):
        """0 = automatic, 1 = one game per wavefront, 2 / 4 = a team of wavefronts per game (fused solver kernels)."""
        self.lib.check(self.lib.set_waves_per_game(self.h, int(nw)))

    def get_waves_per_game(self):
        v = C.c_int32()
        self.lib.check(self.lib.get_waves_per_game(self.h, C.byref(v)))
        return v.value

    def set_refinement(self, max_steps=None, tol=None, mu_tight=None):
        """Iterative refinement of the Newton direction (alg_set_refinement): the opt-u rows of every direction are evaluated and the
        direction is corrected (at most `max_steps` correction solves) while their row-wise backward error exceeds `tol` (relaxed up to
        256 x while the game's largest penalty is below `mu_tight`); max_steps = 0 switches gate and refinement off.  None keeps a value."""
        ms0, tol0, mu0 = self.get_refinement()
        self.lib.check(self.lib.set_refinement(self.h, int(ms0 if max_steps is None else max_steps), float(tol0 if tol is None else tol),
                                               float(mu0 if mu_tight is None else mu_tight)))

    def get_refinement(self):
        ms = C.c_int32(); tol = C.c_double(); mu = C.c_double()
        self.lib.check(self.lib.get_refinement(self.h, C.byref(ms), C.byref(tol), C.byref(mu)))
        return ms.value, tol.value, mu.value

    def set_line_search_groups(self, on):
        """Line search of the team / one-wavefront unicycle kernels: step sizes tried four at a time once the first one was rejected (True,
        default) or one after another (False); bit-identical norms either way (alg_set_line_search_groups)."""
        self.lib.check(self.lib.set_line_search_groups(self.h, 1 if on else 0))

    def get_line_search_groups(self):
        v = C.c_int32()
        self.lib.check(self.lib.get_line_search_groups(self.h, C.byref(v)))
        return bool(v.value)

    def set_handoff(self, iters):
        """Straggler hand-off for heterogeneous batches (alg_set_handoff): games that need more than `iters` inner iterations in the
        one-wavefront kernel park and a second launch finishes them with the team kernel; 0 = off (default)."""
        self.lib.check(self.lib.set_handoff(self.h, int(iters)))

    def get_handoff(self):
        """(budget, number of games the most recent solve handed over)"""
        k = C.c_int32(); n = C.c_int32()
        self.lib.check(self.lib.get_handoff(self.h, C.byref(k), C.byref(n)))
        return k.value, n.value

    def get_direction_gate(self):
        """(B, 3): [max |rho|, row-wise backward error omega, largest row scale] of the opt-u rows of the last Newton direction
        (alg_get_direction_gate)."""
        out = np.zeros((self.B, 3))
        self.lib.check(self.lib.get_direction_gate(self.h, _dptr(out)))
        return out

    def set_stream(self, stream_ptr):
        self.lib.check(self.lib.set_stream(self.h, _P(stream_ptr)))

    def set_x0(self, x0):
        x0 = _f64(x0)
        if x0.ndim == 1:
            x0 = np.ascontiguousarray(np.broadcast_to(x0, (self.B, self.n)))
        x0 = _f64(x0, (self.B, self.n))
        self.lib.check(self.lib.set_x0(self.h, _dptr(x0)))

    def set_lqr(self, Qdiag, Rdiag, xf, uf):
        Qdiag, Rdiag, xf, uf = _f64(Qdiag), _f64(Rdiag), _f64(xf), _f64(uf)
        per_game = int(Qdiag.ndim == 3)
        lead = (self.B,) if per_game else ()
        Qdiag = _f64(Qdiag, lead + (self.p, self.ni)); xf = _f64(np.broadcast_to(xf, lead + (self.p, self.ni)))
        Rdiag = _f64(Rdiag, lead + (self.p, self.mi)); uf = _f64(np.broadcast_to(uf, lead + (self.p, self.mi)))
        self.lib.check(self.lib.set_lqr(self.h, _dptr(Qdiag), _dptr(Rdiag), _dptr(xf), _dptr(uf), per_game))

    def add_collision_cost(self, radius, mu):
        self.lib.check(self.lib.add_collision_cost(self.h, _dptr(_f64(radius, (self.p,))), _dptr(_f64(mu, (self.p,)))))

    def add_collision_avoidance(self, radius):
        r = _f64(np.broadcast_to(np.asarray(radius, dtype=np.float64), (self.p,)))
        self.lib.check(self.lib.add_collision_avoidance(self.h, _dptr(r)))

    def add_control_bound(self, u_max, u_min):
        self.lib.check(self.lib.add_control_bound(self.h, _dptr(_f64(u_max, (self.m,))), _dptr(_f64(u_min, (self.m,)))))

    def _refresh_con_len(self):
        v = C.c_int32()
        self.lib.check(self.lib.get_con_len(self.h, C.byref(v)))
        self.con_len = v.value

    def set_bicycle(self, lf, lr):
        self.lib.check(self.lib.set_bicycle(self.h, float(lf), float(lr)))

    def set_quadrotor(self, mass):
        self.lib.check(self.lib.set_quadrotor(self.h, float(mass)))

    def add_state_bound(self, player, x_max, x_min):
        self.lib.check(self.lib.add_state_bound(self.h, int(player), _dptr(_f64(x_max, (self.n,))), _dptr(_f64(x_min, (self.n,)))))
        self._refresh_con_len()

    def add_wall_constraint(self, x1, y1, x2, y2, xv, yv):
        arrs = [_f64(a) for a in (x1, y1, x2, y2, xv, yv)]
        self.lib.check(self.lib.add_wall_constraint(self.h, len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_circle_constraint(self, xc, yc, radius):
        arrs = [_f64(a) for a in (xc, yc, radius)]
        self.lib.check(self.lib.add_circle_constraint(self.h, len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_wall_constraint_player(self, player, x1, y1, x2, y2, xv, yv):
        arrs = [_f64(a) for a in (x1, y1, x2, y2, xv, yv)]
        self.lib.check(self.lib.add_wall_constraint_player(self.h, int(player), len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_circle_constraint_player(self, player, xc, yc, radius):
        arrs = [_f64(a) for a in (xc, yc, radius)]
        self.lib.check(self.lib.add_circle_constraint_player(self.h, int(player), len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_collision_avoidance_pair(self, i, j, radius, spherical=False):
        """One ordered pair (0-based players), own radius: add_collision_avoidance!(game_con, i, j, radius)."""
        fn = self.lib.add_spherical_collision_avoidance_pair if spherical else self.lib.add_collision_avoidance_pair
        self.lib.check(fn(self.h, int(i), int(j), float(radius)))
        if spherical:
            self._refresh_con_len()

    def add_spherical_collision_avoidance(self, radius):
        r = _f64(np.broadcast_to(np.asarray(radius, dtype=np.float64), (self.p,)))
        self.lib.check(self.lib.add_spherical_collision_avoidance(self.h, _dptr(r)))

    def add_wall3d_constraint(self, p1, p2, p3, v):
        arrs = [_f64(np.asarray(a, dtype=np.float64).reshape(-1, 3)) for a in (p1, p2, p3, v)]
        self.lib.check(self.lib.add_wall3d_constraint(self.h, len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_cylinder_constraint(self, p, axis, l, r):
        p = _f64(np.asarray(p, dtype=np.float64).reshape(-1, 3))
        ax = np.ascontiguousarray(axis, dtype=np.int32)
        self.lib.check(self.lib.add_cylinder_constraint(self.h, len(p), _dptr(p), ax.ctypes.data_as(_I), _dptr(_f64(l, (len(p),))), _dptr(_f64(r, (len(p),)))))
        self._refresh_con_len()

    def add_wall3d_constraint_player(self, player, p1, p2, p3, v):
        """add_wall_constraint!(game_con, i, walls::Vector{Wall3D}): player i (0-based) only."""
        arrs = [_f64(np.asarray(a, dtype=np.float64).reshape(-1, 3)) for a in (p1, p2, p3, v)]
        self.lib.check(self.lib.add_wall3d_constraint_player(self.h, int(player), len(arrs[0]), *[_dptr(a) for a in arrs]))
        self._refresh_con_len()

    def add_cylinder_constraint_player(self, player, p, axis, l, r):
        """add_wall_constraint!(game_con, i, walls::Vector{CylinderWall}): player i (0-based) only."""
        p = _f64(np.asarray(p, dtype=np.float64).reshape(-1, 3))
        ax = np.ascontiguousarray(axis, dtype=np.int32)
        self.lib.check(self.lib.add_cylinder_constraint_player(self.h, int(player), len(p), _dptr(p), ax.ctypes.data_as(_I), _dptr(_f64(l, (len(p),))), _dptr(_f64(r, (len(p),)))))
        self._refresh_con_len()

    # ---- data movement -----------------------------------------------------------------------
    def set_traj(self, z, which=ALG_TRAJ_PD):
        z = _f64(z, (self.B, self.traj_len))
        self.lib.check(self.lib.set_traj(self.h, which, _dptr(z)))

    def get_traj(self, which=ALG_TRAJ_PD):
        z = np.empty((self.B, self.traj_len))
        self.lib.check(self.lib.get_traj(self.h, which, _dptr(z)))
        return z

    def set_con_duals(self, lam=None, mu=None):
        lam = None if lam is None else _f64(lam, (self.B, self.con_len))
        mu = None if mu is None else _f64(mu, (self.B, self.con_len))
        self.lib.check(self.lib.set_con_duals(self.h, _dptr(lam), _dptr(mu)))

    def get_con_duals(self):
        lam = np.empty((self.B, self.con_len)); mu = np.empty((self.B, self.con_len))
        self.lib.check(self.lib.get_con_duals(self.h, _dptr(lam), _dptr(mu)))
        return lam, mu

    # ---- the path ----------------------------------------------------------------------------
    def init_traj(self, game_id0=0, use_shift=False):
        self.lib.check(self.lib.init_traj(self.h, game_id0, int(use_shift)))

    def rollout(self, which=ALG_TRAJ_PD):
        self.lib.check(self.lib.rollout(self.h, which))

    def residual(self, which=ALG_TRAJ_PD, reg=0.0, want_res=True):
        res = np.empty((self.B, self.S)) if want_res else None
        rn = np.empty(self.B)
        self.lib.check(self.lib.residual(self.h, which, reg, _dptr(res), _dptr(rn)))
        return res, rn

    def residual_jacobian(self, reg=0.0, games=None):
        """Dense KKT Jacobians [g, row, col]; games = (first, count) restricts the work and the memory to that range."""
        first, cnt = (0, self.B) if games is None else (int(games[0]), int(games[1]))
        jac = np.empty((cnt, self.S, self.S))
        self.lib.check(self.lib.residual_jacobian_games(self.h, reg, first, cnt, _dptr(jac)))
        return jac.transpose(0, 2, 1)     # column-major S x S per game -> [g, row, col]

    def release_scratch(self):
        self.lib.check(self.lib.release_scratch(self.h))

    def violation_profile(self):
        """Per-knot .vio vectors at pdtraj: dict(dyn (B, N-1), con (B, N-1), sta (B, N), opt (B, N))."""
        out = dict(dyn=np.empty((self.B, self.N - 1)), con=np.empty((self.B, self.N - 1)), sta=np.empty((self.B, self.N)), opt=np.empty((self.B, self.N)))
        self.lib.check(self.lib.get_violation_profile(self.h, _dptr(out["dyn"]), _dptr(out["con"]), _dptr(out["sta"]), _dptr(out["opt"])))
        return out

    def newton_direction(self, reg=0.0):
        delta = np.empty((self.B, self.S)); st = np.empty(self.B, dtype=np.int32)
        self.lib.check(self.lib.newton_direction(self.h, reg, _dptr(delta), _iptr(st)))
        return delta, st

    def line_search(self, res_norm, reg=0.0):
        rn = _f64(res_norm, (self.B,)); a = np.empty(self.B); j = np.empty(self.B, dtype=np.int32)
        self.lib.check(self.lib.line_search(self.h, reg, _dptr(rn), _dptr(a), _iptr(j)))
        return a, j

    def update_traj(self, alpha, target=ALG_TRAJ_PD, source=ALG_TRAJ_PD):
        a = _f64(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (self.B,)))
        self.lib.check(self.lib.update_traj(self.h, target, source, _dptr(a)))

    def record(self):
        rec = np.zeros(self.B, dtype=record_dtype)
        self.lib.check(self.lib.record_stats(self.h, rec.ctypes.data_as(_P)))
        return rec

    def reset_con(self):
        self.lib.check(self.lib.reset_con(self.h))

    def dual_penalty_update(self):
        vals = np.empty((self.B, self.con_len))
        self.lib.check(self.lib.dual_penalty_update(self.h, _dptr(vals)))
        return vals

    def newton_step(self, k_outer=1, l_inner=1, delta=None):
        """inner_iteration for every game; `delta` (B,) is the caller's Δ that record! stores (solver_methods.jl:75), default 0."""
        info = np.zeros(self.B, dtype=step_info_dtype)
        d = None if delta is None else _f64(np.broadcast_to(np.asarray(delta, dtype=np.float64), (self.B,)))
        self.lib.check(self.lib.newton_step(self.h, k_outer, l_inner, None if d is None else _dptr(d), info.ctypes.data_as(_P)))
        return info

    def newton_solve(self, init=True, game_id0=0):
        st = np.zeros(self.B, dtype=game_stats_dtype)
        self.lib.check(self.lib.newton_solve(self.h, int(init), game_id0, st.ctypes.data_as(_P)))
        return st

    def newton_solve_async(self, init=True, game_id0=0):
        self.lib.check(self.lib.newton_solve_async(self.h, int(init), game_id0))

    def get_stats(self):
        st = np.zeros(self.B, dtype=game_stats_dtype)
        self.lib.check(self.lib.get_stats(self.h, st.ctypes.data_as(_P)))
        return st

    def get_history(self, game, max_records=None):
        """Statistics history of one game.  max_records=None: everything the game recorded (alg_get_stats first); a history
        the device buffer could not hold completely raises a warning (alg_get_history returns fewer records than were made)."""
        made = None
        if max_records is None:
            made = int(self.get_stats()["records"][game])
            max_records = max(made, 1)
        out = np.zeros(max_records, dtype=record_dtype); cnt = C.c_int32()
        self.lib.check(self.lib.get_history(self.h, game, max_records, out.ctypes.data_as(_P), C.byref(cnt)))
        if made is not None and cnt.value < made:
            import warnings
            warnings.warn(f"alg_get_history: game {game} made {made} records, the device history holds {cnt.value} (truncated)")
        return out[:cnt.value]

    def synchronize(self):
        self.lib.check(self.lib.synchronize(self.h))

    def ibr_solve_player(self, player):
        st = np.zeros(self.B, dtype=game_stats_dtype)
        self.lib.check(self.lib.ibr_solve_player(self.h, int(player), st.ctypes.data_as(_P)))
        return st

    def ibr_newton_solve(self, ibr_iter=100, ordering=None, delta_min=1e-9, init=True, game_id0=0):
        order = np.ascontiguousarray(np.arange(self.p) if ordering is None else np.asarray(ordering)[:self.p], dtype=np.int32)
        st = np.zeros(self.B, dtype=game_stats_dtype)
        self.lib.check(self.lib.ibr_newton_solve(self.h, int(init), game_id0, int(ibr_iter), _iptr(order), float(delta_min), st.ctypes.data_as(_P)))
        return st

    def mpc_advance(self):
        self.lib.check(self.lib.mpc_advance(self.h))

    def mpc_solve(self, steps, game_id0=0, record_states=False):
        """The whole receding-horizon loop in one call; returns the states (steps+1, B, n) or None (asynchronous)."""
        states = np.empty((steps + 1, self.B, self.n)) if record_states else None
        self.lib.check(self.lib.mpc_solve(self.h, int(steps), int(game_id0), _dptr(states)))
        return states

    def mpc_totals(self, reset=False):
        it = np.zeros(self.B, dtype=np.int64); cv = np.zeros(self.B, dtype=np.int64)
        self.lib.check(self.lib.mpc_totals(self.h, it.ctypes.data_as(C.POINTER(C.c_int64)), cv.ctypes.data_as(C.POINTER(C.c_int64)), int(reset)))
        return it, cv

    def get_x0(self):
        return self.get_traj(ALG_TRAJ_PD)[:, :self.n].copy()

    # ---- views of a traj buffer (primal_dual_traj.jl layout) ----------------------------------
    def split_traj(self, z):
        """z: (B, n+S) -> states (B,N,n), controls (B,N-1,m) joint order, duals (B,p,N-1,n)."""
        B, n, m, p, N, mi = z.shape[0], self.n, self.m, self.p, self.N, self.mi
        blk = z[:, n:].reshape(B, N - 1, self.b)
        X = np.concatenate([z[:, None, :n], blk[:, :, :n]], axis=1)
        ug = blk[:, :, n:n + m].reshape(B, N - 1, p, mi)         # [player, j] -> joint index i + j*p
        U = ug.transpose(0, 1, 3, 2).reshape(B, N - 1, m)
        L = blk[:, :, n + m:].reshape(B, N - 1, p, n).transpose(0, 2, 1, 3)
        return X, U, L

    def join_traj(self, X, U, L):
        B, n, m, p, N, mi = X.shape[0], self.n, self.m, self.p, self.N, self.mi
        z = np.zeros((B, self.traj_len))
        z[:, :n] = X[:, 0]
        blk = z[:, n:].reshape(B, N - 1, self.b)
        blk[:, :, :n] = X[:, 1:]
        blk[:, :, n:n + m] = U.reshape(B, N - 1, mi, p).transpose(0, 1, 3, 2).reshape(B, N - 1, m)
        blk[:, :, n + m:] = L.transpose(0, 2, 1, 3).reshape(B, N - 1, p * n)
        return z
