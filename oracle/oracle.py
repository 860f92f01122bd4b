"""Python binding of the CPU oracle (oracle/lib/liboracle.so).  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import algames_jl_amd  # noqa: E402
from algames_jl_amd._abi import CLib, Batch, alg_desc, _dptr, _f64, _P  # noqa: E402

LIB_PATH = os.path.join(_here, "lib", "liboracle.so")
# arbiter builds: the same source with the scalar type swapped (oracle/Makefile): long double / __float128
LIB_X_PATH = os.path.join(_here, "lib", "liboracle_x.so")
LIB_Q_PATH = os.path.join(_here, "lib", "liboracle_q.so")


def build(force=False):
    """Builds the oracle and the long-double arbiter (`make`).  The __float128 arbiter (libquadmath) is built on demand by lib("q")."""
    src = os.path.join(_here, "algames_oracle.cpp")
    hdr = os.path.join(os.path.dirname(_here), "include", "algames_hip.h")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr))
    stale = any((not os.path.exists(q)) or os.path.getmtime(q) < newest for q in (LIB_PATH, LIB_X_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _here, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def _build_q():
    src = os.path.join(_here, "algames_oracle.cpp")
    if (not os.path.exists(LIB_Q_PATH)) or os.path.getmtime(LIB_Q_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _here, "-s", "lib/liboracle_q.so"])


_lib = None
_arb = {}


def lib(kind=""):
    """kind "": the oracle (double).  "x" / "q": the arbiter builds (long double / __float128 arithmetic behind the same ABI)."""
    global _lib
    if kind:
        if kind not in _arb:
            path = {"x": LIB_X_PATH, "q": LIB_Q_PATH}[kind]
            if kind == "q":
                _build_q()
            elif not os.path.exists(path):
                build()
            _arb[kind] = CLib(path, "orc_")
            _arb[kind].dll.orc_set_threads.restype = C.c_int
            _arb[kind].dll.orc_set_threads.argtypes = [C.c_int]
        return _arb[kind]
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = CLib(LIB_PATH, "orc_")
        d = _lib.dll
        d.orc_kat_dynamics.restype = C.c_int
        d.orc_kat_dynamics.argtypes = [C.POINTER(alg_desc)] + [C.POINTER(C.c_double)] * 6
        d.orc_kat_dynamics_h.restype = C.c_int
        d.orc_kat_dynamics_h.argtypes = [_P] + [C.POINTER(C.c_double)] * 6
        d.orc_kat_cost.restype = C.c_int
        d.orc_kat_cost.argtypes = [_P, C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_double)] * 5
        d.orc_kat_collision_cost.restype = C.c_double
        d.orc_kat_collision_cost.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        d.orc_kat_evaluate_con.restype = C.c_int
        d.orc_kat_evaluate_con.argtypes = [_P, C.POINTER(C.c_double)]
        d.orc_kat_delta_step.restype = C.c_double
        d.orc_kat_delta_step.argtypes = [_P, C.c_int32, C.c_double]
        d.orc_counter_uniform.restype = C.c_double
        d.orc_counter_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        d.orc_set_threads.restype = C.c_int
        d.orc_set_threads.argtypes = [C.c_int]
        d.orc_kat_dense_direction.restype = C.c_int
        d.orc_kat_dense_direction.argtypes = [_P, C.c_int32, C.c_double, C.POINTER(C.c_double)]
    return _lib


class OracleBatch(Batch):
    def __init__(self, model, p, N, dt, batch, d=2, kind=""):
        super().__init__(lib(kind), model, p, N, dt, batch, d=d, device=0)

    # fine-grained known-answer hooks ------------------------------------------------------------
    def kat_dynamics(self, x, u):
        x, u = _f64(x, (self.n,)), _f64(u, (self.m,))
        xd, x2, x3 = np.empty(self.n), np.empty(self.n), np.empty(self.n)
        J = np.empty((self.n, self.n + self.m))
        self.lib.check(self.lib.dll.orc_kat_dynamics_h(self.h, _dptr(x), _dptr(u), _dptr(xd), _dptr(x2), _dptr(x3), _dptr(J)))
        return xd, x2, x3, J

    def kat_cost(self, i, k, x, u, game=0):
        x, u = _f64(x, (self.n,)), _f64(u, (self.m,))
        q, r, Q = np.empty(self.n), np.empty(self.mi), np.empty((self.n, self.n))
        self.lib.check(self.lib.dll.orc_kat_cost(self.h, game, i, k, _dptr(x), _dptr(u), _dptr(q), _dptr(r), _dptr(Q)))
        return q, r, Q

    def kat_evaluate_con(self):
        v = np.empty((self.B, self.con_len))
        self.lib.check(self.lib.dll.orc_kat_evaluate_con(self.h, _dptr(v)))
        return v

    def kat_delta_step(self, alpha, game=0):
        return self.lib.dll.orc_kat_delta_step(self.h, game, alpha)

    def kat_dense_direction(self, reg, game=0):
        d = np.empty(self.S)
        self.lib.check(self.lib.dll.orc_kat_dense_direction(self.h, game, reg, _dptr(d)))
        return d


def collision_cost_value(mu, r, xi, xj):
    return lib().dll.orc_kat_collision_cost(mu, r, _dptr(_f64(xi)), _dptr(_f64(xj)))


def counter_uniform(seed, game, counter):
    return lib().dll.orc_counter_uniform(seed, game, counter)


def set_threads(n):
    """OpenMP threads for the oracle's loop over games; returns the previous setting."""
    return lib().dll.orc_set_threads(int(n))
