"""Cycle profile of the quad-team direction per role (library built with tests/probes/qt_prof.sh build). usage: qt_prof.py [games]"""
import sys, os, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import torch
import algames_jl_amd as alg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prob = alg.scenarios.make_problem("C2", np.arange(G)); prob.batch.set_waves_per_game(1); prob.batch.set_quad_team(1)
import time
alg.newton_solve(prob); torch.cuda.synchronize()
prob = alg.scenarios.make_problem("C2", np.arange(G)); prob.batch.set_waves_per_game(1); prob.batch.set_quad_team(1)
torch.cuda.synchronize(); t0 = time.perf_counter(); alg.newton_solve(prob); torch.cuda.synchronize(); print("solve ms", (time.perf_counter() - t0) * 1e3)
b = prob.batch
fn = b.lib.dll.alg_debug_read_res; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
out = np.zeros((G, 48)); assert fn(b.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 48) == 0
it = prob.stats.summary["newton_iters"].astype(float)
print("raw game 0:", out[0, 16:48].tolist()); print("raw game 5:", out[5, 16:48].tolist())
steps = b.N - 1
pn = ["product", "A'/Q^/y/U/S rows", "wait B2+B1 (control solve)", "G + forward (wave 0) / wait", "costate compute", "costate barrier", "setup", "-"]
cn = ["stage next record/table", "wait B2 (players)", "control solve + gains out", "wait B1", "G + stage costate + wait F", "costate stage", "costate barrier", "setup"]
for wv in range(4):
    a = out[:, 16 + 8 * wv: 24 + 8 * wv] / it[:, None]
    m = a.mean(0)
    print("wavefront %d (%s): %.0f cycles per direction" % (wv, "control" if wv == 3 else "player", m.sum()))
    for j in range(8):
        if m[j] > 0: print("   %-34s %9.0f cycles/iter  %7.0f per step" % ((cn if wv == 3 else pn)[j], m[j], m[j] / steps))
