"""Per-phase cycle profile of newton_direction_dense (library built with -DALG_PHASE_PROF: tests/probes/phase_prof.sh build).
usage: python tests/probes/phase_prof_dense.py P GAMES [NW]"""
import sys, os, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
P, G = int(sys.argv[1]), int(sys.argv[2]); nw = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prob = alg.scenarios.make_problem("Q", np.arange(G), p=P); prob.batch.set_waves_per_game(nw)
alg.newton_solve(prob)
b = prob.batch
fn = b.lib.dll.alg_debug_read_res; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
out = np.zeros((G, 12)); assert fn(b.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 12) == 0
it = prob.stats.summary["newton_iters"].astype(float)
names = ["value recursion: MFMA product", "value recursion: A' apply", "Q-add", "V rows, y_i", "control system build", "pivoted solve (LDS Gauss-Jordan)", "gains out + closed loop",
         "forward sweep", "costate sweep", "-", "-", "-"]
steps = b.N - 1
ok = (out < 1e13).all(1)                       # a few games see a wrapped counter difference: dropped
print(f"games with sane counters: {ok.sum()} of {G}")
out, it = out[ok], it[ok]
tot = out.sum(1)
print(f"Q p={P} {G} games nw={nw}: cycles per Newton iteration in newton_direction_dense (mean over games) = {np.mean(tot / it):.0f}")
for j in range(9):
    per_it = np.mean(out[:, j] / it)
    print(f"  {names[j]:40s} {per_it:9.0f} cycles/iter  {100 * per_it / np.mean(tot / it):5.1f} %   {per_it / steps:7.0f} per step")
