"""Per-phase cycle profile of the Newton-direction sweeps (library built with -DALG_PHASE_PROF: tests/probes/phase_prof.sh).
usage: python tests/probes/phase_prof.py CONFIG GAMES"""
import sys, os, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
cfg, G = sys.argv[1], int(sys.argv[2])
prob = alg.scenarios.make_problem(cfg, np.arange(G)); prob.batch.set_waves_per_game(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
alg.newton_solve(prob)
b = prob.batch
fn = b.lib.dll.alg_debug_read_res; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
out = np.zeros((G, 12)); assert fn(b.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 12) == 0
it = prob.stats.summary["newton_iters"].astype(float)
names = ["value recursion (MFMA + A' + write-back)", "Q-add (+ s_i)", "V rows, y_i, A' table", "g_c", "column build", "pivoted solve", "gains out + closed loop + record copy",
         "forward sweep", "costate sweep", "  (closed loop: K negate, B K + A -> Fx)", "  (record copy: wait for the prefetch)", "set-up before the backward loop"]
steps = (b.N - 1)
tot = out[:, :11].sum(1) + out[:, 11]
print(f"{cfg} {G} games: cycles per Newton iteration in newton_direction (mean over games) = {np.mean(tot / it):.0f}")
for j in list(range(12)):
    per_it = np.mean(out[:, j] / it)
    print(f"  {names[j]:45s} {per_it:9.0f} cycles/iter  {100 * per_it / np.mean(tot / it):5.1f} %   {per_it / steps:7.0f} per step")
# pass-level slots (res[16..31], flushed at the end of newton_solve)
out2 = np.zeros((G, 32)); assert fn(b.h, out2.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 32) == 0
pn = {19: "init_traj + rollout (per solve / iters)", 31: "whole solve (per solve / iters)", 16: "axpy + barrier", 17: "trial assemble (fused pass)", 20: "  phase A", 21: "  rows x", 22: "  rows u", 23: "  rows d", 24: "  reductions", 26: "direction", 27: "record pass"}
cn = {18: "#trials", 25: "#assemble passes", 28: "#directions", 29: "#record passes"}
print("pass level (thread 0's clock), cycles per Newton iteration:")
for s, nm in pn.items(): print(f"  {nm:32s} {np.mean(out2[:, s] / it):10.0f}")
for s, nm in cn.items(): print(f"  {nm:32s} {np.mean(out2[:, s] / it):10.2f} per iteration")
