"""Pass-level cycle accounts of the fused receding-horizon loop (library built with -DALG_PHASE_PROF): python tests/probes/r05_mpc_prof.py [games] [steps] [waves]"""
import sys, os, ctypes, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100; w = int(sys.argv[3]) if len(sys.argv) > 3 else 4
prob = alg.scenarios.make_problem("C5", np.arange(G)); prob.batch.set_waves_per_game(w)
b = prob.batch
alg.mpc_solve(prob, 5)                       # warm-up launch
fn = b.lib.dll.alg_debug_read_res; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
o0 = np.zeros((G, 32)); assert fn(b.h, o0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 32) == 0
t0 = time.perf_counter(); it, cv, _ = alg.mpc_solve(prob, steps); b.synchronize() if hasattr(b, "synchronize") else None; t1 = time.perf_counter()
o1 = np.zeros((G, 32)); assert fn(b.h, o1.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 32) == 0
d = (o1 - o0).mean(0)
its = it.mean()
print(f"C5 loop {G} games x {steps} steps, {w} wavefronts per game: {1e3 * (t1 - t0):.1f} ms wall, {its / steps:.2f} Newton iterations per step, converged {cv.mean() / steps:.3f}")
print(f"  per Newton iteration: {1e6 * (t1 - t0) / its:.1f} us wall")
pn = {16: "axpy + barrier", 17: "trial passes (one step size)", 20: "  phase A", 21: "  rows x", 22: "  rows u", 23: "  rows d", 24: "  reductions", 26: "direction (incl. corrections)", 27: "record pass", 19: "init_traj + rollout", 30: "group passes of the line search"}
cn = {18: "#trials", 25: "#assemble passes", 28: "#directions", 29: "#record passes"}
tot = 0
for s, nm in pn.items():
    print(f"  {nm:32s} {d[s] / its:10.0f} cycles per Newton iteration")
    if s in (16, 17, 26, 27, 19, 30): tot += d[s] / its
for s, nm in cn.items(): print(f"  {nm:32s} {d[s] / its:10.2f} per Newton iteration")
print(f"  accounted: {tot:.0f} cycles per Newton iteration = {tot / 2.3e3:.1f} us at 2.3 GHz")

gs = int(np.argmax((o1 - o0)[:, 31])); ds = (o1 - o0)[gs]; i_s = float(it[gs])
print(f"longest-running game {gs}: {i_s:.0f} Newton iterations ({i_s / steps:.2f} per step), {ds[18] / i_s:.2f} trials per iteration")
tot = 0
for s_ in (16, 17, 26, 27, 19, 30):
    print(f"  {pn[s_]:32s} {ds[s_] / i_s:10.0f} cycles per Newton iteration   total {ds[s_] / 2.3e6:8.1f} ms at 2.3 GHz"); tot += ds[s_]
print(f"  accounted {tot / 2.3e6:.1f} ms of {1e3 * (t1 - t0):.1f} ms wall")
tt = (o1 - o0)[:, 31] / 2.39e6
print("whole-solve time per game (slot 31, ms at 2.39 GHz): min %.1f median %.1f max %.1f (game %d, %d iterations); wall %.1f ms" % (tt.min(), np.median(tt), tt.max(), int(np.argmax(tt)), int(it[int(np.argmax(tt))]), 1e3 * (t1 - t0)))
print("  iterations per game: min %d median %d max %d; trials per iteration of the longest-running game %.2f" % (it.min(), np.median(it), it.max(), (o1 - o0)[int(np.argmax(tt)), 18] / it[int(np.argmax(tt))]))
