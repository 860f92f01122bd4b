"""Pass-level time accounting of the fused solver (library built with -DALG_PHASE_PROF, tests/probes/phase_prof.sh build):
usage: ALGAMES_HIP_LIB=tests/probes/lib_prof.so python tests/probes/ls_prof.py CONFIG GAMES WAVES [MPC_STEPS]"""
import sys, os, ctypes, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
cfg, G, nw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); T = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prob = alg.scenarios.make_problem(cfg, np.arange(G)); prob.batch.set_waves_per_game(nw)
t0 = time.time()
if T: it, cv, _ = alg.mpc_solve(prob, T); it = it.astype(float)
else: alg.newton_solve(prob); it = prob.stats.summary["newton_iters"].astype(float)
wall = time.time() - t0
b = prob.batch
fn = b.lib.dll.alg_debug_read_res; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
out = np.zeros((G, 32)); assert fn(b.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 32) == 0
g = int(np.argmax(it))                               # the slowest game sets the launch time
print(f"{cfg} {G} games, {nw} wavefronts per game, mpc steps {T}: wall {wall*1e3:.1f} ms (incl. first-call overhead); directions mean {it.mean():.0f} max {it.max():.0f}")
for name, sel in (("mean over games", slice(None)), ("slowest game", slice(g, g + 1))):
    o = out[sel].mean(0); d = max(o[28], 1.0)
    print(f"  [{name}] per direction: trials {o[18]/d:.2f}, record passes {o[29]/d:.2f}, assemble passes {o[25]/d:.2f}")
    print(f"     cycles per direction: Newton direction {o[26]/d:.0f} | record passes {o[27]/d:.0f} | trial axpy+barrier {o[16]/d:.0f} | trial assemble {o[17]/d:.0f}  -> sum {(o[26]+o[27]+o[16]+o[17])/d:.0f}")
    p = max(o[25], 1.0)
    print(f"     cycles per assemble pass: phase A {o[20]/p:.0f} | rows x {o[21]/p:.0f} | rows u {o[22]/p:.0f} | rows d {o[23]/p:.0f} | reductions + team combine {o[24]/p:.0f}  -> {o[20:25].sum()/p:.0f};  axpy+barrier per trial {o[16]/max(o[18],1):.0f}")
