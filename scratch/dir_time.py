import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import algames_jl_amd as alg
prob = alg.scenarios.make_problem("C2", np.arange(4096)); b = prob.batch; prob._sync_options()
b.init_traj(0)
for l in range(1, 3):
    b.newton_step(1, l)
import ctypes
lib = alg.hip_lib()
st = np.empty(b.B, dtype=np.int32)
# time k_direction via the ABI (includes a d2h of status; subtract by timing residual-only record call)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
td = t(lambda: lib.newton_direction(b.h, 1e-3, None, None))
tr = t(lambda: b.record())
print(os.environ.get("ALGAMES_HIP_LIB","")[-12:], "direction(incl assemble) ms %.3f   record ms %.3f   sweeps ~ %.3f" % (td, tr, td - tr))
