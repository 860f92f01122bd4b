"""Per-phase kernel timing of the stepwise API at a BASELINE config (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import algames_jl_amd as alg
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prob = alg.scenarios.make_problem(cfg, np.arange(G))
b = prob.batch
prob._sync_options()
b.init_traj(0)
for l in range(1, 4):
    reg = 1e-3 * l ** 4
    rec = b.record()
    d, st = b.newton_direction(reg)
    a, j = b.line_search(rec["res"], reg)
    b.update_traj(a)
t = time.time(); st = b.newton_solve(); print("solve s", time.time() - t, st["newton_iters"].sum())
