"""Throughput of the intro_example.jl scenario (3-player bicycle, all constraint types) as a batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import algames_jl_amd as alg
from test_gpu_parity_ext import _intro_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0 = np.array([0.1, 0.0, 0.5, -0.4, 0.0, 0.7, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
X0 = np.tile(x0, (G, 1)); X0[1:, :6] += 0.05 * (np.random.default_rng(4).random((G - 1, 6)) - 0.5)
prob = _intro_problem(alg, None, X0)
for rep in range(3):
    t = time.time(); alg.newton_solve(prob); dt = time.time() - t
    s = prob.stats.summary
    print(f"intro x{G}: {dt*1e3:.1f} ms, iters {s['newton_iters'].sum()} ({s['newton_iters'].sum()/dt/1e6:.3f} M it/s), converged {s['converged'].sum()}, status!=0 {np.count_nonzero(s['status'])}, iters min/max {s['newton_iters'].min()}/{s['newton_iters'].max()}")
