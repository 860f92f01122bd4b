"""Distribution of per-game solve durations inside one launch (sum of the device-measured t_elap of a game's inner iterations):
do the wavefronts that share a SIMD progress equally?  usage: python tests/probes/finish_times.py [CONFIG GAMES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import algames_jl_amd as alg
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prob = alg.scenarios.make_problem(cfg, np.arange(B)); prob.batch.set_waves_per_game(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
for rep in range(2): alg.newton_solve(prob)
dur = np.array([prob.batch.get_history(g)["t_elap"].sum() for g in range(B)]) * 1e3
q = np.percentile(dur, [0, 5, 25, 50, 75, 95, 100])
print(f"{cfg} {B} games: per-game solve duration (ms): min {q[0]:.2f} p5 {q[1]:.2f} p25 {q[2]:.2f} median {q[3]:.2f} p75 {q[4]:.2f} p95 {q[5]:.2f} max {q[6]:.2f}; mean/max = {dur.mean()/dur.max():.3f}")
for lo in range(0, B, B // 8): print(f"   games {lo:5d}..{lo + B // 8 - 1:5d}: mean {dur[lo:lo + B // 8].mean():.2f} ms")
