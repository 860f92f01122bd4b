"""Line-search trials per Newton iteration of the bench shapes (history of a sample of games): python tests/probes/r05_trials.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
for cfg, G, fam, kw in (("C2", 256, "C2", {}), ("C3", 256, "C3", {}), ("C5", 256, "C5", {}), ("Q2", 256, "Q", {"p": 2}), ("Q4", 128, "Q", {"p": 4})):
    prob = alg.scenarios.make_problem(fam, np.arange(G), **kw); alg.newton_solve(prob)
    st = prob.batch.get_stats(); js = []
    for g in range(min(G, 64)):
        h = prob.batch.get_history(g); js.append(h["ls_j"][h["ls_j"] > 0])
    js = np.concatenate(js)
    print(f"{cfg}: iterations per game mean {st['newton_iters'].mean():.1f} max {st['newton_iters'].max()}; trials per iteration mean {js.mean():.2f} max {js.max()}; share of iterations with more than one trial {np.mean(js > 1):.2f}; failed searches {st['ls_failures'].sum()}")
