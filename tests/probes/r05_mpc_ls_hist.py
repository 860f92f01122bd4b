"""Line-search trial counts of the receding-horizon loop (step-by-step form, history of every solve): python tests/probes/r05_mpc_ls_hist.py [games] [steps]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, root)
import numpy as np
import algames_jl_amd as alg
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prob = alg.scenarios.make_problem("C5", np.arange(G)); b = prob.batch
hist = np.zeros(16, dtype=np.int64); per_game_it = np.zeros(G); per_game_tr = np.zeros(G); outer = np.zeros(G); fails = np.zeros(G)
for t in range(steps):
    if t == 1:
        prob.opts.shift, prob.opts.dual_reset = 1, False; prob._sync_options()
    b.newton_solve_async(init=True, game_id0=prob.game_id0 + t * 1000003)
    st = b.get_stats()
    for g in range(G):
        h = b.get_history(g)
        js = h["ls_j"][h["ls_j"] > 0]
        for j in js: hist[min(int(j), 15)] += 1
        per_game_tr[g] += np.minimum(js, 9).sum()
    per_game_it += st["newton_iters"]; outer += st["outer_iters"]; fails += st["ls_failures"]
    b.mpc_advance()
print("ls_j histogram over all Newton iterations (j = ls_iter: failed):", {j: int(c) for j, c in enumerate(hist) if c})
print("Newton iterations per game per step: mean %.2f, max %.2f; outer iterations per solve %.2f; failed line searches per Newton iteration %.3f" % (per_game_it.mean() / steps, per_game_it.max() / steps, outer.mean() / steps, fails.sum() / per_game_it.sum()))
o = np.argsort(-per_game_it)
print("slowest games: iterations", per_game_it[o[:4]].astype(int), "failed searches", fails[o[:4]].astype(int), "trials", per_game_tr[o[:4]].astype(int))
print("fastest games: iterations", per_game_it[o[-4:]].astype(int), "failed searches", fails[o[-4:]].astype(int), "trials", per_game_tr[o[-4:]].astype(int))
