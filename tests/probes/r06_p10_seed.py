"""Round-6 probe (GPU box): the one case of the ten-player family of tests/probes/fuzz_long_r6.py that ended outside the rule (seed 900022, Unicycle, extended
set; Newton-iteration counts 12 against 14 in one game).  Prints, per game, the first record where the HIP path, the double oracle and the long-double
arbiter take different line-search decisions, and the distance of both double programs from the arbiter in the records before it.
usage: python tests/probes/r06_p10_seed.py [seed]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 900022
s = seed - 900000
model = (F.DI, F.UNI, F.BIC)[s % 3]
g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), ext=(model == F.BIC or bool((s // 3) % 2)), force=(model, 10), force_d3=False, arb="x")
print("tag", tag[:5])
st = [b.newton_solve(init=True, game_id0=7) for b in (g, o, x)]
for f in ("status", "outer_iters", "newton_iters", "ls_failures"):
    print("%-13s hip %s oracle %s arbiter %s" % (f, st[0][f], st[1][f], st[2][f]))
for game in range(g.B):
    hg, ho, hx = g.get_history(game), o.get_history(game), x.get_history(game)
    split = None
    for rec in range(min(len(hg), len(ho), len(hx))):
        dg, do, dx = int(hg["ls_j"][rec]), int(ho["ls_j"][rec]), int(hx["ls_j"][rec])
        if not (dg == do == dx):
            split = (rec, dg, do, dx); break
    upto = split[0] if split else min(len(hg), len(ho), len(hx))
    worst = [0.0, 0.0]
    for rec in range(upto):
        for f in F.ARB_FIELDS:
            sc = abs(hx[f][rec]) + 1e-300
            worst[0] = max(worst[0], abs(hg[f][rec] - hx[f][rec]) / sc); worst[1] = max(worst[1], abs(ho[f][rec] - hx[f][rec]) / sc)
    print("game %d: records hip %d oracle %d arbiter %d; first split (record, j_hip, j_oracle, j_arbiter) %s; before it max rel distance from the arbiter: hip %.2e oracle %.2e"
          % (game, len(hg), len(ho), len(hx), split, worst[0], worst[1]))
    if split:
        rec = split[0]
        for f in ("res",):
            print("    record %d %s: hip %.17g oracle %.17g arbiter %.17g; previous record res: hip %.17g oracle %.17g arbiter %.17g"
                  % (rec, f, hg[f][rec], ho[f][rec], hx[f][rec], hg[f][rec - 1], ho[f][rec - 1], hx[f][rec - 1]))
zx = x.get_traj(0)
print("arbiter iterate max |z| per game:", np.abs(zx).max(axis=1))
