"""Per seed: deviation of the HIP path and of the double oracle from the long-double arbiter, record by record, normalised by
max(|value|, 1e-3 * first residual) so that fields sitting at zero do not pollute the picture."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
def run(seed, kw):
    g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), arb="x", **kw)
    for b in (g, o, x): b.newton_solve(init=True, game_id0=7)
    print(seed, tag[:4], "res_x=", ["%.2g" % v for v in x.get_history(0)["res"][:8]])
    for game in range(g.B):
        hg, ho, hx = g.get_history(game), o.get_history(game), x.get_history(game)
        s0 = 1e-3 * abs(hx["res"][0]); out = []; split = None
        for rec in range(min(len(hg), len(ho), len(hx), 14)):
            eg = max(abs(hg[f][rec] - hx[f][rec]) / max(abs(hx[f][rec]), s0) for f in F.ARB_FIELDS)
            eo = max(abs(ho[f][rec] - hx[f][rec]) / max(abs(hx[f][rec]), s0) for f in F.ARB_FIELDS)
            out.append("%.0e/%.0e" % (eg, eo))
            if not (hg["ls_j"][rec] == ho["ls_j"][rec] == hx["ls_j"][rec]): split = (rec, int(hg["ls_j"][rec]), int(ho["ls_j"][rec]), int(hx["ls_j"][rec])); break
        print("   game", game, "records", len(hx), "split", split, " ".join(out))
QUAD = [s for s in range(13000, 13040) if F.DENSE_FAMILIES[s % 7][0] == 3][:10]
for s in QUAD: run(s, dict(ext=bool(s % 2), force=F.DENSE_FAMILIES[s % 7]))
for s in (13000, 13001, 13002): run(s, dict(ext=bool(s % 2), force=F.DENSE_FAMILIES[s % 7]))
for s in (1000, 1003, 1004): run(s, dict(ext=False))
