"""python tests/probes/dump_solve.py CFG NGAMES OUT.npy [waves] -- solves the first NGAMES scenarios of a BASELINE configuration with the library
ALGAMES_HIP_LIB selects and saves [trajectory | newton_iters]: bitwise comparisons of build variants (tests/probes/build_variant.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import algames_jl_amd as alg
cfg, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
p = alg.scenarios.make_problem(cfg, np.arange(n))
if len(sys.argv) > 4: p.batch.set_waves_per_game(int(sys.argv[4]))
alg.newton_solve(p)
np.save(out, np.concatenate([p.batch.get_traj(0), p.stats.summary["newton_iters"][:, None].astype(float)], axis=1))
print(cfg, n, "waves", p.batch.get_waves_per_game(), "iters", int(p.stats.summary["newton_iters"].sum()))
