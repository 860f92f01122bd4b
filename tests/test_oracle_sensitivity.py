"""CPU-only: how much the ORACLE itself amplifies a perturbation of x0 by at most one ulp per entry on the fuzz problems that sit outside the parity rule
of tests/test_gpu_fuzz.py (the long generator at 4x size, profiles/r06_fuzz_1600_other_families.txt, r06_fuzz_base_2000.txt).  A case where two correct
double programs part by more than the tolerance is acceptable only when the problem, not an arithmetic defect, does it: the oracle is run against itself
with x0 (1 + eps s), eps = 1e-15 and 1e-13, s = +-1 per entry, and must move the final iterate of some game by more than 1e-9 (relative) or change
its own discrete decisions -- while ordinary seeds of the same families stay put.  The GPU companion (test_long_run_outliers_are_problems_that_amplify_one_ulp)
bounds |hip - oracle| per game by this envelope."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_fuzz as F                                             # (module-level gpu mark applies to ITS tests only)


@pytest.mark.parametrize("seed", F.LONG_RUN_OUTLIERS)
def test_outlier_seed_amplifies_one_ulp_of_x0(alg, orc, seed):
    env, flipped = F.oracle_sensitivity(alg, orc, seed)
    ref = F._long_run_pair(F._OracleAsHip(alg, orc), orc, seed)[1]
    ref.newton_solve(init=True, game_id0=7)
    scale = np.maximum(1.0, np.abs(ref.get_traj(0)).max(axis=1))
    assert (flipped | (env > 1e-9 * scale)).any(), (seed, env, scale, flipped)


def test_seed_101156_envelope(alg, orc):
    # the GPU test measures |hip - arbiter| = 0.139 in game 2 (scale 232), 5e-7 in game 1, 1e-11 in game 0
    env, flipped = F.oracle_sensitivity(alg, orc, 101156, eps_list=(1e-15,))
    assert env[2] > 1e-2 and env[0] < 1e-8 and env[1] < 1e-5 and not flipped.any(), (env, flipped)


@pytest.mark.parametrize("seed", [100001, 100002, 200001, 200003, 600001, 700002])
def test_ordinary_seeds_do_not_amplify(alg, orc, seed):
    env, flipped = F.oracle_sensitivity(alg, orc, seed, eps_list=(1e-13,))
    ref = F._long_run_pair(F._OracleAsHip(alg, orc), orc, seed)[1]
    ref.newton_solve(init=True, game_id0=7)
    assert not flipped.any() and (env <= 1e-9 * np.maximum(1.0, np.abs(ref.get_traj(0)).max(axis=1))).all(), (seed, env)
