"""CPU-only: how much the ORACLE itself amplifies a sub-ulp perturbation of x0 on the fuzz problems that sit outside the parity rule of
tests/test_gpu_fuzz.py.  A case where two correct double programs part by more than the tolerance is only acceptable when the problem, not an
arithmetic defect, does it: here the oracle is run against itself with x0 (1 + 1e-15 s), s = +-1 per entry, and must move the final iterate of the
differing game by MORE than the HIP path differs from the long-double arbiter (measured on the GPU, recorded in the GPU test), while the games that
agree on the GPU stay put and ordinary seeds do not move at all."""
import os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class _OracleAsBoth:
    """`alg` whose HIP library is the oracle's: _random_pair then builds the same problem twice on the CPU."""
    def __init__(self, alg, orc): self._alg, self._orc = alg, orc
    def __getattr__(self, k): return getattr(self._alg, k)
    def hip_lib(self): return self._orc.lib()


def _perturbed_pair(alg, orc, seed, eps):
    import test_gpu_fuzz as F
    g, o, tag = F._random_pair(_OracleAsBoth(alg, orc), orc, np.random.default_rng(seed), False)
    x0 = o.get_x0()
    g.set_x0(x0 * (1 + eps * np.sign(np.sin(np.arange(x0.size).reshape(x0.shape)))))
    sg, so = g.newton_solve(init=True, game_id0=7), o.newton_solve(init=True, game_id0=7)
    return g.get_traj(0), o.get_traj(0), sg, so, tag


def test_seed_101156_game_2_amplifies_a_sub_ulp_perturbation_of_x0(alg, orc):
    zp, zo, sp, so, tag = _perturbed_pair(alg, orc, 101156, 1e-15)
    assert tag[:4] == (1, 4, 14, 0.2) and np.array_equal(sp["newton_iters"], so["newton_iters"])
    dz, scale = np.abs(zp - zo).max(axis=1), np.abs(zo).max(axis=1)
    # the GPU test measures |hip - arbiter| = 0.139 in game 2 (scale 232), 5e-7 in game 1, 1e-11 in game 0
    assert dz[2] > 1e-2 and dz[2] / scale[2] > 1e-5, (dz, scale)
    assert dz[0] < 1e-8 and dz[1] < 1e-5, (dz, scale)


def test_ordinary_base_seeds_do_not_amplify(alg, orc):
    for seed in (100001, 100002):
        zp, zo, sp, so, tag = _perturbed_pair(alg, orc, seed, 1e-13)
        assert np.abs(zp - zo).max() <= 1e-10 * np.abs(zo).max(), (seed, tag[:4], np.abs(zp - zo).max())
