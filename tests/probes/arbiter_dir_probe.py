"""Where does the quadrotor family lose its digits?  At the initial point and after one accepted step: Jacobian and Newton direction of
the HIP path and of the double oracle against the long-double arbiter, plus the backward error of each direction in the arbiter's
Jacobian."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
def run(seed, kw):
    g, o, x, tag = F._random_pair(alg, orc, np.random.default_rng(seed), arb="x", **kw)
    reg = 1e-6
    for b in (g, o, x): b.init_traj(game_id0=7); b.rollout()
    for it in range(2):
        Jx = x.residual_jacobian(reg); rx = x.residual(reg=reg)[0]
        Jg, Jo = g.residual_jacobian(reg), o.residual_jacobian(reg)
        dg, do_, dx = g.newton_direction(reg)[0], o.newton_direction(reg)[0], x.newton_direction(reg)[0]
        for game in range(g.B):
            sJ = np.abs(Jx[game]).max(); sd = np.abs(dx[game]).max()
            be = lambda d: np.abs(Jx[game] @ d + rx[game]).max() / (np.abs(Jx[game]).sum(1).max() * np.abs(d).max() + np.abs(rx[game]).max())
            c = np.linalg.cond(Jx[game])
            print(seed, tag[:4], "it", it, "game", game, "cond %.1e" % c, "J: hip %.0e orc %.0e" % (np.abs(Jg[game] - Jx[game]).max() / sJ, np.abs(Jo[game] - Jx[game]).max() / sJ),
                  "dir: hip %.0e orc %.0e" % (np.abs(dg[game] - dx[game]).max() / sd, np.abs(do_[game] - dx[game]).max() / sd),
                  "bwd: hip %.0e orc %.0e x %.0e" % (be(dg[game]), be(do_[game]), be(dx[game])))
        for b in (g, o, x): b.update_traj(0.5)      # same step with each backend's OWN direction held in its buffers? no: set the arbiter's
        z = x.get_traj()
        for b in (g, o): b.set_traj(z)
QUAD = [s for s in range(13000, 13040) if F.DENSE_FAMILIES[s % 7][0] == 3][:6]
for s in QUAD: run(s, dict(ext=bool(s % 2), force=F.DENSE_FAMILIES[s % 7]))
for s in (13000, 13001): run(s, dict(ext=bool(s % 2), force=F.DENSE_FAMILIES[s % 7]))
for s in (1003, 1004): run(s, dict(ext=False))
