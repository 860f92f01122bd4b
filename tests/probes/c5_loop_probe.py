"""Where does the C5 closed loop of the HIP path leave the oracle's?  Step-wise loops of both, per-step iteration counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
os.system("nproc; lscpu | head -20")
ids = np.arange(128, 192); T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pg = alg.scenarios.make_problem("C5", ids); po = alg.scenarios.make_problem("C5", ids, backend=orc.lib())
def loop(prob):
    b = prob.batch; b.mpc_totals(reset=True); prob._sync_options()
    its, xs, lsf = [], [b.get_x0()], []
    for t in range(T):
        if t == 1:
            prob.opts.shift, prob.opts.dual_reset = 1, False; prob._sync_options()
        st = b.newton_solve(init=True, game_id0=prob.game_id0 + t * 1000003)
        its.append(st["newton_iters"].copy()); lsf.append(st["ls_failures"].copy())
        b.mpc_advance(); xs.append(b.get_x0())
    return np.array(its), np.array(xs), np.array(lsf)
t0 = time.time(); ig, xg, lg = loop(pg); t1 = time.time(); io, xo, lo = loop(po); t2 = time.time()
print("gpu loop %.1fs oracle loop %.1fs" % (t1 - t0, t2 - t1))
dx = np.abs(xg - xo).max(axis=2)            # (T+1, B)
for g in range(len(ids)):
    mism = np.nonzero(ig[:, g] != io[:, g])[0]
    if mism.size:
        t = mism[0]
        print("game %2d first iter mismatch at step %3d: gpu %d oracle %d | ls_fail gpu %d orc %d | dx before %.2e after %.2e final %.2e | iters so far %d" % (g, t, ig[t, g], io[t, g], lg[t, g], lo[t, g], dx[t, g], dx[t + 1, g], dx[-1, g], ig[:t, g].sum()))
print("games with identical per-step counts:", int((ig == io).all(axis=0).sum()), "of", len(ids), " max dx over identical games %.2e" % dx[:, (ig == io).all(axis=0)].max())
print("mean iters/step", ig.mean(), " max", ig.max(), "total ls failures", lg.sum(), lo.sum())
