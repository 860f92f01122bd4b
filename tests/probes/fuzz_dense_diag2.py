import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import algames_jl_amd as alg, oracle as orc
import test_gpu_fuzz as F
np.set_printoptions(linewidth=220, precision=5)
seed = int(sys.argv[1]); game = int(sys.argv[2])
rng = np.random.default_rng(seed)
fam = F.DENSE_FAMILIES[(seed - 400000) % len(F.DENSE_FAMILIES)]
g, o, tag = F._random_pair(alg, orc, rng, ext=bool((seed - 400000) % 2), force=fam)
print(tag)
g.init_traj(game_id0=7); o.init_traj(game_id0=7)
zg, zo = g.get_traj(0), o.get_traj(0)
print("init traj max diff", np.abs(zg - zo).max())
g.set_traj(zo); 
rg, ng = g.residual(0, 0.0); ro, no = o.residual(0, 0.0)
print("norms", ng, no)
d = np.abs(rg - ro)[game]
idx = np.nonzero(d > 1e-10 * (1 + np.abs(ro[game])))[0]
print("rows differing", idx, rg[game][idx], ro[game][idx])
p, n, N = g.p, g.n, g.N
X = g.split_traj(zo)[0][game]
print("positions knot 2:", X[1, :3 * p].reshape(3, p))
P3 = X[1, :3 * p].reshape(3, p)
for i in range(p):
    for j in range(p):
        if i != j:
            dl = P3[:, i] - P3[:, j]
            print(i, j, "planar dist %.4f  3d dist %.4f" % (np.hypot(dl[0], dl[1]), np.linalg.norm(dl)))
vg, vo = g.kat_evaluate_con() if hasattr(g, "kat_evaluate_con") else None, o.kat_evaluate_con()
print("oracle con vals", vo[game])
