import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'oracle'))
import numpy as np
import algames_jl_amd as alg, oracle as orc
np.set_printoptions(linewidth=200)
ids=np.arange(40,56)
for rep in range(3):
    pg = alg.scenarios.make_problem("C2", ids, N=12)
    alg.newton_solve(pg)
    s=pg.stats.summary
    print(rep, 'outer', s['outer_iters'], 'iters', s['newton_iters'], 'rec', s['records'], 'conv', s['converged'], 'status', s['status'])
po = alg.scenarios.make_problem("C2", ids, N=12, backend=orc.lib()); alg.newton_solve(po)
so=po.stats.summary
print('orc outer', so['outer_iters'], 'iters', so['newton_iters'], 'rec', so['records'])
print('traj diff', np.abs(pg.batch.get_traj()-po.batch.get_traj()).max(axis=1))
print(pg.stats.history(3)); print(po.stats.history(3))
