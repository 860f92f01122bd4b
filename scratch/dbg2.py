import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import algames_jl_amd as alg
ids=np.arange(40,56)
pg = alg.scenarios.make_problem("C2", ids, N=12)
bad=0
for rep in range(30):
    alg.newton_solve(pg)
    s=pg.stats.summary
    if not (np.all(s['outer_iters']==1) and np.all(s['newton_iters']==3)): bad+=1; print('same-handle rep',rep,s['outer_iters'][:6], s['newton_iters'][:6])
print('same handle bad',bad)
bad=0; keep=[]
for rep in range(30):
    p2 = alg.scenarios.make_problem("C2", ids, N=12); keep.append(p2)
    alg.newton_solve(p2)
    s=p2.stats.summary
    if not (np.all(s['outer_iters']==1) and np.all(s['newton_iters']==3)): bad+=1; print('new-handle rep',rep,s['outer_iters'][:6])
    if rep%3==0: keep.clear()
print('new handle bad',bad)
