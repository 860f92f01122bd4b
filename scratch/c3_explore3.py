import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import algames_jl_amd as alg, oracle as orc
from algames_jl_amd import host
from algames_jl_amd.scenarios import _uniform

def build(ids, p, N, off, v0, rad_ca, reg0, outer, qpos=1.0, seed=100):
    model = host.UnicycleGame(p=p); dt=0.1; B=len(ids)
    ang = 2*np.pi*np.arange(p)/p + _uniform(seed, ids, p, -0.1, 0.1)
    x0=np.zeros((B,model.n)); x0[:,0:p]=np.cos(ang); x0[:,p:2*p]=np.sin(ang)
    tgt = ang + np.pi + off           # target direction rotated by `off` from the antipode
    tx, ty = np.cos(tgt), np.sin(tgt)
    head = np.arctan2(ty - x0[:,p:2*p], tx - x0[:,0:p])
    x0[:,2*p:3*p]=head; x0[:,3*p:4*p]=v0
    Q=[np.array([qpos,qpos,1.0,1.0]) for _ in range(p)]; R=[0.5*np.ones(2) for _ in range(p)]
    obj=host.GameObjective(Q,R,[np.zeros(4)]*p,[np.zeros(2)]*p,N,model)
    xf=np.zeros((B,p,4)); xf[:,:,0]=tx; xf[:,:,1]=ty; xf[:,:,2]=head; xf[:,:,3]=0.0
    obj.xf=xf; obj.Qdiag=np.broadcast_to(obj.Qdiag,(B,p,4)).copy(); obj.Rdiag=np.broadcast_to(obj.Rdiag,(B,p,2)).copy(); obj.uf=np.broadcast_to(obj.uf,(B,p,2)).copy()
    con=host.GameConstraintValues(host.ProblemSize(N,model))
    host.add_collision_avoidance(con, rad_ca); host.add_control_bound(con, np.ones(model.m), -np.ones(model.m))
    opts=host.Options(inner_print=False,outer_print=False,outer_iter=outer,inner_iter=20,ls_iter=25,reg_0=reg0,seed=seed)
    return host.GameProblem(N,dt,x0,model,opts,obj,con,backend=orc.lib())


ids=np.arange(64)
for (p,N) in ((4,50),(3,30)):
  for reg0 in (1e-3,1e-7):
        pr=build(ids,p,N,0.1,0.5,0.05,reg0,7)
        t=time.time(); alg.newton_solve(pr); dt=time.time()-t
        s=pr.stats.summary
        lam,_=pr.batch.get_con_duals()
        print(f"p={p} N={N} reg0={reg0}: conv {s['converged'].sum()}/{len(ids)} iters {s['newton_iters'].min()}-{s['newton_iters'].max()} outerit {s['outer_iters'].min()}-{s['outer_iters'].max()} lsfail {s['ls_failures'].sum()} opt {s['last']['opt_vio'].max():.1e} sta {s['last']['sta_vio'].max():.1e} con {s['last']['con_vio'].max():.1e} lam_max {lam.max():.2f} t={dt:.1f}s")
