import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import oracle as orc

def structured_direction(b, J, res):
    n, m, p, N, mi, bb = b.n, b.m, b.p, b.N, b.mi, b.b
    vx = lambda i,k: i*(N-1)*(n+mi) + k*(n+mi)
    vu = lambda i,k: vx(i,k) + n
    vd = lambda k: p*(N-1)*(n+mi) + k*n
    hx = lambda k: k*bb
    hu = lambda k,i: k*bb + n + i*mi
    hl = lambda k,i: k*bb + n + m + i*n
    def blocks(k):
        Qh = [J[vx(i,k):vx(i,k)+n, hx(k):hx(k)+n] for i in range(p)]
        Rh = [J[vu(i,k):vu(i,k)+mi, hu(k,i):hu(k,i)+mi] for i in range(p)]
        A = J[vd(k):vd(k)+n, hx(k-1):hx(k-1)+n] if k>=1 else np.zeros((n,n))
        B = [J[vd(k):vd(k)+n, hu(k,i):hu(k,i)+mi] for i in range(p)]
        rx = [res[vx(i,k):vx(i,k)+n] for i in range(p)]
        ru = [res[vu(i,k):vu(i,k)+mi] for i in range(p)]
        rd = res[vd(k):vd(k)+n]
        return Qh,Rh,A,B,rx,ru,rd
    K = [None]*(N-1); kap=[None]*(N-1)
    P=None; s=None; F=None; f=None; Anext=None
    for k in range(N-2,-1,-1):
        Qh,Rh,A,B,rx,ru,rd = blocks(k)
        if k == N-2:
            P = [Qh[i].copy() for i in range(p)]; s=[rx[i].copy() for i in range(p)]
        else:
            P = [Qh[i] + Anext.T @ (P[i] @ F) for i in range(p)]
            s = [rx[i] + Anext.T @ (Pold_i @ f + s_i) for i,(Pold_i,s_i) in enumerate(zip(Pold,s))]
        Bj = np.hstack(B)   # n x m, player-blocked columns
        W = np.zeros((m,m)); rhsA = np.zeros((m,n)); rhsb = np.zeros(m)
        for i in range(p):
            V = B[i].T @ P[i]            # mi x n
            W[i*mi:(i+1)*mi,:] = V @ Bj
            W[i*mi:(i+1)*mi, i*mi:(i+1)*mi] += Rh[i]
            rhsA[i*mi:(i+1)*mi,:] = V @ A
            rhsb[i*mi:(i+1)*mi] = V @ rd + B[i].T @ s[i] + ru[i]
        Y = np.linalg.solve(W, np.hstack([rhsA, rhsb[:,None]]))
        K[k] = -Y[:,:n]; kap[k] = -Y[:,n]
        F = A + Bj @ K[k]; f = rd + Bj @ kap[k]
        Anext = A; Pold = P
    # forward
    dx = np.zeros(n); DX=[None]*(N-1); DU=[None]*(N-1)
    for k in range(N-1):
        Qh,Rh,A,B,rx,ru,rd = blocks(k)
        du = K[k] @ dx + kap[k]
        dxn = A @ dx + np.hstack(B) @ du + rd
        DX[k]=dxn; DU[k]=du; dx=dxn
    # costate backward
    DL=[[None]*p for _ in range(N-1)]
    for k in range(N-2,-1,-1):
        Qh,Rh,A,B,rx,ru,rd = blocks(k)
        for i in range(p):
            v = Qh[i] @ DX[k] + rx[i]
            if k < N-2:
                An = J[vd(k+1):vd(k+1)+n, hx(k):hx(k)+n]
                v = v + An.T @ DL[k+1][i]
            DL[k][i]=v
    d = np.zeros(b.S)
    for k in range(N-1):
        d[hx(k):hx(k)+n]=DX[k]; d[hu(k,0):hu(k,0)+m]=DU[k]
        for i in range(p): d[hl(k,i):hl(k,i)+n]=DL[k][i]
    return d

rng=np.random.default_rng(0)
for model,p,N in [(0,2,8),(1,2,6),(1,3,7),(0,3,10)]:
    b = orc.OracleBatch(model,p,N,0.1,1)
    ni=b.n//p
    b.set_lqr(1+rng.random((p,ni)),0.5+rng.random((p,b.mi)),rng.random((p,ni)),rng.random((p,b.mi)))
    b.set_x0(rng.random(b.n))
    b.add_collision_cost(np.full(p,3.0),np.full(p,2.0))
    b.add_collision_avoidance(np.full(p,0.4))
    b.add_control_bound(np.full(b.m,0.5),np.full(b.m,-0.5))
    b.set_traj(rng.random((1,b.traj_len)))
    b.set_con_duals(rng.random((1,b.con_len)),2.0*np.ones((1,b.con_len)))
    reg=1e-3
    d0,st=b.newton_direction(reg)
    J=b.residual_jacobian(reg)[0]; res=b.residual()[0][0]
    d1=structured_direction(b,J,res)
    print(model,p,N,'max diff',np.abs(d0[0]-d1).max(),'scale',np.abs(d0).max(), 'lin resid', np.abs(J@d1+res).max())
