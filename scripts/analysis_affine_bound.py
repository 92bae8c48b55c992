"""Prototype (numpy, CPU) of an affine-corrected group bound for the score path with the outliers pooled into their own
(u, v)-sorted groups: per group a robust affine fit (u, v) ~ A (X - c), exceptions moved out, final groups
re-fitted; the bound |u_i z_i - x_i| >= |ub z_c - x_c| - ||z_c a_u + ub p_z - p_x|| rho - ||a_u|| ||p_z|| rho^2 - eps_u (|z_c| + d_z)
is evaluated next to the shipped box bound on sampled groups and checked to never reject a group that holds an inlier.
Committed run: core groups 128 survivors with the affine bound alone, 84 with (affine OR box), ideal 19; junk groups 335 / 58."""
import os
import sys, time
sys.path[:0]=[os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'progressive-x_amd')]
import numpy as np
from pyprogressivex import datasets
x1, x2, K, lab, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f; T2 = 9.0/4.0*thr*thr; T=np.sqrt(T2)
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
P=hyps.reshape(-1,3,4)
n,d=pts.shape
def morton(X,bits):
    lo=X.min(0); hi=X.max(0)
    q=np.minimum(((X-lo)*((1<<bits)/(hi-lo))).astype(np.int64),(1<<bits)-1)
    key=np.zeros(len(X),np.int64); dd=X.shape[1]
    for k in range(dd):
        for b in range(bits):
            key |= ((q[:,k]>>b)&1) << (b*dd+dd-1-k)
    return key
order=np.argsort(morton(pts,6),kind='stable')
sp=pts[order]; slab=lab[order]
G=n//64
def group_fit(S):   # S: [G,64,5] -> affine fit of u,v on X; returns c, a_u, a_v, ub, vb, residuals
    X=S[:,:,2:5]; u=S[:,:,0]; v=S[:,:,1]
    c=0.5*(X.min(1)+X.max(1))
    dl=X-c[:,None,:]
    # LSQ with intercept: columns [1, dl]
    A=np.concatenate([np.ones((S.shape[0],64,1)),dl],axis=2)
    AtA=np.einsum('gik,gil->gkl',A,A)+1e-9*np.eye(4)
    Atu=np.einsum('gik,gi->gk',A,u); Atv=np.einsum('gik,gi->gk',A,v)
    cu=np.linalg.solve(AtA,Atu[...,None])[...,0]; cv=np.linalg.solve(AtA,Atv[...,None])[...,0]
    ru=u-np.einsum('gik,gk->gi',A,cu); rv=v-np.einsum('gik,gk->gi',A,cv)
    return c,dl,cu,cv,ru,rv
S=sp[:G*64].reshape(G,64,5)
c,dl,cu,cv,ru,rv=group_fit(S)
res=np.hypot(ru,rv)*1075
print("plain LSQ residual px: median of group-max",np.median(res.max(1)),"median of medians",np.median(np.median(res,1)))
# robust: iterate trimmed
w=np.ones((G,64),bool)
for it in range(3):
    X=S[:,:,2:5]; u=S[:,:,0]; v=S[:,:,1]
    A=np.concatenate([np.ones((G,64,1)),dl],axis=2)*w[...,None]
    AtA=np.einsum('gik,gil->gkl',A,A)+1e-9*np.eye(4)
    cu=np.linalg.solve(AtA,np.einsum('gik,gi->gk',A,u*w)[...,None])[...,0]
    cv=np.linalg.solve(AtA,np.einsum('gik,gi->gk',A,v*w)[...,None])[...,0]
    A1=np.concatenate([np.ones((G,64,1)),dl],axis=2)
    ru=u-np.einsum('gik,gk->gi',A1,cu); rv=v-np.einsum('gik,gk->gi',A1,cv)
    res=np.hypot(ru,rv)*1075
    med=np.median(res,1,keepdims=True)
    w=res<np.maximum(4*med,3.0)
core=w.copy()
print("core fraction",core.mean(),"true inlier fraction",(slab[:G*64]>0).mean(), "core&inlier",(core.reshape(-1)&(slab[:G*64]>0)).mean())
# regroup: core first in morton order, then junk sorted by uv morton
idx=np.arange(G*64)
core_idx=idx[core.reshape(-1)]
junk_idx=np.concatenate([idx[~core.reshape(-1)],np.arange(G*64,n)])
jk=morton(sp[junk_idx][:,:2],12)
junk_idx=junk_idx[np.argsort(jk,kind='stable')]
new=np.concatenate([core_idx,junk_idx])
sp2=sp[new]; lab2=slab[new]
G2=n//64
S2=sp2[:G2*64].reshape(G2,64,5)
c,dl,cu,cv,ru,rv=group_fit(S2)
eps_u=np.abs(ru).max(1); eps_v=np.abs(rv).max(1)
rho=np.sqrt((dl**2).sum(2).max(1))
print("final groups: eps_u px median",np.median(eps_u)*1075,"p90",np.percentile(eps_u,90)*1075,"rho median",np.median(rho))
# evaluate survivors on sampled groups with the affine bound
rng=np.random.default_rng(0)
ncore_groups=len(core_idx)//64
def survivors(gs):
    ta=ti=tb=0
    for g in gs:
        p=S2[g]; X=p[:,2:5]
        proj=np.einsum('mij,nj->mni',P[:,:,:3],X)+P[:,None,:,3]
        uu=proj[...,0]/proj[...,2]; vv=proj[...,1]/proj[...,2]
        r2=(p[None,:,0]-uu)**2+(p[None,:,1]-vv)**2
        ideal=(r2<T2).any(1)
        cc=P[:,:,:3]@c[g]+P[:,:,3]      # M,3  (x_c,y_c,z_c)
        px,py,pz=P[:,0,:3],P[:,1,:3],P[:,2,:3]
        ub,au=cu[g,0],cu[g,1:]; vb,av=cv[g,0],cv[g,1:]
        nz=np.linalg.norm(pz,axis=1); dz=nz*rho[g]
        zs=np.abs(cc[:,2])+dz
        gu=cc[:,2:3]*au[None,:]+ub*pz-px; gv=cc[:,2:3]*av[None,:]+vb*pz-py
        lbu=np.abs(ub*cc[:,2]-cc[:,0])-np.linalg.norm(gu,axis=1)*rho[g]-np.linalg.norm(au)*nz*rho[g]**2-eps_u[g]*zs
        lbv=np.abs(vb*cc[:,2]-cc[:,1])-np.linalg.norm(gv,axis=1)*rho[g]-np.linalg.norm(av)*nz*rho[g]**2-eps_v[g]*zs
        rej=(lbu>T*zs)|(lbv>T*zs)
        # box bound for comparison
        ub2=0.5*(p[:,0].min()+p[:,0].max()); vb2=0.5*(p[:,1].min()+p[:,1].max())
        ru2=np.abs(p[:,0]-ub2).max(); rv2=np.abs(p[:,1]-vb2).max()
        nx=np.linalg.norm(px,axis=1); ny=np.linalg.norm(py,axis=1)
        ex=np.abs(ub2*cc[:,2]-cc[:,0]); ey=np.abs(vb2*cc[:,2]-cc[:,1])
        mx=ru2*zs+abs(ub2)*dz+nx*rho[g]; my=rv2*zs+abs(vb2)*dz+ny*rho[g]
        rejb=(ex-mx>T*zs)|(ey-my>T*zs)
        assert not (rej&ideal).any(), "affine bound rejected a group with inliers!"
        ta+=(~rej).sum(); ti+=ideal.sum(); tb+=(~(rej|rejb)).sum()
    return ta/len(gs),ti/len(gs),tb/len(gs)
gs=rng.choice(ncore_groups-1,300,replace=False)
print("core groups: survivors affine %.1f ideal %.1f  (affine OR box) %.1f"%survivors(gs))
gs=ncore_groups+1+rng.choice(G2-ncore_groups-2,100,replace=False)
print("junk groups: survivors affine %.1f ideal %.1f  (affine OR box) %.1f"%survivors(gs))
print("groups core",ncore_groups,"junk",G2-ncore_groups)
