"""T-LESS (tests/golden/scenes/tless*.txt): is the 1.9 deg of the second pose (recorded upstream: 0.95 deg, example_multi_pose_6d.ipynb:104-109) a solver deficit?\nFor each ground-truth pose: the returned pose nearest to it, the sites labelled with it, and the reprojection residuals of BOTH poses on those sites\n(CPU oracle harness).  usage: python scripts/probe_tless.py"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT+"/progressive-x_amd", ROOT+"/tests", ROOT+"/oracle", ROOT+"/scripts"]
import pyprogressivex as px
from pyprogressivex import _api, datasets
from oracle_ctx import OracleContext
import pgx_oracle as O
_api._ctx = OracleContext()
S=ROOT+"/tests/golden/scenes"
M=np.loadtxt(S+"/tless.txt", skiprows=1); K=np.loadtxt(S+"/tless_intrinsics.txt"); gt=np.loadtxt(S+"/tless_poses.txt", skiprows=1).reshape(-1,3,4)
with contextlib.redirect_stdout(io.StringIO()):
    P, lab = px.find6DPoses(M[:, :2], M[:, 2:5], K, 4.0, seed=1)
pts, f = datasets.normalize_pnp(M[:, :2], M[:, 2:5], K)
thr=4.0/f
nP=P.shape[0]//3
def ang(a,b): return float(np.degrees(np.arccos(np.clip(0.5*(np.trace(a[:,:3].T@b[:,:3])-1),-1,1))))
for gi,g in enumerate(gt):
    best=min(range(nP), key=lambda k: ang(g,P[3*k:3*k+3])+np.linalg.norm(g[:,3]-P[3*k:3*k+3][:,3]))
    Pk=P[3*best:3*best+3]
    inl=np.flatnonzero(lab==best)
    r_our=O.squared_residuals(O.PNP, pts[inl], Pk.reshape(-1)); r_gt=O.squared_residuals(O.PNP, pts[inl], g.reshape(-1))
    rg_all=O.squared_residuals(O.PNP, pts, g.reshape(-1)); ro_all=O.squared_residuals(O.PNP, pts, Pk.reshape(-1))
    T2=2.25*thr*thr
    print(f"gt {gi}: model {best} ang {ang(g,Pk):.2f} deg trans {np.linalg.norm(g[:,3]-Pk[:,3]):.1f}; labelled {len(inl)}; sum r^2 on them: ours {r_our.sum()*f*f:.1f} px^2, GT pose {r_gt.sum()*f*f:.1f} px^2; inliers(<T2) ours {int((ro_all<T2).sum())} GT pose {int((rg_all<T2).sum())}; at thr^2 ours {int((ro_all<thr*thr).sum())} GT {int((rg_all<thr*thr).sum())}")
