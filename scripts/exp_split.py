import os, sys, time
sys.path[:0] = ['/root/repo/progressive-x_amd']
import numpy as np
from pyprogressivex import _lib, datasets
x1, x2, K, _, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f; T2 = 9.0/4.0*thr*thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
ctx = _lib.Context(0); ctx.set_points(_lib.PNP, pts)
ctx.preference(gt[0], T2, slot=0); ctx.compound_update([0])
ctx.score_upload(hyps)
for name, t2 in (("T2", T2), ("T2*1e-6", T2*1e-6), ("T2*100", T2*100)):
    ms=[]
    for _ in range(8):
        ctx.timer_start(); ctx.score_launch(t2, has_compound=True); ms.append(ctx.timer_stop())
    res = ctx.score_fetch(exponent=2)
    print(name, "kernel ms", np.median(ms), "mean inliers", res["counts"].mean())
