#!/usr/bin/env python3
"""the inlier / outlier graph cut of GC-RANSAC's local optimisation (pgx_gc_inliers) at C4 size (10^6 correspondences): wall time per cut of
ground-truth poses, for rocprofv3 kernel stats.  usage: c4_cut.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
from pyprogressivex import _lib, datasets
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x1, x2, K, gt, poses = datasets.make_poses(seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
ctx = _lib.Context(0)
ctx.set_points(_lib.PNP, pts)
ctx.graph_build(np.column_stack([x1, x2]), _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
T2 = (4.0 / f) ** 2 * 2.25
for m in poses[:3]:
    ctx.gc_inliers(np.asarray(m).reshape(-1), T2, 0.1)
ctx.sync()
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    inl = ctx.gc_inliers(np.asarray(poses[r % len(poses)]).reshape(-1), T2, 0.1)
    ts.append(time.perf_counter() - t0)
print("cut ms: median %.3f  min %.3f  max %.3f; inliers of the last %d" % (1e3 * np.median(ts), 1e3 * min(ts), 1e3 * max(ts), len(inl)))
ctx.close()
