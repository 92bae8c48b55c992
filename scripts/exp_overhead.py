import sys, time
sys.path[:0]=['/root/repo/progressive-x_amd']
import numpy as np
from pyprogressivex import _lib, datasets, parallel
x1, x2, K, _, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f; T2 = 9.0/4.0*thr*thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
ctx = _lib.Context(0); ctx.set_points(_lib.PNP, pts)
ctx.preference(gt[0], T2, slot=0); ctx.compound_update([0]); ctx.score_upload(hyps)
for _ in range(5):
    ctx.score_launch(T2, has_compound=True); ctx.score_fetch(2)
N=200
t0=time.perf_counter()
for _ in range(N): ctx.score_launch(T2, has_compound=True); ctx.sync()
t1=time.perf_counter()
for _ in range(N): ctx.score_launch(T2, has_compound=True); r=ctx.score_fetch(2)
t2=time.perf_counter()
for _ in range(N): parallel.select_best(r["scores"], r["counts"])
t3=time.perf_counter()
for _ in range(N): ctx.timer_start(); ctx.score_launch(T2, has_compound=True); ctx.timer_stop()
t4=time.perf_counter()
print("launch+sync us", 1e6*(t1-t0)/N, " launch+fetch us", 1e6*(t2-t1)/N, " select us", 1e6*(t3-t2)/N, " timed launch us", 1e6*(t4-t3)/N)
