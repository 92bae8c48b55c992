"""Host-side split of one bench step (call by call), 300 iterations."""
import sys, time
sys.path[:0] = ['/root/repo/progressive-x_amd']
import numpy as np
from pyprogressivex import _lib, datasets, parallel
x1, x2, K, _, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f; T2 = 9.0 / 4.0 * thr * thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
ctx = _lib.Context(0); ctx.set_points(_lib.PNP, pts)
ctx.preference(gt[0], T2, slot=0); ctx.compound_update([0]); ctx.score_upload(hyps)
buf = ctx.score_buffers()
names = ["timer_start", "score_launch", "timer_mark", "score_fetch", "timer_elapsed", "select_best"]
acc = np.zeros(len(names)); N = 300
for it in range(N + 10):
    t = [time.perf_counter()]
    ctx.timer_start(); t.append(time.perf_counter())
    ctx.score_launch(T2, has_compound=True); t.append(time.perf_counter())
    ctx.timer_mark(); t.append(time.perf_counter())
    res = ctx.score_fetch(exponent=2, out=buf); t.append(time.perf_counter())
    k = ctx.timer_elapsed(); t.append(time.perf_counter())
    parallel.select_best(res["scores"], res["counts"]); t.append(time.perf_counter())
    if it >= 10:
        acc += np.diff(t)
print({n: round(1e6 * a / N, 1) for n, a in zip(names, acc)}, "total us", round(1e6 * acc.sum() / N, 1), "kernel us", round(1e3 * k, 1))
