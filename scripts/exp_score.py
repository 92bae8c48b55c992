"""A/B of the score path on the metric batch (and a RANSAC-like P3P batch) under environment switches.
usage: exp_score.py "PGX_SCORE_GROUP_XCD=1 PGX_SCORE_SPLIT=4" "..." ...   (one context per configuration)"""
import os
import sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'progressive-x_amd')]
import numpy as np
from pyprogressivex import _lib, datasets

x1, x2, K, lab, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f
T2 = 9.0 / 4.0 * thr * thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
rng = np.random.default_rng(7)
smp = rng.integers(0, len(pts), (512, 3))
for r in range(0, 512, 2):
    smp[r] = rng.choice(np.nonzero(lab == 1 + (r // 2) % 16)[0], 3, replace=False)
smp = smp.astype(np.int32)
ref = None
for cfg in sys.argv[1:] or [""]:
    keys = []
    for kv in cfg.split():
        k, v = kv.split("=")
        os.environ[k] = v
        keys.append(k)
    ctx = _lib.Context(0)
    ctx.score_profile(True)
    ctx.set_points(_lib.PNP, pts)
    ctx.preference(gt[0], T2, slot=0)
    ctx.compound_update([0])
    ctx.score_upload(hyps)
    kt = []
    for it in range(25):
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2)
        kt.append(ctx.score_kernel_times())
    kt = np.median(np.array(kt[5:]), axis=0)
    if ref is None:
        ref = res
    same = bool(np.array_equal(res["counts"], ref["counts"]) and np.array_equal(res["values"], ref["values"]) and
                np.array_equal(res["shared"], ref["shared"]))
    ctx.solve_minimal(smp, fetch=False)
    kr = []
    for it in range(15):
        ctx.score_launch(T2, has_compound=True)
        ctx.score_fetch(exponent=2)
        kr.append(ctx.score_kernel_times())
    kr = np.median(np.array(kr[3:]), axis=0)
    print(f"{cfg or 'default':60s} metric: cull {kt[0]*1e3:6.1f} group {kt[1]*1e3:6.1f} exact {kt[3]*1e3:6.1f} finish {kt[2]*1e3:5.1f} "
          f"sum {kt.sum()*1e3:6.1f} us  ransac-like: cull {kr[0]*1e3:6.1f} group {kr[1]*1e3:6.1f} exact {kr[3]*1e3:6.1f} "
          f"sum {kr.sum()*1e3:6.1f} us  bitwise-same {same}", flush=True)
    ctx.close()
    for k in keys:
        os.environ.pop(k, None)
