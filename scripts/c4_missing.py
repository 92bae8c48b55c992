#!/usr/bin/env python3
"""Why does the cap-lifted C4 call stop at 12-14 of 16 objects?  For the ground-truth objects no returned model owns: how their GT pose
scores against the final compound instance, and what P3P makes of all-inlier samples drawn from them."""
import contextlib, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
import pyprogressivex as px
from pyprogressivex import _api, _lib, datasets

x1, x2, K, gt, poses = datasets.make_poses(seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    P, lab = px.find6DPoses(x1, x2, K, seed=1, minimum_point_number=5000, max_iters=int(os.environ.get("C4_MAX_ITERS", "2048")), max_outer_iterations=20)
k = P.shape[0] // 3
owners = [int(np.bincount(gt[lab == m], minlength=17)[1:].argmax()) + 1 for m in range(k)]
missing = [o for o in range(1, 17) if o not in owners]
print("models", k, "owners", owners, "missing objects", missing)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f
T2 = 2.25 * thr * thr
ctx = _api._ctx
ctx.set_points(_lib.PNP, pts)
for slot in range(k):
    ctx.preference(P[3 * slot:3 * slot + 3].reshape(-1), T2, slot)
ctx.compound_update(list(range(k)))
gtm = np.asarray(poses, dtype=np.float64).reshape(16, 12)
r = ctx.score(gtm, T2, has_compound=True, exponent=2)
for o in range(1, 17):
    print(f"object {o:2d} {'MISSING' if o in missing else 'found  '} GT pose: inliers {int(r['counts'][o-1]):6d} value {r['values'][o-1]:10.1f} shared {r['shared'][o-1]:10.1f} score {r['scores'][o-1]:12.1f}  labelled-as histogram {np.bincount(lab[gt == o], minlength=k + 1).tolist()}")
rng = np.random.default_rng(0)
for o in missing:
    idx = np.nonzero(gt == o)[0]
    smp = np.array([rng.choice(idx, 3, replace=False) for _ in range(64)], dtype=np.int32)
    models = ctx.solve_minimal(smp)
    ok = ~np.isnan(models[:, 0])
    s = ctx.score(models[ok], T2, has_compound=True, exponent=2)
    print(f"object {o}: 64 all-inlier P3P samples -> {int(ok.sum())} finite roots; best root inliers {int(s['counts'].max())}, best score {s['scores'].max():.1f}; roots with > 25 000 inliers: {int((s['counts'] > 25000).sum())}")
