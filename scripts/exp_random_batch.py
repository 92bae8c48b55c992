"""Score-path timing on a RANSAC-like batch: P3P hypotheses from random minimal samples (mostly garbage poses) instead of
the metric batch's perturbed ground truth.  Prints kernel ms for the group-major path and the chunked kernel."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'progressive-x_amd')]
import numpy as np
from pyprogressivex import _lib, _estimators, datasets
x1, x2, K, _, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f; T2 = 9.0/4.0*thr*thr
rng = np.random.default_rng(3)
est = _estimators.PnPEstimator()
samples = rng.integers(0, len(pts), (1500, 3))
# half of the samples from single objects (all-inlier samples), as a RANSAC run would eventually draw
lab = np.repeat(np.arange(16), 50000)
for s in range(0, 1500, 2):
    o = rng.integers(0, 16); samples[s] = rng.choice(np.nonzero(lab == o)[0], 3, replace=False)
models, src = est.minimal(pts, samples)
hyps = models[:2048] if len(models) >= 2048 else np.vstack([models, models])[:2048]
print("hypotheses", hyps.shape, "finite", np.isfinite(hyps).all())
ctx = _lib.Context(0); ctx.set_points(_lib.PNP, pts)
ctx.score_upload(hyps)
ms = []
for _ in range(8):
    ctx.timer_start(); ctx.score_launch(T2); ms.append(ctx.timer_stop())
res = ctx.score_fetch()
print(os.environ.get("PGX_NO_GROUP", "0"), "kernel ms", np.median(ms), "mean inliers", res["counts"].mean(), "max", res["counts"].max())
