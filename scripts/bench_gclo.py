#!/usr/bin/env python3
"""Cost of one round of the graph-cut local optimisation (pgx_gc_labeling + the inner-RANSAC refits + one scoring launch)
on the C4 / C5 synthetic configs, with a ground-truth model as the so-far-best."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
from pyprogressivex import _estimators, _lib, datasets  # noqa: E402


def run(name, mt, pts, gpts, model, thr, lam, est, radius):
    ctx = _lib.Context(0)
    ctx.set_points(mt, pts)
    ctx.graph_build(gpts, _lib.GRAPH_KNN_IN_BALL, radius=radius, k=5, fetch=False)
    T2 = 2.25 * thr * thr
    ctx.gc_labeling(model, T2, lam)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        flags = ctx.gc_labeling(model, T2, lam)
    t_cut = (time.perf_counter() - t0) / reps
    st = ctx.expansion_stats() if hasattr(ctx, "expansion_stats") else {}
    inl = np.nonzero(flags)[0]
    rng = np.random.default_rng(0)
    size = 7 * est.sample_size
    t0 = time.perf_counter()
    picks = np.array([np.sort(rng.choice(inl, size, replace=False)) for _ in range(50)])
    cands = [m for fits in est.nonminimal_batch(ctx, picks, None, init=model) for m in fits]
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in range(50):
        est.nonminimal(ctx, ("index", picks[b]), None, init=model)
    t_single = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.score(np.asarray(cands), T2, has_compound=False, exponent=2)
    t_score = time.perf_counter() - t0
    print(f"{name}: n={len(pts)} inliers={len(inl)} cut {t_cut * 1e3:.2f} ms, 50 refits batched {t_fit * 1e3:.2f} ms / one by one {t_single * 1e3:.2f} ms "
          f"({len(cands)} candidates), scoring {t_score * 1e3:.2f} ms  stats={st}", flush=True)
    ctx.close()


if __name__ == "__main__":
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    norm, f = datasets.normalize_pnp(x1, x2, K)
    run("C4 pnp", _lib.PNP, norm, np.column_stack([x1, x2]), poses[0], 4.0 / f, 0.1, _estimators.PnPEstimator(), 20.0)
    pts, gt, vps = datasets.make_vanishing_points(seed=0)
    run("C5 vp", _lib.VANISHING_POINT, pts, pts, vps[0], 1.5, 0.05, _estimators.VanishingPointEstimator(), 10.0)
