#!/usr/bin/env python3
"""Wall time per libpgx call at the size of the reference's own scenes (unionhouse: 332 correspondences, homographies), one line
per entry point a findHomographies call spends its time in: what a call costs when the data is tiny and everything is launch,
copy and synchronisation overhead.  usage: bench_small_calls.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
from pyprogressivex import _lib, datasets

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
corrs, g = datasets.load_points_with_labels(os.path.join(ROOT, "tests", "golden", "scenes", "unionhouse.txt"))
pts = np.ascontiguousarray(corrs[:, :4], dtype=np.float64)
n = pts.shape[0]
ctx = _lib.Context(0)
ctx.set_points(_lib.HOMOGRAPHY, pts)
ctx.graph_build(pts, _lib.GRAPH_KNN_IN_BALL if hasattr(_lib, "GRAPH_KNN_IN_BALL") else 0, radius=200.0, k=5)
H = np.eye(3).reshape(-1) + 1e-3 * np.random.default_rng(0).normal(size=9)
models = np.stack([H * (1 + 0.01 * k) for k in range(4)])
T2 = 16.0
idx = np.arange(0, n, 3, dtype=np.int32)
norm = np.array([500.0, 400.0, 0.01, 500.0, 400.0, 0.01])
labels = (np.arange(n) % 4).astype(np.int32)
ctx.set_labels(labels)
ctx.pearl_unary(models, 4.0, 0.05)


def timeit(name, fn):
    for _ in range(20):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:32s} {1e6 * dt:8.1f} us per call")


timeit("score (1 model)", lambda: ctx.score(H, T2))
timeit("score (1 model, masks)", lambda: ctx.score(H, T2, want_masks=True))
timeit("score_inliers", lambda: ctx.score_inliers(0))
timeit("gram DLT (110 indices)", lambda: ctx.gram(_lib.GRAM_DLT_H, ("index", idx), params=norm))
timeit("gram DLT (label 1)", lambda: ctx.gram(_lib.GRAM_DLT_H, ("label", 1), params=norm))
timeit("gram_labels DLT (4 labels)", lambda: ctx.gram_labels(_lib.GRAM_DLT_H, 4, params=np.tile(norm, (4, 1))))
timeit("residual_sums (4 models)", lambda: ctx.residual_sums(models))
timeit("gc_inliers", lambda: ctx.gc_inliers(H, T2, 0.05))
timeit("pearl_unary (4 models)", lambda: ctx.pearl_unary(models, 4.0, 0.05))


def expansion():
    ctx.set_labels(labels)
    ctx.expansion(0.05, 10.0)


timeit("set_labels + expansion", expansion)
timeit("set_labels", lambda: ctx.set_labels(labels))
timeit("get_labels", lambda: ctx.get_labels())
timeit("bucket", lambda: ctx.bucket(5))
timeit("numpy eigh 9x9", lambda: np.linalg.eigh(np.eye(9) + 0.1))
ctx.close()
