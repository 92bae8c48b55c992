#!/usr/bin/env python3
"""Launch-geometry sweep of the group-major scoring kernel on the C5 vanishing-point batch and the C3 Sampson batch (waves per group,
the candidate count from which a step is evaluated in place instead of through the queue): results are bitwise independent of both
(tests/test_gpu_parity.py geometry-invariance tests); this prices them after the Hough ordering of round 6."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "progressive-x_amd"))
from pyprogressivex import _lib, datasets


def batch(ctx, gt, n, m, S, rng):
    smp = rng.integers(0, n, (S, m))
    for r in range(0, S, 2):
        idx = np.flatnonzero(gt == 1 + (r // 2) % int(gt.max()))
        smp[r] = rng.choice(idx, m, replace=False)
    ctx.solve_minimal(smp.astype(np.int32), fetch=False)


def measure(ctx, T2, reps=40):
    buf = ctx.score_buffers()
    ctx.score_profile(0)
    for _ in range(10):
        ctx.score_launch(T2, has_compound=False); ctx.score_fetch(exponent=2, out=buf)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        ctx.score_launch(T2, has_compound=False); ctx.score_fetch(exponent=2, out=buf)
    ctx.sync(); step = (time.perf_counter() - t0) / reps * 1e3
    ctx.score_profile(2)
    ks = []
    for _ in range(4):
        ctx.score_launch(T2, has_compound=False); ctx.score_fetch(exponent=2, out=buf); ks.append(ctx.score_kernel_times())
    return step, np.mean(np.array(ks[1:]), axis=0)


for name, mt, make, m, S, thr in (("C5 vanishing points", _lib.VANISHING_POINT, datasets.make_vanishing_points, 2, 2048, 1.5),
                                  ("C3 Sampson", _lib.FUNDAMENTAL, datasets.make_two_view_motions, 7, 683, 0.75)):
    pts, gt, _ = make(seed=0)
    ctx = _lib.Context(0)
    ctx.set_points(mt, pts)
    batch(ctx, gt, len(pts), m, S, np.random.default_rng(7))
    T2 = 2.25 * thr * thr
    st = ctx.score_stats(T2, has_compound=False)
    print(name, {k: st[k] for k in ("group_pairs", "surviving_group_steps", "exact_evaluations", "inlier_pairs")}, flush=True)
    for split in (0, 16):
        for dense in (32,):
            ctx.score_debug_geometry(split=split, dense_min=dense)
            step, k = measure(ctx, T2)
            print(f"  split {split:2d} dense_min {dense:2d}: step {step:.3f} ms, cull {1e3 * k[0]:.0f} us, group {1e3 * k[1]:.0f} us", flush=True)
    ctx.close()
