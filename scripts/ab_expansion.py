#!/usr/bin/env python3
"""Best-of-N wall time of one alpha-expansion from zeros on the BASELINE shapes (first-cycle memo off: every repetition solves every
min-cut), for A/B runs of the schedule switches (PGX_MF_*: read once per process).  usage: ab_expansion.py [C1 C2] C3 C5 C4 [--reps 4]"""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
os.environ.setdefault("PGX_MF_MEMO", "0")
from pyprogressivex import _lib, datasets  # noqa: E402


def run(name, mt, pts, models, thr, lam, h, graph, reps):
    ctx = _lib.Context(0)
    ctx.set_points(mt, pts)
    ctx.graph_build(graph[0], graph[1], radius=graph[2], k=graph[3])
    n = pts.shape[0]
    ctx.pearl_unary(models, thr, lam)
    ts = []
    for _ in range(reps):
        ctx.set_labels(np.zeros(n, np.int32))
        ctx.sync()
        t0 = time.perf_counter()
        eq, e, cycles = ctx.expansion(lam, h)
        ts.append(time.perf_counter() - t0)
    st = ctx.expansion_stats()
    print(json.dumps(dict(config=name, best_ms=round(1e3 * min(ts), 2), all_ms=[round(1e3 * t, 1) for t in ts], cycles=cycles,
                          sweeps=st["sweeps"] // reps, relabels=st["global_relabels"] // reps, levels=st["bfs_levels"] // reps,
                          crc=zlib.crc32(ctx.get_labels().tobytes()), schedule={k: v // reps for k, v in ctx.expansion_schedule().items()}, paths=ctx.expansion_paths(), env={k: v for k, v in os.environ.items() if k.startswith("PGX_MF")})), flush=True)
    ctx.close()


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C3", "C5", "C4"]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 4
    if "C2" in which:
        pts, gt, models = datasets.make_homographies(seed=0)
        run("C2", _lib.HOMOGRAPHY, pts, models, 3.0, 0.05, 10.0, (pts, _lib.GRAPH_KNN_IN_BALL, 200.0, 5), reps)
    if "C1" in which:
        pts, gt, models = datasets.make_lines(seed=0)
        run("C1", _lib.LINE2D, pts, models, 2.0, 0.05, 10.0, (pts, _lib.GRAPH_KNN_IN_BALL, 50.0, 5), reps)
    if "C3" in which:
        pts, gt, models = datasets.make_two_view_motions(seed=0)
        run("C3", _lib.FUNDAMENTAL, pts, models, 0.75, 0.1, 14.0, (pts, _lib.GRAPH_KNN_IN_BALL, 50.0, 5), reps)
    if "C5" in which:
        pts, gt, models = datasets.make_vanishing_points(seed=0)
        run("C5", _lib.VANISHING_POINT, pts, models, 1.5, 0.1, 20.0, (0.5 * (pts[:, :2] + pts[:, 2:]), _lib.GRAPH_KNN, 0.0, 8), reps)
    if "C4" in which:
        x1, x2, K, gt, poses = datasets.make_poses(seed=0)
        pts, f = datasets.normalize_pnp(x1, x2, K)
        run("C4", _lib.PNP, pts, poses[:10], 4.0 / f, 0.1, 6.0, (np.column_stack([x1, x2]), _lib.GRAPH_KNN_IN_BALL, 20.0, 5), reps)
