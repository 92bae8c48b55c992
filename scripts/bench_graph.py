#!/usr/bin/env python3
"""Neighbourhood graph build: pgx_graph_build on the GPU vs the host kd-tree construction it replaced
(scipy cKDTree, all cores) on the BASELINE configs' shapes.  One JSON line per config; numbers go to DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
from pyprogressivex import _lib, datasets  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_graph as _graph  # noqa: E402  (scipy kd-tree reference constructions)


def run(ctx, name, pts, kind, radius, k):
    ctx.graph_build(pts[:1000], kind, radius=radius, k=k)   # warm up
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        arcs = ctx.graph_build(pts, kind, radius=radius, k=k, fetch=False)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    g = ctx.graph_build(pts, kind, radius=radius, k=k)
    t_fetch = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = _graph.flann_like_graph(pts, radius, k) if kind == _lib.GRAPH_KNN_IN_BALL else _graph.knn_graph(pts, k)
    t_host = time.perf_counter() - t0
    same = all(np.array_equal(a, b) for a, b in zip(g, ref))
    print(json.dumps(dict(config=name, n=int(pts.shape[0]), d=int(pts.shape[1]), k=k, radius=radius, arcs=int(arcs),
                          gpu_build_ms=1e3 * float(np.median(ts)), gpu_build_and_fetch_ms=1e3 * t_fetch,
                          host_kdtree_ms=1e3 * t_host, identical_to_host=bool(same),
                          speedup=t_host / float(np.median(ts)))), flush=True)


if __name__ == "__main__":
    ctx = _lib.Context(0)
    pts, _, _ = datasets.make_homographies(seed=0)
    run(ctx, "C2 homography 5k x 4-D, ball 200, k 5", pts, _lib.GRAPH_KNN_IN_BALL, 200.0, 5)
    pts, _, _ = datasets.make_two_view_motions(seed=0)
    run(ctx, "C3 two-view 1e5 x 4-D, ball 200, k 5", pts, _lib.GRAPH_KNN_IN_BALL, 200.0, 5)
    segs, _, _ = datasets.make_vanishing_points(seed=0)
    mid = np.ascontiguousarray(0.5 * (segs[:, :2] + segs[:, 2:]))
    run(ctx, "C5 VP 2e5 midpoints, k-NN 8", mid, _lib.GRAPH_KNN, 0.0, 8)
    run(ctx, "C5 VP 2e5 x 4-D segments, ball 10, k 5", segs, _lib.GRAPH_KNN_IN_BALL, 10.0, 5)
    x1, x2, K, _, _ = datasets.make_poses(seed=0)
    run(ctx, "C4 PnP 1e6 x 5-D, ball 20, k 5", np.column_stack([x1, x2]), _lib.GRAPH_KNN_IN_BALL, 20.0, 5)
