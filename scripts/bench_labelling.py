#!/usr/bin/env python3
"""Labelling throughput (SURVEY §8d metric 3): PEARL unary table + alpha-expansion on the GPU vs the CPU oracle (Dinic)
on the BASELINE configs' shapes.  Prints one JSON line per config.  Not the driver's bench — numbers go to DESIGN.md."""
import json
import os
import zlib
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle")]
from pyprogressivex import _lib, datasets  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_graph as _graph  # noqa: E402  (scipy kd-tree reference constructions)
import pgx_oracle as O  # noqa: E402


def run(name, mt, pts, models, thr, lam, h, graph, with_oracle=True):
    """graph: a host CSR (off, idx, mult), or (points, kind, radius, k) to build it on the device as the API does (the sites
    of the tile-resident min-cut are then ordered along the Morton curve of those points)"""
    ctx = _lib.Context(0)
    ctx.set_points(mt, pts)
    if len(graph) == 4:
        t0 = time.perf_counter()
        graph = ctx.graph_build(graph[0], graph[1], radius=graph[2], k=graph[3])
        print(json.dumps(dict(note=name + ": device graph build + fetch", seconds=time.perf_counter() - t0)), flush=True)
    else:
        ctx.set_graph(*graph)
    n = pts.shape[0]
    t0 = time.perf_counter()
    ctx.pearl_unary(models, thr, lam)
    ctx.sync()
    t_unary = time.perf_counter() - t0
    ctx.set_labels(np.zeros(n, np.int32))
    t0 = time.perf_counter()
    eq, e, cycles = ctx.expansion(lam, h)
    t_gpu = time.perf_counter() - t0
    st = ctx.expansion_stats()
    labels = ctx.get_labels()
    out = dict(config=name, n=n, K=len(models), arcs=int(graph[0][-1]), lam=lam, h=h, gpu_unary_ms=1e3 * t_unary,
               gpu_expansion_ms=1e3 * t_gpu, cycles=cycles, energy=e, **st,
               gpu_ms_per_mincut=1e3 * t_gpu / max(1, st["mincuts"]), labels_crc=zlib.crc32(labels.tobytes()))
    if with_oracle and "--no-oracle" not in sys.argv:
        t0 = time.perf_counter()
        Dq = O.unary_q(mt, pts, models, thr, lam)
        t_ou = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref, re, rc = O.expansion(Dq, graph, O.quantize_lambda(lam), O.quantize(h), np.zeros(n, np.int32))
        t_cpu = time.perf_counter() - t0
        out.update(cpu_unary_ms=1e3 * t_ou, cpu_expansion_ms=1e3 * t_cpu, identical=bool(np.array_equal(ref, labels)) and re == eq,
                   speedup=t_cpu / t_gpu)
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C2", "C3", "C5", "C4"]
    # --no-oracle: skip the Dinic reference (minutes at C4 size)
    if "C2" in which:
        pts, gt, models = datasets.make_homographies(seed=0)
        run("C2 homography 5k/5 planes", _lib.HOMOGRAPHY, pts, models, 3.0, 0.05, 10.0, (pts, _lib.GRAPH_KNN_IN_BALL, 200.0, 5))
    if "C3" in which:
        pts, gt, models = datasets.make_two_view_motions(seed=0)
        run("C3 two-view 1e5/8 motions", _lib.FUNDAMENTAL, pts, models, 0.75, 0.1, 14.0, (pts, _lib.GRAPH_KNN_IN_BALL, 50.0, 5))
    if "C5" in which:
        pts, gt, models = datasets.make_vanishing_points(seed=0)
        mid = 0.5 * (pts[:, :2] + pts[:, 2:])
        run("C5 vanishing points 2e5/6 VPs, k-NN(8) on midpoints", _lib.VANISHING_POINT, pts, models, 1.5, 0.1, 20.0,
            (mid, _lib.GRAPH_KNN, 0.0, 8))
    if "C4" in which:
        x1, x2, K, gt, poses = datasets.make_poses(seed=0)
        pts, f = datasets.normalize_pnp(x1, x2, K)
        raw = np.column_stack([x1, x2])
        run("C4 6D pose 1e6/10 of 16 objects", _lib.PNP, pts, poses[:10], 4.0 / f, 0.1, 6.0, (raw, _lib.GRAPH_KNN_IN_BALL, 20.0, 5))
