#!/usr/bin/env python3
"""Dependent-step census of the level-synchronous min-cut WITHOUT a GPU: the sequential emulation (tests/emu, the product's own
per-site bodies and driver) on the C3 problem and on a quarter-size C4 (2.5e5 correspondences, 16 objects, ball radius scaled by
4^(1/5)), one line per move: sweeps, global relabels, BFS levels.  The emulation reproduces the device's counts closely (C3: 46
relabels / 2 222 levels / 960 sweeps against the GPU's 42 / 2 145 / 872), so a schedule idea can be priced in launches before any
kernel is written.  MF_EMU_TRACE=1 adds, per round, the excess that still reaches t, the stranded excess and how deep the sites
with excess sit.  Knobs: MF_EMU_SWEEPS_LIST, MF_EMU_STALL, MF_EMU_LIST_DIV.  Round 6 notebook: "hard moves, once more".

usage: exp_emu_hard_moves.py C3|C4s [first-cycle move to trace alone]"""
import ctypes as C
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle")]
import pgx_oracle as O  # noqa: E402
from pyprogressivex import datasets  # noqa: E402

SO = os.path.join(ROOT, "tests", "emu", "libmf_emu.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, os.path.join(ROOT, "tests", "emu", "mf_emu.cpp")])
emu = C.CDLL(SO)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def emu_expand(Dq, graph, lq, hq, alpha, labels, seed=1):
    n, L = Dq.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32).copy()
    off, idx, mult = graph
    ch = C.c_int64()
    st = np.zeros(8, np.int64)
    r = emu.emu_expand_alpha(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), _p(off, C.c_int32), _p(idx, C.c_int32), _p(mult, C.c_int32),
                             C.c_int64(lq), C.c_int64(hq), C.c_int(alpha), _p(lab, C.c_int32), C.c_uint64(seed), C.c_int(0), C.byref(ch),
                             _p(st, C.c_int64))
    assert r == 0, r
    return lab, ch.value, st


def problem(which):
    if which == "C3":
        pts, gt, models = datasets.make_two_view_motions(seed=0)
        mt, thr, lam, h = O.FUNDAMENTAL, 0.75, 0.1, 14.0
        graph = O.graph_build(pts, 0, radius=50.0, k=5)
    else:
        x1, x2, K, gt, poses = datasets.make_poses(n_per_object=12500, n_objects=16, n_outliers=50000, seed=0)
        pts, f = datasets.normalize_pnp(x1, x2, K)
        models = poses[:10]
        mt, thr, lam, h = O.PNP, 4.0 / f, 0.1, 6.0
        graph = O.graph_build(np.column_stack([x1, x2]), 0, radius=20.0 * 4 ** 0.2, k=5)
    return O.unary_q(mt, pts, models, thr, lam), tuple(np.ascontiguousarray(g, dtype=np.int32) for g in graph), lam, h


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "C4s"
    only = int(sys.argv[2]) if len(sys.argv) > 2 else None
    Dq, graph, lam, h = problem(which)
    n, L = Dq.shape
    lq, hq = O.quantize_lambda(lam), O.quantize(h)
    lab = np.zeros(n, np.int32)
    trace = os.environ.pop("MF_EMU_TRACE", None)
    tot = np.zeros(8, np.int64)
    for cyc in range(10):
        moved = 0
        for alpha in range(L):
            if trace and (only is None or (cyc == 0 and alpha == only)):
                os.environ["MF_EMU_TRACE"] = trace
            t0 = time.time()
            lab, ch, st = emu_expand(Dq, graph, lq, hq, alpha, lab)
            os.environ.pop("MF_EMU_TRACE", None)
            tot += st
            moved += ch
            print(f"cycle {cyc} alpha {alpha}: relabelled {ch}, sweeps {st[1]} (list {st[6]}), global relabels {st[2]}, BFS levels {st[3]} "
                  f"({time.time() - t0:.1f} s)", flush=True)
            if only is not None and cyc == 0 and alpha == only:
                sys.exit(0)
        if moved == 0:
            break
    print(f"total: {tot[0]} moves, {tot[1]} sweeps, {tot[2]} global relabels, {tot[3]} BFS levels = {tot[1] + tot[3]} dependent steps; "
          f"labels crc {zlib.crc32(lab.tobytes())}")
