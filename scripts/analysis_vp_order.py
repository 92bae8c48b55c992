#!/usr/bin/env python3
"""Which ORDER of the segments makes the vanishing-point group bound (score_filters.hip.h Filter32<kVanishingPoint>::group_reject) cull
most: CPU prototype on the C5 set (2e5 segments, 6 vanishing points, 50 % outliers) with a 2048-hypothesis batch made like the
bench leg's (half two-segment samples from one vanishing point's inliers, half random pairs).  The bound is valid for ANY grouping
(its radii are those of the actual members), so only the sort key changes; this script prices candidate keys in surviving
(hypothesis, group) pairs before the kernel is touched.  Round 6 notebook: "a cull that works for vanishing points"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
from pyprogressivex import datasets  # noqa: E402


def features(seg):
    dx, dy = seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1]
    flip = (dx < 0) | ((dx == 0) & (dy < 0))
    a, b, c = (seg[:, 1] - seg[:, 3]) / 2, (seg[:, 2] - seg[:, 0]) / 2, (seg[:, 0] * seg[:, 3] - seg[:, 2] * seg[:, 1]) / 2
    s = np.where(flip, -1.0, 1.0)
    h = 0.5 * np.hypot(dx, dy)
    return np.column_stack([a * s / h, b * s / h, c * s / h, (seg[:, 0] + seg[:, 2]) / 2, (seg[:, 1] + seg[:, 3]) / 2, h, np.maximum(np.abs(seg).max(1), 1.0)])


def interleave(cols, bits):
    """bit-interleaved key, most significant bits first: a column quantised to fewer bits simply runs out earlier (the remaining
    low positions belong to the finer columns); columns are given least significant first within a round"""
    key = np.zeros(len(cols[0]), np.uint64)
    nb = max(bits)
    for level in range(nb):                  # level 0 = the top bit of every column
        for v, bk in zip(reversed(cols), reversed(bits)):
            if level < bk:
                key = (key << np.uint64(1)) | ((v.astype(np.uint64) >> np.uint64(bk - 1 - level)) & np.uint64(1))
    return key


def quant(x, bits):
    lo, hi = x.min(), x.max()
    return np.clip(((x - lo) / (hi - lo + 1e-300) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)


def survival(F, order, V, T):
    F = F[order]
    n = len(F) // 64 * 64
    G = F[:n].reshape(-1, 64, 7)
    lo, hi = G[:, :, :5].min(1), G[:, :, :5].max(1)
    c = 0.5 * (lo + hi)
    r = np.abs(G[:, :, :3] - c[:, None, :3]).max(1)
    rM = np.hypot(G[:, :, 3] - c[:, None, 3], G[:, :, 4] - c[:, None, 4]).max(1)
    hmin = G[:, :, 5].min(1)
    Pmax = G[:, :, 6].max(1)
    kept = 0
    floor = 0
    for v in V:
        Nm = G[:, :, :3] @ v
        Dm = np.hypot(G[:, :, 4] * v[2] - v[1], v[0] - G[:, :, 3] * v[2])
        floor += np.count_nonzero((np.abs(Nm) * G[:, :, 5] <= T * Dm).any(1))
        N = c[:, :3] @ v
        RN = r @ np.abs(v)
        D = np.hypot(c[:, 4] * v[2] - v[1], v[0] - c[:, 3] * v[2])
        RD = abs(v[2]) * rM
        Gq = (np.abs(N) - RN) * hmin / T - RD
        # the trust test of group_reject: the members' D must stay above the f32 / f64 error floor, D_c - R_D >= t(Pmax)
        V1, V2 = abs(v[0]) + abs(v[1]), abs(v[2])
        tt = 2.0 ** -38 * V2 / T * Pmax * Pmax + (2.0 ** -11 * V2 + 2.0 ** -38 * V1 / T) * Pmax + 2.0 ** -11 * V1 + RD
        tt0 = tt - RD
        rej = (Gq > 0) & (Gq > D) & ((D >= tt) | ((np.abs(N) - RN) >= tt0 * 1.01) if TRUST_BY_LINE else (D >= tt))
        kept += np.count_nonzero(~rej)
    return kept / (len(V) * len(G)), floor / (len(V) * len(G))


TRUST_BY_LINE = "--trust-by-line" in sys.argv     # D_i >= |N^_i|: the line distance bounds the midpoint distance from below

if __name__ == "__main__":
    seg, gt, vps = datasets.make_vanishing_points(seed=0)
    F = features(seg)
    rng = np.random.default_rng(7)
    L = np.cross(np.column_stack([seg[:, :2], np.ones(len(seg))]), np.column_stack([seg[:, 2:], np.ones(len(seg))]))
    V = []
    for m in range(1024):
        if m % 2 == 0:
            idx = np.flatnonzero(gt == 1 + (m // 2) % 6)
            i, j = rng.choice(idx, 2, replace=False)
        else:
            i, j = rng.choice(len(seg), 2, replace=False)
        v = np.cross(L[i], L[j])
        V.append(v / np.abs(v).max())
    V = np.array(V)
    T = 1.5 * (1 + 1 / 64)
    th = np.arctan2(-F[:, 0], F[:, 1])
    rho = F[:, 2]
    mx, my = F[:, 3], F[:, 4]
    along = mx * F[:, 1] - my * F[:, 0]          # position of the midpoint ALONG its line
    cands = {"shipped: Morton(mx, my, theta), 10 bits each": interleave([quant(th, 10), quant(my, 10), quant(mx, 10)], [10, 10, 10])}
    for bt, bp in ((10, 8), (10, 7), (10, 6), (10, 5), (10, 4)):
        cands[f"Morton(mx, my, theta) with {bp} position bits under {bt} of theta"] = interleave([quant(mx, bp), quant(my, bp), quant(th, bt)], [bp, bp, bt])
    cands["theta only"] = quant(th, 20).astype(np.uint64)
    for bt, br in ((10, 10), (10, 8), (10, 6)):
        cands[f"Hough: Morton(theta {bt} bits, rho {br} bits)"] = interleave([quant(rho, br), quant(th, bt)], [br, bt])
    for bt, br, ba in ((10, 9, 5), (10, 8, 6), (10, 7, 7), (9, 9, 9)):
        cands[f"Morton(theta {bt}, rho {br}, along {ba})"] = interleave([quant(along, ba), quant(rho, br), quant(th, bt)], [ba, br, bt])
    lh = np.log(F[:, 5])
    cx, cy = 0.5 * (mx.min() + mx.max()), 0.5 * (my.min() + my.max())
    rc = F[:, 0] * cx + F[:, 1] * cy + F[:, 2]
    for bt, br, ba, bl in ((10, 10, 0, 3), (10, 10, 2, 3), (10, 10, 3, 3), (10, 10, 4, 3), (10, 9, 5, 3), (9, 9, 6, 3), (8, 8, 8, 3), (10, 10, 10, 3)):
        cands[f"Morton(theta {bt}, rho centred {br}, along {ba}, log length {bl})"] = interleave([quant(lh, bl), quant(along, max(ba, 1)) if ba else np.zeros(len(lh), np.int64), quant(rc, br), quant(th, bt)], [bl, ba, br, bt])
    for bt, br, bh in ((10, 10, 3), (10, 10, 5), (9, 9, 9)):
        cands[f"Morton(theta {bt}, rho {br}, log length {bh})"] = interleave([quant(lh, bh), quant(rho, br), quant(th, bt)], [bh, br, bt])
    cands["Morton(theta 10, rho 10, log length 4, along 4)"] = interleave([quant(along, 4), quant(lh, 4), quant(rho, 10), quant(th, 10)], [4, 4, 10, 10])
    for name, key in cands.items():
        order = np.argsort(key, kind="stable")
        sv, fl = survival(F, order, V, T)
        print(f"{100 * sv:6.2f} % of the (hypothesis, group) pairs survive ({100 * fl:5.2f} % hold an inlier)   {name}", flush=True)
