#!/usr/bin/env python3
"""Machine check of the filter error budgets.

score_filters.hip.h removes (hypothesis, 64-point group) pairs with a bound and single pairs with an f32 test; both rest on
hand-derived inequalities.  Here every residual type gets batches of >= 1e7 (point, hypothesis) pairs CONSTRUCTED to sit at
|r^2 / T^2 - 1| < 1e-4 - the only place where a too-optimistic error term can show - and the device re-decides every pair
with the exact FP64 residual (PGX_VERIFY=1, score_verify_kernel): a pair the chain discarded although the exact residual calls
it an inlier is a contradiction.  Cases: the coordinates' native scale (pixels / normalised), hypotheses scaled by powers of
two across the bands of pow2_normaliser (2^-252 .. 2^+252 incl. the edges 1e-75 / 1e75 where the filters switch themselves
off), thresholds from 1e-3 to 1e3 times the nominal one, and the same points with the threshold moved ONTO individual residuals.

How the near-threshold pairs are made: for a hypothesis h0 every point is moved along a random direction of its OBSERVED
coordinates by bisection (40 steps, vectorised, CPU oracle residual) until r^2(h0) = T^2 (1 + eps), eps uniform in +-1e-4; the
other hypotheses of the batch are h0 (1 + 1e-10 ... 1e-8 relative noise), which moves r^2 / T^2 by far less than 1e-4.

usage: verify_filters.py [--points 200000] [--hyps 64]    -> one JSON line per case + a summary; zero contradictions expected"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle")]
import numpy as np

os.environ["PGX_VERIFY"] = "1"
from helpers import MODEL_CASES, make_case   # noqa: E402
from pyprogressivex import _lib              # noqa: E402
import pgx_oracle as O                       # noqa: E402

OBSERVED = {"line": [0, 1], "homography": [2, 3], "homography_sym": [0, 1, 2, 3], "fundamental": [0, 1, 2, 3], "pnp": [0, 1],
            "vanishing_point": [0, 1, 2, 3]}


def near_threshold_points(mt, name, pts, h0, T2, rng, width=1e-4):
    """moves every point along a line in its OBSERVED coordinates until r^2(h0) = T2 (1 + eps), |eps| < width: points inside the
    threshold outwards along a random direction, points outside it inwards along the (numerical) gradient of r^2.  Points that
    cannot be bracketed are left alone.  Returns (points, how many were moved)."""
    n = pts.shape[0]
    cols = OBSERVED[name]
    target = T2 * (1.0 + rng.uniform(-width, width, n))
    scale = max(1.0, float(np.abs(pts[:, cols]).max()))

    def r2_at(q):
        v = O.squared_residuals(mt, q, h0)
        return np.where(np.isfinite(v), v, np.inf)
    base = r2_at(pts)
    d = rng.normal(size=(n, len(cols)))
    grad = np.zeros_like(d)
    step = 1e-6 * scale
    for k, c in enumerate(cols):
        qp, qm = pts.copy(), pts.copy()
        qp[:, c] += step
        qm[:, c] -= step
        grad[:, k] = r2_at(qp) - r2_at(qm)
    inward = base > target
    gn = np.linalg.norm(grad, axis=1)
    usable = inward & np.isfinite(gn) & (gn > 0)
    d[usable] = -grad[usable]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)

    def r2(t):
        q = pts.copy()
        q[:, cols] += t[:, None] * d
        return r2_at(q)
    lo = np.zeros(n)
    hi = np.where(inward, 0.0, 4.0 * np.sqrt(T2) + 1e-6 * scale)
    for _ in range(30):                                    # outward movers: grow the bracket until the residual exceeds the target
        top = r2(hi)
        grow = ~inward & ~(top > target)
        if not grow.any():
            break
        hi[grow] *= 2.0
    # inward movers: first step along -gradient at which the residual is below the target (a Newton step r^2 / |grad| and multiples)
    newton = np.where(usable, base / np.maximum(gn / (2.0 * step), 1e-300), 0.0)
    found = np.zeros(n, dtype=bool)
    for mult in (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 4.0, 8.0):
        t = newton * mult
        v = r2(t)
        hit = usable & ~found & (v < target)
        hi = np.where(hit, t, hi)
        lo = np.where(usable & ~found & ~hit, t, lo)       # still above the target: the bracket starts here
        found |= hit
    top = r2(hi)
    ok = np.where(inward, found & (top < target) & (r2(lo) > target), (base < target) & (top > target))
    for _ in range(70):
        mid = 0.5 * (lo + hi)
        v = r2(mid)
        towards_hi = np.where(inward, v > target, v < target)
        lo = np.where(towards_hi, mid, lo)
        hi = np.where(towards_hi, hi, mid)
    out = pts.copy()
    out[:, cols] += np.where(ok, 0.5 * (lo + hi), 0.0)[:, None] * d
    return out, int(ok.sum())


def main():
    npts = int(sys.argv[sys.argv.index("--points") + 1]) if "--points" in sys.argv else 200000
    M = int(sys.argv[sys.argv.index("--hyps") + 1]) if "--hyps" in sys.argv else 64
    rng = np.random.default_rng(2026)
    ctx = _lib.Context(0)
    total_pairs = total_bad = total_near = 0
    t0 = time.time()
    for name in MODEL_CASES:
        mt, pts, models, thr = make_case(name, npts, 4, seed=11)
        h0 = models[0].copy()
        for tscale in (1.0, 1e-3, 1e3):
            T2 = 2.25 * thr * thr * tscale * tscale
            moved, nmoved = near_threshold_points(mt, name, pts, h0, T2, rng)
            hyps = h0[None, :] * (1.0 + rng.normal(0, 1, (M, h0.shape[0])) * 10.0 ** rng.uniform(-10, -8, (M, 1)))
            hyps[0] = h0
            r = O.squared_residuals(mt, moved, h0) / T2
            near = int((np.abs(r - 1.0) < 1e-4).sum())
            exps = [0] if tscale != 1.0 else [0, -252, -249, -126, -24, 24, 126, 249, 252]   # 1e-75 ~ 2^-249.1, 1e75 ~ 2^249.1
            for e in exps:
                batch = np.ldexp(hyps, e)
                T2e = T2
                ctx.set_points(mt, moved)
                ctx.set_compound(np.zeros(len(moved)))
                a = ctx.score(batch, float(T2e), has_compound=True, exponent=2)
                ref = O.score(mt, moved, batch[:2], float(T2e), compound=np.zeros(len(moved)), has_compound=True, exponent=2)
                st = ctx.score_stats(float(T2e), has_compound=True)
                counts_ok = bool(np.array_equal(a["counts"][:2], ref["counts"]))
                rec = dict(residual=name, threshold_scale=tscale, hypothesis_scale_log2=e, points=int(len(moved)), hypotheses=M,
                           pairs=st["pairs"], near_threshold_pairs=near * M, moved_points=nmoved, path=st["path"], filter=st["filter"],
                           surviving_group_steps=st["surviving_group_steps"], exact_evaluations=st["exact_evaluations"],
                           inlier_pairs=st["inlier_pairs"], contradictions=st["contradictions"], counts_equal_oracle_on_2_hypotheses=counts_ok)
                print(json.dumps(rec), flush=True)
                total_pairs += st["pairs"]
                total_near += near * M
                total_bad += max(st["contradictions"], 0) + (0 if counts_ok else 1)
        # the threshold moved ONTO residuals of the unmoved points: T2 = r_i^2 exactly, and one ulp either side
        sq = O.squared_residuals(mt, pts, h0)
        fin = np.sort(sq[np.isfinite(sq) & (sq > 0)])
        for q in (0.1, 0.5, 0.9):
            mid = float(fin[int(q * (len(fin) - 1))])
            for T2 in (mid, float(np.nextafter(mid, np.inf)), float(np.nextafter(mid, 0))):
                ctx.set_points(mt, pts)
                ctx.set_compound(np.zeros(len(pts)))
                ctx.score(np.repeat(h0[None, :], M, axis=0), T2, has_compound=True, exponent=2)
                st = ctx.score_stats(T2, has_compound=True)
                total_pairs += st["pairs"]
                total_bad += max(st["contradictions"], 0)
                if st["contradictions"] != 0:
                    print(json.dumps(dict(residual=name, threshold_on_residual=T2, **st)), flush=True)
    ctx.close()
    print(json.dumps(dict(summary=True, verified_pairs=total_pairs, near_threshold_pairs=total_near, contradictions=total_bad,
                          seconds=round(time.time() - t0, 1))), flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
