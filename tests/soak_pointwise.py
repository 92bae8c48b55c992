"""Soak of the other device kernels against the CPU oracle (tests/test_gpu_fuzz.py runs a bounded slice; run by hand for longer):
preference vectors + compound maximum, the quantised unary table, residual sums, the minimal solvers and the inlier/outlier cut,
on random cases of every model type with the hypotheses and point sets of tests/soak_scoring.py (a hair from the truth, garbage,
rescaled by up to 10^+-160, entries of wildly different magnitude; duplicated, rescaled, huge and tiny points).
usage: python tests/soak_pointwise.py <seed> <trials>"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "progressive-x_amd"), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]
import numpy as np
from helpers import MODEL_CASES, make_case
from pyprogressivex import _lib
import pgx_oracle as O


def wild_models(rng, models, M):
    gt = models[0].copy()
    P = gt.shape[0]
    for k in range(min(M, 40)):
        kind = rng.integers(0, 5)
        if kind == 0:
            models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-13, -2), P))
        elif kind == 1:
            models[k] = rng.normal(0, 1, P) * 10.0 ** rng.uniform(-6, 4)
        elif kind == 2:
            models[k] = gt * 10.0 ** (rng.uniform(-40, 40) if rng.random() < 0.7 else rng.uniform(-160, 160))
        elif kind == 4:
            models[k] = (gt if rng.random() < 0.5 else rng.normal(0, 1, P)) * 10.0 ** rng.uniform(-rng.choice([3, 30, 120]), rng.choice([3, 30, 120]), P)
    return models


def same(a, b):
    """bitwise, NaN == NaN"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def soak(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    bad = 0
    checked = 0
    t0 = time.time()

    def report(what, *info):
        nonlocal bad
        bad += 1
        print("MISMATCH", what, *info, flush=True)

    for trial in range(trials):
        name = list(MODEL_CASES)[trial % len(MODEL_CASES)]
        n = int(rng.choice([1, 2, 63, 64, 65, 500, 3000, 4097, 20011]))
        M = int(rng.choice([1, 3, 9]))
        mt, pts, models, thr = make_case(name, n, M, seed=int(rng.integers(1 << 30)))
        pts, models = pts.copy(), wild_models(rng, models.copy(), M)
        mode = trial % 4
        if n >= 500 and mode == 1:
            pts[rng.integers(0, n, 20)] *= 10.0 ** rng.uniform(-6, 8)
            pts[10:30] = pts[10]
        elif mode == 2 and rng.random() < 0.5:
            pts *= 10.0 ** rng.uniform(-30, 30)          # the whole data set at an absurd scale
        elif mode == 3 and rng.random() < 0.3:
            pts[rng.integers(0, n, 3)] = rng.choice([np.nan, np.inf, -np.inf, 1e200, 1e-200])
        T2 = 2.25 * thr * thr * 10.0 ** (rng.uniform(-3, 3) if rng.random() < 0.8 else rng.uniform(-14, 14))
        tag = (name, n, M, mode, trial)
        ctx.set_points(mt, pts)
        comp = rng.random(n) * (rng.random(n) < 0.5)
        ctx.set_compound(comp)
        with np.errstate(all="ignore"):
            # preference + compound
            prefs = []
            for k, model in enumerate(models):
                got = ctx.preference(model, float(T2), slot=k, want_pref=True)
                ref = O.preference(mt, pts, model * (1.0 + 1e-9) if os.environ.get("SOAK_INJECT") else model, float(T2))
                checked += ref.size
                prefs.append(ref)
                if not same(got["pref"], ref):
                    report("preference", tag, "model", model[:4], int((got["pref"] != ref).sum()))
            if not same(ctx.compound_update(np.arange(M), want_compound=True), O.compound_max(np.stack(prefs))):
                report("compound", tag)
            # unary table
            lam = float(rng.choice([0.0, 0.1, 0.5, 0.9]))
            t = float(thr * 10.0 ** rng.uniform(-2, 2))
            got = ctx.pearl_unary(models, t, lam, want_table=True)
            ref = O.unary_q(mt, pts, models, t, lam)
            checked += ref.size
            if not np.array_equal(got, ref):
                w = np.argwhere(got != ref)
                report("unary", tag, "thr", t, "lam", lam, len(w), "first", w[0], got[tuple(w[0])], ref[tuple(w[0])], "model", models[min(w[0][1], M - 1)][:4])
            # residual sums
            labels = rng.integers(0, M + 1, n).astype(np.int32)
            ctx.set_labels(labels)
            for k, model in enumerate(models[:3]):
                a, b = ctx.residual_sum(model, k), O.residual_sum(mt, pts, model, labels, k)
                if not (a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-9 * max(abs(b), 1e-300)):
                    report("residual_sum", tag, a, b, "model", model[:4])
            # minimal solvers
            if name != "homography_sym":
                m = {"fundamental": 7, "homography": 4, "pnp": 3}.get(name, 2)
                S = 300
                samples = rng.integers(0, n, (S, m)).astype(np.int32)
                samples[:10, 1] = samples[:10, 0]
                got = ctx.solve_minimal(samples)
                ref = O.solve_minimal(mt, pts, samples)
                checked += ref.size
                if not same(got, ref):
                    w = np.nonzero(~((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1))[0]
                    report("solve_minimal", tag, len(w), "rows", w[:4], got[w[0]][:4], ref[w[0]][:4])
            # neighbourhood graph (finite points only) + the inlier/outlier cut of GC-RANSAC's local optimisation on it
            if np.isfinite(pts).all() and 2 <= n <= 4000:   # the oracle ranks by brute force up to 4000 points (no tie margin)
                k = int(rng.choice([1, 3, 6]))
                kind = int(rng.choice([_lib.GRAPH_KNN, _lib.GRAPH_KNN_IN_BALL]))
                span = float(np.abs(pts).max()) or 1.0
                radius = float(span * 10.0 ** rng.uniform(-3, 0.5))
                ref = O.graph_build(pts, kind, radius=radius, k=k)
                got = ctx.graph_build(pts, kind, radius=radius, k=k)
                checked += sum(a.size for a in ref)
                if not all(np.array_equal(a, b) for a, b in zip(got, ref)):
                    report("graph_build", tag, kind, radius, k)
                else:
                    lam_gc = float(rng.choice([0.05, 0.14, 0.5, 0.9]))
                    for model in models[:3]:
                        a, b = ctx.gc_labeling(model, float(T2), lam_gc), O.gc_labeling(mt, pts, model, float(T2), lam_gc, ref)
                        checked += b.size
                        if not np.array_equal(a, b):
                            report("gc_labeling", tag, lam_gc, "model", model[:4], int((a != b).sum()))
                    # greedy lambda = 0 labelling on the unary table of this case
                    Dq = O.unary_q(mt, pts, models, t, 0.0)
                    ctx.pearl_unary(models, t, 0.0)
                    h = float(rng.choice([0.0, 3.0, 40.0]))
                    rl, re, _ = O.greedy_labeling(Dq, O.quantize(h))
                    ctx.set_labels(np.full(n, M, np.int32))
                    ge = ctx.greedy_labeling(h)[0]
                    checked += n
                    if not (np.array_equal(ctx.get_labels(), rl) and ge == re):
                        report("greedy", tag, h)
    ctx.close()
    if verbose:
        print(f"pointwise soak done: seed {seed}, {trials} cases, {bad} mismatches, {checked} values compared, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    soak(int(sys.argv[1]), int(sys.argv[2]))
