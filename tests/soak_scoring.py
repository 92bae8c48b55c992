"""Scoring soak (tests/test_gpu_fuzz.py runs a bounded slice of it; run it by hand for longer): the cull + f32 filter + exact path against the CPU oracle on randomly generated
cases of all six model types - random sizes, hypotheses a hair from ground truth, garbage hypotheses at several scales,
thresholds exactly on residuals.  usage: python tests/soak_scoring.py <seed> <trials>"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "progressive-x_amd"), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]
import numpy as np
from helpers import MODEL_CASES, make_case
from pyprogressivex import _lib
import pgx_oracle as O


def soak(seed, trials, verbose=True, verify=True):
    """verify: the context is created with PGX_VERIFY=1 and every batch is also re-decided pair by pair on the device
    (pgx_score_stats[5]): a pair the group bound or the f32 filter discarded although the exact residual calls it an inlier is a
    contradiction - a hole in a filter proof - whether or not it happens to change a count."""
    rng = np.random.default_rng(seed)
    saved = os.environ.get("PGX_VERIFY")
    if verify:
        os.environ["PGX_VERIFY"] = "1"
    try:
        ctx = _lib.Context(0)
    finally:
        if saved is None:
            os.environ.pop("PGX_VERIFY", None)
        else:
            os.environ["PGX_VERIFY"] = saved
    bad = 0
    contradictions = 0
    checked_pairs = 0
    t0 = time.time()
    for trial in range(trials):
        name = list(MODEL_CASES)[trial % len(MODEL_CASES)]
        n = int(rng.choice([1, 63, 64, 65, 500, 4097, 20011, 60000]))
        M = int(rng.choice([1, 3, 64, 65, 257, 700]))
        mt, pts, models, thr = make_case(name, n, M, seed=int(rng.integers(1 << 30)))
        pts, models = pts.copy(), models.copy()
        gt = models[0].copy()
        P = gt.shape[0]
        for k in range(min(M, 40)):
            kind = rng.integers(0, 5)
            if kind == 0:
                models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-13, -2), P))
            elif kind == 1:
                models[k] = rng.normal(0, 1, P) * 10.0 ** rng.uniform(-6, 4)
            elif kind == 4:   # entries of wildly different magnitude
                models[k] = (gt if rng.random() < 0.5 else rng.normal(0, 1, P)) * 10.0 ** rng.uniform(-rng.choice([3, 30, 120]), rng.choice([3, 30, 120]), P)
            elif kind == 2:
                models[k] = gt * 10.0 ** (rng.uniform(-40, 40) if rng.random() < 0.7 else rng.uniform(-160, 160))
        if n >= 500 and trial % 3 == 0:
            pts[rng.integers(0, n, 20)] *= 10.0 ** rng.uniform(-6, 8)
            pts[10:30] = pts[10]
        sq0 = O.squared_residuals(mt, pts, gt)
        fin = np.sort(sq0[np.isfinite(sq0) & (sq0 > 0)])
        T2s = [2.25 * thr * thr * 10.0 ** (rng.uniform(-3, 3) if rng.random() < 0.8 else rng.uniform(-14, 14))]
        if len(fin):
            mid = fin[int(rng.integers(0, len(fin)))]
            T2s += [mid, np.nextafter(mid, np.inf), np.nextafter(mid, 0)]
        comp = rng.random(n) * (rng.random(n) < 0.5)
        ctx.set_points(mt, pts)
        ctx.set_compound(comp)
        for T2 in T2s:
            ref = O.score(mt, pts, models, float(T2), compound=comp, has_compound=True, exponent=2, want_masks=True)
            a = ctx.score(models, float(T2), has_compound=True, exponent=2, want_masks=True)
            b = ctx.score(models, float(T2), has_compound=True, exponent=2)
            ok = np.array_equal(a["counts"], ref["counts"]) and np.array_equal(a["masks"], ref["masks"]) and np.array_equal(b["counts"], ref["counts"])
            tol = 1e-9 * np.maximum(np.abs(ref["values"]), 1e-4)
            ok = ok and np.all(np.abs(a["values"] - ref["values"]) <= tol) and np.all(np.abs(b["values"] - ref["values"]) <= tol)
            if verify:
                st = ctx.score_stats(float(T2), has_compound=True)
                if st["contradictions"] >= 0:          # (-1: this batch went the dense / unsorted way: nothing is discarded there)
                    checked_pairs += st["pairs"]
                    if st["contradictions"] != 0:
                        contradictions += st["contradictions"]
                        ok = False
                        print("CONTRADICTION", name, n, M, T2, st, flush=True)
            if not ok:
                bad += 1
                w = np.nonzero((a["counts"] != ref["counts"]) | (b["counts"] != ref["counts"]))[0]
                vw = np.nonzero(np.abs(a["values"] - ref["values"]) > tol)[0]
                print("MISMATCH", name, n, M, T2, "count idx", w[:4], "gpu", a["counts"][w[:4]], b["counts"][w[:4]], "ref", ref["counts"][w[:4]],
                      "value idx", vw[:4], a["values"][vw[:4]], ref["values"][vw[:4]], "model", models[(w if len(w) else vw)[0]][:4] if (len(w) or len(vw)) else None, flush=True)
                if os.environ.get("SOAK_DENSE"):
                    os.environ["PGX_NO_GROUP"] = "1"
                    d = _lib.Context(0); d.set_points(mt, pts); d.set_compound(comp)
                    dd = d.score(models, float(T2), has_compound=True, exponent=2)
                    print("   dense path counts", dd["counts"][w[:4]], "values", dd["values"][vw[:4]]); d.close()
                    del os.environ["PGX_NO_GROUP"]
    ctx.close()
    if verbose:
        print(f"soak done: seed {seed}, {trials} cases, {bad} mismatches, {contradictions} contradictions in {checked_pairs:.3g} verified pairs, "
              f"{time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    soak(int(sys.argv[1]), int(sys.argv[2]))
