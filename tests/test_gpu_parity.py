"""GPU parity: the HIP hot path (through the C ABI of libpgx.so) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer work — inlier counts/masks, quantised unary costs, labels, energies, bucket
orders — bit-exact; floating-point sums (scores, Tanimoto terms, residual sums) within 1e-5 relative (asserted at
1e-9, only the summation order differs).  Residual-derived values that involve no reduction (preference vectors,
compound max) are required to be bit-identical: the kernels use the oracle's operation order with FMA contraction off.
"""
import numpy as np
import pytest

from helpers import (MODEL_CASES, csr_from_pairs, make_case, random_sym_graph, realistic_labeling_problem)
from pyprogressivex import _lib, datasets

pytestmark = pytest.mark.gpu

REL = 1e-9


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(1e-300, np.maximum(np.abs(a), np.abs(b)))) if a.size else 0.0


# ----------------------------------------------------------------------------------------------------------------------
# a2 / a3 / a4 : preference vector (bit-exact), Tanimoto terms, compound max
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20011])
def test_preference_bit_exact(gpu_ctx, oracle, name, n):
    mt, pts, models, thr = make_case(name, n, 3, seed=n)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    comp = np.random.default_rng(7).uniform(0, 1, n)
    gpu_ctx.set_compound(comp)
    for k, model in enumerate(models):
        got = gpu_ctx.preference(model, T2, slot=k, want_pref=True)
        ref = oracle.preference(mt, pts, model, T2)
        assert np.array_equal(got["pref"], ref), f"{name}: preference vector differs from the oracle"
        d, a, b = oracle.tanimoto_terms(ref, comp)
        assert abs(got["dot"] - d) <= REL * max(abs(d), 1e-300)
        assert abs(got["pref_sqnorm"] - a) <= REL * max(abs(a), 1e-300)
        assert abs(got["comp_sqnorm"] - b) <= REL * max(abs(b), 1e-300)
        assert np.array_equal(gpu_ctx.get_preference(k), ref)
    prefs = np.stack([oracle.preference(mt, pts, m, T2) for m in models])
    got_c = gpu_ctx.compound_update(np.arange(len(models)), want_compound=True)
    assert np.array_equal(got_c, oracle.compound_max(prefs))
    # K == 0 leaves the compound vector untouched (progressive_x.h:600-601)
    assert np.array_equal(gpu_ctx.compound_update(np.zeros(0, np.int32), want_compound=True), got_c)


def test_preference_huge_threshold_exercises_every_lane(gpu_ctx, oracle):
    # with T2 huge every pref is in (0,1): bit-equality of pref == bit-equality of every squared residual
    for name in MODEL_CASES:
        mt, pts, models, thr = make_case(name, 5000, 2, seed=3)
        gpu_ctx.set_points(mt, pts)
        for model in models:
            sq = oracle.squared_residuals(mt, pts, model)
            T2 = float(np.nanmax(sq[np.isfinite(sq)]) * 4 + 1)
            got = gpu_ctx.preference(model, T2, slot=0, want_pref=True)["pref"]
            assert np.array_equal(got, oracle.preference(mt, pts, model, T2)), name


# ----------------------------------------------------------------------------------------------------------------------
# a1 : batched scoring
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("n,M", [(1, 1), (64, 3), (65, 257), (4097, 64), (20000, 300)])
def test_score_matches_oracle(gpu_ctx, oracle, name, n, M):
    mt, pts, models, thr = make_case(name, n, M, seed=n + M)
    T2 = 2.25 * thr * thr
    comp = np.random.default_rng(11).uniform(0, 1, n) * (np.random.default_rng(12).uniform(0, 1, n) < 0.5)
    gpu_ctx.set_points(mt, pts)
    for has_compound, exponent in ((False, 2), (True, 2), (True, 3), (True, 1)):
        gpu_ctx.set_compound(comp if has_compound else None)
        got = gpu_ctx.score(models, T2, has_compound=has_compound, exponent=exponent, want_masks=True)
        ref = oracle.score(mt, pts, models, T2, compound=comp, has_compound=has_compound, exponent=exponent,
                           want_masks=True)
        assert np.array_equal(got["counts"], ref["counts"]), f"{name}: inlier counts differ"
        assert np.array_equal(got["masks"], ref["masks"]), f"{name}: inlier masks differ"
        assert _rel(got["values"], ref["values"]) <= REL
        assert _rel(got["shared"], ref["shared"]) <= REL
        scale = np.maximum(np.abs(ref["values"]), np.abs(ref["shared"]) ** exponent) + 1e-300
        assert np.max(np.abs(got["scores"] - ref["scores"]) / scale) <= 1e-9
        # counts == popcount of the mask rows (checksum of the mask against the counter)
        pop = np.array([sum(bin(int(w)).count("1") for w in row) for row in got["masks"]])
        assert np.array_equal(pop, got["counts"])
        # same launch without masks gives identical numbers (different template instance)
        again = gpu_ctx.score(models, T2, has_compound=has_compound, exponent=exponent, want_masks=False)
        assert np.array_equal(again["counts"], got["counts"]) and np.array_equal(again["values"], got["values"])


def test_score_threshold_boundary_is_strict(gpu_ctx, oracle):
    # line x = 0, points at |x| = 2 exactly: r^2 == T2 = 4 -> NOT an inlier for the scorer (strict <, :85) ...
    mt = 0
    pts = np.array([[2.0, 5.0], [-2.0, 1.0], [1.999999, 0.0], [0.0, 0.0], [2.0000001, 3.0]])
    model = np.array([[1.0, 0.0, 0.0]])
    gpu_ctx.set_points(mt, pts)
    got = gpu_ctx.score(model, 4.0, want_masks=True)
    ref = oracle.score(mt, pts, model, 4.0, want_masks=True)
    assert got["counts"][0] == ref["counts"][0] == 2
    assert np.array_equal(got["masks"], ref["masks"])
    # ... but IS within the threshold for PEARL's data term (r^2 > T2 test, PEARL.h:123): thr with 9/4 thr^2 == 4
    thr = 4.0 / 3.0
    Dq = gpu_ctx.pearl_unary(model, thr, 0.25, want_table=True)
    assert np.array_equal(Dq, oracle.unary_q(mt, pts, model, thr, 0.25))


@pytest.mark.parametrize("name", ["pnp", "homography"])
def test_score_filter_adversarial(gpu_ctx, oracle, name):
    """The rejection filter (score.hip Filter<>) must never change a result: thresholds placed exactly ON residuals,
    hypotheses a hair away from ground truth, vanishing denominators, NaN/Inf inputs, and magnitudes that must trip the
    host guard all have to reproduce the oracle's counts and masks bit for bit."""
    mt, pts, models, thr = make_case(name, 4096, 64, seed=77)
    rng = np.random.default_rng(5)
    pts = pts.copy()
    models = models.copy()
    gt = models[0].copy()
    # hypotheses at relative distance 1e-12 .. 1e-3 from a ground-truth model: residuals straddle the threshold
    for k in range(8, 40):
        models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-12, -3), gt.shape))
    # denominators that cancel to ~0 for the ground-truth model (pz = 0 / t3 = 0), and exact zeros
    if name == "pnp":
        P = gt.reshape(3, 4)
        for i in range(0, 200):
            X, Y = pts[i, 2], pts[i, 3]
            pts[i, 4] = -(P[2, 0] * X + P[2, 1] * Y + P[2, 3]) / P[2, 2] * (1.0 + (i % 5) * 1e-16)
    else:
        H = gt.reshape(3, 3)
        for i in range(0, 200):
            pts[i, 1] = -(H[2, 0] * pts[i, 0] + H[2, 2]) / H[2, 1] * (1.0 + (i % 5) * 1e-16) if H[2, 1] != 0 else 0.0
    pts[200:210] = 0.0
    pts[210, 0] = np.nan
    pts[211, -1] = np.inf
    pts[212, 0] = -np.inf
    models[40] = np.nan
    models[41] = 0.0
    models[42, 0] = np.inf
    models[43] = gt * 1e150
    models[44] = gt * 1e-150
    models[45] = gt * 1e150          # entries that overflow f32 with mixed signs: inf - inf inside the f32 bound tests
    models[45, ::2] *= -1.0
    models[46] = -gt * 1e150
    # scales that ARE representable in f32 but whose squares are not (found by tests/soak_scoring.py: pz^2 underflowed to 0 before
    # the multiplication by T2 and "lhs > 0" rejected true inliers; the f32 copies are now made from a power-of-two-normalised model)
    for k, sc in enumerate((1e-24, 1e-30, 1e-37, 1e22, 1e30, 1e-110, 1e110)):
        models[47 + k] = gt * sc
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.set_compound(None)
    sq0 = oracle.squared_residuals(mt, pts, gt)
    finite = np.sort(sq0[np.isfinite(sq0) & (sq0 > 0)])
    T2s = [2.25 * thr * thr, finite[len(finite) // 3], np.nextafter(finite[len(finite) // 3], np.inf),
           np.nextafter(finite[len(finite) // 3], 0), finite[0], finite[-1] * 4, 1e-30, 1e30]
    for T2 in T2s:
        got = gpu_ctx.score(models, float(T2), want_masks=True)
        ref = oracle.score(mt, pts, models, float(T2), want_masks=True)
        assert np.array_equal(got["counts"], ref["counts"]), (name, T2)
        assert np.array_equal(got["masks"], ref["masks"]), (name, T2)
        assert _rel(got["values"], ref["values"]) <= REL
    # huge observed coordinates: Umax / T exceeds the guard => the library must fall back to the unfiltered instance
    big = pts.copy()
    big[300:310, 0 if name == "pnp" else 2] = 1e13
    gpu_ctx.set_points(mt, big)
    got = gpu_ctx.score(models, 2.25 * thr * thr, want_masks=True)
    ref = oracle.score(mt, big, models, 2.25 * thr * thr, want_masks=True)
    assert np.array_equal(got["counts"], ref["counts"]) and np.array_equal(got["masks"], ref["masks"])


def test_vanishing_point_filter_and_cull_adversarial(oracle, monkeypatch):
    """Filter32<kVanishingPoint> + the group test on oriented, length-normalised segment features must never change a
    result.  Stress: thresholds exactly ON residuals, vanishing points a hair from ground truth, AT a segment's midpoint
    (D = 0) and at infinity (v2 = 0), zero-length / reversed / duplicated / huge / tiny segments, NaN and Inf, garbage
    hypotheses, thresholds from 1e-30 to 1e30 - all against the oracle bit for bit, with and without masks, and against
    the dense kernel (PGX_NO_GROUP=1)."""
    rng = np.random.default_rng(11)
    monkeypatch.setenv("PGX_NO_GROUP", "1")
    plain = _lib.Context(0)
    monkeypatch.delenv("PGX_NO_GROUP")
    culled = _lib.Context(0)
    try:
        for trial in range(10):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30011]))
            mt, pts, models, thr = make_case("vanishing_point", n, 64, seed=500 + trial)
            pts, models = pts.copy(), models.copy()
            gt = models[0].copy()
            for k in range(8, 30):                    # a hair away from ground truth
                models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-12, -3), 3))
            if n >= 1000:
                pts[5, 2:] = pts[5, :2]               # zero-length segment
                pts[6] = pts[7][[2, 3, 0, 1]]         # the reverse of its neighbour
                pts[8:40] = pts[8]                    # duplicates
                pts[40:50] *= 1e6                     # huge coordinates
                pts[50:60] *= 1e-6                    # tiny segments near the origin
                m = 0.5 * (pts[60, :2] + pts[60, 2:])
                models[30] = np.array([m[0], m[1], 1.0])              # the vanishing point AT a midpoint: D = 0 there
                models[31] = np.array([m[0], m[1], 1.0]) * (1 + 1e-15)
            models[32] = np.array([1.0, 0.0, 0.0])    # at infinity
            models[33] = np.array([0.0, 1.0, 0.0])
            models[34] = np.array([3.0, -2.0, 1e-300])
            models[35] = np.nan
            models[36] = 0.0
            models[37] = np.array([np.inf, 1.0, 1.0])
            models[38] = gt * 1e150
            models[39] = gt * 1e-150
            models[40:] = rng.normal(0, 1, (models.shape[0] - 40, 3)) * rng.choice([1e-3, 1.0, 1e3], (models.shape[0] - 40, 1))
            for k, sc in enumerate((1e-24, 1e-30, 1e-37, 1e22, 1e30)):     # f32-representable, squares are not (soak_scoring.py)
                models[40 + k] = gt * sc
            sq0 = oracle.squared_residuals(mt, pts, gt)
            finite = np.sort(sq0[np.isfinite(sq0) & (sq0 > 0)])
            T2s = [2.25 * thr * thr, 1e-30, 1e30]
            if len(finite):
                mid = finite[len(finite) // 3]
                T2s += [mid, np.nextafter(mid, np.inf), np.nextafter(mid, 0), finite[0], finite[-1] * 4]
            for ctx in (plain, culled):
                ctx.set_points(mt, pts)
            if n >= 64:
                assert culled.score_debug_fetch("order").shape == (n,)     # the sorted path is active
            for T2 in T2s:
                ref = oracle.score(mt, pts, models, float(T2), want_masks=True)
                b = culled.score(models, float(T2), want_masks=True)
                assert np.array_equal(b["counts"], ref["counts"]), (trial, T2)
                assert np.array_equal(b["masks"], ref["masks"]), (trial, T2)
                # sums are 2^-q fixed point (q = 50 for small n): absolute 1e-13, relative 1e-9 on sums of order >= 1
                assert np.all(np.abs(b["values"] - ref["values"]) <= REL * np.maximum(np.abs(ref["values"]), 1e-4))
                c = culled.score(models, float(T2))
                assert np.array_equal(c["counts"], ref["counts"]), (trial, T2, "queued path")
                assert np.array_equal(c["values"], b["values"])
                a = plain.score(models, float(T2))
                assert np.array_equal(a["counts"], ref["counts"])
        # non-finite data: no sorted copies, the dense kernel answers
        mt, pts, models, thr = make_case("vanishing_point", 3000, 16, seed=3)
        pts = pts.copy()
        pts[17, 1] = np.nan
        culled.set_points(mt, pts)
        with pytest.raises(_lib.PgxError, match="no sorted copies"):
            culled.score_debug_fetch("order")
        got = culled.score(models, 2.25 * thr * thr)
        assert np.array_equal(got["counts"], oracle.score(mt, pts, models, 2.25 * thr * thr)["counts"])
    finally:
        plain.close()
        culled.close()


@pytest.mark.parametrize("name", ["line", "homography_sym"])
def test_line_and_symmetric_filters_adversarial(oracle, name, monkeypatch):
    """Filter32<kLine2D> (three f32 FMAs + the group bound on 2-D boxes) and the symmetric transfer error on the forward
    homography filters: thresholds exactly ON residuals, hypotheses a hair from ground truth, zero / NaN / Inf / 1e+-150 /
    f32-overflowing hypotheses, huge / tiny / duplicated points, vanishing denominators of BOTH directions (symmetric),
    thresholds 1e-30 .. 1e30 - against the oracle bit for bit, with and without masks, and against the dense kernel."""
    rng = np.random.default_rng(17)
    monkeypatch.setenv("PGX_NO_GROUP", "1")
    plain = _lib.Context(0)
    monkeypatch.delenv("PGX_NO_GROUP")
    culled = _lib.Context(0)
    try:
        for trial in range(8):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30011]))
            mt, pts, models, thr = make_case(name, n, 64, seed=900 + trial)
            pts, models = pts.copy(), models.copy()
            gt = models[0].copy()
            P = gt.shape[0]
            for k in range(8, 28):
                models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-12, -3), P))
            if n >= 1000:
                pts[8:40] = pts[8]
                pts[40:50] *= 1e6
                pts[50:60] *= 1e-6
                pts[60:64, 0] *= 1e10
                if trial % 2:
                    pts[64:70] *= 1e30
                if name == "homography_sym":      # t3 = 0 for the forward map, s3 = 0 for the inverse
                    H, Hi = gt[:9].reshape(3, 3), gt[9:].reshape(3, 3)
                    for i in range(100, 140):
                        if H[2, 1] != 0:
                            pts[i, 1] = -(H[2, 0] * pts[i, 0] + H[2, 2]) / H[2, 1] * (1.0 + (i % 5) * 1e-16)
                    for i in range(140, 180):
                        if Hi[2, 1] != 0:
                            pts[i, 3] = -(Hi[2, 0] * pts[i, 2] + Hi[2, 2]) / Hi[2, 1] * (1.0 + (i % 5) * 1e-16)
            models[30] = 0.0
            models[31] = np.nan
            models[32] = gt
            models[32, P - 1] = np.nan            # NaN in the last entry only (the inverse part of a symmetric model)
            models[33] = gt
            models[33, 2] = np.inf
            models[34] = gt * 1e150
            models[35] = gt * 1e-150
            models[36] = gt * 1e150
            models[36, ::2] *= -1.0
            models[37] = gt * 1e33
            models[38] = gt * 1e-43
            models[40:] = rng.normal(0, 1, (models.shape[0] - 40, P)) * rng.choice([1e-3, 1.0, 1e3], (models.shape[0] - 40, 1))
            for k, sc in enumerate((1e-24, 1e-30, 1e22, 1e30)):          # f32-representable, squares are not (soak_scoring.py)
                models[40 + k] = gt * sc
            sq0 = oracle.squared_residuals(mt, pts, gt)
            finite = np.sort(sq0[np.isfinite(sq0) & (sq0 > 0)])
            T2s = [2.25 * thr * thr, 1e-30, 1e30, 1e-23, 1e23]
            if len(finite):
                mid = finite[len(finite) // 3]
                T2s += [mid, np.nextafter(mid, np.inf), np.nextafter(mid, 0), finite[0], finite[-1] * 4]
            for ctx in (plain, culled):
                ctx.set_points(mt, pts)
            for T2 in T2s:
                ref = oracle.score(mt, pts, models, float(T2), want_masks=True)
                b = culled.score(models, float(T2), want_masks=True)
                assert np.array_equal(b["counts"], ref["counts"]), (trial, T2)
                assert np.array_equal(b["masks"], ref["masks"]), (trial, T2)
                assert np.all(np.abs(b["values"] - ref["values"]) <= REL * np.maximum(np.abs(ref["values"]), 1e-4))
                c = culled.score(models, float(T2))
                assert np.array_equal(c["counts"], ref["counts"]), (trial, T2, "queued path")
                a = plain.score(models, float(T2))
                assert np.array_equal(a["counts"], ref["counts"])
        mt, pts, models, thr = make_case(name, 5000, 16, seed=3)
        culled.set_points(mt, pts)
        culled.score_upload(models)
        st = culled.score_stats(2.25 * thr * thr)
        assert st["path"] == "cull + group-major" and st["filter"] == "f32" and st["exact_evaluations"] < st["pairs"] // 2
    finally:
        plain.close()
        culled.close()


def test_fundamental_filter_and_cull_adversarial(oracle, monkeypatch):
    """Filter32<kFundamental> (division-free f32 Sampson test) + the bilinear group bound on the 4-D boxes must never change
    a result.  Stress: thresholds exactly ON residuals, matrices a hair from ground truth, correspondences AT the two
    epipoles (zero gradient: 0 / 0), matrices whose gradient vanishes everywhere (only f8) or that are rank 1 / zero /
    NaN / Inf / 1e+-150 / f32-overflowing with mixed signs, duplicated / huge / tiny coordinates, thresholds from 1e-30 to
    1e30 (outside [1e-12, 1e12] the library must fall back to the dense kernel) - all against the oracle bit for bit, with
    and without masks, and against the dense kernel (PGX_NO_GROUP=1)."""
    rng = np.random.default_rng(13)
    monkeypatch.setenv("PGX_NO_GROUP", "1")
    plain = _lib.Context(0)
    monkeypatch.delenv("PGX_NO_GROUP")
    culled = _lib.Context(0)
    try:
        for trial in range(10):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30011]))
            mt, pts, models, thr = make_case("fundamental", n, 64, seed=700 + trial)
            pts, models = pts.copy(), models.copy()
            gt = models[0].copy()
            for k in range(8, 28):                    # a hair away from ground truth
                models[k] = gt * (1.0 + rng.normal(0, 10.0 ** rng.uniform(-12, -3), 9))
            F = gt.reshape(3, 3)
            ea = np.linalg.svd(F)[2][-1]              # F ea = 0: (rx, ry) vanish at ea in the first image
            eb = np.linalg.svd(F.T)[2][-1]            # F^T eb = 0: (rxc, ryc, rwc) vanish at eb in the second image
            if n >= 1000:
                if abs(ea[2]) > 1e-12 and abs(eb[2]) > 1e-12:
                    pts[5] = np.concatenate([ea[:2] / ea[2], eb[:2] / eb[2]])     # zero gradient: r^2 = 0 / 0
                    pts[6, :2] = ea[:2] / ea[2]
                    pts[7, 2:] = eb[:2] / eb[2]
                pts[8:40] = pts[8]                    # duplicates
                pts[40:50] *= 1e6                     # huge coordinates
                pts[50:60] *= 1e-6
                pts[60:64, 0] *= 1e10                 # products that overflow f32 once squared
                if trial % 2:
                    pts[64:70] *= 1e13                # ... and P^2 Pmax^2 scales that must switch the filter off per hypothesis
            models[28] = 0.0
            models[28, 8] = 1.0                       # n = 1 everywhere, zero gradient: r^2 = inf
            models[29] = np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0]).reshape(-1)   # rank 1
            models[30] = 0.0
            models[31] = np.nan
            models[32] = gt
            models[32, 4] = np.nan
            models[33] = gt
            models[33, 2] = np.inf
            models[34] = gt * 1e150
            models[35] = gt * 1e-150
            models[36] = gt * 1e150
            models[36, ::2] *= -1.0                   # entries that overflow f32 with mixed signs
            models[37] = gt * 1e33                    # beyond the per-hypothesis overflow guard, finite in f32
            models[38] = gt * 1e-33
            models[39] = gt * 1e-43                   # denormal in f32
            models[40:] = rng.normal(0, 1, (models.shape[0] - 40, 9)) * rng.choice([1e-6, 1e-3, 1.0], (models.shape[0] - 40, 9))
            for k, sc in enumerate((1e-24, 1e-30, 1e22, 1e30)):          # f32-representable, squares are not (soak_scoring.py)
                models[40 + k] = gt * sc
            sq0 = oracle.squared_residuals(mt, pts, gt)
            finite = np.sort(sq0[np.isfinite(sq0) & (sq0 > 0)])
            T2s = [2.25 * thr * thr, 1e-30, 1e30, 1e-11, 1e11]
            if len(finite):
                mid = finite[len(finite) // 3]
                T2s += [mid, np.nextafter(mid, np.inf), np.nextafter(mid, 0), finite[0], finite[-1] * 4]
            for ctx in (plain, culled):
                ctx.set_points(mt, pts)
            if n >= 64:
                assert culled.score_debug_fetch("order").shape == (n,)     # the sorted path is active
            for T2 in T2s:
                ref = oracle.score(mt, pts, models, float(T2), want_masks=True)
                b = culled.score(models, float(T2), want_masks=True)
                assert np.array_equal(b["counts"], ref["counts"]), (trial, T2)
                assert np.array_equal(b["masks"], ref["masks"]), (trial, T2)
                assert np.all(np.abs(b["values"] - ref["values"]) <= REL * np.maximum(np.abs(ref["values"]), 1e-4))
                c = culled.score(models, float(T2))
                assert np.array_equal(c["counts"], ref["counts"]), (trial, T2, "queued path")
                if 1e-12 < T2 < 1e12:
                    assert np.array_equal(c["values"], b["values"])
                a = plain.score(models, float(T2))
                assert np.array_equal(a["counts"], ref["counts"])
        # the filter is in use at ordinary thresholds and not outside its range
        mt, pts, models, thr = make_case("fundamental", 5000, 16, seed=3)
        culled.set_points(mt, pts)
        culled.score_upload(models)
        st = culled.score_stats(2.25 * thr * thr)
        assert st["path"] == "cull + group-major" and st["filter"] == "f32" and st["exact_evaluations"] < st["pairs"] // 4
        assert culled.score_stats(1e-20)["path"] == "every pair"
        # non-finite data: no sorted copies, the dense kernel answers
        pts = pts.copy()
        pts[17, 1] = np.nan
        culled.set_points(mt, pts)
        with pytest.raises(_lib.PgxError, match="no sorted copies"):
            culled.score_debug_fetch("order")
        got = culled.score(models, 2.25 * thr * thr)
        assert np.array_equal(got["counts"], oracle.score(mt, pts, models, 2.25 * thr * thr)["counts"])
    finally:
        plain.close()
        culled.close()


@pytest.mark.parametrize("name", ["pnp", "homography", "fundamental", "homography_sym", "line"])
def test_group_culling_never_changes_a_count(oracle, name, monkeypatch):
    # The group-major path (Morton-sorted points, bound test per 64-point group, fixed-point accumulation) against the
    # plain chunked kernel on data built to stress the bound: wide magnitude ranges, duplicated points, groups of one,
    # garbage hypotheses (points behind the camera, vanishing denominators), tiny and huge thresholds.
    monkeypatch.setenv("PGX_NO_GROUP", "1")
    plain = _lib.Context(0)
    monkeypatch.delenv("PGX_NO_GROUP")
    culled = _lib.Context(0)
    rng = np.random.default_rng(77)
    try:
        for trial in range(12):
            n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 30011]))
            mt, pts, models, thr = make_case(name, n, 96, seed=1000 + trial)
            if trial % 3 == 1:                       # clustered duplicates: zero-radius groups
                pts = pts[rng.integers(0, max(1, n // 50), n)]
            if trial % 4 == 2:                       # wide magnitude range in the multiplied coordinates
                pts = pts.copy()
                pts[:, -3 if name == "pnp" else 0] *= rng.choice([1e-3, 1.0, 1e3], n)
            garbage = rng.normal(0, 1, models.shape) * rng.choice([1e-3, 1.0, 1e2], (models.shape[0], 1))
            hyps = np.vstack([models, garbage, models * (1 + 1e-9 * rng.normal(0, 1, models.shape))])
            T2 = (9.0 / 4.0 * thr * thr) * float(rng.choice([1e-6, 1.0, 1.0, 25.0, 1e6]))
            for ctx in (plain, culled):
                ctx.set_points(mt, pts)
            a = plain.score(hyps, T2, want_masks=True)
            b = culled.score(hyps, T2, want_masks=True)
            assert np.array_equal(a["counts"], b["counts"]), f"trial {trial}: counts differ"
            assert np.array_equal(a["masks"], b["masks"]), f"trial {trial}: masks differ"
            assert _rel(a["values"], b["values"]) < REL
            ref = oracle.score(mt, pts, hyps, T2)
            assert np.array_equal(b["counts"], ref["counts"])
            # without masks the candidates go through the compaction queue (64 exact evaluations per wave step): the
            # same per-pair fixed-point integers are accumulated, so counts AND sums are bitwise those of the mask variant
            c = culled.score(hyps, T2)
            assert np.array_equal(c["counts"], b["counts"]), f"trial {trial}: queued path counts differ"
            if name in ("pnp", "homography"):
                assert np.array_equal(c["values"], b["values"]) and np.array_equal(c["shared"], b["shared"])
    finally:
        plain.close()
        culled.close()


@pytest.mark.parametrize("name", ["pnp", "homography", "fundamental", "homography_sym", "line"])
def test_group_path_is_independent_of_the_launch_geometry(name, monkeypatch, oracle):
    # per-pair fixed point + integer accumulation: the results of the group-major path do not depend on how many waves share
    # a group, on queue boundaries, on the order in which groups finish, on whether a step is evaluated in place or through the
    # wave's candidate queue, on how many accumulator replicas there are or on the order of the batch: bitwise equal
    mt, pts, models, thr = make_case(name, 50000, 700, seed=21)
    T2 = 2.25 * thr * thr
    comp = np.random.default_rng(3).random(len(pts))
    configs = [dict(split=s) for s in (1, 3, 8, 16)] + [
        dict(group_xcd=1), dict(group_xcd=0), dict(group_xcd=1, split=2), dict(nrep=64), dict(dense_min=65), dict(dense_min=1),
        dict(dense_min=8, split=5), dict(cull_segs=31), dict(no_sort=1)]
    outs = []
    for cfg in configs:
        cfg = dict(cfg)
        monkeypatch.delenv("PGX_NO_SORT", raising=False)
        if cfg.pop("no_sort", 0):
            monkeypatch.setenv("PGX_NO_SORT", "1")       # hypotheses scored in the caller's order instead of locality order
        ctx = _lib.Context(0)
        try:
            ctx.score_debug_geometry(**cfg)
            ctx.set_points(mt, pts)
            ctx.set_compound(comp)
            outs.append(ctx.score(models, T2, has_compound=True, exponent=2))
            outs.append(ctx.score(models, T2, has_compound=True, exponent=2))
        finally:
            ctx.close()
    monkeypatch.delenv("PGX_NO_SORT", raising=False)
    for o in outs[1:]:
        for key in ("counts", "values", "shared", "scores"):
            assert np.array_equal(o[key], outs[0][key]), key
    ref = oracle.score(mt, pts, models, T2, compound=comp, has_compound=True, exponent=2)
    assert np.array_equal(outs[0]["counts"], ref["counts"]) and _rel(outs[0]["values"], ref["values"]) <= REL


def test_score_early_exit_predicate_is_order_free(oracle):
    # scoring_function_with_compound_model.h:105-106 fires iff count + 1 < best: pure function of the full count,
    # which is why the batched kernel needs no point ordering (checked on the oracle itself; host logic applies it).
    mt, pts, models, thr = make_case("line", 500, 20, seed=5)
    T2 = 2.25 * thr * thr
    full = oracle.score(mt, pts, models, T2)
    for best in (0, 1, 5, 50, 400):
        cut = oracle.score(mt, pts, models, T2, best_inlier_number=np.full(len(models), best))
        interrupted = full["counts"] + 1 < best
        assert np.array_equal(cut["counts"], np.where(interrupted, 0, full["counts"]))


def test_score_is_reproducible_and_batch_invariant(gpu_ctx):
    mt, pts, models, thr = make_case("pnp", 30000, 512, seed=9)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    a = gpu_ctx.score(models, T2)
    b = gpu_ctx.score(models, T2)
    assert np.array_equal(a["values"], b["values"]) and np.array_equal(a["counts"], b["counts"])
    # scoring a sub-batch gives bitwise the same per-hypothesis numbers when the point chunking is unchanged
    c = gpu_ctx.score(models[:256], T2)
    assert np.array_equal(c["counts"], a["counts"][:256])
    assert _rel(c["values"], a["values"][:256]) <= REL


# ----------------------------------------------------------------------------------------------------------------------
# a6 : unary table
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_unary_table_bit_exact(gpu_ctx, oracle, name):
    mt, pts, models, thr = make_case(name, 3001, 7, seed=21)
    gpu_ctx.set_points(mt, pts)
    for lam in (0.0, 0.1, 0.5):
        got = gpu_ctx.pearl_unary(models, thr, lam, want_table=True)
        assert np.array_equal(got, oracle.unary_q(mt, pts, models, thr, lam)), name
    got0 = gpu_ctx.pearl_unary(None, thr, 0.3, want_table=True)  # K = 0: only the outlier label
    assert got0.shape == (3001, 1) and np.all(got0 == oracle.quantize(0.7))


# ----------------------------------------------------------------------------------------------------------------------
# a9 : bucket + residual sums
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,L", [(1, 2), (255, 3), (256, 4), (257, 11), (100003, 7)])
def test_bucket_matches_oracle(gpu_ctx, oracle, n, L):
    rng = np.random.default_rng(n)
    labels = rng.integers(0, L + 1, n).astype(np.int32)  # includes values >= L-1 -> outlier bucket
    gpu_ctx.set_labels(labels)
    counts, order = gpu_ctx.bucket(L)
    rc, ro = oracle.bucket(labels, L)
    assert np.array_equal(counts, rc)
    assert np.array_equal(order, ro)  # stable: ascending point index inside every bucket (PEARL.h:342-352)
    sizes, none = gpu_ctx.bucket(L, want_order=False)   # the sizes-only form PEARL uses (integer atomics, no scan)
    assert none is None and np.array_equal(sizes, rc)


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_residual_sum(gpu_ctx, oracle, name):
    mt, pts, models, thr = make_case(name, 5003, 2, seed=31)
    labels = np.random.default_rng(1).integers(0, 3, 5003).astype(np.int32)
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.set_labels(labels)
    for k, model in enumerate(models):
        got = gpu_ctx.residual_sum(model, k)
        ref = oracle.residual_sum(mt, pts, model, labels, k)
        assert abs(got - ref) <= REL * max(abs(ref), 1e-300), name


# ----------------------------------------------------------------------------------------------------------------------
# a8 / a19 : energy, single expansion moves, full expansion
# ----------------------------------------------------------------------------------------------------------------------
MINCUT_PATHS = {"one_workgroup": {"PGX_TILE_EXPANSION_MAX": "8192"},    # every move of a graph of <= 8 192 sites on the whole-graph kernels (default: region first beyond 1 024)
                "level_synchronous": {"PGX_MF_TILE": "0", "PGX_MF_REGION": "0"},
                "region": {"PGX_MF_TILE": "0"},
                # the one-workgroup solver that keeps the capacities in memory, also where the LDS-resident one would take the move
                "one_workgroup_memory": {"PGX_TILE_MINI": "0", "PGX_TILE_EXPANSION_MAX": "8192"},
                # region moves solved by the memory-resident instance (default: the LDS-resident t_region_mini_kernel first)
                "region_memory": {"PGX_MF_TILE": "0", "PGX_TILE_MINI": "0"}}
MINCUT_COUNTED_AS = {"one_workgroup_memory": "one_workgroup", "region_memory": "region"}


@pytest.fixture
def mincut_ctx(request, monkeypatch):
    """A context per min-cut schedule (the switches are read when the context is created): the default (a graph of <= 8192
    sites is one workgroup, one launch per move), maxflow.hip's level-synchronous launches for everything, and region moves
    (the open sites of a move compacted and solved by one workgroup, enqueued a cycle at a time) even for small graphs."""
    for key in ("PGX_MF_TILE", "PGX_MF_REGION", "PGX_TILE_MINI", "PGX_TILE_EXPANSION_MAX"):
        monkeypatch.delenv(key, raising=False)
    for key, val in MINCUT_PATHS[request.param].items():
        monkeypatch.setenv(key, val)
    ctx = _lib.Context(0)
    ctx.path_name = MINCUT_COUNTED_AS.get(request.param, request.param)
    ctx.fixture_name = request.param
    yield ctx
    ctx.close()


@pytest.mark.parametrize("mincut_ctx", list(MINCUT_PATHS), indirect=True)
def test_energy_and_moves_random_small(mincut_ctx, oracle):
    # the max-flow schedule must not show in the result: the cut (minimal sink side) is unique
    gpu_ctx = mincut_ctx
    rng = np.random.default_rng(2024)
    for trial in range(60):
        n = int(rng.integers(2, 300))
        L = int(rng.integers(2, 7))
        Dq = (rng.integers(0, 1 << 20, (n, L)) << 12).astype(np.int64)
        graph = random_sym_graph(rng, n, float(rng.choice([0.0, 0.02, 0.2])))
        lam = float(rng.choice([0.0, 0.1, 0.45]))
        h = float(rng.choice([0.0, 0.0005, 0.01, 2.0]))
        lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
        labels = rng.integers(0, L, n).astype(np.int32)
        if trial % 3 == 0:
            labels[:] = rng.integers(0, L)
        gpu_ctx.set_unary_q(Dq)
        gpu_ctx.set_graph(*graph)
        gpu_ctx.set_labels(labels)
        eq, e = gpu_ctx.energy(lam, h)
        assert eq == oracle.energy(Dq, graph, lq, hq, labels)
        assert e == eq / 2.0 ** 32
        for alpha in rng.permutation(L):
            ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, int(alpha), labels)
            changed = gpu_ctx.expand_alpha(lam, h, int(alpha))
            got = gpu_ctx.get_labels()
            assert np.array_equal(got, ref), f"trial {trial} alpha {alpha}: labels differ from the oracle min-cut"
            assert changed == ref_changed
            labels = ref


@pytest.mark.parametrize("mincut_ctx", list(MINCUT_PATHS), indirect=True)
@pytest.mark.parametrize("n,lam,h", [(3000, 0.3, 10.0), (3000, 0.0, 10.0), (6000, 0.2, 3.0), (20000, 0.1, 6.0), (20000, 0.45, 0.0)])
def test_full_expansion_matches_oracle(mincut_ctx, oracle, n, lam, h):
    gpu_ctx = mincut_ctx
    Dq, graph = realistic_labeling_problem(n, L=6, lam=lam, seed=n)
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    labels0 = np.zeros(n, dtype=np.int32)
    ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, labels0)
    gpu_ctx.set_unary_q(Dq)
    gpu_ctx.set_graph(*graph)
    gpu_ctx.set_labels(labels0)
    eq, e, cycles = gpu_ctx.expansion(lam, h)
    assert np.array_equal(gpu_ctx.get_labels(), ref_labels)
    assert eq == ref_e and cycles == ref_cycles
    # idempotence: a second run from the optimum changes nothing and needs exactly one (empty) cycle
    eq2, _, cycles2 = gpu_ctx.expansion(lam, h)
    assert eq2 == eq and cycles2 == 1 and np.array_equal(gpu_ctx.get_labels(), ref_labels)
    st = gpu_ctx.expansion_stats()
    assert st["relabelled_sites"] == 0


@pytest.mark.parametrize("mincut_ctx", list(MINCUT_PATHS), indirect=True)
def test_each_mincut_path_is_the_one_that_ran(mincut_ctx, oracle):
    # pgx_expansion_paths: the schedule a fixture asks for must be the one that solved the moves
    Dq, graph = realistic_labeling_problem(3000, L=4, lam=0.2, seed=5)
    mincut_ctx.set_unary_q(Dq)
    mincut_ctx.set_graph(*graph)
    mincut_ctx.set_labels(np.zeros(3000, np.int32))
    mincut_ctx.expansion(0.2, 3.0)
    paths = mincut_ctx.expansion_paths()
    assert paths[mincut_ctx.path_name] > 0
    if mincut_ctx.path_name == "one_workgroup":
        assert paths["level_synchronous"] == 0 and paths["region"] == 0 and paths["tile_handed_back"] == 0
        # 3000 sites: beyond the LDS-resident kernel either way
        assert mincut_ctx.one_workgroup_launches()["lds_resident"] == 0 and mincut_ctx.one_workgroup_launches()["memory_resident"] > 0


def _dense_rows_graph(rng, n, deg):
    """Symmetric graph in which every site has about `deg` neighbours, some rows well beyond 16 arcs."""
    iu, ju = [], []
    for i in range(n):
        k = int(rng.integers(max(1, deg // 2), deg + 1)) if i % 7 else min(n - 1, 3 * deg)
        nb = rng.choice(n, size=min(k, n - 1), replace=False)
        for j in nb:
            if j != i:
                iu.append(min(i, j)); ju.append(max(i, j))
    pairs = np.unique(np.stack([iu, ju], 1), axis=0)
    return csr_from_pairs(n, pairs[:, 0], pairs[:, 1], rng.integers(1, 3, pairs.shape[0]))


@pytest.mark.parametrize("mini", ["1", "0"])
def test_lds_resident_one_workgroup_moves_match_oracle(oracle, monkeypatch, mini):
    """Round 6, csrc/maxflow_tile.hip t_mini_kernel: graphs of <= 1024 sites and <= 8192 arcs are solved with the whole move in LDS
    (one site per thread; 256, 512 or 1024 threads).  Single moves against the oracle's min-cut, bit for bit, at the widths' edges
    (255..257, 511..513, 1023..1024 sites), on rows longer than the 16 arcs a thread keeps in registers, with label costs from
    zero to larger than any data term (a hub that reaches t only through members WITHOUT a t-link is passed on inside this kernel;
    the memory-resident kernel hands such a move back), on labellings that do not use alpha (the gate) and on uniform labellings.
    The launch counters say which kernel was compared; PGX_TILE_MINI=0 runs the same cases through t_move_kernel."""
    monkeypatch.setenv("PGX_TILE_MINI", mini)
    ctx = _lib.Context(0)
    try:
        rng = np.random.default_rng(606)
        sizes = [2, 3, 64, 65, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024]
        moves = 0
        for trial, n in enumerate(sizes * 2):
            L = int(rng.integers(2, 8))
            kind = trial % 4
            if kind == 0:
                graph = random_sym_graph(rng, n, min(1.0, 6.0 / max(n, 2)))
            elif kind == 1:
                graph = _dense_rows_graph(rng, n, 3 if n > 600 else 7)       # rows of up to 3 * deg arcs: beyond the register table
            elif kind == 2:
                graph = realistic_labeling_problem(n, L=max(L, 2), lam=0.2, seed=trial)[1] if 64 <= n <= 600 else random_sym_graph(rng, n, min(1.0, 5.0 / n))
            else:
                graph = random_sym_graph(rng, n, min(1.0, 3.0 / max(n, 2)))
            assert graph[1].shape[0] <= 8192, "the case is meant for the LDS-resident kernel"
            Dq = (rng.integers(0, 1 << 20, (n, L)) << 12).astype(np.int64)
            if trial % 5 == 0:
                Dq[:, rng.integers(0, L)] >>= 6      # one label nearly free: most sites lose their t-link in its move
            lam = float(rng.choice([0.0, 0.1, 0.45]))
            h = float(rng.choice([0.0, 0.0005, 0.01, 2.0, 300.0]))
            lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
            labels = rng.integers(0, L, n).astype(np.int32)
            if trial % 3 == 0:
                labels[:] = rng.integers(0, L)
            elif trial % 3 == 1:
                labels[labels == L - 1] = 0          # the last label is not in use: its move goes through the gate
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(labels)
            for alpha in list(rng.permutation(L)) + [L - 1]:
                ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, int(alpha), labels)
                changed = ctx.expand_alpha(lam, h, int(alpha))
                assert np.array_equal(ctx.get_labels(), ref), f"trial {trial} n {n} alpha {alpha} h {h}: labels differ from the oracle min-cut"
                assert changed == ref_changed
                labels = ref
                moves += 1
        launches = ctx.one_workgroup_launches()
        if mini == "1":
            assert launches["lds_resident"] > 0 and launches["memory_resident"] == 0 and moves > 100
            # a graph of few sites but more arcs than the LDS tables hold goes to the memory-resident kernel
            n, L = 600, 3
            graph = _dense_rows_graph(rng, n, 14)
            assert graph[1].shape[0] > 8192
            Dq = (rng.integers(0, 1 << 20, (n, L)) << 12).astype(np.int64)
            labels = rng.integers(0, L, n).astype(np.int32)
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(labels)
            lq, hq = oracle.quantize_lambda(0.1), oracle.quantize(0.01)
            for alpha in range(L):
                ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, labels)
                assert ctx.expand_alpha(0.1, 0.01, alpha) == ref_changed and np.array_equal(ctx.get_labels(), ref)
                labels = ref
            assert ctx.one_workgroup_launches()["memory_resident"] == L
            launches = ctx.one_workgroup_launches()
            launches["memory_resident"] = 0
            assert ctx.expansion_paths()["tile_handed_back"] == 0
        else:
            assert launches["lds_resident"] == 0 and launches["memory_resident"] > 0
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["rounds", "searches", "both", "both_one_xcd_budget"])
def test_one_launch_schedules_match_oracle(oracle, monkeypatch, mode):
    """Round 6 (csrc/maxflow_xcd.hip.h): the later rounds of a move and whole global relabels as ONE persistent launch on one XCD
    (fence-free barrier over the XCD's workgroups, plain stores + L1-bypassing loads).  Forced on for every level-synchronous move
    (depth threshold 0, rounds cut to 2 sweeps so that moves end them with work left): full expansions on realistic energies and
    single moves on small problems full of ties must give the oracle's labels, energies and cycle counts, and
    pgx_expansion_schedule must show that the new launches ran."""
    for key, val in {"PGX_MF_TILE": "0", "PGX_MF_REGION": "0", "PGX_MF_XCD": "0" if mode == "searches" else "1",
                     "PGX_MF_XCD_SEARCH": "0" if mode == "rounds" else "1", "PGX_MF_XCD_MIN_DEPTH": "0", "PGX_MF_MEMO": "0",
                     "PGX_MF_SWEEPS": "96" if mode == "both_one_xcd_budget" else "2"}.items():
        monkeypatch.setenv(key, val)
    ctx = _lib.Context(0)
    try:
        for n, lam, h in ((3000, 0.3, 10.0), (20000, 0.1, 6.0), (20000, 0.45, 0.0), (60000, 0.2, 3.0)):
            Dq, graph = realistic_labeling_problem(n, L=6, lam=lam, seed=n)
            lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
            ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, np.zeros(n, np.int32))
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(np.zeros(n, np.int32))
            eq, e, cycles = ctx.expansion(lam, h)
            assert np.array_equal(ctx.get_labels(), ref_labels) and eq == ref_e and cycles == ref_cycles, (mode, n)
        rng = np.random.default_rng(606)
        for trial in range(40):
            n, L = int(rng.integers(2, 400)), int(rng.integers(2, 7))
            Dq = (rng.integers(0, 1 << 8, (n, L)) << 24).astype(np.int64)      # coarse costs: ties everywhere
            graph = random_sym_graph(rng, n, float(rng.choice([0.02, 0.2])))
            lam, h = float(rng.choice([0.1, 0.45])), float(rng.choice([0.0, 0.01, 2.0]))
            lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
            labels = rng.integers(0, L, n).astype(np.int32)
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(labels)
            for alpha in rng.permutation(L):
                ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, int(alpha), labels)
                assert ctx.expand_alpha(lam, h, int(alpha)) == ref_changed
                labels = ctx.get_labels()
                assert np.array_equal(labels, ref), (mode, trial, alpha)
        sched = ctx.expansion_schedule()
        if mode != "searches":
            assert sched["xcd_round_launches"] > 0 and sched["xcd_rounds"] >= sched["xcd_round_launches"]
        else:
            assert sched["xcd_round_launches"] == 0
        if mode != "rounds":
            assert sched["xcd_searches"] > 0
        else:
            assert sched["xcd_searches"] == 0
        assert ctx.expansion_paths()["level_synchronous"] > 0
    finally:
        ctx.close()


@pytest.mark.parametrize("n,lam,h,L", [(700, 0.3, 2.0, 3), (3000, 0.2, 3.0, 7), (8000, 0.1, 10.0, 5)])
def test_batched_one_workgroup_moves_equal_unbatched(oracle, monkeypatch, n, lam, h, L):
    """Graphs of <= 8192 sites: pgx_expansion enqueues the one-workgroup moves of a cycle back to back (the region moves' batch
    slots, poison word and on-device skip rule; one read-back per batch).  Same labels, energy, cycles and work counters as one
    host round trip per move (PGX_MF_TILE_BATCH=0), and the oracle's result; a second expansion from the optimum is one empty cycle."""
    Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=7 * n + L)
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, np.zeros(n, np.int32))
    seen = {}
    monkeypatch.setenv("PGX_TILE_EXPANSION_MAX", "8192")   # every move on the whole-graph kernel (default: beyond 4 096 sites the region path goes first)
    for batch in ("1", "0"):
        monkeypatch.setenv("PGX_MF_TILE_BATCH", batch)
        ctx = _lib.Context(0)
        try:
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(np.zeros(n, np.int32))
            eq, _, cycles = ctx.expansion(lam, h)
            assert np.array_equal(ctx.get_labels(), ref_labels) and eq == ref_e and cycles == ref_cycles, batch
            st, paths = ctx.expansion_stats(), ctx.expansion_paths()
            assert paths["one_workgroup"] > 0 and paths["region"] == 0 and paths["level_synchronous"] == 0
            seen[batch] = (st["mincuts"], st["relabelled_sites"], st["skipped_moves"], paths["one_workgroup"])
            eq2, _, cycles2 = ctx.expansion(lam, h)
            assert eq2 == eq and cycles2 == 1
        finally:
            ctx.close()
    assert seen["1"] == seen["0"]


@pytest.mark.parametrize("n,lam,h,L", [(5000, 0.2, 3.0, 6), (8000, 0.1, 10.0, 5), (4500, 0.45, 0.5, 4), (1025, 0.3, 1.0, 5), (2084, 0.1, 4.0, 7)])
def test_mid_size_graphs_region_first_then_whole_graph_kernel(oracle, n, lam, h, L):
    """Graphs of 1 025 .. 8 192 sites (C1: 2 000, C2: 5 000): an expansion move tries the region path first and, declined, is solved by the one-workgroup
    whole-graph kernel (not by the level-synchronous launches).  Labels, energy, cycles of the oracle; both solvers took moves."""
    Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=11 * n + L)
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, np.zeros(n, np.int32))
    ctx = _lib.Context(0)
    try:
        ctx.set_unary_q(Dq)
        ctx.set_graph(*graph)
        ctx.set_labels(np.zeros(n, np.int32))
        eq, _, cycles = ctx.expansion(lam, h)
        assert np.array_equal(ctx.get_labels(), ref_labels) and eq == ref_e and cycles == ref_cycles
        paths = ctx.expansion_paths()
        assert paths["region"] > 0 and paths["level_synchronous"] == 0, paths
        assert paths["region_declined"] == 0 or paths["one_workgroup"] > 0, paths
        # single moves (unbatched) take the same route
        rng = np.random.default_rng(n)
        labels = rng.integers(0, L, n).astype(np.int32)
        ctx.set_labels(labels)
        for alpha in range(L):
            ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, labels)
            assert ctx.expand_alpha(lam, h, alpha) == ref_changed and np.array_equal(ctx.get_labels(), ref)
            labels = ref
        assert ctx.expansion_paths()["level_synchronous"] == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("n,lam,h,L", [(30000, 0.15, 4.0, 6), (60000, 0.3, 0.0, 5), (60000, 0.05, 12.0, 9)])
def test_region_moves_match_oracle(oracle, monkeypatch, n, lam, h, L):
    """Graphs beyond the one-workgroup limit: a move whose OPEN sites (no t-link left after the source / sink saturation)
    number <= 8192 is solved by one workgroup on their compacted sub-graph (maxflow_tile.hip expand_alpha_region), the rest by
    maxflow.hip - the labels, energy and cycle count are the oracle's either way, and with the region path switched off."""
    Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=n + L)
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, np.zeros(n, np.int32))
    seen = {}
    for region in ("1", "0"):
        monkeypatch.setenv("PGX_MF_REGION", region)
        ctx = _lib.Context(0)
        try:
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(np.zeros(n, np.int32))
            eq, _, cycles = ctx.expansion(lam, h)
            assert np.array_equal(ctx.get_labels(), ref_labels) and eq == ref_e and cycles == ref_cycles
            seen[region] = ctx.expansion_paths()
        finally:
            ctx.close()
    assert seen["1"]["region"] > 0 and seen["1"]["one_workgroup"] == 0
    assert seen["0"]["region"] == 0 and seen["0"]["level_synchronous"] > 0


def test_region_moves_decline_a_domino_of_weak_sinks(oracle, monkeypatch):
    """A path on which every site but the first barely prefers its label (sink capacity of one unit) while the arcs are strong: the
    first site's excess saturates its neighbour's t-link, which makes that neighbour a member of the region, whose arc saturates
    the next ... the region grows by one site per promotion round.  After its round limit the region must be DECLINED - the last
    members have not charged their arcs to their neighbours yet, so "this neighbour keeps its t-link" cannot be checked - and the
    general path solves the move (regression test: the limit used to fall through to the build with incomplete sums)."""
    monkeypatch.setenv("PGX_MF_TILE", "0")          # small graph through the region path
    n, lam = 3000, 0.5
    a = np.arange(n - 1)
    graph = csr_from_pairs(n, a, a + 1, np.full(n - 1, 2))
    Dq = np.zeros((n, 2), np.int64)
    Dq[:, 1] = 1                                      # every site: label 1 costs one unit (2^-32) more than label 0 ...
    Dq[0] = (1 << 40, 0)                              # ... except the first, which wants label 1 badly
    lq, hq = oracle.quantize_lambda(lam), 0
    start = np.zeros(n, np.int32)
    ref, re, rc = oracle.expansion(Dq, graph, lq, hq, start.copy())
    ctx = _lib.Context(0)
    try:
        ctx.set_unary_q(Dq)
        ctx.set_graph(*graph)
        ctx.set_labels(start.copy())
        eq, _, cyc = ctx.expansion(lam, 0.0)
        assert np.array_equal(ctx.get_labels(), ref) and eq == re and cyc == rc
        assert ref.sum() == n                         # the domino: everybody follows the first site
        paths = ctx.expansion_paths()
        assert paths["region_declined"] > 0 and paths["level_synchronous"] > 0
    finally:
        ctx.close()


def test_region_moves_decline_wide_graphs(oracle):
    """A graph with a site of more than 32 neighbours: the region path's arc rows do not fit, every move goes to maxflow.hip
    (which must then initialise the move itself - the regression of the fused initialisation)."""
    rng = np.random.default_rng(9)
    n, L, lam, h = 12000, 4, 0.2, 2.0
    Dq, (off, idx, mult) = realistic_labeling_problem(n, L=L, lam=lam, seed=99)
    src = np.repeat(np.arange(n), np.diff(off))
    iu, ju, mu = src[src < idx], idx[src < idx], mult[src < idx]
    for hub in (17, 5000):   # two stars of ~60 arcs each
        spokes = np.setdiff1d(rng.choice(n, 60, replace=False), np.concatenate([[hub], idx[off[hub]:off[hub + 1]]]))
        iu = np.concatenate([iu, np.minimum(hub, spokes)])
        ju = np.concatenate([ju, np.maximum(hub, spokes)])
        mu = np.concatenate([mu, np.ones(spokes.size, mu.dtype)])
    graph = csr_from_pairs(n, iu, ju, mu)
    assert np.diff(graph[0]).max() > 32
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    ref_labels, ref_e, ref_cycles = oracle.expansion(Dq, graph, lq, hq, np.zeros(n, np.int32))
    ctx = _lib.Context(0)
    try:
        ctx.set_unary_q(Dq)
        ctx.set_graph(*graph)
        ctx.set_labels(np.zeros(n, np.int32))
        eq, _, cycles = ctx.expansion(lam, h)
        assert np.array_equal(ctx.get_labels(), ref_labels) and eq == ref_e and cycles == ref_cycles
        paths = ctx.expansion_paths()
        assert paths["region"] == 0 and paths["region_declined"] == 0 and paths["level_synchronous"] > 0
    finally:
        ctx.close()


def test_expansion_energy_never_increases(gpu_ctx, oracle):
    Dq, graph = realistic_labeling_problem(5000, L=5, lam=0.2, seed=77)
    lam, h = 0.2, 8.0
    gpu_ctx.set_unary_q(Dq)
    gpu_ctx.set_graph(*graph)
    gpu_ctx.set_labels(np.random.default_rng(0).integers(0, 5, 5000).astype(np.int32))
    prev, _ = gpu_ctx.energy(lam, h)
    for cycle in range(3):
        for alpha in range(5):
            gpu_ctx.expand_alpha(lam, h, alpha)
            cur, _ = gpu_ctx.energy(lam, h)
            assert cur <= prev
            prev = cur


# ----------------------------------------------------------------------------------------------------------------------
# a20: neighbourhood graph built on the GPU — index work, bit-exact against the oracle's lists
# ----------------------------------------------------------------------------------------------------------------------
def _graph_case(rng, n, d, style):
    if style == "uniform":
        return rng.random((n, d)) * 100.0
    if style == "integer":      # many exact ties: ranking by (squared distance, index) decides
        return rng.integers(0, 12, (n, d)).astype(np.float64)
    if style == "clustered":
        c = rng.random((8, d)) * 1000.0
        return c[rng.integers(0, 8, n)] + rng.normal(0, 3.0, (n, d))
    raise ValueError(style)


@pytest.mark.parametrize("style", ["uniform", "integer", "clustered"])
@pytest.mark.parametrize("n,d,k", [(1, 2, 5), (2, 4, 5), (7, 5, 5), (300, 2, 8), (2500, 4, 5), (2500, 5, 16), (3000, 3, 1)])
def test_graph_build_matches_oracle(gpu_ctx, oracle, style, n, d, k):
    rng = np.random.default_rng(n * 31 + d * 7 + k)
    pts = _graph_case(rng, n, d, style)
    for kind, radius in ((_lib.GRAPH_KNN_IN_BALL, 0.5), (_lib.GRAPH_KNN_IN_BALL, 9.0), (_lib.GRAPH_KNN_IN_BALL, 1e4),
                         (_lib.GRAPH_KNN, 0.0)):
        ref = oracle.graph_build(pts, kind, radius=radius, k=k)
        got = gpu_ctx.graph_build(pts, kind, radius=radius, k=k)
        for name, a, b in zip(("off", "idx", "mult"), got, ref):
            assert np.array_equal(a, b), f"{style} n={n} d={d} k={k} kind={kind} r={radius}: {name} differs"


@pytest.mark.parametrize("style", ["uniform", "integer", "clustered"])
@pytest.mark.parametrize("n,d", [(1, 2), (2, 4), (300, 2), (2500, 4), (3000, 5), (30000, 3)])
def test_graph_ball_matches_oracle(gpu_ctx, oracle, style, n, d):
    # exhaustive ball (PGX_GRAPH_BALL): variable degree, symmetric lists, multiplicity 2 everywhere
    rng = np.random.default_rng(n * 17 + d)
    if n > 5000 and style == "integer":
        pytest.skip("a 12^3 lattice with 30000 points is one dense clique: quadratic output")
    pts = _graph_case(rng, n, d, style)
    for radius in (0.5, 3.0, 9.0) if n <= 5000 else (0.5, 3.0):
        ref = oracle.graph_build(pts, _lib.GRAPH_BALL, radius=radius)
        got = gpu_ctx.graph_build(pts, _lib.GRAPH_BALL, radius=radius)
        for name, a, b in zip(("off", "idx", "mult"), got, ref):
            assert np.array_equal(a, b), f"{style} n={n} d={d} r={radius}: {name} differs"
    if n == 2500:   # an independent construction: scipy's kd-tree pairs (distances well away from the radius)
        import host_graph
        ref = host_graph.radius_graph(pts, 3.0)
        got = gpu_ctx.graph_build(pts, _lib.GRAPH_BALL, radius=3.0)
        if style == "uniform":
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))


def test_graph_build_feeds_the_expansion(gpu_ctx, oracle):
    # the resident graph of pgx_graph_build is the one the moves run on: same labels as with pgx_set_graph(oracle CSR)
    rng = np.random.default_rng(5)
    n, L = 4000, 4
    pts = rng.random((n, 4)) * 50.0
    Dq = (rng.integers(0, 1 << 20, (n, L)) << 12).astype(np.int64)
    lam, h = 0.2, 0.5
    ref_graph = oracle.graph_build(pts, _lib.GRAPH_KNN_IN_BALL, radius=4.0, k=5)
    ref_labels, _, _ = oracle.expansion(Dq, ref_graph, oracle.quantize_lambda(lam), oracle.quantize(h), np.zeros(n, np.int32))
    gpu_ctx.set_unary_q(Dq)
    gpu_ctx.graph_build(pts, _lib.GRAPH_KNN_IN_BALL, radius=4.0, k=5, fetch=False)
    gpu_ctx.set_labels(np.zeros(n, np.int32))
    gpu_ctx.expansion(lam, h)
    assert np.array_equal(gpu_ctx.get_labels(), ref_labels)


def test_graph_build_c4_size_properties(gpu_ctx, oracle):
    x1, x2, K, gt, _ = datasets.make_poses(seed=0)
    pts = np.column_stack([x1, x2])                      # the un-normalised 5-D rows the reference hands to FLANN
    n = pts.shape[0]
    off, idx, mult = gpu_ctx.graph_build(pts, _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5)
    deg = np.diff(off)
    assert off[0] == 0 and off[-1] == idx.size and deg.min() >= 0
    rows = np.repeat(np.arange(n), deg)
    assert (idx != rows).all() and ((mult == 1) | (mult == 2)).all()
    assert (np.diff(idx.astype(np.int64) + rows * n) > 0).all()                       # rows sorted, no duplicates
    key_fwd = np.sort(rows * n + idx)
    key_bwd = np.sort(idx.astype(np.int64) * n + rows)
    assert np.array_equal(key_fwd, key_bwd)                                           # symmetric
    # every directed list entry is counted once: sum of multiplicities = 2 * number of list entries
    sample = np.random.default_rng(0).choice(n, 2000, replace=False)
    lists = oracle.graph_lists(pts, 5, radius=20.0, rows=sample)
    back = {}
    for r, i in enumerate(sample):
        mine = [int(j) for j in lists[r] if j >= 0]
        row = idx[off[i]:off[i + 1]].tolist()
        assert set(mine) <= set(row)                                                   # own list is part of the row
        back[int(i)] = (mine, row, mult[off[i]:off[i + 1]].tolist())
    # multiplicities: 2 iff both lists hold the pair (checked on the neighbours' own lists)
    others = sorted({j for mine, row, _ in back.values() for j in row})
    olists = dict(zip(others, oracle.graph_lists(pts, 5, radius=20.0, rows=np.array(others)).tolist()))
    for i, (mine, row, ms) in back.items():
        for j, m in zip(row, ms):
            assert m == int(j in mine) + int(i in olists[j])


def test_next_rows_against_committed_vectors(gpu_ctx):
    # the GPU against tests/golden/kat_next_v1.npz directly (no oracle in the loop)
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_next_v1.npz"))
    for d in (2, 4, 5):
        for tag, kind, radius, k in (("ball", _lib.GRAPH_KNN_IN_BALL, 6.0, 5), ("knn", _lib.GRAPH_KNN, 0.0, 8)):
            got = gpu_ctx.graph_build(g[f"g{d}_pts"], kind, radius=radius, k=k)
            for name, a in zip(("off", "idx", "mult"), got):
                assert np.array_equal(a, g[f"g{d}_{tag}_{name}"]), (d, tag, name)
    for name in ("line", "vanishing_point", "homography", "fundamental", "pnp"):
        gpu_ctx.set_points(MODEL_CASES[name], g[f"s_{name}_pts"])
        assert np.array_equal(gpu_ctx.solve_minimal(g[f"s_{name}_samples"]), g[f"s_{name}_models"], equal_nan=True)
    prm = np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0])
    for name, kind, p in (("vanishing_point", _lib.GRAM_VP, None), ("homography", _lib.GRAM_DLT_H, prm),
                          ("fundamental", _lib.GRAM_EPI_F, prm), ("pnp", _lib.GRAM_PNP_GN, "model")):
        gpu_ctx.set_points(MODEL_CASES[name], g[f"m_{name}_pts"])
        p = g[f"m_{name}_model"][:12] if isinstance(p, str) else p
        G, cnt, bad = gpu_ctx.gram(kind, ("index", g[f"m_{name}_idx"]), params=p, weights=g[f"m_{name}_w"], wpow=2)
        ref = g[f"m_{name}_G{kind}"]
        assert cnt == 120 and bad == 0 and np.abs(G - ref).max() <= REL * np.abs(ref).max()


def test_graph_build_error_paths(gpu_ctx):
    pts = np.random.default_rng(0).random((10, 4))
    with pytest.raises(_lib.PgxError):
        gpu_ctx.graph_build(pts, 7, radius=1.0, k=5)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.graph_build(pts, _lib.GRAPH_KNN_IN_BALL, radius=0.0, k=5)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.graph_build(pts, _lib.GRAPH_KNN, k=17)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.graph_build(np.random.default_rng(0).random((10, 6)), _lib.GRAPH_KNN, k=3)
    bad = pts.copy()
    bad[3, 0] = np.nan
    with pytest.raises(_lib.PgxError):
        gpu_ctx.graph_build(bad, _lib.GRAPH_KNN, k=3)


# ----------------------------------------------------------------------------------------------------------------------
# SURVEY 8f rank 1 (first slice): minimal solvers on the GPU — bit-exact models, then scored where they are
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["line", "vanishing_point", "homography", "fundamental", "pnp"])
def test_solve_minimal_matches_oracle_and_scores_in_place(gpu_ctx, oracle, name):
    mt, pts, models, thr = make_case(name, 5000, 4, seed=9)
    rng = np.random.default_rng(4)
    m = {"fundamental": 7, "homography": 4, "pnp": 3}.get(name, 2)
    slots = {"fundamental": 3, "pnp": 4}.get(name, 1)
    samples = rng.integers(0, 5000, (3000, m)).astype(np.int32)
    if name in ("fundamental", "pnp"):                      # some all-inlier samples of one structure as well
        for s in range(100, 400):
            samples[s] = rng.choice(np.nonzero(np.arange(5000) % 5 == s % 3)[0], m, replace=False)
    samples[:40, 1] = samples[:40, 0]                      # degenerate: the same point / segment twice
    gpu_ctx.set_points(mt, pts)
    got = gpu_ctx.solve_minimal(samples)
    ref = oracle.solve_minimal(mt, pts, samples)
    assert got.shape == ref.shape == (3000 * slots, {"fundamental": 9, "homography": 9, "pnp": 12}.get(name, 3))
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got[:40 * slots]).all()
    ok = ~np.isnan(ref[:, 0])
    assert np.array_equal(got[ok], ref[ok]), "generated hypotheses must be bit-identical to the oracle's"
    T2 = 9.0 / 4.0 * thr * thr
    gpu_ctx.score_launch(T2)                              # scores the resident, device-generated batch
    a = gpu_ctx.score_fetch()
    b = oracle.score(mt, pts, np.where(np.isnan(ref), np.nan, ref), T2)
    assert np.array_equal(a["counts"], b["counts"]) and a["counts"][:40 * slots].max() == 0
    assert _rel(a["values"], b["values"]) < REL
    up = gpu_ctx.score(ref[ok], T2)                       # the same models through the upload path
    assert np.array_equal(up["counts"], a["counts"][ok])


def test_solve_minimal_error_paths(gpu_ctx):
    mt, pts, models, thr = make_case("homography_sym", 100, 1, seed=1)
    gpu_ctx.set_points(mt, pts)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.solve_minimal(np.zeros((4, 2), np.int32))          # 18-parameter models are generated on the host
    mt, pts, models, thr = make_case("line", 100, 1, seed=1)
    gpu_ctx.set_points(mt, pts)
    out = gpu_ctx.solve_minimal(np.array([[0, 1], [5, 100], [-1, 2]], np.int32))
    assert np.isfinite(out[0]).all() and np.isnan(out[1]).all() and np.isnan(out[2]).all()   # out-of-range index -> no model


# ----------------------------------------------------------------------------------------------------------------------
# a9: Gram pass of the non-minimal refits — floating-point sums, 1e-9 relative to the matrix scale
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("n", [1, 257, 20011])
def test_gram_matches_oracle(gpu_ctx, oracle, name, n):
    mt, pts, models, thr = make_case(name, n, 2, seed=n + 11)
    rng = np.random.default_rng(n)
    gpu_ctx.set_points(mt, pts)
    labels = rng.integers(0, 3, n).astype(np.int32)
    gpu_ctx.set_labels(labels)
    weights = rng.random(n) + 0.5
    kinds = {"line": [(_lib.GRAM_AFFINE, None)],
             "vanishing_point": [(_lib.GRAM_VP, None), (_lib.GRAM_AFFINE, None)],
             "homography": [(_lib.GRAM_AFFINE, None), (_lib.GRAM_DLT_H, np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0]))],
             "homography_sym": [(_lib.GRAM_DLT_H, np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0]))],
             "fundamental": [(_lib.GRAM_EPI_F, np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0]))],
             "pnp": [(_lib.GRAM_AFFINE, None), (_lib.GRAM_PNP_GN, models[0][:12])]}[name]
    index = rng.permutation(n)[: max(1, n // 2)]
    for kind, prm in kinds:
        for sel, ref_index in ((("index", index), index), (("label", 1), np.nonzero(labels == 1)[0])):
            for w, wpow in ((None, 2), (weights, 1), (weights, 2)):
                G, cnt, bad = gpu_ctx.gram(kind, sel, params=prm, weights=w, wpow=wpow)
                Gr, cntr, badr = oracle.gram(kind, pts, ref_index, params=prm, weights=w, wpow=wpow)
                assert (cnt, bad) == (cntr, badr)
                scale = max(np.abs(Gr).max(), 1e-300)
                assert np.abs(G - Gr).max() <= REL * scale, f"{name} kind {kind} sel {sel[0]} wpow {wpow}"
                assert np.array_equal(G, G.T)
    # reproducible run to run (fixed reduction tree)
    kind, prm = kinds[-1]
    a = gpu_ctx.gram(kind, ("label", 1), params=prm, weights=weights)[0]
    b = gpu_ctx.gram(kind, ("label", 1), params=prm, weights=weights)[0]
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("B,m", [(1, 1), (3, 14), (50, 49), (7, 64), (5, 200)])
def test_gram_batch_matches_oracle(gpu_ctx, oracle, name, B, m):
    """pgx_gram_batch (one wave per selection, per-selection parameter blocks) against the oracle's per-selection Gram."""
    n = 5000
    mt, pts, models, thr = make_case(name, n, 2, seed=B + m)
    rng = np.random.default_rng(B * 1000 + m)
    gpu_ctx.set_points(mt, pts)
    weights = rng.random(n) + 0.5
    norm = lambda: np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0]) * (1.0 + 0.1 * rng.random(6))
    kinds = {"line": [(_lib.GRAM_AFFINE, None)],
             "vanishing_point": [(_lib.GRAM_VP, None)],
             "homography": [(_lib.GRAM_AFFINE, None), (_lib.GRAM_DLT_H, norm)],
             "homography_sym": [(_lib.GRAM_DLT_H, norm)],
             "fundamental": [(_lib.GRAM_EPI_F, norm)],
             "pnp": [(_lib.GRAM_PNP_GN, lambda: models[0][:12] + 1e-3 * rng.random(12))]}[name]
    index = np.array([rng.choice(n, m, replace=False) for _ in range(B)])
    for kind, make in kinds:
        prm = None if make is None else np.array([make() for _ in range(B)])
        for w, wpow in ((None, 2), (weights, 1), (weights, 2)):
            G, bad = gpu_ctx.gram_batch(kind, index, params=prm, weights=w, wpow=wpow)
            assert G.shape[0] == B and bad.shape == (B,)
            for b in range(B):
                Gr, _, badr = oracle.gram(kind, pts, index[b], params=None if prm is None else prm[b], weights=w, wpow=wpow)
                assert int(bad[b]) == badr
                assert np.abs(G[b] - Gr).max() <= REL * max(np.abs(Gr).max(), 1e-300), f"{name} kind {kind} b={b} wpow {wpow}"
                assert np.array_equal(G[b], G[b].T)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.gram_batch(kinds[0][0], np.full((2, 3), n, np.int32), params=None if kinds[0][1] is None else np.zeros((2, 6 if name != "pnp" else 12)))


def test_batched_refits_equal_single_refits(gpu_ctx):
    """Estimator.nonminimal_batch (coroutines in lockstep over pgx_gram_batch) returns what nonminimal returns per
    selection (same algebra; only the reduction tree of the Gram pass differs: 1e-9)."""
    from pyprogressivex import _estimators
    rng = np.random.default_rng(4)
    for name in ("line", "vanishing_point", "homography", "homography_sym", "fundamental", "pnp"):
        mt, pts, models, thr = make_case(name, 4000, 1, seed=9)
        est = _estimators.ESTIMATORS[name]()
        gpu_ctx.set_points(mt, pts)
        m = 7 * est.sample_size
        one = gpu_ctx.score(models[:1], 2.25 * thr * thr, want_masks=True)
        inl = np.nonzero(np.unpackbits(one["masks"][0].view(np.uint8), bitorder="little")[:len(pts)])[0]
        assert len(inl) > m
        picks = np.array([np.sort(rng.choice(inl, m, replace=False)) for _ in range(20)])
        init = models[0] if name == "pnp" else None
        batch = est.nonminimal_batch(gpu_ctx, picks, None, init=init)
        for b in range(len(picks)):
            single = est.nonminimal(gpu_ctx, ("index", picks[b]), None, init=init)
            assert len(single) == len(batch[b]) == 1
            a, c = np.asarray(single[0]), np.asarray(batch[b][0])
            if np.dot(a, c) < 0 and name in ("line", "vanishing_point", "fundamental"):
                c = -c                                             # eigenvector sign
            assert np.abs(a - c).max() <= 1e-6 * max(1.0, np.abs(a).max()), name


def test_epipolar_support_matches_oracle(gpu_ctx, oracle):
    """pgx_epipolar_support (U-14, the F estimator's symmetric-epipolar support): both counts equal to the oracle's over random and
    near-true fundamental matrices, degenerate ones included (rank one, NaN)."""
    mt, pts, models, thr = make_case("fundamental", 30011, 6, seed=5)
    gpu_ctx.set_points(mt, pts)
    rng = np.random.default_rng(3)
    T2 = 2.25 * thr * thr
    cands = list(models) + [rng.normal(0, 1, 9) for _ in range(4)]
    cands.append(np.outer([0.0, 1.0, -500.0], [0.0, 1.0, -480.0]).reshape(-1))
    cands.append(np.full(9, np.nan))
    seen = 0
    for F in cands:
        for S2 in (T2, 4.0 * T2, 1e-9):
            got = gpu_ctx.epipolar_support(F, T2, S2)
            ref = oracle.epipolar_support(pts, F, T2, S2)
            assert got == ref
            seen = max(seen, got[0])
    assert seen > 1000
    mt2, pts2, _, _ = make_case("homography", 500, 1, seed=1)
    gpu_ctx.set_points(mt2, pts2)
    with pytest.raises(_lib.PgxError):
        gpu_ctx.epipolar_support(cands[0], T2, T2)


def test_device_pose_refits_reproduce_the_host_iteration(gpu_ctx):
    """pgx_pnp_refine_batch (all Gauss-Newton steps of a batch in one launch, 6x6 pseudo-inverse by Jacobi on the device)
    against PnPEstimator._fit_many driven step by step through pgx_gram_batch with numpy's pinv on the host: the same
    iterates up to rounding (1e-9), with and without weights, from perturbed starts; selections that fail (a point on the
    camera plane, fewer than 4 points) fail in both."""
    from pyprogressivex import _estimators
    rng = np.random.default_rng(12)
    mt, pts, models, thr = make_case("pnp", 6000, 3, seed=21)
    est = _estimators.ESTIMATORS["pnp"]()
    gpu_ctx.set_points(mt, pts)
    weights = rng.random(len(pts)) + 0.5
    for k in range(3):
        one = gpu_ctx.score(models[k:k + 1], 2.25 * thr * thr, want_masks=True)
        inl = np.nonzero(np.unpackbits(one["masks"][0].view(np.uint8), bitorder="little")[:len(pts)])[0]
        for m in (21, 64, 150):
            if len(inl) <= m:
                continue
            picks = np.array([np.sort(rng.choice(inl, m, replace=False)) for _ in range(33)])
            init = models[k].reshape(3, 4).copy()
            init[:, 3] += rng.normal(0, 0.01, 3)             # a start slightly off: several steps to converge
            init = init.reshape(-1)
            for w in (None, weights):
                dev = est.nonminimal_batch(gpu_ctx, picks, w, init=init)

                def gram(kind, prm, use_w, wpow, rows):
                    G, bad = gpu_ctx.gram_batch(kind, picks[rows], params=prm, weights=w if use_w else None, wpow=wpow)
                    return G, np.full(len(rows), m, dtype=np.int64), bad
                host = est._fit_many(gram, len(picks), [init] * len(picks))
                for b in range(len(picks)):
                    assert len(dev[b]) == len(host[b]) == 1
                    assert np.abs(dev[b][0] - host[b][0]).max() <= 1e-9 * max(1.0, np.abs(host[b][0]).max())
    # failures: fewer than 4 points; a point exactly on the camera plane of the start
    P, ok = gpu_ctx.pnp_refine_batch(np.tile(models[0], (2, 1)), inl[:6].reshape(2, 3).astype(np.int32))
    assert not ok.any()
    X = pts[inl[0], 2:5]
    flat = models[0].reshape(3, 4).copy()
    flat[2, 3] = -float(flat[2, :3] @ X)                     # z_c = 0 for that point
    P, ok = gpu_ctx.pnp_refine_batch(np.stack([flat.reshape(-1), models[0]]), np.stack([inl[:8], inl[:8]]).astype(np.int32))
    assert not ok[0] and ok[1]
    with pytest.raises(_lib.PgxError):
        gpu_ctx.pnp_refine_batch(models[:1], np.full((1, 5), len(pts), np.int32))


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_label_batched_gram_and_residual_sums_are_bitwise_the_single_label_calls(gpu_ctx, name):
    # pgx_gram_labels / pgx_residual_sums: all instances of a PEARL iteration in one launch, same trees as the single calls
    n, K = 30011, 5
    mt, pts, models, thr = make_case(name, n, K, seed=4)
    rng = np.random.default_rng(8)
    gpu_ctx.set_points(mt, pts)
    labels = rng.integers(0, K + 1, n).astype(np.int32)        # label K = outliers, label 3 stays empty
    labels[labels == 3] = K
    gpu_ctx.set_labels(labels)
    weights = rng.random(n) + 0.5
    sums = gpu_ctx.residual_sums(models)
    for k in range(K):
        assert sums[k] == gpu_ctx.residual_sum(models[k], k)
    norm = np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0])
    kinds = {"line": [(_lib.GRAM_AFFINE, None)], "vanishing_point": [(_lib.GRAM_VP, None)],
             "homography": [(_lib.GRAM_AFFINE, None), (_lib.GRAM_DLT_H, np.array([norm * (1 + 0.1 * k) for k in range(K)]))],
             "homography_sym": [(_lib.GRAM_DLT_H, np.array([norm * (1 + 0.1 * k) for k in range(K)]))],
             "fundamental": [(_lib.GRAM_EPI_F, np.array([norm * (1 + 0.1 * k) for k in range(K)]))],
             "pnp": [(_lib.GRAM_PNP_GN, models[:, :12])]}[name]
    for kind, prm in kinds:
        for w, wpow in ((None, 2), (weights, 1), (weights, 2)):
            G, cnt, bad = gpu_ctx.gram_labels(kind, K, params=prm, weights=w, wpow=wpow)
            for k in range(K):
                Gk, ck, bk = gpu_ctx.gram(kind, ("label", k), params=None if prm is None else prm[k], weights=w, wpow=wpow)
                assert np.array_equal(G[k], Gk) and (int(cnt[k]), int(bad[k])) == (ck, bk), f"{name} kind {kind} label {k}"
            assert cnt[3] == 0 and not G[3].any()


def test_gram_error_paths_and_empty_selection(gpu_ctx):
    mt, pts, models, thr = make_case("homography", 100, 1, seed=1)
    gpu_ctx.set_points(mt, pts)
    G, cnt, bad = gpu_ctx.gram(_lib.GRAM_AFFINE, ("index", np.zeros(0, np.int32)))
    assert cnt == 0 and not G.any()
    with pytest.raises(_lib.PgxError):
        gpu_ctx.gram(_lib.GRAM_AFFINE, ("index", np.array([100], np.int32)))
    with pytest.raises(_lib.PgxError):
        gpu_ctx.gram(_lib.GRAM_DLT_H, ("index", np.array([1], np.int32)), params=np.ones(5))
    with pytest.raises(_lib.PgxError):
        gpu_ctx.gram(_lib.GRAM_VP + 9, ("index", np.array([1], np.int32)))
    with pytest.raises(_lib.PgxError):
        gpu_ctx.gram(_lib.GRAM_AFFINE, ("index", np.array([1], np.int32)), weights=np.ones(100), wpow=3)
    # weights are resident and length-checked (a short host array used to be read for n doubles)
    with pytest.raises(ValueError, match="one entry per point"):
        gpu_ctx.gram(_lib.GRAM_AFFINE, ("index", np.array([1], np.int32)), weights=np.ones(5))
    with pytest.raises(ValueError, match="one entry per point"):
        gpu_ctx.gram_batch(_lib.GRAM_AFFINE, np.array([[1, 2]], np.int32), weights=np.ones(5))
    import ctypes as C
    w = np.ones(5)
    assert gpu_ctx._lib.pgx_set_weights(gpu_ctx._h, w.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(5)) != 0
    assert b"5 weights for 100 points" in gpu_ctx._lib.pgx_last_error(gpu_ctx._h)
    out = np.zeros(15)
    cnt, bad = C.c_int64(), C.c_int64()
    idx = np.array([1], np.int32)
    gpu_ctx.set_weights(None)      # (the wpow = 3 call above had uploaded its weights before failing)
    rc = gpu_ctx._lib.pgx_gram(gpu_ctx._h, C.c_int(_lib.GRAM_AFFINE), None, C.c_int(0), C.c_int(0),
                               idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(1), C.c_int(0), C.c_int(1), C.c_int(2),
                               out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.byref(bad))
    assert rc != 0 and b"no weights are resident" in gpu_ctx._lib.pgx_last_error(gpu_ctx._h)
    # set_points invalidates the resident weights of the previous point set
    gpu_ctx.set_weights(np.ones(100))
    gpu_ctx.set_points(mt, pts[:50])
    rc = gpu_ctx._lib.pgx_gram(gpu_ctx._h, C.c_int(_lib.GRAM_AFFINE), None, C.c_int(0), C.c_int(0),
                               idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(1), C.c_int(0), C.c_int(1), C.c_int(2),
                               out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.byref(bad))
    assert rc != 0


def test_compound_update_takes_more_than_32_models(gpu_ctx, oracle):
    """the 32-pointer kernel argument is chunked (max is associative and exact)."""
    mt, pts, models, thr = make_case("line", 3000, 40, seed=2)
    gpu_ctx.set_points(mt, pts)
    T2 = 2.25 * thr * thr
    prefs = []
    for k in range(40):
        prefs.append(gpu_ctx.preference(models[k], T2, slot=k, want_pref=True)["pref"])
    comp = gpu_ctx.compound_update(list(range(40)), want_compound=True)
    assert np.array_equal(comp, np.max(np.stack(prefs), axis=0))
    comp = gpu_ctx.compound_update([39, 3, 35], want_compound=True)
    assert np.array_equal(comp, np.max(np.stack([prefs[39], prefs[3], prefs[35]]), axis=0))


def test_asymmetric_graph_is_rejected(gpu_ctx):
    from pyprogressivex._lib import PgxError
    off = np.array([0, 1, 1], dtype=np.int32)
    with pytest.raises(PgxError):
        gpu_ctx.set_graph(off, np.array([1], np.int32), np.array([1], np.int32))


# ----------------------------------------------------------------------------------------------------------------------
# GC-RANSAC's inlier/outlier graph cut (pgx_gc_labeling) — flags bit-exact against the oracle, which
# builds upstream's add_term1/add_term2 graph while the device solves the re-parameterised Potts form
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
@pytest.mark.parametrize("n,lam", [(2, 0.5), (65, 0.1), (3000, 0.3), (20011, 0.14), (20011, 0.9)])
def test_gc_labeling_matches_oracle(gpu_ctx, oracle, name, n, lam):
    mt, pts, models, thr = make_case(name, n, 6, seed=n)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    graph = gpu_ctx.graph_build(pts, _lib.GRAPH_KNN, k=min(6, max(1, n - 1)))
    seen = set()
    for m in models:
        ref = oracle.gc_labeling(mt, pts, m, T2, lam, graph)
        got = gpu_ctx.gc_labeling(m, T2, lam)
        assert got.dtype == np.int32 and np.array_equal(got, ref), f"{name} n={n} lam={lam}: {int((got != ref).sum())} flags differ"
        idx = gpu_ctx.gc_inliers(m, T2, lam)      # the same cut as ascending indices (device compaction)
        assert idx.dtype == np.int64 and np.array_equal(idx, np.flatnonzero(ref))
        seen.add(int(ref.sum()))
    if n >= 3000:
        assert max(seen) > n // 20          # the ground-truth hypotheses keep a real inlier set
    # a degenerate (NaN) model: every residual counts as beyond the threshold (e = 1), on both sides
    bad = np.full(models.shape[1], np.nan)
    assert np.array_equal(gpu_ctx.gc_labeling(bad, T2, lam), oracle.gc_labeling(mt, pts, bad, T2, lam, graph))
    assert np.array_equal(gpu_ctx.gc_inliers(bad, T2, lam), np.flatnonzero(oracle.gc_labeling(mt, pts, bad, T2, lam, graph)))


def test_gc_labeling_orientations_agree_including_ties(oracle, monkeypatch):
    """The cut is solved with the terminals swapped (every site starts "outlier", alpha = "inlier", alpha goes to the minimal
    SOURCE side - maxflow.hip mf_k_src_*) because the stated orientation makes 95 % of the sites hold excess; PGX_GC_FLIP=0
    keeps the stated one.  Both must give the oracle's flags bit for bit - also where min cuts are NOT unique: duplicated
    points, integer lattices (exact ties between unary and pairwise terms), lambda = 0.5, models exactly through points."""
    monkeypatch.setenv("PGX_GC_FLIP", "0")
    plain = _lib.Context(0)
    monkeypatch.setenv("PGX_GC_FLIP", "1")
    flipped = _lib.Context(0)
    rng = np.random.default_rng(31)
    try:
        for trial in range(24):
            n = int(rng.choice([2, 7, 64, 500, 3000, 12000]))
            kind = trial % 4
            if kind == 0:      # integer lattice + the line x = k: residuals are integers, T2 an integer: ties galore
                pts = rng.integers(0, 12, (n, 2)).astype(np.float64)
                mt, model, T2 = _lib.LINE2D, np.array([1.0, 0.0, -float(rng.integers(0, 12))]), float(rng.choice([1.0, 4.0, 9.0]))
            elif kind == 1:    # heavy duplication
                mt, p0, models, thr = make_case("homography", max(4, n // 10), 3, seed=trial)
                pts = p0[rng.integers(0, len(p0), n)]
                model, T2 = models[0], 2.25 * thr * thr
            else:
                name = ["pnp", "fundamental", "vanishing_point", "line"][trial % 4]
                mt, pts, models, thr = make_case(name, n, 3, seed=100 + trial)
                model, T2 = models[trial % 3], 2.25 * thr * thr
            lam = float(rng.choice([0.5, 0.25, 0.1, 0.75]))
            k = min(6, max(1, n - 1))
            for ctx in (plain, flipped):
                ctx.set_points(mt, pts)
            graph = plain.graph_build(pts, _lib.GRAPH_KNN, k=k)
            flipped.graph_build(pts, _lib.GRAPH_KNN, k=k, fetch=False)
            ref = oracle.gc_labeling(mt, pts, model, T2, lam, graph)
            a = plain.gc_labeling(model, T2, lam)
            b = flipped.gc_labeling(model, T2, lam)
            assert np.array_equal(a, ref), (trial, "stated orientation", int((a != ref).sum()))
            assert np.array_equal(b, ref), (trial, "swapped terminals", int((b != ref).sum()))
    finally:
        plain.close()
        flipped.close()


def test_gc_labeling_smooths_and_keeps_the_pearl_state(gpu_ctx, oracle):
    # the pairwise term changes the outcome of plain thresholding (points just beyond the threshold inside an inlier
    # neighbourhood are pulled in) but not wholesale; a following expansion still sees its own tables
    rng = np.random.default_rng(2)
    n = 20000
    pts = np.column_stack([rng.random(n) * 1000, rng.normal(0, 2.0, n)])
    pts[rng.random(n) < 0.3, 1] = 50.0                                       # gross outliers
    model, T2, lam = np.array([0.0, 1.0, 0.0]), 9.0, 0.14
    gpu_ctx.set_points(_lib.LINE2D, pts)
    graph = gpu_ctx.graph_build(np.column_stack([pts[:, 0], np.zeros(n)]), _lib.GRAPH_KNN, k=6)   # neighbours along x
    Dq = (rng.integers(0, 1 << 20, (n, 3)) << 12).astype(np.int64)
    gpu_ctx.set_unary_q(Dq)
    gpu_ctx.set_labels(np.zeros(n, np.int32))
    flags = gpu_ctx.gc_labeling(model, T2, lam)
    assert np.array_equal(flags, oracle.gc_labeling(_lib.LINE2D, pts, model, T2, lam, graph))
    thr_inl = pts[:, 1] ** 2 <= T2
    assert 0 < int((flags != thr_inl).sum()) < n // 4
    gpu_ctx.expansion(0.2, 0.5)
    ref, _, _ = oracle.expansion(Dq, graph, oracle.quantize_lambda(0.2), oracle.quantize(0.5), np.zeros(n, np.int32))
    assert np.array_equal(gpu_ctx.get_labels(), ref)


def test_gc_labeling_error_paths(gpu_ctx):
    from pyprogressivex._lib import PgxError
    pts = np.random.default_rng(0).random((100, 2))
    gpu_ctx.set_points(_lib.LINE2D, pts)
    gpu_ctx.graph_build(pts, _lib.GRAPH_KNN, k=3)
    for lam in (0.0, 1.0, -0.1, float("nan")):
        with pytest.raises(PgxError):
            gpu_ctx.gc_labeling(np.array([0.0, 1.0, 0.0]), 1.0, lam)
    with pytest.raises(PgxError):
        gpu_ctx.gc_labeling(np.array([0.0, 1.0, 0.0]), 0.0, 0.5)
    gpu_ctx.set_points(_lib.LINE2D, pts[:50])            # graph over 100 sites no longer matches
    with pytest.raises(PgxError):
        gpu_ctx.gc_labeling(np.array([0.0, 1.0, 0.0]), 1.0, 0.5)


# ----------------------------------------------------------------------------------------------------------------------
# full BASELINE size: 1e6 2D-3D correspondences x 2048 pose hypotheses, size-independent properties
# ----------------------------------------------------------------------------------------------------------------------
def test_metric_batch_full_size_properties(gpu_ctx, oracle):
    from pyprogressivex import datasets
    x1, x2, K, gt_labels, gt_poses = datasets.make_poses()
    pts, f = datasets.normalize_pnp(x1, x2, K)
    hyps = datasets.make_pose_hypotheses(gt_poses, M=2048)
    thr = 4.0 / f
    T2 = 2.25 * thr * thr
    n = pts.shape[0]
    assert n == 1000000 and hyps.shape == (2048, 12)
    gpu_ctx.set_points(3, pts)
    comp = np.zeros(n)
    comp[:50000] = 0.5
    gpu_ctx.set_compound(comp)
    full = gpu_ctx.score(hyps, T2, has_compound=True, exponent=2)
    # (1) the 16 ground-truth poses recover (almost all of) their 5e4 inliers
    assert np.all(full["counts"][:16] > 45000)
    # (2) oracle on a bounded sample of hypotheses over ALL points: counts exact, sums to 1e-9
    sample = np.array([0, 5, 15, 16, 100, 2047])
    ref = oracle.score(3, pts, hyps[sample], T2, compound=comp, has_compound=True, exponent=2)
    assert np.array_equal(full["counts"][sample], ref["counts"])
    assert _rel(full["values"][sample], ref["values"]) <= REL
    assert _rel(full["shared"][sample], ref["shared"]) <= REL
    # (3) additivity over a split of the points (integer counts add exactly)
    half = n // 2
    gpu_ctx.set_points(3, pts[:half])
    a = gpu_ctx.score(hyps, T2)
    gpu_ctx.set_points(3, pts[half:])
    b = gpu_ctx.score(hyps, T2)
    assert np.array_equal(a["counts"] + b["counts"], full["counts"])
    assert _rel(a["values"] + b["values"], full["values"]) <= 1e-9
    # (4) masks of a 64-hypothesis slice agree with the counters and with the oracle's mask for one of them
    gpu_ctx.set_points(3, pts)
    m64 = gpu_ctx.score(hyps[:64], T2, want_masks=True)
    pop = np.unpackbits(m64["masks"].view(np.uint8), axis=1).sum(axis=1)
    assert np.array_equal(pop, m64["counts"])
    refm = oracle.score(3, pts, hyps[3:4], T2, want_masks=True)
    assert np.array_equal(refm["masks"][0], m64["masks"][3])


# ----------------------------------------------------------------------------------------------------------------------
# RCCL plumbing with a single rank (the 1-GPU box cannot run N > 1; the N > 1 host logic is covered by the gloo test)
# ----------------------------------------------------------------------------------------------------------------------
def test_rccl_single_rank_allgather(gpu_ctx):
    from pyprogressivex import _lib, parallel
    mt, pts, models, thr = make_case("pnp", 5000, 100, seed=2)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.comm_init(1, 0, _lib.comm_unique_id())
    try:
        gpu_ctx.comm_barrier()
        assert gpu_ctx.comm_allreduce_max(3.25) == 3.25
        table = parallel.score_sharded(parallel.RcclExchange(gpu_ctx), models, T2)
        gpu_ctx.score_upload(models)
        gpu_ctx.score_launch(T2)
        gpu_ctx.score_allgather()
        allg = gpu_ctx.score_fetch_all()
        direct = gpu_ctx.score_fetch()
        for k in ("counts", "values", "shared", "scores"):
            assert np.array_equal(allg[k], direct[k]) and np.array_equal(table[k], direct[k])
        gpu_ctx.set_compound(np.linspace(0, 1, 5000))
        gpu_ctx.compound_allreduce_max()
        assert np.array_equal(gpu_ctx.get_compound(), np.linspace(0, 1, 5000))
    finally:
        gpu_ctx.comm_destroy()


@pytest.mark.parametrize("n", [1023, 1024, 1025, 2049, 8191, 8192, 8193])
def test_inlier_indices_at_the_one_launch_compaction_limits(gpu_ctx, oracle, n):
    """Up to 8 192 points the inlier indices come from one launch (pointwise.hip compact_small_kernel: a run of <= 8 consecutive items per
    thread, 1 024 threads), beyond from hipcub's select: the same ascending sets at the sizes where the run length and the path change,
    for the scorer's mask rows and for the cut's flags (the cut's compaction rides behind the one-workgroup move)."""
    mt, pts, models, thr = make_case("line", n, 3, seed=n)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.score(models, T2, want_masks=True)
    for row in range(len(models)):
        with np.errstate(invalid="ignore"):
            assert np.array_equal(gpu_ctx.score_inliers(row), np.flatnonzero(oracle.squared_residuals(mt, pts, models[row]) < T2))
    graph = gpu_ctx.graph_build(pts, _lib.GRAPH_KNN, k=4)
    for m in models:
        assert np.array_equal(gpu_ctx.gc_inliers(m, T2, 0.2), np.flatnonzero(oracle.gc_labeling(mt, pts, m, T2, 0.2, graph)))


@pytest.mark.parametrize("name", ["pnp", "homography", "line"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 5000, 100003])
def test_score_inliers_are_the_mask_row(gpu_ctx, oracle, name, n):
    """pgx_score_inliers: the inlier vector of getScore (scoring_function_with_compound_model.h:88) as ascending indices,
    compacted on the device = the mask row unpacked on the host = the oracle's strict r^2 < T^2 set."""
    from pyprogressivex._proposal import mask_to_indices
    mt, pts, models, thr = make_case(name, n, 3, seed=n + 1)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    got = gpu_ctx.score(models, T2, want_masks=True)
    for row in range(len(models)):
        idx = gpu_ctx.score_inliers(row)
        assert np.array_equal(idx, mask_to_indices(got["masks"][row], n))
        with np.errstate(invalid="ignore"):
            assert np.array_equal(idx, np.flatnonzero(oracle.squared_residuals(mt, pts, models[row]) < T2))
    with pytest.raises(_lib.PgxError, match="row"):
        gpu_ctx.score_inliers(len(models))
    gpu_ctx.score(models, T2)
    with pytest.raises(_lib.PgxError, match="no masks"):
        gpu_ctx.score_inliers(0)


def test_first_cycle_memo_is_transparent(monkeypatch):
    """pgx_expansion keeps the labels after every first-cycle move of an expansion from the all-zero labelling and restores the
    state behind the leading moves whose unary columns are unchanged (PEARL re-runs such expansions with mostly the same
    models).  The memo must be invisible: energy, cycles and labels of a sequence of expansions with a growing / partly refitted
    model list equal those of a context with the memo switched off - and the oracle's, on the last one."""
    import pgx_oracle as O
    from pyprogressivex import _lib
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=3000, n_objects=6, n_outliers=6000, seed=3)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    n = pts.shape[0]
    lam, h, thr = 0.1, 6.0, 4.0 / f
    raw = np.column_stack([x1, x2])
    rng = np.random.default_rng(5)
    jig = lambda P: P + rng.normal(0, 1e-4, P.shape)          # a refit: the model moves a little
    lists = [poses[:3], np.vstack([poses[:3], poses[3:4]]),                       # one model appended: 3 leading columns kept
             np.vstack([poses[:2], jig(poses[2:3]), poses[3:4], poses[4:5]]),     # column 2 refitted: only 2 kept
             np.vstack([poses[:2], jig(poses[2:3]), poses[3:4], poses[4:5]])[[0, 1, 2, 3, 4]],
             poses[:6]]
    lists[3] = lists[2].copy()                                                    # identical list: every first-cycle move restored

    def run(memo):
        monkeypatch.setenv("PGX_MF_MEMO", "1" if memo else "0")
        ctx = _lib.Context(0)
        out = []
        try:
            ctx.set_points(_lib.PNP, pts)
            ctx.graph_build(raw, _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
            for models in lists:
                ctx.pearl_unary(models, thr, lam)
                ctx.set_labels(np.zeros(n, np.int32))
                eq, e, cycles = ctx.expansion(lam, h)
                out.append((eq, cycles, ctx.get_labels()))
            hits = ctx.expansion_paths()["memo"]
        finally:
            ctx.close()
        return out, hits
    # a labelling written by anything but pgx_set_labels(zeros) is not a memo start: set zeros, let the greedy solver relabel, expand
    c = _lib.Context(0)
    try:
        c.set_points(_lib.PNP, pts)
        c.graph_build(raw, _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
        Dq = c.pearl_unary(lists[0], thr, lam, want_table=True)
        c.set_labels(np.zeros(n, np.int32))
        c.expansion(lam, h)                                   # fills the memo for these columns
        c.set_labels(np.zeros(n, np.int32))
        c.greedy_labeling(h)                                  # relabels: the next expansion does NOT start from zeros
        start = c.get_labels()
        eq, e, cyc = c.expansion(lam, h)
        ref, ref_e, ref_cyc = O.expansion(Dq, O.graph_build(raw, 0, radius=20.0, k=5), O.quantize_lambda(lam), O.quantize(h), start)
        assert np.array_equal(c.get_labels(), ref) and eq == ref_e and cyc == ref_cyc
    finally:
        c.close()
    with_memo, hits = run(True)
    without, none = run(False)
    assert none == 0 and hits >= 3 + 2 + 6, hits
    for (eq, cycles, lab), (eq0, cycles0, lab0) in zip(with_memo, without):
        assert eq == eq0 and cycles == cycles0 and np.array_equal(lab, lab0)
    Dq = O.unary_q(O.PNP, pts, lists[-1], thr, lam)
    graph = O.graph_build(raw, 0, radius=20.0, k=5)
    ref, ref_e, ref_cyc = O.expansion(Dq, graph, O.quantize_lambda(lam), O.quantize(h), np.zeros(n, np.int32))
    assert np.array_equal(with_memo[-1][2], ref) and with_memo[-1][0] == ref_e and with_memo[-1][1] == ref_cyc


def test_point_sharded_accumulators_are_exact(gpu_ctx, oracle):
    """pgx_score_allreduce adds the integer accumulators of ranks that hold different points.  On one GPU: the accumulators of
    the whole point set (a) ARE the oracle's per-inlier terms in the path's fixed point, integer for integer, and (b) equal
    the sum of the accumulators of three slices scored with the job's scale (pgx_score_set_global_n): what the ranks of a
    point-sharded job all-reduce is bitwise the single-GPU table."""
    from helpers import fixed_point_accumulators
    from pyprogressivex import parallel
    for name in ("pnp", "fundamental", "vanishing_point"):
        mt, pts, models, thr = make_case(name, 20011, 96, seed=9)
        T2 = 2.25 * thr * thr
        n = pts.shape[0]
        comp = np.random.default_rng(3).uniform(0, 1, n)
        gpu_ctx.score_set_global_n(0)
        gpu_ctx.set_points(mt, pts)
        gpu_ctx.set_compound(comp)
        full = gpu_ctx.score(models, T2, has_compound=True, exponent=2)
        acc = gpu_ctx.score_accumulators()
        ref = fixed_point_accumulators(oracle, mt, pts, models, T2, comp)
        for k in ("counts", "values_q", "shared_q"):
            assert np.array_equal(acc[k].astype(np.int64), ref[k]), (name, k)
        total = {k: np.zeros(len(models), np.uint64) for k in acc}
        try:
            for r in range(3):
                lo, hi = parallel.point_slice(n, 3, r)
                gpu_ctx.set_points(mt, pts[lo:hi])
                gpu_ctx.score_set_global_n(n)
                gpu_ctx.set_compound(comp[lo:hi])
                gpu_ctx.score(models, T2, has_compound=True, exponent=2)
                part = gpu_ctx.score_accumulators()
                for k in total:
                    total[k] += part[k]
        finally:
            gpu_ctx.score_set_global_n(0)
        for k in total:
            assert np.array_equal(total[k], acc[k]), (name, k)
        table = parallel.table_from_accumulators(total["counts"], total["values_q"], total["shared_q"], n, True, 2)
        for k in ("counts", "values", "shared"):
            assert np.array_equal(table[k], full[k]), (name, k)


def test_rccl_single_rank_allreduce(gpu_ctx):
    """the point-sharded exchange with RCCL on the path (one rank: the all-reduce is the identity): serial and pipelined forms"""
    from pyprogressivex import _lib, parallel
    mt, pts, models, thr = make_case("pnp", 20011, 300, seed=2)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.set_compound(np.linspace(0, 1, pts.shape[0]))
    direct = gpu_ctx.score(models, T2, has_compound=True, exponent=2)
    gpu_ctx.comm_init(1, 0, _lib.comm_unique_id())
    try:
        gpu_ctx.force_comm = True
        ex = parallel.RcclExchange(gpu_ctx)
        for pieces in (1, 2, 5):
            table = parallel.score_point_sharded(ex, models, T2, has_compound=True, exponent=2, pieces=pieces)
            for k in ("counts", "values", "shared", "scores"):
                assert np.array_equal(table[k], direct[k]), (pieces, k)
        # a collective on the context's stream while a pipelined exchange is in flight is refused, not reordered
        gpu_ctx.score_upload(models)
        gpu_ctx.score_launch(T2, has_compound=True)
        gpu_ctx.score_allreduce_begin(0)
        with pytest.raises(_lib.PgxError, match="in flight"):
            gpu_ctx.comm_barrier()
        got = gpu_ctx.score_allreduce_end(0, 2)
        assert np.array_equal(got["scores"], direct["scores"])
        gpu_ctx.comm_barrier()
    finally:
        gpu_ctx.force_comm = False
        gpu_ctx.comm_destroy()


# ----------------------------------------------------------------------------------------------------------------------
# error paths of the C ABI (status codes + messages, never a crash) and a BASELINE-size labelling move
# ----------------------------------------------------------------------------------------------------------------------
def test_abi_error_paths():
    from pyprogressivex import _lib
    ctx = _lib.Context(0)
    try:
        with pytest.raises(_lib.PgxError, match="points not set"):
            ctx.model_type = 0
            ctx.score(np.zeros((1, 3)), 1.0)
        with pytest.raises(ValueError):
            ctx.set_points(0, np.zeros((4, 3)))                      # wrong point dimension for lines
        with pytest.raises(_lib.PgxError, match="empty input"):
            ctx.set_points(0, np.zeros((0, 2)))
        ctx.set_points(0, np.random.default_rng(0).random((100, 2)))
        with pytest.raises(_lib.PgxError, match="unary table not set"):
            ctx.set_labels(np.zeros(100, np.int32))
            ctx.expansion(0.1, 1.0)
        ctx.pearl_unary(np.array([[1.0, 0.0, -0.5]]), 0.1, 0.2)
        with pytest.raises(_lib.PgxError, match="no graph set|needs a graph"):
            ctx.expansion(0.2, 1.0)                                   # lambda > 0 without pgx_set_graph
        with pytest.raises(_lib.PgxError, match="out of range"):
            ctx.expand_alpha(0.0, 1.0, 7)
        with pytest.raises(_lib.PgxError, match="fixed-point range"):
            ctx.energy(0.0, 1e12)                                     # label cost beyond the 2^30 budget
        with pytest.raises(_lib.PgxError, match="fixed-point range"):
            big = np.zeros((100, 2), np.int64)
            big[0, 0] = 1 << 61                                       # an injected table whose sums would not fit int64
            ctx.set_unary_q(big)
            ctx.set_labels(np.zeros(100, np.int32))
            ctx.energy(0.0, 0.0)
        ctx.pearl_unary(np.array([[1.0, 0.0, -0.5]]), 0.1, 0.2)
        with pytest.raises(_lib.PgxError, match="slot"):
            ctx.compound_update([5])
        with pytest.raises(_lib.PgxError, match="at most"):
            ctx.set_unary_q(np.zeros((10, 65), np.int64))
            ctx.set_labels(np.zeros(10, np.int32))
            ctx.expand_alpha(0.0, 0.0, 0)
        with pytest.raises(_lib.PgxError):
            ctx.comm_barrier()                                        # communicator not initialised
        with pytest.raises(_lib.PgxError, match="communicator not initialised"):
            ctx.score_allreduce()
        with pytest.raises(_lib.PgxError, match="negative point count"):
            ctx.score_set_global_n(-1)
        with pytest.raises(_lib.PgxError, match="negative label"):
            ctx.set_labels(np.array([0, -1, 0], np.int32))
        ctx.set_unary_q(np.zeros((100, 2), np.int64))
        ctx.set_labels(np.full(100, 2, np.int32))                     # label 2 of a 2-label table: indexes per-label tables out of range
        with pytest.raises(_lib.PgxError, match="out of range"):
            ctx.expansion(0.0, 1.0)
        with pytest.raises(_lib.PgxError, match="out of range"):
            ctx.energy(0.0, 1.0)
    finally:
        ctx.close()


def test_single_move_at_c5_size_matches_oracle(gpu_ctx, oracle):
    """BASELINE config C5 shape: 2e5 line segments, 6 vanishing points, k-NN(8) graph on the midpoints; one expansion
    move from a noisy labelling must reproduce the oracle's min-cut labels exactly (the full expansion is covered at
    smaller sizes; Dinic needs ~17 s for it here)."""
    import host_graph as _graph
    from pyprogressivex import datasets
    pts, gt, vps = datasets.make_vanishing_points(seed=0)
    n = pts.shape[0]
    graph = _graph.knn_graph(0.5 * (pts[:, :2] + pts[:, 2:]), 8)
    lam, h, thr = 0.1, 20.0, 1.5
    gpu_ctx.set_points(4, pts)
    Dq = gpu_ctx.pearl_unary(vps, thr, lam, want_table=True)
    assert np.array_equal(Dq, oracle.unary_q(4, pts, vps, thr, lam))
    labels = np.where(gt == 0, 6, gt - 1).astype(np.int32)
    flip = np.random.default_rng(1).random(n) < 0.3
    labels[flip] = np.random.default_rng(2).integers(0, 7, int(flip.sum()))
    gpu_ctx.set_graph(*graph)
    gpu_ctx.set_labels(labels)
    lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
    assert gpu_ctx.energy(lam, h)[0] == oracle.energy(Dq, graph, lq, hq, labels)
    for alpha in (2, 6):
        ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, labels)
        assert gpu_ctx.expand_alpha(lam, h, alpha) == ref_changed
        labels = gpu_ctx.get_labels()
        assert np.array_equal(labels, ref)


def test_expansion_at_c4_size_is_schedule_invariant(gpu_ctx, monkeypatch):
    """BASELINE config C4 shape (1e6 sites, 10 labels, 6.1 M arcs): the oracle's Dinic needs minutes here, so the full
    size is covered by properties the fixed-point construction guarantees: the minimal sink side of every move is
    unique, so the labels must not depend on the max-flow schedule (work lists, wave pass, the gate for an unused
    label); a second expansion from the optimum changes nothing; the reported energy is the energy of the labels;
    the resident graph of pgx_graph_build and the Gram pass agree with themselves across selections."""
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    n = pts.shape[0]
    lam, h = 0.1, 6.0
    gpu_ctx.set_points(_lib.PNP, pts)
    arcs = gpu_ctx.graph_build(np.column_stack([x1, x2]), _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
    assert arcs > 4 * n
    gpu_ctx.pearl_unary(poses[:9], 4.0 / f, lam)
    results = []
    gpu_ctx.set_labels(np.zeros(n, np.int32))
    eq, e, cycles = gpu_ctx.expansion(lam, h)
    results.append((eq, cycles, gpu_ctx.get_labels()))
    assert gpu_ctx.energy(lam, h)[0] == eq
    for env in ({"PGX_MF_REGION": "0"},):     # level-synchronous launches only (the default context mixes in region moves)
        for key in ("PGX_MF_TILE", "PGX_MF_REGION"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        other = _lib.Context(0)
        try:
            other.set_points(_lib.PNP, pts)
            other.graph_build(np.column_stack([x1, x2]), _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
            other.pearl_unary(poses[:9], 4.0 / f, lam)
            other.set_labels(np.zeros(n, np.int32))
            eq, e, cycles = other.expansion(lam, h)
            results.append((eq, cycles, other.get_labels()))
            assert other.energy(lam, h)[0] == eq
        finally:
            other.close()
    for eq, cycles, labels in results[1:]:
        assert eq == results[0][0] and cycles == results[0][1] and np.array_equal(labels, results[0][2])
    eq2, _, cycles2 = gpu_ctx.expansion(lam, h)            # idempotent at the optimum
    assert eq2 == results[-1][0] and cycles2 == 1 and np.array_equal(gpu_ctx.get_labels(), results[0][2])
    # ground truth sanity: the first nine objects are recovered almost everywhere
    lab = results[0][2]
    agree = np.mean(lab[(gt >= 1) & (gt <= 9)] == gt[(gt >= 1) & (gt <= 9)] - 1)
    assert agree > 0.97
    # Gram pass: label selection == index selection of the same sites (fixed reduction tree => identical up to the
    # different partial grouping; both against the same matrix scale)
    sel = np.nonzero(lab == 3)[0]
    Ga, ca, _ = gpu_ctx.gram(_lib.GRAM_PNP_GN, ("label", 3), params=poses[3])
    Gb, cb, _ = gpu_ctx.gram(_lib.GRAM_PNP_GN, ("index", sel), params=poses[3])
    assert ca == cb == len(sel) and np.abs(Ga - Gb).max() <= REL * np.abs(Ga).max()


# ----------------------------------------------------------------------------------------------------------------------
# U-8: the lambda = 0 labelling GCO-v3 takes (greedy facility location over the label costs / per-site argmin)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,L", [(1, 2), (64, 3), (255, 2), (4097, 5), (20011, 11), (100003, 9)])
@pytest.mark.parametrize("h", [0.0, 3.0, 40.0])
def test_greedy_labeling_matches_oracle(gpu_ctx, oracle, n, L, h):
    rng = np.random.default_rng(n + L)
    D = rng.integers(0, 2 << 32, (n, L)).astype(np.int64)
    cheap = rng.integers(0, L, n)
    D[np.arange(n), cheap] >>= 6                                   # clusters
    D[:, L - 1] = 1 << 32                                           # the outlier label's constant cost
    if n > 64:
        D[:64] = (D[:64] >> 30) << 30                               # ties
    gpu_ctx.set_unary_q(D)
    gpu_ctx.set_labels(np.full(n, L - 1, np.int32))                 # ignored by the heuristic
    eq, e, opened = gpu_ctx.greedy_labeling(h)
    ref_labels, ref_e, ref_opened = oracle.greedy_labeling(D, oracle.quantize(h))
    assert np.array_equal(gpu_ctx.get_labels(), ref_labels) and eq == ref_e
    assert opened == (ref_opened if h > 0 else len(set(ref_labels.tolist())))
    assert gpu_ctx.energy(0.0, h)[0] == eq


def test_greedy_labeling_on_a_real_unary_table_and_error_paths(gpu_ctx, oracle):
    mt, pts, models, thr = make_case("homography", 20011, 3, seed=3)
    gpu_ctx.set_points(mt, pts)
    Dq = gpu_ctx.pearl_unary(models, thr, 0.0, want_table=True)
    eq, e, opened = gpu_ctx.greedy_labeling(10.0)
    ref_labels, ref_e, ref_opened = oracle.greedy_labeling(Dq, oracle.quantize(10.0))
    assert np.array_equal(gpu_ctx.get_labels(), ref_labels) and eq == ref_e and opened == ref_opened == 4
    gpu_ctx.set_labels(np.zeros(len(pts), np.int32))
    eq2, _, _ = gpu_ctx.expansion(0.0, 10.0)
    print(f"lambda = 0 labelling energies: greedy {eq / 2 ** 32:.3f}, alpha-expansion from zeros {eq2 / 2 ** 32:.3f}")
    with pytest.raises(_lib.PgxError, match="fixed-point range"):
        gpu_ctx.set_unary_q(np.full((10, 2), 1 << 40, np.int64))
        gpu_ctx.greedy_labeling(1.0)
    with pytest.raises(_lib.PgxError, match="negative cost"):
        gpu_ctx.set_unary_q(np.full((10, 2), -1, np.int64))


# ----------------------------------------------------------------------------------------------------------------------
# pgx_set_points on the device (setpoints.hip) == the host preprocessing of round 1, bit for bit
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n", [("pnp", 1), ("pnp", 63), ("pnp", 64), ("pnp", 5000), ("pnp", 100003), ("homography", 513),
                                    ("homography", 20011)])
def test_set_points_device_equals_host_preprocessing(name, n, monkeypatch, oracle):
    mt, pts, models, thr = make_case(name, n, 70, seed=n)
    if n > 1000:
        pts[7] = pts[3]                       # exact duplicates: the stable sort must keep them in index order
        pts[11, 0] = pts[:, 0].max() * 3.0    # a far point: the quantisation range
    T2 = 2.25 * thr * thr
    got = {}
    monkeypatch.setenv("PGX_SP_KD", "0")        # the host path orders by the Morton key: compare like with like (k-d order: next test)
    for mode in ("0", "1"):
        monkeypatch.setenv("PGX_SETPOINTS_HOST", mode)
        ctx = _lib.Context(0)
        try:
            ctx.set_points(mt, pts)
            got[mode] = {k: ctx.score_debug_fetch(k) for k in ("order", "bounds", "rows64", "rows32", "rows32_sorted")}
            got[mode]["score"] = ctx.score(models, T2, want_masks=True)
        finally:
            ctx.close()
    monkeypatch.delenv("PGX_SETPOINTS_HOST")
    for k in ("order", "bounds", "rows64", "rows32", "rows32_sorted"):
        assert np.array_equal(got["0"][k], got["1"][k]), k
    for k in ("counts", "values", "masks"):
        assert np.array_equal(got["0"]["score"][k], got["1"]["score"][k]), k
    ref = oracle.score(mt, pts, models, T2, want_masks=True)
    assert np.array_equal(got["0"]["score"]["counts"], ref["counts"]) and np.array_equal(got["0"]["score"]["masks"], ref["masks"])


@pytest.mark.parametrize("n", [129, 130, 4097, 5000, 100003])
def test_kd_order_of_pose_points_is_a_valid_grouping(n, monkeypatch, oracle):
    """The k-d order of a pose problem's points (setpoints.hip, the default) is another permutation under the same group
    bounds: a permutation of all points, duplicates and a far outlier included, and scores, counts and masks bitwise those of
    the Morton order and equal to the oracle's - with no more surviving (hypothesis, group) steps at the larger sizes."""
    mt, pts, models, thr = make_case("pnp", n, 70, seed=n)
    pts[7] = pts[3]
    pts[11, 0] = pts[:, 0].max() * 3.0
    T2 = 2.25 * thr * thr
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PGX_SP_KD", mode)
        ctx = _lib.Context(0)
        try:
            ctx.set_points(mt, pts)
            got[mode] = {"order": ctx.score_debug_fetch("order"), "score": ctx.score(models, T2, want_masks=True),
                         "steps": ctx.score_stats(T2)["surviving_group_steps"]}
        finally:
            ctx.close()
    monkeypatch.delenv("PGX_SP_KD")
    assert np.array_equal(np.sort(got["1"]["order"]), np.arange(n))
    assert not np.array_equal(got["0"]["order"], got["1"]["order"])
    for k in ("counts", "values", "masks"):
        assert np.array_equal(got["0"]["score"][k], got["1"]["score"][k]), k
    ref = oracle.score(mt, pts, models, T2, want_masks=True)
    assert np.array_equal(got["1"]["score"]["counts"], ref["counts"]) and np.array_equal(got["1"]["score"]["masks"], ref["masks"])
    if n >= 100000:
        assert got["1"]["steps"] <= got["0"]["steps"]


def test_set_points_device_non_finite_points_disable_the_sorted_path(oracle):
    mt, pts, models, thr = make_case("pnp", 3000, 40, seed=9)
    T2 = 2.25 * thr * thr
    for bad in (np.nan, np.inf):
        q = pts.copy()
        q[17, 3] = bad
        ctx = _lib.Context(0)
        try:
            ctx.set_points(mt, q)
            with pytest.raises(_lib.PgxError, match="no sorted copies"):
                ctx.score_debug_fetch("order")
            got = ctx.score(models, T2)
        finally:
            ctx.close()
        ref = oracle.score(mt, q, models, T2)
        assert np.array_equal(got["counts"], ref["counts"])


# ----------------------------------------------------------------------------------------------------------------------
# the point-sharded exchange never skips its collective, stale accumulators are refused, labels_max follows the moves
# ----------------------------------------------------------------------------------------------------------------------
def test_allreduce_raises_on_every_rank_instead_of_skipping_the_collective(gpu_ctx):
    """A rank whose launch did not take the integer-accumulator path (here: a slice with an infinite coordinate, which switches the
    sorted copies and the f32 filter off) used to return an error BEFORE ncclAllReduce while its peers waited in it for ever.
    Now every rank enters the reduction - this one with a zero block and a poison word - and the error is raised after it, in the
    serial and in the pipelined form; a healthy launch afterwards reduces normally."""
    from pyprogressivex import _lib
    mt, pts, models, thr = make_case("pnp", 20011, 200, seed=3)
    T2 = 2.25 * thr * thr
    bad = pts.copy()
    bad[17, 2] = np.inf
    gpu_ctx.comm_init(1, 0, _lib.comm_unique_id())
    try:
        gpu_ctx.set_points(mt, bad)
        gpu_ctx.score_upload(models)
        gpu_ctx.score_launch(T2, has_compound=False)
        with pytest.raises(_lib.PgxError, match="could not take the group-major path.*this rank is one of them"):
            gpu_ctx.score_allreduce()
        gpu_ctx.score_launch(T2, has_compound=False)
        gpu_ctx.score_allreduce_begin(0)                               # enqueued: the collective runs
        with pytest.raises(_lib.PgxError, match="could not take the group-major path"):
            gpu_ctx.score_allreduce_end(0, 2)
        gpu_ctx.comm_barrier()                                          # the slot was released: the communicator is usable
        gpu_ctx.set_points(mt, pts)
        direct = gpu_ctx.score(models, T2, has_compound=False, exponent=2)
        gpu_ctx.score_launch(T2, has_compound=False)
        gpu_ctx.score_allreduce()
        got = gpu_ctx.score_fetch(2)
        assert np.array_equal(got["counts"], direct["counts"]) and np.array_equal(got["values"], direct["values"])
        # accumulators belong to the batch they were launched for: an upload in between makes them stale
        gpu_ctx.score_launch(T2, has_compound=False)
        gpu_ctx.score_upload(models[:100])
        with pytest.raises(_lib.PgxError, match="could not take the group-major path"):
            gpu_ctx.score_allreduce()
    finally:
        gpu_ctx.comm_destroy()


def test_labels_written_by_moves_are_range_checked_against_a_smaller_table(gpu_ctx):
    """labels_max used to follow pgx_set_labels only: after moves on a 4-label table wrote label 3, a warm start on a 3-label table
    passed the guard and indexed the per-label tables out of range"""
    from pyprogressivex import _lib
    pts, gt, lines = datasets_lines()
    gpu_ctx.set_points(_lib.LINE2D, pts)
    three = np.asarray(lines, dtype=np.float64).reshape(3, 3)
    gpu_ctx.pearl_unary(three, 2.0, 0.0)                                # L = 4
    gpu_ctx.set_labels(np.zeros(len(pts), np.int32))
    gpu_ctx.expansion(0.0, 5.0)                                         # writes labels up to 3
    assert int(gpu_ctx.get_labels().max()) == 3
    gpu_ctx.pearl_unary(three[:2], 2.0, 0.0)                            # L = 3, labels kept (a warm start)
    with pytest.raises(_lib.PgxError, match="out of range"):
        gpu_ctx.expansion(0.0, 5.0)
    with pytest.raises(_lib.PgxError, match="out of range"):
        gpu_ctx.energy(0.0, 5.0)


def datasets_lines():
    from pyprogressivex import datasets
    return datasets.make_lines(n_per_line=300, n_lines=3, n_outliers=300, seed=2)


def test_identical_expansion_call_is_answered_from_the_fixed_point(monkeypatch):
    """PEARL ends every run with a labelling call whose models and warm start are those of the call before (the iteration that finds
    "nothing changed", PEARL.h:429-467).  pgx_expansion answers such a call - same unary columns, weights, graph, and the labels the
    last expansion left at a fixed point - without running it: one verifying cycle of no-ops, the same energy.  Must be invisible:
    equal to a context with the memo off and to the oracle; anything that touches labels, columns or weights in between runs for real."""
    import pgx_oracle as O
    from pyprogressivex import _lib
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=2500, n_objects=4, n_outliers=4000, seed=8)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    n = pts.shape[0]
    lam, h, thr = 0.1, 6.0, 4.0 / f
    raw = np.column_stack([x1, x2])
    graph = O.graph_build(raw, 0, radius=20.0, k=5)

    def run(memo):
        monkeypatch.setenv("PGX_MF_MEMO", "1" if memo else "0")
        ctx = _lib.Context(0)
        out = []
        try:
            ctx.set_points(_lib.PNP, pts)
            ctx.graph_build(raw, _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)
            Dq = ctx.pearl_unary(poses[:4], thr, lam, want_table=True)
            ctx.set_labels(np.zeros(n, np.int32))
            out.append(ctx.expansion(lam, h) + (ctx.get_labels(),))                  # 0: from zeros
            hits0 = ctx.expansion_paths()["memo"]
            ctx.pearl_unary(poses[:4], thr, lam)                                      # the same models again, labels untouched
            out.append(ctx.expansion(lam, h) + (ctx.get_labels(),))                  # 1: the identical call
            hits1 = ctx.expansion_paths()["memo"]
            mincuts1 = ctx.expansion_stats()["mincuts"]
            out.append(ctx.expansion(lam, 7.0) + (ctx.get_labels(),))                # 2: another label cost: runs
            ctx.pearl_unary(poses[:4], thr, lam)
            ctx.set_labels(out[1][3])                                                 # the caller's labels (same content): runs
            out.append(ctx.expansion(lam, h) + (ctx.get_labels(),))                  # 3
            mincuts3 = ctx.expansion_stats()["mincuts"] + ctx.expansion_stats().get("skipped_moves", 0)
            ctx.pearl_unary(np.vstack([poses[:3], poses[3:4] + 1e-6]), thr, lam)      # a refitted model: runs
            out.append(ctx.expansion(lam, h) + (ctx.get_labels(),))                  # 4
        finally:
            ctx.close()
        return out, Dq, (hits0, hits1, mincuts1, mincuts3)
    with_memo, Dq, (hits0, hits1, mincuts1, mincuts3) = run(True)
    without, _, (w0, w1, wm1, _) = run(False)
    for k, (a, b) in enumerate(zip(with_memo, without)):
        assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[3], b[3]), k
    assert hits1 - hits0 == 5 and mincuts1 == 0                  # the identical call solved no min-cut (5 labels = 5 moves answered)
    assert w1 == w0 and wm1 > 0                                  # ... and with the memo off it did
    assert mincuts3 > 0
    assert with_memo[1][2] == 1 and with_memo[1][0] == with_memo[0][0] and np.array_equal(with_memo[1][3], with_memo[0][3])
    ref, ref_e, ref_cyc = O.expansion(Dq, graph, O.quantize_lambda(lam), O.quantize(h), with_memo[0][3])    # the oracle, from the fixed point
    assert np.array_equal(ref, with_memo[1][3]) and ref_e == with_memo[1][0] and ref_cyc == 1


def test_graph_fetch_returns_the_resident_graph(gpu_ctx, oracle):
    """Context.graph_fetch (bench.py hands the device-built graph to the CPU labelling baseline): the CSR pgx_graph_build left resident"""
    from pyprogressivex import _lib
    rng = np.random.default_rng(3)
    pts = rng.random((5000, 2)) * 300
    arcs = gpu_ctx.graph_build(pts, _lib.GRAPH_KNN, k=6, fetch=False)
    got = gpu_ctx.graph_fetch()
    ref = oracle.graph_build(pts, 2, k=6)
    assert arcs == len(ref[1])
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_global_n_of_a_sharded_job_does_not_outlive_its_point_set(gpu_ctx):
    """pgx_score_set_global_n fixes the fixed-point scale of a point-sharded job; it used to persist across pgx_set_points, so an
    unrelated later problem on the same context summed at another scale.  After set_points the scale is the point set's own:
    the integer accumulators equal those of a context that never saw the sharded job."""
    mt, pts, models, thr = make_case("pnp", 9001, 150, seed=5)
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(mt, pts)
    gpu_ctx.score_upload(models)
    gpu_ctx.score_launch(T2, has_compound=False)
    clean = gpu_ctx.score_accumulators()
    gpu_ctx.score_set_global_n(50_000_000)
    gpu_ctx.score_launch(T2, has_compound=False)
    sharded = gpu_ctx.score_accumulators()
    assert np.array_equal(clean["counts"], sharded["counts"]) and not np.array_equal(clean["values_q"], sharded["values_q"])   # another scale while the job is declared
    gpu_ctx.set_points(mt, pts)                                 # a new problem: the declaration is gone
    gpu_ctx.score_upload(models)
    gpu_ctx.score_launch(T2, has_compound=False)
    again = gpu_ctx.score_accumulators()
    assert np.array_equal(again["counts"], clean["counts"]) and np.array_equal(again["values_q"], clean["values_q"])


@pytest.mark.gpu
def test_identical_call_shortcut_under_its_verifying_mode_and_across_every_label_writer(monkeypatch):
    """the shortcut's precondition is one counter, pgx_ctx::labels_version, bumped by every writer of the labels
    (pgx_set_labels, pgx_expand_alpha, the greedy labelling, pgx_expansion's own moves) and compared on a hit.  PGX_MF_DONE_VERIFY=1
    runs the real verifying cycle behind every hit and fails the call unless it relabels nothing at the same energy: with it on, a
    PEARL-like sequence must behave exactly as with it off, and a hit must never follow any of the writers."""
    import pgx_oracle as O
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=1500, n_objects=3, n_outliers=2500, seed=11)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    n = pts.shape[0]
    lam, h, thr = 0.1, 6.0, 4.0 / f
    raw = np.column_stack([x1, x2])

    def sequence(verify):
        monkeypatch.setenv("PGX_MF_DONE_VERIFY", "1" if verify else "0")
        monkeypatch.setenv("PGX_MF_MEMO", "1")
        ctx = _lib.Context(0)
        out = []
        try:
            ctx.set_points(_lib.PNP, pts)
            ctx.graph_build(raw, _lib.GRAPH_KNN_IN_BALL, radius=20.0, k=5, fetch=False)

            def again():
                ctx.pearl_unary(poses[:3], thr, lam)
                h0 = ctx.expansion_paths()["memo"]
                r = ctx.expansion(lam, h)
                out.append(r + (ctx.get_labels(), ctx.expansion_paths()["memo"] - h0))
            ctx.pearl_unary(poses[:3], thr, lam)
            ctx.set_labels(np.zeros(n, np.int32))
            out.append(ctx.expansion(lam, h) + (ctx.get_labels(), 0))
            again()                                        # 1: a hit
            ctx.expand_alpha(lam, h, 1)                    # a single move (changes nothing at the fixed point, but it is a writer)
            again()                                        # 2: must run
            again()                                        # 3: a hit again
            ctx.set_labels(out[-1][3])                     # the caller's labels, same content
            again()                                        # 4: must run
            ctx.greedy_labeling(h)                         # another writer
            ctx.pearl_unary(poses[:3], thr, lam)
            ctx.set_labels(out[1][3])
            again()                                        # 5: must run
        finally:
            ctx.close()
        return out
    plain, checked = sequence(False), sequence(True)
    for k, (a, b) in enumerate(zip(plain, checked)):
        assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[3], b[3]), k
    assert [r[4] > 0 for r in plain] == [False, True, False, True, False, False]      # hits exactly where nothing wrote the labels
    graph = O.graph_build(raw, 0, radius=20.0, k=5)
    Dq = O.unary_q(O.PNP, pts, poses[:3], thr, lam)
    ref_labels, ref_e, _ = O.expansion(Dq, graph, O.quantize_lambda(lam), O.quantize(h), np.zeros(n, np.int32))
    assert np.array_equal(plain[1][3], ref_labels) and plain[1][0] == ref_e


@pytest.mark.gpu
def test_graph_fetch_sizes_follow_the_resident_graph(gpu_ctx, oracle):
    """graph_fetch sized its buffers from the last graph_build; a larger graph set since (pgx_set_graph) overran them.
    The sizes now come from pgx_graph_size at the time of the fetch."""
    rng = np.random.default_rng(5)
    small = rng.random((300, 2)) * 100
    built = gpu_ctx.graph_build(small, _lib.GRAPH_KNN, k=3)
    assert gpu_ctx.graph_size() == (300, len(built[1]))
    big = oracle.graph_build(rng.random((5000, 2)) * 100, 2, k=8)
    gpu_ctx.set_graph(*big)
    assert gpu_ctx.graph_size() == (5000, len(big[1]))
    for a, b in zip(gpu_ctx.graph_fetch(), big):
        assert np.array_equal(a, b)
    again = gpu_ctx.graph_build(small, _lib.GRAPH_KNN, k=3)
    for a, b in zip(again, built):
        assert np.array_equal(a, b)


def test_device_jacobi_eigen_solver_is_bitwise_the_oracles(gpu_ctx, oracle):
    """pgx_eigh_smallest_batch (round 6: the refits' small dense solve behind the C ABI): one lane per matrix runs the cyclic Jacobi
    of the oracle's pgxo_eigh_smallest in the same operation order - eigenvectors and eigenvalues must be array_equal, for every
    size the refits use, badly scaled and degenerate matrices included (the LAPACK agreement of that algorithm, with its stated
    tolerance, is the CPU test tests/test_oracle.py)."""
    rng = np.random.default_rng(44)
    for q in (1, 2, 3, 6, 9):
        X = rng.standard_normal((257, 25, q))
        A = np.einsum("bni,bnj->bij", X, X)
        A[:30] *= 1e-12
        A[30:60] *= 1e12
        A[60:70] = 0.0
        A[70] = np.eye(q)
        A[71, 0, 0] = np.nan
        A[72:80, :, 0] *= 1e-5
        A[72:80, 0, :] *= 1e-5
        vec, val = gpu_ctx.eigh_smallest_batch(A)
        rvec, rval, _ = oracle.eigh_smallest(A)
        assert np.array_equal(vec, rvec, equal_nan=True) and np.array_equal(val, rval, equal_nan=True), q
    with pytest.raises(_lib.PgxError):
        gpu_ctx.eigh_smallest_batch(np.zeros((2, 10, 10)))


def test_refit_solver_jacobi_on_the_gpu_equals_the_cpu_restatement(monkeypatch):
    """the drop-in calls with refit_solver="jacobi": GPU (device Jacobi) against the same host code on the oracle context (the oracle's
    Jacobi) - labels equal, models to 1e-9 (the Gram sums differ by summation order, as with LAPACK) - and against the default solver"""
    import pyprogressivex as px
    from oracle_ctx import OracleContext
    from pyprogressivex import _api
    pts, gt, _ = datasets.make_homographies(n_per_plane=300, n_planes=3, n_outliers=400, seed=5)
    kw = dict(threshold=3.0, conf=0.99, sampler_id=0, seed=2, minimum_point_number=40)
    monkeypatch.setattr(_api, "_ctx", None)
    Hg, lg = px.findHomographies(pts, 1000, 1000, 1000, 1000, refit_solver="jacobi", **kw)
    Hd, ld = px.findHomographies(pts, 1000, 1000, 1000, 1000, **kw)
    monkeypatch.setattr(_api, "_ctx", OracleContext())
    Hc, lc = px.findHomographies(pts, 1000, 1000, 1000, 1000, refit_solver="jacobi", **kw)
    assert Hg.shape[0] >= 9 and np.array_equal(lg, lc) and np.allclose(Hg, Hc, rtol=1e-9, atol=1e-12)
    assert np.array_equal(lg, ld) and np.allclose(Hg, Hd, rtol=1e-9, atol=1e-12)
    p3, g3, _ = datasets.make_two_view_motions(n_per_motion=400, n_motions=2, n_outliers=300, seed=9)
    kw = dict(threshold=0.75, conf=0.99, sampler_id=0, seed=3, minimum_point_number=50, max_iters=500)
    monkeypatch.setattr(_api, "_ctx", None)
    Fg, lg = px.findTwoViewMotions(p3, 1000, 1000, 1000, 1000, refit_solver="jacobi", **kw)
    monkeypatch.setattr(_api, "_ctx", OracleContext())
    Fc, lc = px.findTwoViewMotions(p3, 1000, 1000, 1000, 1000, refit_solver="jacobi", **kw)
    assert Fg.shape == Fc.shape and Fg.shape[0] >= 3 and np.array_equal(lg, lc)
