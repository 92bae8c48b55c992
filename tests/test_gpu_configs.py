"""BASELINE.json configs at FULL size on the GPU: C3 (1e5 two-view correspondences, 8 motions,
fundamental matrices / Sampson distance) and the scoring half of C5 (2e5 segments, 6 vanishing points) against the
oracle on the same seeded inputs; C4 and the labelling half of C5 live in test_gpu_parity.py."""
import numpy as np
import pytest

import pyprogressivex as px
from oracle_ctx import OracleContext
from pyprogressivex import _api, _estimators, _lib, datasets

pytestmark = pytest.mark.gpu
REL = 1e-9


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


@pytest.fixture(scope="module")
def c3():
    pts, gt, Fs = datasets.make_two_view_motions(seed=0)          # 8 motions x 1e4 inliers + 2e4 outliers
    assert pts.shape == (100000, 4) and len(Fs) == 8
    return pts, gt, Fs


def _f_hypotheses(pts, gt, Fs, M, seed):
    """M fundamental matrices: the 8 ground-truth ones, 7-point solutions of all-inlier samples, of mixed samples and of
    random samples (counts from a handful to 1e4) - the oracle's own minimal solver makes them, NaN rows dropped."""
    import pgx_oracle as O
    rng = np.random.default_rng(seed)
    smp = []
    for s in range(3 * M):
        kind = s % 3
        if kind == 0:
            smp.append(rng.choice(np.nonzero(gt == 1 + s % 8)[0], 7, replace=False))
        elif kind == 1:
            smp.append(np.concatenate([rng.choice(np.nonzero(gt == 1 + s % 8)[0], 5, replace=False),
                                       rng.choice(len(gt), 2, replace=False)]))
        else:
            smp.append(rng.choice(len(gt), 7, replace=False))
    sol = O.solve_minimal(O.FUNDAMENTAL, pts, np.array(smp, dtype=np.int32))
    sol = sol[~np.isnan(sol[:, 0])]
    return np.ascontiguousarray(np.vstack([Fs.reshape(8, 9), sol])[:M])


def test_c3_scoring_all_points_vs_oracle(gpu_ctx, oracle, c3):
    pts, gt, Fs = c3
    hyps = _f_hypotheses(pts, gt, Fs, 96, seed=3)
    assert hyps.shape == (96, 9)
    thr = 0.75
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(_lib.FUNDAMENTAL, pts)
    # a non-empty compound instance: the preference vector of motion 0
    pref = gpu_ctx.preference(Fs[0].reshape(-1), T2, slot=0, want_pref=True)["pref"]
    assert np.array_equal(pref, oracle.preference(oracle.FUNDAMENTAL, pts, Fs[0].reshape(-1), T2))
    comp = gpu_ctx.compound_update([0], want_compound=True)
    got = gpu_ctx.score(hyps, T2, has_compound=True, exponent=2, want_masks=True)
    ref = oracle.score(oracle.FUNDAMENTAL, pts, hyps, T2, compound=comp, has_compound=True, exponent=2, want_masks=True)
    assert np.array_equal(got["counts"], ref["counts"]) and np.array_equal(got["masks"], ref["masks"])
    assert _rel(got["values"], ref["values"]) <= REL and _rel(got["shared"], ref["shared"]) <= REL
    assert np.all(got["counts"][:8] > 9000) and got["counts"].min() < 3000      # the batch spans the range
    # without masks the same numbers come back (the mask-free kernel variant)
    again = gpu_ctx.score(hyps, T2, has_compound=True, exponent=2)
    assert np.array_equal(again["counts"], ref["counts"]) and _rel(again["values"], ref["values"]) <= REL


def test_c3_unary_table_and_lambda0_expansion_vs_oracle(gpu_ctx, oracle, c3):
    pts, gt, Fs = c3
    thr, lam, h = 0.75, 0.0, 1000.0                       # findTwoViewMotions default lambda = 0 (bindings.cpp:445-461)
    models = Fs.reshape(8, 9)
    gpu_ctx.set_points(_lib.FUNDAMENTAL, pts)
    Dq = gpu_ctx.pearl_unary(models, thr, lam, want_table=True)
    assert Dq.shape == (100000, 9)
    assert np.array_equal(Dq, oracle.unary_q(oracle.FUNDAMENTAL, pts, models, thr, lam))      # 1e5 x 9, bit-exact
    empty = (np.zeros(len(pts) + 1, np.int32), np.zeros(1, np.int32), np.ones(1, np.int32))
    gpu_ctx.set_labels(np.zeros(len(pts), np.int32))
    eq, e, cyc = gpu_ctx.expansion(lam, h)
    ref_labels, ref_e, ref_cyc = oracle.expansion(Dq, empty, oracle.quantize_lambda(lam), oracle.quantize(h),
                                                  np.zeros(len(pts), np.int32))
    assert np.array_equal(gpu_ctx.get_labels(), ref_labels) and eq == ref_e and cyc == ref_cyc
    counts, order = gpu_ctx.bucket(9)
    rc, ro = oracle.bucket(ref_labels, 9)
    assert np.array_equal(counts, rc) and np.array_equal(order, ro)
    sums = gpu_ctx.residual_sums(models)
    for k in range(8):
        assert abs(sums[k] - oracle.residual_sum(oracle.FUNDAMENTAL, pts, models[k], ref_labels, k)) <= REL * max(1.0, abs(sums[k]))
    agree = np.mean(ref_labels[gt > 0] == gt[gt > 0] - 1)
    assert agree > 0.9


def test_c3_end_to_end_identical_to_cpu_restatement(monkeypatch, c3):
    """findTwoViewMotions at C3 size, GPU vs the same host code on the CPU port for the same seed: identical labelling and
    model set (max_iters kept small so that the CPU leg stays under a minute)."""
    pts, gt, Fs = c3
    kw = dict(threshold=0.75, conf=0.99, sampler_id=0, seed=1, minimum_point_number=1000, max_iters=300)
    monkeypatch.setattr(_api, "_ctx", None)
    F, lab = px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, **kw)
    gpu = _api._ctx
    monkeypatch.setattr(_api, "_ctx", OracleContext())
    Fr, labr = px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, **kw)
    monkeypatch.setattr(_api, "_ctx", gpu)
    assert F.shape == Fr.shape and F.shape[0] >= 3 * 4
    assert np.array_equal(lab, labr)
    assert np.allclose(F, Fr, rtol=1e-7, atol=1e-9)


def test_c3_quality_with_the_full_iteration_budget():
    """Sanity only.  On this synthetic scene the method itself (not the GPU path: the CPU restatement returns the identical
    labelling, previous test) ends near ME 0.47: one fundamental matrix absorbs two motions whose epipolar geometry nearly
    coincides (9.7k + 9.5k of their points, scripts/dbg_c3.py) and the remaining instances, refitted on mixtures, keep
    45-60 % of their motions - the weak (codimension-1) epipolar constraint at work, as on the bundled cubetoy scene."""
    pts, gt, Fs = datasets.make_two_view_motions(seed=0)
    # the measured band at this seed, not a sanity bound (a regression 0.47 -> 0.53 used to pass silently):
    # default (validity="off", the strict restatement): 7 motions, ME 0.466 (profiles/round2_api.txt and the round-4 GPU run);
    # with the opt-in U-14 stages 8 motions, ME 0.527
    for validity, motions, lo, hi in (("off", 7, 0.44, 0.49), ("full", 8, 0.50, 0.55)):
        F, lab = px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1,
                                       minimum_point_number=1000, max_iters=2000, validity=validity)
        K = F.shape[0] // 3
        me = datasets.misclassification(np.where(lab == K, 0, lab + 1), gt)
        print(f"C3 findTwoViewMotions validity={validity}: {K} motions, misclassification {me:.4f}")
        assert K == motions and lo <= me <= hi, (validity, K, me)


def test_c5_vanishing_point_scoring_all_segments_vs_oracle(gpu_ctx, oracle):
    pts, gt, vps = datasets.make_vanishing_points(seed=0)          # 2e5 segments, 6 vanishing points, 50 % outliers
    assert pts.shape == (199996, 4)                                 # 6 x 16666 inliers + 1e5 outliers
    rng = np.random.default_rng(4)
    smp = np.array([rng.choice(np.nonzero(gt == 1 + s % 6)[0], 2, replace=False) if s % 2 == 0 else
                    rng.choice(len(gt), 2, replace=False) for s in range(250)], dtype=np.int32)
    hyps = np.vstack([vps, oracle.solve_minimal(oracle.VANISHING_POINT, pts, smp)])
    hyps = np.ascontiguousarray(hyps[~np.isnan(hyps[:, 0])][:200])
    thr = 1.5
    T2 = 2.25 * thr * thr
    gpu_ctx.set_points(_lib.VANISHING_POINT, pts)
    pref = gpu_ctx.preference(vps[0], T2, slot=0, want_pref=True)["pref"]
    assert np.array_equal(pref, oracle.preference(oracle.VANISHING_POINT, pts, vps[0], T2))
    comp = gpu_ctx.compound_update([0], want_compound=True)
    got = gpu_ctx.score(hyps, T2, has_compound=True, exponent=2, want_masks=True)
    ref = oracle.score(oracle.VANISHING_POINT, pts, hyps, T2, compound=comp, has_compound=True, exponent=2, want_masks=True)
    assert np.array_equal(got["counts"], ref["counts"]) and np.array_equal(got["masks"], ref["masks"])
    assert _rel(got["values"], ref["values"]) <= REL and _rel(got["shared"], ref["shared"]) <= REL
    assert np.all(got["counts"][:6] > 12000)


def test_c4_with_the_proposal_cap_lifted():
    """BASELINE config C4 names 16 objects; the reference's outer loop stops after 10 proposals (progressive_x.h:272), so the drop-in
    default returns at most 10.  With the cap lifted (keyword-only max_outer_iterations=20) the same call keeps going: 13 proposals
    accepted, 12 objects kept at this seed (measured, scripts/c4_objects.py: the proposal engine stops finding the last objects
    at max_iters = 2048 and the loop ends on its 20 proposals), every kept model on a distinct ground-truth object, and the
    labels agree with the ground truth on the objects found."""
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    P, lab = px.find6DPoses(x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048, max_outer_iterations=20)
    k = P.shape[0] // 3
    assert 12 <= k <= 16
    owners = [int(np.bincount(gt[lab == m], minlength=17)[1:].argmax()) + 1 for m in range(k)]
    assert len(set(owners)) == k                                   # one model per object
    for m, o in enumerate(owners):
        sel = gt == o
        assert np.mean(lab[sel] == m) > 0.97                       # the object's points carry its model's label
    me = datasets.misclassification(np.where(lab == k, 0, lab + 1), gt)
    assert me <= (16 - k) * 0.05 + 0.02                            # what is missing is the objects not found (5 % of the points each)


def test_c4_all_sixteen_objects():
    """BASELINE config C4's 16 objects.  What stops the default call short of them is the reference's own scoring at this size
    (scripts/c4_missing.py): score = value - shared^2 (scoring_function_with_compound_model.h:110-121, exponent 2 for this driver:
    progressive_x.h:183), and a new object's incidental overlap with a dozen accepted models - 300-500 units of 10^6 points - squared
    exceeds its whole value (~47 000): its score is negative, junk wins the proposal.  With the two keyword-only extensions
    (the 10-proposal cap lifted, the penalty's exponent 1) the same call returns all 16, one model per object."""
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    P, lab = px.find6DPoses(x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048, max_outer_iterations=32, scoring_exponent=1)
    k = P.shape[0] // 3
    assert k == 16
    owners = [int(np.bincount(gt[lab == m], minlength=17)[1:].argmax()) + 1 for m in range(k)]
    assert sorted(owners) == list(range(1, 17))
    assert datasets.misclassification(np.where(lab == k, 0, lab + 1), gt) < 0.01
