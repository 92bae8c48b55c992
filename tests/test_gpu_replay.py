"""tests/test_replay.py on the MI355X: the drop-in calls run through libpgx.so, their decisions (accept / reject per proposal,
PEARL iteration by iteration: energy, refit acceptance, rejections, model count, convergence; break reason), final labels and
model set must equal the independent replay of progressive_x.h / PEARL.h (oracle/progx_replay.c) - VERDICT r4 item 1."""
import numpy as np
import pytest

import progx_replay as R
import pyprogressivex as px
import replay_helpers as H
from pyprogressivex import _api, datasets

pytestmark = pytest.mark.gpu


def _rr(fn, *a, **kw):
    """PEARL.h:393 compares two sums whose last bits depend on the summation order (a fixed tree on the GPU, sequential in the
    replay and upstream): comparisons that agree to 1e-12 are ties and follow the recorded run (progx_replay.h refit_tie_rtol);
    every other decision is the replay's own."""
    return H.run_and_replay(fn, *a, refit_tie_rtol=1e-12, **kw)


@pytest.fixture(autouse=True)
def _tie_tolerance(monkeypatch):
    monkeypatch.setattr(H, "DEFAULT_TIE", 1e-12)


@pytest.fixture()
def gpu_api(monkeypatch):
    monkeypatch.setattr(_api, "_ctx", None)       # the package creates its libpgx context lazily (raises without a GPU)


def test_c1_lines_gpu_decisions_equal_the_replay(gpu_api):
    pts, gt, _ = datasets.make_lines(seed=0)
    out, rec, rep = _rr(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1,
                                     minimum_point_number=50)
    assert H.assert_agree(out, rec, rep, 1) == 3
    verdicts, after, brk = H.summary(rec.events)
    assert verdicts[:3] == [1, 1, 1] and after[-1] == 3 and brk == [R.BREAK_LOOP_RAN_OUT]


@pytest.mark.parametrize("l0", ["greedy", "expansion"])
def test_c2_homographies_gpu_decisions_equal_the_replay(gpu_api, l0):
    pts, gt, _ = datasets.make_homographies(seed=0)
    out, rec, rep = _rr(px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0,
                                     seed=1, minimum_point_number=50, labeling_l0=l0)
    assert H.assert_agree(out, rec, rep, 3) == 5


def test_three_object_pnp_gpu_decisions_equal_the_replay(gpu_api):
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=600, n_objects=3, n_outliers=600, seed=0)
    out, rec, rep = _rr(px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30)
    assert H.assert_agree(out, rec, rep, 3) == 3


def test_six_vanishing_points_gpu_decisions_equal_the_replay(gpu_api):
    pts, gt, _ = datasets.make_vanishing_points(n_inliers=3000, n_vps=6, n_outliers=3000, seed=0)
    out, rec, rep = _rr(px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0,
                                     seed=1, minimum_point_number=100, spatial_coherence_weight=0.05, neighborhood_ball_radius=15.0)
    assert H.assert_agree(out, rec, rep, 1) >= 4


def test_philox_device_sampled_run_gpu_decisions_equal_the_replay(gpu_api):
    pts, gt, _ = datasets.make_homographies(seed=0)
    out, rec, rep = _rr(px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0,
                                     seed=1, minimum_point_number=50, sampler_rng="philox")
    assert H.assert_agree(out, rec, rep, 3) >= 4


def test_u16_int_abs_gpu(gpu_api):
    pts, gt, _ = datasets.make_lines(seed=0)
    out, rec, rep = _rr(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1,
                                     minimum_point_number=50, pearl_abs="int")
    H.assert_agree(out, rec, rep, 1)


@pytest.mark.parametrize("scenario", H.SCENARIOS, ids=lambda f: f.__name__[9:])
def test_scripted_quirk_gpu(gpu_ctx, monkeypatch, scenario):
    scenario(gpu_ctx, monkeypatch)


def test_scripted_stale_preference_vectors_gpu(gpu_ctx, monkeypatch, oracle):
    H.scenario_stale_preference_vectors(gpu_ctx, monkeypatch, oracle)


def test_c3_full_size_gpu_decisions_equal_the_replay(gpu_api):
    """BASELINE config C3 at full size (1e5 correspondences, 8 motions): the whole findTwoViewMotions call - 10 proposals, ~244 PEARL
    iterations with their refit decisions, rejections and convergence tests - against the independent replay (lambda = 0: GCO's
    special-case labelling, so the replay's labellings are the oracle's greedy solver on 1e5 x 9 tables)."""
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    out, rec, rep = _rr(px.findTwoViewMotions, pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1,
                        minimum_point_number=1000, max_iters=2000)
    assert H.assert_agree(out, rec, rep, 3) == 7
    assert sum(e[0] == R.EV_PEARL_ITER for e in rec.events) > 200


def test_c5_full_size_gpu_decisions_equal_the_replay(gpu_api):
    """BASELINE config C5 at full size (2e5 segments, 6 vanishing points, spatial coherence on): the whole findVanishingPoints call -
    10 proposals, 229 PEARL iterations, ~1 500 refit decisions - against the independent replay, whose 229 labellings are the oracle's
    Dinic expansions on the 2e5-site graph (about two minutes of host time: the longest test of the suite, and the one that ties the
    device's region moves, the first-cycle memo and the identical-call answer to an independent computation of every PEARL step)."""
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    out, rec, rep = _rr(px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1,
                        minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
    assert H.assert_agree(out, rec, rep, 1) == 9
    assert sum(e[0] == R.EV_PEARL_ITER for e in rec.events) > 200


# ---- the proposal engine against ITS independent restatement (oracle/progx_proposal.c) -----------------------
def _walks_agree(fn, *a, **kw):
    import progx_proposal as Q
    rec = Q.WalkRecorder()
    out = fn(*a, trace=rec, **kw)
    assert rec.walks
    for k, w in enumerate(rec.walks):
        diff = Q.compare(w)
        assert diff is None, f"proposal {k}: {diff}"
    return out, rec


def test_proposal_walks_on_the_gpu_equal_the_replay_c1_c2_pnp_philox(gpu_api):
    """every proposal of C1, C2, a 3-object pose scene (min-cut local optimisation) and a run on device-drawn Philox batches: which
    hypothesis becomes the so-far-best and when, the iteration bound, when the local optimisation fires, which refit it keeps, the
    final least squares, the iteration count handed to ProgressiveX::run - recomputed by the replay from the score tables"""
    import progx_proposal as Q
    pts, gt, _ = datasets.make_lines(seed=0)
    out, rec = _walks_agree(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    assert len(rec.walks) == 10 and out[0].shape[0] == 3
    pts, gt, _ = datasets.make_homographies(seed=0)
    out, rec = _walks_agree(px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    assert out[0].shape[0] // 3 == 5
    assert any(e[0] == Q.EV_LO_ROUND and e[1] == 1 and e[3] == 1 for w in rec.walks for e in w["events"])
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=400, n_objects=3, n_outliers=400, seed=0)
    out, rec = _walks_agree(px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30)
    assert out[0].shape[0] // 3 == 3
    pts, gt, _ = datasets.make_lines(seed=3)
    _walks_agree(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.95, sampler_id=0, seed=5, minimum_point_number=40, sampler_rng="philox")


def test_proposal_walks_at_full_size_equal_the_replay_c3_c5_c4(gpu_api):
    """the same at the BASELINE sizes: C3 (1e5 correspondences, 2 000 seven-point samples x 3 slots per proposal), C5 (2e5 segments) and
    C4 (1e6 correspondences, 2 048 P3P samples x 4 slots): the replay is a pass over the score table, so full size costs nothing"""
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    out, rec = _walks_agree(px.findTwoViewMotions, pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1,
                            minimum_point_number=1000, max_iters=2000)
    assert max(len(w["counts"]) for w in rec.walks) == 6000
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    _walks_agree(px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1,
                 minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    out, rec = _walks_agree(px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048)
    assert max(len(w["counts"]) for w in rec.walks) == 8192 and out[0].shape[0] // 3 == 9


def test_c4_full_size_events_match_the_committed_replay_pin(gpu_api):
    """`find6DPoses` on 1e6 correspondences (the metric's configuration) against the independent control-flow oracle
    WITHOUT the 35 minutes of host Dinic: tests/golden/kat_c4_replay_v1.npz holds the ORACLE's decision stream, computed once from the
    recorded proposals / refits of this very call (scripts/pin_c4_replay.py record + replay: 228 events, 23 PEARL iterations whose
    labellings of 1e6 sites the oracle's expansion recomputed; agreed event for event, 2 summation-order ties followed).  This test
    runs the call again (~1 s): the replay's inputs must be the recorded ones byte for byte (else the pin does not apply and must be
    regenerated - a different failure from a wrong decision), every decision must equal the pinned oracle stream, and the labels and
    models must be the oracle's."""
    import hashlib
    import importlib.util
    import os
    pin = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_c4_replay_v1.npz"), allow_pickle=False)
    spec = importlib.util.spec_from_file_location("pin_c4_replay", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "pin_c4_replay.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    rec = R.TraceRecorder()
    models, labels = px.find6DPoses(x1, x2, K, trace=rec, **tool.ARGS)
    assert tool.inputs_digest(rec.proposals, rec.refits) == str(pin["inputs_sha256"]), \
        "the proposals / refits of the call are not the recorded ones: regenerate the pin (scripts/pin_c4_replay.py), this is not a decision mismatch"
    oracle_events = [tuple(int(v) for v in e[:4]) + (float(e[4]), float(e[5])) for e in pin["oracle_events"]]
    diff = R.compare_events(rec.events, oracle_events)
    assert diff is None, diff
    assert len(rec.events) == 228 and sum(e[0] == R.EV_PEARL_ITER for e in rec.events) == 23
    assert hashlib.sha256(np.asarray(labels, dtype=np.int64).tobytes()).hexdigest() == str(pin["labels_sha256"])
    assert np.array_equal(np.bincount(np.asarray(labels, dtype=np.int64), minlength=len(pin["label_histogram"])), pin["label_histogram"])
    assert models.shape[0] // 3 == pin["models"].shape[0] == 9
    assert np.array_equal(models.reshape(9, -1), pin["models"].reshape(9, -1))
