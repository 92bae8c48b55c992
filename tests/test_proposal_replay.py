"""the proposal engine (pyprogressivex/_proposal.py: sampling -> batch scoring -> walk -> local optimisation ->
final least squares) checked against a SECOND restatement of the loop, oracle/progx_proposal.c, written from the sequential algorithm
(every hypothesis in generation order, scoring_function_with_compound_model.h:105-106, the iteration bound, when the local
optimisation fires and what it keeps).  The product hands the replay what its walk saw - the score table of the batch, and for every
local-optimisation round / least-squares step what the cut and the refit solver returned - and the replay recomputes every decision.
CPU file: the oracle-backed context; tests/test_gpu_replay.py runs the same scenes on the GPU."""
import os
import re

import numpy as np
import pytest

import progx_proposal as Q
import pyprogressivex as px
from oracle_ctx import OracleContext
from pyprogressivex import _api, _proposal, datasets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_api(monkeypatch):
    monkeypatch.setattr(_api, "_ctx", OracleContext())


def run_and_check(fn, *a, **kw):
    rec = Q.WalkRecorder()
    out = fn(*a, trace=rec, **kw)
    assert rec.walks, "the proposal engine recorded nothing"
    for k, w in enumerate(rec.walks):
        diff = Q.compare(w)
        assert diff is None, f"proposal {k}: {diff}"
    return out, rec


def test_event_codes_agree_and_the_replay_is_independent():
    hdr = open(os.path.join(ROOT, "oracle", "progx_proposal.h")).read()
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"PGXQ_EV_([A-Z_]+) = (\d+)", hdr)}
    assert len(enum) == 6
    for name, code in enum.items():
        assert getattr(Q, "EV_" + name) == code and getattr(_proposal, "WK_" + name) == code
    assert not re.search(r"^\s*(import|from)\s+pyprogressivex", open(os.path.join(ROOT, "oracle", "progx_proposal.py")).read(), re.M)
    assert "pgx.h" not in open(os.path.join(ROOT, "oracle", "progx_proposal.c")).read()


def test_c1_lines_proposals_equal_the_replay(cpu_api):
    pts, gt, _ = datasets.make_lines(seed=0)
    out, rec = run_and_check(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    assert len(rec.walks) == 10 and out[0].shape[0] == 3
    codes = [e[0] for w in rec.walks for e in w["events"]]
    assert Q.EV_BEST in codes and Q.EV_LO_ROUND in codes and Q.EV_LSQ in codes and Q.EV_FINAL in codes


def test_c2_homographies_proposals_equal_the_replay(cpu_api):
    pts, gt, _ = datasets.make_homographies(seed=0)
    out, rec = run_and_check(px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0, seed=1,
                             minimum_point_number=50, max_iters=300)
    assert out[0].shape[0] // 3 >= 4
    assert any(e[0] == Q.EV_LO_ROUND and e[1] == 1 and e[3] == 1 for w in rec.walks for e in w["events"])     # an inner RANSAC of refits improved a model


def test_pnp_scene_with_the_graph_cut_proposals_equal_the_replay(cpu_api):
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=400, n_objects=3, n_outliers=400, seed=0)
    out, rec = run_and_check(px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30)      # lambda = 0.1: the cut is a real min-cut
    assert out[0].shape[0] // 3 == 3


def test_philox_sampler_run_proposals_equal_the_replay(cpu_api):
    pts, gt, _ = datasets.make_lines(seed=3)
    run_and_check(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.95, sampler_id=0, seed=5, minimum_point_number=40,
                  sampler_rng="philox")


def test_the_replay_notices_a_wrong_decision(cpu_api):
    """the comparison is not vacuous: tamper with one recorded decision of each kind and the replay objects"""
    pts, gt, _ = datasets.make_lines(seed=0)
    _, rec = run_and_check(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    w = rec.walks[0]
    for code in (Q.EV_BEST, Q.EV_LO_ROUND, Q.EV_WALK_END, Q.EV_LSQ):
        k = next(i for i, e in enumerate(w["events"]) if e[0] == code)
        bad = dict(w, events=list(w["events"]))
        e = bad["events"][k]
        bad["events"][k] = (e[0], e[1] + 1) + tuple(e[2:])
        assert Q.compare(bad) is not None, code
    # a so-far-best the count test should have passed over: lower the best's count in the table so that an earlier rival qualifies
    bad = dict(w, counts=w["counts"].copy())
    first = next(e for e in w["events"] if e[0] == Q.EV_BEST)
    bad["counts"][first[1]] = 0
    assert Q.compare(bad) is not None


def test_hand_made_walk():
    """a table small enough to decide by hand: 6 hypotheses from 6 samples of 2 points out of n = 100, no local optimisation before
    iteration 20: h0 (10 inliers) becomes the best; h1 has a better score but 8 + 1 < 10 inliers: passed over (:105-106); h2 is not
    strictly better; h3 (30 inliers, better) wins; then the bound log(0.01) / log(1 - 0.09) = 48.8 stays above every iteration;
    the local optimisation runs once after the loop (one round, too few inliers for the non-minimal solver: branch 0)"""
    rec = dict(n=100, samples=6, sample_size=2, nonminimal_sample_size=3, confidence=0.99, max_iters=1000, min_iters=0, lo_after=20, every_best=True,
               max_cuts=10, lsq_budget=10, counts=np.array([10, 8, 10, 30, 29, 0]), scores=np.array([5.0, 6.0, 5.0, 20.0, 19.0, -np.inf]),
               src=np.arange(6), rounds=[(2, np.zeros(0, np.int64), np.zeros(0))], lsq=[(2, 0, 0, -np.inf)], events=[])
    ev, consumed = Q.replay(rec)
    assert ev == [(Q.EV_BEST, 0, 1, 10, 5.0), (Q.EV_BEST, 3, 4, 30, 20.0), (Q.EV_WALK_END, 49, 3, 0, 20.0),
                  (Q.EV_LO_ROUND, 0, 0, 0, 20.0), (Q.EV_LO_END, 1, 30, 1, 20.0), (Q.EV_LSQ, 1, 0, 0, 20.0), (Q.EV_FINAL, 0, 0, 0, 20.0)]
    assert consumed == (1, 1)
