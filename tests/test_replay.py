"""Rows a5, a10-a13 (ProgressiveX::run / PEARL::run and their helpers) against an oracle of their own - VERDICT r4 item 1.

pyprogressivex/_engine.py (the product's host loop) is run on the oracle-backed context and every decision it reports through
the `trace=` hook is compared with oracle/progx_replay.c, a second restatement written from progressive_x.h:251-624 and
PEARL.h:218-555 that replays the recorded proposals / refit results and recomputes everything else from the points.  The
scripted scenarios pin each quirk of the reference's control flow to a hand-derived expectation on BOTH streams.
tests/test_gpu_replay.py repeats all of it with libpgx.so on the MI355X."""
import os
import re

import numpy as np
import pytest

import progx_replay as R
import pyprogressivex as px
import replay_helpers as H
from oracle_ctx import OracleContext
from pyprogressivex import _api, _engine, datasets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_api(monkeypatch):
    monkeypatch.setattr(_api, "_ctx", OracleContext())


def test_event_tables_agree():
    hdr = open(os.path.join(ROOT, "oracle", "progx_replay.h")).read()
    enum = {m.group(1): int(m.group(2)) for m in re.finditer(r"PGXR_EV_([A-Z_]+) = (\d+)", hdr)}
    assert len(enum) == 15
    for name, code in enum.items():
        assert getattr(R, "EV_" + name) == code and getattr(_engine, "EV_" + name) == code and R.EVENT_NAMES[code] == name


def test_replay_module_is_independent_of_the_product():
    py = open(os.path.join(ROOT, "oracle", "progx_replay.py")).read()
    assert not re.search(r"^\s*(import|from)\s+pyprogressivex", py, re.M)
    c = open(os.path.join(ROOT, "oracle", "progx_replay.c")).read()
    assert "#include \"progx_replay.h\"" in c and "pgx.h" not in c.replace("pgx_oracle.h", "")


def test_c1_lines_decisions_equal_the_replay(cpu_api):
    pts, gt, _ = datasets.make_lines(seed=0)                                      # BASELINE config C1
    out, rec, rep = H.run_and_replay(px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1,
                                     minimum_point_number=50)
    assert H.assert_agree(out, rec, rep, 1) == 3
    verdicts, after, brk = H.summary(rec.events)
    assert verdicts[:3] == [1, 1, 1] and len(verdicts) == 10 and after[-1] == 3 and brk == [R.BREAK_LOOP_RAN_OUT]


def test_c1_abs_switch_u16(cpu_api):
    """[U-16] PEARL.h:465 with int abs(int): |dE| < 1 counts as no change - fewer PEARL iterations on the same scene; both
    restatements follow the switch"""
    pts, gt, _ = datasets.make_lines(seed=0)
    kw = dict(threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    out_d, rec_d, rep_d = H.run_and_replay(px.findLines, pts, np.array(0), 1000, 1000, **kw)
    out_i, rec_i, rep_i = H.run_and_replay(px.findLines, pts, np.array(0), 1000, 1000, pearl_abs="int", **kw)
    H.assert_agree(out_i, rec_i, rep_i, 1)
    n_d = sum(e[0] == R.EV_PEARL_ITER for e in rec_d.events)
    n_i = sum(e[0] == R.EV_PEARL_ITER for e in rec_i.events)
    assert n_i < n_d
    with pytest.raises(ValueError):
        px.findLines(pts, np.array(0), 1000, 1000, pearl_abs="float", **kw)


@pytest.mark.parametrize("l0", ["greedy", "expansion"])
def test_c2_homographies_decisions_equal_the_replay(cpu_api, l0):
    pts, gt, _ = datasets.make_homographies(seed=0)                               # BASELINE config C2
    out, rec, rep = H.run_and_replay(px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0,
                                     seed=1, minimum_point_number=50, labeling_l0=l0, max_iters=300)
    assert H.assert_agree(out, rec, rep, 3) >= 4
    assert any(e[0] == R.EV_REFIT and e[3] == 3 for e in rec.events)              # refits were accepted along the way


def test_three_object_pnp_decisions_equal_the_replay(cpu_api):
    x1, x2, K, gt, poses = datasets.make_poses(n_per_object=400, n_objects=3, n_outliers=400, seed=0)
    out, rec, rep = H.run_and_replay(px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=30)      # lambda = 0.1: alpha-expansion on the graph
    assert H.assert_agree(out, rec, rep, 3) == 3


def test_six_vanishing_points_decisions_equal_the_replay(cpu_api):
    pts, gt, _ = datasets.make_vanishing_points(n_inliers=1500, n_vps=6, n_outliers=1500, seed=0)
    out, rec, rep = H.run_and_replay(px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0,
                                     seed=1, minimum_point_number=60, spatial_coherence_weight=0.05, neighborhood_ball_radius=15.0)
    assert H.assert_agree(out, rec, rep, 1) >= 4


@pytest.mark.parametrize("scenario", H.SCENARIOS, ids=lambda f: f.__name__[9:])
def test_scripted_quirk(monkeypatch, scenario):
    scenario(OracleContext(), monkeypatch)


def test_scripted_stale_preference_vectors(monkeypatch, oracle):
    H.scenario_stale_preference_vectors(OracleContext(), monkeypatch, oracle)


def test_predicted_unseen_inliers_with_zero_iterations_is_the_literal_formula():
    """progressive_x.h:503: 1.0 / 0 = +inf, pow(x, inf) = 0 for x < 1, so the ratio is 1: no guard upstream"""
    assert _engine.predicted_unseen_inliers(0.05, 4, 0, 10, 100) == 90


def test_a_divergent_trace_is_reported_not_swallowed():
    pts, L, I = H.line_scene()
    s = dict(model_type=0, max_outer_iterations=10, pearl_maximum_iteration_number=100, labeling_l0=0, pearl_abs_int=0, sample_size=2,
             nonminimal_sample_size=2, minimum_number_of_inliers=20, max_proposal_number_without_change=10,
             maximum_model_number=(1 << 64) - 1, maximum_tanimoto_similarity=0.5, one_minus_confidence=0.01,
             inlier_outlier_threshold=2.0, spatial_coherence_weight=0.0)
    props = [(L["A"], I["A"], 2), (L["D"], I["D"], 2)]
    with pytest.raises(R.ReplayError) as e:                # two models -> PEARL asks for refits the trace does not hold
        R.replay(s, pts, None, props + [(None, None, 0)] * 8, [])
    assert e.value.code == -4
    with pytest.raises(R.ReplayError) as e:                # a refit computed for another inlier count: the labellings differ
        R.replay(s, pts, None, props + [(None, None, 0)] * 8, [(59, [L["A"]]), (40, [L["D"]])] * 4)
    assert e.value.code == -3
    with pytest.raises(R.ReplayError) as e:                # the loop wants more proposals than were recorded
        R.replay(s, pts, None, props[:1], [])
    assert e.value.code == -2


def test_replay_soak_slice_cpu(cpu_api):
    """25 random small calls (the generator of tests/soak_replay.py) on the oracle-backed context: every decision equals the replay,
    with NO tie tolerance (both sides sum sequentially)"""
    import soak_replay
    assert soak_replay.soak(7, 25, verbose=False, tie=0.0) == 0


def test_trace_hook_contract_is_checked_when_the_loops_are_built():
    """the hook's documented interface is .proposal / .refit(inlier_number, fits, accepted) / .event (+ optional .begin);
    an object that lacks one of them is refused up front instead of raising in the middle of PEARL"""
    from pyprogressivex import _engine

    class Half:
        def event(self, *a, **k):
            pass

        def proposal(self, *a):
            pass
    with pytest.raises(TypeError, match=r"\.refit"):
        _engine.check_trace_hook(Half())
    assert _engine.check_trace_hook(None) is None
    rec = R.TraceRecorder()
    assert _engine.check_trace_hook(rec) is rec
    rec.refit(7, [np.zeros(3)], True)          # the three-argument form the loops call
    assert rec.refits[0][0] == 7 and rec.refits[0][2] is True
