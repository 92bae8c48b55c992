"""Shared by tests/test_replay.py (CPU) and tests/test_gpu_replay.py: run a drop-in call (or the bare outer loop with a
scripted proposal engine) with a TraceRecorder, replay the trace through the independent restatement of
progressive_x.h / PEARL.h (oracle/progx_replay.c) and compare every decision.  TEST INFRASTRUCTURE."""
import numpy as np

import pgx_oracle as O
import progx_replay as R
from pyprogressivex import _engine, _estimators, _lib, _proposal


def graph_for(info):
    """the neighbourhood graph the run used, rebuilt by the ORACLE's builder from what the hook was told"""
    s = info["settings"]
    if not s.spatial_coherence_weight > 0.0:
        return None
    nb = info["neighborhood"]
    if nb == "radius":
        return O.graph_build(info["graph_points"], 1, radius=info["radius"])
    if nb.startswith("knn:"):
        return O.graph_build(info["graph_points"], 2, k=int(nb[4:]))
    return O.graph_build(info["graph_points"], 0, radius=info["radius"], k=5)


def run_and_replay(fn, *args, refit_tie_rtol=0.0, **kw):
    """fn(*args, trace=recorder, **kw) -> (result of the call, recorder, replay dict).  refit_tie_rtol: progx_replay.h - the GPU
    files pass 1e-12 (tree sums against sequential sums at PEARL.h:393), the CPU file leaves every decision to the replay."""
    rec = R.TraceRecorder()
    out = fn(*args, trace=rec, **kw)
    info = rec.info
    if info is None:          # an unknown sampler id: the call returns zero models before any loop runs (progressivex_python.cpp:240-245)
        return out, rec, None
    rep = R.replay(R.settings_from(info, refit_tie_rtol=refit_tie_rtol), info["points"], graph_for(info), rec.proposals, rec.refits)
    return out, rec, rep


def assert_agree(out, rec, rep, rows_per_model, model_rtol=0.0):
    models, labels = out
    diff = R.compare_events(rec.events, rep["events"])
    assert diff is None, diff + "\n--- run ---\n" + R.narrate(rec.events[-40:]) + "\n--- replay ---\n" + R.narrate(rep["events"][-40:])
    assert rep["consumed"] == (len(rec.proposals), len(rec.refits)), (rep["consumed"], len(rec.proposals), len(rec.refits))
    assert np.array_equal(np.asarray(labels, dtype=np.int64), rep["labels"])
    K = models.shape[0] // rows_per_model
    assert K == rep["models"].shape[0]
    return K


def summary(events):
    """(accepted per proposal, models after every PEARL iteration, break reason) - what VERDICT r4 item 1 names"""
    verdicts = [e[1] for e in events if e[0] == R.EV_VALIDATION]
    after = [e[3] // 2 for e in events if e[0] == R.EV_PEARL_END]
    brk = [e[1] for e in events if e[0] == R.EV_BREAK]
    return verdicts, after, brk


# ---- scripted runs: the bare loops with a proposal engine (and optionally a refit solver) that read from a script ------------
class ScriptedEngine:
    """stands in for gcransac::GCRANSAC::run (progressive_x.h:294-299): returns the script's proposals in order"""
    script = []

    def __init__(self, *a, **k):
        self.k = 0

    def run(self, T2, has_compound=False, exponent=2, weights=None):
        item = type(self).script[self.k]
        self.k += 1
        if item is None:
            return None
        model, inliers, iterations = item
        return dict(model=np.asarray(model, dtype=np.float64), inliers=np.asarray(inliers, dtype=np.int64), iterations=int(iterations))


class IdentityRefitLines(_estimators.LineEstimator):
    """estimateModelNonminimal that hands back the model it was given (sum after == sum before: PEARL.h:393 must not accept)"""

    def nonminimal_labels(self, ctx, K, weights, inits=None, skip=()):
        return [[] if k in skip else [np.asarray(inits[k], dtype=np.float64).copy()] for k in range(K)]


DEFAULT_TIE = 0.0     # the GPU file raises it to 1e-12 (tree sums against sequential sums: progx_replay.h refit_tie_rtol)


def scripted_run(ctx, monkeypatch, pts, script, est=None, refit_tie_rtol=None, **settings):
    """ProgressiveX.run on `ctx` with the scripted engine; settings are attributes of MultiModelSettings.
    Returns (models, statistics, recorder, replay)."""
    est = est or _estimators.LineEstimator()
    s = _engine.MultiModelSettings()
    for k, v in settings.items():
        if k == "confidence":
            s.set_confidence(v)
        else:
            assert hasattr(s, k), k
            setattr(s, k, v)
    engine = type("Engine", (ScriptedEngine,), dict(script=list(script)))
    monkeypatch.setattr(_proposal, "ProposalEngine", engine)
    rec = R.TraceRecorder()
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    px = _engine.ProgressiveX(ctx, est, pts, None, None, s, trace=rec)
    models, st = px.run()
    info = dict(model_type=est.model_type, points=pts, sample_size=est.sample_size, nonminimal_sample_size=est.nonminimal_sample_size,
                settings=s)
    rep = R.replay(R.settings_from(info, refit_tie_rtol=DEFAULT_TIE if refit_tie_rtol is None else refit_tie_rtol), pts, None,
                   rec.proposals, rec.refits)
    diff = R.compare_events(rec.events, rep["events"])
    assert diff is None, diff + "\n--- run ---\n" + R.narrate(rec.events) + "\n--- replay ---\n" + R.narrate(rep["events"])
    assert np.array_equal(np.asarray(st.labeling, dtype=np.int64), rep["labels"])
    assert len(models) == rep["models"].shape[0]
    for m, r in zip(models, rep["models"]):
        assert np.array_equal(m.descriptor, r)
    return models, st, rec, rep


def line_scene():
    """Hand-checkable 2-D line scene (threshold 2.0, minimum 20 inliers throughout):
       A: y = 500, 60 points (indices 0..59)
       B: y = 100: 19 points exactly on it (60..78) + the point (125, 101) one pixel off it (79): 20 points
       C: y = 101 + 0.04 (x - 125): through (125, 101) - the only point of B it is nearer to than B - and through its own 18
          points (80..97), which lie 19+ px from B
       D: x = 900 (vertical), 40 points (98..137);  8 stray points (138..145)."""
    A = np.column_stack([np.arange(60) * 10.0 + 5.0, np.full(60, 500.0)])
    B_on = np.column_stack([np.arange(300, 490, 10).astype(float), np.full(19, 100.0)])
    B_off = np.array([[125.0, 101.0]])
    xc = np.arange(600, 780, 10).astype(float)
    C_own = np.column_stack([xc, 101.0 + 0.04 * (xc - 125.0)])
    D = np.column_stack([np.full(40, 900.0), np.arange(40) * 7.0 + 200.0])
    stray = np.array([[30.0, 30.0], [700.0, 50.0], [650.0, 800.0], [300.0, 300.0], [50.0, 900.0], [770.0, 333.0], [420.0, 640.0], [610.0, 610.0]])
    pts = np.vstack([A, B_on, B_off, C_own, D, stray])
    nC = np.array([-0.04, 1.0]) / np.hypot(0.04, 1.0)
    lines = dict(A=np.array([0.0, 1.0, -500.0]), B=np.array([0.0, 1.0, -100.0]),
                 C=np.array([nC[0], nC[1], -(nC[0] * 125.0 + nC[1] * 101.0)]), D=np.array([1.0, 0.0, -900.0]))
    idx = dict(A=np.arange(0, 60), B=np.arange(60, 80), B_on=np.arange(60, 79), B_off=np.arange(79, 80), C_own=np.arange(80, 98),
               D=np.arange(98, 138), stray=np.arange(138, 146))
    return pts, lines, idx


# ---- the scripted quirk scenarios (run by the CPU file on the oracle-backed context and by the GPU file on libpgx) -----------
BASE = dict(minimum_number_of_inliers=20, inlier_outlier_threshold=2.0, maximum_tanimoto_similarity=0.5,
            spatial_coherence_weight=0.0, confidence=0.99)
GARBAGE1 = np.array([0.6, 0.8, -400.0])      # lines through nothing: every preference is 0
GARBAGE2 = np.array([0.8, -0.6, 100.0])


def _codes(events, code):
    return [e for e in events if e[0] == code]


def scenario_reject_counter_is_never_reset(ctx, monkeypatch):
    """progressive_x.h:342-345: ++unaccepted on every invalid proposal, NO reset on an accepted one; break when it EQUALS
    max_proposal_number_without_change (10).  5 invalid, 1 valid, 5 invalid: with the cap lifted to 20 the loop must stop
    at the 11th proposal with one model."""
    pts, L, I = line_scene()
    small = (L["B"], I["B"][:5], 3)                      # 5 inliers < max(sampleSize, 20): the size gate (:574)
    script = [small] * 5 + [(L["A"], I["A"], 2)] + [small] * 5 + [(L["D"], I["D"], 2)] * 9
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, max_outer_iterations=20, **BASE)
    assert [e[1] for e in _codes(rec.events, R.EV_UNACCEPTED)] == list(range(1, 11))
    assert _codes(rec.events, R.EV_BREAK)[0][1] == R.BREAK_REJECT_COUNTER
    assert len(_codes(rec.events, R.EV_OUTER)) == 11 and len(models) == 1 and rep["consumed"][0] == 11
    assert all(e[2] == 1 and np.isnan(e[4]) for e in _codes(rec.events, R.EV_VALIDATION) if e[1] == 0)   # reason: size gate


def scenario_empty_proposals_neither_count_nor_break(ctx, monkeypatch):
    """:301-303 `continue` before anything is counted: ten empty proposals use up the ten iterations, nothing else happens"""
    pts, L, I = line_scene()
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, [None] * 10, **BASE)
    assert [e[0] for e in rec.events] == [R.EV_OUTER, R.EV_PROPOSAL_EMPTY] * 10 + [R.EV_BREAK]
    assert rec.events[-1][1] == R.BREAK_LOOP_RAN_OUT and len(models) == 0 and not np.any(st.labeling)


def scenario_single_model_labels_and_covered_count(ctx, monkeypatch):
    """:375-385 the first model's labelling is GC-RANSAC's inlier list as given (0 = inlier, 1 = outlier, no PEARL), and
    :447-451 passes inliers_of_each_model.size() - the NUMBER of stored inlier sets (1) - as the covered inlier count."""
    pts, L, I = line_scene()
    subset = I["A"][::2]                                  # 30 of A's 60 points: the list is used as it is, not recomputed
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, [(L["A"], subset, 7)] + [None] * 9, **BASE)
    expect = np.ones(len(pts), dtype=np.int64)
    expect[subset] = 0
    assert np.array_equal(st.labeling, expect)
    covered, unseen = _codes(rec.events, R.EV_UNSEEN)[0][1:3]
    ratio = (1.0 - 0.01 ** (1.0 / 7)) ** 0.5
    assert covered == 1 and unseen == int(np.floor((len(pts) - 1) * ratio + 0.5))
    assert not _codes(rec.events, R.EV_PEARL_ITER)


def scenario_tanimoto_gate(ctx, monkeypatch):
    """:583-587: an all-zero preference against an all-zero compound is 0/0 = NaN and passes (`max < NaN` is false); a repeat
    of an accepted model has similarity 1 and is refused for reason 2; the refused proposal counts as unaccepted."""
    pts, L, I = line_scene()
    script = [(GARBAGE1, np.arange(22), 1), (L["A"], I["A"], 2), (L["A"], I["A"], 2)] + [None] * 7
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, **BASE)
    v = _codes(rec.events, R.EV_VALIDATION)
    assert v[0][1] == 1 and np.isnan(v[0][4])             # NaN -> valid
    assert v[2][1:3] == (0, 2) and abs(v[2][4] - 1.0) < 1e-12
    assert [e[1] for e in _codes(rec.events, R.EV_UNACCEPTED)] == [1]


def scenario_both_instances_rejected_in_reverse_order(ctx, monkeypatch):
    """PEARL.h:283-312 walks the instances from the last to the first; two lines through nothing (accepted on their scripted
    inlier lists) end with 0 points each: REJECT 1, then REJECT 0.  What follows is the reference's behaviour with an EMPTY
    model set: iteration 2 labels nothing (:486-487) and converges; getLabeling returns the stale labels of the two-model
    engine (every site carries the old outlier label 2, instance_number 2 with zero models); the compound vector is left as
    it was (:600-601); covered = N - N = 0.  The next accepted model is again a 'first' one: a second stored inlier set, so
    covered = 2."""
    pts, L, I = line_scene()
    script = [(GARBAGE1, np.arange(22), 1), (GARBAGE2, np.arange(30, 55), 2), (L["D"], I["D"], 2), (L["A"], I["A"], 2)] + [None] * 6
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, maximum_tanimoto_similarity=1.0,
                                        **{k: v for k, v in BASE.items() if k != "maximum_tanimoto_similarity"})
    rej = _codes(rec.events, R.EV_REJECT)
    assert [(e[1], e[2]) for e in rej] == [(1, 0), (0, 0)]
    ends = _codes(rec.events, R.EV_PEARL_END)
    assert ends[0][1:4] == (1, 1, 0) and ends[1][1:4] == (2, 0, 1)          # it 1: rejected, 0 models left; it 2: converged
    iters = _codes(rec.events, R.EV_PEARL_ITER)
    assert iters[1][2] == 0 and iters[1][3] == 0 and iters[1][4] == iters[0][4]   # no models, fresh flag, energy untouched
    lab = _codes(rec.events, R.EV_LABELING)[0]
    assert lab[1:3] == (2, 0)
    uns = _codes(rec.events, R.EV_UNSEEN)
    assert uns[1][1] == 0 and uns[2][1] == 2                                 # covered: N - outliers = 0, then TWO stored sets
    assert len(models) == 2 and np.array_equal(np.bincount(st.labeling, minlength=3), [40, 60, len(pts) - 100])


def scenario_refit_acceptance_is_strict_and_warm_start_rule(ctx, monkeypatch):
    """PEARL.h:393 accepts a refit only if the sum of residuals got strictly smaller: a refit that returns the same model is
    not a change, so the second iteration converges; :429-431 the labelling is warm-started iff iteration > 1 and nothing was
    rejected in the previous one."""
    pts, L, I = line_scene()
    script = [(L["A"], I["A"], 2), (L["D"], I["D"], 2)] + [None] * 8
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, est=IdentityRefitLines(), **BASE)
    refits = _codes(rec.events, R.EV_REFIT)
    assert refits and all(e[3] == 2 and e[4] == e[5] for e in refits)        # one model returned, equal sums, NOT accepted
    ends = _codes(rec.events, R.EV_PEARL_END)
    assert [e[1:4] for e in ends] == [(1, 0, 4), (2, 0, 5)]
    assert [e[3] for e in _codes(rec.events, R.EV_PEARL_ITER)] == [0, 1]


def scenario_stale_preference_vectors(ctx, monkeypatch, oracle):
    """progressive_x.h:620-621 maximises over the STORED preference vectors; PEARL's setDescriptor (PEARL.h:396) does not
    recompute them.  Slightly wrong proposals of noisy lines get refitted (accepted: strictly smaller sums); the compound
    vector must still be the maximum of the preferences of the descriptors AS PROPOSED."""
    from pyprogressivex import datasets
    pts, gt, lines = datasets.make_lines(n_per_line=150, n_lines=2, n_outliers=100, seed=3)
    lines = np.asarray(lines, dtype=np.float64).reshape(2, 3)
    prop = []
    for k in range(2):
        a, b, c = lines[k]
        th = np.arctan2(b, a) + 0.004                     # rotate the normal a little about the line's centroid
        cen = pts[gt == k + 1].mean(axis=0)
        n = np.array([np.cos(th), np.sin(th)])
        prop.append(np.array([n[0], n[1], -n @ cen]))
    inl = [np.nonzero(gt == k + 1)[0] for k in range(2)]
    script = [(prop[0], inl[0], 2), (prop[1], inl[1], 2)] + [None] * 8
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, **BASE)
    assert any(e[3] == 3 for e in _codes(rec.events, R.EV_REFIT))             # at least one refit was accepted ...
    assert not np.array_equal(models[0].descriptor, prop[0]) or not np.array_equal(models[1].descriptor, prop[1])
    T2 = 2.25 * 2.0 * 2.0
    stale = np.maximum(oracle.preference(oracle.LINE2D, pts, prop[0], T2), oracle.preference(oracle.LINE2D, pts, prop[1], T2))
    fresh = np.maximum(oracle.preference(oracle.LINE2D, pts, models[0].descriptor, T2),
                       oracle.preference(oracle.LINE2D, pts, models[1].descriptor, T2))
    comp = _codes(rec.events, R.EV_COMPOUND)[-1]
    assert abs(comp[4] - stale.sum()) <= 1e-9 * stale.sum() and abs(stale.sum() - fresh.sum()) > 1e-6
    assert np.array_equal(ctx.get_compound(), stale)


def scenario_break_rules(ctx, monkeypatch):
    """:468 unseen < minimum_number_of_inliers and :472 getModelNumber() >= maximum_model_number"""
    pts, L, I = line_scene()
    script = [(L["A"], I["A"], 2), (L["D"], I["D"], 2), (L["B"], I["B"], 2)] + [None] * 7
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, maximum_model_number=2, **BASE)
    assert _codes(rec.events, R.EV_BREAK)[0][1] == R.BREAK_MODEL_NUMBER and len(models) == 2 and rep["consumed"][0] == 2
    script = [(L["A"], I["A"], 5000)] + [None] * 9          # many RANSAC iterations: the predicted unseen count falls below 20
    models, st, rec, rep = scripted_run(ctx, monkeypatch, pts, script, **BASE)
    assert _codes(rec.events, R.EV_BREAK)[0][1] == R.BREAK_UNSEEN and _codes(rec.events, R.EV_UNSEEN)[0][2] < 20
    assert rep["consumed"][0] == 1


SCENARIOS = [scenario_reject_counter_is_never_reset, scenario_empty_proposals_neither_count_nor_break,
             scenario_single_model_labels_and_covered_count, scenario_tanimoto_gate,
             scenario_both_instances_rejected_in_reverse_order, scenario_refit_acceptance_is_strict_and_warm_start_rule,
             scenario_break_rules]
