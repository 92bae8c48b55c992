"""ctypes view of oracle/progx_replay.c — the independent oracle of the control-flow rows (SURVEY §8 a5, a10–a13) — and the
recorder that captures a run's trace through the `trace=` hook of the drop-in API.

TEST INFRASTRUCTURE ONLY: importable from tests/ (and scripts/), never from the product package.  PARITY UNPINNED.
This module does not import pyprogressivex: the replay is a second restatement of progressive_x.h:251-624 /
PEARL.h:218-555, not a re-run of _engine.py.
"""
import ctypes as C

import numpy as np

import pgx_oracle as O

(EV_OUTER, EV_PROPOSAL_EMPTY, EV_PROPOSAL, EV_VALIDATION, EV_UNACCEPTED, EV_SINGLE_MODEL, EV_PEARL_ITER, EV_REFIT_SKIP, EV_REFIT,
 EV_REJECT, EV_PEARL_END, EV_LABELING, EV_COMPOUND, EV_UNSEEN, EV_BREAK) = range(1, 16)
EVENT_NAMES = {1: "OUTER", 2: "PROPOSAL_EMPTY", 3: "PROPOSAL", 4: "VALIDATION", 5: "UNACCEPTED", 6: "SINGLE_MODEL", 7: "PEARL_ITER",
               8: "REFIT_SKIP", 9: "REFIT", 10: "REJECT", 11: "PEARL_END", 12: "LABELING", 13: "COMPOUND", 14: "UNSEEN", 15: "BREAK"}
BREAK_LOOP_RAN_OUT, BREAK_REJECT_COUNTER, BREAK_UNSEEN, BREAK_MODEL_NUMBER = 0, 1, 2, 3


class _Settings(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("max_outer_iterations", C.c_int32), ("pearl_maximum_iteration_number", C.c_int32),
                ("labeling_l0", C.c_int32), ("pearl_abs_int", C.c_int32), ("pad_", C.c_int32),
                ("sample_size", C.c_uint64), ("nonminimal_sample_size", C.c_uint64), ("minimum_number_of_inliers", C.c_uint64),
                ("max_proposal_number_without_change", C.c_uint64), ("maximum_model_number", C.c_uint64),
                ("maximum_tanimoto_similarity", C.c_double), ("one_minus_confidence", C.c_double),
                ("inlier_outlier_threshold", C.c_double), ("spatial_coherence_weight", C.c_double), ("refit_tie_rtol", C.c_double)]


class _Trace(C.Structure):
    _fields_ = [("n_proposals", C.c_int32), ("pad_", C.c_int32), ("models", C.POINTER(C.c_double)), ("empty", C.POINTER(C.c_uint8)),
                ("inlier_off", C.POINTER(C.c_int64)), ("inliers", C.POINTER(C.c_int64)), ("iterations", C.POINTER(C.c_uint64)),
                ("n_refits", C.c_int64), ("refit_inliers", C.POINTER(C.c_int64)), ("refit_models_n", C.POINTER(C.c_int32)),
                ("refit_models", C.POINTER(C.c_double)), ("refit_accepted", C.POINTER(C.c_int8))]


class _Event(C.Structure):
    _fields_ = [("code", C.c_int32), ("pad_", C.c_int32), ("a", C.c_int64), ("b", C.c_int64), ("c", C.c_int64),
                ("x", C.c_double), ("y", C.c_double)]


class ReplayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pgxr_replay: {code}: {msg}")
        self.code = code


class TraceRecorder:
    """The `trace=` hook (pyprogressivex._engine): keeps what the run took from the proposal engine and from the refit solver
    (the replay's inputs) and every decision event the run reported (to be compared with the replay's)."""

    def __init__(self):
        self.info = None
        self.proposals = []     # (descriptor or None, inliers, iterations)
        self.refits = []        # (inlier_number, [models], accepted or None)
        self.events = []        # (code, a, b, c, x, y)
        self.walks = []         # one record per proposal from the proposal engine (oracle/progx_proposal.py replays them)

    def begin(self, info):
        self.info = info

    def proposal(self, model, inliers, iterations):
        self.proposals.append((None if model is None else np.array(model, dtype=np.float64).reshape(-1),
                               None if inliers is None else np.array(inliers, dtype=np.int64).reshape(-1), int(iterations)))

    def refit(self, inlier_number, fits, accepted=None):
        self.refits.append((int(inlier_number), [np.array(f, dtype=np.float64).reshape(-1) for f in fits],
                            None if accepted is None else bool(accepted)))

    def walk(self, rec):
        self.walks.append(rec)

    def event(self, code, a=0, b=0, c=0, x=0.0, y=0.0):
        self.events.append((int(code), int(a), int(b), int(c), float(x), float(y)))


def settings_from(info, max_outer_iterations=None, refit_tie_rtol=0.0):
    """the replay's settings out of what TraceRecorder.begin received (plain attribute reads of the run's MultiModelSettings)"""
    s = info["settings"]
    big = (1 << 64) - 1
    return dict(model_type=int(info["model_type"]),
                max_outer_iterations=int(s.max_outer_iterations if max_outer_iterations is None else max_outer_iterations),
                pearl_maximum_iteration_number=100,
                labeling_l0=0 if s.labeling_l0 == "greedy" else 1,
                pearl_abs_int=1 if getattr(s, "pearl_abs", "double") == "int" else 0,
                sample_size=int(info["sample_size"]), nonminimal_sample_size=int(info["nonminimal_sample_size"]),
                minimum_number_of_inliers=int(s.minimum_number_of_inliers),
                max_proposal_number_without_change=int(s.max_proposal_number_without_change),
                maximum_model_number=min(int(s.maximum_model_number), big),
                maximum_tanimoto_similarity=float(s.maximum_tanimoto_similarity),
                one_minus_confidence=float(s.one_minus_confidence),
                inlier_outlier_threshold=float(s.inlier_outlier_threshold),
                spatial_coherence_weight=float(s.spatial_coherence_weight), refit_tie_rtol=float(refit_tie_rtol))


def replay(settings, pts, graph, proposals, refits, max_events=200000, max_models=256):
    """Runs pgxr_replay.  settings: dict (settings_from); graph: (off, idx, mult) or None; proposals / refits as TraceRecorder
    keeps them.  Returns dict(events=[(code, a, b, c, x, y)], labels int64[n], models [K, p], consumed (proposals, refits))."""
    lib = O.lib()
    fn = lib.pgxr_replay
    fn.restype = C.c_int64
    lib.pgxr_last_error.restype = C.c_char_p
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n = pts.shape[0]
    p = O.PARAM_DIM[settings["model_type"]]
    st = _Settings()
    for k, v in settings.items():
        setattr(st, k, v)
    P = len(proposals)
    models = np.zeros((max(P, 1), p))
    empty = np.zeros(max(P, 1), dtype=np.uint8)
    off = np.zeros(P + 1, dtype=np.int64)
    its = np.zeros(max(P, 1), dtype=np.uint64)
    chunks = []
    for q, (m, inl, it) in enumerate(proposals):
        if m is None:
            empty[q] = 1
            off[q + 1] = off[q]
            continue
        models[q] = m
        chunks.append(np.asarray(inl, dtype=np.int64))
        off[q + 1] = off[q] + len(inl)
        its[q] = it
    inliers = np.ascontiguousarray(np.concatenate(chunks) if chunks else np.zeros(1, dtype=np.int64))
    if inliers.size == 0:
        inliers = np.zeros(1, dtype=np.int64)
    R = len(refits)
    r_inl = np.zeros(max(R, 1), dtype=np.int64)
    r_n = np.zeros(max(R, 1), dtype=np.int32)
    r_m = np.zeros((max(R, 1), p))
    r_acc = np.full(max(R, 1), -1, dtype=np.int8)
    for r, rec in enumerate(refits):
        cnt, fits = rec[0], rec[1]
        r_inl[r] = cnt
        r_n[r] = len(fits)
        if len(fits) >= 1:
            r_m[r] = fits[-1]        # current_models.back()
        if len(rec) > 2 and rec[2] is not None:
            r_acc[r] = 1 if rec[2] else 0
    tr = _Trace(P, 0, O._p(models, C.c_double), O._p(empty, C.c_uint8), O._p(off, C.c_int64), O._p(inliers, C.c_int64),
                O._p(its, C.c_uint64), R, O._p(r_inl, C.c_int64), O._p(r_n, C.c_int32), O._p(r_m, C.c_double), O._p(r_acc, C.c_int8))
    g = None
    if graph is not None:
        g = tuple(np.ascontiguousarray(a, dtype=np.int32) for a in graph)
    ev = (_Event * max_events)()
    labels = np.zeros(n, dtype=np.int64)
    out_models = np.zeros((max_models, p))
    out_n = C.c_int32(0)
    consumed = (C.c_int64 * 2)()
    rc = fn(C.byref(st), O._p(pts, C.c_double), C.c_int64(n),
            None if g is None else O._p(g[0], C.c_int32), None if g is None else O._p(g[1], C.c_int32),
            None if g is None else O._p(g[2], C.c_int32), C.byref(tr), ev, C.c_int64(max_events),
            O._p(labels, C.c_int64), O._p(out_models, C.c_double), C.c_int32(max_models), C.byref(out_n), consumed)
    if rc < 0:
        raise ReplayError(int(rc), lib.pgxr_last_error().decode())
    events = [(e.code, e.a, e.b, e.c, e.x, e.y) for e in ev[:rc]]
    ties = sum(1 for e in events if e[0] == EV_REFIT and e[3] & 4)
    return dict(events=events, labels=labels, models=out_models[:out_n.value].copy(), consumed=(consumed[0], consumed[1]), ties=ties)


def compare_events(got, ref, rtol_sums=1e-9, atol_tanimoto=1e-12):
    """None when the two decision streams agree, else a description of the first difference.  Integers (codes, counts, flags,
    indices) must be equal; energies are 2^-32 fixed-point values and must be EQUAL; the Tanimoto similarity and the residual /
    compound sums are floating-point reductions whose summation order differs between a GPU tree and a sequential loop."""
    for k, (g, r) in enumerate(zip(got, ref)):
        if r[0] == EV_REFIT and r[3] & 4:
            r = r[:3] + (r[3] & 3,) + r[4:]          # a tie the replay resolved the recording's way (refit_tie_rtol): same decision
        if g[:4] != r[:4]:
            return f"event {k}: {EVENT_NAMES.get(g[0], g[0])}{g[1:]} != {EVENT_NAMES.get(r[0], r[0])}{r[1:]}"
        code = g[0]
        for gv, rv, which in ((g[4], r[4], "x"), (g[5], r[5], "y")):
            if np.isnan(gv) and np.isnan(rv):
                continue
            if code == EV_PEARL_ITER:
                ok = gv == rv
            elif code == EV_VALIDATION:
                ok = abs(gv - rv) <= atol_tanimoto + 1e-12 * abs(rv) or (np.isinf(gv) and gv == rv)
            else:
                ok = abs(gv - rv) <= rtol_sums * max(abs(rv), 1e-300) or gv == rv
            if not ok:
                return f"event {k} ({EVENT_NAMES.get(code, code)}{g[1:4]}): {which} = {gv!r} != {rv!r}"
    if len(got) != len(ref):
        k = min(len(got), len(ref))
        longer = got if len(got) > len(ref) else ref
        return f"stream lengths differ: {len(got)} != {len(ref)}; first extra event: {EVENT_NAMES.get(longer[k][0])}{longer[k][1:]}"
    return None


def narrate(events):
    return "\n".join(f"{k:4d} {EVENT_NAMES.get(e[0], e[0]):14s} a={e[1]} b={e[2]} c={e[3]} x={e[4]:.12g} y={e[5]:.12g}"
                     for k, e in enumerate(events))
