"""ctypes view of oracle/progx_proposal.c - the independent restatement of one GC-RANSAC proposal as Progressive-X drives it
(progressive_x.h:294-299, scoring_function_with_compound_model.h:105-106; loop and local optimisation from memory of the absent
submodule: U-9, U-12) - and the comparison of its decisions with the record the product's proposal engine hands to `trace.walk`.

TEST INFRASTRUCTURE ONLY; does not import pyprogressivex.  PARITY UNPINNED."""
import ctypes as C

import numpy as np

import pgx_oracle as O

EV_BEST, EV_LO_ROUND, EV_LO_END, EV_WALK_END, EV_LSQ, EV_FINAL = range(1, 7)
EVENT_NAMES = {1: "BEST", 2: "LO_ROUND", 3: "LO_END", 4: "WALK_END", 5: "LSQ", 6: "FINAL"}


class _Settings(C.Structure):
    _fields_ = [("n", C.c_int64), ("samples", C.c_int64), ("sample_size", C.c_int32), ("nonminimal_sample_size", C.c_int32),
                ("confidence", C.c_double), ("max_iters", C.c_int64), ("min_iters", C.c_int64), ("lo_after", C.c_int64),
                ("every_best", C.c_int32), ("max_cuts", C.c_int32), ("lsq_budget", C.c_int32), ("pad_", C.c_int32)]


class _Event(C.Structure):
    _fields_ = [("code", C.c_int32), ("pad_", C.c_int32), ("a", C.c_int64), ("b", C.c_int64), ("c", C.c_int64), ("x", C.c_double)]


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def replay(rec, max_events=100000):
    """rec: the dict ProposalEngine.run hands to trace.walk.  Returns (events [(code, a, b, c, x)], (rounds, steps) consumed)."""
    fn = O.lib().pgxq_replay
    fn.restype = C.c_int64
    st = _Settings()
    for k in ("n", "samples", "sample_size", "nonminimal_sample_size", "confidence", "max_iters", "min_iters", "lo_after", "max_cuts", "lsq_budget"):
        setattr(st, k, rec[k])
    st.every_best = 1 if rec["every_best"] else 0
    counts = np.ascontiguousarray(rec["counts"], dtype=np.int64)
    scores = np.ascontiguousarray(rec["scores"], dtype=np.float64)
    src = np.ascontiguousarray(rec["src"], dtype=np.int64)
    rounds = rec["rounds"]
    r_inl = np.array([r[0] for r in rounds] + [0], dtype=np.int64)
    r_off = np.cumsum([0] + [len(r[1]) for r in rounds]).astype(np.int64)
    c_cnt = np.concatenate([np.asarray(r[1], dtype=np.int64) for r in rounds] + [np.zeros(1, np.int64)])
    c_sco = np.concatenate([np.asarray(r[2], dtype=np.float64) for r in rounds] + [np.zeros(1)])
    lsq = rec["lsq"]
    q_inl = np.array([q[0] for q in lsq] + [0], dtype=np.int64)
    q_fit = np.array([q[1] for q in lsq] + [0], dtype=np.int64)
    q_cnt = np.array([q[2] for q in lsq] + [0], dtype=np.int64)
    q_sco = np.array([q[3] for q in lsq] + [0.0], dtype=np.float64)
    ev = (_Event * max_events)()
    consumed = (C.c_int64 * 2)()
    got = fn(C.byref(st), C.c_int64(len(counts)), _p(counts, C.c_int64), _p(scores, C.c_double), _p(src, C.c_int64),
             C.c_int64(len(rounds)), _p(r_inl, C.c_int64), _p(r_off, C.c_int64), _p(c_cnt, C.c_int64), _p(c_sco, C.c_double),
             C.c_int64(len(lsq)), _p(q_inl, C.c_int64), _p(q_fit, C.c_int64), _p(q_cnt, C.c_int64), _p(q_sco, C.c_double),
             ev, C.c_int64(max_events), consumed)
    if got < 0:
        raise RuntimeError(f"pgxq_replay: the recording ran out of {'local-optimisation rounds' if got == -1 else 'least-squares steps'} "
                           f"(consumed {consumed[0]} rounds, {consumed[1]} steps)")
    return [(e.code, e.a, e.b, e.c, e.x) for e in ev[:min(got, max_events)]], (int(consumed[0]), int(consumed[1]))


def compare(rec, max_events=100000):
    """None when the product's decisions (rec['events']) are the replay's, else a description of the first difference.  Everything is
    compared exactly: counts, indices and flags are integers, scores are read from the same tables on both sides - except FINAL, whose
    score the product obtains by scoring the returned model once more (with masks): on the GPU that is the same integer accumulation
    (equal bits), on a context that sums in floating point the mask-producing pass may order its sum differently: 1e-12 relative."""
    ref, consumed = replay(rec, max_events)
    got = [tuple(e) for e in rec["events"]]
    for k, (g, r) in enumerate(zip(got, ref)):
        same_x = g[4] == r[4] or (np.isnan(g[4]) and np.isnan(r[4]))
        if not same_x and g[0] == EV_FINAL and r[0] == EV_FINAL:
            same_x = abs(g[4] - r[4]) <= 1e-12 * max(abs(r[4]), 1e-300)
        if tuple(int(v) for v in g[:4]) != tuple(int(v) for v in r[:4]) or not same_x:
            return f"event {k}: product {EVENT_NAMES.get(g[0], g[0])}{tuple(g[1:])} != replay {EVENT_NAMES.get(r[0], r[0])}{tuple(r[1:])}"
    if len(got) != len(ref):
        k = min(len(got), len(ref))
        longer = got if len(got) > len(ref) else ref
        return f"stream lengths differ: product {len(got)}, replay {len(ref)}; first extra: {EVENT_NAMES.get(longer[k][0])}{tuple(longer[k][1:])}"
    if consumed != (len(rec["rounds"]), len(rec["lsq"])):
        return f"the replay consumed {consumed} of ({len(rec['rounds'])}, {len(rec['lsq'])}) recorded rounds / steps"
    return None


class WalkRecorder:
    """a trace hook that keeps the proposal records only (the three required methods do nothing)"""

    def __init__(self):
        self.walks = []

    def proposal(self, *a):
        pass

    def refit(self, *a):
        pass

    def event(self, *a, **k):
        pass

    def walk(self, rec):
        self.walks.append(rec)
