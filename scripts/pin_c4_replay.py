#!/usr/bin/env python3
"""The C4 full-size decision stream as a committed pin: `find6DPoses` on 1e6 correspondences (the metric's
configuration) agrees with the independent control-flow oracle decision by decision - 35 minutes of host Dinic for its 23 labellings
of 1e6 sites, too long for the GPU suite.  Two steps:
  record  (GPU box, ~2 s):   run the call with the decision trace recorded; write the trace - what the run took from the proposal
                             engine and the refit solver, its events, labels, models - to gpurun_out/c4_trace.npz
  replay  (any host, ~35 min, no GPU): oracle/progx_replay.c recomputes every decision from the points and the recorded
                             proposals / refits (its labellings by the oracle's Dinic expansion), compares with the recorded events
                             and writes tests/golden/kat_c4_replay_v1.npz: the ORACLE's event stream, SHA-256 of its labels, its
                             models, and SHA-256 of the replay's inputs (so that a run whose proposals differ is told apart
                             from a run whose decisions differ)
tests/test_gpu_replay.py::test_c4_full_size_events_match_the_committed_replay_pin runs the call again and compares."""
import contextlib
import hashlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
TRACE = os.path.join(ROOT, "gpurun_out", "c4_trace.npz")
PIN = os.path.join(ROOT, "tests", "golden", "kat_c4_replay_v1.npz")
ARGS = dict(seed=1, minimum_point_number=5000, max_iters=2048)


def inputs_digest(proposals, refits):
    """SHA-256 over the replay's inputs exactly as TraceRecorder keeps them"""
    h = hashlib.sha256()
    for model, inliers, iterations in proposals:
        h.update(b"P" + (b"-" if model is None else np.ascontiguousarray(model, dtype=np.float64).tobytes()))
        h.update(b"-" if inliers is None else np.ascontiguousarray(inliers, dtype=np.int64).tobytes())
        h.update(int(iterations).to_bytes(8, "little", signed=True))
    for inlier_number, fits, accepted in refits:
        h.update(b"R" + int(inlier_number).to_bytes(8, "little", signed=True))
        for f in fits:
            h.update(np.ascontiguousarray(f, dtype=np.float64).tobytes())
        h.update(b"?" if accepted is None else (b"1" if accepted else b"0"))
    return h.hexdigest()


def pack(proposals, refits):
    """flat arrays for np.savez (ragged lists -> values + offsets)"""
    pm = np.array([np.full(12, np.nan) if m is None else m for m, _, _ in proposals], dtype=np.float64).reshape(len(proposals), -1)
    pnone = np.array([m is None for m, _, _ in proposals], dtype=np.uint8)
    pin = [np.zeros(0, np.int64) if i is None else i for _, i, _ in proposals]
    poff = np.cumsum([0] + [len(i) for i in pin]).astype(np.int64)
    pit = np.array([it for _, _, it in proposals], dtype=np.int64)
    rn = np.array([r[0] for r in refits], dtype=np.int64)
    rcount = np.array([len(r[1]) for r in refits], dtype=np.int64)
    rfits = np.array([f for r in refits for f in r[1]], dtype=np.float64).reshape(int(rcount.sum()), -1)
    racc = np.array([-1 if r[2] is None else int(r[2]) for r in refits], dtype=np.int8)
    return dict(p_models=pm, p_none=pnone, p_inliers=np.concatenate(pin) if pin else np.zeros(0, np.int64), p_off=poff, p_iterations=pit,
                r_inlier_number=rn, r_count=rcount, r_fits=rfits, r_accepted=racc)


def unpack(z):
    proposals, refits = [], []
    for k in range(len(z["p_iterations"])):
        none = bool(z["p_none"][k])
        proposals.append((None if none else z["p_models"][k].copy(), None if none else z["p_inliers"][z["p_off"][k]:z["p_off"][k + 1]].copy(),
                          int(z["p_iterations"][k])))
    at = 0
    for k in range(len(z["r_inlier_number"])):
        c = int(z["r_count"][k])
        refits.append((int(z["r_inlier_number"][k]), [z["r_fits"][at + j].copy() for j in range(c)],
                       None if z["r_accepted"][k] < 0 else bool(z["r_accepted"][k])))
        at += c
    return proposals, refits


def scene():
    from pyprogressivex import datasets
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    return x1, x2, K


def record():
    import pyprogressivex as px
    import progx_replay as R
    x1, x2, K = scene()
    rec = R.TraceRecorder()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        models, labels = px.find6DPoses(x1, x2, K, trace=rec, **ARGS)
    dt = time.perf_counter() - t0
    s = rec.info["settings"]
    settings = R.settings_from(rec.info, refit_tie_rtol=1e-12)
    os.makedirs(os.path.dirname(TRACE), exist_ok=True)
    np.savez_compressed(TRACE, events=np.array(rec.events, dtype=np.float64), labels=np.asarray(labels, dtype=np.int16), models=models,
                        settings=json.dumps(settings), neighborhood=rec.info["neighborhood"], radius=float(rec.info.get("radius", 0.0)),
                        inputs_sha256=inputs_digest(rec.proposals, rec.refits),
                        points_sha256=hashlib.sha256(np.ascontiguousarray(rec.info["points"], dtype=np.float64).tobytes()).hexdigest(),
                        graph_points_sha256=hashlib.sha256(np.ascontiguousarray(rec.info["graph_points"], dtype=np.float64).tobytes()).hexdigest(),
                        **pack(rec.proposals, rec.refits))
    print(json.dumps(dict(step="record", seconds=round(dt, 2), models=int(models.shape[0] // 3), events=len(rec.events), proposals=len(rec.proposals),
                          refits=len(rec.refits), inputs_sha256=inputs_digest(rec.proposals, rec.refits), lam=float(s.spatial_coherence_weight))))


def replay():
    import pgx_oracle as O
    import progx_replay as R
    from pyprogressivex import datasets
    z = np.load(TRACE, allow_pickle=False)
    proposals, refits = unpack(z)
    assert inputs_digest(proposals, refits) == str(z["inputs_sha256"])
    x1, x2, K = scene()
    pts, f = datasets.normalize_pnp(x1, x2, K)          # what the run's `points` are (progressivex_python.cpp:96-98)
    graph_pts = np.column_stack([x1, x2])
    assert hashlib.sha256(np.ascontiguousarray(pts, dtype=np.float64).tobytes()).hexdigest() == str(z["points_sha256"])
    assert hashlib.sha256(np.ascontiguousarray(graph_pts, dtype=np.float64).tobytes()).hexdigest() == str(z["graph_points_sha256"])
    settings = json.loads(str(z["settings"]))
    t0 = time.perf_counter()
    nb = str(z["neighborhood"])
    graph = O.graph_build(graph_pts, 0, radius=float(z["radius"]), k=5) if not nb.startswith("knn:") and nb != "radius" else None
    assert graph is not None, nb
    print(json.dumps(dict(step="graph", seconds=round(time.perf_counter() - t0, 1), arcs=int(graph[0][-1]))), flush=True)
    rep = R.replay(settings, pts, graph, proposals, refits)
    events = [tuple(int(v) for v in e[:4]) + (float(e[4]), float(e[5])) for e in z["events"]]
    diff = R.compare_events(events, rep["events"])
    labels_ok = np.array_equal(z["labels"].astype(np.int64), rep["labels"])
    out = dict(step="replay", seconds=round(time.perf_counter() - t0, 1), events=len(rep["events"]), first_difference=diff, labels_equal=bool(labels_ok),
               models=int(rep["models"].shape[0]), ties_followed=int(rep["ties"]), consumed=[int(c) for c in rep["consumed"]])
    print(json.dumps(out), flush=True)
    assert diff is None and labels_ok and rep["consumed"] == (len(proposals), len(refits))
    np.savez_compressed(PIN, oracle_events=np.array(rep["events"], dtype=np.float64),
                        labels_sha256=hashlib.sha256(rep["labels"].astype(np.int64).tobytes()).hexdigest(),
                        label_histogram=np.bincount(rep["labels"].astype(np.int64), minlength=rep["models"].shape[0] + 1),
                        models=rep["models"], inputs_sha256=str(z["inputs_sha256"]), ties_followed=int(rep["ties"]),
                        call=json.dumps(dict(fn="find6DPoses", scene="datasets.make_poses(seed=0)", **ARGS)),
                        generator="scripts/pin_c4_replay.py record (GPU) + replay (host, oracle/progx_replay.c + Dinic expansion)")
    print("wrote", PIN)


if __name__ == "__main__":
    {"record": record, "replay": replay}[sys.argv[1]]()
