#!/usr/bin/env python3
"""Full-size replays of the drop-in calls against the independent control-flow oracle (oracle/progx_replay.c): the GPU runs the call
with the decision trace recorded, the replay recomputes every decision on the host (its labellings are the oracle's Dinic / greedy
solvers: minutes at these sizes - a one-off campaign, not a test).  usage: python scripts/replay_fullsize.py C3 C5"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import pyprogressivex as px
from pyprogressivex import datasets
import progx_replay as R
import replay_helpers as H

def run(name, rows, fn, *a, **kw):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        out, rec, rep = H.run_and_replay(fn, *a, refit_tie_rtol=1e-12, **kw)
    dt = time.perf_counter() - t0
    diff = R.compare_events(rec.events, rep["events"])
    ok = diff is None and np.array_equal(np.asarray(out[1], dtype=np.int64), rep["labels"]) and out[0].shape[0] // rows == rep["models"].shape[0]
    print(json.dumps(dict(config=name, points=int(len(out[1])), models=int(out[0].shape[0] // rows), decision_events=len(rec.events),
                          pearl_iterations=sum(e[0] == R.EV_PEARL_ITER for e in rec.events), refits=len(rec.refits), ties_followed=rep["ties"],
                          agree=bool(ok), first_difference=diff, seconds=round(dt, 1))), flush=True)

which = sys.argv[1:] or ["C3", "C5"]
px.findLines(np.random.default_rng(0).random((50, 2)) * 100, np.array(0), 100, 100, sampler_id=0, seed=0)
if "C3" in which:
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    run("C3 findTwoViewMotions 1e5/8", 3, px.findTwoViewMotions, pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1,
        minimum_point_number=1000, max_iters=2000)
if "C5" in which:
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    run("C5 findVanishingPoints 2e5/6", 1, px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1,
        minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
if "C4" in which:
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    run("C4 find6DPoses 1e6/16 (cap 10)", 3, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048)
