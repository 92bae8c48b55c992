#!/usr/bin/env python3
"""findVanishingPoints at C5 with bench.py's api-leg arguments, wall time of 3 calls: for rocprofv3 kernel stats and A/B runs.
usage: c5_call.py [C5|C2|C3]"""
import contextlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
import pyprogressivex as px
from pyprogressivex import datasets
which = sys.argv[1] if len(sys.argv) > 1 else "C5"
if which == "C5":
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    f = lambda: px.findVanishingPoints(pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1, minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
elif which == "C2":
    pts, gt, _ = datasets.make_homographies(seed=0)
    f = lambda: px.findHomographies(pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
else:
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    f = lambda: px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1, minimum_point_number=1000, max_iters=2000)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        f()
    ts.append(time.perf_counter() - t0)
print(which, "wall ms", [round(1e3 * t, 1) for t in ts])
