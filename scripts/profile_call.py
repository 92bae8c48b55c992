#!/usr/bin/env python3
"""cProfile of one drop-in call at a BASELINE shape (bench.py's api-leg arguments): totals per libpgx entry point and per host function.
usage: profile_call.py C1|C2|C3|C5|C4 [top]"""
import cProfile, contextlib, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
import pyprogressivex as px
from pyprogressivex import datasets
which = sys.argv[1] if len(sys.argv) > 1 else "C3"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if which == "C5":
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    f = lambda: px.findVanishingPoints(pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1, minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
elif which == "C2":
    pts, gt, _ = datasets.make_homographies(seed=0)
    f = lambda: px.findHomographies(pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
elif which == "C1":
    pts, gt, _ = datasets.make_lines(seed=0)
    f = lambda: px.findLines(pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
elif which == "C4":
    x1, x2, K, gt = datasets.make_poses(seed=0)[:4]
    f = lambda: px.find6DPoses(x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048)
else:
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    f = lambda: px.findTwoViewMotions(pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1, minimum_point_number=1000, max_iters=2000)
def call():
    with contextlib.redirect_stdout(io.StringIO()):
        return f()
call()
t0 = time.perf_counter(); call(); print(which, "wall ms %.1f" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
call()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(top)
print(out.getvalue()[:7000])
