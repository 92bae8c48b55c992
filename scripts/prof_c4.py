"""cProfile of find6DPoses at C4 (1e6 correspondences) and findVanishingPoints at C5.  usage: python scripts/prof_c4.py [C4] [C5]"""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
import numpy as np
import pyprogressivex as px
from pyprogressivex import datasets
px.findLines(np.random.default_rng(0).random((50, 2)) * 100, np.array(0), 100, 100, sampler_id=0, seed=0)
which = sys.argv[1:] or ["C4", "C5"]
def prof(name, fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        fn(*a, **kw)
    ts = []
    for _ in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter(); fn(*a, **kw); ts.append(time.perf_counter() - t0)
    print(name, "wall", [round(t, 4) for t in ts])
    pr = cProfile.Profile(); pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        fn(*a, **kw)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
if "C4" in which:
    x1, x2, K, gt, poses = datasets.make_poses(seed=0)
    prof("C4", px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048)
if "C5" in which:
    pts, gt, _ = datasets.make_vanishing_points(seed=0)
    prof("C5", px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1, minimum_point_number=2000,
         spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
