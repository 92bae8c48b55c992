#!/usr/bin/env python3
"""Where a drop-in call on one of the reference's own scenes spends its wall time (VERDICT r5 missing 2 / weak 5): cProfile of
findHomographies on unionhouse with the notebook's arguments, totals per libpgx entry point (ctypes call sites in _lib.py) and
per host function.  usage: profile_small_call.py [scene]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
import pyprogressivex as px
from pyprogressivex import datasets

scene = sys.argv[1] if len(sys.argv) > 1 else "unionhouse"
corrs, g = datasets.load_points_with_labels(os.path.join(ROOT, "tests", "golden", "scenes", f"{scene}.txt"))
kw = dict(threshold=4.0, conf=0.5, spatial_coherence_weight=0.05, neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
          minimum_point_number=10, maximum_model_number=6, scoring_exponent=2, sampler_id=3)


def call(seed):
    with contextlib.redirect_stdout(io.StringIO()):
        return px.findHomographies(corrs, 1024, 768, 1024, 768, seed=seed, **kw)


call(0)
ts = []
for s in range(5):
    t0 = time.perf_counter(); call(s); ts.append(time.perf_counter() - t0)
print("wall ms per call (seeds 0-4):", [round(1e3 * t, 1) for t in ts], "points", len(corrs))
pr = cProfile.Profile()
pr.enable()
for s in range(5):
    call(s)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(r"_lib.py")
print(out.getvalue()[:5000])
