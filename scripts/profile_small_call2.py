#!/usr/bin/env python3
"""cProfile of one drop-in call on a bundled scene, cumulative view (who calls what); companion of profile_small_call.py.
usage: profile_small_call2.py [scene] [top]"""
import cProfile, contextlib, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))
import pyprogressivex as px
from pyprogressivex import datasets
scene = sys.argv[1] if len(sys.argv) > 1 else "unionhouse"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
corrs, g = datasets.load_points_with_labels(os.path.join(ROOT, "tests", "golden", "scenes", f"{scene}.txt"))
kw = dict(threshold=4.0, conf=0.5, spatial_coherence_weight=0.05, neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
          minimum_point_number=10, maximum_model_number=6, scoring_exponent=2, sampler_id=3)
def call(seed):
    with contextlib.redirect_stdout(io.StringIO()):
        return px.findHomographies(corrs, 1024, 768, 1024, 768, seed=seed, **kw)
call(0)
pr = cProfile.Profile()
pr.enable()
for s in range(5):
    call(s)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(top)
print(out.getvalue()[:9000])
