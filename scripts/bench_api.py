#!/usr/bin/env python3
"""End-to-end wall time of the drop-in API on the BASELINE configs (synthetic data of SURVEY §8d) with the stage timers
the reference keeps in MultiModelStatistics (progressive_x.h:84-104).  One JSON line per config."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd")]
import pyprogressivex as px  # noqa: E402
from pyprogressivex import _api, datasets  # noqa: E402


def me(labels, K, gt):
    return float(datasets.misclassification(np.where(labels == K, 0, labels + 1), gt))


def timed(name, fn, gt, rows, *args, **kw):
    stats = {}
    orig = _api._engine.ProgressiveX.run

    def run(self):
        models, st = orig(self)
        stats.update(proposal=st.total_time_of_proposal_engine, validation=st.total_time_of_model_validation,
                     optimization=st.total_time_of_optimization, compound=st.total_time_of_compound_model_calculation,
                     pearl_iterations=st.pearl_iterations, expansion_cycles=st.expansion_cycles, core=st.processing_time)
        return models, st
    _api._engine.ProgressiveX.run = run
    acc = {}
    if os.environ.get("BENCH_MF_STATS"):   # accumulate the max-flow counters of every PEARL labelling call
        from pyprogressivex import _lib
        orig_exp = _lib.Context.expansion

        def expansion(self, *a, **k):
            r = orig_exp(self, *a, **k)
            for key, val in self.expansion_stats().items():
                acc[key] = acc.get(key, 0) + val
            return r
        _lib.Context.expansion = expansion
        stats["maxflow"] = acc
    kw.setdefault("local_optimization", os.environ.get("BENCH_LO", "auto"))   # "lsq": refit-only local optimisation
    kw.setdefault("labeling_l0", os.environ.get("BENCH_L0", "greedy"))        # U-8 switch (lambda = 0 only)
    t0 = time.perf_counter()
    models, labels = fn(*args, **kw)
    dt = time.perf_counter() - t0
    _api._engine.ProgressiveX.run = orig
    K = models.shape[0] // rows
    print(json.dumps(dict(config=name, n=len(labels), models=K, misclassification=me(labels, K, gt), wall_s=dt,
                          **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()})), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C1", "C2", "C3", "C5", "C4"]
    if "--oracle" in sys.argv:   # the same host code on the single-threaded CPU port (test infrastructure) instead of libpgx
        sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
        from oracle_ctx import OracleContext
        _api._ctx = OracleContext()
    px.findLines(np.random.default_rng(0).random((50, 2)) * 100, np.array(0), 100, 100, sampler_id=0, seed=0)  # warm up
    if "C1" in which:
        pts, gt, _ = datasets.make_lines(seed=0)
        timed("C1 lines 2k/3", px.findLines, gt, 1, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0,
              seed=1, minimum_point_number=50)
    if "C2" in which:
        pts, gt, _ = datasets.make_homographies(seed=0)
        timed("C2 homographies 5k/5", px.findHomographies, gt, 3, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99,
              sampler_id=0, seed=1, minimum_point_number=50)
    if "C3" in which:
        pts, gt, _ = datasets.make_two_view_motions(seed=0)
        timed("C3 two-view 1e5/8", px.findTwoViewMotions, gt, 3, pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99,
              sampler_id=0, seed=1, minimum_point_number=1000, max_iters=2000)
    if "C5" in which:
        pts, gt, _ = datasets.make_vanishing_points(seed=0)
        timed("C5 vanishing points 2e5/6", px.findVanishingPoints, gt, 1, pts, np.array(0), 1000, 1000, threshold=1.5,
              conf=0.99, sampler_id=0, seed=1, minimum_point_number=2000, spatial_coherence_weight=0.05,
              neighborhood_ball_radius=10.0)
    if "C4" in which:
        x1, x2, K, gt, poses = datasets.make_poses(seed=0)
        timed("C4 6D poses 1e6/16 (cap 10)", px.find6DPoses, gt, 3, x1, x2, K, seed=1, minimum_point_number=5000,
              max_iters=2048)
    if "C4all" in which:   # the same scene with the reference's hard-wired cap of 10 proposals (progressive_x.h:272) lifted: all 16 objects
        x1, x2, K, gt, poses = datasets.make_poses(seed=0)
        timed("C4 6D poses 1e6/16 (max_outer_iterations=20)", px.find6DPoses, gt, 3, x1, x2, K, seed=1, minimum_point_number=5000,
              max_iters=2048, max_outer_iterations=20)
