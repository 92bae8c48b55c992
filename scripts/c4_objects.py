#!/usr/bin/env python3
"""BASELINE config C4 names 16 objects; the reference's outer loop is capped at 10 proposals (progressive_x.h:272).  This runs
find6DPoses on the C4 scene with the cap lifted (max_outer_iterations) and prints why the loop ended and what it kept."""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle")]
import pyprogressivex as px  # noqa: E402
from pyprogressivex import datasets  # noqa: E402
import progx_replay as R  # noqa: E402  (the recorder only: TEST / diagnostics infrastructure)

x1, x2, K, gt, poses = datasets.make_poses(seed=0)
for cap in [int(a) for a in sys.argv[1:]] or [20, 32]:
    rec = R.TraceRecorder()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        P, lab = px.find6DPoses(x1, x2, K, seed=1, minimum_point_number=5000, max_iters=int(os.environ.get("C4_MAX_ITERS", "2048")), max_outer_iterations=cap, trace=rec, conf=float(os.environ.get("C4_CONF", "0.9")), maximum_tanimoto_similarity=float(os.environ.get("C4_TANI", "0.9")), scoring_exponent=int(os.environ.get("C4_EXPONENT", "2")))
        dt = time.perf_counter() - t0
    k = P.shape[0] // 3
    verdicts = [(e[1], e[2]) for e in rec.events if e[0] == R.EV_VALIDATION]
    brk = [e[1] for e in rec.events if e[0] == R.EV_BREAK]
    unseen = [e[2] for e in rec.events if e[0] == R.EV_UNSEEN]
    props = [e[1] for e in rec.events if e[0] == R.EV_PROPOSAL]
    me = float(datasets.misclassification(np.where(lab == k, 0, lab + 1), gt))
    found = sorted(set(int(np.bincount(gt[lab == m], minlength=17)[1:].argmax()) + 1 for m in range(k)))
    print(json.dumps(dict(max_outer_iterations=cap, wall_s=round(dt, 3), models=k, misclassification=round(me, 4), break_reason=brk,
                          proposals=len(props), accepted=sum(v[0] for v in verdicts), rejected_by=[v[1] for v in verdicts if not v[0]],
                          proposal_inliers=props, unseen=unseen, distinct_objects_found=len(found))), flush=True)
