#!/usr/bin/env python3
"""three bounded experiments on the bundled cubetoy scene (recorded ME 0.012, adelaideF.ipynb:149-157), on
the CPU-oracle harness (the GPU path returns bit-identical labellings), 10 seeds each, the notebook's exact arguments.
  (i)   so-far-best ranked by inlier count first, MSAC value second            _proposal.U15["rank"] = "count"
  (ii)  the most aggressive early termination a confidence rule could produce   _proposal.U15["stop_at_first"] = True
  (iii) local optimisation variants: graph-cut LO with inner RANSAC on the cut's inliers (shipped) / least-squares refits only
Prints one line per variant; the table goes to docs/experiments-cubetoy.md §5."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")]
from pyprogressivex import _api, _proposal  # noqa: E402
from oracle_ctx import OracleContext  # noqa: E402
import eval_scenes  # noqa: E402

_api._ctx = OracleContext()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variants = [("shipped", {}, {}), ("(i) count first", {"rank": "count"}, {}), ("(ii) stop at first valid best", {"stop_at_first": True}, {}),
            ("(i)+(ii)", {"rank": "count", "stop_at_first": True}, {}), ("(iii) LO = least-squares refits only", {}, {"local_optimization": "lsq"}),
            ("(ii)+(iii)", {"stop_at_first": True}, {"local_optimization": "lsq"})]
for name, u15, extra in variants:
    _proposal.U15.update({"rank": "value", "stop_at_first": False})
    _proposal.U15.update(u15)
    for scene in ("cubetoy", "breadcube", "book"):
        r = [eval_scenes.two_view_scene(scene, s, **extra) for s in range(seeds)]
        me = [round(float(x[0]), 3) for x in r]
        print(f"{name:40s} {scene:10s} median {np.median(me):.3f}  <= 0.036: {sum(m <= 0.036 for m in me)}/{seeds}  per seed {me}  models {[x[1] for x in r]}", flush=True)
