#!/usr/bin/env python3
"""Round-2 known-answer vectors (regression fixtures of the BUILD, not of upstream - parity stays unpinned):
the greedy lambda = 0 labelling [U-8] on fixed tables, the PROSAC growth function, sampler draws for fixed seeds, both
getMisclassificationError overloads on a fixed labelling.  Run from the repo root: python tests/golden/make_golden_r2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle")]
import pgx_oracle as O  # noqa: E402
from pyprogressivex import _proposal, datasets  # noqa: E402

out = {}
rng = np.random.default_rng(2026)
for k, (n, L, h) in enumerate(((40, 3, 2.5), (257, 6, 10.0), (64, 4, 0.0))):
    D = rng.integers(0, 2 << 32, (n, L)).astype(np.int64)
    D[np.arange(n), rng.integers(0, L, n)] >>= 6
    lab, e, opened = O.greedy_labeling(D, O.quantize(h))
    out[f"greedy{k}_D"], out[f"greedy{k}_h"], out[f"greedy{k}_labels"] = D, np.array([h]), lab
    out[f"greedy{k}_energy"], out[f"greedy{k}_opened"] = np.array([e]), np.array([opened])
out["prosac_growth_249_7"] = _proposal.prosac_growth_function(249, 7, 100000)
out["prosac_growth_2000_4"] = _proposal.prosac_growth_function(2000, 4, 100000)[::50]
out["prosac_draw"] = _proposal.ProsacSampler(300, np.random.default_rng(1)).draw(400, 4)
pts = np.random.default_rng(3).random((300, 4)) * [640, 480, 640, 480]
out["pnapsac_pts"] = pts
out["pnapsac_draw"] = _proposal.ProgressiveNapsacSampler(300, np.random.default_rng(2), pts, (640, 480, 640, 480), 4).draw(400, 4)
lab = rng.integers(-1, 4, 200)
ann = rng.integers(0, 4, 200)
out["me_lab"], out["me_ann"] = lab, ann
out["me_values"] = np.array([datasets.misclassification(np.where(lab < 0, 3, lab), ann),
                             datasets.misclassification_labeling(lab, ann, 3, 3),
                             datasets.misclassification_models((rng.random((3, 200)) < 0.3).astype(float), ann, 3)])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kat_r2.npz"), **out)
print("wrote kat_r2.npz:", sorted(out))
