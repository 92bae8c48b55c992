"""Regenerates tests/golden/kat_v1.npz from the CPU oracle (run in the build container: `python tests/golden/make_golden.py`).

The reference holds no golden vectors for this path, and it can be neither built nor imported here,
so these vectors pin the oracle against ITSELF over time (regression) and CPU<->GPU; the hand-checkable and
exact-rational cases that pin the oracle's formulas against the mathematics live in tests/test_oracle.py.
Fixture = data only: seeded inputs + the oracle's outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "tests")]
import pgx_oracle as O  # noqa: E402
from helpers import MODEL_CASES, make_case, realistic_labeling_problem  # noqa: E402

out = {}
for name, mt in MODEL_CASES.items():
    mt, pts, models, thr = make_case(name, 257, 5, seed=42)
    T2 = 2.25 * thr * thr
    comp = np.random.default_rng(3).uniform(0, 1, 257) * (np.arange(257) % 3 == 0)
    sc = O.score(mt, pts, models, T2, compound=comp, has_compound=True, exponent=2, want_masks=True)
    out[f"{name}_pts"] = pts
    out[f"{name}_models"] = models
    out[f"{name}_thr"] = np.array([thr])
    out[f"{name}_comp"] = comp
    out[f"{name}_sq0"] = O.squared_residuals(mt, pts, models[0])
    out[f"{name}_counts"] = sc["counts"]
    out[f"{name}_values"] = sc["values"]
    out[f"{name}_shared"] = sc["shared"]
    out[f"{name}_scores"] = sc["scores"]
    out[f"{name}_masks"] = sc["masks"]
    out[f"{name}_pref0"] = O.preference(mt, pts, models[0], T2)
    out[f"{name}_unary_q"] = O.unary_q(mt, pts, models[:3], thr, 0.1)
Dq, graph = realistic_labeling_problem(400, L=5, lam=0.25, seed=5)
lq, hq = O.quantize_lambda(0.25), O.quantize(6.0)
labels, e, cyc = O.expansion(Dq, graph, lq, hq, np.zeros(400, np.int32))
out.update(exp_Dq=Dq, exp_off=graph[0], exp_idx=graph[1], exp_mult=graph[2], exp_labels=labels,
           exp_energy=np.array([e]), exp_cycles=np.array([cyc]), exp_lq=np.array([lq]), exp_hq=np.array([hq]))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_v1.npz"), **out)
print("wrote kat_v1.npz with", len(out), "arrays")
