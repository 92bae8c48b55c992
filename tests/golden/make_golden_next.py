"""Regenerates tests/golden/kat_next_v1.npz: seeded inputs + the oracle's outputs for the SURVEY 8f rows built in round 1
(neighbourhood graph, Gram pass of the refits, minimal solvers).  Run in the build container:
`python tests/golden/make_golden_next.py`.  Same status as kat_v1.npz: the reference holds no vectors for these (its
FLANN graph and its solvers are absent from the snapshot), so they pin the oracle against itself over time and
CPU<->GPU; hand-checkable cases live in tests/test_oracle.py.  Fixture = data only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "tests")]
import pgx_oracle as O  # noqa: E402
from helpers import MODEL_CASES, make_case  # noqa: E402

out = {}
rng = np.random.default_rng(11)
for d in (2, 4, 5):
    pts = np.round(rng.random((300, d)) * 40.0, 1)          # one decimal: plenty of exact distance ties
    out[f"g{d}_pts"] = pts
    for tag, kind, radius, k in (("ball", 0, 6.0, 5), ("knn", 2, 0.0, 8)):
        off, idx, mult = O.graph_build(pts, kind, radius=radius, k=k)
        out[f"g{d}_{tag}_off"], out[f"g{d}_{tag}_idx"], out[f"g{d}_{tag}_mult"] = off, idx, mult
prm = np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0])
for name, kinds in (("line", [(O.GRAM_AFFINE, None)]), ("vanishing_point", [(O.GRAM_VP, None)]),
                    ("homography", [(O.GRAM_AFFINE, None), (O.GRAM_DLT_H, prm)]), ("fundamental", [(O.GRAM_EPI_F, prm)]),
                    ("pnp", [(O.GRAM_PNP_GN, "model")])):
    mt, pts, models, thr = make_case(name, 200, 2, seed=21)
    idx = rng.permutation(200)[:120]
    w = rng.random(200) + 0.5
    out[f"m_{name}_pts"], out[f"m_{name}_idx"], out[f"m_{name}_w"], out[f"m_{name}_model"] = pts, idx, w, models[0]
    for kind, p in kinds:
        p = models[0][:12] if isinstance(p, str) else p
        G, cnt, bad = O.gram(kind, pts, idx, params=p, weights=w, wpow=2)
        out[f"m_{name}_G{kind}"] = G
for name in ("line", "vanishing_point", "homography", "fundamental", "pnp"):
    mt, pts, models, thr = make_case(name, 200, 2, seed=31)
    smp = rng.integers(0, 200, (64, {"homography": 4, "fundamental": 7, "pnp": 3}.get(name, 2))).astype(np.int32)
    smp[:4, 1] = smp[:4, 0]
    out[f"s_{name}_pts"], out[f"s_{name}_samples"] = pts, smp
    out[f"s_{name}_models"] = O.solve_minimal(mt, pts, smp)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_next_v1.npz"), **out)
print("wrote kat_next_v1.npz with", len(out), "arrays")
