import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "progressive-x_amd")]
import numpy as np
from pyprogressivex import _lib, datasets
rng = np.random.default_rng(0)
p2, g2, models = datasets.make_vanishing_points(seed=0)
c2 = _lib.Context(0)
c2.set_points(_lib.VANISHING_POINT, p2)
T2b = 2.25 * 1.5 * 1.5
c2.preference(np.asarray(models[0]).reshape(-1), T2b, slot=0)
c2.compound_update([0])
K2 = int(g2.max())
smp = np.array([rng.choice(np.nonzero(g2 == 1 + r % K2)[0], 2, replace=False) if r % 2 == 0 else rng.choice(len(g2), 2, replace=False) for r in range(2048)], dtype=np.int32)
c2.solve_minimal(smp, fetch=False)
c2.score_launch(T2b, has_compound=True)
out = c2.score_fetch(exponent=2)
print("sum counts", int(out["counts"].sum()), "max", int(out["counts"].max()), "n", len(p2))
print(c2.score_stats(T2b, has_compound=True))
srt = np.sort(out["counts"])[::-1]
print("top counts", srt[:5], "median", srt[len(srt)//2], "groups sizes", np.bincount(g2))
