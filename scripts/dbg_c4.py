"""Diagnostics for single expansion moves at C4 scale (PGX_MF_DEBUG=1|2|3 prints one line per global relabel).
mode 'steady': near-converged labelling; mode 'newlabel': one object's points are still outliers and its model is
expanded for the first time (the move that hands a new instance its points)."""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'progressive-x_amd')]
import numpy as np
from pyprogressivex import _lib, datasets
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import host_graph as _graph
mode = sys.argv[1] if len(sys.argv) > 1 else "steady"
x1, x2, K, gt, poses = datasets.make_poses(seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
g = _graph.flann_like_graph(np.column_stack([x1, x2]), 20.0)
ctx = _lib.Context(0); ctx.set_points(_lib.PNP, pts); ctx.set_graph(*g)
lam, h = 0.1, float(sys.argv[2]) if len(sys.argv) > 2 else 5000.0
ctx.pearl_unary(poses[:9], 4.0/f, lam)
lab = np.where(gt == 0, 9, np.minimum(gt - 1, 9)).astype(np.int32)   # GT (objects >= 10 -> outlier)
alphas = (3, 9)
if mode == "newlabel":
    lab[lab == 4] = 9
    alphas = (4,)
ctx.set_labels(lab)
for alpha in alphas:
    t=time.perf_counter(); ch = ctx.expand_alpha(lam, h, alpha); print("alpha", alpha, "changed", ch, "ms", 1e3*(time.perf_counter()-t), ctx.expansion_stats(), flush=True)
