import sys,time; sys.path[:0]=['/root/repo/progressive-x_amd']
import numpy as np
from pyprogressivex import _lib, datasets
x1, x2, K, _, gt = datasets.make_poses(seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
ctx=_lib.Context(0)
for i in range(3):
    t=time.perf_counter(); ctx.set_points(_lib.PNP, pts); print("set_points ms", 1e3*(time.perf_counter()-t))
