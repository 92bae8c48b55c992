"""Step time and kernel times of the metric batch (6D poses 1e6 x 2048) over the group kernel's launch geometry.
usage: python scripts/sweep_score.py [split ...]   (0 = the default)"""
import os, sys, time, zlib
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "progressive-x_amd")]
import numpy as np
from pyprogressivex import _lib, datasets

# arguments: plain integers = waves per group of score_group_kernel; key=value sets any geometry knob of pgx_score_debug_geometry
# for that run (e.g. "transposed=1,tsplit=3,dense_min=24")
splits = [a for a in sys.argv[1:]] or ["0"]
x1, x2, K, lab, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f
T2 = 9.0 / 4.0 * thr * thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
ctx = _lib.Context(0)
ctx.set_points(_lib.PNP, pts)
ctx.preference(gt[0], T2, slot=0)
ctx.compound_update([0])
ctx.score_upload(hyps)
buf = ctx.score_buffers()
ref = None
for sp in splits:
    if "=" in sp:
        ctx.score_debug_geometry(**{k: int(v) for k, v in (kv.split("=") for kv in sp.split(","))})
    else:
        ctx.score_debug_geometry(split=int(sp))
    ctx.score_profile(0)
    for _ in range(300):
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(300):
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 300 * 1e3
    ctx.score_profile(2)
    kts = []
    for _ in range(10):
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
        kts.append(ctx.score_kernel_times())
    kt = np.median(np.array(kts), axis=0)
    crc = zlib.crc32(res["counts"].tobytes() + res["scores"].tobytes())
    if ref is None:
        ref = crc
    print(f" split={sp} step={ms:.4f} ms  cull={kt[0]*1e3:.1f} group={kt[1]*1e3:.1f} finish={kt[2]*1e3:.1f} us  crc={crc:08x} {'same' if crc == ref else 'DIFFERENT'}", flush=True)
