"""Sampson group bound in (position, flow) coordinates - CPU prototype (round 6, after the Hough order did it for vanishing points).
Correspondences of one rigid motion have a flow w = x_b - x_a that varies smoothly with x_a, and for a flow-like epipolar geometry
grad_a n ~ -grad_b n: in coordinates (x_a, w) a group is tight in w and the first-order term is |g_a + g_b| r_a + |g_b| r_w instead of
|g_a| r_a + |g_b| r_b.  Prices: Morton(x_a, x_b) + ball bound (shipped), Morton(x_a, w) + ball bound, Morton(x_a, w) + flow bound,
and the floor (pairs whose group holds an inlier), on the C3 set with 7-point hypotheses (half from one motion's inliers)."""
import sys
import numpy as np
sys.path.insert(0, "progressive-x_amd")
from pyprogressivex import datasets, _estimators

pts, gt, models = datasets.make_two_view_motions(seed=0)
n = len(pts)
rng = np.random.default_rng(1)
K = int(gt.max())
S = 200
smp = np.array([rng.choice(np.nonzero(gt == 1 + r % K)[0], 7, replace=False) if r % 2 == 0 else rng.choice(n, 7, replace=False)
                for r in range(S)], dtype=np.int32)
Fs = np.asarray(_estimators.FundamentalEstimator().minimal(pts, smp)[0]).reshape(-1, 9)
Fs = Fs[np.isfinite(Fs).all(1) & (np.abs(Fs).sum(1) > 0)]
T = 1.5 * 0.75 * (1 + 1 / 64)


def morton(p, bits):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    d = p.shape[1]
    key = np.zeros(len(p), dtype=np.int64)
    for k in range(d):
        for b in range(bits):
            key |= ((q[:, k] >> b) & 1) << (b * d + d - 1 - k)
    return key


def evaluate(order, label, flow_bound):
    sp = pts[order]
    G = n // 64
    P = sp[:G * 64].reshape(G, 64, 4)
    xa, xb = P[:, :, :2], P[:, :, 2:]
    w = xb - xa
    ca = 0.5 * (xa.min(1) + xa.max(1)); cb = 0.5 * (xb.min(1) + xb.max(1)); cw = 0.5 * (w.min(1) + w.max(1))
    ra = np.sqrt(((xa - ca[:, None]) ** 2).sum(2)).max(1)
    rb = np.sqrt(((xb - cb[:, None]) ** 2).sum(2)).max(1)
    rw = np.sqrt(((w - cw[:, None]) ** 2).sum(2)).max(1)
    R = np.sqrt(((xa - ca[:, None]) ** 2).sum(2) + ((xb - cb[:, None]) ** 2).sum(2)).max(1)
    kept = floor = 0
    for F in Fs:
        F = F.reshape(3, 3)
        A = np.linalg.norm(F[:2, :2], 2)
        # all points: exact Sampson
        X1 = np.concatenate([xa, np.ones((G, 64, 1))], 2); X2 = np.concatenate([xb, np.ones((G, 64, 1))], 2)
        L1 = X2 @ F; L2 = X1 @ F.T
        r = np.abs((L1 * X1).sum(2)) / np.sqrt(L1[..., 0] ** 2 + L1[..., 1] ** 2 + L2[..., 0] ** 2 + L2[..., 1] ** 2)
        floor += np.count_nonzero((r < 1.5 * 0.75).any(1))
        if flow_bound:
            c1 = np.column_stack([ca, np.ones(G)]); c2 = np.column_stack([ca + cw, np.ones(G)])    # centre: x_a = ca, x_b = ca + cw
        else:
            c1 = np.column_stack([ca, np.ones(G)]); c2 = np.column_stack([cb, np.ones(G)])
        l1 = c2 @ F; l2 = c1 @ F.T
        nc = np.abs((l1 * c1).sum(1))
        ga, gb = l1[:, :2], l2[:, :2]
        g = np.sqrt((ga ** 2).sum(1) + (gb ** 2).sum(1))
        if flow_bound:
            lb = nc - np.sqrt(((ga + gb) ** 2).sum(1)) * ra - np.sqrt((gb ** 2).sum(1)) * rw - A * (ra + rw) * ra
            ub = g + A * np.sqrt(ra ** 2 + (ra + rw) ** 2)
        else:
            lb = nc - g * R - A * ra * rb
            ub = g + A * R
        kept += np.count_nonzero(~(lb > T * ub))
    tot = len(Fs) * G
    print(f"{100 * kept / tot:6.2f} % survive ({100 * floor / tot:5.2f} % hold an inlier)   {label}", flush=True)


print("hypotheses", len(Fs))
evaluate(np.argsort(morton(pts, 7), kind="stable"), "Morton(x_a, x_b), ball bound (shipped)", False)
pw = np.column_stack([pts[:, :2], pts[:, 2:] - pts[:, :2]])
o = np.argsort(morton(pw, 7), kind="stable")
evaluate(o, "Morton(x_a, w), ball bound", False)
evaluate(o, "Morton(x_a, w), flow bound", True)
for scale in (2.0, 4.0, 8.0):
    pw2 = pw.copy(); pw2[:, 2:] *= scale     # finer cells in flow than in position
    lo, hi = pw2.min(0), pw2.max(0)
    # common scale for all four columns so that the scaling matters
    q = np.clip(((pw2 - lo) / (hi - lo).max() * 128).astype(np.int64), 0, 127 * int(scale))
    key = np.zeros(n, dtype=np.int64)
    bits = 10
    for k in range(4):
        for b in range(bits):
            key |= ((q[:, k] >> b) & 1) << (b * 4 + 3 - k)
    evaluate(np.argsort(key, kind="stable"), f"Morton(x_a, {scale:g} w) common grid, flow bound", True)
