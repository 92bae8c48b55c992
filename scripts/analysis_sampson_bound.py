"""CPU prototype of the Sampson group bound: how many (hypothesis, 64-point group) pairs the bilinear
bound would cull on the C3 set, and how many pairs the f32 point filter would pass on to the exact path.

n(x1, x2) = x2^T F x1 is bilinear and the Sampson denominator is |grad n|^2 (the gradient over the four image
coordinates).  Around a group centre c with per-view radii r1, r2 (R^2 = r1^2 + r2^2), A = F[:2,:2]:
    |n(x)| >= |n(c)| - |g(c)| R - ||A|| r1 r2,     |grad n(x)| <= |g(c)| + ||A|| R,
so the group has no inlier when |n(c)| - |g(c)| R - ||A|| r1 r2 > T (|g(c)| + ||A|| R).
"""
import sys
import numpy as np
sys.path.insert(0, "progressive-x_amd")
sys.path.insert(0, ".")
from pyprogressivex import datasets, _estimators

pts, gt, models = datasets.make_two_view_motions(seed=0)
n = len(pts)
rng = np.random.default_rng(1)
K = int(gt.max())
S = 200
smp = np.array([rng.choice(np.nonzero(gt == 1 + r % K)[0], 7, replace=False) if r % 2 == 0 else rng.choice(n, 7, replace=False)
                for r in range(S)], dtype=np.int32)
Fs = np.asarray(_estimators.FundamentalEstimator().minimal(pts, smp)[0]).reshape(-1, 9)   # host 7-point solver of the package
Fs = Fs[np.isfinite(Fs).all(1) & (np.abs(Fs).sum(1) > 0)]
print("hypotheses", len(Fs))
T = 1.5 * 0.75

def morton(p, bits):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    d = p.shape[1]
    key = np.zeros(len(p), dtype=np.int64)
    for k in range(d):
        for b in range(bits):
            key |= ((q[:, k] >> b) & 1) << (b * d + d - 1 - k)
    return key

def run(order, label):
    sp = pts[order]
    G = (n + 63) // 64
    pad = np.vstack([sp, np.repeat(sp[-1:], G * 64 - n, 0)]).reshape(G, 64, 4)
    lo, hi = pad.min(1), pad.max(1)
    c = 0.5 * (lo + hi)
    d = pad - c[:, None, :]
    r1 = np.sqrt((d[:, :, :2] ** 2).sum(2)).max(1)
    r2 = np.sqrt((d[:, :, 2:] ** 2).sum(2)).max(1)
    R = np.sqrt((d ** 2).sum(2)).max(1)
    culled = 0; total = 0; inl = 0; near = 0
    for F in Fs:
        F = F.reshape(3, 3)
        # the residual's own layout: rxc = f0 x2 + f3 y2 + f6 etc.
        x1 = np.column_stack([c[:, 0], c[:, 1], np.ones(G)])
        x2 = np.column_stack([c[:, 2], c[:, 3], np.ones(G)])
        l1 = x2 @ F          # (F^T x2): rxc, ryc, rwc
        l2 = x1 @ F.T        # (F x1): rx, ry, .
        nc = np.abs((l1 * x1).sum(1))
        g = np.sqrt(l1[:, 0] ** 2 + l1[:, 1] ** 2 + l2[:, 0] ** 2 + l2[:, 1] ** 2)
        A = np.linalg.norm(F[:2, :2], 2)
        lb = nc - g * R - A * r1 * r2
        ub = g + A * R
        cull = lb > T * ub
        culled += cull.sum(); total += G
        # pair level
        X1 = np.column_stack([sp[:, 0], sp[:, 1], np.ones(n)]); X2 = np.column_stack([sp[:, 2], sp[:, 3], np.ones(n)])
        L1 = X2 @ F; L2 = X1 @ F.T
        r = np.abs((L1 * X1).sum(1)) / np.sqrt(L1[:, 0] ** 2 + L1[:, 1] ** 2 + L2[:, 0] ** 2 + L2[:, 1] ** 2)
        inl += (r < T).sum(); near += (r < T * 1.02).sum()
    print(f"{label}: culled {culled / total:.3f} of (hyp, group) pairs; inlier pairs {inl / (len(Fs) * n):.4f}; f32-filter candidates ~{near / (len(Fs) * n):.4f}")

run(np.arange(n), "unsorted")
for bits in (7,):
    run(np.argsort(morton(pts, bits), kind="stable"), f"morton4d/{bits}")

def run_box(order, label):
    """per-coordinate half extents instead of radii"""
    sp = pts[order]
    G = (n + 63) // 64
    pad = np.vstack([sp, np.repeat(sp[-1:], G * 64 - n, 0)]).reshape(G, 64, 4)
    lo, hi = pad.min(1), pad.max(1)
    c = 0.5 * (lo + hi)
    h = 0.5 * (hi - lo)
    culled = 0; total = 0
    for F in Fs:
        F = F.reshape(3, 3)
        x1 = np.column_stack([c[:, 0], c[:, 1], np.ones(G)])
        x2 = np.column_stack([c[:, 2], c[:, 3], np.ones(G)])
        l1 = x2 @ F; l2 = x1 @ F.T
        nc = np.abs((l1 * x1).sum(1))
        g = np.abs(np.column_stack([l1[:, 0], l1[:, 1], l2[:, 0], l2[:, 1]]))   # d n / d(x1, y1, x2, y2)
        A = np.abs(F[:2, :2])     # d2 n / d x2_i d x1_j = F[i][j]
        quad = (h[:, 2:] @ A * h[:, :2]).sum(1)
        lb = nc - (g * h).sum(1) - quad
        # gradient growth: d/dx1_j changes by sum_i |A_ij| h2_i ; d/dx2_i by sum_j |A_ij| h1_j
        gu = np.column_stack([g[:, 0] + h[:, 2:] @ A[:, 0], g[:, 1] + h[:, 2:] @ A[:, 1], g[:, 2] + h[:, :2] @ A[0, :], g[:, 3] + h[:, :2] @ A[1, :]])
        ub = np.sqrt((gu ** 2).sum(1))
        culled += (lb > T * ub).sum(); total += G
    print(f"{label}: box bound culled {culled / total:.3f}")

o = np.argsort(morton(pts, 7), kind="stable")
run_box(o, "morton4d/7")
