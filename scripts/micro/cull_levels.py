"""Survival statistics of the cull kernel's two bound levels on the metric batch (numpy emulation of
Filter32<PnP>::group_reject in f64 without the trust test: statistics, not decisions).
usage (GPU box): python scripts/micro/cull_levels.py"""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "progressive-x_amd")]
import numpy as np
from pyprogressivex import _lib, datasets

x1, x2, K, lab, gt = datasets.make_poses(n_per_object=50000, n_objects=16, n_outliers=200000, seed=0)
pts, f = datasets.normalize_pnp(x1, x2, K)
thr = 4.0 / f
T2 = 9.0 / 4.0 * thr * thr
hyps = datasets.make_pose_hypotheses(gt, M=2048, seed=1)
ctx = _lib.Context(0)
ctx.set_points(_lib.PNP, pts)
b = ctx.score_debug_fetch("bounds").astype(np.float64)
groups = (pts.shape[0] + 63) // 64
gb, sb = b[:groups], b[groups:]
Tup = np.sqrt(T2) * (1 + 1 / 64)
infl = 1.0 + 1e-5


def survive(rows, H):
    """rows [G,12], H [M,12] -> bool [M,G]: not rejected"""
    m = H
    c = rows[:, :3]
    cx = m[:, None, 0] * c[None, :, 0] + m[:, None, 1] * c[None, :, 1] + m[:, None, 2] * c[None, :, 2] + m[:, None, 3]
    cy = m[:, None, 4] * c[None, :, 0] + m[:, None, 5] * c[None, :, 1] + m[:, None, 6] * c[None, :, 2] + m[:, None, 7]
    cz = m[:, None, 8] * c[None, :, 0] + m[:, None, 9] * c[None, :, 1] + m[:, None, 10] * c[None, :, 2] + m[:, None, 11]
    n0 = np.linalg.norm(m[:, 0:3], axis=1)[:, None] * infl
    n1 = np.linalg.norm(m[:, 4:7], axis=1)[:, None] * infl
    n2 = np.linalg.norm(m[:, 8:11], axis=1)[:, None] * infl
    rho, ub, vb, ru, rv = (rows[None, :, k] for k in (3, 4, 5, 6, 7))
    dz, dx, dy = n2 * rho, n0 * rho, n1 * rho
    zs = np.abs(cz) + dz
    ex, ey = np.abs(ub * cz - cx), np.abs(vb * cz - cy)
    mx = ru * zs + np.abs(ub) * dz + dx
    my = rv * zs + np.abs(vb) * dz + dy
    tol = Tup * zs
    return ~((ex - mx > tol) | (ey - my > tol))


M = hyps.shape[0]
S = np.zeros((M, sb.shape[0]), dtype=bool)
G = np.zeros((M, groups), dtype=bool)
for i in range(0, M, 128):
    S[i:i + 128] = survive(sb, hyps[i:i + 128])
    G[i:i + 128] = survive(gb, hyps[i:i + 128])
Sg = np.repeat(S, 8, axis=1)[:, :groups]
print("hyp x super-group pairs:", S.size, "surviving", int(S.sum()), f"({S.mean():.4f})")
print("hyp x group pairs:", G.size, "surviving the group bound", int(G.sum()), f"({G.mean():.4f})", " both bounds", int((G & Sg).sum()))
print("group survivors inside surviving super-groups / 8 x surviving super-groups:", (G & Sg).sum() / (8 * S.sum()))
# a 64-group level (8 super-groups): emulate with the union box? not available; report how clustered the survivors are instead
S64 = S[:, : S.shape[1] // 8 * 8].reshape(M, -1, 8).any(axis=2)
print("hyp x 64-group blocks with any surviving super-group:", int(S64.sum()), "of", S64.size, f"({S64.mean():.4f})")
# word-level statistics in the caller's order (the device reorders the batch by locality; this is the unsorted figure)
Sw = S.reshape(M // 64, 64, -1).any(axis=1)
print("unsorted words: (word, super-group) with any survivor:", f"{Sw.mean():.4f}")
