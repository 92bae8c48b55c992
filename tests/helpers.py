"""Shared input builders for the tests (seeded, small)."""
import numpy as np

from pyprogressivex import datasets

MODEL_CASES = {"line": 0, "homography": 1, "fundamental": 2, "pnp": 3, "vanishing_point": 4, "homography_sym": 5}


def _fit_n(rng, arr, n):
    idx = rng.permutation(arr.shape[0])
    if arr.shape[0] >= n:
        return np.ascontiguousarray(arr[idx[:n]])
    extra = rng.integers(0, arr.shape[0], n - arr.shape[0])
    return np.ascontiguousarray(np.vstack([arr, arr[extra]])[rng.permutation(n)])


def make_case(name, n, M, seed=0):
    """(model_type, points[n,d], models[M,p], threshold): a few ground-truth structures with inliers, outliers, and
    hypotheses that are ground truth, perturbed ground truth or random (so counts span 0 .. many)."""
    rng = np.random.default_rng(seed)
    mt = MODEL_CASES[name]
    per = max(2, n // 5)
    if name == "line":
        pts, _, gt = datasets.make_lines(n_per_line=per, n_lines=3, n_outliers=per, seed=seed)
        thr = 2.0
    elif name in ("homography", "homography_sym"):
        pts, _, gt = datasets.make_homographies(n_per_plane=per, n_planes=3, n_outliers=per, seed=seed)
        thr = 3.0
        if name == "homography_sym":
            gt = np.array([np.concatenate([h, np.linalg.inv(h.reshape(3, 3)).reshape(-1)]) for h in gt])
    elif name == "fundamental":
        pts, _, gt = datasets.make_two_view_motions(n_per_motion=per, n_motions=3, n_outliers=per, seed=seed)
        thr = 0.75
    elif name == "pnp":
        x1, x2, K, _, gt = datasets.make_poses(n_per_object=per, n_objects=3, n_outliers=per, seed=seed)
        pts, f = datasets.normalize_pnp(x1, x2, K)
        thr = 4.0 / f
    elif name == "vanishing_point":
        pts, _, gt = datasets.make_vanishing_points(n_inliers=3 * per, n_vps=3, n_outliers=per, seed=seed)
        thr = 1.5
    else:
        raise KeyError(name)
    pts = _fit_n(rng, pts, n)
    models = []
    for m in range(M):
        g = gt[m % len(gt)]
        kind = m % 4 if m >= len(gt) else 0
        if kind == 0:
            models.append(g.copy())
        elif kind == 1:
            models.append(g * (1.0 + rng.normal(0, 1e-4, g.shape)))
        elif kind == 2:
            models.append(g * (1.0 + rng.normal(0, 1e-2, g.shape)))
        else:
            models.append(rng.normal(0, 1, g.shape) * np.abs(g).max())
    return mt, pts, np.ascontiguousarray(np.array(models)), thr


def random_sym_graph(rng, n, p):
    """Symmetric CSR (off, idx, mult) with random multiplicities 1..2 per undirected pair."""
    if p <= 0 or n < 2:
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random(iu.shape[0]) < p
    iu, ju = iu[keep], ju[keep]
    mult = rng.integers(1, 3, iu.shape[0])
    return csr_from_pairs(n, iu, ju, mult)


def csr_from_pairs(n, iu, ju, mult):
    a = np.concatenate([iu, ju])
    b = np.concatenate([ju, iu])
    m = np.concatenate([mult, mult])
    o = np.lexsort((b, a))
    a, b, m = a[o], b[o], m[o]
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, a + 1, 1)
    return np.cumsum(off).astype(np.int32), b.astype(np.int32), m.astype(np.int32)


def realistic_labeling_problem(n, L, lam, seed=0):
    """Points in the unit square with a radius graph; clusters of points prefer one of L-1 model labels, the last
    label is the outlier label with constant cost (1 - lambda), as PEARL's data term prices it."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 2))
    pairs = cKDTree(pts).query_pairs(r=2.0 / np.sqrt(n), output_type="ndarray")
    graph = csr_from_pairs(n, pairs[:, 0], pairs[:, 1], np.full(pairs.shape[0], 2))
    D = rng.random((n, L)) * 2 * (1 - lam)
    D[:, L - 1] = 1 - lam
    cl = (pts[:, 0] * 5).astype(int) % (L - 1)
    D[np.arange(n), cl] *= 0.1
    Dq = np.rint(D * 2.0 ** 32).astype(np.int64)
    return Dq, graph


def fixed_point_accumulators(O, mt, pts, models, T2, comp=None, n_total=None):
    """The integer accumulators the group-major score path builds (score.hip: count, and value / shared support as sums of
    round-to-nearest-even(term * 2^q) per inlier), restated from the ORACLE's residuals: getScore's terms
    (scoring_function_with_compound_model.h:85-97, 115-117) in the fixed point of a job of n_total points."""
    from pyprogressivex import parallel
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    models = np.ascontiguousarray(models, dtype=np.float64)
    q = parallel.fixed_point_scale(pts.shape[0] if n_total is None else n_total)
    M = models.shape[0]
    counts = np.zeros(M, np.int64)
    values_q = np.zeros(M, np.int64)
    shared_q = np.zeros(M, np.int64)
    for m in range(M):
        sq = O.squared_residuals(mt, pts, models[m])
        with np.errstate(invalid="ignore"):
            inl = sq < T2
        sc = np.maximum(0.0, 1.0 - sq[inl] / T2)
        counts[m] = int(inl.sum())
        values_q[m] = int(np.rint(sc * q).astype(np.int64).sum())
        if comp is not None:
            shared_q[m] = int(np.rint(np.minimum(np.asarray(comp)[inl], sc) * q).astype(np.int64).sum())
    return dict(counts=counts, values_q=values_q, shared_q=shared_q)
