"""Host-side (scipy kd-tree) neighbourhood graphs — TEST / BENCH INFRASTRUCTURE ONLY: independent constructions used to
cross-check and to time pgx_graph_build (csrc/graph.hip), which is what the package itself uses for every variant.

Replaces: gcransac::neighborhood::FlannNeighborhoodGraph(&points, radius) + getNeighbors(i)
(/root/reference/src/pyprogressivex/src/progressivex_python.cpp:104,207,339,458,571; PEARL.h:534).  The FLANN
implementation is absent from the snapshot [U-7, UPSTREAM-MEMORY]: upstream runs OpenCV's FlannBasedMatcher
(4 kd-trees, SearchParams(checks = 6)).radiusMatch in the full d-dimensional data space and drops the first match
(the point itself) — an APPROXIMATE search that inspects about six candidates per query, so every list holds at most a
handful of points inside the ball and the lists are not symmetric.  Three deterministic restatements are provided:

  flann_like_graph(points, radius, k=5)   default of the drop-in API: the k nearest neighbours inside the ball
                                          (exact), i.e. the list length upstream's checks=6 search can return at most
  radius_graph(points, radius)            every point inside the ball (symmetric lists => multiplicity 2, U-6)
  knn_graph(points, k)                    plain k-NN (BASELINE config C5 asks for a k-NN graph)

PEARL's setNeighbors loop (PEARL.h:532-536) inserts one entry per DIRECTED list element, so an undirected pair gets
multiplicity 1 or 2 depending on whether one or both lists contain it [U-6].
"""
import numpy as np


def csr_from_pairs(n, iu, ju, mult):
    """Undirected pairs (iu[k], ju[k]) with multiplicities -> symmetric CSR (off, idx, mult), rows sorted."""
    iu = np.asarray(iu, dtype=np.int64)
    ju = np.asarray(ju, dtype=np.int64)
    mult = np.asarray(mult, dtype=np.int64)
    a = np.concatenate([iu, ju])
    b = np.concatenate([ju, iu])
    m = np.concatenate([mult, mult])
    o = np.lexsort((b, a))
    a, b, m = a[o], b[o], m[o]
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, a + 1, 1)
    return np.cumsum(off).astype(np.int32), b.astype(np.int32), m.astype(np.int32)


def symmetrize(n, src, dst):
    """Directed neighbour entries (src -> dst), as the raw getNeighbors lists, -> symmetric CSR whose multiplicity is
    the number of directed entries between the two sites (self loops dropped, PEARL.h:535)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    key = lo * n + hi
    uniq, counts = np.unique(key, return_counts=True)
    return csr_from_pairs(n, uniq // n, uniq % n, counts)


def radius_graph(points, radius):
    from scipy.spatial import cKDTree
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    pairs = cKDTree(pts).query_pairs(r=float(radius), output_type="ndarray")
    if pairs.shape[0] == 0:
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    return csr_from_pairs(n, pairs[:, 0], pairs[:, 1], np.full(pairs.shape[0], 2))


def flann_like_graph(points, radius, k=5):
    from scipy.spatial import cKDTree
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    kk = min(k + 1, n)
    if kk <= 1:
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    dist, nbr = cKDTree(pts).query(pts, k=kk, distance_upper_bound=float(radius), workers=-1)
    src = np.repeat(np.arange(n), kk)
    dst = nbr.reshape(-1)
    ok = np.isfinite(dist.reshape(-1)) & (dst < n) & (dst != src)
    if not ok.any():
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    return symmetrize(n, src[ok], dst[ok])


def knn_graph(points, k):
    from scipy.spatial import cKDTree
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    k = min(k, n - 1)
    if k <= 0:
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    _, nbr = cKDTree(pts).query(pts, k=k + 1, workers=-1)
    src = np.repeat(np.arange(n), k)
    return symmetrize(n, src, nbr[:, 1:].reshape(-1))


def neighbour_lists(graph):
    """CSR -> python list of index arrays (used by the NAPSAC sampler)."""
    off, idx, _ = graph
    return [idx[off[i]:off[i + 1]] for i in range(len(off) - 1)]
