"""ctypes view of the CPU oracle (oracle/pgx_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from the product package (progressive-x_amd/).
PARITY UNPINNED (see pgx_oracle.h header and DESIGN.md §3).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PGX_ORACLE_SO") or os.path.join(_HERE, "libpgx_oracle.so")   # (scripts/sanitize.sh points it at the ASAN/UBSAN build)

LINE2D, HOMOGRAPHY, FUNDAMENTAL, PNP, VANISHING_POINT, HOMOGRAPHY_SYM = range(6)
POINT_DIM = {0: 2, 1: 4, 2: 4, 3: 5, 4: 4, 5: 4}
PARAM_DIM = {0: 3, 1: 9, 2: 9, 3: 12, 4: 3, 5: 18}
FIXED_ONE = 1 << 32


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("pgx_oracle.c", "pgx_oracle.h", "progx_replay.c", "progx_replay.h", "bk_maxflow.c")]
    if os.environ.get("PGX_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs if os.path.exists(f)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.pgxo_squared_residual.restype = C.c_double
        _lib.pgxo_residual.restype = C.c_double
        _lib.pgxo_residual_sum.restype = C.c_double
        _lib.pgxo_epipolar_support.restype = None
        _lib.pgxo_quantize.restype = C.c_int64
        _lib.pgxo_quantize.argtypes = [C.c_double]
        _lib.pgxo_energy.restype = C.c_int64
        _lib.pgxo_maxflow.restype = C.c_int64
        _lib.pgxo_predicted_unseen_inliers.restype = C.c_uint64
        _lib.pgxo_philox4x32.restype = None
        _lib.pgxo_sample_uniform.restype = None
        _lib.pgxo_sample_napsac.restype = None
        _lib.pgxo_sample_prosac.restype = None
        _lib.pgxo_predicted_unseen_inliers.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def squared_residuals(model_type, pts, model):
    pts = _f64(pts); model = _f64(model)
    out = np.empty(pts.shape[0], dtype=np.float64)
    lib().pgxo_squared_residuals(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(pts.shape[0]),
                                 _p(model, C.c_double), _p(out, C.c_double))
    return out


def residual(model_type, pt, model):
    pt = _f64(pt); model = _f64(model)
    return lib().pgxo_residual(C.c_int(model_type), _p(pt, C.c_double), _p(model, C.c_double))


def score(model_type, pts, models, T2, compound=None, has_compound=None, exponent=2,
          best_inlier_number=None, want_masks=False):
    pts = _f64(pts); models = _f64(models).reshape(-1, PARAM_DIM[model_type])
    n, M = pts.shape[0], models.shape[0]
    if has_compound is None:
        has_compound = compound is not None
    comp = _f64(compound) if compound is not None else np.zeros(n, dtype=np.float64)
    best = None if best_inlier_number is None else np.ascontiguousarray(best_inlier_number, dtype=np.int64)
    counts = np.zeros(M, dtype=np.int64)
    values = np.zeros(M, dtype=np.float64)
    shared = np.zeros(M, dtype=np.float64)
    scores = np.zeros(M, dtype=np.float64)
    masks = np.zeros((M, (n + 63) // 64), dtype=np.uint64) if want_masks else None
    lib().pgxo_score(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(n), _p(models, C.c_double),
                     C.c_int(M), C.c_double(T2), _p(comp, C.c_double), C.c_int(1 if has_compound else 0),
                     C.c_int(int(exponent)), _p(best, C.c_int64), _p(counts, C.c_int64),
                     _p(values, C.c_double), _p(shared, C.c_double), _p(scores, C.c_double),
                     _p(masks, C.c_uint64))
    return dict(counts=counts, values=values, shared=shared, scores=scores, masks=masks)


def preference(model_type, pts, model, T2):
    pts = _f64(pts); model = _f64(model)
    out = np.empty(pts.shape[0], dtype=np.float64)
    lib().pgxo_preference(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(pts.shape[0]),
                          _p(model, C.c_double), C.c_double(T2), _p(out, C.c_double))
    return out


def tanimoto_terms(pref, compound):
    pref = _f64(pref); compound = _f64(compound)
    d, a, b = C.c_double(), C.c_double(), C.c_double()
    lib().pgxo_tanimoto_terms(_p(pref, C.c_double), _p(compound, C.c_double), C.c_int64(pref.shape[0]),
                              C.byref(d), C.byref(a), C.byref(b))
    return d.value, a.value, b.value


def is_valid_tanimoto(dot, pn, cn, max_tanimoto):
    t = C.c_double()
    ok = lib().pgxo_is_valid_tanimoto(C.c_double(dot), C.c_double(pn), C.c_double(cn),
                                      C.c_double(max_tanimoto), C.byref(t))
    return bool(ok), t.value


def compound_max(prefs):
    prefs = _f64(prefs)
    K, n = prefs.shape
    out = np.empty(n, dtype=np.float64)
    lib().pgxo_compound_max(_p(prefs, C.c_double), C.c_int(K), C.c_int64(n), _p(out, C.c_double))
    return out


def predicted_unseen_inliers(one_minus_conf, sample_size, iteration_number, covered, point_number):
    return int(lib().pgxo_predicted_unseen_inliers(one_minus_conf, sample_size, iteration_number,
                                                   covered, point_number))


def unary(model_type, pts, models, threshold, lam):
    pts = _f64(pts); models = _f64(models).reshape(-1, PARAM_DIM[model_type])
    n, K = pts.shape[0], models.shape[0]
    D = np.empty((n, K + 1), dtype=np.float64)
    lib().pgxo_unary(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(n), _p(models, C.c_double),
                     C.c_int(K), C.c_double(threshold), C.c_double(lam), _p(D, C.c_double))
    return D


def unary_q(model_type, pts, models, threshold, lam):
    pts = _f64(pts); models = _f64(models).reshape(-1, PARAM_DIM[model_type])
    n, K = pts.shape[0], models.shape[0]
    Dq = np.empty((n, K + 1), dtype=np.int64)
    lib().pgxo_unary_q(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(n), _p(models, C.c_double),
                       C.c_int(K), C.c_double(threshold), C.c_double(lam), _p(Dq, C.c_int64))
    return Dq


def quantize(x):
    return int(lib().pgxo_quantize(float(x)))


def quantize_lambda(lam):
    """weight of ONE directed neighbour entry; forced even so that w/2 is exact."""
    return 2 * int(np.rint(float(lam) * (1 << 31)))


def _graph_args(graph):
    if graph is None:
        return None, None, None, ()
    off, idx, mult = (_i32(g) for g in graph)
    return _p(off, C.c_int32), _p(idx, C.c_int32), _p(mult, C.c_int32), (off, idx, mult)


def energy(Dq, graph, lambda_q, h_q, labels):
    Dq = np.ascontiguousarray(Dq, dtype=np.int64); labels = _i32(labels)
    n, L = Dq.shape
    po, pi, pm, keep = _graph_args(graph)
    return int(lib().pgxo_energy(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), po, pi, pm,
                                 C.c_int64(lambda_q), C.c_int64(h_q), _p(labels, C.c_int32)))


def expand_alpha(Dq, graph, lambda_q, h_q, alpha, labels):
    Dq = np.ascontiguousarray(Dq, dtype=np.int64); labels = _i32(labels).copy()
    n, L = Dq.shape
    po, pi, pm, keep = _graph_args(graph)
    flow = C.c_int64()
    changed = lib().pgxo_expand_alpha(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), po, pi, pm,
                                      C.c_int64(lambda_q), C.c_int64(h_q), C.c_int(alpha),
                                      _p(labels, C.c_int32), C.byref(flow))
    return labels, int(changed), int(flow.value)


def gc_labeling(model_type, pts, model, T2, lam, graph):
    """GC-RANSAC's inlier/outlier cut ([U-12]): flags[n] int32, 1 = inlier"""
    pts = _f64(pts); model = _f64(model)
    n = pts.shape[0]
    off, idx = _i32(graph[0]), _i32(graph[1])
    flags = np.zeros(n, dtype=np.int32)
    lib().pgxo_gc_labeling.restype = C.c_int64
    lib().pgxo_gc_labeling(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(n), _p(model, C.c_double),
                           C.c_double(T2), C.c_double(lam), _p(off, C.c_int32), _p(idx, C.c_int32), _p(flags, C.c_int32))
    return flags


def expansion(Dq, graph, lambda_q, h_q, labels, max_cycles=1000):
    Dq = np.ascontiguousarray(Dq, dtype=np.int64); labels = _i32(labels).copy()
    n, L = Dq.shape
    po, pi, pm, keep = _graph_args(graph)
    e = C.c_int64(); cyc = C.c_int()
    lib().pgxo_expansion(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), po, pi, pm, C.c_int64(lambda_q),
                         C.c_int64(h_q), _p(labels, C.c_int32), C.c_int(max_cycles), C.byref(e),
                         C.byref(cyc))
    return labels, int(e.value), int(cyc.value)


def expansion_bk(Dq, graph, lambda_q, h_q, labels, max_cycles=1000):
    """pgxo_expansion with Boykov-Kolmogorov as the min-cut solver (oracle/bk_maxflow.c) -> (labels, energy_q, cycles, mincuts)"""
    Dq = np.ascontiguousarray(Dq, dtype=np.int64); labels = _i32(labels).copy()
    n, L = Dq.shape
    po, pi, pm, keep = _graph_args(graph)
    e = C.c_int64(); cyc = C.c_int(); cuts = C.c_int64()
    lib().pgxo_expansion_bk(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), po, pi, pm, C.c_int64(lambda_q),
                            C.c_int64(h_q), _p(labels, C.c_int32), C.c_int(max_cycles), C.byref(e), C.byref(cyc), C.byref(cuts))
    return labels, int(e.value), int(cyc.value), int(cuts.value)


def expand_alpha_bk(Dq, graph, lambda_q, h_q, alpha, labels):
    Dq = np.ascontiguousarray(Dq, dtype=np.int64); labels = _i32(labels).copy()
    n, L = Dq.shape
    po, pi, pm, keep = _graph_args(graph)
    flow = C.c_int64()
    changed = lib().pgxo_expand_alpha_bk(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), po, pi, pm,
                                         C.c_int64(lambda_q), C.c_int64(h_q), C.c_int(alpha),
                                         _p(labels, C.c_int32), C.byref(flow))
    return labels, int(changed), int(flow.value)


def maxflow_bk(nnodes, frm, to, cap, s, t):
    frm = _i32(frm); to = _i32(to); cap = np.ascontiguousarray(cap, dtype=np.int64)
    side = np.zeros(nnodes, dtype=np.uint8)
    fn = lib().pgxo_maxflow_bk
    fn.restype = C.c_int64
    f = fn(C.c_int(nnodes), C.c_int64(len(frm)), _p(frm, C.c_int32), _p(to, C.c_int32),
           _p(cap, C.c_int64), C.c_int(s), C.c_int(t), _p(side, C.c_uint8))
    return int(f), side


def greedy_labeling(Dq, h_q):
    """U-8: GCO-v3's labelling of an energy without smooth costs (greedy facility location) -> (labels, energy_q, opened)"""
    Dq = np.ascontiguousarray(Dq, dtype=np.int64)
    n, L = Dq.shape
    labels = np.zeros(n, dtype=np.int32)
    e = C.c_int64()
    opened = lib().pgxo_greedy_labeling(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), C.c_int64(h_q), _p(labels, C.c_int32),
                                        C.byref(e))
    return labels, int(e.value), int(opened)


def maxflow(nnodes, frm, to, cap, s, t):
    frm = _i32(frm); to = _i32(to); cap = np.ascontiguousarray(cap, dtype=np.int64)
    side = np.zeros(nnodes, dtype=np.uint8)
    f = lib().pgxo_maxflow(C.c_int(nnodes), C.c_int64(len(frm)), _p(frm, C.c_int32), _p(to, C.c_int32),
                           _p(cap, C.c_int64), C.c_int(s), C.c_int(t), _p(side, C.c_uint8))
    return int(f), side


def bucket(labels, L):
    labels = _i32(labels)
    counts = np.zeros(L, dtype=np.int64)
    order = np.empty(labels.shape[0], dtype=np.int32)
    lib().pgxo_bucket(_p(labels, C.c_int32), C.c_int64(labels.shape[0]), C.c_int(L),
                      _p(counts, C.c_int64), _p(order, C.c_int32))
    return counts, order


def epipolar_support(pts, F, T2, S2):
    """(Sampson inliers, those also within S2 of the symmetric epipolar distance) of F over the correspondences [n, 4]"""
    pts = _f64(pts); F = _f64(F).reshape(-1)
    out = np.zeros(2, dtype=np.int64)
    lib().pgxo_epipolar_support(_p(pts, C.c_double), C.c_int64(pts.shape[0]), _p(F, C.c_double), C.c_double(T2), C.c_double(S2),
                                _p(out, C.c_int64))
    return int(out[0]), int(out[1])


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr]); k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    out = (C.c_uint32 * 4)()
    lib().pgxo_philox4x32(c, k, out)
    return [int(x) for x in out]


def sample_uniform(key, batch, first, count, n, m):
    out = np.empty((count, m), dtype=np.int32)
    lib().pgxo_sample_uniform(C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int64(first), C.c_int64(count),
                              C.c_int64(n), C.c_int(m), _p(out, C.c_int32))
    return out


def sample_napsac(key, batch, first, count, n, off, idx, m):
    out = np.empty((count, m), dtype=np.int32)
    off32, idx32 = _i32(off), _i32(idx)
    lib().pgxo_sample_napsac(C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int64(first), C.c_int64(count),
                             C.c_int64(n), _p(off32, C.c_int32), _p(idx32, C.c_int32), C.c_int(m), _p(out, C.c_int32))
    return out


def sample_prosac(key, batch, first, count, n, tops, m):
    out = np.empty((count, m), dtype=np.int32)
    tops32 = _i32(tops)
    assert len(tops32) >= count
    lib().pgxo_sample_prosac(C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int64(first), C.c_int64(count),
                             C.c_int64(n), _p(tops32, C.c_int32), C.c_int(m), _p(out, C.c_int32))
    return out


def sample_pnapsac(key, batch, count, pts, sizes, m, tops, growth_local, max_local, layers=(16, 8, 4, 2)):
    """Progressive NAPSAC on the in-repo generator, one draw: [count, m] int32 (pgxo_sample_pnapsac)"""
    pts = _f64(pts)
    n, d = pts.shape
    sz = np.zeros(4)
    sv = _f64(sizes).reshape(-1)[:4]
    sz[:len(sv)] = sv
    lay = _i32(layers)
    tops32 = _i32(tops)
    growth = np.ascontiguousarray(growth_local, dtype=np.int64)
    assert len(tops32) >= count and len(growth) >= n
    out = np.empty((count, m), dtype=np.int32)
    lib().pgxo_sample_pnapsac.restype = C.c_int
    r = lib().pgxo_sample_pnapsac(_p(pts, C.c_double), C.c_int64(n), C.c_int(d), _p(sz, C.c_double), _p(lay, C.c_int32), C.c_int(len(lay)), C.c_int(m),
                                  C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int32(int(count)),
                                  _p(tops32, C.c_int32), _p(growth, C.c_int64), C.c_int64(int(max_local)), _p(out, C.c_int32))
    if r != 0:
        raise ValueError(f"pgxo_sample_pnapsac: {r}")
    return out


def residual_sum(model_type, pts, model, labels, label):
    pts = _f64(pts); model = _f64(model); labels = _i32(labels)
    return lib().pgxo_residual_sum(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(pts.shape[0]),
                                   _p(model, C.c_double), _p(labels, C.c_int32), C.c_int(label))


def solve_minimal(model_type, pts, samples):
    """[S,3] models of the 2-point line / 2-segment vanishing point solvers, [3S,9] (three slots per sample) of the
    7-point fundamental matrix solver; NaN rows = no model"""
    pts = _f64(pts); samples = _i32(samples)
    shape = {FUNDAMENTAL: (samples.shape[0] * 3, 9), HOMOGRAPHY: (samples.shape[0], 9),
             PNP: (samples.shape[0] * 4, 12)}.get(model_type, (samples.shape[0], 3))
    out = np.empty(shape, dtype=np.float64)
    r = lib().pgxo_solve_minimal(C.c_int(model_type), _p(pts, C.c_double), C.c_int64(pts.shape[0]), _p(samples, C.c_int32),
                                 C.c_int(samples.shape[0]), _p(out, C.c_double))
    if r != 0:
        raise ValueError("solve_minimal: model type without a device solver")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# neighbourhood graph — numpy restatement of the deterministic lists the build defines for
# FlannNeighborhoodGraph [U-7] (/root/reference/src/pyprogressivex/src/progressivex_python.cpp:104,207,339,458,571) and of
# the setNeighbors multiplicities (/root/reference/src/pyprogressivex/include/PEARL.h:532-536) [U-6].
# ---------------------------------------------------------------------------------------------------------------------
def _sqdist_rows(a, b):
    """squared distances of row a against rows b, summed in dimension order (the contract of graph.hip)"""
    s = (a[0] - b[:, 0]) * (a[0] - b[:, 0])
    for j in range(1, b.shape[1]):
        df = a[j] - b[:, j]
        s = s + df * df
    return s


def graph_lists(points, k, radius=None, rows=None):
    """[n,k] int32 (or [len(rows),k] for the listed query rows), -1 padded: the k nearest neighbours ranked by (squared distance, index), inside the ball when
    `radius` is given (squared distance <= radius*radius).  Candidates come from a kd-tree with a safety margin, the
    ranking itself is recomputed with the exact arithmetic; brute force for small n."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = pts.shape[0]
    k = min(k, n - 1)
    rows = np.arange(n) if rows is None else np.asarray(rows)
    out = np.full((len(rows), max(k, 0)), -1, dtype=np.int32)
    if k <= 0:
        return out
    r2 = np.inf if radius is None else float(radius) * float(radius)
    if n <= 4000:   # brute force: the whole distance matrix, ranking by a stable sort (ties -> lower index)
        S = (pts[rows, 0][:, None] - pts[None, :, 0]) * (pts[rows, 0][:, None] - pts[None, :, 0])
        for j in range(1, pts.shape[1]):
            df = pts[rows, j][:, None] - pts[None, :, j]
            S = S + df * df
        S[np.arange(len(rows)), rows] = np.inf
        S[~(S <= r2)] = np.inf
        order = np.argsort(S, axis=1, kind="stable")[:, :k]
        ok = np.take_along_axis(S, order, axis=1) < np.inf
        out[:] = np.where(ok, order, -1)
        return out
    from scipy.spatial import cKDTree
    tree = cKDTree(pts)
    kk = min(n, k + 17)
    if radius is None:
        _, cands = tree.query(pts[rows], k=kk)
    else:
        _, cands = tree.query(pts[rows], k=kk, distance_upper_bound=float(radius) * (1.0 + 1e-9))
    for r, i in enumerate(rows):
        cand = np.asarray(cands[r])
        cand = cand[(cand < n) & (cand != i)]
        s = _sqdist_rows(pts[i], pts[cand])
        keep = s <= r2
        cand, s = cand[keep], s[keep]
        full = np.lexsort((cand, s))
        order = full[:k]
        if len(order) == k and len(cand) >= kk - 1:
            # all candidates tied with the k-th must be inside the margin for the ranking to be decided here
            assert s[full[-1]] > s[order[-1]], "tie margin exhausted: raise the candidate margin"
        out[r, :len(order)] = cand[order]
    return out


def graph_from_lists(lists):
    """directed lists -> symmetric CSR (off, idx, mult) with rows sorted; multiplicity = number of directed entries"""
    n, k = lists.shape
    src = np.repeat(np.arange(n, dtype=np.int64), k)
    dst = lists.reshape(-1).astype(np.int64)
    ok = dst >= 0
    src, dst = src[ok], dst[ok]
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    key, cnt = np.unique(lo * n + hi, return_counts=True)
    a = np.concatenate([key // n, key % n])
    b = np.concatenate([key % n, key // n])
    m = np.concatenate([cnt, cnt])
    o = np.lexsort((b, a))
    a, b, m = a[o], b[o], m[o]
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, a + 1, 1)
    return np.cumsum(off).astype(np.int32), b.astype(np.int32), m.astype(np.int32)


def graph_ball(points, radius):
    """exhaustive ball (kind 1): every point with squared distance <= radius^2, symmetric lists -> multiplicity 2.
    Candidate pairs from a kd-tree with a safety margin, the decision itself with the exact arithmetic."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = pts.shape[0]
    r2 = float(radius) * float(radius)
    if n <= 4000:
        S = (pts[:, 0][:, None] - pts[None, :, 0]) * (pts[:, 0][:, None] - pts[None, :, 0])
        for j in range(1, pts.shape[1]):
            df = pts[:, j][:, None] - pts[None, :, j]
            S = S + df * df
        a, b = np.nonzero((S <= r2) & ~np.eye(n, dtype=bool))
    else:
        from scipy.spatial import cKDTree
        pairs = cKDTree(pts).query_pairs(r=float(radius) * (1.0 + 1e-9), output_type="ndarray")
        i, j = pairs[:, 0], pairs[:, 1]
        s = (pts[i, 0] - pts[j, 0]) * (pts[i, 0] - pts[j, 0])
        for c in range(1, pts.shape[1]):
            df = pts[i, c] - pts[j, c]
            s = s + df * df
        keep = s <= r2
        a = np.concatenate([i[keep], j[keep]])
        b = np.concatenate([j[keep], i[keep]])
    o = np.lexsort((b, a))
    a, b = a[o], b[o]
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, a + 1, 1)
    return np.cumsum(off).astype(np.int32), b.astype(np.int32), np.full(len(b), 2, dtype=np.int32)


def graph_build(points, kind, radius=0.0, k=5):
    """kind 0: k nearest inside the ball; kind 1: the whole ball; kind 2: plain k-NN (the constants of include/pgx.h)"""
    if kind == 1:
        return graph_ball(points, radius)
    lists = graph_lists(points, k, radius if kind == 0 else None)
    return graph_from_lists(lists)


# ---------------------------------------------------------------------------------------------------------------------
# Gram matrices of the non-minimal refits — numpy restatement of include/pgx.h pgx_gram.
# Row definitions: vanishing point = /root/reference/src/pyprogressivex/include/solver_vanishing_point_two_lines.h:212-217;
# the other estimators' solvers are absent from the snapshot (graph-cut-ransac submodule) and restated from the
# literature: normalised DLT / 8-point rows, Gauss-Newton rows of the reprojection error at [R|t].
# ---------------------------------------------------------------------------------------------------------------------
GRAM_AFFINE, GRAM_DLT_H, GRAM_EPI_F, GRAM_VP, GRAM_PNP_GN = 0, 1, 2, 3, 4


def gram_rows(kind, p, params=None):
    """list of row blocks [m,q] (one per row of a point) and a `bad` mask"""
    bad = np.zeros(p.shape[0], dtype=bool)
    one = np.ones(p.shape[0])
    if kind == GRAM_AFFINE:
        return [np.column_stack([one, p])], bad
    if kind in (GRAM_DLT_H, GRAM_EPI_F):
        s1, cx1, cy1, s2, cx2, cy2 = [float(v) for v in params]
        x1, y1 = (p[:, 0] - cx1) * s1, (p[:, 1] - cy1) * s1
        x2, y2 = (p[:, 2] - cx2) * s2, (p[:, 3] - cy2) * s2
        z = np.zeros_like(x1)
        if kind == GRAM_DLT_H:
            return [np.column_stack([-x1, -y1, -one, z, z, z, x2 * x1, x2 * y1, x2]),
                    np.column_stack([z, z, z, -x1, -y1, -one, y2 * x1, y2 * y1, y2])], bad
        return [np.column_stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, one])], bad
    if kind == GRAM_VP:
        x0, y0, x1, y1 = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
        mx, my, mz = (x0 + x1) / 2.0, (y0 + y1) / 2.0, 1.0
        return [np.column_stack([y0 * mz - my, mx - x0 * mz, x0 * my - y0 * mx])], bad
    if kind == GRAM_PNP_GN:
        P = np.asarray(params, dtype=np.float64).reshape(3, 4)
        Xr = p[:, 2:] @ P[:, :3].T
        Xc = Xr + P[:, 3]
        bad = ~(np.abs(Xc[:, 2]) >= 1e-12)
        zc = np.where(bad, 1.0, Xc[:, 2])
        inv = 1.0 / zc
        du, dv = Xc[:, 0] * inv - p[:, 0], Xc[:, 1] * inv - p[:, 1]
        a, b, c = inv, -Xc[:, 0] * inv * inv, -Xc[:, 1] * inv * inv
        rx, ry, rz = Xr[:, 0], Xr[:, 1], Xr[:, 2]
        z = np.zeros_like(a)
        ju = np.column_stack([b * ry, a * rz - b * rx, -a * ry, a, z, b, du])
        jv = np.column_stack([-a * rz + c * ry, -c * rx, a * rx, z, a, c, dv])
        return [ju, jv], bad
    raise ValueError(kind)


def gram(kind, pts, index, params=None, weights=None, wpow=2):
    """(G [q,q], count, bad) over pts[index]"""
    index = np.asarray(index, dtype=np.int64)
    p = np.ascontiguousarray(pts, dtype=np.float64)[index]
    blocks, bad = gram_rows(kind, p, params)
    w = np.ones(len(index)) if weights is None or len(weights) == 0 else np.asarray(weights, dtype=np.float64)[index] ** wpow
    w = np.where(bad, 0.0, w)
    q = blocks[0].shape[1]
    G = np.zeros((q, q))
    for A in blocks:
        A = np.where(bad[:, None], 0.0, A)
        G = G + (A * w[:, None]).T @ A
    return G, int(len(index)), int(bad.sum())


def eigh_smallest(A):
    """pgxo_eigh_smallest: (vec [B, q], val [B], sweeps [B]) of the symmetric matrices A [B, q, q] (q <= 9), cyclic Jacobi in the
    operation order of the device kernel (csrc/fit.hip eigh_smallest_kernel)"""
    A = np.ascontiguousarray(A, dtype=np.float64)
    B, q, _ = A.shape
    assert q <= 9 and A.shape[2] == q
    vec = np.zeros((B, q))
    val = np.zeros(B)
    sw = np.zeros(B, dtype=np.int32)
    fn = lib().pgxo_eigh_smallest
    fn.restype = None
    fn(A.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(q), C.c_int64(B), vec.ctypes.data_as(C.POINTER(C.c_double)),
       val.ctypes.data_as(C.POINTER(C.c_double)), sw.ctypes.data_as(C.POINTER(C.c_int32)))
    return vec, val, sw
