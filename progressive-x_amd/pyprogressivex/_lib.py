"""ctypes binding of libpgx.so (include/pgx.h) — the only door from the Python host code to the GPU.

There is no CPU fallback: if the shared library is missing or no HIP device is present the calls raise
`PgxError`; nothing in this package ever routes to the test oracle.
"""
import ctypes as C
import os

import numpy as np

# Multi-process GPU work (one rank per GPU over RCCL) needs dmabuf IPC on hosts whose driver has no legacy IPC: without this
# RCCL's bootstrap fails with "hipIpcGetMemHandle: invalid argument".  Set before the HIP runtime initialises; a value the
# launcher exported wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGX_LIBPGX") or os.path.join(_HERE, "libpgx.so")   # PGX_LIBPGX: A/B of kernel builds

LINE2D, HOMOGRAPHY, FUNDAMENTAL, PNP, VANISHING_POINT, HOMOGRAPHY_SYM = range(6)
POINT_DIM = {0: 2, 1: 4, 2: 4, 3: 5, 4: 4, 5: 4}
PARAM_DIM = {0: 3, 1: 9, 2: 9, 3: 12, 4: 3, 5: 18}
FIXED_ONE = float(1 << 32)
UNIQUE_ID_BYTES = 128

# every symbol include/pgx.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "pgx_version", "pgx_device_count", "pgx_global_error", "pgx_create", "pgx_destroy", "pgx_last_error",
    "pgx_model_dims", "pgx_sync", "pgx_timer_start", "pgx_timer_stop", "pgx_timer_mark", "pgx_timer_elapsed", "pgx_device_info",
    "pgx_set_points", "pgx_set_compound", "pgx_get_compound",
    "pgx_score", "pgx_score_upload", "pgx_score_launch", "pgx_score_fetch", "pgx_score_algorithmic_bytes", "pgx_score_stats", "pgx_score_profile", "pgx_score_kernel_times", "pgx_score_debug_fetch", "pgx_score_debug_geometry",
    "pgx_preference", "pgx_get_preference", "pgx_compound_update",
    "pgx_pearl_unary", "pgx_set_unary_q", "pgx_set_graph", "pgx_graph_build", "pgx_graph_fetch", "pgx_set_weights", "pgx_gram", "pgx_solve_minimal",
    "pgx_set_labels", "pgx_get_labels", "pgx_energy", "pgx_expand_alpha", "pgx_expansion", "pgx_greedy_labeling", "pgx_expansion_stats", "pgx_expansion_paths", "pgx_expansion_schedule", "pgx_one_workgroup_launches", "pgx_host_rows_with_duplicates", "pgx_host_fisher_yates_rows", "pgx_graph_size", "pgx_eigh_smallest_batch",
    "pgx_bucket", "pgx_residual_sum", "pgx_gc_labeling", "pgx_gc_inliers", "pgx_epipolar_support", "pgx_gram_batch", "pgx_gram_labels", "pgx_residual_sums", "pgx_pnp_refine_batch",
    "pgx_comm_unique_id", "pgx_comm_init", "pgx_comm_destroy", "pgx_comm_barrier", "pgx_comm_allreduce_max_f64",
    "pgx_score_allgather", "pgx_score_fetch_all", "pgx_score_allgather_begin", "pgx_score_allgather_end", "pgx_compound_allreduce_max",
    "pgx_score_inliers", "pgx_solve_minimal_sampled", "pgx_sampler_prosac_set", "pgx_score_set_global_n", "pgx_score_allreduce", "pgx_score_allreduce_begin", "pgx_score_allreduce_end",
    "pgx_pnapsac_create", "pgx_pnapsac_draw", "pgx_pnapsac_destroy",
]


GRAPH_KNN_IN_BALL, GRAPH_BALL, GRAPH_KNN = 0, 1, 2
GRAM_AFFINE, GRAM_DLT_H, GRAM_EPI_F, GRAM_VP, GRAM_PNP_GN = 0, 1, 2, 3, 4
GRAM_Q = {GRAM_DLT_H: 9, GRAM_EPI_F: 9, GRAM_VP: 3, GRAM_PNP_GN: 7}   # GRAM_AFFINE: point dimension + 1


_TRIU = {}


def tri_to_sym(tri, q):
    """upper triangle (row-major) -> full symmetric q x q matrix (index pairs cached per q: the two fancy assignments are a fifth of
    the zeros + triu_indices + triu + transpose + add this replaces - 28 us of Python per Gram call on the reference's own scenes)"""
    iu = _TRIU.get(q)
    if iu is None:
        iu = _TRIU[q] = np.triu_indices(q)
    G = np.empty((q, q))
    G[iu[1], iu[0]] = tri
    G[iu[0], iu[1]] = tri
    return G


class PgxError(RuntimeError):
    pass


_lib = None


def load():
    """Loads libpgx.so (built by `make -C progressive-x_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgxError(
            f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C progressive-x_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.pgx_global_error.restype = C.c_char_p
    lib.pgx_last_error.restype = C.c_char_p
    lib.pgx_last_error.argtypes = [C.c_void_p]
    lib.pgx_destroy.restype = None
    lib.pgx_destroy.argtypes = [C.c_void_p]
    lib.pgx_pnapsac_destroy.restype = None
    lib.pgx_pnapsac_destroy.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _ptr(a, ct):
    """the array's address for a pointer argument (no argtypes are declared, so a c_void_p is passed as is; data_as() builds a typed
    pointer object through ctypes.cast: 1 200 of them per drop-in call on a small scene were 1 ms).  `ct` documents the element type."""
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_count():
    lib = load()
    n = C.c_int(0)
    lib.pgx_device_count(C.byref(n))
    return n.value


def comm_unique_id():
    lib = load()
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    r = lib.pgx_comm_unique_id(buf)
    if r != 0:
        raise PgxError(f"pgx_comm_unique_id failed ({r}): {lib.pgx_global_error().decode()}")
    return bytes(buf)


def host_rows_with_duplicates(s):
    """pgx_host_rows_with_duplicates: bool [count] - which rows of the int64 array s [count, m] hold a repeated value (host code, no GPU)."""
    s = np.ascontiguousarray(s, dtype=np.int64)
    out = np.empty(s.shape[0], dtype=np.uint8)
    r = load().pgx_host_rows_with_duplicates(_ptr(s, C.c_int64), C.c_int64(s.shape[0]), C.c_int(s.shape[1]), _ptr(out, C.c_uint8))
    if r != 0:
        raise PgxError(f"pgx_host_rows_with_duplicates failed ({r})")
    return out.view(np.bool_)


def host_fisher_yates_rows(draws, rows, s):
    """pgx_host_fisher_yates_rows: fills the rows `rows` of s [count, m] (int64, C-contiguous, in place) from the offsets draws [k, m]."""
    draws = np.ascontiguousarray(draws, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    if s.dtype != np.int64 or not s.flags.c_contiguous:
        raise ValueError("host_fisher_yates_rows: s must be a C-contiguous int64 array")
    r = load().pgx_host_fisher_yates_rows(_ptr(draws, C.c_int64), _ptr(rows, C.c_int64), C.c_int64(rows.shape[0]), C.c_int(s.shape[1]), _ptr(s, C.c_int64))
    if r != 0:
        raise PgxError(f"pgx_host_fisher_yates_rows failed ({r})")


class PnapsacSampler:
    """pgx_pnapsac_*: Progressive NAPSAC on the in-repo generator, host code of libpgx.so (csrc/sampler_host.hip) - needs no GPU
    context.  draw() = the samples of one proposal; _rng.pnapsac_samples is the same function in numpy."""

    def __init__(self, pts, sizes, m, layers=(16, 8, 4, 2)):
        self._lib = load()
        pts = _f64(pts)
        self.n, self.m = int(pts.shape[0]), int(m)
        sz = np.zeros(4)
        s = _f64(sizes).reshape(-1)[:4]
        sz[:len(s)] = s
        lay = _i32(layers)
        h = C.c_void_p()
        r = self._lib.pgx_pnapsac_create(_ptr(pts, C.c_double), C.c_int64(self.n), C.c_int(int(pts.shape[1])), _ptr(sz, C.c_double),
                                         _ptr(lay, C.c_int32), C.c_int(len(lay)), C.c_int(self.m), C.byref(h))
        if r != 0:
            raise PgxError(f"pgx_pnapsac_create failed ({r}): {self._lib.pgx_global_error().decode()}")
        self._h = h

    def draw(self, key, batch, count, tops, growth_local, max_local):
        tops = _i32(tops)
        growth = np.ascontiguousarray(growth_local, dtype=np.int64)
        if len(tops) < count or len(growth) < self.n:
            raise ValueError("pgx_pnapsac_draw: one subset size per sample and one growth entry per point")
        out = np.empty((int(count), self.m), dtype=np.int32)
        r = self._lib.pgx_pnapsac_draw(self._h, C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF), C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int32(int(count)),
                                       _ptr(tops, C.c_int32), _ptr(growth, C.c_int64), C.c_int64(int(max_local)), _ptr(out, C.c_int32))
        if r != 0:
            raise PgxError(f"pgx_pnapsac_draw failed ({r}): {self._lib.pgx_global_error().decode()}")
        return out.astype(np.int64)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pgx_pnapsac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU context (pgx_ctx): resident points, compound vector, preference slots, unary table, graph, labels."""

    def __init__(self, device_id=0):
        self._lib = load()
        h = C.c_void_p()
        r = self._lib.pgx_create(C.c_int(int(device_id)), C.byref(h))
        if r != 0:
            raise PgxError(f"pgx_create(device={device_id}) failed ({r}): {self._lib.pgx_global_error().decode()}")
        self._h = h
        self.device_id = int(device_id)
        self.model_type = None
        self.n = 0
        self.M = 0
        self.nranks = 1
        self.rank = 0

    # -- plumbing ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pgx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, r, what):
        if r != 0:
            raise PgxError(f"{what} failed ({r}): {self._lib.pgx_last_error(self._h).decode()}")

    def sync(self):
        self._ck(self._lib.pgx_sync(self._h), "pgx_sync")

    def timer_start(self):
        self._ck(self._lib.pgx_timer_start(self._h), "pgx_timer_start")

    def timer_mark(self):
        self._ck(self._lib.pgx_timer_mark(self._h), "pgx_timer_mark")

    def timer_elapsed(self):
        ms = C.c_float()
        self._ck(self._lib.pgx_timer_elapsed(self._h, C.byref(ms)), "pgx_timer_elapsed")
        return float(ms.value)

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self._lib.pgx_timer_stop(self._h, C.byref(ms)), "pgx_timer_stop")
        return float(ms.value)

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        hbm = C.c_int64()
        self._ck(self._lib.pgx_device_info(self._h, name, C.c_int(256), C.byref(cu), C.byref(hbm)), "pgx_device_info")
        return dict(name=name.value.decode(), cu_count=cu.value, hbm_bytes=hbm.value)

    # -- resident data -------------------------------------------------------------------------------------------
    def set_points(self, model_type, points):
        pts = _f64(points)
        if pts.ndim != 2 or pts.shape[1] != POINT_DIM[model_type]:
            raise ValueError(f"points must be [n,{POINT_DIM[model_type]}] for model type {model_type}")
        self._ck(self._lib.pgx_set_points(self._h, C.c_int(model_type), _ptr(pts, C.c_double), C.c_int64(pts.shape[0])),
                 "pgx_set_points")
        self.model_type = model_type
        self.n = pts.shape[0]
        self._w_obj = None          # resident weights belong to a point set

    def set_weights(self, weights):
        """pgx_set_weights: per-point weights of the weighted refits, uploaded once and checked against n (None clears)."""
        if weights is None or len(weights) == 0:
            self._ck(self._lib.pgx_set_weights(self._h, None, C.c_int64(0)), "pgx_set_weights")
            self._w_obj = None
            return
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.shape[0] != self.n:
            raise ValueError(f"weights must have one entry per point ({self.n}), got {w.shape[0]}")
        self._ck(self._lib.pgx_set_weights(self._h, _ptr(w, C.c_double), C.c_int64(w.shape[0])), "pgx_set_weights")
        self._w_obj = weights

    def _use_weights(self, weights):
        """1 when `weights` (the same array object as last time, or a new one that is uploaded now) is to be used.  The
        arrays are treated as immutable: a caller that edits its weights in place calls set_weights again."""
        if weights is None or len(weights) == 0:
            return 0
        if weights is not getattr(self, "_w_obj", None):
            self.set_weights(weights)
        return 1

    def set_compound(self, compound=None):
        c = None if compound is None else _f64(compound)
        self._ck(self._lib.pgx_set_compound(self._h, _ptr(c, C.c_double)), "pgx_set_compound")

    def get_compound(self):
        out = np.empty(self.n, dtype=np.float64)
        self._ck(self._lib.pgx_get_compound(self._h, _ptr(out, C.c_double)), "pgx_get_compound")
        return out

    # -- scoring ---------------------------------------------------------------------------------------------------
    def _models(self, models):
        m = _f64(models).reshape(-1, PARAM_DIM[self.model_type])
        return m

    def score(self, models, T2, has_compound=False, exponent=2, want_masks=False):
        m = self._models(models)
        M = m.shape[0]
        counts = np.empty(M, dtype=np.int64)
        values = np.empty(M, dtype=np.float64)
        shared = np.empty(M, dtype=np.float64)
        scores = np.empty(M, dtype=np.float64)
        masks = np.empty((M, (self.n + 63) // 64), dtype=np.uint64) if want_masks else None
        self._ck(self._lib.pgx_score(self._h, _ptr(m, C.c_double), C.c_int(M), C.c_double(T2),
                                     C.c_int(1 if has_compound else 0), C.c_int(int(exponent)),
                                     _ptr(counts, C.c_int64), _ptr(values, C.c_double), _ptr(shared, C.c_double),
                                     _ptr(scores, C.c_double), _ptr(masks, C.c_uint64)), "pgx_score")
        self.M = M
        return dict(counts=counts, values=values, shared=shared, scores=scores, masks=masks)

    def score_upload(self, models):
        m = self._models(models)
        self._ck(self._lib.pgx_score_upload(self._h, _ptr(m, C.c_double), C.c_int(m.shape[0])), "pgx_score_upload")
        self.M = m.shape[0]

    def score_launch(self, T2, has_compound=False, want_masks=False):
        self._ck(self._lib.pgx_score_launch(self._h, C.c_double(T2), C.c_int(1 if has_compound else 0),
                                            C.c_int(1 if want_masks else 0)), "pgx_score_launch")

    def score_buffers(self):
        """Reusable result buffers for score_fetch(out=...): a loop that fetches every step saves the four allocations and
        pointer conversions per call (the arrays are overwritten by the next fetch)."""
        M = self.M
        buf = dict(counts=np.empty(M, dtype=np.int64), values=np.empty(M, dtype=np.float64),
                   shared=np.empty(M, dtype=np.float64), scores=np.empty(M, dtype=np.float64), masks=None)
        buf["_ptrs"] = (_ptr(buf["counts"], C.c_int64), _ptr(buf["values"], C.c_double), _ptr(buf["shared"], C.c_double),
                        _ptr(buf["scores"], C.c_double))
        return buf

    def score_fetch(self, exponent=2, want_masks=False, out=None):
        if out is not None and not want_masks and out["counts"].shape[0] == self.M:
            pc, pv, ps, pq = out["_ptrs"]
            self._ck(self._lib.pgx_score_fetch(self._h, C.c_int(int(exponent)), pc, pv, ps, pq, None), "pgx_score_fetch")
            return out
        M = self.M
        counts = np.empty(M, dtype=np.int64)
        values = np.empty(M, dtype=np.float64)
        shared = np.empty(M, dtype=np.float64)
        scores = np.empty(M, dtype=np.float64)
        masks = np.empty((M, (self.n + 63) // 64), dtype=np.uint64) if want_masks else None
        self._ck(self._lib.pgx_score_fetch(self._h, C.c_int(int(exponent)), _ptr(counts, C.c_int64),
                                           _ptr(values, C.c_double), _ptr(shared, C.c_double),
                                           _ptr(scores, C.c_double), _ptr(masks, C.c_uint64)), "pgx_score_fetch")
        return dict(counts=counts, values=values, shared=shared, scores=scores, masks=masks)

    def score_algorithmic_bytes(self, want_masks=False):
        b = C.c_int64()
        p = C.c_int64()
        self._ck(self._lib.pgx_score_algorithmic_bytes(self._h, C.c_int(1 if want_masks else 0), C.byref(b),
                                                       C.byref(p)), "pgx_score_algorithmic_bytes")
        return b.value, p.value

    def score_inliers(self, row=0):
        """pgx_score_inliers: ascending indices (int64) of the inliers of hypothesis `row` of the last launch with masks"""
        if not hasattr(self, "_inl_index") or self._inl_index.shape[0] != self.n:
            self._inl_index = np.empty(self.n, dtype=np.int32)
        cnt = C.c_int64()
        self._ck(self._lib.pgx_score_inliers(self._h, C.c_int(int(row)), _ptr(self._inl_index, C.c_int32), C.byref(cnt)), "pgx_score_inliers")
        return self._inl_index[:cnt.value].astype(np.int64)

    def score_debug_fetch(self, what):
        """pgx_score_debug_fetch: 'order' | 'bounds' | 'rows64' | 'rows32' | 'rows32_sorted' of the resident point set"""
        d = POINT_DIM[self.model_type]
        groups = (self.n + 63) // 64
        supers = (groups + 7) // 8
        spec = {"order": (0, np.int32, (self.n,)), "bounds": (1, np.float32, (groups + supers, 12)),
                "rows64": (2, np.float64, (groups, d, 64)), "rows32": (3, np.float32, (groups, 8, 64)),
                "rows32_sorted": (4, np.float32, (self.n, 8))}[what]
        out = np.empty(spec[2], dtype=spec[1])
        self._ck(self._lib.pgx_score_debug_fetch(self._h, C.c_int(spec[0]), out.ctypes.data_as(C.c_void_p), C.c_int64(out.nbytes)),
                 "pgx_score_debug_fetch")
        return out

    def score_debug_geometry(self, **kw):
        """pgx_score_debug_geometry (test hook): split=, group_xcd=, nrep=, dense_min=, cull_segs="""
        keys = {"split": 0, "group_xcd": 1, "nrep": 2, "dense_min": 3, "cull_segs": 4}
        for k, v in kw.items():
            self._ck(self._lib.pgx_score_debug_geometry(self._h, C.c_int(keys[k]), C.c_int(int(v))), "pgx_score_debug_geometry")

    def score_profile(self, on=1):
        """0 off; 1 = HIP events around the dominant scoring kernel only; 2 (or True) = around every kernel of a launch"""
        level = 2 if on is True else int(on)
        self._ck(self._lib.pgx_score_profile(self._h, C.c_int(level)), "pgx_score_profile")

    def score_kernel_times(self):
        """ms of (cull or chunked kernel, group-major kernel, finish/reduce, exact evaluation of the queued candidates) of
        the last profiled scoring launch"""
        ms = (C.c_float * 4)()
        self._ck(self._lib.pgx_score_kernel_times(self._h, ms), "pgx_score_kernel_times")
        return float(ms[0]), float(ms[1]), float(ms[2]), float(ms[3])

    def score_stats(self, T2, has_compound=False):
        """pgx_score_stats: work counters of one (untimed) scoring launch of the resident batch."""
        st = np.zeros(8, dtype=np.int64)
        self._ck(self._lib.pgx_score_stats(self._h, C.c_double(T2), C.c_int(1 if has_compound else 0), _ptr(st, C.c_int64)),
                 "pgx_score_stats")
        return dict(pairs=int(st[0]), group_pairs=int(st[1]), surviving_group_steps=int(st[2]), exact_evaluations=int(st[3]),
                    inlier_pairs=int(st[4]), contradictions=int(st[5]), path={1: "every pair", 2: "cull + group-major"}.get(int(st[6]), "?"),
                    filter={0: "none", 1: "f64", 2: "f32"}.get(int(st[7]), "?"))

    # -- preference / compound -------------------------------------------------------------------------------------
    def preference(self, model, T2, slot, want_pref=False):
        m = _f64(model).reshape(-1)
        out = np.empty(self.n, dtype=np.float64) if want_pref else None
        d, a, b = C.c_double(), C.c_double(), C.c_double()
        self._ck(self._lib.pgx_preference(self._h, _ptr(m, C.c_double), C.c_double(T2), C.c_int(int(slot)),
                                          _ptr(out, C.c_double), C.byref(d), C.byref(a), C.byref(b)),
                 "pgx_preference")
        return dict(pref=out, dot=d.value, pref_sqnorm=a.value, comp_sqnorm=b.value)

    def get_preference(self, slot):
        out = np.empty(self.n, dtype=np.float64)
        self._ck(self._lib.pgx_get_preference(self._h, C.c_int(int(slot)), _ptr(out, C.c_double)),
                 "pgx_get_preference")
        return out

    def compound_update(self, slots, want_compound=False):
        s = _i32(slots).reshape(-1)
        out = np.empty(self.n, dtype=np.float64) if want_compound else None
        self._ck(self._lib.pgx_compound_update(self._h, _ptr(s, C.c_int32), C.c_int(s.shape[0]),
                                               _ptr(out, C.c_double)), "pgx_compound_update")
        return out

    # -- PEARL -------------------------------------------------------------------------------------------------------
    def pearl_unary(self, models, threshold, lam, want_table=False):
        K = 0 if models is None else int(np.asarray(models).size // PARAM_DIM[self.model_type])
        m = self._models(models) if K > 0 else None
        out = np.empty((self.n, K + 1), dtype=np.int64) if want_table else None
        self._ck(self._lib.pgx_pearl_unary(self._h, _ptr(m, C.c_double), C.c_int(K), C.c_double(threshold),
                                           C.c_double(lam), _ptr(out, C.c_int64)), "pgx_pearl_unary")
        self.L = K + 1
        self._dq_n = self.n
        return out

    def set_unary_q(self, Dq):
        Dq = np.ascontiguousarray(Dq, dtype=np.int64)
        self._ck(self._lib.pgx_set_unary_q(self._h, _ptr(Dq, C.c_int64), C.c_int64(Dq.shape[0]),
                                           C.c_int(Dq.shape[1])), "pgx_set_unary_q")
        self.L = Dq.shape[1]
        self._dq_n = Dq.shape[0]
        if self.n == 0:
            self.n = Dq.shape[0]

    def set_graph(self, off, idx, mult):
        off, idx, mult = _i32(off), _i32(idx), _i32(mult)
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.int32)
            mult = np.ones(1, dtype=np.int32)
        self._ck(self._lib.pgx_set_graph(self._h, C.c_int64(off.shape[0] - 1), _ptr(off, C.c_int32),
                                         _ptr(idx, C.c_int32), _ptr(mult, C.c_int32)), "pgx_set_graph")

    def graph_build(self, points, kind, radius=0.0, k=5, fetch=True):
        """Neighbourhood graph on the GPU (pgx_graph_build); leaves it resident and returns the CSR when `fetch`."""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        arcs = C.c_int64()
        self._ck(self._lib.pgx_graph_build(self._h, _ptr(pts, C.c_double), C.c_int64(pts.shape[0]), C.c_int(pts.shape[1]),
                                           C.c_int(int(kind)), C.c_double(float(radius)), C.c_int(int(k)), C.byref(arcs)),
                 "pgx_graph_build")
        if not fetch:
            return arcs.value
        return self.graph_fetch()

    def graph_size(self):
        """(sites, directed arcs) of the graph resident now (pgx_graph_size); (0, 0) when none."""
        n, arcs = C.c_int64(), C.c_int64()
        self._ck(self._lib.pgx_graph_size(self._h, C.byref(n), C.byref(arcs)), "pgx_graph_size")
        return int(n.value), int(arcs.value)

    def graph_fetch(self):
        """CSR (off, idx, mult) of the resident graph (pgx_graph_fetch).  The buffers are sized by what is resident NOW - asked of
        the library, not remembered from the last graph_build: set_graph / set_points may have replaced the graph since."""
        n, arcs = self.graph_size()
        if n <= 0:
            raise PgxError("graph_fetch: no graph is resident on this context")
        off = np.empty(n + 1, dtype=np.int32)
        idx = np.empty(max(arcs, 1), dtype=np.int32)
        mult = np.empty(max(arcs, 1), dtype=np.int32)
        self._ck(self._lib.pgx_graph_fetch(self._h, _ptr(off, C.c_int32), _ptr(idx, C.c_int32), _ptr(mult, C.c_int32)),
                 "pgx_graph_fetch")
        return off, idx[:arcs], mult[:arcs]

    def solve_minimal(self, samples, fetch=True):
        """pgx_solve_minimal: hypotheses of the 2-point line / 2-segment vanishing point solvers, generated from the
        resident points straight into the resident hypothesis buffer (score_launch can follow).  NaN rows mark
        degenerate samples."""
        smp = _i32(samples)
        want = {FUNDAMENTAL: 7, HOMOGRAPHY: 4, PNP: 3}.get(self.model_type, 2)
        if smp.ndim != 2 or smp.shape[1] != want:
            raise ValueError(f"samples must be [S,{want}]")
        rows = smp.shape[0] * {FUNDAMENTAL: 3, PNP: 4}.get(self.model_type, 1)   # root slots per 7-point / P3P sample
        out = np.empty((rows, PARAM_DIM[self.model_type]), dtype=np.float64) if fetch else None
        self._ck(self._lib.pgx_solve_minimal(self._h, _ptr(smp, C.c_int32), C.c_int(smp.shape[0]), _ptr(out, C.c_double)),
                 "pgx_solve_minimal")
        self.M = rows
        return out

    def solve_minimal_sampled(self, key, batch, S, fetch=True, fetch_samples=False, sampler="uniform"):
        """pgx_solve_minimal_sampled: S minimal samples drawn on the device by the in-repo generator (_rng.py gives the same rows:
        sampler "uniform", "napsac" on the resident neighbourhood graph, or "prosac" with the table of sampler_prosac_set) and solved into the resident hypothesis buffer.
        Returns (models or None, samples or None)."""
        m = {FUNDAMENTAL: 7, HOMOGRAPHY: 4, PNP: 3}.get(self.model_type, 2)
        rows = int(S) * {FUNDAMENTAL: 3, PNP: 4}.get(self.model_type, 1)
        out = np.empty((rows, PARAM_DIM[self.model_type]), dtype=np.float64) if fetch else None
        smp = np.empty((int(S), m), dtype=np.int32) if fetch_samples else None
        self._ck(self._lib.pgx_solve_minimal_sampled(self._h, C.c_int({"uniform": 0, "napsac": 1, "prosac": 2}[sampler]), C.c_uint64(int(key) & 0xFFFFFFFFFFFFFFFF),
                                                     C.c_uint32(int(batch) & 0xFFFFFFFF), C.c_int(int(S)), _ptr(smp, C.c_int32), _ptr(out, C.c_double)),
                 "pgx_solve_minimal_sampled")
        self.M = rows
        return out, smp

    def sampler_prosac_set(self, subset_sizes):
        """pgx_sampler_prosac_set: PROSAC's subset size n_k per sample number k = 1 .. len (0 = uniform), for the resident points"""
        t = np.ascontiguousarray(subset_sizes, dtype=np.int32)
        self._ck(self._lib.pgx_sampler_prosac_set(self._h, _ptr(t, C.c_int32), C.c_int(len(t))), "pgx_sampler_prosac_set")

    def gram(self, kind, sel, params=None, weights=None, wpow=2):
        """pgx_gram: weighted Gram matrix of the design rows of the selected resident points.
        sel = ("index", int array) or ("label", k).  Returns (G [q,q] symmetric, count, bad)."""
        q = GRAM_Q.get(kind, POINT_DIM[self.model_type] + 1)
        out = np.zeros(q * (q + 1) // 2, dtype=np.float64)
        prm = None if params is None else np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        use_w = self._use_weights(weights)
        cnt, bad = C.c_int64(), C.c_int64()
        if sel[0] == "index":
            idx = _i32(sel[1])
            mode, iptr, m, label = 0, _ptr(idx, C.c_int32) if idx.size else None, idx.size, 0
        else:
            mode, iptr, m, label = 1, None, 0, int(sel[1])
        self._ck(self._lib.pgx_gram(self._h, C.c_int(int(kind)), _ptr(prm, C.c_double), C.c_int(0 if prm is None else prm.size),
                                    C.c_int(mode), iptr, C.c_int64(m), C.c_int(label), C.c_int(use_w),
                                    C.c_int(int(wpow)), _ptr(out, C.c_double), C.byref(cnt), C.byref(bad)), "pgx_gram")
        return tri_to_sym(out, q), cnt.value, bad.value

    def gram_labels(self, kind, K, params=None, weights=None, wpow=2):
        """pgx_gram_labels: the Gram matrices of the points with label k under parameter block k, k = 0..K-1, in one launch.
        Returns (G [K, q, q], count [K], bad [K]); G[k] is bit-identical to gram(kind, ("label", k), params[k])[0]."""
        q = GRAM_Q.get(kind, POINT_DIM[self.model_type] + 1)
        nv = q * (q + 1) // 2
        out = np.zeros((K, nv), dtype=np.float64)
        cnt = np.zeros(K, dtype=np.int64)
        bad = np.zeros(K, dtype=np.int64)
        prm = None if params is None else np.ascontiguousarray(params, dtype=np.float64).reshape(K, -1)
        use_w = self._use_weights(weights)
        self._ck(self._lib.pgx_gram_labels(self._h, C.c_int(int(kind)), _ptr(prm, C.c_double), C.c_int(0 if prm is None else prm.shape[1]),
                                           C.c_int(int(K)), C.c_int(use_w), C.c_int(int(wpow)), _ptr(out, C.c_double),
                                           _ptr(cnt, C.c_int64), _ptr(bad, C.c_int64)), "pgx_gram_labels")
        G = np.zeros((K, q, q))
        iu = np.triu_indices(q)
        G[:, iu[0], iu[1]] = out
        G[:, iu[1], iu[0]] = out
        return G, cnt, bad

    def residual_sums(self, models):
        """pgx_residual_sums: sum of the unsquared residuals of model k over the points labelled k, all k in one launch."""
        m = np.ascontiguousarray(models, dtype=np.float64)
        K = m.shape[0]
        out = np.zeros(K, dtype=np.float64)
        self._ck(self._lib.pgx_residual_sums(self._h, _ptr(m, C.c_double), C.c_int(K), _ptr(out, C.c_double)), "pgx_residual_sums")
        return out

    def gram_batch(self, kind, index, params=None, weights=None, wpow=2):
        """pgx_gram_batch: B selections of m resident points each (index [B, m]) in one launch; params [B, np] or None;
        weights = the full per-point weight vector (gathered here).  Returns (G [B, q, q] symmetric, bad [B])."""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        B, m = idx.shape
        q = GRAM_Q.get(kind, POINT_DIM[self.model_type] + 1)
        nv = q * (q + 1) // 2
        out = np.zeros((B, nv), dtype=np.float64)
        bad = np.zeros(B, dtype=np.int32)
        prm = None if params is None else np.ascontiguousarray(params, dtype=np.float64).reshape(B, -1)
        w = None
        if weights is not None and len(weights) > 0:
            wf = np.asarray(weights, dtype=np.float64).reshape(-1)
            if wf.shape[0] != self.n:
                raise ValueError(f"weights must have one entry per point ({self.n}), got {wf.shape[0]}")
            w = np.ascontiguousarray(wf[idx])
        self._ck(self._lib.pgx_gram_batch(self._h, C.c_int(int(kind)), _ptr(prm, C.c_double),
                                          C.c_int(0 if prm is None else prm.shape[1]), _ptr(idx, C.c_int32) if idx.size else None,
                                          C.c_int(B), C.c_int(m), _ptr(w, C.c_double), C.c_int(int(wpow)),
                                          _ptr(out, C.c_double), _ptr(bad, C.c_int32)), "pgx_gram_batch")
        G = np.zeros((B, q, q))
        iu = np.triu_indices(q)
        G[:, iu[0], iu[1]] = out
        G[:, iu[1], iu[0]] = out
        return G, bad

    def pnp_refine_batch(self, inits, index, weights=None, wpow=2, iterations=10):
        """pgx_pnp_refine_batch: the Gauss-Newton pose refit of B selections (index [B, m]) from inits [B, 12] in one launch.
        weights = the full per-point weight vector (gathered here).  Returns (poses [B, 12], ok [B] bool)."""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        B, m = idx.shape
        ini = np.ascontiguousarray(inits, dtype=np.float64).reshape(B, 12)
        out = np.zeros((B, 12), dtype=np.float64)
        status = np.zeros(B, dtype=np.int32)
        w = None
        if weights is not None and len(weights) > 0:
            wf = np.asarray(weights, dtype=np.float64).reshape(-1)
            if wf.shape[0] != self.n:
                raise ValueError(f"weights must have one entry per point ({self.n}), got {wf.shape[0]}")
            w = np.ascontiguousarray(wf[idx])
        self._ck(self._lib.pgx_pnp_refine_batch(self._h, _ptr(ini, C.c_double), _ptr(idx, C.c_int32) if idx.size else None,
                                                C.c_int(B), C.c_int(m), _ptr(w, C.c_double), C.c_int(int(wpow)),
                                                C.c_int(int(iterations)), _ptr(out, C.c_double), _ptr(status, C.c_int32)),
                 "pgx_pnp_refine_batch")
        return out, status != 0

    def set_labels(self, labels):
        lab = _i32(labels)
        self._ck(self._lib.pgx_set_labels(self._h, _ptr(lab, C.c_int32), C.c_int64(lab.shape[0])), "pgx_set_labels")
        self._nlabels = lab.shape[0]

    def get_labels(self):
        out = np.empty(self._nlabels, dtype=np.int32)
        self._ck(self._lib.pgx_get_labels(self._h, _ptr(out, C.c_int32)), "pgx_get_labels")
        return out

    def energy(self, lam, label_cost):
        eq = C.c_int64()
        e = C.c_double()
        self._ck(self._lib.pgx_energy(self._h, C.c_double(lam), C.c_double(label_cost), C.byref(eq), C.byref(e)),
                 "pgx_energy")
        return eq.value, e.value

    def expand_alpha(self, lam, label_cost, alpha):
        ch = C.c_int64()
        self._ck(self._lib.pgx_expand_alpha(self._h, C.c_double(lam), C.c_double(label_cost), C.c_int(int(alpha)),
                                            C.byref(ch)), "pgx_expand_alpha")
        return ch.value

    def expansion(self, lam, label_cost, max_cycles=1000):
        eq = C.c_int64()
        e = C.c_double()
        cyc = C.c_int()
        self._ck(self._lib.pgx_expansion(self._h, C.c_double(lam), C.c_double(label_cost), C.c_int(int(max_cycles)),
                                         C.byref(eq), C.byref(e), C.byref(cyc)), "pgx_expansion")
        return eq.value, e.value, cyc.value

    def greedy_labeling(self, label_cost):
        """pgx_greedy_labeling [U-8]: GCO-v3's labelling of an energy without smooth costs -> (energy_q, energy, opened)"""
        eq, e, op = C.c_int64(), C.c_double(), C.c_int()
        self._ck(self._lib.pgx_greedy_labeling(self._h, C.c_double(label_cost), C.byref(eq), C.byref(e), C.byref(op)),
                 "pgx_greedy_labeling")
        self._nlabels = self._dq_n          # the labelling now covers the sites of the unary table
        return eq.value, e.value, op.value

    def expansion_stats(self):
        st = np.zeros(8, dtype=np.int64)
        self._ck(self._lib.pgx_expansion_stats(self._h, _ptr(st, C.c_int64)), "pgx_expansion_stats")
        return dict(mincuts=int(st[0]), sweeps=int(st[1]), global_relabels=int(st[2]), bfs_levels=int(st[3]),
                    relabelled_sites=int(st[4]), wave_passes=int(st[5]), list_sweeps=int(st[6]), skipped_moves=int(st[7]))

    def expansion_paths(self):
        """Which min-cut solver finished the moves of this context so far (include/pgx.h pgx_expansion_paths)."""
        st = np.zeros(6, dtype=np.int64)
        self._ck(self._lib.pgx_expansion_paths(self._h, _ptr(st, C.c_int64)), "pgx_expansion_paths")
        return dict(one_workgroup=int(st[0]), memo=int(st[1]), region=int(st[2]), level_synchronous=int(st[3]),
                    region_declined=int(st[4]), tile_handed_back=int(st[5]))

    def eigh_smallest_batch(self, A):
        """pgx_eigh_smallest_batch: (vec [B, q], val [B]) - eigenvector and eigenvalue of the smallest eigenvalue of the symmetric
        matrices A [B, q, q], q <= 9 (cyclic Jacobi on the device, one lane per matrix; bitwise the CPU restatement the tests hold)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        if A.ndim != 3 or A.shape[1] != A.shape[2]:
            raise PgxError("eigh_smallest_batch: A must be [B, q, q]")
        B, q = A.shape[0], A.shape[1]
        vec = np.zeros((B, q), dtype=np.float64)
        val = np.zeros(B, dtype=np.float64)
        self._ck(self._lib.pgx_eigh_smallest_batch(self._h, _ptr(A, C.c_double), C.c_int(q), C.c_int64(B), _ptr(vec, C.c_double),
                                                   _ptr(val, C.c_double)), "pgx_eigh_smallest_batch")
        return vec, val

    def expansion_schedule(self):
        """How the level-synchronous min-cut spent its dependent steps (include/pgx.h pgx_expansion_schedule)."""
        st = np.zeros(8, dtype=np.int64)
        self._ck(self._lib.pgx_expansion_schedule(self._h, _ptr(st, C.c_int64)), "pgx_expansion_schedule")
        return dict(xcd_round_launches=int(st[0]), xcd_rounds=int(st[1]), xcd_launches_finished=int(st[2]), xcd_searches=int(st[3]),
                    xcd_list_sites=int(st[4]), list_sites=int(st[5]), xcd_levels=int(st[6]), xcd_sweeps=int(st[7]))

    def one_workgroup_launches(self):
        """Whole-graph one-workgroup moves enqueued so far, by kernel (include/pgx.h pgx_one_workgroup_launches)."""
        st = np.zeros(2, dtype=np.int64)
        self._ck(self._lib.pgx_one_workgroup_launches(self._h, _ptr(st, C.c_int64)), "pgx_one_workgroup_launches")
        return dict(lds_resident=int(st[0]), memory_resident=int(st[1]))

    def bucket(self, L, want_order=True):
        counts = np.zeros(L, dtype=np.int64)
        order = np.empty(self._nlabels, dtype=np.int32) if want_order else None
        self._ck(self._lib.pgx_bucket(self._h, C.c_int(int(L)), _ptr(counts, C.c_int64), _ptr(order, C.c_int32)),
                 "pgx_bucket")
        return counts, order

    def residual_sum(self, model, label):
        m = _f64(model).reshape(-1)
        s = C.c_double()
        self._ck(self._lib.pgx_residual_sum(self._h, _ptr(m, C.c_double), C.c_int(int(label)), C.byref(s)),
                 "pgx_residual_sum")
        return s.value

    def epipolar_support(self, F, T2, S2):
        """pgx_epipolar_support: (Sampson inliers of F, those also within S2 of the symmetric epipolar distance)."""
        f = _f64(F).reshape(-1)
        out = np.zeros(2, dtype=np.int64)
        self._ck(self._lib.pgx_epipolar_support(self._h, _ptr(f, C.c_double), C.c_double(float(T2)), C.c_double(float(S2)),
                                                _ptr(out, C.c_int64)), "pgx_epipolar_support")
        return int(out[0]), int(out[1])

    def gc_inliers(self, model, T2, lam):
        """pgx_gc_inliers: the inliers' indices (ascending, int64) of the inlier/outlier graph cut of `model`."""
        m = _f64(model).reshape(-1)
        if not hasattr(self, "_gc_index") or self._gc_index.shape[0] != self.n:
            self._gc_index = np.empty(self.n, dtype=np.int32)
        cnt = C.c_int64()
        self._ck(self._lib.pgx_gc_inliers(self._h, _ptr(m, C.c_double), C.c_double(float(T2)), C.c_double(float(lam)),
                                          _ptr(self._gc_index, C.c_int32), C.byref(cnt)), "pgx_gc_inliers")
        return self._gc_index[:cnt.value].astype(np.int64)

    def gc_labeling(self, model, T2, lam):
        """GC-RANSAC's inlier/outlier graph cut of `model` on the resident graph (include/pgx.h pgx_gc_labeling):
        returns the int32 flags (1 = inlier)."""
        m = _f64(model).reshape(-1)
        flags = np.empty(self.n, dtype=np.int32)
        cnt = C.c_int64()
        self._ck(self._lib.pgx_gc_labeling(self._h, _ptr(m, C.c_double), C.c_double(float(T2)), C.c_double(float(lam)),
                                           _ptr(flags, C.c_int32), C.byref(cnt)), "pgx_gc_labeling")
        return flags

    # -- multi-GPU ---------------------------------------------------------------------------------------------------
    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self._lib.pgx_comm_init(self._h, C.c_int(int(nranks)), C.c_int(int(rank)), buf), "pgx_comm_init")
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_destroy(self):
        self._ck(self._lib.pgx_comm_destroy(self._h), "pgx_comm_destroy")
        self.nranks, self.rank = 1, 0

    def comm_barrier(self):
        self._ck(self._lib.pgx_comm_barrier(self._h), "pgx_comm_barrier")

    def comm_allreduce_max(self, value):
        v = C.c_double(float(value))
        self._ck(self._lib.pgx_comm_allreduce_max_f64(self._h, C.byref(v)), "pgx_comm_allreduce_max_f64")
        return v.value

    def score_allgather(self):
        self._ck(self._lib.pgx_score_allgather(self._h), "pgx_score_allgather")

    def score_fetch_all(self, exponent=2):
        T = self.M * self.nranks
        counts = np.empty(T, dtype=np.int64)
        values = np.empty(T, dtype=np.float64)
        shared = np.empty(T, dtype=np.float64)
        scores = np.empty(T, dtype=np.float64)
        self._ck(self._lib.pgx_score_fetch_all(self._h, C.c_int(int(exponent)), _ptr(counts, C.c_int64),
                                               _ptr(values, C.c_double), _ptr(shared, C.c_double),
                                               _ptr(scores, C.c_double)), "pgx_score_fetch_all")
        return dict(counts=counts, values=values, shared=shared, scores=scores)

    def score_allgather_begin(self, slot):
        """pgx_score_allgather_begin: the exchange of the launch just issued, on the exchange stream (slot 0 / 1)"""
        self._ck(self._lib.pgx_score_allgather_begin(self._h, C.c_int(int(slot))), "pgx_score_allgather_begin")
        self._slot_M = getattr(self, "_slot_M", {})
        self._slot_M[int(slot)] = self.M

    def score_allgather_end(self, slot, exponent=2):
        T = self._slot_M[int(slot)] * self.nranks
        counts = np.empty(T, dtype=np.int64)
        values = np.empty(T, dtype=np.float64)
        shared = np.empty(T, dtype=np.float64)
        scores = np.empty(T, dtype=np.float64)
        self._ck(self._lib.pgx_score_allgather_end(self._h, C.c_int(int(slot)), C.c_int(int(exponent)), _ptr(counts, C.c_int64),
                                                   _ptr(values, C.c_double), _ptr(shared, C.c_double), _ptr(scores, C.c_double)),
                 "pgx_score_allgather_end")
        return dict(counts=counts, values=values, shared=shared, scores=scores)

    # -- point-sharded scoring: this context holds a slice of the job's points (include/pgx.h pgx_score_allreduce)
    def score_set_global_n(self, n_total):
        """the point count of the whole job: the ranks then accumulate in one fixed-point scale (0 = this context's own n)"""
        self._ck(self._lib.pgx_score_set_global_n(self._h, C.c_int64(int(n_total))), "pgx_score_set_global_n")

    def score_allreduce(self):
        """after score_launch: sums the launch's integer accumulators over the ranks; score_fetch then returns the job's table"""
        self._ck(self._lib.pgx_score_allreduce(self._h), "pgx_score_allreduce")

    def score_allreduce_begin(self, slot):
        self._ck(self._lib.pgx_score_allreduce_begin(self._h, C.c_int(int(slot))), "pgx_score_allreduce_begin")
        self._slot_M = getattr(self, "_slot_M", {})
        self._slot_M[int(slot)] = self.M

    def score_allreduce_end(self, slot, exponent=2):
        M = self._slot_M[int(slot)]
        counts = np.empty(M, dtype=np.int64)
        values = np.empty(M, dtype=np.float64)
        shared = np.empty(M, dtype=np.float64)
        scores = np.empty(M, dtype=np.float64)
        self._ck(self._lib.pgx_score_allreduce_end(self._h, C.c_int(int(slot)), C.c_int(int(exponent)), _ptr(counts, C.c_int64),
                                                   _ptr(values, C.c_double), _ptr(shared, C.c_double), _ptr(scores, C.c_double)),
                 "pgx_score_allreduce_end")
        return dict(counts=counts, values=values, shared=shared, scores=scores)

    def score_accumulators(self):
        """pgx_score_debug_fetch(5): the last launch's integer accumulators in the caller's order, dict of uint64 [M] arrays"""
        Mpad = (self.M + 255) // 256 * 256
        out = np.empty((3, Mpad), dtype=np.uint64)
        self._ck(self._lib.pgx_score_debug_fetch(self._h, C.c_int(5), out.ctypes.data_as(C.c_void_p), C.c_int64(out.nbytes)),
                 "pgx_score_debug_fetch")
        return dict(counts=out[0, :self.M].copy(), values_q=out[1, :self.M].copy(), shared_q=out[2, :self.M].copy())

    def compound_allreduce_max(self):
        self._ck(self._lib.pgx_compound_allreduce_max(self._h), "pgx_compound_allreduce_max")
