"""Estimators: batched minimal solvers (hypothesis generation) and non-minimal refits, host side (numpy).

Replaces the Estimator concept of the reference (`sampleSize()`, `nonMinimalSampleSize()`, `estimateModel`,
`estimateModelNonminimal`; exemplar in-tree: vanishing_point_estimator.h:34-226 and
solver_vanishing_point_two_lines.h:128-237, paths relative to /root/reference/src/pyprogressivex/include/).
Only the vanishing-point solver exists in the snapshot; the line / homography / fundamental / PnP solvers live in the
absent graph-cut-ransac submodule and are restated from the literature [UPSTREAM-MEMORY] — SURVEY.md §8f ranks them
"next" (GPU versions), so round 1 keeps them on the host: they produce the hypothesis batches that the HIP scorer
consumes and the refits PEARL asks for.  Residuals are NOT computed here — that is libpgx's job.
"""
import numpy as np

from . import _lib


class Estimator:
    model_type = None
    sample_size = 0
    nonminimal_sample_size = 0
    device_minimal = False     # True: ctx.solve_minimal(samples) replaces minimal() in the proposal engine
    device_slots = 1           # hypotheses the device solver emits per sample (NaN = none)
    rows_per_model = 1       # rows of the returned model array per instance (3 for 3x3 / 3x4 matrices)
    cols = 3

    # The small dense solve of the refits: "lapack" (default) = numpy.linalg.eigh on the host; "jacobi" (round 6, keyword
    # refit_solver= of the drop-in calls) = the context's batched cyclic-Jacobi solver (pgx_eigh_smallest_batch: on the device,
    # bitwise the CPU restatement the tests hold; agrees with LAPACK to ~1e-13 on the eigenvector) - the eigen-solves behind the
    # C ABI, the first step towards one C call per local-optimisation round.
    refit_solver = "lapack"
    _eig_ctx = None            # the context of the refit being run (set by nonminimal / nonminimal_labels / nonminimal_batch)

    def _smallest(self, A):
        """eigenvectors of the smallest eigenvalue of the symmetric matrices A [B, q, q] -> [B, q]"""
        ctx = self._eig_ctx
        if self.refit_solver == "jacobi" and ctx is not None and hasattr(ctx, "eigh_smallest_batch"):
            return ctx.eigh_smallest_batch(A)[0]
        return np.linalg.eigh(A)[1][:, :, 0]

    def minimal(self, pts, samples):
        """samples [S, m] -> (models [H, P], sample_of_model [H]); several or zero solutions per sample allowed."""
        raise NotImplementedError

    # -- model validity (gcransac Estimator::isValidModel, both overloads) [U-14] -------------------------------------------
    validity = "off"

    def valid_samples(self, pts, samples, models, src):
        """Quick test of hypotheses against their own minimal samples, before they are scored (isValidModel(model, data,
        sample, threshold)).  models [H, P], src [H] = row of `samples` each came from -> bool [H].  Default: all valid."""
        return np.ones(len(models), dtype=bool)

    def valid_best(self, ctx, pts, model, sample, threshold, rescore):
        """Test of a model that is about to become the so-far-best (isValidModel(model, data, inliers, sample, threshold,
        model_updated)).  Returns (valid, model): the model may come back replaced.  `rescore(models) -> scores` evaluates
        candidates with the proposal's scoring function.  Default: valid, unchanged."""
        return True, model

    def _fit(self, init):
        """The refit as a coroutine: yields Gram requests (kind, params, use_weights, wpow), receives (G, count, bad)
        and returns the list of models.  Written once, driven either by one pgx_gram call per request (`nonminimal`) or
        in lockstep over a batch of selections with one pgx_gram_batch launch per step (`nonminimal_batch`)."""
        raise NotImplementedError
        yield  # pragma: no cover

    def nonminimal(self, ctx, sel, weights=None, init=None):
        """Least-squares fit to the selected resident points -> list of models (the reference accepts the refit only
        if exactly 1).  sel = ("index", indices) or ("label", k); the data pass runs on the device (ctx.gram =
        pgx_gram: weighted Gram matrix of the design rows), the small dense solve here."""
        self._eig_ctx = ctx
        gen = self._fit(init)
        try:
            req = next(gen)
            while True:
                kind, params, use_w, wpow = req
                req = gen.send(ctx.gram(kind, sel, params=params, weights=weights if use_w else None, wpow=wpow))
        except StopIteration as done:
            return done.value

    def nonminimal_labels(self, ctx, K, weights=None, inits=None, skip=()):
        """`nonminimal(ctx, ("label", k), weights, init=inits[k])` for k = 0..K-1 with the coroutines advancing in lockstep:
        every step is ONE pgx_gram_labels launch for all labels (PEARL refits every instance per iteration; one host round
        trip per instance and step made that loop latency-bound).  Labels in `skip` are not fitted.  The Gram matrices
        are bit-identical to the single-label calls, so the results are those of K separate `nonminimal` calls."""
        self._eig_ctx = ctx
        if hasattr(self, "_fit_many"):          # refit vectorised over the instances (same Gram launches, stacked small solves)
            live = [k for k in range(K) if k not in skip]
            out = [[] for _ in range(K)]
            if not live:
                return out

            def gram(kind, prm, use_w, wpow, rows):
                sel = [live[r] for r in rows]
                full = None
                if prm is not None:
                    full = np.zeros((K, prm.shape[1]))
                    full[sel] = prm
                G, cnt, bad = ctx.gram_labels(kind, K, params=full, weights=weights if use_w else None, wpow=wpow)
                return G[sel], cnt[sel], bad[sel]
            res = self._fit_many(gram, len(live), [None if inits is None else inits[k] for k in live])
            for r, k in enumerate(live):
                out[k] = res[r]
            return out
        gens = {k: self._fit(None if inits is None else inits[k]) for k in range(K) if k not in skip}
        results, pending = {k: [] for k in range(K)}, {}
        for k, g in gens.items():
            try:
                pending[k] = next(g)
            except StopIteration as done:
                results[k] = done.value
        while pending:
            groups = {}
            for k, (kind, params, use_w, wpow) in pending.items():
                groups.setdefault((kind, bool(use_w), wpow, None if params is None else len(np.ravel(params))), []).append(k)
            for (kind, use_w, wpow, plen), ks in groups.items():
                prm = None
                if plen is not None:
                    prm = np.zeros((K, plen))
                    for k in ks:
                        prm[k] = np.asarray(pending[k][1], dtype=np.float64).reshape(-1)
                G, cnt, bad = ctx.gram_labels(kind, K, params=prm, weights=weights if use_w else None, wpow=wpow)
                for k in ks:
                    try:
                        pending[k] = gens[k].send((G[k], int(cnt[k]), int(bad[k])))
                    except StopIteration as done:
                        results[k] = done.value
                        del pending[k]
        return [results[k] for k in range(K)]

    def nonminimal_batch(self, ctx, index, weights=None, init=None):
        """`nonminimal` for B index selections of equal size (index [B, m]) sharing `init`: the B coroutines advance in
        lockstep, each step is ONE pgx_gram_batch launch.  Returns a list of B model lists."""
        index = np.asarray(index)
        B, m = index.shape
        self._eig_ctx = ctx
        if hasattr(self, "_fit_many"):           # refit vectorised over the batch (same Gram launches, stacked small solves)
            def gram(kind, prm, use_w, wpow, rows):
                G, bad = ctx.gram_batch(kind, index[rows], params=prm, weights=weights if use_w else None, wpow=wpow)
                return G, np.full(len(rows), m, dtype=np.int64), bad
            return self._fit_many(gram, B, [init] * B)
        gens = [self._fit(init) for _ in range(B)]
        results, pending = [None] * B, {}
        for b, g in enumerate(gens):
            try:
                pending[b] = next(g)
            except StopIteration as done:
                results[b] = done.value
        while pending:
            groups = {}
            for b, (kind, params, use_w, wpow) in pending.items():
                groups.setdefault((kind, bool(use_w), wpow, params is None), []).append(b)
            for (kind, use_w, wpow, no_params), bs in groups.items():
                prm = None if no_params else np.array([np.asarray(pending[b][1], dtype=np.float64).reshape(-1) for b in bs])
                G, bad = ctx.gram_batch(kind, index[bs], params=prm, weights=weights if use_w else None, wpow=wpow)
                for k, b in enumerate(bs):
                    try:
                        pending[b] = gens[b].send((G[k], m, int(bad[k])))
                    except StopIteration as done:
                        results[b] = done.value
                        del pending[b]
        return results

    def descriptor(self, model):
        return np.asarray(model, dtype=np.float64)

    def output(self, model):
        return np.asarray(model, dtype=np.float64).reshape(self.rows_per_model, self.cols)


# ---------------------------------------------------------------------------------------------------------------------
# 2D line  (Default2DLineEstimator, progressivex_python.cpp:489)  [U-4]
# ---------------------------------------------------------------------------------------------------------------------
class LineEstimator(Estimator):
    model_type = _lib.LINE2D
    sample_size = 2
    nonminimal_sample_size = 2
    device_minimal = True      # pgx_solve_minimal generates the hypotheses on the GPU (same formulas as minimal())

    def minimal(self, pts, samples):
        a, b = pts[samples[:, 0]], pts[samples[:, 1]]
        d = b - a
        ln = np.linalg.norm(d, axis=1)
        ok = ln > 0
        nrm = np.column_stack([-d[:, 1], d[:, 0]]) / np.where(ok, ln, 1.0)[:, None]
        c = -(nrm * a).sum(axis=1)
        models = np.column_stack([nrm, c])
        return models[ok], np.nonzero(ok)[0]

    def _fit(self, init):
        G, cnt, _ = yield (_lib.GRAM_AFFINE, None, True, 1)                    # sum w [1,x,y][1,x,y]^T
        W = G[0, 0]
        if cnt < 2 or not W > 0:
            return []
        mean = G[0, 1:] / W
        evals, evecs = np.linalg.eigh(G[1:, 1:] - W * np.outer(mean, mean))     # weighted scatter about the mean
        nrm = evecs[:, 0]
        return [np.array([nrm[0], nrm[1], -nrm @ mean])]

    def _fit_many(self, gram, B, inits):
        """`_fit` for B items at once: one Gram launch, stacked 2x2 eigh (LAPACK per matrix: bitwise the single-call models; the
        coroutine per item was half of a findLines call at C1 - 5 800 fits of ~7 us of Python each)."""
        out = [[] for _ in range(B)]
        G, cnt, _ = gram(_lib.GRAM_AFFINE, None, True, 1, np.arange(B))
        W = G[:, 0, 0]
        idx = np.nonzero((np.asarray(cnt) >= 2) & (W > 0))[0]
        if idx.size == 0:
            return out
        Wk = W[idx]
        mean = G[idx, 0, 1:] / Wk[:, None]
        scatter = G[idx][:, 1:, 1:] - Wk[:, None, None] * (mean[:, :, None] * mean[:, None, :])
        ok = np.isfinite(scatter).all(axis=(1, 2))
        evecs = np.zeros_like(scatter)
        if ok.any():
            evecs[ok] = np.linalg.eigh(scatter[ok])[1]
        for k, b in enumerate(idx):
            if not ok[k]:                         # (eigh raises on non-finite input: the single-item path would too)
                np.linalg.eigh(scatter[k])
            nrm = evecs[k][:, 0]
            out[b] = [np.array([nrm[0], nrm[1], -nrm @ mean[k]])]
        return out


# ---------------------------------------------------------------------------------------------------------------------
# vanishing point: solver_vanishing_point_two_lines.h (in-tree, exact restatement)
# ---------------------------------------------------------------------------------------------------------------------
class VanishingPointEstimator(Estimator):
    model_type = _lib.VANISHING_POINT
    sample_size = 2            # solver_vanishing_point_two_lines.h:70-73
    nonminimal_sample_size = 2  # the same solver class is used for both (progressivex_python.cpp:343-346)
    device_minimal = True      # pgx_solve_minimal generates the hypotheses on the GPU (same formulas as minimal())

    @staticmethod
    def _cross(a1, b1, c1, a2, b2, c2):  # :106-121
        return b1 * c2 - c1 * b2, -(a1 * c2 - c1 * a2), a1 * b2 - b1 * a2

    def minimal(self, pts, samples):
        s0, s1 = pts[samples[:, 0]], pts[samples[:, 1]]
        one = np.ones(len(samples))
        l0 = self._cross(s0[:, 0], s0[:, 1], one, s0[:, 2], s0[:, 3], one)      # :174-176
        l1 = self._cross(s1[:, 0], s1[:, 1], one, s1[:, 2], s1[:, 3], one)      # :177-179
        v = np.column_stack(self._cross(l0[0], l0[1], l0[2], l1[0], l1[1], l1[2]))  # :180-182
        ln = np.sqrt((v * v).sum(axis=1))                                        # :123-131 vec_norm
        ok = ln > 0
        v = v / np.where(ok, ln, 1.0)[:, None]
        return v[ok], np.nonzero(ok)[0]

    def _fit(self, init):
        # rows A = [y0*mz-my, mx-x0*mz, x0*my-y0*mx] * w (:212-218) are generated and accumulated on the device
        AtA, cnt, _ = yield (_lib.GRAM_VP, None, True, 2)
        if cnt < 2:
            return []
        if self.refit_solver == "jacobi":
            v = self._smallest(AtA[None])[0]
        else:
            evals, evecs = np.linalg.eigh(AtA)                                    # :227 SelfAdjointEigenSolver
            v = evecs[:, int(np.argmin(evals))]                                   # :230-233
        n = np.linalg.norm(v)
        return [v / n] if n > 0 else []

    def _fit_many(self, gram, B, inits):
        """`_fit` for B items at once (gram = the Gram provider of nonminimal_batch / nonminimal_labels): one launch, stacked
        3x3 eigh (LAPACK per matrix: bitwise the single-call models)."""
        out = [[] for _ in range(B)]
        AtA, cnt, _ = gram(_lib.GRAM_VP, None, True, 2, np.arange(B))
        idx = np.nonzero((cnt >= 2) & np.isfinite(AtA).all(axis=(1, 2)))[0]
        if idx.size == 0:
            return out
        if self.refit_solver == "jacobi":
            vs = self._smallest(AtA[idx])
        else:
            evals, evecs = np.linalg.eigh(AtA[idx])
            pick = np.argmin(evals, axis=1)
            vs = np.array([evecs[k][:, int(pick[k])] for k in range(len(idx))])
        for k, b in enumerate(idx):
            v = vs[k]
            n = np.linalg.norm(v)
            if n > 0:
                out[b] = [v / n]
        return out


# ---------------------------------------------------------------------------------------------------------------------
# homography (DefaultHomographyEstimator, progressivex_python.cpp:252)  [UPSTREAM-MEMORY]: 4-point, h33 = 1
# ---------------------------------------------------------------------------------------------------------------------
def _dlt_rows(x1, y1, x2, y2):
    z = np.zeros_like(x1)
    o = np.ones_like(x1)
    r1 = np.stack([-x1, -y1, -o, z, z, z, x2 * x1, x2 * y1, x2], axis=-1)
    r2 = np.stack([z, z, z, -x1, -y1, -o, y2 * x1, y2 * y1, y2], axis=-1)
    return r1, r2


def _hartley_from_moments(G, cnt):
    """Normalising similarities of both images from one pass of first/second moments (the affine Gram matrix G;
    isotropic scaling to an RMS distance of sqrt(2) from the centroid).  Returns (T1, T2, params for the DLT /
    epipolar rows, count)."""
    if cnt < 1:
        return None, None, None, 0
    c = G[0, 1:] / cnt
    var = np.maximum(np.diag(G)[1:] / cnt - c * c, 0.0)
    Ts, prm = [], []
    for k in (0, 2):
        rms = np.sqrt(var[k] + var[k + 1])
        sc = np.sqrt(2.0) / rms if rms > 0 else 1.0
        Ts.append(np.array([[sc, 0, -sc * c[k]], [0, sc, -sc * c[k + 1]], [0, 0, 1.0]]))
        prm += [sc, c[k], c[k + 1]]
    return Ts[0], Ts[1], np.array(prm), cnt


def _hartley_batch(G, cnt):
    """`_hartley_from_moments` for a stack of Gram matrices G [B, 5, 5] with counts cnt [B] (all >= 1): (T1 [B,3,3],
    T2 [B,3,3], params [B,6]) - elementwise the same arithmetic, so every item gets bitwise the single-call result."""
    B = G.shape[0]
    cnt = np.asarray(cnt, dtype=np.float64).reshape(B, 1)
    c = G[:, 0, 1:] / cnt
    var = np.maximum(np.diagonal(G, axis1=1, axis2=2)[:, 1:] / cnt - c * c, 0.0)
    Ts, prm = [], []
    for k in (0, 2):
        rms = np.sqrt(var[:, k] + var[:, k + 1])
        sc = np.where(rms > 0, np.sqrt(2.0) / np.where(rms > 0, rms, 1.0), 1.0)
        T = np.zeros((B, 3, 3))
        T[:, 0, 0] = sc
        T[:, 0, 2] = -sc * c[:, k]
        T[:, 1, 1] = sc
        T[:, 1, 2] = -sc * c[:, k + 1]
        T[:, 2, 2] = 1.0
        Ts.append(T)
        prm += [sc, c[:, k], c[:, k + 1]]
    return Ts[0], Ts[1], np.stack(prm, axis=1)


def _dlt_homography(p):
    """normalised DLT homography of correspondences p [k, 4] (host, k small: DEGENSAC's plane refit) or None"""
    x1, x2 = p[:, 0:2], p[:, 2:4]
    def norm(x):
        c = x.mean(0)
        d = np.sqrt(((x - c) ** 2).sum(1)).mean()
        if not (d > 0):
            return None, None
        sc = np.sqrt(2.0) / d
        T = np.array([[sc, 0, -sc * c[0]], [0, sc, -sc * c[1]], [0, 0, 1.0]])
        return (x - c) * sc, T
    a, T1 = norm(x1)
    b, T2 = norm(x2)
    if a is None or b is None:
        return None
    z = np.zeros(len(p))
    o = np.ones(len(p))
    r1 = np.stack([a[:, 0], a[:, 1], o, z, z, z, -b[:, 0] * a[:, 0], -b[:, 0] * a[:, 1], -b[:, 0]], axis=1)
    r2 = np.stack([z, z, z, a[:, 0], a[:, 1], o, -b[:, 1] * a[:, 0], -b[:, 1] * a[:, 1], -b[:, 1]], axis=1)
    A = np.vstack([r1, r2])
    try:
        h = np.linalg.svd(A)[2][-1].reshape(3, 3)
        H = np.linalg.inv(T2) @ h @ T1
    except np.linalg.LinAlgError:
        return None
    return H if np.isfinite(H).all() else None


def _smallest_eigenvector(G):
    evals, evecs = np.linalg.eigh(G)
    return evecs[:, 0]


class HomographyEstimator(Estimator):
    model_type = _lib.HOMOGRAPHY
    sample_size = 4
    nonminimal_sample_size = 4
    rows_per_model = 3
    device_minimal = True      # pgx_solve_minimal: 8x8 Gaussian elimination per sample on the GPU

    def minimal(self, pts, samples):
        scale = max(1.0, float(np.abs(pts).max()))     # isotropic pre-scaling: coordinates O(1) for the 8x8 solve
        p = pts[samples] / scale                       # [S,4,4]
        r1, r2 = _dlt_rows(p[..., 0], p[..., 1], p[..., 2], p[..., 3])
        A = np.concatenate([r1, r2], axis=1)           # [S,8,9]
        lhs, rhs = A[:, :, :8], -A[:, :, 8]
        ok = np.abs(np.linalg.det(lhs)) > 1e-14
        h = np.zeros((len(samples), 9))
        if ok.any():
            h[ok, :8] = np.linalg.solve(lhs[ok], rhs[ok][..., None])[..., 0]
            h[ok, 8] = 1.0
            # undo the scaling: H = S^-1 Hn S with S = diag(1/scale, 1/scale, 1)
            h[ok, 2] *= scale
            h[ok, 5] *= scale
            h[ok, 6] /= scale
            h[ok, 7] /= scale
        ok &= np.isfinite(h).all(axis=1)
        return h[ok], np.nonzero(ok)[0]

    def _fit(self, init):
        G, cnt, _ = yield (_lib.GRAM_AFFINE, None, False, 2)
        T1, T2, prm, cnt = _hartley_from_moments(G, cnt)
        if cnt < 4:
            return []
        AtA, _, _ = yield (_lib.GRAM_DLT_H, prm, True, 2)                                  # normalised DLT rows
        Hn = (self._smallest(AtA[None])[0] if self.refit_solver == "jacobi" else _smallest_eigenvector(AtA)).reshape(3, 3)
        H = np.linalg.inv(T2) @ Hn @ T1
        if not np.isfinite(H).all() or abs(H[2, 2]) < 1e-300:
            return []
        return [(H / H[2, 2]).reshape(-1)]


    def _fit_many(self, gram, B, inits):
        """`_fit` for B items at once: two Gram launches, stacked eigh / inv (numpy runs LAPACK per matrix, so each item's
        model is bitwise the single-call one)."""
        out = [[] for _ in range(B)]
        G, cnt, _ = gram(_lib.GRAM_AFFINE, None, False, 2, np.arange(B))
        rows = np.nonzero((cnt >= 4) & np.isfinite(G).all(axis=(1, 2)))[0]
        if rows.size == 0:
            return out
        T1, T2, prm = _hartley_batch(G[rows], cnt[rows])
        AtA, _, _ = gram(_lib.GRAM_DLT_H, prm, True, 2, rows)
        fin = np.isfinite(AtA).all(axis=(1, 2)) & np.isfinite(T2).all(axis=(1, 2))
        if not fin.any():
            return out
        sel = np.nonzero(fin)[0]
        Hn = self._smallest(AtA[sel]).reshape(-1, 3, 3)
        H = np.linalg.inv(T2[sel]) @ Hn @ T1[sel]
        for k, r in enumerate(sel):
            Hb = H[k]
            if np.isfinite(Hb).all() and abs(Hb[2, 2]) >= 1e-300:
                out[rows[r]] = [(Hb / Hb[2, 2]).reshape(-1)]
        return out


class SymmetricHomographyEstimator(HomographyEstimator):
    """Same solvers; the model carried to the kernels is [H | H^-1] (symmetric transfer error switch)."""
    model_type = _lib.HOMOGRAPHY_SYM
    device_minimal = False     # 18-parameter models: generated on the host

    @staticmethod
    def _augment(models):
        out = []
        keep = []
        for k, h in enumerate(models):
            H = h.reshape(3, 3)
            if abs(np.linalg.det(H)) < 1e-300:
                continue
            out.append(np.concatenate([h, np.linalg.inv(H).reshape(-1)]))
            keep.append(k)
        return (np.array(out).reshape(-1, 18), np.array(keep, dtype=np.int64))

    def minimal(self, pts, samples):
        models, src = super().minimal(pts, samples)
        aug, keep = self._augment(models)
        return aug, src[keep]

    def _fit(self, init):
        res = yield from super()._fit(init)
        return [m for m in self._augment(np.array(res).reshape(-1, 9))[0]]

    def _fit_many(self, gram, B, inits):
        return [[m for m in self._augment(np.array(res).reshape(-1, 9))[0]] for res in super()._fit_many(gram, B, inits)]

    def output(self, model):
        return np.asarray(model[:9], dtype=np.float64).reshape(3, 3)


# ---------------------------------------------------------------------------------------------------------------------
# fundamental matrix (DefaultFundamentalMatrixEstimator, progressivex_python.cpp:616)  [UPSTREAM-MEMORY]
# 7-point minimal (up to 3 solutions), normalised 8-point non-minimal
# ---------------------------------------------------------------------------------------------------------------------
class FundamentalEstimator(Estimator):
    model_type = _lib.FUNDAMENTAL
    sample_size = 7
    nonminimal_sample_size = 8
    rows_per_model = 3
    device_minimal = True      # pgx_solve_minimal: Gauss-Jordan null space + cubic by bisection, three slots per sample
    device_slots = 3

    @staticmethod
    def _rows(p):
        x1, y1, x2, y2 = p[..., 0], p[..., 1], p[..., 2], p[..., 3]
        o = np.ones_like(x1)
        return np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, o], axis=-1)

    def minimal(self, pts, samples):
        p = pts[samples]
        scale = max(1.0, float(np.abs(pts).max()))
        A = self._rows(p / scale)                      # [S,7,9], isotropic pre-scaling for conditioning
        _, _, vt = np.linalg.svd(A, full_matrices=True)
        F1, F2 = vt[:, 7].reshape(-1, 3, 3), vt[:, 8].reshape(-1, 3, 3)
        # det(l F1 + (1-l) F2) is a cubic in l: interpolate it at 4 nodes
        ls = np.array([0.0, 1.0, -1.0, 2.0])
        vals = np.stack([np.linalg.det(l * F1 + (1 - l) * F2) for l in ls], axis=1)
        V = np.vander(ls, 4)
        coef = np.linalg.solve(V, vals.T).T            # [S,4] highest power first
        D = np.diag([1 / scale, 1 / scale, 1.0])
        # roots of all cubics at once: eigenvalues of the stacked companion matrices (np.roots does the same one by
        # one); samples with a vanishing leading coefficient (a quadratic) take the slow path
        finite = np.isfinite(coef).all(axis=1)
        lead = np.abs(coef[:, 0]) > 1e-14 * np.abs(coef).max(axis=1)
        reg = np.nonzero(finite & lead)[0]
        sidx, roots = [], []
        if len(reg):
            comp = np.zeros((len(reg), 3, 3))
            comp[:, 0, :] = -coef[reg, 1:] / coef[reg, :1]
            comp[:, 1, 0] = 1.0
            comp[:, 2, 1] = 1.0
            ev = np.linalg.eigvals(comp)                                   # [R,3]
            real = np.abs(ev.imag) <= 1e-9 * np.maximum(1.0, np.abs(ev.real))
            rr, cc = np.nonzero(real)
            sidx.append(reg[rr])
            roots.append(ev.real[rr, cc])
        for s in np.nonzero(finite & ~lead)[0]:
            for r in np.roots(coef[s, 1:]):
                if abs(r.imag) <= 1e-9 * max(1.0, abs(r.real)):
                    sidx.append(np.array([s]))
                    roots.append(np.array([r.real]))
        if not sidx:
            return np.zeros((0, 9)), np.zeros(0, dtype=np.int64)
        sidx, roots = np.concatenate(sidx), np.concatenate(roots)
        o = np.argsort(sidx, kind="stable")                                # by sample; roots keep LAPACK's order (as np.roots)
        sidx, roots = sidx[o], roots[o]
        F = D @ (roots[:, None, None] * F1[sidx] + (1.0 - roots)[:, None, None] * F2[sidx]) @ D
        nrm = np.sqrt((F * F).sum(axis=(1, 2)))
        ok = (nrm > 0) & np.isfinite(nrm)
        models = (F[ok] / nrm[ok][:, None, None]).reshape(-1, 9)
        return models, sidx[ok].astype(np.int64)

        return np.array(models).reshape(-1, 9), np.array(src, dtype=np.int64)

    # -- validity [U-14] ------------------------------------------------------------------------------------------------------
    # DefaultFundamentalMatrixEstimator (progressivex_python.cpp:616) = gcransac's FundamentalMatrixEstimator, whose source is
    # absent from the snapshot.  Restated from the literature the upstream comments cite [UPSTREAM-MEMORY]:
    #   "oriented"  Chum, Werner, Matas, "Epipolar geometry estimation via RANSAC benefits from the oriented epipolar
    #               constraint" (ICPR 2004): e' x x' and F x must point the same way for every correspondence of the
    #               minimal sample - a hypothesis whose sample disagrees is dropped before it is scored;
    #   "symmetric" every so-far-best F must keep at least max(7, ratio * |inliers|) of its Sampson inliers (r^2 < T^2, the
    #               scorer's threshold) when they are tested with the SYMMETRIC EPIPOLAR distance d(x', F x)^2 + d(x, F^T x')^2
    #               - the Sampson distance stays small next to a degenerate (near rank-one) F's pseudo-epipoles, where one of
    #               the two point-line distances explodes; ratio = 0.5 upstream.  The symmetric distance is 4 x the Sampson
    #               distance when both gradients are equal, so it is compared with 4 T^2: against threshold^2 (what the
    #               recollection of upstream says) the CLEAN ground-truth models of the bundled cubetoy scene keep only
    #               48-60 % of their inliers and are rejected (docs/experiments-cubetoy.md);
    #   "degensac"  Chum, Werner, Matas, "Two-view geometry estimation unaffected by a dominant plane" (CVPR 2005): if five of
    #               the seven sample points agree with a homography H compatible with F, F is H-degenerate: H is re-estimated
    #               on its inliers, epipoles are drawn from pairs of off-plane correspondences (plane and parallax:
    #               e' = (x1' x H x1) x (x2' x H x2), F = [e']x H) and the best-scoring F replaces the model.
    # validity = "off" | "oriented" | "symmetric" (oriented + symmetric) | "full" (all three).  Default "off": nothing of these
    # stages can be checked against the snapshot, and on the three bundled AdelaideRMF scenes plus C3 no setting measures better
    # than none (docs/experiments-cubetoy.md section 1, round-4 rows) - a recollection does not get to change default results.
    validity = "off"
    minimum_inlier_ratio_in_validity_check = 0.5
    homography_threshold = 2.0

    @staticmethod
    def _hom(p):
        o = np.ones(p.shape[:-1] + (1,))
        return np.concatenate([p[..., 0:2], o], axis=-1), np.concatenate([p[..., 2:4], o], axis=-1)

    def valid_samples(self, pts, samples, models, src):
        ok = np.isfinite(models).all(axis=1)
        if self.validity == "off" or not ok.any():
            return ok
        idx = np.nonzero(ok)[0]
        F = models[idx].reshape(-1, 3, 3)
        x1, x2 = self._hom(pts[np.asarray(samples)[src[idx]]])                 # [H, 7, 3] each
        e2 = np.linalg.svd(F)[0][:, :, 2]                                       # left null vector: e2^T F = 0
        l = np.einsum("hij,hkj->hki", F, x1)                                    # F x1
        m = np.cross(e2[:, None, :], x2)                                        # e2 x x2
        sg = np.sign((l * m).sum(-1))
        good = (sg > 0).all(axis=1) | (sg < 0).all(axis=1)
        ok[idx[~good]] = False
        return ok

    @staticmethod
    def _sampson_and_symmetric(F, pts):
        x1, x2 = FundamentalEstimator._hom(pts)
        l2 = x1 @ F.T                                                           # F x1: lines in image 2
        l1 = x2 @ F                                                             # F^T x2: lines in image 1
        r = (x2 * l2).sum(-1)
        a = l2[:, 0] ** 2 + l2[:, 1] ** 2
        b = l1[:, 0] ** 2 + l1[:, 1] ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            return r * r / (a + b), r * r * (1.0 / a + 1.0 / b)

    def _h_from_f(self, F, e2, x1, x2):
        """H compatible with F through three correspondences (Hartley & Zisserman, result 13.6): H = A - e2 (M^-1 b)^T,
        A = [e2]x F, M = rows x1_i, b_i = (x2_i x A x1_i) . (x2_i x e2) / |x2_i x e2|^2."""
        ex = np.array([[0.0, -e2[2], e2[1]], [e2[2], 0.0, -e2[0]], [-e2[1], e2[0], 0.0]])
        A = ex @ F
        c1 = np.cross(x2, (A @ x1.T).T)
        c2 = np.cross(x2, e2[None, :])
        den = (c2 * c2).sum(-1)
        if not np.all(den > 0):
            return None
        b = (c1 * c2).sum(-1) / den
        try:
            return A - np.outer(e2, np.linalg.solve(x1, b))
        except np.linalg.LinAlgError:
            return None

    @staticmethod
    def _transfer2(H, x1, x2):
        y = x1 @ H.T
        with np.errstate(divide="ignore", invalid="ignore"):
            d = (y[:, 0] / y[:, 2] - x2[:, 0]) ** 2 + (y[:, 1] / y[:, 2] - x2[:, 1]) ** 2
        return np.where(np.isfinite(d), d, np.inf)

    def valid_best(self, ctx, pts, model, sample, threshold, rescore):
        if self.validity in ("off", "oriented"):
            return True, model
        F = np.asarray(model, dtype=np.float64).reshape(3, 3)
        T2 = 2.25 * threshold * threshold
        inliers, supported = ctx.epipolar_support(F, T2, 4.0 * T2)           # one pass over all points on the device
        need = max(self.sample_size, int(inliers * self.minimum_inlier_ratio_in_validity_check))
        if supported < need:
            return False, model
        if self.validity != "full" or sample is None:
            return True, model
        # ---- DEGENSAC: is the sample H-degenerate?
        x1, x2 = self._hom(pts[np.asarray(sample)])
        e2 = np.linalg.svd(F)[0][:, 2]
        h2 = self.homography_threshold ** 2
        H = None
        for tri in ((0, 1, 2), (3, 4, 5), (0, 1, 6), (3, 4, 6), (2, 5, 6)):
            cand = self._h_from_f(F, e2, x1[list(tri)], x2[list(tri)])
            if cand is not None and np.isfinite(cand).all() and int((self._transfer2(cand, x1, x2) < h2).sum()) >= 5:
                H = cand
                break
        if H is None:
            return True, model
        # the plane: inliers of H among all correspondences, H refitted on them (normalised DLT)
        a1, a2 = self._hom(pts)
        on = self._transfer2(H, a1, a2) < h2
        if int(on.sum()) >= 4:
            Hr = _dlt_homography(pts[on])
            if Hr is not None and int((self._transfer2(Hr, a1, a2) < h2).sum()) >= int(on.sum()):
                H = Hr
                on = self._transfer2(H, a1, a2) < h2
        off = np.nonzero(~on)[0]
        if len(off) < 2:
            return True, model
        # plane and parallax: epipoles from pairs of off-plane correspondences
        rng = np.random.default_rng(int(on.sum()) * 7919 + len(off))            # deterministic in the data (no stream consumed)
        trials = min(100, len(off) * (len(off) - 1) // 2)
        pa = rng.integers(0, len(off), trials)
        pb = (pa + 1 + rng.integers(0, len(off) - 1, trials)) % len(off)
        la = np.cross(a2[off[pa]], (a1[off[pa]] @ H.T))
        lb = np.cross(a2[off[pb]], (a1[off[pb]] @ H.T))
        e = np.cross(la, lb)
        nrm = np.linalg.norm(e, axis=1)
        e = e[nrm > 0] / nrm[nrm > 0][:, None]
        if len(e) == 0:
            return True, model
        ex = np.zeros((len(e), 3, 3))
        ex[:, 0, 1], ex[:, 0, 2], ex[:, 1, 0], ex[:, 1, 2], ex[:, 2, 0], ex[:, 2, 1] = -e[:, 2], e[:, 1], e[:, 2], -e[:, 0], -e[:, 1], e[:, 0]
        cands = ex @ H
        cn = np.linalg.norm(cands.reshape(len(e), -1), axis=1)
        cands = (cands[cn > 0] / cn[cn > 0][:, None, None]).reshape(-1, 9)
        if len(cands) == 0:
            return True, model
        sc = np.asarray(rescore(np.vstack([np.asarray(model, dtype=np.float64)[None, :], cands])), dtype=np.float64)
        k = int(np.argmax(sc[1:])) + 1
        if np.isfinite(sc[k]) and sc[k] > sc[0]:
            return True, cands[k - 1].copy()
        return True, model

    def _fit(self, init):
        G, cnt, _ = yield (_lib.GRAM_AFFINE, None, False, 2)
        T1, T2, prm, cnt = _hartley_from_moments(G, cnt)
        if cnt < 8:
            return []
        AtA, _, _ = yield (_lib.GRAM_EPI_F, prm, True, 2)                                  # normalised 8-point rows
        F = (self._smallest(AtA[None])[0] if self.refit_solver == "jacobi" else _smallest_eigenvector(AtA)).reshape(3, 3)
        u, s, v = np.linalg.svd(F)
        F = u @ np.diag([s[0], s[1], 0.0]) @ v
        F = T2.T @ F @ T1
        nrm = np.linalg.norm(F)
        if not np.isfinite(nrm) or nrm == 0:
            return []
        return [(F / nrm).reshape(-1)]

    def _fit_many(self, gram, B, inits):
        """`_fit` for B items at once (a local-optimisation round refits 50 samples, a PEARL iteration every instance: 5 200
        eigh + svd calls were a third of findTwoViewMotions' proposal time at C3): two Gram launches, stacked eigh / svd -
        LAPACK per matrix, so each item's model is bitwise the single-call one."""
        out = [[] for _ in range(B)]
        G, cnt, _ = gram(_lib.GRAM_AFFINE, None, False, 2, np.arange(B))
        rows = np.nonzero((cnt >= 8) & np.isfinite(G).all(axis=(1, 2)))[0]
        if rows.size == 0:
            return out
        T1, T2, prm = _hartley_batch(G[rows], cnt[rows])
        AtA, _, _ = gram(_lib.GRAM_EPI_F, prm, True, 2, rows)
        sel = np.nonzero(np.isfinite(AtA).all(axis=(1, 2)))[0]
        if sel.size == 0:
            return out
        F = self._smallest(AtA[sel]).reshape(-1, 3, 3)
        okf = np.isfinite(F).all(axis=(1, 2))
        sel, F = sel[okf], F[okf]
        if sel.size == 0:
            return out
        u, sv, v = np.linalg.svd(F)
        D = np.zeros_like(F)
        D[:, 0, 0], D[:, 1, 1] = sv[:, 0], sv[:, 1]
        F = u @ D @ v
        F = T2[sel].transpose(0, 2, 1) @ F @ T1[sel]
        for k, r in enumerate(sel):
            nrm = np.linalg.norm(F[k])
            if np.isfinite(nrm) and nrm != 0:
                out[rows[r]] = [(F[k] / nrm).reshape(-1)]
        return out


# ---------------------------------------------------------------------------------------------------------------------
# PnP (DefaultPnPEstimator, progressivex_python.cpp:119)  [UPSTREAM-MEMORY]: P3P minimal (Grunert's quartic + Horn
# alignment, up to 4 solutions), Gauss-Newton reprojection refinement as the non-minimal fit
# ---------------------------------------------------------------------------------------------------------------------
def _kabsch(A, B):
    """Batched rigid alignment B ~ R A + t for [S,3,3] point triples (rows = points)."""
    ca, cb = A.mean(axis=1, keepdims=True), B.mean(axis=1, keepdims=True)
    H = np.einsum("sij,sik->sjk", A - ca, B - cb)
    U, _, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(np.einsum("sij,sjk->sik", Vt.transpose(0, 2, 1), U.transpose(0, 2, 1))))
    Dm = np.zeros_like(H)
    Dm[:, 0, 0] = 1
    Dm[:, 1, 1] = 1
    Dm[:, 2, 2] = d
    R = Vt.transpose(0, 2, 1) @ Dm @ U.transpose(0, 2, 1)
    t = cb[:, 0] - np.einsum("sij,sj->si", R, ca[:, 0])
    return R, t


class PnPEstimator(Estimator):
    model_type = _lib.PNP
    sample_size = 3
    nonminimal_sample_size = 6
    rows_per_model = 3
    cols = 4
    device_minimal = True      # pgx_solve_minimal: Grunert's quartic by bisection, pose from the triangle frames; 4 slots
    device_slots = 4

    def minimal(self, pts, samples):
        p = pts[samples]                                     # [S,3,5]
        f = np.concatenate([p[..., :2], np.ones(p.shape[:2] + (1,))], axis=-1)
        f = f / np.linalg.norm(f, axis=-1, keepdims=True)    # unit bearings
        X = p[..., 2:]
        a = np.linalg.norm(X[:, 1] - X[:, 2], axis=1)
        b = np.linalg.norm(X[:, 0] - X[:, 2], axis=1)
        c = np.linalg.norm(X[:, 0] - X[:, 1], axis=1)
        ca = (f[:, 1] * f[:, 2]).sum(axis=1)
        cb = (f[:, 0] * f[:, 2]).sum(axis=1)
        cg = (f[:, 0] * f[:, 1]).sum(axis=1)
        ok = (a > 0) & (b > 0) & (c > 0)
        with np.errstate(all="ignore"):
            a2, b2, c2 = a * a, b * b, c * c
            q = (a2 - c2) / b2
            r = (a2 + c2) / b2
            # Grunert / Fischler-Bolles quartic in v = s3 / s1
            A4 = (q - 1) ** 2 - 4 * c2 / b2 * ca ** 2
            A3 = 4 * (q * (1 - q) * cb - (1 - r) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb)
            A2 = 2 * (q ** 2 - 1 + 2 * q ** 2 * cb ** 2 + 2 * (b2 - c2) / b2 * ca ** 2
                      - 4 * r * ca * cb * cg + 2 * (b2 - a2) / b2 * cg ** 2)
            A1 = 4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - r) * ca * cg)
            A0 = (1 + q) ** 2 - 4 * a2 / b2 * cg ** 2
        coef = np.stack([A4, A3, A2, A1, A0], axis=1)
        ok &= np.isfinite(coef).all(axis=1) & (np.abs(A4) > 1e-14)
        S = len(samples)
        comp = np.zeros((S, 4, 4))
        comp[:, 1, 0] = comp[:, 2, 1] = comp[:, 3, 2] = 1.0
        safeA4 = np.where(ok, A4, 1.0)
        comp[:, 0, 3] = -A0 / safeA4
        comp[:, 1, 3] = -A1 / safeA4
        comp[:, 2, 3] = -A2 / safeA4
        comp[:, 3, 3] = -A3 / safeA4
        comp[~ok] = 0
        roots = np.linalg.eigvals(comp)                      # [S,4]
        models, src = [], []
        for k in range(4):
            v = roots[:, k]
            real = ok & (np.abs(v.imag) < 1e-8 * np.maximum(1.0, np.abs(v.real))) & (v.real > 0)
            if not real.any():
                continue
            sidx = np.nonzero(real)[0]
            vv = v.real[sidx]
            qq, cga, cba, caa = q[sidx], cg[sidx], cb[sidx], ca[sidx]
            a2s, b2s, c2s = a2[sidx], b2[sidx], c2[sidx]
            with np.errstate(all="ignore"):
                u = ((-1 + qq) * vv * vv - 2 * qq * cba * vv + 1 + qq) / (2 * (cga - vv * caa))
                s1 = np.sqrt(b2s / (1 + vv * vv - 2 * vv * cba))
            s2, s3 = u * s1, vv * s1
            good = np.isfinite(s1) & np.isfinite(s2) & (s1 > 0) & (s2 > 0) & (s3 > 0)
            # consistency with the two unused side constraints
            e1 = s1 ** 2 + s2 ** 2 - 2 * s1 * s2 * cga - c2s
            e2 = s2 ** 2 + s3 ** 2 - 2 * s2 * s3 * caa - a2s
            good &= (np.abs(e1) < 1e-6 * c2s) & (np.abs(e2) < 1e-6 * a2s)
            if not good.any():
                continue
            sidx = sidx[good]
            depth = np.stack([s1[good], s2[good], s3[good]], axis=1)
            Y = f[sidx] * depth[..., None]                    # camera-frame points
            R, t = _kabsch(X[sidx], Y)
            P = np.concatenate([R, t[..., None]], axis=2).reshape(-1, 12)
            fin = np.isfinite(P).all(axis=1)
            models.append(P[fin])
            src.append(sidx[fin])
        if not models:
            return np.zeros((0, 12)), np.zeros(0, dtype=np.int64)
        models, src = np.vstack(models), np.concatenate(src)
        o = np.argsort(src, kind="stable")
        return models[o], src[o]

    def _fit(self, init):
        """Gauss-Newton on the reprojection error from `init` (PEARL and the local optimisation always have one; the
        DLT start of an un-initialised fit is not needed on this path).  Per iteration the device accumulates the
        normal equations  sum (J,r)^T (J,r)  (rows in fit.hip GenPnpGn), the 6x6 solve and the pose update run here."""
        if init is None:
            return []
        P = np.asarray(init, dtype=np.float64).reshape(3, 4).copy()
        R, t = P[:, :3], P[:, 3]
        for _ in range(10):
            G, cnt, bad = yield (_lib.GRAM_PNP_GN, np.column_stack([R, t]).reshape(-1), True, 2)
            if cnt < 4 or bad > 0:
                return []
            try:
                delta = np.linalg.lstsq(G[:6, :6], -G[:6, 6], rcond=None)[0]
            except np.linalg.LinAlgError:
                return []
            om = delta[:3]
            ang = np.linalg.norm(om)
            if ang > 0:                                       # R <- exp([omega]_x) R ; t <- t + dt
                k = om / ang
                Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
                R = (np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)) @ R
            t = t + delta[3:]
            if np.linalg.norm(delta) < 1e-12:
                break
        out = np.column_stack([R, t])
        return [out.reshape(-1)] if np.isfinite(out).all() else []


    def nonminimal_batch(self, ctx, index, weights=None, init=None):
        """The local optimisation's B refits from one start: ONE pgx_pnp_refine_batch launch that runs all Gauss-Newton steps
        on the device when the context offers it (the GPU context); otherwise - the CPU oracle context of the tests - the
        lockstep iteration of `_fit_many`, whose iterates the kernel reproduces up to rounding."""
        refine = getattr(ctx, "pnp_refine_batch", None)
        if refine is None or init is None:
            return super().nonminimal_batch(ctx, index, weights, init)
        index = np.asarray(index)
        B = index.shape[0]
        start = np.tile(np.asarray(init, dtype=np.float64).reshape(1, 12), (B, 1))
        P, ok = refine(start, index, weights=weights, wpow=2, iterations=10)
        return [[P[b]] if ok[b] else [] for b in range(B)]

    def _fit_many(self, gram, B, inits, iterations=10):
        """`_fit` for B items at once: the same Gauss-Newton iteration with the 6x6 solves, the rotation updates and the
        convergence tests done on [B, ...] arrays - 50 refits x 10 iterations per graph-cut round were 13 us of lstsq + 15 us
        of small-array numpy each, a third of C4's proposal time.  The solve is the pseudo-inverse with lstsq's cut-off
        (singular values below eps * 6 * s_max dropped), so an item's iterates are those of `_fit` up to rounding."""
        out = [[] for _ in range(B)]
        have = np.array([ini is not None for ini in inits], dtype=bool)
        if not have.any():
            return out
        R = np.zeros((B, 3, 3))
        t = np.zeros((B, 3))
        for b in np.nonzero(have)[0]:
            P0 = np.asarray(inits[b], dtype=np.float64).reshape(3, 4)
            R[b], t[b] = P0[:, :3], P0[:, 3]
        running = have.copy()                   # still iterating
        failed = ~have
        eye = np.eye(3)
        for _ in range(iterations):
            act = np.nonzero(running)[0]
            if act.size == 0:
                break
            prm = np.concatenate([R[act], t[act][:, :, None]], axis=2).reshape(act.size, 12)
            G, cnt, bad = gram(_lib.GRAM_PNP_GN, prm, True, 2, act)
            A, b = G[:, :6, :6], -G[:, :6, 6]
            ok = (bad == 0) & (cnt >= 4) & np.isfinite(A).all(axis=(1, 2)) & np.isfinite(b).all(axis=1)
            failed[act[~ok]] = True
            running[act[~ok]] = False
            if not ok.any():
                break
            act, A, b = act[ok], A[ok], b[ok]
            delta = np.einsum("bij,bj->bi", np.linalg.pinv(A, rcond=6 * np.finfo(np.float64).eps), b)
            om = delta[:, :3]
            ang = np.linalg.norm(om, axis=1)
            k = om / np.where(ang > 0, ang, 1.0)[:, None]
            Kx = np.zeros((act.size, 3, 3))
            Kx[:, 0, 1], Kx[:, 0, 2] = -k[:, 2], k[:, 1]
            Kx[:, 1, 0], Kx[:, 1, 2] = k[:, 2], -k[:, 0]
            Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 1], k[:, 0]
            rot = eye + np.sin(ang)[:, None, None] * Kx + (1 - np.cos(ang))[:, None, None] * (Kx @ Kx)
            rot[ang == 0] = eye                   # R <- exp([omega]_x) R ; t <- t + dt
            R[act] = rot @ R[act]
            t[act] = t[act] + delta[:, 3:]
            running[act[np.linalg.norm(delta, axis=1) < 1e-12]] = False
        P = np.concatenate([R, t[:, :, None]], axis=2).reshape(B, 12)
        good = ~failed & np.isfinite(P).all(axis=1)
        return [[P[b]] if good[b] else [] for b in range(B)]


ESTIMATORS = {
    "line": LineEstimator,
    "vanishing_point": VanishingPointEstimator,
    "homography": HomographyEstimator,
    "homography_sym": SymmetricHomographyEstimator,
    "fundamental": FundamentalEstimator,
    "pnp": PnPEstimator,
}
