"""End-to-end differential soak (tests/test_gpu_fuzz.py runs a bounded slice): the five drop-in calls on random small
synthetic problems with random arguments (thresholds, spatial coherence, samplers, neighbourhoods, local optimisation, the
lambda = 0 switch, model caps) - once through libpgx.so on the GPU and once through the same host code driven by the CPU
oracle, same seed.  Labels must be identical, models equal to 1e-7; a call that differs is replayed with every context call recorded on both
sides and classified (see classify()).  usage: python tests/soak_api.py <seed> <trials>"""
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "progressive-x_amd"), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]
import numpy as np
import pyprogressivex as px
from oracle_ctx import OracleContext
from pyprogressivex import _api, datasets


class Recorder:
    """proxy of a context that logs (method, arguments, result) of every call"""
    def __init__(self, inner):
        self._i, self.log = inner, []

    def __getattr__(self, name):
        a = getattr(self._i, name)
        if not callable(a):
            return a

        def f(*args, **kw):
            r = a(*args, **kw)
            self.log.append((name, _flat(list(args)), _flat(kw), _flat(r)))
            return r
        return f


def _flat(r):
    if isinstance(r, dict):
        return {k: _flat(v) for k, v in r.items() if k != "path"}
    if isinstance(r, (tuple, list)):
        return [_flat(v) for v in r]
    if isinstance(r, np.ndarray):
        return r.copy()
    return r


def _differ(a, b, tol):
    """integers, shapes and structure exactly; floats to `tol` relative (0 = bitwise, NaN == NaN)"""
    if isinstance(a, dict):
        return set(a) != set(b) or any(_differ(a[k], b[k], tol) for k in a)
    if isinstance(a, list):
        return not isinstance(b, list) or len(a) != len(b) or any(_differ(x, y, tol) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        if not isinstance(b, np.ndarray) or a.shape != b.shape:
            return True
        if a.dtype.kind == "f":
            return not np.allclose(a, b, rtol=tol, atol=0, equal_nan=True)
        return not np.array_equal(a, b)
    if isinstance(a, float):
        return not (a == b or (a != a and b != b) or abs(a - b) <= tol * abs(b))
    return a != b


def _refine_amplification(gpu, before, call):
    """The Gauss-Newton pose refit iterates inside ONE call: its 6 x 6 normal equations are summed in lane order on the device and in
    index order on the oracle, and a selection that mixes two objects makes the iteration diverge (poses at 1e9: scored, never
    chosen) and amplifies the last-bit difference.  Evidence asked for before calling it that: replayed with 1, 2, .. iterations
    the two sides agree to 1e-12 after the first one and the distance then grows step by step (no jump from zero)."""
    pts = next((pa for name, pa, _, _ in reversed(before) if name == "set_points"), None)
    if pts is None:
        return None
    oc = OracleContext()
    oc.set_points(pts[0], pts[1])
    gpu.set_points(pts[0], pts[1])
    kw = dict(call[2])
    total = int(kw.pop("iterations", 10))
    dist = []
    for it in range(1, total + 1):
        pd, okd = gpu.pnp_refine_batch(*call[1], iterations=it, **kw)
        po, oko = oc.pnp_refine_batch(*call[1], iterations=it, **kw)
        if not np.array_equal(okd, oko):
            return None
        both = np.asarray(okd, dtype=bool)
        scale = np.maximum(np.abs(po[both]).max(axis=1, keepdims=True), 1e-300) if both.any() else 1.0
        dist.append(float((np.abs(pd[both] - po[both]) / scale).max()) if both.any() else 0.0)
    if dist[0] > 1e-12 or dist[-1] <= 1e-9:
        return None
    first = next(k for k, d in enumerate(dist) if d > 1e-9)
    if first > 0 and dist[first - 1] == 0.0:
        return None                                   # from bitwise equal to far apart in one step: not rounding
    return ("rounding amplified by a diverging Gauss-Newton iteration - max relative distance after 1 .. %d iterations: %s"
            % (total, " ".join(f"{d:.1e}" for d in dist)))


class _WithoutShortcuts:
    """the GPU context without its device-only shortcuts (pgx_score_inliers: the inlier list compacted on the device, where the
    oracle-backed context unpacks the mask row - the same set, tests/test_gpu_parity.py): both sides then make the SAME sequence of
    context calls and classify() can compare them call by call"""
    HIDDEN = ("score_inliers", "solve_minimal_sampled")

    def __init__(self, inner):
        self._i = inner

    def __getattr__(self, name):
        if name in self.HIDDEN:
            raise AttributeError(name)
        return getattr(self._i, name)


def classify(fn, args, kw, gpu, _aligned=False):
    """Replays a call that returned different results on the two sides and finds the first context call that explains it.
    "bug": a call whose arguments - and those of every call before it - were bitwise the same on both sides returned different
    integers, or floats further than 1e-9 apart.  "fp-order": the first difference is a floating-point sum within 1e-9 (Gram
    matrices and score sums are reduced in another order on the device) that the iteration then amplified."""
    logs = []
    for inner in (gpu, OracleContext()):
        rec = Recorder(inner)
        _api._ctx = rec
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fn(*args, **kw)
        logs.append(rec.log)
    _api._ctx = gpu
    for i, (a, b) in enumerate(zip(*logs)):
        if a[0] != b[0]:
            if not _aligned and (a[0] in _WithoutShortcuts.HIDDEN or b[0] in _WithoutShortcuts.HIDDEN):
                # the sequences part at a device-only shortcut, not at a result: compare again with the shortcut hidden (round 5: two
                # find6DPoses calls of a 1 200-call campaign were reported as "bug" here; aligned, both are the Gauss-Newton
                # amplification below - profiles/round5_soak_campaign.txt)
                kind, why = classify(fn, args, kw, _WithoutShortcuts(gpu), _aligned=True)
                _api._ctx = gpu
                return kind, why
            return "bug", f"call {i}: {a[0]} on the device, {b[0]} on the oracle, identical inputs so far"
        if _differ(a[1], b[1], 0.0) or _differ(a[2], b[2], 0.0):
            return "fp-order", f"call {i} {a[0]} is the first whose arguments differ (in rounding, or in the sign of a homogeneous model: every result before it agreed to 1e-9)"
        if _differ(a[3], b[3], 1e-9):
            if os.environ.get("SOAK_DUMP"):   # the arguments of the call and where the two results part
                np.set_printoptions(precision=17, linewidth=220)
                print("   call", i, a[0], "kwargs", {k: v for k, v in a[2].items() if not isinstance(v, np.ndarray)})
                as_dict = lambda r: r if isinstance(r, dict) else ({f"result{j}": v for j, v in enumerate(r)} if isinstance(r, list) else {"result": r})   # noqa: E731
                ra, rb = as_dict(a[3]), as_dict(b[3])
                for k in ra:
                    if k in rb and _differ(ra[k], rb[k], 1e-9):
                        x, y = np.asarray(ra[k]), np.asarray(rb[k])
                        w = np.nonzero(~np.isclose(x, y, rtol=1e-9, atol=0, equal_nan=True))[0] if x.shape == y.shape else []
                        print("   field", k, "differs at", w[:6], "device", x[w[:6]] if len(w) else x, "oracle", y[w[:6]] if len(w) else y)
                        if len(w) and a[1] and isinstance(a[1][0], np.ndarray) and a[1][0].ndim == 2:
                            print("   hypotheses", a[1][0][w[:3]], "other positional args", [v for v in a[1][1:] if not isinstance(v, np.ndarray)])
                for j, v in enumerate(a[1]):
                    if isinstance(v, np.ndarray):
                        np.save(os.environ["SOAK_DUMP"] + f"_call{i}_arg{j}.npy", v)
                for k in ra:
                    if isinstance(ra[k], np.ndarray):
                        np.save(os.environ["SOAK_DUMP"] + f"_call{i}_device_{k}.npy", ra[k])
                        np.save(os.environ["SOAK_DUMP"] + f"_call{i}_oracle_{k}.npy", rb[k])
                for name, pa, _, _ in reversed(logs[0][:i]):                 # the resident points of that call
                    if name == "set_points":
                        np.save(os.environ["SOAK_DUMP"] + "_points.npy", pa[1])
                        break
            if a[0] == "pnp_refine_batch" and int(a[2].get("iterations", 10)) > 1:
                why = _refine_amplification(gpu, logs[0][:i], a)
                if why:
                    return "fp-order", f"call {i} {a[0]}: {why}"
            return "bug", f"call {i} {a[0]}: identical inputs so far, results differ"
    return "fp-order", "every call agreed to 1e-9; the returned arrays differ beyond the soak's 1e-7"


def soak(seed, trials, verbose=True, only=None):
    rng = np.random.default_rng(seed)
    bad = 0
    chaos = 0
    found = 0
    t0 = time.time()
    _api._ctx = None
    gpu = None
    for trial in range(trials):
        which = trial % 5
        s = int(rng.integers(1 << 30))
        kw = dict(conf=float(rng.choice([0.5, 0.9, 0.99])), seed=int(rng.integers(1000)),
                  spatial_coherence_weight=float(rng.choice([0.0, 0.0, 0.05, 0.14, 0.5])),
                  maximum_tanimoto_similarity=float(rng.choice([0.2, 0.4, 0.9])),
                  max_iters=int(rng.choice([50, 200, 600])),
                  maximum_model_number=int(rng.choice([-1, -1, 1, 2, 4])),
                  neighborhood=str(rng.choice(["flann_like", "knn:6", "radius"])),
                  local_optimization=str(rng.choice(["auto", "lsq"])),
                  labeling_l0=str(rng.choice(["greedy", "expansion"])),
                  sampler_rng=str(rng.choice(["numpy", "philox"])), refit_solver=str(rng.choice(["lapack", "lapack", "jacobi"])))
        K = int(rng.integers(1, 5))
        per = int(rng.choice([40, 150, 400, 1500]))
        nout = int(rng.choice([0, 50, 400]))
        if which == 0:
            pts, gt, _ = datasets.make_lines(n_per_line=per, n_lines=K, n_outliers=nout, seed=s)
            fn, args = px.findLines, (pts, np.array(0), 1000, 1000)
            kw.update(threshold=float(rng.choice([1.0, 2.0, 4.0])), sampler_id=int(rng.choice([0, 0, 1, 1, 2, 2, 2, 3])),
                      minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([20.0, 60.0, 200.0])))
        elif which == 1:
            pts, gt, _ = datasets.make_homographies(n_per_plane=per, n_planes=K, n_outliers=nout, seed=s)
            fn, args = px.findHomographies, (pts, 1000, 1000, 1000, 1000)
            kw.update(threshold=float(rng.choice([1.0, 3.0, 6.0])), sampler_id=int(rng.choice([0, 0, 1, 1, 2, 2, 2, 3])),
                      minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([60.0, 200.0])),
                      residual=str(rng.choice(["transfer", "symmetric"])), scoring_exponent=int(rng.choice([1, 2, 3])))
        elif which == 2:
            pts, gt, _ = datasets.make_two_view_motions(n_per_motion=per, n_motions=min(K, 3), n_outliers=nout, seed=s)
            fn, args = px.findTwoViewMotions, (pts, 1000, 1000, 1000, 1000)
            kw.update(threshold=float(rng.choice([0.5, 0.75, 2.0])), sampler_id=int(rng.choice([0, 0, 1, 1, 2, 2, 2, 3])),
                      minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([50.0, 200.0])))
        elif which == 3:
            pts, gt, _ = datasets.make_vanishing_points(n_inliers=per * K, n_vps=K, n_outliers=nout, seed=s)
            fn, args = px.findVanishingPoints, (pts, np.array(0) if rng.random() < 0.5 else rng.random(len(pts)), 1000, 1000)
            kw.update(threshold=float(rng.choice([0.5, 1.5, 3.0])), sampler_id=int(rng.choice([0, 1, 3])),
                      minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([15.0, 100.0])))
        else:
            x1, x2, Kc, gt, _ = datasets.make_poses(n_per_object=per, n_objects=min(K, 3), n_outliers=nout, seed=s)
            fn, args = px.find6DPoses, (x1, x2, Kc)
            for k in ("sampler_id",):
                kw.pop(k, None)
            kw.update(threshold=float(rng.choice([2.0, 4.0, 8.0])), minimum_point_number=int(rng.choice([6, 30])),
                      neighborhood_ball_radius=float(rng.choice([20.0, 60.0])))
        if only is not None and not (only[0] <= trial <= only[1]):
            continue
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                _api._ctx = gpu
                M, lab = fn(*args, **kw)
                gpu = _api._ctx
                _api._ctx = OracleContext()
                Mr, labr = fn(*args, **kw)
        finally:
            _api._ctx = gpu
        found += M.shape[0]
        ok = np.array_equal(lab, labr) and M.shape == Mr.shape
        if ok and M.size:
            # lines, vanishing points, homographies and fundamental matrices are homogeneous: a refit's eigenvector comes with
            # either sign (the Gram sums' rounding decides), so those are compared up to the sign of each model
            rows = 3 if which in (1, 2, 4) else 1
            A, B = M.reshape(-1, rows * M.shape[1]), Mr.reshape(-1, rows * M.shape[1])
            tol = 1e-7 * np.abs(B).max(axis=1, keepdims=True) + 1e-9
            same = np.all(np.abs(A - B) <= tol, axis=1)
            if which != 4:
                same |= np.all(np.abs(A + B) <= tol, axis=1)
            ok = bool(same.all())
        if not ok:
            kind, why = classify(fn, args, kw, gpu)
            bad += kind == "bug"
            chaos += kind != "bug"
            print("MISMATCH" if kind == "bug" else "fp-order divergence", "trial", trial, "-", why, "|", fn.__name__, "data seed", s, "K", K, "per", per, "nout", nout, kw, "models", M.shape, Mr.shape,
                  "labels differing", int((lab != labr).sum()) if lab.shape == labr.shape else "shape",
                  "max model diff", float(np.abs(M - Mr).max()) if M.shape == Mr.shape and M.size else None, flush=True)
    if verbose:
        print(f"api soak done: seed {seed}, {trials} calls, {bad} mismatches, {chaos} fp-order divergences, {found} model rows returned, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    soak(int(sys.argv[1]), int(sys.argv[2]), only=(int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else None)   # optional: first and last trial to run
