"""The five entry points of the reference's pybind11 module, same names / argument order / defaults / return layout /
error messages (/root/reference/src/pyprogressivex/src/bindings.cpp:9-392 wrappers, :410-491 defaults) and the
parameter plumbing of the problem drivers (/root/reference/src/pyprogressivex/src/progressivex_python.cpp:41-666),
including their quirks: unknown sampler ids print to stderr and return zero models,
findTwoViewMotions ignores scoring_exponent, findLines ignores weights, only findVanishingPoints forwards do_logging.

Extensions that do not change the reference behaviour when left at their defaults (keyword-only):
  seed=None                     reproducible sampling (the reference seeds from std::random_device)
  sampler_rng="numpy"           "philox": the samplers draw from the in-repo counter-based generator (_rng.py) - uniform, NAPSAC and PROSAC on the device inside the solver's launch, Progressive NAPSAC (sequential) in libpgx's host code
  refit_solver="lapack"         "jacobi": the refits' smallest-eigenvector solves run through pgx_eigh_smallest_batch (batched cyclic Jacobi
                                on the device, bitwise a CPU restatement, ~1e-13 from LAPACK) instead of numpy.linalg.eigh: round 6
  distributed=None              True: shard the proposal batches over the ranks of this launch (every rank must make the same
                                call on the same data; checked).  None: only if PGX_MULTI_GPU=1.  Never implicit.
  max_outer_iterations=10       the reference's hard cap on proposals per call (progressive_x.h:272)
  residual="transfer"           findHomographies: "symmetric" switches to the symmetric transfer error (U-1 switch)
  neighborhood="flann_like"     U-7 switch: "flann_like" (<= 5 nearest neighbours inside the ball, what upstream's
                                checks=6 approximate search can return at most), "radius" (all points in the ball),
                                or "knn:<k>"
  labeling_l0="greedy"          U-8 switch, only matters when spatial_coherence_weight == 0 (the default of four of the
                                five entry points): PEARL then sets no smooth cost, and GCO-v3's expansion() labels such
                                an energy by its special-case solver (greedy facility location over the label costs)
                                instead of alpha-expansion; "expansion" = alpha-expansion moves (round 1's behaviour)
  local_optimization="auto"     U-12 switch: "auto" = GC-RANSAC's graph-cut local optimisation whenever the spatial
                                coherence weight is in (0, 1) (the cut runs on the GPU, pgx_gc_labeling), iterated
                                least-squares refits otherwise; "lsq" = refits only
"""
import sys

import numpy as np

from . import _engine, _estimators, _lib, _proposal

_ctx = None


def _context():
    """One GPU context per process (device = $PGX_DEVICE or $LOCAL_RANK or 0).  Raises if libpgx.so or the GPU is
    missing — there is no CPU path."""
    global _ctx
    if _ctx is None:
        import os
        dev = int(os.environ.get("PGX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ctx = _lib.Context(dev)
    return _ctx


def _as_f64(a):
    # py::array_t<double> with forcecast (bindings.cpp:10): any dtype is converted; the raw buffer is then read as if
    # C-contiguous, which np.ascontiguousarray makes explicit
    return np.ascontiguousarray(np.asarray(a), dtype=np.float64)


def _shape2(a):
    if a.ndim < 2:
        raise ValueError("array must be 2-dimensional")  # pybind11 would fail on buf.shape[1]
    return a.shape[0], a.shape[1]


def _weights(weights_, n=None):
    """bindings.cpp:200-208: a 0-d / empty array means "no weights".  The reference then reads weights[i] for every point
    without a length check (undefined behaviour for a short array); here a length other than n is a ValueError."""
    w = np.asarray(weights_, dtype=np.float64)
    if w.ndim == 0 or w.size == 0:
        return None
    w = np.ascontiguousarray(w).reshape(-1)
    if n is not None and w.shape[0] != n:
        raise ValueError(f"weights should have one entry per row ({n}), got {w.shape[0]}")
    return w


def _unknown_sampler(sampler_id):
    # progressivex_python.cpp:240-245 (message kept literally, including its omission of id 3)
    sys.stderr.write(f"Unknown sampler identifier: {sampler_id}. The accepted samplers are 0 (uniform sampling), "
                     "1 (PROSAC sampling), 2 (P-NAPSAC sampling)\n")


def _run(estimator, pts, graph_points, radius, sampler_factory, *, threshold, conf, spatial_coherence_weight,
         maximum_tanimoto_similarity, max_iters, minimum_point_number, maximum_model_number, scoring_exponent=2,
         do_logging=False, weights=None, seed=None, max_outer_iterations=10, neighborhood="flann_like",
         local_optimization="auto", labeling_l0="greedy", distributed=None, sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    n = pts.shape[0]
    if sampler_rng not in ("numpy", "philox"):
        raise ValueError("sampler_rng should be 'numpy' or 'philox'")
    if refit_solver not in ("lapack", "jacobi"):
        raise ValueError("refit_solver should be 'lapack' or 'jacobi'")
    estimator.refit_solver = str(refit_solver)   # the refits' small eigen-solves: numpy on the host / pgx_eigh_smallest_batch (_estimators.py)
    if getattr(sampler_factory, "unknown", False):
        # progressivex_python.cpp:240-245: message on stderr, zero models, labelling left at its initial zeros
        _unknown_sampler(sampler_factory.sampler_id)
        return [], np.zeros(n, dtype=np.int32), None
    ctx = _context()
    # multi-GPU, OPT-IN (distributed=True or PGX_MULTI_GPU=1, one process per GPU, WORLD_SIZE > 1): the proposal batches are
    # sharded over the ranks and the score triples all-gathered over RCCL (parallel.py); everything else runs replicated, so
    # every rank returns the same result.  Every rank must make this call with the same data (checked) and draw the same
    # samples: an unset seed is replaced by one the launch agrees on.
    from . import parallel
    exchange = parallel.default_exchange(ctx, distributed)
    if exchange is not None:
        parallel.check_same_problem(exchange, pts, params=(
            type(estimator).__name__, float(radius), getattr(sampler_factory, "sampler_id", None), float(threshold), float(conf),
            float(spatial_coherence_weight), float(maximum_tanimoto_similarity), int(max_iters), int(minimum_point_number),
            int(maximum_model_number), int(scoring_exponent), seed, int(max_outer_iterations), str(neighborhood),
            str(local_optimization), str(labeling_l0), str(sampler_rng), getattr(estimator, "validity", None), str(pearl_abs), str(refit_solver)))
        if seed is None:
            seed = parallel.shared_seed()
    rng = np.random.default_rng(seed)
    # FlannNeighborhoodGraph(&points, radius) [U-7]: built on the GPU (pgx_graph_build) and left resident there; the
    # CSR comes back for the neighbourhood samplers.
    resident = True
    # ... and it only comes BACK to the host for the sampler that walks it (NAPSAC): 74 MB of CSR at N = 1e6 otherwise cross PCIe
    # for nothing (find6DPoses always samples uniformly)
    fetch = bool(getattr(sampler_factory, "needs_graph", True))
    if neighborhood == "radius":
        graph = ctx.graph_build(graph_points, _lib.GRAPH_BALL, radius=radius, fetch=fetch)
    elif str(neighborhood).startswith("knn:"):
        graph = ctx.graph_build(graph_points, _lib.GRAPH_KNN, k=int(str(neighborhood)[4:]), fetch=fetch)
    else:
        graph = ctx.graph_build(graph_points, _lib.GRAPH_KNN_IN_BALL, radius=radius, k=5, fetch=fetch)
    if not fetch:
        graph = None
    if sampler_rng == "philox" and getattr(sampler_factory, "kind", None) == "pnapsac":   # (sequential: drawn by libpgx's host code)
        sampler = _proposal.PhiloxProgressiveNapsacSampler(n, rng, *sampler_factory.pnapsac_args)
    else:
        sampler = sampler_factory(n, rng, graph)
    if sampler_rng == "philox" and type(sampler) is _proposal.UniformSampler:   # the in-repo counter-based generator (device-drawable)
        sampler = _proposal.PhiloxUniformSampler(n, rng)
    elif sampler_rng == "philox" and type(sampler) is _proposal.NapsacSampler:
        sampler = _proposal.PhiloxNapsacSampler(n, rng, graph)
    elif sampler_rng == "philox" and type(sampler) is _proposal.ProsacSampler:
        sampler = _proposal.PhiloxProsacSampler(n, rng, sample_size=sampler.prosac_m, convergence_iterations=sampler.t_n)
    s = _engine.MultiModelSettings()
    s.minimum_number_of_inliers = int(minimum_point_number)          # progressivex_python.cpp:261
    s.inlier_outlier_threshold = float(threshold)                    # :263
    s.set_confidence(float(conf))                                    # :265
    s.maximum_tanimoto_similarity = float(maximum_tanimoto_similarity)   # :267
    s.spatial_coherence_weight = float(spatial_coherence_weight)     # :269
    s.max_iteration_number = int(max_iters)                          # :271
    if maximum_model_number > 0:                                     # :273-274
        s.maximum_model_number = int(maximum_model_number)
    s.point_weights = weights
    s.max_outer_iterations = int(max_outer_iterations)
    s.local_optimization = str(local_optimization)   # "auto": GC-RANSAC's graph-cut LO; "lsq": refits only
    if labeling_l0 not in ("greedy", "expansion"):
        raise ValueError("labeling_l0 should be 'greedy' or 'expansion'")
    s.labeling_l0 = str(labeling_l0)
    if pearl_abs not in ("double", "int"):
        raise ValueError("pearl_abs should be 'double' or 'int'")
    s.pearl_abs = str(pearl_abs)        # [U-16] which abs() PEARL.h:465 resolves to
    if trace is not None and hasattr(trace, "begin"):
        # diagnostics hook (_engine.EV_*): what the run is made of, for a replay of its decisions (tests/)
        trace.begin(dict(model_type=estimator.model_type, points=pts, graph_points=graph_points, radius=float(radius),
                         neighborhood=str(neighborhood), sample_size=int(estimator.sample_size),
                         nonminimal_sample_size=int(estimator.nonminimal_sample_size), settings=s))
    px = _engine.ProgressiveX(ctx, estimator, pts, graph, sampler, s, scoring_exponent=scoring_exponent,
                              do_logging=do_logging, graph_resident=resident, exchange=exchange, trace=trace)
    models, stats = px.run()
    labeling = np.asarray(stats.labeling, dtype=np.int64).astype(np.int32)     # bindings.cpp:152-156
    return models, labeling, stats


def _stack(estimator, models, cols):
    rows = estimator.rows_per_model
    out = np.zeros((rows * len(models), cols), dtype=np.float64)
    for k, m in enumerate(models):
        out[rows * k: rows * (k + 1)] = estimator.output(m.descriptor)
    return out


def _sampler_factory(sampler_id, valid, pts=None, sizes=None, sample_size=None, prosac_sample_size=None):
    def make(n, rng, graph):
        kind = valid[sampler_id]
        if kind == "uniform":
            return _proposal.UniformSampler(n, rng)
        if kind == "prosac":
            return _proposal.ProsacSampler(n, rng, sample_size=prosac_sample_size)
        if kind == "pnapsac":                # ProgressiveNapsacSampler<4>(&points, {16,8,4,2}, m, {w1,h1,w2,h2}, 0.5)
            return _proposal.ProgressiveNapsacSampler(n, rng, pts, sizes, sample_size)
        return _proposal.NapsacSampler(n, rng, graph)
    make.unknown = sampler_id not in valid
    make.kind = valid.get(sampler_id)
    make.pnapsac_args = (pts, sizes, sample_size)
    make.needs_graph = valid.get(sampler_id) == "napsac"
    make.sampler_id = sampler_id
    return make


def findHomographies(corrs, w1, h1, w2, h2, threshold=4.0, conf=0.5, spatial_coherence_weight=0.0,
                     neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
                     minimum_point_number=10, maximum_model_number=-1, sampler_id=3, scoring_exponent=2,
                     do_logging=False, *, seed=None, max_outer_iterations=10, residual="transfer", neighborhood="flann_like",
                     local_optimization="auto", labeling_l0="greedy", distributed=None, sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    """bindings.cpp:99-166, progressivex_python.cpp:173-304.  Returns (H[(3K),3] float64, labels[n] int32)."""
    corrs = _as_f64(corrs)
    n, dim = _shape2(corrs)
    if dim != 4 or n < 4:
        raise ValueError("corrs should be an array with dims [n,4], n>=4")           # bindings.cpp:119-124
    if do_logging and sampler_id == 1:
        print("Note: PROSAC sampler requires the correspondences to be order by quality, e.g., SNN ratio.")
    if do_logging and sampler_id == 2:
        print("Note: Progressive NAPSAC sampler requires the correspondences to be order by quality, e.g., SNN ratio.")
    est = (_estimators.SymmetricHomographyEstimator() if residual == "symmetric" else _estimators.HomographyEstimator())
    # NB: the driver never calls progressive_x.log(): do_logging only triggers the sampler notes (:219-226)
    models, labels, _ = _run(est, corrs, corrs, neighborhood_ball_radius,
                             _sampler_factory(sampler_id, {0: "uniform", 1: "prosac", 2: "pnapsac", 3: "napsac"}, corrs,
                                              (w1, h1, w2, h2), est.sample_size),
                             threshold=threshold, conf=conf, spatial_coherence_weight=spatial_coherence_weight,
                             maximum_tanimoto_similarity=maximum_tanimoto_similarity, max_iters=max_iters,
                             minimum_point_number=minimum_point_number, maximum_model_number=maximum_model_number,
                             scoring_exponent=scoring_exponent, do_logging=False, seed=seed,
                             max_outer_iterations=max_outer_iterations, neighborhood=neighborhood, local_optimization=local_optimization, labeling_l0=labeling_l0, distributed=distributed, sampler_rng=sampler_rng, trace=trace, pearl_abs=pearl_abs, refit_solver=refit_solver)
    return _stack(est, models, 3), labels


def findTwoViewMotions(corrs, w1, h1, w2, h2, threshold=4.0, conf=0.5, spatial_coherence_weight=0.0,
                       neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
                       minimum_point_number=10, maximum_model_number=-1, sampler_id=3, scoring_exponent=3,
                       do_logging=False, *, seed=None, max_outer_iterations=10, neighborhood="flann_like",
                     local_optimization="auto", labeling_l0="greedy", distributed=None, validity="off", sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    """bindings.cpp:324-392, progressivex_python.cpp:537-666.  Returns (F[(3K),3], labels[n]).
    validity [U-14, keyword-only, not in the reference's signature]: which of the estimator's model-validity stages run -
    "off" (the default: strict restatement of what is in the snapshot, i.e. none), "oriented", "symmetric" (oriented +
    symmetric-epipolar support) or "full" (+ DEGENSAC: the recollection of gcransac's FundamentalMatrixEstimator::isValidModel;
    its source is absent from the snapshot and no bundled scene measures better with it, so it is opt-in - INTEGRATION.md)."""
    if validity not in ("off", "oriented", "symmetric", "full"):
        raise ValueError("validity should be 'off', 'oriented', 'symmetric' or 'full'")
    corrs = _as_f64(corrs)
    n, dim = _shape2(corrs)
    if dim != 4 or n < 7:
        raise ValueError("corrs should be an array with dims [n,4], n>=7")           # bindings.cpp:344-349
    if do_logging and sampler_id == 1:
        print("Note: PROSAC sampler requires the correspondences to be order by quality, e.g., SNN ratio.")
    if do_logging and sampler_id == 2:
        print("Note: Progressive NAPSAC sampler requires the correspondences to be order by quality, e.g., SNN ratio.")
    est = _estimators.FundamentalEstimator()
    est.validity = validity
    # the driver ignores scoring_exponent (never calls setScoringExponent: :621-638) => ProgressiveX's default 2
    models, labels, _ = _run(est, corrs, corrs, neighborhood_ball_radius,
                             _sampler_factory(sampler_id, {0: "uniform", 1: "prosac", 2: "pnapsac", 3: "napsac"}, corrs,
                                              (w1, h1, w2, h2), est.sample_size),
                             threshold=threshold, conf=conf, spatial_coherence_weight=spatial_coherence_weight,
                             maximum_tanimoto_similarity=maximum_tanimoto_similarity, max_iters=max_iters,
                             minimum_point_number=minimum_point_number, maximum_model_number=maximum_model_number,
                             scoring_exponent=2, do_logging=False, seed=seed,
                             max_outer_iterations=max_outer_iterations, neighborhood=neighborhood, local_optimization=local_optimization, labeling_l0=labeling_l0, distributed=distributed, sampler_rng=sampler_rng, trace=trace, pearl_abs=pearl_abs, refit_solver=refit_solver)
    return _stack(est, models, 3), labels


findFundamentalMatrices = findTwoViewMotions   # name used by BASELINE.json's north star


def findVanishingPoints(lines, weights, w, h, threshold=4.0, conf=0.5, spatial_coherence_weight=0.0,
                        neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
                        minimum_point_number=10, maximum_model_number=-1, sampler_id=3, scoring_exponent=2,
                        do_logging=False, *, seed=None, max_outer_iterations=10, neighborhood="flann_like",
                     local_optimization="auto", labeling_l0="greedy", distributed=None, sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    """bindings.cpp:168-245, progressivex_python.cpp:306-423.  Returns (vp[K,3], labels[n]).  Only sampler ids 0/1
    exist for this driver, so the DEFAULT id 3 returns zero models, as in the reference."""
    lines = _as_f64(lines)
    n, dim = _shape2(lines)
    if dim != 4 or n < 2:
        raise ValueError("lines should be an array with dims [n,4], n>=2")           # bindings.cpp:188-193
    if do_logging and sampler_id == 1:
        print("Note: PROSAC sampler requires the correspondences to be order by quality, e.g., SNN ratio.")
    est = _estimators.VanishingPointEstimator()
    models, labels, _ = _run(est, lines, lines, neighborhood_ball_radius,
                             _sampler_factory(sampler_id, {0: "uniform", 1: "prosac"}),
                             threshold=threshold, conf=conf, spatial_coherence_weight=spatial_coherence_weight,
                             maximum_tanimoto_similarity=maximum_tanimoto_similarity, max_iters=max_iters,
                             minimum_point_number=minimum_point_number, maximum_model_number=maximum_model_number,
                             scoring_exponent=scoring_exponent, do_logging=bool(do_logging),     # :401
                             weights=_weights(weights, n), seed=seed, max_outer_iterations=max_outer_iterations,
                             neighborhood=neighborhood, local_optimization=local_optimization, labeling_l0=labeling_l0, distributed=distributed, sampler_rng=sampler_rng, trace=trace, pearl_abs=pearl_abs, refit_solver=refit_solver)
    return _stack(est, models, 3), labels


def findLines(points, weights, w, h, threshold=2.0, conf=0.5, spatial_coherence_weight=0.0,
              neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
              minimum_point_number=10, maximum_model_number=-1, sampler_id=3, scoring_exponent=2,
              do_logging=False, *, seed=None, max_outer_iterations=10, neighborhood="flann_like",
                     local_optimization="auto", labeling_l0="greedy", distributed=None, sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    """bindings.cpp:247-322, progressivex_python.cpp:425-535.  Returns (lines[K,3], labels[n]).  Sampler ids 0/1/2
    (2 = NAPSAC here); the default 3 returns zero models; `weights` is parsed and ignored, as in the reference."""
    points = _as_f64(points)
    n, dim = _shape2(points)
    if dim != 2 or n < 2:
        raise ValueError("Points should be an array with dims [n,3], n>=2")          # bindings.cpp:267-270 (sic)
    _weights(weights)         # parsed and unused (progressivex_python.cpp:466-482): any length is accepted, as upstream
    if do_logging and sampler_id == 1:
        print("Note: PROSAC sampler requires the points to be order by quality, e.g., SNN ratio.")
    est = _estimators.LineEstimator()
    models, labels, _ = _run(est, points, points, neighborhood_ball_radius,
                             _sampler_factory(sampler_id, {0: "uniform", 1: "prosac", 2: "napsac"},
                                              prosac_sample_size=4),     # the HOMOGRAPHY sample size (quirk, :463)
                             threshold=threshold, conf=conf, spatial_coherence_weight=spatial_coherence_weight,
                             maximum_tanimoto_similarity=maximum_tanimoto_similarity, max_iters=max_iters,
                             minimum_point_number=minimum_point_number, maximum_model_number=maximum_model_number,
                             scoring_exponent=scoring_exponent, do_logging=False, seed=seed,
                             max_outer_iterations=max_outer_iterations, neighborhood=neighborhood, local_optimization=local_optimization, labeling_l0=labeling_l0, distributed=distributed, sampler_rng=sampler_rng, trace=trace, pearl_abs=pearl_abs, refit_solver=refit_solver)
    return _stack(est, models, 3), labels


def find6DPoses(x1y1, x2y2z2, K, threshold=4.0, conf=0.90, spatial_coherence_weight=0.1,
                neighborhood_ball_radius=20.0, maximum_tanimoto_similarity=0.9, max_iters=400,
                minimum_point_number=2 * 3, maximum_model_number=-1, *, seed=None, max_outer_iterations=10, scoring_exponent=2,
                neighborhood="flann_like",
                     local_optimization="auto", labeling_l0="greedy", distributed=None, sampler_rng="numpy", trace=None, pearl_abs="double", refit_solver="lapack"):
    """bindings.cpp:9-97, progressivex_python.cpp:41-171.  Returns (P[(3K),4], labels[n]).
    scoring_exponent [keyword-only, not in the reference's signature]: the driver never calls setScoringExponent, so the class default
    2 applies (progressive_x.h:183) - the default here.  The compound-model score is value - shared^exponent
    (scoring_function_with_compound_model.h:110-121): at 10^6 points a new object's incidental overlap with a dozen accepted models
    (~300-500 units of shared support) squared exceeds its whole value (~47 000), its score goes negative and junk hypotheses win the
    proposal - the call stops at 12-14 of BASELINE config C4's 16 objects (scripts/c4_missing.py).  exponent = 1 keeps the penalty
    on the scale of the support."""
    import time
    x1 = _as_f64(x1y1)
    n, dim = _shape2(x1)
    if dim != 2 or n < 3:
        raise ValueError("x1y1 should be an array with dims [n,2], n>=3")            # bindings.cpp:26-31
    x2 = _as_f64(x2y2z2)
    na, dima = _shape2(x2)
    if dima != 3:
        raise ValueError("x2y2z2 should be an array with dims [n,3], n>=3")          # :37-39
    if na != n:
        raise ValueError("x1y1 and x2y2z2 should be the same size")                  # :40-42
    Km = _as_f64(K)
    if Km.ndim != 2 or Km.shape != (3, 3):
        raise ValueError("K should be an array with dims [3,3]")                     # :48-50
    Kinv = np.linalg.inv(Km)                                                          # progressivex_python.cpp:69-70
    un = Kinv[0, 0] * x1[:, 0] + Kinv[0, 1] * x1[:, 1] + Kinv[0, 2]                    # :88
    vn = Kinv[1, 0] * x1[:, 0] + Kinv[1, 1] * x1[:, 1] + Kinv[1, 2]                    # :89
    normalized = np.ascontiguousarray(np.column_stack([un, vn, x2]))                   # :88-92
    raw = np.column_stack([x1, x2])                                                    # :82-86 (graph on RAW points)
    f = 0.5 * (Km[0, 0] + Km[1, 1])                                                    # :96
    t0 = time.perf_counter()
    est = _estimators.PnPEstimator()

    def factory(n_, rng, graph):
        print("Neighborhood calculation time = %f secs." % (time.perf_counter() - t0))   # :109
        return _proposal.UniformSampler(n_, rng)                                          # :112 always uniform
    factory.needs_graph = False

    models, labels, _ = _run(est, normalized, raw, neighborhood_ball_radius, factory,
                             threshold=threshold / f, conf=conf, spatial_coherence_weight=spatial_coherence_weight,
                             maximum_tanimoto_similarity=maximum_tanimoto_similarity, max_iters=max_iters,
                             minimum_point_number=minimum_point_number, maximum_model_number=maximum_model_number,
                             scoring_exponent=int(scoring_exponent), do_logging=False, seed=seed,
                             max_outer_iterations=max_outer_iterations, neighborhood=neighborhood, local_optimization=local_optimization, labeling_l0=labeling_l0, distributed=distributed, sampler_rng=sampler_rng, trace=trace, pearl_abs=pearl_abs, refit_solver=refit_solver)
    return _stack(est, models, 4), labels
