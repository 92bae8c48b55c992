#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: batched hypothesis scoring, 1e6 2D-3D correspondences x 2048 pose
hypotheses per GPU (config C4 points + the metric batch of SURVEY.md §8d), through the C ABI of libpgx.so.

One step = one pass of the proposal hot path over one batch: score every hypothesis of the batch against all points
(MSAC + compound-model score, scoring_function_with_compound_model.h:61-125), exchange the per-hypothesis results
when N > 1, fetch them and select the winner on the host.  Inputs (points, compound preference vector, hypotheses) are
resident in HBM before the timed region starts.  --gpus N (default mode, --scaling strong-points = BASELINE's fixed total
work "1e6 pts x 2048 hyps, 1/2/4/8 GPU"): rank r holds 1/N of the points and scores all 2048 hypotheses against them, the
integer accumulators are summed by an RCCL all-reduce (bitwise the 1-GPU table), value = (points x hypotheses of the JOB) /
time, "scaling": "strong".  --scaling strong / weak: the hypotheses split over the ranks (all-gather) / 2048 per rank.

The JSON line carries, next to the headline:
  roofline       HBM roofline of the dominant kernel: ALGORITHMIC bytes of SURVEY 8(d) (N d 8 + M p 8 + M 16 + N 8 for the
                 compound vector; no derived copies) / that kernel's HIP-event time measured live; PMC traffic from the
                 committed rocprofv3 passes; executed work from device counters (pgx_score_stats)
  cpu_baseline   the oracle (single-threaded C port of the reference path) on this box's host cores, bounded sample
  legs           secondary measurements on 1 GPU (never the headline): the same batch uploaded and sorted inside the step,
                 a RANSAC-like batch generated inside the step by the device P3P solver (samples -> solve -> score -> fetch
                 -> select: end-to-end models/s), the dense unfiltered kernel with its true FP64 fraction, Sampson scoring at
                 C3 size, vanishing-point scoring at C5 size, one pgx_set_points

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 via
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
(RANK / LOCAL_RANK / WORLD_SIZE from the env; torch itself is not imported: the data plane is RCCL inside libpgx).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "progressive-x_amd"))

# HBM bytes per launch of the default workload from the committed PMC passes, summed over the kernels of one launch:
# FETCH_SIZE x 2 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, in KiB.
PMC = {"source": "profiles/round1_bench_v12_final.txt",
       "fetch_kib": 267746.8 + 3439.0 + 36.5, "write_kib": 27527.2 + 4276.5 + 158.8,
       "valu_busy_frac": 0.56}       # SQ pass of the same file: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_current.json")   # refreshed by scripts/profile_bench.sh when re-profiled
if os.path.exists(PMC_FILE):
    with open(PMC_FILE) as _f:
        PMC = json.load(_f)
F32_FLOPS_PER_FILTER_TEST = 31   # Filter32<PnP>::reject: 13 FMA (2 flops) + 3 mul + 2 compares
# Counter calibration (profiles/round5_fetch_calibration.txt, scripts/micro/fetch_calib.hip: known-bytes kernels in this path's access
# patterns under the same --pmc passes): FETCH_SIZE reports exactly 1/2 of a streaming read at 16, 8 (group-blocked rows included)
# and 4 bytes per lane alike -> x2 holds for the 8 B/lane group kernel too; WRITE_SIZE reports coalesced stores 1:1 but tallies every
# L2 atomic as a 32-byte write although the 48 KB accumulator table never leaves L2 -> the group kernel's WRITE_SIZE (it writes nothing
# but accumulator atomics) is not HBM traffic.  traffic = 2 FETCH + WRITE - atomic WRITE; traffic_raw = 2 FETCH + WRITE.
PMC_TRAFFIC_RAW = int((2 * PMC["fetch_kib"] + PMC["write_kib"]) * 1024)
PMC_TRAFFIC_DEFAULT = int((2 * PMC["fetch_kib"] + PMC["write_kib"] - PMC.get("atomic_write_kib", 0.0)) * 1024)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes of ONE expansion from zeros per config from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over every min-cut
# kernel (scripts/profile_labelling.sh): {"c4": {"bytes": ..., "source": ...}, ...}; absent until the pass has been run
PMC_LABELLING = {}
_pl = os.path.join(ROOT, "profiles", "pmc_labelling.json")
if os.path.exists(_pl):
    with open(_pl) as _f:
        PMC_LABELLING = json.load(_f)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64 counting an FMA as 2 flops; parity mode may not contract => 39.3 usable
FLOPS_PER_PAIR = {"pnp": 25, "fundamental": 33, "vanishing_point": 24}   # exact residual + score update, docs/lab-notebook.md 5.1


def cpu_baseline(pts, hyps, T2, comp, budget_s=15.0):
    """The oracle (a single-threaded C restatement of the reference path, kind = 'port') on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pgx_oracle as O
    O.lib()
    t0 = time.perf_counter()
    O.score(O.PNP, pts, hyps[:8], T2, compound=comp, has_compound=True, exponent=2)
    per_hyp = (time.perf_counter() - t0) / 8
    runs = 5                                                  # median of >= 5 runs
    m = int(max(8, min(hyps.shape[0], budget_s / runs / max(per_hyp, 1e-9))))
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        O.score(O.PNP, pts, hyps[:m], T2, compound=comp, has_compound=True, exponent=2)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": pts.shape[0] * m / dt, "unit": "residual-evals/s", "cores": 1, "kind": "port",
           "models_per_sec": m / dt,
           "sample": f"{m} of {hyps.shape[0]} hypotheses x all {pts.shape[0]} points, median of {runs} runs ({dt:.2f} s each), 1 thread, "
                     "oracle built -O3 -ffp-contract=off (the reference: -O3, no -march => no FMA)",
           "run_seconds": [round(t, 3) for t in times],
           "host_cpu": cpu, "host_cores_available": os.cpu_count()}
    # NOT the reference's configuration (it is single-threaded): the same port with the hypotheses split over all
    # host cores (threads calling the C function, which runs without the GIL), reported for context only
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, hyps.shape[0]))
        chunks = np.array_split(np.arange(hyps.shape[0]), cores)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda ix: O.score(O.PNP, pts, hyps[ix], T2, compound=comp, has_compound=True, exponent=2), chunks))
        dta = time.perf_counter() - t0
        out["all_cores"] = {"value": pts.shape[0] * hyps.shape[0] / dta, "unit": "residual-evals/s", "cores": cores,
                            "sample": f"all {hyps.shape[0]} hypotheses x all {pts.shape[0]} points, {dta:.2f} s, {cores} threads",
                            "note": "context only: the reference is single-threaded"}
    except Exception as e:   # the single-thread figure is the baseline; never fail the bench over the context number
        out["all_cores"] = {"error": str(e)}
    return out


def cpu_labelling_baseline(bk_c5=False):
    """SURVEY 8(d) metric 3 / BASELINE.md 3 on the host: ONE alpha-expansion from the all-zero labelling (what every PEARL::run starts
    with, PEARL.h:507-551) at C3 and C5 size, single thread, on the problems the GPU legs solved (CPU_LABELLING).  Two solvers of the
    oracle: Boykov-Kolmogorov (oracle/bk_maxflow.c: the algorithm GCoptimization uses behind PEARL.h:550) and Dinic; they return
    identical labels (unique minimal sink side) and the FASTER one is the baseline of a config - BK wins by 3-15x on the plain
    pairwise networks but loses to Dinic when the label-cost hubs (one node joined to every site of a label) are in play at 2e5
    sites, so C5's BK run (~110 s) is opt-in (--cpu-labelling-bk-c5; measured once: profiles/).  C4 (1e6 sites, Dinic 402 s) is
    beyond a bench budget: see DESIGN.md."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pgx_oracle as O
    out = {"kind": "port", "cores": 1, "solver": "faster of Boykov-Kolmogorov (oracle/bk_maxflow.c) and Dinic (oracle/pgx_oracle.c) per config",
           "unit": "expansion cycles/s, min-cuts/s of one expansion from zeros", "configs": {}}
    for key, (Dq, graph, lam, h, gpu_labels) in CPU_LABELLING.items():
        n, L = Dq.shape
        lq, hq = O.quantize_lambda(lam), O.quantize(h)
        z = np.zeros(n, np.int32)
        rec = {"sites": int(n), "labels": int(L), "arcs": int(len(graph[1])), "lambda": lam, "label_cost": h}
        runs = {}
        if key != "c5" or bk_c5:
            t0 = time.perf_counter()
            lab, e, cyc, cuts = O.expansion_bk(Dq, graph, lq, hq, z)
            runs["bk"] = (time.perf_counter() - t0, lab, cyc, cuts)
        t0 = time.perf_counter()
        lab, e, cyc = O.expansion(Dq, graph, lq, hq, z)
        runs["dinic"] = (time.perf_counter() - t0, lab, cyc, cyc * L)
        for name, (t, lab, cyc, cuts) in runs.items():
            rec[name + "_s"] = t
            rec[name + "_labels_equal_gpu"] = bool(np.array_equal(lab, gpu_labels))
        best = min(runs, key=lambda k2: runs[k2][0])
        t, lab, cyc, cuts = runs[best]
        rec.update(solver=best, seconds=t, cycles=int(cyc), mincuts=int(cuts), cycles_per_s=cyc / t, mincuts_per_s=cuts / t)
        if "bk" not in runs:
            rec["bk_s"] = None
            rec["bk_note"] = "not run by default (label-cost hubs: ~110 s, profiles/round5_cpu_labelling.txt); --cpu-labelling-bk-c5 runs it"
        out["configs"][key] = rec
    if "c5" in out["configs"]:      # the headline pair of the block: C5 is BASELINE's alpha-expansion config ("full spatial k-NN graph")
        out["cycles_per_s"] = out["configs"]["c5"]["cycles_per_s"]
        out["mincuts_per_s"] = out["configs"]["c5"]["mincuts_per_s"]
        out["sample"] = "one whole expansion from zeros at C5 size (2e5 sites, k-NN(8) graph, 7 labels), not a sub-sample"
    return out


def api_legs(datasets, c3=None, c5=None, c4=None):
    """Wall time of the drop-in calls: the five find* entry points on the BASELINE configs C1-C5 with
    the arguments of scripts/bench_api.py, the cap-lifted C4 call (16 objects), and the reference's bundled scenes with the
    notebooks' exact arguments (tests/golden/scenes: data files of the reference kept as fixtures; recorded = the wall time the
    reference's notebooks print, unstated CPU, one thread)."""
    import contextlib
    import io
    import pyprogressivex as px
    legs = {}
    px.findLines(np.random.default_rng(0).random((50, 2)) * 100, np.array(0), 100, 100, sampler_id=0, seed=0)   # the package's context

    def me(labels, K, gt):
        return float(datasets.misclassification(np.where(labels == K, 0, labels + 1), gt))

    def timed(key, rows, gt, fn, *a, **kw):
        try:
            with contextlib.redirect_stdout(io.StringIO()):        # find6DPoses prints its neighbourhood time (progressivex_python.cpp:109)
                t0 = time.perf_counter()
                models, labels = fn(*a, **kw)
                dt = time.perf_counter() - t0
            K = models.shape[0] // rows
            legs[key] = {"wall_s": dt, "models": int(K), "points": int(len(labels)), "misclassification": me(labels, K, gt)}
        except Exception as e:       # never fail the bench over a secondary leg
            legs[key] = {"error": str(e)}
    pts, gt, _ = datasets.make_lines(seed=0)
    timed("c1_findLines", 1, gt, px.findLines, pts, np.array(0), 1000, 1000, threshold=2.0, conf=0.99, sampler_id=0, seed=1, minimum_point_number=50)
    pts, gt, _ = datasets.make_homographies(seed=0)
    timed("c2_findHomographies", 3, gt, px.findHomographies, pts, 1000, 1000, 1000, 1000, threshold=3.0, conf=0.99, sampler_id=0, seed=1,
          minimum_point_number=50)
    pts, gt, _ = c3 if c3 is not None else datasets.make_two_view_motions(seed=0)
    timed("c3_findTwoViewMotions", 3, gt, px.findTwoViewMotions, pts, 1000, 1000, 1000, 1000, threshold=0.75, conf=0.99, sampler_id=0, seed=1,
          minimum_point_number=1000, max_iters=2000)
    pts, gt, _ = c5 if c5 is not None else datasets.make_vanishing_points(seed=0)
    timed("c5_findVanishingPoints", 1, gt, px.findVanishingPoints, pts, np.array(0), 1000, 1000, threshold=1.5, conf=0.99, sampler_id=0, seed=1,
          minimum_point_number=2000, spatial_coherence_weight=0.05, neighborhood_ball_radius=10.0)
    x1, x2, K, gt = c4 if c4 is not None else datasets.make_poses(seed=0)[:4]
    timed("c4_find6DPoses", 3, gt, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048)
    # BASELINE config C4 names 16 objects; the reference's outer loop stops at 10 proposals (progressive_x.h:272): the same call with
    # the cap lifted (keyword-only extension) returns all of them
    timed("c4_find6DPoses_cap_lifted", 3, gt, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048,
          max_outer_iterations=20)
    # ... and stops at 12-14: at 10^6 points the reference's score value - shared^2 (exponent 2 for this driver) turns negative for every
    # further object (scripts/c4_missing.py).  With the penalty's exponent at 1 (second keyword-only extension) all 16 come back.
    timed("c4_find6DPoses_16_objects", 3, gt, px.find6DPoses, x1, x2, K, seed=1, minimum_point_number=5000, max_iters=2048,
          max_outer_iterations=32, scoring_exponent=1)
    scenes = os.path.join(ROOT, "tests", "golden", "scenes")
    recorded = {"unionhouse": 0.030, "unihouse": 0.308, "oldclassicswing": 0.089, "breadcube": 0.737, "cubetoy": 0.514, "book": 0.582,
                "tless": 57.57}      # dataset_comparison/adelaideH.ipynb:137-142, adelaideF.ipynb:149-157, example_multi_pose_6d.ipynb
    out = {}
    for scene in ("unionhouse", "unihouse", "oldclassicswing", "breadcube", "cubetoy", "book", "tless"):
        try:
            ts = []
            for seed in range(3):
                with contextlib.redirect_stdout(io.StringIO()):
                    if scene == "tless":
                        M = np.loadtxt(os.path.join(scenes, "tless.txt"), skiprows=1)
                        Km = np.loadtxt(os.path.join(scenes, "tless_intrinsics.txt"))
                        t0 = time.perf_counter()
                        px.find6DPoses(M[:, :2], M[:, 2:5], Km, 4.0, seed=seed)
                    elif scene in ("unionhouse", "unihouse", "oldclassicswing"):
                        corrs, g = datasets.load_points_with_labels(os.path.join(scenes, f"{scene}.txt"))
                        t0 = time.perf_counter()
                        px.findHomographies(corrs, 1024, 768, 1024, 768, threshold=4.0, conf=0.5, spatial_coherence_weight=0.05,
                                            neighborhood_ball_radius=200.0, maximum_tanimoto_similarity=0.4, max_iters=1000,
                                            minimum_point_number=10, maximum_model_number=6, scoring_exponent=2, sampler_id=3, seed=seed)
                    else:
                        corrs, g = datasets.load_points_with_labels(os.path.join(scenes, f"{scene}.txt"))
                        c32 = corrs.astype(np.float32)      # the notebook's density ordering (sampler_id == 2 branch), outside the timed call
                        d = np.sqrt(((c32[:, None, :] - c32[None, :, :]) ** 2).sum(-1))
                        corrs = np.ascontiguousarray(corrs[np.argsort((d <= 50.0).sum(1))[::-1]])
                        t0 = time.perf_counter()
                        px.findTwoViewMotions(corrs, 1024, 768, 1024, 768, threshold=0.75, conf=0.5, spatial_coherence_weight=0.5,
                                              neighborhood_ball_radius=50.0, maximum_tanimoto_similarity=0.4, max_iters=10000,
                                              minimum_point_number=7, maximum_model_number=4, sampler_id=2, scoring_exponent=1.0, seed=seed)
                    ts.append(time.perf_counter() - t0)
            out[scene] = {"wall_s_median": float(np.median(ts)), "wall_s": [round(t, 4) for t in ts], "recorded_s": recorded[scene]}
        except Exception as e:
            out[scene] = {"error": str(e)}
    legs["bundled_scenes"] = out
    legs["note"] = ("wall time of the find* call alone (host marshalling, graph build, proposals, PEARL, read-backs), library already loaded; "
                    "bundled scenes: median of seeds 0-2, notebooks' arguments; recorded_s: what the reference's notebooks print (unstated CPU)")
    return legs


def timed_steps(ctx, step, steps, warmup):
    """(seconds per step by the wall clock, mean HIP-event ms of the scoring kernels) of `step` on one context.  The wall clock
    runs with the events off - two event records around every kernel cost ~30 us of a step - and five more steps with the
    events on give the per-kernel split."""
    ctx.score_profile(0)
    for _ in range(warmup):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    wall = (time.perf_counter() - t0) / steps
    ctx.score_profile(2)
    kt = []
    for _ in range(6):
        step()
        kt.append(ctx.score_kernel_times())
    return wall, np.mean(np.array(kt[1:]), axis=0)


def secondary_legs(_lib, datasets, parallel, ctx, pts, hyps, gt, T2, steps, warmup):
    """Bounded extra measurements on one GPU; every entry says what the step contained."""
    legs = {}
    n, M = pts.shape[0], hyps.shape[0]
    steps = max(5, min(steps, 20))
    buf = ctx.score_buffers()

    # (1) the headline batch, but uploaded (and locality-sorted) inside every step: what a proposal with host-made hypotheses pays
    def step_upload():
        ctx.score_upload(hyps)
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
        return parallel.select_best(res["scores"], res["counts"])
    s, kt = timed_steps(ctx, step_upload, steps, warmup)
    legs["upload_inclusive"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s, "kernel_ms": [float(x) for x in kt],
                                "step": "pgx_score_upload (196 KB of models + locality sort) + launch + fetch + select"}

    # (2) RANSAC-like batch generated INSIDE the step: random minimal samples -> device P3P (4 slots per sample) -> cull ->
    #     score -> fetch -> sequential selection.  The number SURVEY 8(f)1 calls end-to-end models/s.
    rng = np.random.default_rng(7)
    S = M // 4
    per_obj_min = int(min(np.count_nonzero(gt == 1 + k) for k in range(16)))
    inl = np.stack([np.nonzero(gt == 1 + k)[0][:per_obj_min] for k in range(16)]) if per_obj_min > 0 else None

    def draw():
        # half of the samples all-inlier (as a converging RANSAC sees them), half uniformly random; vectorised: the host's
        # share of a proposal is one RNG call per half
        smp = rng.integers(0, n, (S, 3))
        if inl is not None:
            rows = np.arange(0, S, 2)
            smp[rows] = inl[((rows // 2) % 16)[:, None], rng.integers(0, per_obj_min, (len(rows), 3))]
        return smp.astype(np.int32)

    def step_ransac():
        ctx.solve_minimal(draw(), fetch=False)
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
        return parallel.select_best(res["scores"], res["counts"])
    s, kt = timed_steps(ctx, step_ransac, steps, warmup)
    st = ctx.score_stats(T2, has_compound=True)
    legs["ransac_like_end_to_end"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s, "residual_evals_per_sec": n * M / s,
                                      "kernel_ms": [float(x) for x in kt], "work": st,
                                      "step": f"{S} minimal samples drawn on the host (half all-inlier, half random) -> "
                                              "pgx_solve_minimal (device P3P, 4 slots each) -> launch -> fetch -> select"}
    # (2b) the same step with the samples drawn ON THE DEVICE by the in-repo counter-based generator (csrc/rng.hip.h): no host RNG,
    #      no index upload.  Uniform samples (what gcransac's UniformSampler draws: a new batch number every step), so nearly all
    #      hypotheses are outlier-contaminated - a different batch from (2), listed for the step's host share, not for its kernel time
    state = {"b": 0}

    def step_sampled():
        state["b"] += 1
        ctx.solve_minimal_sampled(0x5EEDC0DE, state["b"], S, fetch=False)
        ctx.score_launch(T2, has_compound=True)
        res = ctx.score_fetch(exponent=2, out=buf)
        return parallel.select_best(res["scores"], res["counts"])
    try:
        s, kt = timed_steps(ctx, step_sampled, steps, warmup)
        legs["device_sampled_end_to_end"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s, "kernel_ms": [float(x) for x in kt],
                                             "step": f"pgx_solve_minimal_sampled: {S} uniform minimal samples drawn on the device (Philox4x32-10, key + batch "
                                                     "number) -> device P3P (4 slots each) -> launch -> fetch -> select; no host RNG, no index upload"}
    except Exception as e:
        legs["device_sampled_end_to_end"] = {"error": str(e)}
    ctx.score_upload(hyps)    # leave the metric batch resident

    # (3) one pgx_set_points (the once-per-problem preprocessing of the group-major path)
    t0 = time.perf_counter()
    ctx.set_points(_lib.PNP, pts)
    legs["set_points"] = {"ms": 1e3 * (time.perf_counter() - t0),
                          "what": "pgx_set_points, once per problem: upload of 40 MB + filter scales, f32 rows, Morton keys, "
                                  "radix sort, sorted / group-blocked copies and group bounds on the device (setpoints.hip)"}

    # (4) dense mode: no group test, no rejection filter - every pair through the exact FP64 path (true FP64 fraction)
    saved = {k: os.environ.get(k) for k in ("PGX_NO_GROUP", "PGX_NO_FILTER")}
    os.environ["PGX_NO_GROUP"], os.environ["PGX_NO_FILTER"] = "1", "1"
    try:
        dense = _lib.Context(ctx.device_id)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        dense.score_profile(True)
        dense.set_points(_lib.PNP, pts)
        dense.preference(gt_pose0, T2, slot=0)
        dense.compound_update([0])
        dense.score_upload(hyps)
        dbuf = dense.score_buffers()

        def step_dense():
            dense.score_launch(T2, has_compound=True)
            return dense.score_fetch(exponent=2, out=dbuf)
        s, kt = timed_steps(dense, step_dense, min(steps, 10), 2)
        tf = n * M * FLOPS_PER_PAIR["pnp"] / (kt[0] * 1e-3) / 1e12
        legs["dense_unfiltered"] = {"ms_per_step": 1e3 * s, "kernel_ms": float(kt[0]), "residual_evals_per_sec": n * M / s,
                                    "fp64_tflops": tf, "fp64_frac_of_no_fma_peak": tf / (FP64_VALU_PEAK_TFLOPS / 2),
                                    "step": "PGX_NO_GROUP=1 PGX_NO_FILTER=1: score_kernel<PnP, FILT=0>, all 2.05e9 pairs exact"}
    finally:
        dense.close()

    # (6) RCCL on the timed path with a single rank (the N > 1 runs are the driver's): launch + all-gather + fetch of the table
    try:
        cc = _lib.Context(ctx.device_id)
        try:
            cc.score_profile(True)
            cc.set_points(_lib.PNP, pts)
            cc.preference(gt_pose0, T2, slot=0)
            cc.compound_update([0])
            cc.score_upload(hyps)
            parallel.init_rccl(cc, 0, 1)

            def step_comm():
                cc.score_launch(T2, has_compound=True)
                cc.score_allgather()
                res = cc.score_fetch_all(exponent=2)
                return parallel.select_best(res["scores"], res["counts"])
            s, kt = timed_steps(cc, step_comm, steps, warmup)
            legs["rccl_single_rank_serial"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s, "kernel_ms": [float(x) for x in kt],
                                               "step": "launch + ncclAllGather of (count, value, shared) over a 1-rank communicator + fetch + select, one after the other"}
            # the same exchange overlapped: batch i is all-gathered and copied out on the exchange stream while batch i + 1 is scored
            state = {"i": 0}
            cc.score_launch(T2, has_compound=True)
            cc.score_allgather_begin(0)

            def step_pipe():
                i = state["i"] = state["i"] + 1
                cc.score_launch(T2, has_compound=True)
                cc.score_allgather_begin(i & 1)
                res = cc.score_allgather_end((i - 1) & 1, exponent=2)
                return parallel.select_best(res["scores"], res["counts"])
            cc.score_profile(0)
            for _ in range(warmup):
                step_pipe()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_pipe()
            cc.sync()
            s = (time.perf_counter() - t0) / steps
            cc.score_allgather_end(state["i"] & 1, exponent=2)
            legs["rccl_single_rank"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s,
                                        "step": "two batches in flight: launch(i) + pgx_score_allgather_begin(i) [copy aside, ncclAllGather and copy to pinned memory "
                                                "on the exchange stream] + pgx_score_allgather_end(i - 1) + select"}
            # the point-sharded exchange (bench.py --gpus N's default): integer accumulators exported, all-reduced (sum, uint64) and
            # converted on the exchange stream while the next batch is scored
            state["i"] = 0
            cc.score_launch(T2, has_compound=True)
            cc.score_allreduce_begin(0)

            def step_red():
                i = state["i"] = state["i"] + 1
                cc.score_launch(T2, has_compound=True)
                cc.score_allreduce_begin(i & 1)
                res = cc.score_allreduce_end((i - 1) & 1, exponent=2)
                return parallel.select_best(res["scores"], res["counts"])
            for _ in range(warmup):
                step_red()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_red()
            cc.sync()
            s = (time.perf_counter() - t0) / steps
            cc.score_allreduce_end(state["i"] & 1, exponent=2)
            legs["rccl_single_rank_allreduce"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s,
                                                  "step": "two batches in flight: launch(i) + pgx_score_allreduce_begin(i) [export of the integer accumulators, "
                                                          "ncclAllReduce(sum, uint64), conversion and copy to pinned memory on the exchange stream] + "
                                                          "pgx_score_allreduce_end(i - 1) + select"}
            # the RANSAC-like step (leg 2) with two batches in flight: the samples of batch i are drawn and solved and its
            # scoring launched BEFORE the table of batch i - 1 is taken and walked - the host's share (draw, fetch, select) overlaps the
            # device's.  Its own leg: the serial step stays the headline, and the drop-in calls cannot use it (one proposal's walk and
            # compound update precede the next proposal's samples: DESIGN.md, round-6 table)
            try:
                state["i"] = 0
                cc.solve_minimal(draw(), fetch=False)
                cc.score_launch(T2, has_compound=True)
                cc.score_allgather_begin(0)
                picks = []

                def step_pipe_ransac():
                    i = state["i"] = state["i"] + 1
                    cc.solve_minimal(draw(), fetch=False)
                    cc.score_launch(T2, has_compound=True)
                    cc.score_allgather_begin(i & 1)
                    res = cc.score_allgather_end((i - 1) & 1, exponent=2)
                    return parallel.select_best(res["scores"], res["counts"])
                for _ in range(warmup):
                    step_pipe_ransac()
                t0 = time.perf_counter()
                for _ in range(steps):
                    picks.append(step_pipe_ransac())
                cc.sync()
                s = (time.perf_counter() - t0) / steps
                cc.score_allgather_end(state["i"] & 1, exponent=2)
                legs["pipelined_end_to_end"] = {"ms_per_step": 1e3 * s, "models_per_sec": M / s, "pipelined": True,
                                                "step": f"the ransac_like_end_to_end step with two batches in flight on a 1-rank communicator: draw {S} samples + "
                                                        "pgx_solve_minimal + launch of batch i, then pgx_score_allgather_end + select of batch i - 1"}
            except Exception as e:
                legs["pipelined_end_to_end"] = {"error": str(e)}
            cc.comm_destroy()
        finally:
            cc.close()
    except Exception as e:       # never fail the bench over a secondary leg
        legs["rccl_single_rank"] = {"error": str(e)}

    # (5) Sampson scoring at C3 size and vanishing-point scoring at C5 size: 2048 hypotheses from the device solvers
    for name, mt, make, m, slots, thr in (("c3_sampson", _lib.FUNDAMENTAL, datasets.make_two_view_motions, 7, 3, 0.75),
                                          ("c5_vanishing_point", _lib.VANISHING_POINT, datasets.make_vanishing_points, 2, 1, 1.5)):
        p2, g2, models = make(seed=0)
        c2 = _lib.Context(ctx.device_id)
        try:
            c2.score_profile(True)
            c2.set_points(mt, p2)
            T2b = 2.25 * thr * thr
            c2.preference(np.asarray(models[0]).reshape(-1), T2b, slot=0)
            c2.compound_update([0])
            S2 = (M + slots - 1) // slots
            K2 = int(g2.max())
            smp = np.array([rng.choice(np.nonzero(g2 == 1 + r % K2)[0], m, replace=False) if r % 2 == 0 else
                            rng.choice(len(g2), m, replace=False) for r in range(S2)], dtype=np.int32)
            c2.solve_minimal(smp, fetch=False)
            b2 = c2.score_buffers()

            def step2():
                c2.score_launch(T2b, has_compound=True)
                return c2.score_fetch(exponent=2, out=b2)
            s, kt = timed_steps(c2, step2, min(steps, 50), min(warmup, 20))
            Mb = c2.M
            key = "fundamental" if mt == _lib.FUNDAMENTAL else "vanishing_point"
            tf = len(p2) * Mb * FLOPS_PER_PAIR[key] / (float(kt[0] + kt[1] + kt[3]) * 1e-3) / 1e12
            legs[name] = {"points": int(len(p2)), "hypotheses": int(Mb), "ms_per_step": 1e3 * s,
                          "kernel_ms": [float(x) for x in kt], "residual_evals_per_sec": len(p2) * Mb / s,
                          "models_per_sec": Mb / s, "work": c2.score_stats(T2b, has_compound=True),
                          "effective_fp64_tflops": tf,
                          "step": "resident batch from pgx_solve_minimal (half all-inlier, half random samples): launch + fetch"}
        finally:
            c2.close()
    return legs


CPU_LABELLING = {}      # config key -> (Dq [n, L] int64, (off, idx, mult), lam, h, labels the GPU ended with): inputs of cpu_labelling_baseline


def labelling_leg(_lib, name, mt, pts, models, thr, lam, h, graph_points, kind, radius, k, keep_for_cpu=None):
    """SURVEY 8(d) metric 3 at one config's (N, K, E): the PEARL labelling step (PEARL.h:476-555) = unary table + one full
    alpha-expansion from the all-zero labelling (what the first iteration of every PEARL::run does), on the neighbourhood
    graph built on the device.  expansion cycles/s and min-cuts/s are of that expansion; a PEARL iteration's labelling part
    is unary + expansion (its refits are host-size solves, not measured here)."""
    # The leg repeats ONE expansion from the all-zero labelling on the same models: pgx_expansion's first-cycle memo
    # would answer every repetition after the first from its kept labels - cached work inside a timed region.  The context of
    # this leg is created with the memo off (PGX_MF_MEMO is read once, in pgx_create): every repetition solves every min-cut.
    saved = os.environ.get("PGX_MF_MEMO")
    os.environ["PGX_MF_MEMO"] = "0"
    try:
        c = _lib.Context(0)
    finally:
        if saved is None:
            os.environ.pop("PGX_MF_MEMO", None)
        else:
            os.environ["PGX_MF_MEMO"] = saved
    try:
        c.set_points(mt, pts)
        t0 = time.perf_counter()
        arcs = c.graph_build(graph_points, kind, radius=radius, k=k, fetch=False)
        c.sync()
        t_graph = time.perf_counter() - t0
        c.pearl_unary(models, thr, lam)                      # (first call: allocations)
        c.sync()
        t0 = time.perf_counter()
        c.pearl_unary(models, thr, lam)
        c.sync()
        t_unary = time.perf_counter() - t0
        best = None
        for _ in range(3):
            c.set_labels(np.zeros(len(pts), np.int32))
            sched0 = c.expansion_schedule()
            t0 = time.perf_counter()
            eq, e, cycles = c.expansion(lam, h)
            t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        st = c.expansion_stats()
        sched = {k2: v2 - sched0[k2] for k2, v2 in c.expansion_schedule().items()}      # of the last repetition, like `st`
        paths = c.expansion_paths()
        if keep_for_cpu is not None:      # the same problem for the CPU solvers: the device-built graph and unary table (bit-identical to
            graph = c.graph_fetch()       # the oracle's, tests/test_fullsize_pins.py) and the labels to check them against
            CPU_LABELLING[keep_for_cpu] = (c.pearl_unary(models, thr, lam, want_table=True), graph, lam, h, c.get_labels())
        n_sites, n_arcs = int(len(pts)), int(arcs)
        # SURVEY 8(d) "expansion_step": a push-relabel sweep reads N (8 B excess + 4 B height) + E (4 B idx + 8 B cap); a level of the
        # global relabel reads N x 4 B of heights.  Only the level-synchronous solver counts sweeps / levels (one-workgroup and region
        # moves keep their state in LDS): for them the figure is the bytes of the moves that fell back to it.
        # nearly all sweeps are LIST sweeps that visit a few hundred sites, and charging each the whole graph made the
        # fraction meaningless.  A list sweep is charged by its list - sites visited (device counters, pgx_expansion_schedule) x (12 B +
        # the site's share of the arcs x 12 B) - and only the sweeps over all sites by the graph; `frac_survey_formula` keeps the
        # old figure for comparison with rounds 1-5.
        all_sweeps = int(st["sweeps"]) - int(st["list_sweeps"])
        list_sites = int(sched["list_sites"]) + int(sched["xcd_list_sites"])
        per_site = 12.0 + 12.0 * n_arcs / max(1, n_sites)
        alg_survey = (n_sites * 12 + n_arcs * 12) * int(st["sweeps"]) + n_sites * 4 * int(st["bfs_levels"])
        alg = int((n_sites * 12 + n_arcs * 12) * all_sweeps + list_sites * per_site + n_sites * 4 * int(st["bfs_levels"]))
        steps = int(st["sweeps"]) + int(st["bfs_levels"]) + int(st["global_relabels"])
        inside = int(sched["xcd_levels"]) + int(sched["xcd_sweeps"])
        traffic = PMC_LABELLING.get(keep_for_cpu or name[:2].lower())
        roof = {"bound": "hbm", "achieved": alg / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / best / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": alg, "algorithmic_bytes_survey_formula": alg_survey, "frac_survey_formula": alg_survey / best / 1e9 / HBM_PEAK_GBS,
                "formula": "(N 12 + E 12) x sweeps over all sites + list sites x (12 + 12 E / N) + N 4 x bfs_levels, over the expansion's wall time "
                           "(survey formula: every sweep charged the whole graph)",
                "all_site_sweeps": all_sweeps, "list_sweeps": int(st["list_sweeps"]), "list_sites_visited": list_sites,
                "dependent_steps": steps, "steps_inside_persistent_launches": inside,
                "launches_per_expansion": steps - inside + int(sched["xcd_round_launches"]) + int(sched["xcd_searches"]),
                "traffic": traffic["bytes"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                "moves_by_solver": {k2: int(v2) for k2, v2 in paths.items()},
                "note": "latency bound: every sweep / level is one dependent step (a launch of ~12 us at N = 1e6, or ~5 us inside a persistent "
                        "one-XCD launch on graphs <= 3e5 sites) that moves little; the fraction says how far from a streaming pass the solver is"}
        return {"config": name, "sites": int(len(pts)), "labels": int(len(models)) + 1, "arcs": int(arcs), "lambda": lam, "label_cost": h,
                "roofline_labelling": roof,
                "graph_build_ms": 1e3 * t_graph, "unary_ms": 1e3 * t_unary, "expansion_ms": 1e3 * best, "cycles": int(cycles), "energy": e,
                "expansion_cycles_per_sec": cycles / best, "mincuts_per_sec": st["mincuts"] / best,
                "pearl_labelling_iterations_per_sec": 1.0 / (t_unary + best),
                "ms_per_mincut": 1e3 * best / max(1, st["mincuts"]), **st,
                "memo_hits": int(c.expansion_paths()["memo"]),
                "note": "best of 3, first-cycle memo OFF (every repetition solves every min-cut); labels are those of the CPU oracle's Dinic solver (tests). <= 8192 sites: one workgroup, one launch per "
                        "move (maxflow_tile.hip); larger: level-synchronous push-relabel, latency bound (DESIGN.md 4.3)"}
    finally:
        c.close()


def labelling_legs(_lib, datasets, raw, pts, poses, thr):
    legs = {}
    def guard(key, fn):
        try:
            legs[key] = fn()
        except Exception as e:       # never fail the bench over a secondary leg
            legs[key] = {"error": str(e)}
    p2, _, m2 = datasets.make_homographies(seed=0)
    guard("labelling_c2", lambda: labelling_leg(_lib, "C2 homography 5k/5 planes", _lib.HOMOGRAPHY, p2, m2, 3.0, 0.05, 10.0, p2,
                                                _lib.GRAPH_KNN_IN_BALL, 200.0, 5))
    p3, _, m3 = datasets.make_two_view_motions(seed=0)
    guard("labelling_c3", lambda: labelling_leg(_lib, "C3 two-view 1e5/8 motions", _lib.FUNDAMENTAL, p3, m3, 0.75, 0.1, 14.0, p3,
                                                _lib.GRAPH_KNN_IN_BALL, 50.0, 5, keep_for_cpu="c3"))
    p5, _, m5 = datasets.make_vanishing_points(seed=0)
    guard("labelling_c5", lambda: labelling_leg(_lib, "C5 vanishing points 2e5/6, k-NN(8) on midpoints", _lib.VANISHING_POINT, p5, m5, 1.5, 0.1,
                                                20.0, 0.5 * (p5[:, :2] + p5[:, 2:]), _lib.GRAPH_KNN, 0.0, 8, keep_for_cpu="c5"))
    guard("labelling_c4", lambda: labelling_leg(_lib, "C4 6D pose 1e6/10 of 16 objects", _lib.PNP, pts, poses, thr, 0.1, 6.0, raw,
                                                _lib.GRAPH_KNN_IN_BALL, 20.0, 5))
    return legs


def strong_scaling_legs(ctx, parallel, hyps, T2, steps, warmup):
    """What 1/2/4/8 GPUs would each do under STRONG scaling of the 2048-hypothesis batch (BASELINE.json: '1e6 pts x 2048 hyps,
    1/2/4/8 GPU'), measured on this one GPU: the step at M = 2048 / 1024 / 512 / 256 hypotheses with its kernel split - the
    cull's per-group part, the row loads and the finish do not shrink with M, which is what bounds the efficiency."""
    out = {}
    base = None
    for div in (1, 2, 4, 8):
        sub = np.ascontiguousarray(np.array_split(hyps, div)[0])
        ctx.score_upload(sub)
        buf = ctx.score_buffers()

        def step():
            ctx.score_launch(T2, has_compound=True)
            res = ctx.score_fetch(exponent=2, out=buf)
            return parallel.select_best(res["scores"], res["counts"])
        s, kt = timed_steps(ctx, step, max(20, min(steps, 50)), max(10, min(warmup, 50)))
        base = s if base is None else base
        out[f"gpus_{div}"] = {"hypotheses_per_gpu": int(len(sub)), "ms_per_step": 1e3 * s,
                              "kernel_ms": {"cull": float(kt[0]), "group_major": float(kt[1]), "finish": float(kt[2])},
                              "projected_models_per_sec": len(hyps) / s,
                              "projected_efficiency": base / (div * s),
                              "note": "projection: every GPU does this step concurrently; the all-gather of 48 KB is not included"}
    ctx.score_upload(hyps)
    return out


def strong_points_legs(_lib, ctx, parallel, pts, hyps, gt0, T2, steps, warmup):
    """The same question for the POINT-sharded split (pgx_score_allreduce: every GPU scores all 2048 hypotheses against N / G of
    the points, integer accumulators all-reduced): the step at N = 1e6 / 5e5 / 2.5e5 / 1.25e5 points x 2048 hypotheses on this
    one GPU.  Cull, dispatch, group work and row loads all divide by G; what does not is the finish and the launch gaps."""
    out = {}
    base = None
    n = pts.shape[0]
    try:
        for div in (1, 2, 4, 8):
            lo, hi = parallel.point_slice(n, div, 0)
            ctx.set_points(_lib.PNP, pts[lo:hi])
            ctx.score_set_global_n(n)
            ctx.preference(gt0, T2, slot=0)
            ctx.compound_update([0])
            ctx.score_upload(hyps)
            buf = ctx.score_buffers()

            def step():
                ctx.score_launch(T2, has_compound=True)
                res = ctx.score_fetch(exponent=2, out=buf)
                return parallel.select_best(res["scores"], res["counts"])
            s, kt = timed_steps(ctx, step, max(20, min(steps, 50)), max(10, min(warmup, 50)))
            base = s if base is None else base
            out[f"gpus_{div}"] = {"points_per_gpu": int(hi - lo), "hypotheses_per_gpu": int(len(hyps)), "ms_per_step": 1e3 * s,
                                  "kernel_ms": {"cull": float(kt[0]), "group_major": float(kt[1]), "finish": float(kt[2])},
                                  "projected_models_per_sec": len(hyps) / s,
                                  "projected_efficiency": base / (div * s),
                                  "note": "projection: every GPU does this step concurrently on its slice; the all-reduce of 48 KB "
                                          "(overlapped with the next launch in bench.py --gpus N) is not included"}
    finally:
        ctx.score_set_global_n(0)
        ctx.set_points(_lib.PNP, pts)
        ctx.preference(gt0, T2, slot=0)
        ctx.compound_update([0])
        ctx.score_upload(hyps)
    return out


MAX_LINE_BYTES = 8000     # the driver stores an 8 188-byte tail of stdout: a longer last line does not parse (BENCH_r05: parsed = null)


def _r(x, sig=6):
    """numbers to `sig` significant digits (the compact line carries measurements, not prose)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    return x


def compact_line(out):
    """The LAST stdout line: the contract keys, `config`, `roofline` (numbers + traffic), `cpu_baseline` (scoring + the labelling
    numbers per config), a compact `roofline_labelling` and the drop-in calls' wall times.  Everything else of `out` (legs, notes,
    projections) travels in the BENCH_DETAIL line and gpurun_out/bench_detail.json."""
    c = {k: out[k] for k in ("metric", "value", "unit", "models_per_sec", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                             "scaling", "scaling_mode", "vs_baseline", "dtype", "data") if k in out}
    c["value_kind"] = "pairs covered/s (bounds decide most pairs; results = all-pairs exact); models_per_sec is the co-headline"
    cfg = dict(out["config"])
    cfg["workload"] = "C4 6D-pose points (1e6 2D-3D corr., 16 objects, 20% outliers) x 2048 pose hypotheses, PnP reprojection residual, MSAC + compound score"
    cfg["exchange"] = cfg["exchange"].split(" (")[0].split(",")[0]
    c["config"] = cfg
    c["winner"] = out["winner"]
    r = out["roofline"]
    alg = r["algorithmic_bytes_per_launch"]
    c["roofline"] = {k: _r(r[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_raw", "kernel", "kernel_ms",
                                           "kernel_ms_samples", "algorithmic_bytes_per_launch")}
    c["roofline"]["frac"] = c["roofline"]["achieved"] / c["roofline"]["peak"]      # frac = achieved / peak holds exactly on the printed numbers
    c["roofline"]["traffic_over_algorithmic"] = _r(r["traffic"] / alg, 4) if r["traffic"] else None
    c["roofline"]["traffic_raw_over_algorithmic"] = _r(r["traffic_raw"] / alg, 4) if r["traffic_raw"] else None
    c["roofline"]["traffic_source"] = f"{PMC['source']} @ {PMC.get('commit', '?')} (constant, not this run; raw = 2 FETCH + WRITE as r1-r4)"
    c["roofline"]["launch_kernels_ms"] = {k: _r(v, 4) for k, v in r["launch_kernels_ms"].items() if k != "note"}
    c["roofline"]["compute"] = {k: _r(v, 4) for k, v in r["compute"].items() if k != "source"}
    ex = out.get("executed", {})
    c["executed"] = {k: ex[k] for k in ("group_pairs", "surviving_group_steps", "exact_fp64_evaluations", "inlier_pairs") if k in ex}
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        cb = {k: _r(b[k]) for k in ("value", "unit", "cores", "kind", "models_per_sec", "host_cpu", "host_cores_available") if k in b}
        cb["sample"] = b["sample"].split(", oracle built")[0]
        if isinstance(b.get("all_cores"), dict) and "value" in b["all_cores"]:
            cb["all_cores_context"] = {"value": _r(b["all_cores"]["value"]), "cores": b["all_cores"]["cores"]}
        lab = b.get("labelling")
        if isinstance(lab, dict) and "configs" in lab:
            cl = {"kind": lab["kind"], "cores": lab["cores"], "what": "one expansion from zeros, faster of BK / Dinic; labels checked equal to the GPU's"}
            for key, rec in lab["configs"].items():
                cl[key] = {k: _r(rec.get(k), 4) for k in ("solver", "seconds", "bk_s", "dinic_s", "cycles", "mincuts", "gpu_expansion_s", "gpu_over_cpu")}
                cl[key]["labels_equal_gpu"] = bool(rec.get(rec.get("solver", "") + "_labels_equal_gpu"))
            cb["labelling"] = cl
        elif lab is not None:
            cb["labelling"] = lab
        c["cpu_baseline"] = cb
        c["speedup_vs_cpu_port"] = _r(out.get("speedup_vs_cpu_port"), 5)
        c["speedup_executed"] = _r(out.get("speedup_executed"), 5)
    legs = out.get("legs", {})
    rl = {}
    for key in ("c2", "c3", "c5", "c4"):
        g = legs.get("labelling_" + key)
        if isinstance(g, dict) and "roofline_labelling" in g:
            q = g["roofline_labelling"]
            rl[key] = {"frac": _r(q["frac"], 4), "achieved": _r(q["achieved"], 5), "bytes": q["algorithmic_bytes"],
                       "frac_survey_formula": _r(q.get("frac_survey_formula"), 4), "traffic": q.get("traffic"),
                       "steps": q.get("dependent_steps"), "launches": q["launches_per_expansion"], "list_sites": q.get("list_sites_visited"),
                       "expansion_ms": _r(g["expansion_ms"], 5), "mincuts": g.get("mincuts"),
                       "moves": {k: v for k, v in q.get("moves_by_solver", {}).items() if v}}
    if rl:
        rl["unit"] = "achieved GB/s of 8000; bytes: list sweeps charged by the sites they visit (frac_survey_formula: every sweep the whole graph); traffic: PMC bytes per expansion"
        c["roofline_labelling"] = rl
    api = legs.get("api")
    if isinstance(api, dict):
        ca = {}
        for k, v in api.items():
            if isinstance(v, dict) and "wall_s" in v:
                ca[k] = [_r(v["wall_s"], 4), v["models"], _r(v["misclassification"], 3)]
        sc = api.get("bundled_scenes", {})
        ca["scenes_wall_vs_recorded_s"] = {k: [_r(v.get("wall_s_median"), 3), v.get("recorded_s")] for k, v in sc.items() if isinstance(v, dict)}
        ca["fields"] = "[wall_s, models, misclassification]"
        c["api"] = ca
    for key in ("ransac_like_end_to_end", "pipelined_end_to_end", "c3_sampson", "c5_vanishing_point"):
        g = legs.get(key)
        if isinstance(g, dict) and "ms_per_step" in g:
            c.setdefault("legs_ms_per_step", {})[key] = _r(g["ms_per_step"], 4)
    c["detail"] = "BENCH_DETAIL line above + gpurun_out/bench_detail.json"
    line = json.dumps(c)
    if len(line) >= MAX_LINE_BYTES:       # never print a line the driver cannot parse: shed the optional blocks, largest first
        for key in ("api", "executed", "legs_ms_per_step", "roofline_labelling"):
            c.pop(key, None)
            if len(json.dumps(c)) < MAX_LINE_BYTES:
                break
    return c


gt_pose0 = None


def main():
    global gt_pose0
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)   # the first ~30 steps (10 ms) run 6 % slower: the clocks are still ramping
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--hyps", type=int, default=2048)
    ap.add_argument("--scaling", choices=("strong-points", "strong", "weak"), default="strong-points",
                    help="strong-points (default; the BASELINE metric '1e6 pts x 2048 hyps, 1/2/4/8 GPU': fixed total work): every GPU "
                         "scores all --hyps hypotheses against its 1/N slice of the points, the integer accumulators are summed by an "
                         "RCCL all-reduce (bitwise the 1-GPU table); strong: --hyps hypotheses in total, split over the GPUs, every "
                         "GPU holds all points (all-gather); weak: --hyps hypotheses per GPU, all points on every GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--cpu-labelling-bk-c5", action="store_true", help="also time Boykov-Kolmogorov on the C5 expansion (~110 s of CPU)")
    ap.add_argument("--stub", action="store_true", help="TEST HOOK (no GPU): take the context from the python file PGX_BENCH_STUB names; "
                                                        "without this flag the variable is ignored")
    args = ap.parse_args()

    from pyprogressivex import _lib, datasets, parallel
    rank, world, local = parallel.rank_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # ---- synthetic workload (identical points on every rank; per-rank hypothesis batches)
    n_obj = 16
    per_obj = args.points // 20
    x1, x2, K, gt_labels, gt = datasets.make_poses(n_per_object=per_obj, n_objects=n_obj,
                                                   n_outliers=args.points - n_obj * per_obj, seed=0)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    thr = 4.0 / f                       # find6DPoses default threshold 4 px (bindings.cpp:467), normalised (:96-98)
    T2 = 9.0 / 4.0 * thr * thr          # progressive_x.h:523
    by_points = args.scaling == "strong-points"
    n_total = pts.shape[0]
    if args.scaling == "weak":      # every rank scores its own batch of --hyps hypotheses
        hyps = datasets.make_pose_hypotheses(gt, M=args.hyps, seed=1 + rank)
    elif by_points:                 # ONE batch of --hyps hypotheses on every rank, rank r holds slice r of the POINTS
        hyps = datasets.make_pose_hypotheses(gt, M=args.hyps, seed=1)
    else:                           # strong: ONE batch of --hyps hypotheses, rank r scores slice r (BASELINE: 2048 in total)
        hyps = np.ascontiguousarray(np.array_split(datasets.make_pose_hypotheses(gt, M=args.hyps, seed=1), world)[rank])
    gt_pose0 = gt[0]

    # TEST HOOK (tests/test_bench_multirank.py, no GPU): PGX_BENCH_STUB names a python file whose StubContext has the _lib.Context
    # surface this function uses, answered by the CPU oracle with gloo collectives; only honoured together with --stub.  It exists to run THIS function's multi-rank
    # control flow and JSON arithmetic (n_gpus, scaling, parallelism, value) on a CPU box; the line then carries "data": "stub" and
    # is not a measurement.  Never set on the GPU box.
    stub = None
    if args.stub:      # (an environment variable alone must not make bench.main execute a file)
        if not os.environ.get("PGX_BENCH_STUB"):
            raise SystemExit("--stub needs PGX_BENCH_STUB=<python file with a StubContext>")
        import importlib.util
        spec = importlib.util.spec_from_file_location("pgx_bench_stub", os.environ["PGX_BENCH_STUB"])
        stub = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(stub)
    ctx = stub.StubContext(local) if stub else _lib.Context(local)
    info = ctx.device_info()
    ctx.score_profile(1)                # HIP events around the dominant scoring kernel, on the stream the kernels run on
    p_lo, p_hi = parallel.point_slice(n_total, world, rank) if by_points else (0, n_total)
    ctx.set_points(_lib.PNP, pts[p_lo:p_hi] if by_points and world > 1 else pts)
    if by_points:
        ctx.score_set_global_n(n_total)     # one fixed-point scale for the whole job: the all-reduced sums are bitwise the 1-GPU sums
    # a non-empty compound instance: the preference vector of the first accepted model (GT pose 0) - per point, so per slice
    ctx.preference(gt[0], T2, slot=0)
    ctx.compound_update([0])
    comp = ctx.get_compound() if rank == 0 else None
    ctx.score_upload(hyps)
    use_comm = world > 1 or os.environ.get("PGX_FORCE_COMM") == "1"   # PGX_FORCE_COMM: exercise RCCL with 1 rank
    if use_comm:
        (stub.init_comm if stub else parallel.init_rccl)(ctx, rank, world)

    fetch_buf = None if use_comm else ctx.score_buffers()   # reused every step (results are consumed before the next one)

    # HIP events around the dominant kernel on every EVENT_EVERY-th step of the timed region: a pair of event records costs
    # ~9 us of a 0.25 ms step, so timing every launch would put 3.5 % of measurement into the number being measured
    EVENT_EVERY = 5

    # with a communicator: two batches in flight - the exchange of step i - 1 overlaps the scoring of step i.  A pipelined exchange
    # runs on the exchange stream; every other collective (the barriers below) needs the slots collected first (comm.hip refuses
    # it otherwise: one communicator, two streams), so the pipeline is drained before a barrier and refilled by the next step.
    pipe = {"i": 0, "open": False, "res": None}
    begin = ctx.score_allreduce_begin if by_points else ctx.score_allgather_begin
    end = ctx.score_allreduce_end if by_points else ctx.score_allgather_end

    def drain():
        if pipe["open"]:
            pipe["res"] = end(pipe["i"] & 1, exponent=2)
            pipe["open"] = False
        return pipe["res"]

    def step(sample=True):
        ctx.score_profile(1 if sample else 0)
        ctx.score_launch(T2, has_compound=True)
        if use_comm:
            i = pipe["i"] = pipe["i"] + 1
            begin(i & 1)
            if pipe["open"]:
                pipe["res"] = end((i - 1) & 1, exponent=2)
            pipe["open"] = True
            res = pipe["res"]
        else:
            res = ctx.score_fetch(exponent=2, out=fetch_buf)
        kt = ctx.score_kernel_times() if sample else None    # (waits for the launch's last event)
        best = parallel.select_best(res["scores"], res["counts"]) if res is not None else -1
        return kt, best, res

    for i in range(args.warmup):
        step(i % EVENT_EVERY == 0)
    # ... and until the GPU has been busy for >= 50 ms whatever --warmup says: the first ~10 ms of launches run on ramping
    # clocks (6 % slower), which a driver-chosen `--warmup 5` would put inside the timed region
    extra_warmup = 0
    if not use_comm:            # (with a communicator every rank would have to agree on the count: the caller's warm-up stands)
        ctx.sync()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.05:
            step(False)
            extra_warmup += 1
    if use_comm:
        drain()
        ctx.comm_barrier()
    ctx.sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(args.steps):
        kms, best, res = step(i % EVENT_EVERY == 0)
        if kms is not None:
            kernel_ms.append(kms)
    if use_comm:
        res = drain()                 # inside the timed region: K launches, K exchanges
        best = parallel.select_best(res["scores"], res["counts"])
        ctx.comm_barrier()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    if use_comm:
        elapsed = ctx.comm_allreduce_max(elapsed)

    if rank == 0:
        n, M = n_total, hyps.shape[0]
        total_hyps = M * world if args.scaling == "weak" else args.hyps
        pairs_per_step = n * total_hyps
        ms_per_step = 1e3 * elapsed / args.steps
        alg_bytes, pairs = ctx.score_algorithmic_bytes()
        kt = np.mean(np.array(kernel_ms), axis=0)
        k_ms = float(kt[1]) if kt[1] > 0 else float(kt[0])      # the dominant kernel: group-major scoring (timed region)
        ctx.score_profile(2)            # the per-kernel breakdown: events around every kernel, outside the timed region

        def local_step():               # rank 0 alone from here on: no collective (the other ranks are at the final barrier)
            ctx.score_launch(T2, has_compound=True)
            ctx.score_fetch(exponent=2)
            return ctx.score_kernel_times()
        kt = np.mean(np.array([local_step() for _ in range(6)][1:]), axis=0)
        launch_ms = float(kt.sum())
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        work = ctx.score_stats(T2, has_compound=True)
        default_workload = n == 1000000 and M == 2048
        out = {
            "metric": "residual_evals_per_sec",
            "value": pairs_per_step / (elapsed / args.steps),
            "unit": "residual-evals/s",
            "value_kind": "effective: (point, hypothesis) pairs COVERED per second - counts, masks and scores are those of evaluating "
                          "every pair, but a group bound and an f32 filter decide most pairs without the FP64 residual (see executed)",
            "models_per_sec": total_hyps / (elapsed / args.steps),
            "co_headline": "models_per_sec (every hypothesis fully scored against all points): the figure that does not depend on how a pair is decided",
            "executed_pairs_per_sec": (work["surviving_group_steps"] * 64 + work["exact_evaluations"]) * world / (elapsed / args.steps),
            "executed_pairs_note": "f32 filter evaluations + exact FP64 evaluations actually run per second (rank 0's batch x ranks)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_extra_steps_by_time": extra_warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if args.scaling == "weak" else "strong", "scaling_mode": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "stub" if stub else "synthetic",
            "config": {"workload": "C4 multi-6D-pose points (1e6 2D-3D correspondences, 16 objects, 20% outliers) x "
                                   "metric batch of 2048 pose hypotheses (16 GT + perturbed; weak mode: per GPU), PnP reprojection "
                                   "residual, MSAC + compound-model score, compound instance = 1 model",
                       "points": n, "points_per_gpu": int(p_hi - p_lo), "hypotheses_per_gpu": M, "hypotheses_total": total_hyps,
                       "parallelism": (f"point-sharded x{world}" if by_points else f"hypothesis-sharded x{world}") if world > 1 else "1 GPU",
                       "exchange": ("none" if not use_comm else
                                    "rccl all-reduce (sum, uint64) of the integer accumulators (count, 2^-q fixed-point value and shared), overlapped with the next launch (two in flight)"
                                    if by_points else "rccl all-gather of (count,value,shared), overlapped with the next batch's scoring (two in flight)"),
                       "device": info["name"], "cu_count": info["cu_count"]},
            "winner": {"index": best, "inliers": int(res["counts"][best]) if best >= 0 else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": PMC_TRAFFIC_DEFAULT if default_workload else None,
                         "traffic_raw": PMC_TRAFFIC_RAW if default_workload else None,
                         "traffic_source": f"NOT measured in this run: constant from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                           f"{PMC['source']} (taken at commit {PMC.get('commit', '5d18ef3')}); traffic = 2 x FETCH + WRITE - the group "
                                           f"kernel's WRITE ({PMC.get('atomic_write_kib', 0.0):.0f} KiB: L2 atomics tallied at 32 B each, no HBM bytes), "
                                           "factors calibrated on known-bytes kernels in the same access patterns (profiles/round5_fetch_calibration.txt); "
                                           "traffic_raw = 2 x FETCH + WRITE as rounds 1-4 quoted it",
                         "kernel": "pgx::score_group_kernel<PnP>", "kernel_ms": k_ms,
                         "kernel_ms_samples": len(kernel_ms),
                         "kernel_ms_how": f"HIP events around the kernel on every {EVENT_EVERY}th launch of the timed region, on the context's stream",
                         "launch_kernels_ms": {"cull": float(kt[0]), "group_major": float(kt[1]), "finish": float(kt[2]),
                                               "exact_queue": float(kt[3]), "sum": launch_ms,
                                               "note": "breakdown from 5 extra steps with events around every kernel (an event costs "
                                                       "~5 us on the stream); kernel_ms above is from the timed region"},
                         # what actually bounds the kernel travels with `frac`: executed arithmetic per second of
                         # the dominant kernel and the busy fractions of the two units its filter loop is co-bound by
                         "compute": {"fp64_exact_tflops": work["exact_evaluations"] * FLOPS_PER_PAIR["pnp"] / (k_ms * 1e-3) / 1e12,
                                     "f32_filter_tflops": work["surviving_group_steps"] * 64 * F32_FLOPS_PER_FILTER_TEST / (k_ms * 1e-3) / 1e12,
                                     "valu_busy": PMC.get("valu_busy_frac") if default_workload else None,
                                     "lds_busy": PMC.get("lds_busy_frac") if default_workload else None,
                                     "fp64_vector_peak_tflops_no_fma": 39.3,
                                     "source": f"work counters of an untimed launch (pgx_score_stats) over the timed kernel duration; busy fractions from the "
                                               f"SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_LDS passes of {PMC['source']}"},
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_formula": "N d 8 + M p 8 + M 16 + N 8 (compound), SURVEY 8(d); no derived copies",
                         "note": "arithmetic/latency bound by construction (~0.02 algorithmic B/pair): the fraction is small "
                                 "whatever the kernel does; see `executed` for the work actually done"},
            "executed": {"pairs": work["pairs"], "group_pairs": work["group_pairs"],
                         "surviving_group_steps": work["surviving_group_steps"],
                         "f32_filter_evaluations": work["surviving_group_steps"] * 64,
                         "exact_fp64_evaluations": work["exact_evaluations"], "inlier_pairs": work["inlier_pairs"],
                         "exact_fp64_tflops": work["exact_evaluations"] * FLOPS_PER_PAIR["pnp"] / (k_ms * 1e-3) / 1e12,
                         "valu_busy_frac": PMC.get("valu_busy_frac") if default_workload else None,
                         "valu_busy_source": f"SQ_ACTIVE_INST_VALU pass, {PMC['source']}",
                         "source": "device counters of an untimed launch of the same kernels (pgx_score_stats)",
                         "effective_fp64_tflops_if_every_pair_were_exact": pairs * FLOPS_PER_PAIR["pnp"] / (k_ms * 1e-3) / 1e12,
                         "note": "'effective' is NOT a utilisation: a bound test culls (hypothesis, 64-point group) pairs, an f32 "
                                 "filter the rest; only exact_fp64_evaluations run the reference's FP64 residual"},
        }
        if world == 1 and not args.no_legs and default_workload:
            out["legs"] = secondary_legs(_lib, datasets, parallel, ctx, pts, hyps, gt_labels, T2, args.steps, args.warmup)
            out["legs"].update(labelling_legs(_lib, datasets, np.column_stack([x1, x2]), pts, gt[:10], thr))
            try:
                out["legs"]["api"] = api_legs(datasets, c4=(x1, x2, K, gt_labels))
            except Exception as e:
                out["legs"]["api"] = {"error": str(e)}
            for key in ("labelling_c2", "labelling_c3", "labelling_c5", "labelling_c4"):      # next to the contract's `roofline`
                if isinstance(out["legs"].get(key), dict) and "roofline_labelling" in out["legs"][key]:
                    out.setdefault("roofline_labelling", {})[key[10:]] = out["legs"][key]["roofline_labelling"]
            try:
                out["legs"]["strong_scaling_projection"] = strong_scaling_legs(ctx, parallel, hyps, T2, args.steps, args.warmup)
            except Exception as e:       # never fail the bench over a secondary leg
                out["legs"]["strong_scaling_projection"] = {"error": str(e)}
            try:
                out["legs"]["strong_points_projection"] = strong_points_legs(_lib, ctx, parallel, pts, hyps, gt[0], T2, args.steps, args.warmup)
            except Exception as e:
                out["legs"]["strong_points_projection"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(pts, hyps, T2, comp)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_port"] = out["models_per_sec"] / cb["models_per_sec"]
            out["speedup_note"] = ("models/s over models/s: both sides score every hypothesis against all points with identical results; "
                                   "the CPU port evaluates every pair exactly, the GPU path decides most pairs by bounds - "
                                   "pair-for-pair on EXECUTED evaluations the ratio is speedup_executed")
            out["speedup_executed"] = out["executed_pairs_per_sec"] / cb["value"]
            if CPU_LABELLING:
                try:
                    cb["labelling"] = cpu_labelling_baseline(bk_c5=args.cpu_labelling_bk_c5)
                    for key, rec in cb["labelling"]["configs"].items():      # the GPU's expansion of the same problem next to it
                        g = out.get("legs", {}).get("labelling_" + key, {})
                        if "expansion_ms" in g:
                            rec["gpu_expansion_s"] = g["expansion_ms"] * 1e-3
                            rec["gpu_over_cpu"] = rec["seconds"] / (g["expansion_ms"] * 1e-3)
                except Exception as e:
                    cb["labelling"] = {"error": str(e)}
        line = json.dumps(compact_line(out))
        detail = json.dumps(out)
    else:
        line = detail = None
    if use_comm:
        ctx.comm_barrier()
        ctx.comm_destroy()
    ctx.close()
    if line is not None:
        # libraries (RCCL's version banner) write to the C stdio buffer, which is flushed at exit, i.e. AFTER python's own
        # buffer: flush it now so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        # The driver keeps an 8 KB tail of stdout and parses the LAST line: that line is the compact record (contract keys, config,
        # roofline, cpu_baseline, roofline_labelling, api wall times; < 8 000 bytes, asserted by tests/test_bench_contract.py).  The
        # full record (every leg, notes, projections) goes to an EARLIER line, prefixed so that it is not mistaken for the line,
        # and to gpurun_out/bench_detail.json.
        print("BENCH_DETAIL " + detail, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
                f.write(detail + "\n")
        except OSError:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
