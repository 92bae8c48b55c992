#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: batched hypothesis scoring, 1e6 2D-3D correspondences x 2048 pose
hypotheses per GPU (config C4 points + the metric batch of SURVEY.md §8d), through the C ABI of libpgx.so.

One step = one pass of the proposal hot path over one batch: score every hypothesis of the batch against all points
(MSAC + compound-model score, scoring_function_with_compound_model.h:61-125), exchange the per-hypothesis results
(RCCL all-gather when N > 1), fetch them and select the winner on the host.  Inputs (points, compound preference
vector, hypotheses) are resident in HBM before the timed region starts.  Weak scaling: every rank scores its own 2048
hypotheses against all points, value = (points x hypotheses of all ranks) / time.

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

# HBM bytes per launch of the default workload from the committed PMC passes (profiles/round1_bench_v12_final.txt),
# summed over the three kernels of one launch (cull, group-major score, finish): FETCH_SIZE 267746.8 + 3439.0 + 36.5 KiB,
# x 2 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md §HBM), + WRITE_SIZE 27527.2 + 4276.5 + 158.8 KiB
# (writes: survivor bit masks of the cull pass, accumulator atomics; fetch: the 8 waves that share a 64-point group run
# on the 8 XCDs -- wave p of every group on XCD p, which keeps each XCD's hypotheses, their constants and their accumulator
# atomics in its own L2 -- so every XCD's L2 fetches all rows once: 8 x 80 MB.  Co-locating a group's waves cuts the fetch
# 8x but every XCD then updates every accumulator: measured 0.31 instead of 0.28 ms)
PMC_TRAFFIC_DEFAULT = int((2 * (267746.8 + 3439.0 + 36.5) + 27527.2 + 4276.5 + 158.8) * 1024)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64 counting an FMA as 2 flops; parity mode may not contract => 39.3 usable
FLOPS_PER_PAIR_PNP = 25        # 9 mul + 9 add (3x4 projection) + 2 div + 2 sub + 2 mul + 1 add


def cpu_baseline(pts, hyps, T2, comp, budget_s=15.0):
    """The oracle (a single-threaded C restatement of the reference path, kind = 'port') on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pgx_oracle as O
    O.lib()
    t0 = time.perf_counter()
    O.score(O.PNP, pts, hyps[:8], T2, compound=comp, has_compound=True, exponent=2)
    per_hyp = (time.perf_counter() - t0) / 8
    m = int(max(8, min(hyps.shape[0], budget_s / max(per_hyp, 1e-9))))
    t0 = time.perf_counter()
    O.score(O.PNP, pts, hyps[:m], T2, compound=comp, has_compound=True, exponent=2)
    dt = time.perf_counter() - t0
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
           "sample": f"{m} of {hyps.shape[0]} hypotheses x all {pts.shape[0]} points, {dt:.1f} s, 1 thread",
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--hyps", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from pyprogressivex import _lib, datasets, parallel
    rank, world, local = parallel.rank_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # ---- synthetic workload (identical points on every rank; per-rank hypothesis batches)
    n_obj = 16
    per_obj = args.points // 20
    x1, x2, K, _, gt = datasets.make_poses(n_per_object=per_obj, n_objects=n_obj,
                                           n_outliers=args.points - n_obj * per_obj, seed=0)
    pts, f = datasets.normalize_pnp(x1, x2, K)
    thr = 4.0 / f                       # find6DPoses default threshold 4 px (bindings.cpp:467), normalised (:96-98)
    T2 = 9.0 / 4.0 * thr * thr          # progressive_x.h:523
    hyps = datasets.make_pose_hypotheses(gt, M=args.hyps, seed=1 + rank)

    ctx = _lib.Context(local)
    info = ctx.device_info()
    ctx.set_points(_lib.PNP, pts)
    # a non-empty compound instance: the preference vector of the first accepted model (GT pose 0)
    ctx.preference(gt[0], T2, slot=0)
    ctx.compound_update([0])
    comp = ctx.get_compound() if rank == 0 else None
    ctx.score_upload(hyps)
    use_comm = world > 1 or os.environ.get("PGX_FORCE_COMM") == "1"   # PGX_FORCE_COMM: exercise RCCL with 1 rank
    if use_comm:
        parallel.init_rccl(ctx, rank, world)

    fetch_buf = None if use_comm else ctx.score_buffers()   # reused every step (results are consumed before the next one)

    def step():
        ctx.timer_start()
        ctx.score_launch(T2, has_compound=True)
        ctx.timer_mark()                 # HIP events on the stream the kernels run on; no host wait here
        if use_comm:
            ctx.score_allgather()
            res = ctx.score_fetch_all(exponent=2)
        else:
            res = ctx.score_fetch(exponent=2, out=fetch_buf)
        kernel_ms = ctx.timer_elapsed()  # the fetch synchronised the stream: the events are complete
        best = parallel.select_best(res["scores"], res["counts"])
        return kernel_ms, best, res

    for _ in range(args.warmup):
        step()
    if use_comm:
        ctx.comm_barrier()
    ctx.sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kms, best, res = step()
        kernel_ms.append(kms)
    if use_comm:
        ctx.comm_barrier()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    if use_comm:
        elapsed = ctx.comm_allreduce_max(elapsed)

    if rank == 0:
        n, M = pts.shape[0], hyps.shape[0]
        pairs_per_step = n * M * world
        ms_per_step = 1e3 * elapsed / args.steps
        alg_bytes, pairs = ctx.score_algorithmic_bytes()
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "residual_evals_per_sec",
            "value": pairs_per_step / (elapsed / args.steps),
            "unit": "residual-evals/s",
            "models_per_sec": M * world / (elapsed / args.steps),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 multi-6D-pose points (1e6 2D-3D correspondences, 16 objects, 20% outliers) x "
                                   "metric batch of 2048 pose hypotheses per GPU (16 GT + perturbed), PnP reprojection "
                                   "residual, MSAC + compound-model score, compound instance = 1 model",
                       "points": n, "hypotheses_per_gpu": M, "parallelism": f"hypothesis-sharded x{world}",
                       "exchange": "rccl all-gather of (count,value,shared)" if use_comm else "none",
                       "device": info["name"], "cu_count": info["cu_count"]},
            "winner": {"index": best, "inliers": int(res["counts"][best]) if best >= 0 else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": PMC_TRAFFIC_DEFAULT if (n == 1000000 and M == 2048) else None,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/round1_bench_v12_final.txt",
                         "kernel": "pgx::score_group_kernel<PnP> (+ score_cull_kernel, score_finish_kernel)", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "arithmetic bound by construction (~0.04 algorithmic B/pair); 93 % of the (hypothesis, 64-point group) pairs are culled by a bound test, see valu_fp64"},
            "valu_fp64": {"effective_tflops": pairs * FLOPS_PER_PAIR_PNP / (k_ms * 1e-3) / 1e12,
                          "peak_tflops_fma": FP64_VALU_PEAK_TFLOPS, "peak_tflops_no_fma": FP64_VALU_PEAK_TFLOPS / 2,
                          "effective_over_no_fma_peak": pairs * FLOPS_PER_PAIR_PNP / (k_ms * 1e-3) / 1e12 / (FP64_VALU_PEAK_TFLOPS / 2),
                          "flops_per_pair": FLOPS_PER_PAIR_PNP, "pairs_per_launch": pairs,
                          "note": "effective = what evaluating every pair exactly would cost; the kernels evaluate ~7 % of the "
                                  "pairs with the f32 pre-filter and ~0.3 % exactly (DESIGN.md 5.2c), so the ratio may exceed 1"},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(pts, hyps, T2, comp)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_port"] = out["value"] / cb["value"]
        line = json.dumps(out)
    else:
        line = None
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
        print(line, flush=True)


if __name__ == "__main__":
    main()
