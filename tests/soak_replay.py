"""Replay soak (tests/test_gpu_fuzz.py runs a bounded slice): the five drop-in calls on random small problems with random arguments
(the generator of tests/soak_api.py), each run ONCE through libpgx.so with the decision trace recorded, then replayed by the
independent restatement of progressive_x.h / PEARL.h (oracle/progx_replay.c): every accept / reject, every PEARL iteration
(energy, refit acceptance, rejections, convergence), every break reason, the final labels and the model count must agree.
PEARL.h:393's comparison of two ~n-term sums is followed, not decided, where the sums agree to 1e-12 (tree vs sequential order);
the number of such ties is reported.  usage: python tests/soak_replay.py <seed> <trials> [--oracle]"""
import contextlib
import io
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "progressive-x_amd"), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]
import numpy as np
import pyprogressivex as px
from pyprogressivex import _api, datasets
import progx_replay as R
import progx_proposal as Q
import replay_helpers as H


def problem(rng, trial):
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
              sampler_rng=str(rng.choice(["numpy", "philox"])),
              pearl_abs=str(rng.choice(["double", "double", "int"])), refit_solver=str(rng.choice(["lapack", "lapack", "jacobi"])),
              max_outer_iterations=int(rng.choice([10, 10, 14])))
    K = int(rng.integers(1, 5))
    per = int(rng.choice([40, 150, 400, 1000]))
    nout = int(rng.choice([0, 50, 400]))
    if which == 0:
        pts, gt, _ = datasets.make_lines(n_per_line=per, n_lines=K, n_outliers=nout, seed=s)
        fn, args, rows = px.findLines, (pts, np.array(0), 1000, 1000), 1
        kw.update(threshold=float(rng.choice([1.0, 2.0, 4.0])), sampler_id=int(rng.choice([0, 0, 1, 2, 2, 3])),
                  minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([20.0, 60.0, 200.0])))
    elif which == 1:
        pts, gt, _ = datasets.make_homographies(n_per_plane=per, n_planes=K, n_outliers=nout, seed=s)
        fn, args, rows = px.findHomographies, (pts, 1000, 1000, 1000, 1000), 3
        kw.update(threshold=float(rng.choice([1.0, 3.0, 6.0])), sampler_id=int(rng.choice([0, 0, 1, 2, 2, 3])),
                  minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([60.0, 200.0])),
                  scoring_exponent=int(rng.choice([1, 2, 3])))
    elif which == 2:
        pts, gt, _ = datasets.make_two_view_motions(n_per_motion=per, n_motions=min(K, 3), n_outliers=nout, seed=s)
        fn, args, rows = px.findTwoViewMotions, (pts, 1000, 1000, 1000, 1000), 3
        kw.update(threshold=float(rng.choice([0.5, 0.75, 2.0])), sampler_id=int(rng.choice([0, 0, 1, 2, 2, 3])),
                  minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([50.0, 200.0])))
    elif which == 3:
        pts, gt, _ = datasets.make_vanishing_points(n_inliers=per * K, n_vps=K, n_outliers=nout, seed=s)
        fn, args, rows = px.findVanishingPoints, (pts, np.array(0) if rng.random() < 0.5 else rng.random(len(pts)), 1000, 1000), 1
        kw.update(threshold=float(rng.choice([0.5, 1.5, 3.0])), sampler_id=int(rng.choice([0, 1, 3])),
                  minimum_point_number=int(rng.choice([10, 30])), neighborhood_ball_radius=float(rng.choice([15.0, 100.0])))
    else:
        x1, x2, Kc, gt, _ = datasets.make_poses(n_per_object=per, n_objects=min(K, 3), n_outliers=nout, seed=s)
        fn, args, rows = px.find6DPoses, (x1, x2, Kc), 3
        kw.update(threshold=float(rng.choice([2.0, 4.0, 8.0])), minimum_point_number=int(rng.choice([6, 30])),
                  neighborhood_ball_radius=float(rng.choice([20.0, 60.0])))
    return fn, args, kw, rows, s


def soak(seed, trials, verbose=True, tie=1e-12):
    rng = np.random.default_rng(seed)
    bad = ties = events = pearl = walks = walk_events = 0
    t0 = time.time()
    for trial in range(trials):
        fn, args, kw, rows, s = problem(rng, trial)
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                out, rec, rep = H.run_and_replay(fn, *args, refit_tie_rtol=tie, **kw)
        except R.ReplayError as e:
            bad += 1
            print("MISMATCH trial", trial, fn.__name__, "data seed", s, kw, "-", e, flush=True)
            continue
        if rep is None:           # unknown sampler id: zero models, no run (progressivex_python.cpp:240-245)
            continue
        diff = R.compare_events(rec.events, rep["events"])
        ok = diff is None and np.array_equal(np.asarray(out[1], dtype=np.int64), rep["labels"]) and \
            out[0].shape[0] // rows == rep["models"].shape[0] and rep["consumed"] == (len(rec.proposals), len(rec.refits))
        ties += rep["ties"]
        events += len(rec.events)
        pearl += sum(e[0] == R.EV_PEARL_ITER for e in rec.events)
        if not ok:
            bad += 1
            print("MISMATCH trial", trial, fn.__name__, "data seed", s, kw, "-", diff, flush=True)
        # ... and every proposal of the call against the independent restatement of the proposal loop (oracle/progx_proposal.c)
        for k, w in enumerate(rec.walks):
            wd = Q.compare(w)
            walks += 1
            walk_events += len(w["events"])
            if wd is not None:
                bad += 1
                print("MISMATCH trial", trial, fn.__name__, "data seed", s, kw, "- proposal", k, wd, flush=True)
    if verbose:
        print(f"replay soak done: seed {seed}, {trials} calls, {events} decision events ({pearl} PEARL iterations), {ties} summation-order ties "
              f"followed; {walks} proposals ({walk_events} proposal-loop decisions) against the proposal replay; {bad} mismatches, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    if "--oracle" in sys.argv:
        from oracle_ctx import OracleContext
        _api._ctx = OracleContext()
    soak(int(sys.argv[1]), int(sys.argv[2]), tie=0.0 if "--oracle" in sys.argv else 1e-12)
