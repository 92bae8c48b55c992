"""Host restatement of the Progressive-X outer loop and of PEARL, driving libpgx for every O(N) step.

ProgressiveX.run  <- progx::ProgressiveX<...>::run / initialize / isPutativeModelValid / updateCompoundModel /
                     getPredictedUnseenInliers   (/root/reference/src/pyprogressivex/include/progressive_x.h:251-624)
Pearl.run         <- pearl::PEARL<...>::run / labeling / parameterEstimation / rejectInstances / getLabeling
                     (/root/reference/src/pyprogressivex/include/PEARL.h:218-555)

The control flow — including the reference's quirks that SURVEY.md §3.2 lists — is reproduced literally; all N-length
work (preference vectors, Tanimoto reductions, compound max, unary costs, alpha-expansion, label bucketing, residual
sums) runs in HIP kernels behind the C ABI.  Model refits (3x3 / 9x9 solves) stay on the host in round 1.
"""
import math
import sys
import time

import numpy as np

from . import _proposal


# Decision events of the two loops below, handed to the optional `trace=` hook (diagnostics; the reference's analogue is its
# do_logging narration).  The hook is an object with ALL of
#     .proposal(model_or_None, inliers_or_None, iterations)      what the outer loop took from the proposal engine
#     .refit(inlier_number, fits, accepted)                      what PEARL took from the refit solver for one instance, and whether it kept it
#     .event(code, a=0, b=0, c=0, x=0.0, y=0.0)                  a decision (codes below)
# and optionally .begin(info) (the drop-in API describes the run: points, graph, settings) and .walk(record) (one record per proposal from
# the proposal engine: _proposal.WK_* events + what its walk saw; replayed by oracle/progx_proposal.c).  It receives every input the loops take
# from the proposal engine / the refit solver and every decision they make; tests/ compares the stream with an independent replay of
# progressive_x.h / PEARL.h (oracle/progx_replay.c, whose header documents the fields).  check_trace_hook() refuses an object that
# lacks one of the three required methods when the loops are built, not in the middle of a run.
(EV_OUTER, EV_PROPOSAL_EMPTY, EV_PROPOSAL, EV_VALIDATION, EV_UNACCEPTED, EV_SINGLE_MODEL, EV_PEARL_ITER, EV_REFIT_SKIP, EV_REFIT,
 EV_REJECT, EV_PEARL_END, EV_LABELING, EV_COMPOUND, EV_UNSEEN, EV_BREAK) = range(1, 16)


def check_trace_hook(trace):
    """None, or an object with callable .proposal / .refit / .event (see above); .begin is optional."""
    if trace is None:
        return None
    missing = [name for name in ("proposal", "refit", "event") if not callable(getattr(trace, name, None))]
    if missing:
        raise TypeError("trace hook lacks " + ", ".join("." + m for m in missing) +
                        ": it needs .proposal(model, inliers, iterations), .refit(inlier_number, fits, accepted) and .event(code, a, b, c, x, y)")
    return trace


class MultiModelSettings:
    """progx::MultiModelSettings (progressive_x.h:32-73) with the defaults of its constructor."""

    def __init__(self):
        self.minimum_number_of_inliers = 20
        self.max_proposal_number_without_change = 10
        self.maximum_model_number = sys.maxsize
        self.maximum_tanimoto_similarity = 0.5
        self.confidence = 0.95
        self.one_minus_confidence = 0.05
        self.inlier_outlier_threshold = 2.0
        self.spatial_coherence_weight = 0.14
        self.point_weights = None
        # proposal_engine_settings (:66-71)
        self.max_iteration_number = 5000
        self.max_local_optimization_number = 50
        self.max_graph_cut_number = 10        # gcransac::utils::Settings defaults [UPSTREAM-MEMORY]: graph cuts per proposal,
        self.min_iteration_number = 20        # iterations the main loop always runs,
        self.min_iteration_number_before_lo = 20   # iterations before the first local optimisation,
        self.max_least_squares_iterations = 10     # refits of the final iterated least squares
        self.labeling_l0 = "greedy"           # [U-8] lambda = 0: GCO-v3's special-case solver ("expansion": alpha-expansion)
        self.lo_cadence = "every_best"        # "winner": round 1's stand-in (one local optimisation, on the batch winner)
        self.local_optimization = "auto"      # "auto": graph-cut LO when 0 < lambda < 1, else LSQ refits; "lsq": always LSQ
        # not in the reference: its outer loop is hard-capped at 10 proposals (progressive_x.h:272)
        self.max_outer_iterations = 10
        # [U-16] PEARL.h:465 calls an unqualified abs(energy - previous_energy) behind #include <math.h>.  With libstdc++ / MSVC
        # headers the double overload is visible in the global namespace ("double", the default); a toolchain that resolves it
        # to int abs(int) truncates the difference first, so convergence fires whenever |dE| < 1 ("int").
        self.pearl_abs = "double"

    def set_confidence(self, c):  # :49-53
        self.confidence = c
        self.one_minus_confidence = 1.0 - c


class Statistics:
    def __init__(self):
        self.processing_time = 0.0
        self.total_time_of_proposal_engine = 0.0
        self.total_time_of_model_validation = 0.0
        self.total_time_of_optimization = 0.0
        self.total_time_of_compound_model_calculation = 0.0
        self.iteration_statistics = []
        self.inliers_of_each_model = []
        self.labeling = None
        self.pearl_iterations = 0
        self.expansion_cycles = 0


class Model:
    """progx::Model (progx_model.h:43-99): descriptor + the device slot holding its (stale) preference vector."""
    __slots__ = ("descriptor", "slot")

    def __init__(self, descriptor, slot=-1):
        self.descriptor = np.asarray(descriptor, dtype=np.float64).copy()
        self.slot = slot


def predicted_unseen_inliers(one_minus_confidence, sample_size, iteration_number, covered, point_number):
    """progressive_x.h:495-513 (size_t arithmetic: point_number - covered wraps if covered > point_number)."""
    unseen = (point_number - covered) % (1 << 64)
    one_over_iteration_number = 1.0 / iteration_number if iteration_number else math.inf   # (1.0 / 0 in C++: +inf, no guard upstream)
    ratio = math.pow(1.0 - math.pow(one_minus_confidence, one_over_iteration_number), 1.0 / sample_size)
    v = unseen * ratio
    return int(math.floor(v + 0.5)) if v >= 0 else int(math.ceil(v - 0.5))  # std::round: halves away from zero


# ---------------------------------------------------------------------------------------------------------------------
# PEARL
# ---------------------------------------------------------------------------------------------------------------------
class Pearl:
    def __init__(self, ctx, estimator, pts, threshold, spatial_coherence_weight, minimum_inlier_number, point_weights,
                 maximum_iteration_number=100, do_logging=False, labeling_l0="greedy", trace=None, pearl_abs="double"):
        self.ctx, self.est, self.pts = ctx, estimator, pts
        self.threshold = threshold
        self.lam = spatial_coherence_weight
        self.model_complexity_weight = float(minimum_inlier_number)   # PEARL.h:144
        self.epsilon = 1e-5                                           # PEARL.h:145
        self.minimum_inlier_number = int(minimum_inlier_number)
        self.maximum_iteration_number = maximum_iteration_number
        self.point_weights = point_weights
        self.do_logging = do_logging
        self.labeling_l0 = labeling_l0   # U-8 switch, see labeling()
        self.trace = check_trace_hook(trace)
        if pearl_abs not in ("double", "int"):
            raise ValueError("pearl_abs should be 'double' or 'int'")
        self.pearl_abs = pearl_abs       # U-16 switch, see run()
        self.n = pts.shape[0]
        self.has_engine = False      # alpha_expansion_engine != nullptr
        self.outliers_number = 0
        self.points_per_instance = []
        self.iterations = 0
        self.cycles = 0

    # PEARL.h:476-555
    def labeling(self, models, init_with_previous):
        if len(models) == 0:
            return None
        K = len(models)
        # a fresh GCoptimizationGeneralGraph starts from the all-zero labelling (PEARL.h:507-508); the previous labels
        # are only carried over when nothing was rejected (PEARL.h:541-547)
        if not (init_with_previous and self.has_engine):
            self.ctx.set_labels(np.zeros(self.n, dtype=np.int32))
        desc = np.stack([m.descriptor for m in models])
        self.ctx.pearl_unary(desc, self.threshold, self.lam)          # PEARL.h:512-519 data term
        lam = self.lam if self.lam > 0.0 else 0.0                     # :523-525, :532 smooth term only if > 0
        h = self.model_complexity_weight if self.model_complexity_weight > 0.0 else 0.0   # :528-529
        greedy = lam == 0.0 and self.labeling_l0 == "greedy" and K + 1 <= 64   # (pgx_greedy_labeling: label sets as 64-bit masks)
        if greedy:
            # [U-8, UNVERIFIED recollection of GCO-v3] no setSmoothCost / setNeighbors call was made (:523-536), so expansion()
            # (:550-551) leaves through solveSpecialCases(): per-site argmin without label costs, greedy facility location with
            # them - not alpha-expansion.  labeling_l0="expansion" runs the closed-form alpha-expansion moves instead; so does a
            # run with more than 63 instances (max_outer_iterations extension: the greedy solver keeps label sets as 64-bit
            # masks - decided above, never by catching an error: a device fault must not silently change the semantics).
            eq, e, opened = self.ctx.greedy_labeling(h)
            cycles = 1
        else:
            if lam == 0.0 and self.labeling_l0 == "greedy":
                import warnings
                warnings.warn(f"pyprogressivex: {K + 1} labels exceed the greedy lambda = 0 labelling's 64; alpha-expansion moves instead")
            eq, e, cycles = self.ctx.expansion(lam, h, 1000)          # :550-551
        self.has_engine = True
        self.cycles += cycles
        return e

    # PEARL.h:319-401
    def parameter_estimation(self, models):
        if not self.has_engine:
            return False
        K = len(models)
        # :342-352 buckets the points by label; only the bucket SIZES are used on the host (:365 and rejectInstances), the
        # members themselves are selected on the device by label (pgx_gram_labels / pgx_residual_sums): no index traffic
        counts, _ = self.ctx.bucket(K + 1, want_order=False)
        self.points_per_instance = [int(counts[k]) for k in range(K)]
        self.outliers_number = int(counts[K])
        changed = False
        if K == 0:
            return False
        # The per-instance steps of PEARL.h:365-393 run for all instances together: the sums before (:369-371), the refits
        # (:375-380) and the sums after (:388-390) are one launch each (per refit step) instead of one per instance — the
        # same numbers bit for bit, K times fewer host round trips.
        small = {k for k in range(K) if self.points_per_instance[k] < self.est.nonminimal_sample_size}   # :365
        current = np.array([np.asarray(m.descriptor, dtype=np.float64).reshape(-1) for m in models])
        before = self.ctx.residual_sums(current)
        fits = self.est.nonminimal_labels(self.ctx, K, self.point_weights, inits=current, skip=small)
        cand = current.copy()
        tried = []
        for k in range(K):
            if k in small or len(fits[k]) != 1:                       # :384
                continue
            cand[k] = np.asarray(fits[k][0], dtype=np.float64).reshape(-1)
            tried.append(k)
        accepted = set()
        after = None
        if tried:
            after = self.ctx.residual_sums(cand)
            for k in tried:
                if after[k] < before[k]:                              # :393
                    models[k].descriptor = cand[k].copy()
                    accepted.add(k)
                    changed = True
        if self.trace is not None:                                    # in the reference's per-instance order
            for k in range(K):
                cnt = self.points_per_instance[k]
                if k in small:
                    self.trace.event(EV_REFIT_SKIP, k, cnt)
                    continue
                self.trace.refit(cnt, fits[k], k in accepted)
                one = len(fits[k]) == 1
                self.trace.event(EV_REFIT, k, cnt, 2 * len(fits[k]) + (k in accepted), float(before[k]), float(after[k]) if one else 0.0)
        return changed

    # PEARL.h:275-315
    def reject_instances(self, models):
        changed = False
        for k in range(len(models) - 1, -1, -1):
            cnt = self.points_per_instance[k]
            if cnt < self.minimum_inlier_number:
                self.outliers_number += cnt
                del self.points_per_instance[k]
                del models[k]
                changed = True
                if self.trace is not None:
                    self.trace.event(EV_REJECT, k, cnt)
                if self.do_logging:
                    print(f"[Optimization] Instance {k} is rejected due to having too few inliers ({cnt}).")
        return changed

    # PEARL.h:405-472
    def run(self, models):
        iteration_number = 0
        energy, previous_energy = sys.float_info.max, -1.0
        model_rejected = False
        convergence = False
        while not convergence and iteration_number < self.maximum_iteration_number:
            iteration_number += 1
            if self.do_logging:
                print(f"[Optimization] Iteration {iteration_number}.")
            init_prev = iteration_number > 1 and not model_rejected           # :429-431
            models_before = len(models)
            e = self.labeling(models, init_prev)                              # :434
            if e is not None:
                energy = e
            if self.trace is not None:
                self.trace.event(EV_PEARL_ITER, iteration_number, models_before, int(init_prev), float(energy))
            if self.do_logging:
                print(f"[Optimization] The energy of the labeling is {energy}.")
            params_changed = self.parameter_estimation(models)               # :453
            model_rejected = self.reject_instances(models)                   # :458
            delta = energy - previous_energy
            if self.pearl_abs == "int":                                       # [U-16] int abs(int): truncation towards zero first
                abs_delta = float(abs(int(delta))) if -2147483648.0 < delta < 2147483648.0 else sys.float_info.max
            else:
                abs_delta = abs(delta)
            if not model_rejected and not params_changed and abs_delta < self.epsilon and iteration_number > 1:   # :463-467
                convergence = True
            if self.trace is not None:
                self.trace.event(EV_PEARL_END, iteration_number, 2 * int(params_changed) + int(model_rejected),
                                 2 * len(models) + int(convergence))
            previous_energy = energy
        self.iterations += iteration_number
        return True

    # PEARL.h:218-247
    def get_labeling(self):
        if not self.has_engine:
            return np.zeros(self.n, dtype=np.int64), 0
        lab = self.ctx.get_labels().astype(np.int64)
        return lab, int(lab.max()) if lab.size else 0


# ---------------------------------------------------------------------------------------------------------------------
# Progressive-X
# ---------------------------------------------------------------------------------------------------------------------
class ProgressiveX:
    def __init__(self, ctx, estimator, pts, graph, sampler, settings, scoring_exponent=2, do_logging=False,
                 exchange=None, graph_resident=False, trace=None):
        self.ctx, self.est, self.pts, self.graph = ctx, estimator, pts, graph
        self.trace = check_trace_hook(trace)
        self.graph_resident = graph_resident   # built by ctx.graph_build: already on the device
        self.sampler, self.settings = sampler, settings
        self.scoring_exponent = int(scoring_exponent)   # setExponent(const int) truncates (scoring_function...h:39)
        self.do_logging = do_logging
        self.exchange = exchange
        self.models = []
        self.statistics = Statistics()
        self.n = pts.shape[0]

    def _log(self, msg):
        if self.do_logging:
            print(msg)

    # progressive_x.h:519-559
    def initialize(self):
        s = self.settings
        self.statistics.labeling = np.zeros(self.n, dtype=np.int64)                          # :522
        self.T2 = 9.0 / 4.0 * s.inlier_outlier_threshold * s.inlier_outlier_threshold        # :523
        self.ctx.set_points(self.est.model_type, self.pts)                                   # compound := 0 (:524)
        if self.graph is not None and s.spatial_coherence_weight > 0.0 and not self.graph_resident:
            self.ctx.set_graph(*self.graph)
        self.pearl = Pearl(self.ctx, self.est, self.pts, s.inlier_outlier_threshold, s.spatial_coherence_weight,
                           s.minimum_number_of_inliers, s.point_weights, 100, self.do_logging,
                           labeling_l0=getattr(s, "labeling_l0", "greedy"), trace=self.trace,
                           pearl_abs=getattr(s, "pearl_abs", "double"))                          # :527-534
        self.engine = _proposal.ProposalEngine(self.ctx, self.est, self.pts, self.sampler, s, self.exchange, trace=self.trace)

    # progressive_x.h:565-591
    def is_putative_model_valid(self, model, inlier_number):
        s = self.settings
        if inlier_number < max(self.est.sample_size, s.minimum_number_of_inliers):            # :574
            if self.trace is not None:
                self.trace.event(EV_VALIDATION, 0, 1, 0, float("nan"))
            return False
        # lowest preference slot no live model holds: slots of proposals that failed this test and of instances PEARL removed
        # are reused (each is N * 8 bytes on the device)
        used = {m.slot for m in self.models}
        model.slot = next(k for k in range(len(used) + 1) if k not in used)
        r = self.ctx.preference(model.descriptor, self.T2, model.slot)                        # :578-579
        denom = r["pref_sqnorm"] + r["comp_sqnorm"] - r["dot"]
        tanimoto = r["dot"] / denom if denom != 0.0 else float("nan")                          # :583-585 (0/0 -> NaN)
        valid = not (s.maximum_tanimoto_similarity < tanimoto)                                 # :587 (NaN -> valid)
        if self.trace is not None:
            self.trace.event(EV_VALIDATION, int(valid), 0 if valid else 2, 0, float(tanimoto))
        return valid

    # progressive_x.h:597-624
    def update_compound_model(self):
        if len(self.models) == 0:
            if self.trace is not None:
                self.trace.event(EV_COMPOUND, 0, 0, 0, float(np.sum(self.ctx.get_compound())))
            return
        comp = self.ctx.compound_update([m.slot for m in self.models], want_compound=self.trace is not None)   # max over the STORED (stale) preference vectors
        if self.trace is not None:
            self.trace.event(EV_COMPOUND, len(self.models), 0, 0, float(np.sum(comp)))

    # progressive_x.h:251-489
    def run(self):
        t_main = time.perf_counter()
        self._log("Progressive-X is started...")
        self.initialize()
        s, st = self.settings, self.statistics
        number_of_ransac_iterations = 0
        unaccepted = 0
        self._log("The main iteration is started...")
        tr = self.trace
        break_reason = 0
        for current_iteration in range(s.max_outer_iterations):                               # :272 (hard 10 upstream)
            if tr is not None:
                tr.event(EV_OUTER, current_iteration)
            self._log("-------------------------------------------")
            self._log(f"Iteration {current_iteration + 1}.")
            it_stats = dict(time_of_proposal_engine=0.0, time_of_model_validation=0.0, time_of_optimization=0.0,
                            time_of_compound_model_update=0.0, number_of_instances=0)
            # ---- proposal (:294)
            t0 = time.perf_counter()
            prop = self.engine.run(self.T2, has_compound=len(self.models) > 0, exponent=self.scoring_exponent,
                                   weights=s.point_weights)
            it_stats["time_of_proposal_engine"] = time.perf_counter() - t0
            if prop is None or prop["model"] is None:                                         # :301-303
                if tr is not None:
                    tr.proposal(None, None, 0)
                    tr.event(EV_PROPOSAL_EMPTY)
                continue
            putative = Model(prop["model"])
            inliers = prop["inliers"]
            number_of_ransac_iterations += prop["iterations"]                                 # :317-318
            if tr is not None:
                tr.proposal(putative.descriptor, inliers, prop["iterations"])
                tr.event(EV_PROPOSAL, len(inliers), int(prop["iterations"]), int(number_of_ransac_iterations))
            self._log(f"A model proposed with {len(inliers)} inliers\nin {it_stats['time_of_proposal_engine']} "
                      f"seconds ({prop['iterations']} iterations).")
            # ---- validation (:334)
            self._log("Check if the model should be added to the compound instance.")
            t0 = time.perf_counter()
            if not self.is_putative_model_valid(putative, len(inliers)):
                self._log("The model is not accepted to be added to the compound instances. The number of "
                          f"consecutively rejected proposals is {unaccepted} (< {s.max_proposal_number_without_change})")
                unaccepted += 1                                                               # :342 (never reset)
                if tr is not None:
                    tr.event(EV_UNACCEPTED, unaccepted)
                if unaccepted == s.max_proposal_number_without_change:
                    break_reason = 1
                    break
                continue
            it_stats["time_of_model_validation"] = time.perf_counter() - t0
            self._log(f"The model has been accepted in {it_stats['time_of_model_validation']} seconds.")
            # ---- optimisation (:366-403)
            t0 = time.perf_counter()
            self.models.append(putative)                                                      # :369
            self._log("Model optimization started...")
            if len(self.models) == 1:
                st.inliers_of_each_model.append(inliers)                                      # :378-379
                st.labeling[:] = 1                                                            # :382
                st.labeling[inliers] = 0                                                      # :383-384
                if tr is not None:
                    tr.event(EV_SINGLE_MODEL, len(inliers))
            else:
                self.pearl.run(self.models)                                                   # :390
                st.labeling, model_number = self.pearl.get_labeling()                         # :396
                if tr is not None:
                    tr.event(EV_LABELING, model_number, len(self.models))
                if model_number != len(self.models):
                    self._log("Models have been removed during the optimization.\n")
            it_stats["time_of_optimization"] = time.perf_counter() - t0
            self._log(f"Model optimization finished in {it_stats['time_of_optimization']} seconds.")
            # ---- compound update (:423)
            t0 = time.perf_counter()
            self.update_compound_model()
            it_stats["time_of_compound_model_update"] = time.perf_counter() - t0
            self._log(f"Compound instance (containing {len(self.models)} models) is updated in "
                      f"{it_stats['time_of_compound_model_update']} seconds.")
            it_stats["number_of_instances"] = len(self.models)
            # ---- predicted unseen inliers (:447-457)
            if len(self.models) == 1:
                covered = len(st.inliers_of_each_model)    # quirk :451 — the COUNT of stored inlier sets, not of inliers
            else:
                covered = self.n - self.pearl.outliers_number
            unseen = predicted_unseen_inliers(s.one_minus_confidence, self.est.sample_size,
                                              number_of_ransac_iterations, covered, self.n)
            if tr is not None:
                tr.event(EV_UNSEEN, covered, unseen)
            st.iteration_statistics.append(it_stats)
            st.total_time_of_proposal_engine += it_stats["time_of_proposal_engine"]
            st.total_time_of_model_validation += it_stats["time_of_model_validation"]
            st.total_time_of_optimization += it_stats["time_of_optimization"]
            st.total_time_of_compound_model_calculation += it_stats["time_of_compound_model_update"]
            self._log(f"The predicted number of inliers (with confidence {s.confidence})\nnot covered by the compound "
                      f"instance is {unseen}.")
            if unseen < s.minimum_number_of_inliers:                                          # :468
                break_reason = 2
                break
            if len(self.models) >= s.maximum_model_number:                                    # :472
                break_reason = 3
                break
        if tr is not None:
            tr.event(EV_BREAK, break_reason)
        st.processing_time = time.perf_counter() - t_main
        st.pearl_iterations = self.pearl.iterations
        st.expansion_cycles = self.pearl.cycles
        return self.models, st
