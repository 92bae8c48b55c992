/*
 * progx_replay.h — independent CPU oracle for the CONTROL-FLOW rows of the hot path (SURVEY.md §8 a5, a10–a13):
 *   progx::ProgressiveX::run / initialize / isPutativeModelValid / updateCompoundModel / getPredictedUnseenInliers
 *       (/root/reference/src/pyprogressivex/include/progressive_x.h:251-624)
 *   pearl::PEARL::run / labeling / parameterEstimation / rejectInstances / getLabeling
 *       (/root/reference/src/pyprogressivex/include/PEARL.h:218-555)
 *
 * TEST INFRASTRUCTURE ONLY (same rules as pgx_oracle.h): nothing under progressive-x_amd/ may include, link or call it.
 * PARITY UNPINNED like the rest of the oracle: the reference holds no test for these functions; this file is a second,
 * independent restatement of the C++ (written from progressive_x.h / PEARL.h, not from pyprogressivex/_engine.py), in the
 * reference's own statement order, so that a transcription error in either restatement shows up as a disagreement.
 *
 * What is replayed and what is taken from a recorded TRACE.  The two pieces of the loop that live in the absent
 * graph-cut-ransac submodule are not decisions of these rows and are consumed from the trace:
 *   - proposal_engine->run (progressive_x.h:294-299): per outer iteration the putative model (or "empty descriptor"),
 *     the inlier indices of its RANSAC statistics and its iteration number;
 *   - model_estimator_->estimateModelNonminimal (PEARL.h:375-380): per call the models it returned (count + descriptor).
 * Everything else — the size gate, preference vectors, Tanimoto test, the never-reset reject counter, labelling (unary
 * table + alpha-expansion / greedy special case through pgx_oracle.c), the warm-start rule, bucketing, residual sums,
 * strict refit acceptance, reverse-order rejection, convergence test, getLabeling, stale-preference compound update,
 * predicted unseen inliers, both break rules and the 10-proposal cap — is recomputed here from the points.
 * The replay emits one EVENT per decision; the GPU path records the same events through the `trace=` hook of
 * pyprogressivex._engine and the tests compare the two streams.
 */
#ifndef PROGX_REPLAY_H
#define PROGX_REPLAY_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t model_type;                          /* PGXO_* */
    int32_t max_outer_iterations;                /* progressive_x.h:272: the literal 10 */
    int32_t pearl_maximum_iteration_number;      /* progressive_x.h:532: the literal 100 */
    int32_t labeling_l0;                         /* U-8: 0 = GCO's special-case solver when lambda == 0, 1 = alpha-expansion moves */
    int32_t pearl_abs_int;                       /* U-16: PEARL.h:465 `abs(energy - previous_energy)` resolved to int abs(int) */
    int32_t pad_;
    uint64_t sample_size;                        /* _ModelEstimator::sampleSize() */
    uint64_t nonminimal_sample_size;             /* _ModelEstimator::nonMinimalSampleSize() */
    uint64_t minimum_number_of_inliers;          /* MultiModelSettings (progressive_x.h:38-47) */
    uint64_t max_proposal_number_without_change;
    uint64_t maximum_model_number;
    double maximum_tanimoto_similarity;
    double one_minus_confidence;
    double inlier_outlier_threshold;
    double spatial_coherence_weight;
    /* 0: every decision is the replay's own.  > 0 (the GPU tests use 1e-12): PEARL.h:393 compares two sums of ~n doubles whose
     * last bits depend on the summation order (sequential upstream and here, a fixed tree on the GPU); when the two sums agree
     * to this relative tolerance the comparison is a numerical tie and the replay follows the recorded run's choice
     * (pgxr_trace.refit_accepted), marking the event (bit 2 of REFIT's c).  Anything outside the tolerance stays independent. */
    double refit_tie_rtol;
} pgxr_settings;

typedef struct {
    /* proposal_engine->run, one entry per outer iteration the recording run made */
    int32_t n_proposals;
    int32_t pad_;
    const double *models;          /* n_proposals x param_dim */
    const uint8_t *empty;          /* 1: descriptor with 0 rows / cols (progressive_x.h:301-303) */
    const int64_t *inlier_off;     /* n_proposals + 1 offsets into `inliers` */
    const int64_t *inliers;        /* proposal_engine->getRansacStatistics().inliers */
    const uint64_t *iterations;    /* ...iteration_number */
    /* estimateModelNonminimal, in call order */
    int64_t n_refits;
    const int64_t *refit_inliers;  /* inlier_number of the call (checked: a different number here means the labelling diverged) */
    const int32_t *refit_models_n; /* current_models.size() */
    const double *refit_models;    /* n_refits x param_dim: current_models.back().descriptor */
    const int8_t *refit_accepted;  /* what the recording run decided at PEARL.h:393 (1 / 0, -1 unknown); only read for ties, may be NULL */
} pgxr_trace;

/* event codes; fields a, b, c are integers, x, y doubles */
enum {
    PGXR_EV_OUTER = 1,          /* a = current_iteration */
    PGXR_EV_PROPOSAL_EMPTY = 2, /* `continue` at progressive_x.h:303 */
    PGXR_EV_PROPOSAL = 3,       /* a = inlier number, b = iteration number, c = total number_of_ransac_iterations */
    PGXR_EV_VALIDATION = 4,     /* a = 1 valid / 0 not, b = reason (0 valid, 1 size gate :574, 2 Tanimoto :587), x = tanimoto (NaN at the gate) */
    PGXR_EV_UNACCEPTED = 5,     /* a = unaccepted_putative_instances after ++ (:342) */
    PGXR_EV_SINGLE_MODEL = 6,   /* a = inliers written as label 0 (:378-384) */
    PGXR_EV_PEARL_ITER = 7,     /* a = iteration_number, b = models before, c = initialize_with_previous_labeling, x = energy */
    PGXR_EV_REFIT_SKIP = 8,     /* a = instance, b = inlier number (< nonMinimalSampleSize, PEARL.h:365) */
    PGXR_EV_REFIT = 9,          /* a = instance, b = inlier number, c = models returned * 2 + accepted (+ 4: a tie steered by the trace), x = sum before, y = sum after (0 if c/2 != 1) */
    PGXR_EV_REJECT = 10,        /* a = instance index at the time of removal, b = its inlier number (PEARL.h:293-312) */
    PGXR_EV_PEARL_END = 11,     /* a = iteration_number, b = params changed * 2 + model rejected, c = models after * 2 + convergence */
    PGXR_EV_LABELING = 12,      /* a = instance_number_ of getLabeling (max label), b = models.size() */
    PGXR_EV_COMPOUND = 13,      /* a = models in the compound instance, x = sum of the compound vector (after the update) */
    PGXR_EV_UNSEEN = 14,        /* a = inlier_number_of_compound_model_ passed (:447-457), b = unseen_inliers */
    PGXR_EV_BREAK = 15          /* a = 1 reject counter (:343), 2 unseen < minimum (:468), 3 model number (:472), 0 loop ran out */
};

typedef struct {
    int32_t code, pad_;
    int64_t a, b, c;
    double x, y;
} pgxr_event;

/*
 * Replays the trace.  off / idx / mult: the symmetric CSR neighbourhood graph pgx_oracle.c's expansion takes (may be NULL
 * when spatial_coherence_weight == 0).  Outputs: events (capacity max_events), labels_out[n] = statistics.labeling,
 * models_out (capacity max_models x param_dim) and *models_n = the final compound instance, consumed[2] = proposals and
 * refit records read from the trace.
 * Returns the number of events (>= 0), or
 *   -1 bad argument, -2 the trace ran out of proposals, -3 a refit record does not match the replay's inlier number
 *   (the labelling of the recording run differs), -4 the trace ran out of refit records, -5 event / model capacity.
 */
int64_t pgxr_replay(const pgxr_settings *settings, const double *pts, int64_t n,
                    const int32_t *off, const int32_t *idx, const int32_t *mult,
                    const pgxr_trace *trace,
                    pgxr_event *events, int64_t max_events,
                    int64_t *labels_out, double *models_out, int32_t max_models, int32_t *models_n,
                    int64_t consumed[2]);

const char *pgxr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
