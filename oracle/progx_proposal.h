/* progx_proposal.h - an independent restatement of ONE proposal of the GC-RANSAC loop as Progressive-X drives it
 * (gcransac::GCRANSAC::run, called at /root/reference/src/pyprogressivex/include/progressive_x.h:294-299 with the settings of
 * :541-545 and the scoring function of scoring_function_with_compound_model.h), written from the sequential algorithm: iterate the
 * hypotheses in generation order; Score()'s early exit (scoring_function_with_compound_model.h:105-106: a hypothesis whose inlier
 * count + 1 falls below the so-far-best's is not considered); "first strictly better score wins"; the iteration bound
 * log(1 - confidence) / log(1 - q^m) after every so-far-best; the local optimisation at every so-far-best found after
 * min_iteration_number_before_lo iterations, once more after the loop if it never ran, then the final iterated least squares.
 * TEST INFRASTRUCTURE (tests/ only).  PARITY UNPINNED: GC-RANSAC's sources are absent from the snapshot (empty submodule); the parts
 * written from memory of upstream are tagged [U-9] (the loop) and [U-12] (the graph-cut local optimisation), as in the product.
 *
 * It REPLAYS a recorded proposal: the score table of the batch (what pgx_score returned for every hypothesis, in generation
 * order), and for every local-optimisation round / least-squares step what the cut and the refit solver returned (how many
 * inliers the cut kept, the score rows of the candidate models).  Everything decided from those - which hypothesis becomes the
 * so-far-best and when, the bound, when the local optimisation fires, which candidate it keeps, when it stops, the iteration
 * count reported to ProgressiveX::run - is recomputed here and compared event by event with what the product reported. */
#ifndef PROGX_PROPOSAL_H
#define PROGX_PROPOSAL_H
#include <stdint.h>

enum { PGXQ_EV_BEST = 1, PGXQ_EV_LO_ROUND = 2, PGXQ_EV_LO_END = 3, PGXQ_EV_WALK_END = 4, PGXQ_EV_LSQ = 5, PGXQ_EV_FINAL = 6 };

typedef struct {
    int64_t n;                        /* points */
    int64_t samples;                  /* minimal samples drawn for this proposal */
    int32_t sample_size, nonminimal_sample_size;
    double confidence;
    int64_t max_iters, min_iters, lo_after;   /* max_iteration_number, min_iteration_number, min_iteration_number_before_lo */
    int32_t every_best;               /* the local optimisation runs at every so-far-best (the sequential loop's cadence) */
    int32_t max_cuts;                 /* max_graph_cut_number: cuts of one proposal */
    int32_t lsq_budget;               /* max_least_squares_iterations */
    int32_t pad_;
} pgxq_settings;

typedef struct { int32_t code, pad_; int64_t a, b, c; double x; } pgxq_event;

/* counts / scores / src [H]: the table the walk sees (scores: -inf where the count is 0 or the score is NaN), src = sample number
 * of each hypothesis.  Local optimisation rounds, in the order they ran: round_inliers [R] (what the cut kept), round_off [R + 1]
 * into cand_counts / cand_scores (the candidates' rows).  Least-squares steps: lsq_inliers / lsq_fits / lsq_counts / lsq_scores [Q]
 * (fits = models the solver returned; count / score of the one candidate, ignored unless fits == 1).
 * returns the number of events written (<= max_events), or -1 - k when the recording ran out of rounds (k = 0) / steps (k = 1). */
int64_t pgxq_replay(const pgxq_settings* s, int64_t H, const int64_t* counts, const double* scores, const int64_t* src,
                    int64_t R, const int64_t* round_inliers, const int64_t* round_off, const int64_t* cand_counts, const double* cand_scores,
                    int64_t Q, const int64_t* lsq_inliers, const int64_t* lsq_fits, const int64_t* lsq_counts, const double* lsq_scores,
                    pgxq_event* events, int64_t max_events, int64_t consumed[2]);
#endif
