/* progx_proposal.c - see progx_proposal.h.  TEST INFRASTRUCTURE; PARITY UNPINNED. */
#include "progx_proposal.h"

#include <math.h>

typedef struct {
    const pgxq_settings* s;
    double best_score;
    int64_t best_count;
    int64_t lo_runs, cuts;
    int64_t R, r_at;
    const int64_t *round_inliers, *round_off, *cand_counts;
    const double* cand_scores;
    pgxq_event* ev;
    int64_t nev, max_events;
    int error;
} State;

static void emit(State* st, int code, int64_t a, int64_t b, int64_t c, double x)
{
    if (st->nev < st->max_events) {
        pgxq_event* e = &st->ev[st->nev];
        e->code = code; e->pad_ = 0; e->a = a; e->b = b; e->c = c; e->x = x;
    }
    st->nev += 1;
}

/* the standard termination criterion [U-9]: how many iterations make a sample of all-inliers likely enough */
static double iteration_bound(int64_t inliers, int64_t n, int m, double confidence)
{
    double q = (double)inliers / (double)n;
    if (q > 1.0) q = 1.0;
    if (q < 0.0) q = 0.0;
    const double qm = pow(q, (double)m);
    if (qm <= 0.0) return INFINITY;
    if (qm >= 1.0) return 1.0;
    const double den = log1p(-qm);
    return den < 0.0 ? log(1.0 - confidence) / den : INFINITY;
}

/* graphCutLocalOptimization [U-12]: rounds of { cut -> candidates from the inliers -> keep the best strictly better candidate }
 * while the proposal's cut budget lasts and a round still improves */
static void local_optimization(State* st)
{
    const pgxq_settings* s = st->s;
    st->lo_runs += 1;
    const int64_t limit = 7 * (int64_t)s->sample_size;
    while (st->cuts < s->max_cuts) {
        st->cuts += 1;
        if (st->r_at >= st->R) { st->error = 1; return; }          /* the recording holds no such round */
        const int64_t r = st->r_at++;
        const int64_t inl = st->round_inliers[r];
        const int64_t size = inl < limit ? inl : limit;
        /* which candidates can exist: an inner RANSAC of refits on `size` of more inliers, one refit of all of them, or nothing */
        const int branch = (size < inl && size >= s->nonminimal_sample_size) ? 1
                           : ((s->sample_size < inl && inl >= s->nonminimal_sample_size) ? 2 : 0);
        const int64_t c0 = st->round_off[r], c1 = st->round_off[r + 1];
        int updated = 0;
        if (branch != 0)
            for (int64_t k = c0; k < c1; ++k)                        /* walked in order: a strictly better score replaces the best */
                if (st->cand_counts[k] > 0 && st->cand_scores[k] > st->best_score) {
                    st->best_score = st->cand_scores[k];
                    st->best_count = st->cand_counts[k];
                    updated = 1;
                }
        emit(st, PGXQ_EV_LO_ROUND, branch, branch != 0 ? c1 - c0 : 0, updated, st->best_score);
        if (branch == 0 || c1 == c0 || !updated) break;
    }
    emit(st, PGXQ_EV_LO_END, st->cuts, st->best_count, st->lo_runs, st->best_score);
}

int64_t pgxq_replay(const pgxq_settings* s, int64_t H, const int64_t* counts, const double* scores, const int64_t* src,
                    int64_t R, const int64_t* round_inliers, const int64_t* round_off, const int64_t* cand_counts, const double* cand_scores,
                    int64_t Q, const int64_t* lsq_inliers, const int64_t* lsq_fits, const int64_t* lsq_counts, const double* lsq_scores,
                    pgxq_event* events, int64_t max_events, int64_t consumed[2])
{
    State st = {s, -INFINITY, 0, 0, 0, R, 0, round_inliers, round_off, cand_counts, cand_scores, events, 0, max_events, 0};
    int64_t best = -1, it_best = 0;
    double bound = (double)s->max_iters;
    /* ---- the main loop, one hypothesis after the other in generation order */
    for (int64_t h = 0; h < H; ++h) {
        const int64_t it = src[h] + 1;                               /* the iteration that drew this hypothesis' sample */
        if ((double)it > bound && it > s->min_iters) break;         /* enough iterations for the confidence asked */
        const int64_t c = counts[h];
        if (c + 1 < st.best_count) continue;                         /* scoring_function_with_compound_model.h:105-106 */
        if (!(c > 0 && scores[h] > st.best_score)) continue;         /* first strictly better score wins */
        best = h;
        st.best_score = scores[h];
        st.best_count = c;
        it_best = it;
        emit(&st, PGXQ_EV_BEST, h, it, c, scores[h]);
        if (s->every_best && it > s->lo_after && c > s->sample_size) {
            local_optimization(&st);
            if (st.error) { consumed[0] = st.r_at; consumed[1] = 0; return -1; }
        }
        const double b = iteration_bound(st.best_count, s->n, s->sample_size, s->confidence);
        bound = b < (double)s->max_iters ? b : (double)s->max_iters;
    }
    double cb = ceil(bound);
    if (cb > (double)s->max_iters) cb = (double)s->max_iters;
    int64_t iterations = it_best;
    if ((double)iterations < cb) iterations = (int64_t)cb;
    const int64_t floor_it = s->min_iters < s->samples ? s->min_iters : s->samples;
    if (iterations < floor_it) iterations = floor_it;
    if (iterations < 1) iterations = 1;
    emit(&st, PGXQ_EV_WALK_END, iterations, best, st.lo_runs, st.best_score);
    int64_t q_at = 0;
    if (best >= 0) {
        if (st.lo_runs == 0) {                                       /* "apply the local optimisation if it has not been applied yet" */
            local_optimization(&st);
            if (st.error) { consumed[0] = st.r_at; consumed[1] = 0; return -1; }
        }
        /* final iterated least squares: a refit of the inliers is kept while the score improves */
        int64_t budget = s->lsq_budget, steps = 0;
        while (budget > 0) {
            budget -= 1;
            if (q_at >= Q) { consumed[0] = st.r_at; consumed[1] = q_at; return -2; }
            const int64_t q = q_at++;
            steps += 1;
            if (lsq_inliers[q] < s->nonminimal_sample_size) break;
            if (lsq_fits[q] != 1) break;
            if (lsq_scores[q] > st.best_score && lsq_counts[q] > 0) st.best_score = lsq_scores[q];
            else break;
        }
        emit(&st, PGXQ_EV_LSQ, steps, 0, 0, st.best_score);
        emit(&st, PGXQ_EV_FINAL, 0, 0, 0, st.best_score);
    }
    consumed[0] = st.r_at;
    consumed[1] = q_at;
    return st.nev;
}
