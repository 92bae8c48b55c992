/*
 * progx_replay.c — see progx_replay.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 *
 * A second restatement of the reference's host control flow, independent of pyprogressivex/_engine.py: plain C in the
 * statement order of the C++ (std::vector -> malloc'ed arrays, size_t -> uint64_t, Eigen vectors -> double*).  Paths are
 * relative to /root/reference/src/pyprogressivex/include/.  The O(N) arithmetic comes from pgx_oracle.c.
 */
#include "progx_replay.h"
#include "pgx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_PARAM 18
#define MAXV(a, b) ((a) > (b) ? (a) : (b))   /* OpenCV's MAX macro (progressive_x.h:574,621; PEARL.h:243) */

static char g_err[512];
const char *pgxr_last_error(void) { return g_err; }

/* progx::Model (progx_model.h:43-99): descriptor + the preference vector setPreferenceVector last computed */
typedef struct {
    double descriptor[MAX_PARAM];
    double *preference_vector;   /* NULL until setPreferenceVector */
} model_t;

typedef struct {
    model_t *data;
    size_t size, cap;
} model_vec;

static void mv_push(model_vec *v, const model_t *m)   /* emplace_back(copy) */
{
    if (v->size == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 8;
        v->data = (model_t *)realloc(v->data, v->cap * sizeof(model_t));
    }
    v->data[v->size++] = *m;
}

static void mv_erase(model_vec *v, size_t k)          /* erase(begin() + k) */
{
    free(v->data[k].preference_vector);
    memmove(v->data + k, v->data + k + 1, (v->size - k - 1) * sizeof(model_t));
    --v->size;
}

/* ---- event log ------------------------------------------------------------------------------------------------ */
typedef struct {
    pgxr_event *ev;
    int64_t n, cap;
    int overflow;
} evlog;

static void emit(evlog *g, int code, int64_t a, int64_t b, int64_t c, double x, double y)
{
    if (g->n >= g->cap) { g->overflow = 1; return; }
    pgxr_event *e = &g->ev[g->n++];
    e->code = code; e->pad_ = 0; e->a = a; e->b = b; e->c = c; e->x = x; e->y = y;
}

/* ---- trace cursor --------------------------------------------------------------------------------------------- */
typedef struct {
    const pgxr_trace *t;
    int32_t next_proposal;
    int64_t next_refit;
    int status;   /* 0 ok, else the (negative) return code */
} cursor;

/* ---- pearl::PEARL (PEARL.h:130-216) --------------------------------------------------------------------------- */
typedef struct {
    const pgxr_settings *s;
    const double *pts;
    int64_t point_number;
    int d, p;
    const int32_t *off, *idx, *mult;
    /* constructor arguments (PEARL.h:135-152) */
    uint64_t maximum_iteration_number, minimum_inlier_number;
    double inlier_outlier_threshold, spatial_coherence_weight, model_complexity_weight, epsilon;
    /* GCoptimizationGeneralGraph *alpha_expansion_engine: its labelling survives until the next `new` */
    int engine;                 /* != nullptr */
    int32_t *engine_labels;     /* whatLabel(i) */
    /* std::vector<std::vector<size_t>> points_per_instance; std::vector<size_t> outliers */
    int64_t **points_per_instance;
    uint64_t *instance_size;
    size_t instance_number;     /* points_per_instance.size() */
    uint64_t outliers_size;     /* outliers.size() = getOutlierNumber() (PEARL.h:170) */
    evlog *log;
    cursor *cur;
} pearl_t;

static void pearl_clear_instances(pearl_t *P)
{
    for (size_t k = 0; k < P->instance_number; ++k) free(P->points_per_instance[k]);
    free(P->points_per_instance);
    free(P->instance_size);
    P->points_per_instance = NULL;
    P->instance_size = NULL;
    P->instance_number = 0;
}

/* PEARL::labeling, PEARL.h:476-555 */
static int pearl_labeling(pearl_t *P, const model_vec *models_, int initialize_with_previous_labeling_, double *energy_)
{
    /* :486-487 */
    if (models_->size == 0)
        return 0;
    const int64_t n = P->point_number;
    /* :490-497 the previous labelling, if asked for and if an engine exists */
    int32_t *previous_labeling = NULL;
    size_t previous_labeling_size = 0;
    if (initialize_with_previous_labeling_ && P->engine) {
        previous_labeling = (int32_t *)malloc((size_t)n * sizeof(int32_t));
        for (int64_t i = 0; i < n; ++i)
            previous_labeling[i] = P->engine_labels[i];
        previous_labeling_size = (size_t)n;
    }
    /* :502-508 delete + new GCoptimizationGeneralGraph(point_number, models + 1): a fresh engine labels every site 0 */
    const int K = (int)models_->size, L = K + 1;
    memset(P->engine_labels, 0, (size_t)n * sizeof(int32_t));
    P->engine = 1;
    /* :512-519 data term (dataEnergyFunctor, PEARL.h:82-128) in the oracle's 2^-32 fixed point */
    double *desc = (double *)malloc((size_t)K * (size_t)P->p * sizeof(double));
    for (int k = 0; k < K; ++k)
        memcpy(desc + (size_t)k * P->p, models_->data[k].descriptor, (size_t)P->p * sizeof(double));
    int64_t *Dq = (int64_t *)malloc((size_t)n * (size_t)L * sizeof(int64_t));
    pgxo_unary_q(P->s->model_type, P->pts, n, desc, K, P->inlier_outlier_threshold, P->spatial_coherence_weight, Dq);
    free(desc);
    /* :523-525, :532-536 smooth cost and neighbours only if the weight is positive; :528-529 label cost only if positive */
    const double lambda = P->spatial_coherence_weight > 0.0 ? P->spatial_coherence_weight : 0.0;
    const double h = P->model_complexity_weight > 0.0 ? P->model_complexity_weight : 0.0;
    /* :541-547 */
    if (initialize_with_previous_labeling_ && previous_labeling_size > 0) {
        for (int64_t i = 0; i < n; ++i)
            P->engine_labels[i] = previous_labeling[i];
    }
    free(previous_labeling);
    /* :549-551 expansion(iteration_number, 1000) */
    int64_t energy_q = 0;
    int rc = 0;
    if (lambda == 0.0 && P->s->labeling_l0 == 0 && L <= 64) {
        /* U-8: without setSmoothCost / setNeighbors GCO-v3 leaves through its special-case solver (pgx_oracle.c) */
        pgxo_greedy_labeling(n, L, Dq, pgxo_quantize(h), P->engine_labels, &energy_q);
    } else {
        const int64_t lambda_q = 2 * (int64_t)nearbyint(lambda * 2147483648.0);   /* weight of one directed entry, even (U-6) */
        int cycles = 0;
        if (lambda_q > 0) {
            if (!P->off) { free(Dq); return -1; }
            rc = pgxo_expansion(n, L, Dq, P->off, P->idx, P->mult, lambda_q, pgxo_quantize(h), P->engine_labels, 1000, &energy_q, &cycles);
        } else {
            int32_t *zoff = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
            int32_t zidx = 0, zmult = 1;
            rc = pgxo_expansion(n, L, Dq, zoff, &zidx, &zmult, 0, pgxo_quantize(h), P->engine_labels, 1000, &energy_q, &cycles);
            free(zoff);
        }
    }
    free(Dq);
    if (rc != 0) return -1;
    *energy_ = (double)energy_q / 4294967296.0;
    return 1;
}

/* PEARL::parameterEstimation, PEARL.h:319-401 */
static void pearl_parameter_estimation(pearl_t *P, model_vec *models_, int *changed_)
{
    /* :326-331 */
    if (!P->engine)
        return;
    /* :334 */
    const size_t instance_number = models_->size;
    /* :337-339 */
    pearl_clear_instances(P);
    P->points_per_instance = (int64_t **)calloc(instance_number ? instance_number : 1, sizeof(int64_t *));
    P->instance_size = (uint64_t *)calloc(instance_number ? instance_number : 1, sizeof(uint64_t));
    P->instance_number = instance_number;
    P->outliers_size = 0;
    /* :342-352 */
    for (size_t k = 0; k < instance_number; ++k)
        P->points_per_instance[k] = (int64_t *)malloc((size_t)(P->point_number ? P->point_number : 1) * sizeof(int64_t));
    for (int64_t point_idx = 0; point_idx < P->point_number; ++point_idx) {
        const size_t label = (size_t)P->engine_labels[point_idx];
        if (label < instance_number)
            P->points_per_instance[label][P->instance_size[label]++] = point_idx;
        else
            ++P->outliers_size;
    }
    /* :355-400 */
    for (size_t instance_idx = 0; instance_idx < instance_number; ++instance_idx) {
        const int64_t *current_inliers = P->points_per_instance[instance_idx];
        const uint64_t inlier_number = P->instance_size[instance_idx];
        /* :365-366 */
        if (inlier_number < P->s->nonminimal_sample_size) {
            emit(P->log, PGXR_EV_REFIT_SKIP, (int64_t)instance_idx, (int64_t)inlier_number, 0, 0.0, 0.0);
            continue;
        }
        /* :369-371 the sum of UNSQUARED residuals before */
        double sum_of_residuals_before = 0.0;
        for (uint64_t j = 0; j < inlier_number; ++j)
            sum_of_residuals_before += pgxo_residual(P->s->model_type, P->pts + (size_t)current_inliers[j] * P->d,
                                                     models_->data[instance_idx].descriptor);
        /* :374-380 estimateModelNonminimal: from the trace */
        cursor *c = P->cur;
        if (c->next_refit >= c->t->n_refits) {
            snprintf(g_err, sizeof g_err, "trace ran out of refit records (needed #%lld, instance %zu with %llu inliers)",
                     (long long)c->next_refit, instance_idx, (unsigned long long)inlier_number);
            c->status = -4;
            return;
        }
        const int64_t r = c->next_refit++;
        if ((uint64_t)c->t->refit_inliers[r] != inlier_number) {
            snprintf(g_err, sizeof g_err, "refit record #%lld was computed for %lld inliers, the replay's instance %zu has %llu",
                     (long long)r, (long long)c->t->refit_inliers[r], instance_idx, (unsigned long long)inlier_number);
            c->status = -3;
            return;
        }
        const int32_t current_models_size = c->t->refit_models_n[r];
        /* :384-385 */
        if (current_models_size != 1) {
            emit(P->log, PGXR_EV_REFIT, (int64_t)instance_idx, (int64_t)inlier_number, (int64_t)current_models_size * 2,
                 sum_of_residuals_before, 0.0);
            continue;
        }
        const double *fitted = c->t->refit_models + (size_t)r * P->p;
        /* :388-390 */
        double sum_of_residuals_after = 0.0;
        for (uint64_t j = 0; j < inlier_number; ++j)
            sum_of_residuals_after += pgxo_residual(P->s->model_type, P->pts + (size_t)current_inliers[j] * P->d, fitted);
        /* :393-399 strictly smaller; setDescriptor leaves the preference vector as it was (stale) */
        int accepted = sum_of_residuals_after < sum_of_residuals_before, tie = 0;
        if (P->s->refit_tie_rtol > 0.0 && c->t->refit_accepted && c->t->refit_accepted[r] >= 0 &&
            fabs(sum_of_residuals_after - sum_of_residuals_before) <= P->s->refit_tie_rtol * fabs(sum_of_residuals_before)) {
            /* a numerical tie (progx_replay.h, refit_tie_rtol): the order of the additions decides, follow the recording */
            tie = accepted != (c->t->refit_accepted[r] != 0);
            accepted = c->t->refit_accepted[r] != 0;
        }
        if (accepted) {
            memcpy(models_->data[instance_idx].descriptor, fitted, (size_t)P->p * sizeof(double));
            *changed_ = 1;
        }
        emit(P->log, PGXR_EV_REFIT, (int64_t)instance_idx, (int64_t)inlier_number, 2 + accepted + 4 * tie, sum_of_residuals_before,
             sum_of_residuals_after);
    }
}

/* PEARL::rejectInstances, PEARL.h:275-315 */
static void pearl_reject_instances(pearl_t *P, model_vec *models_, int *changed_)
{
    for (int instance_idx = (int)models_->size - 1; instance_idx >= 0; --instance_idx) {
        const uint64_t inlier_number = P->instance_size[instance_idx];
        if (inlier_number < P->minimum_inlier_number) {
            /* :296 outliers.insert(...) */
            P->outliers_size += inlier_number;
            /* :300 points_per_instance.erase */
            free(P->points_per_instance[instance_idx]);
            memmove(P->points_per_instance + instance_idx, P->points_per_instance + instance_idx + 1,
                    (P->instance_number - (size_t)instance_idx - 1) * sizeof(int64_t *));
            memmove(P->instance_size + instance_idx, P->instance_size + instance_idx + 1,
                    (P->instance_number - (size_t)instance_idx - 1) * sizeof(uint64_t));
            --P->instance_number;
            /* :303 models_->erase */
            mv_erase(models_, (size_t)instance_idx);
            /* :307 */
            *changed_ = 1;
            emit(P->log, PGXR_EV_REJECT, instance_idx, (int64_t)inlier_number, 0, 0.0, 0.0);
        }
    }
}

/* PEARL::run, PEARL.h:405-472 */
static int pearl_run(pearl_t *P, model_vec *models_)
{
    size_t iteration_number = 0;
    double energy = DBL_MAX, previous_energy = -1.0;
    int model_parameters_changed = 0, model_rejected = 0, convergenve = 0;
    while (!convergenve && iteration_number++ < P->maximum_iteration_number) {
        /* :429-431 */
        const int initialize_with_previous_labeling = iteration_number > 1 && !model_rejected;
        const size_t models_before = models_->size;
        /* :434-439 */
        if (pearl_labeling(P, models_, initialize_with_previous_labeling, &energy) < 0) {
            snprintf(g_err, sizeof g_err, "labelling failed (spatial weight > 0 needs a graph)");
            P->cur->status = -1;
            return 0;
        }
        emit(P->log, PGXR_EV_PEARL_ITER, (int64_t)iteration_number, (int64_t)models_before, initialize_with_previous_labeling, energy, 0.0);
        /* :446, :450 */
        model_parameters_changed = 0;
        model_rejected = 0;
        /* :453-456 */
        pearl_parameter_estimation(P, models_, &model_parameters_changed);
        if (P->cur->status != 0) return 0;
        /* :458-460 */
        pearl_reject_instances(P, models_, &model_rejected);
        /* :463-467 */
        double delta = energy - previous_energy;
        double abs_delta;
        if (P->s->pearl_abs_int)   /* U-16: int abs(int) — the conversion truncates towards zero (out of range is UB upstream: treated as huge) */
            abs_delta = (delta > -2147483648.0 && delta < 2147483648.0) ? (double)abs((int)delta) : DBL_MAX;
        else
            abs_delta = fabs(delta);
        if (!model_rejected && !model_parameters_changed && abs_delta < P->epsilon && iteration_number > 1)
            convergenve = 1;
        emit(P->log, PGXR_EV_PEARL_END, (int64_t)iteration_number, model_parameters_changed * 2 + model_rejected,
             (int64_t)models_->size * 2 + convergenve, 0.0, 0.0);
        /* :469 */
        previous_energy = energy;
    }
    return 1;
}

/* PEARL::getLabeling(std::vector<size_t>&, size_t&), PEARL.h:218-247 */
static void pearl_get_labeling(const pearl_t *P, int64_t *labeling_, uint64_t *instance_number_)
{
    if (!P->engine) {
        for (int64_t i = 0; i < P->point_number; ++i) labeling_[i] = 0;
        *instance_number_ = 0;
        return;
    }
    *instance_number_ = 0;
    for (int64_t point_idx = 0; point_idx < P->point_number; ++point_idx) {
        const uint64_t label = (uint64_t)P->engine_labels[point_idx];
        labeling_[point_idx] = (int64_t)label;
        *instance_number_ = MAXV(*instance_number_, label);
    }
}

/* ProgressiveX::getPredictedUnseenInliers, progressive_x.h:495-513 */
static uint64_t get_predicted_unseen_inliers(uint64_t point_number, double one_minus_confidence_, uint64_t sample_size_,
                                             uint64_t iteration_number_, uint64_t inlier_number_of_compound_model_)
{
    const uint64_t unseen_point_number = point_number - inlier_number_of_compound_model_;   /* size_t: wraps */
    const double one_over_iteration_number = 1.0 / (double)iteration_number_;
    const double one_over_sample_size = 1.0 / (double)sample_size_;
    const double inlier_ratio = pow(1.0 - pow(one_minus_confidence_, one_over_iteration_number), one_over_sample_size);
    return (uint64_t)round((double)unseen_point_number * inlier_ratio);
}

int64_t pgxr_replay(const pgxr_settings *settings, const double *pts, int64_t n,
                    const int32_t *off, const int32_t *idx, const int32_t *mult,
                    const pgxr_trace *trace,
                    pgxr_event *events, int64_t max_events,
                    int64_t *labels_out, double *models_out, int32_t max_models, int32_t *models_n,
                    int64_t consumed[2])
{
    g_err[0] = 0;
    int d = 0, p = 0;
    if (!settings || !pts || n <= 0 || !trace || !events || !labels_out || !models_out || !models_n ||
        pgxo_model_dims(settings->model_type, &d, &p) != 0 || p > MAX_PARAM) {
        snprintf(g_err, sizeof g_err, "bad argument");
        return -1;
    }
    evlog log = {events, 0, max_events, 0};
    cursor cur = {trace, 0, 0, 0};

    /* ---- ProgressiveX::initialize, progressive_x.h:519-559 ---- */
    const uint64_t point_number = (uint64_t)n;
    int64_t *labeling = labels_out;                                     /* statistics.labeling.resize(point_number, 0) */
    for (int64_t i = 0; i < n; ++i) labeling[i] = 0;
    const double truncated_squared_threshold = 9.0 / 4.0 * settings->inlier_outlier_threshold * settings->inlier_outlier_threshold;
    double *compound_preference_vector = (double *)calloc((size_t)n, sizeof(double));
    pearl_t P;
    memset(&P, 0, sizeof P);
    P.s = settings; P.pts = pts; P.point_number = n; P.d = d; P.p = p; P.off = off; P.idx = idx; P.mult = mult;
    P.inlier_outlier_threshold = settings->inlier_outlier_threshold;          /* :528 */
    P.spatial_coherence_weight = settings->spatial_coherence_weight;          /* :529 */
    P.minimum_inlier_number = settings->minimum_number_of_inliers;            /* :530 */
    P.model_complexity_weight = (double)settings->minimum_number_of_inliers;  /* PEARL.h:144 */
    P.epsilon = 1e-5;                                                         /* PEARL.h:145 */
    P.maximum_iteration_number = (uint64_t)settings->pearl_maximum_iteration_number;   /* :532 */
    P.engine = 0;
    P.engine_labels = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    P.log = &log; P.cur = &cur;

    model_vec models = {NULL, 0, 0};
    size_t inliers_of_each_model_size = 0;     /* statistics.inliers_of_each_model.size() */

    /* ---- ProgressiveX::run, progressive_x.h:251-489 ---- */
    uint64_t number_of_ransac_iterations = 0, unaccepted_putative_instances = 0, unseen_inliers = point_number;
    int break_reason = 0;
    for (uint64_t current_iteration = 0; current_iteration < (uint64_t)settings->max_outer_iterations; ++current_iteration) {
        emit(&log, PGXR_EV_OUTER, (int64_t)current_iteration, 0, 0, 0.0, 0.0);
        /* :294-299 proposal_engine->run: from the trace */
        if (cur.next_proposal >= trace->n_proposals) {
            snprintf(g_err, sizeof g_err, "trace ran out of proposals at outer iteration %llu", (unsigned long long)current_iteration);
            cur.status = -2;
            break;
        }
        const int32_t q = cur.next_proposal++;
        /* :301-303 */
        if (trace->empty[q]) {
            emit(&log, PGXR_EV_PROPOSAL_EMPTY, 0, 0, 0, 0.0, 0.0);
            continue;
        }
        model_t putative_model;
        memset(&putative_model, 0, sizeof putative_model);
        memcpy(putative_model.descriptor, trace->models + (size_t)q * p, (size_t)p * sizeof(double));
        const int64_t *proposal_inliers = trace->inliers + trace->inlier_off[q];
        const uint64_t proposal_inlier_number = (uint64_t)(trace->inlier_off[q + 1] - trace->inlier_off[q]);
        /* :317-318 */
        number_of_ransac_iterations += trace->iterations[q];
        emit(&log, PGXR_EV_PROPOSAL, (int64_t)proposal_inlier_number, (int64_t)trace->iterations[q], (int64_t)number_of_ransac_iterations, 0.0, 0.0);

        /* ---- isPutativeModelValid, :565-591 ---- */
        int valid = 1, reason = 0;
        double tanimoto_similarity = NAN;
        {
            const uint64_t inlier_number = proposal_inlier_number;
            if (inlier_number < MAXV(settings->sample_size, settings->minimum_number_of_inliers)) {   /* :574 */
                valid = 0; reason = 1;
            } else {
                /* :578-579 setPreferenceVector (progx_model.h:70-87) */
                putative_model.preference_vector = (double *)malloc((size_t)n * sizeof(double));
                pgxo_preference(settings->model_type, pts, n, putative_model.descriptor, truncated_squared_threshold,
                                putative_model.preference_vector);
                /* :583-585 */
                double dot_product = 0.0, pref_sq = 0.0, comp_sq = 0.0;
                for (int64_t i = 0; i < n; ++i) {
                    dot_product += putative_model.preference_vector[i] * compound_preference_vector[i];
                    pref_sq += putative_model.preference_vector[i] * putative_model.preference_vector[i];
                    comp_sq += compound_preference_vector[i] * compound_preference_vector[i];
                }
                tanimoto_similarity = dot_product / (pref_sq + comp_sq - dot_product);
                /* :587 (a NaN similarity compares false: valid) */
                if (settings->maximum_tanimoto_similarity < tanimoto_similarity) {
                    valid = 0; reason = 2;
                }
            }
        }
        emit(&log, PGXR_EV_VALIDATION, valid, reason, 0, tanimoto_similarity, 0.0);
        /* :334-346 */
        if (!valid) {
            free(putative_model.preference_vector);
            ++unaccepted_putative_instances;
            emit(&log, PGXR_EV_UNACCEPTED, (int64_t)unaccepted_putative_instances, 0, 0, 0.0, 0.0);
            if (unaccepted_putative_instances == settings->max_proposal_number_without_change) {
                break_reason = 1;
                break;
            }
            continue;
        }
        /* :369 */
        mv_push(&models, &putative_model);
        /* :375-385 */
        if (models.size == 1) {
            ++inliers_of_each_model_size;                       /* inliers_of_each_model.emplace_back(...) */
            for (int64_t i = 0; i < n; ++i) labeling[i] = 1;
            for (uint64_t j = 0; j < proposal_inlier_number; ++j)
                labeling[proposal_inliers[j]] = 0;
            emit(&log, PGXR_EV_SINGLE_MODEL, (int64_t)proposal_inlier_number, 0, 0, 0.0, 0.0);
        } else {
            /* :390-403 */
            pearl_run(&P, &models);
            if (cur.status != 0) break;
            uint64_t model_number = 0;
            pearl_get_labeling(&P, labeling, &model_number);
            emit(&log, PGXR_EV_LABELING, (int64_t)model_number, (int64_t)models.size, 0, 0.0, 0.0);
        }
        /* ---- updateCompoundModel, :597-624 ---- */
        if (models.size != 0) {
            for (int64_t i = 0; i < n; ++i) compound_preference_vector[i] = 0.0;      /* :604 */
            for (size_t k = 0; k < models.size; ++k)                                   /* :608 */
                for (int64_t point_idx = 0; point_idx < n; ++point_idx)               /* :612 (the residual of :615 is unused) */
                    compound_preference_vector[point_idx] =
                        MAXV(compound_preference_vector[point_idx], models.data[k].preference_vector[point_idx]);   /* :620-621 the STORED vector */
        }
        {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += compound_preference_vector[i];
            emit(&log, PGXR_EV_COMPOUND, (int64_t)models.size, 0, 0, s, 0.0);
        }
        /* :447-457 */
        uint64_t covered;
        if (models.size == 1)
            covered = (uint64_t)inliers_of_each_model_size;     /* the NUMBER of stored inlier sets, not an inlier count */
        else
            covered = point_number - P.outliers_size;
        unseen_inliers = get_predicted_unseen_inliers(point_number, settings->one_minus_confidence, settings->sample_size,
                                                      number_of_ransac_iterations, covered);
        emit(&log, PGXR_EV_UNSEEN, (int64_t)covered, (int64_t)unseen_inliers, 0, 0.0, 0.0);
        /* :468-469 */
        if (unseen_inliers < settings->minimum_number_of_inliers) {
            break_reason = 2;
            break;
        }
        /* :472-473 */
        if (models.size >= settings->maximum_model_number) {
            break_reason = 3;
            break;
        }
    }
    if (cur.status == 0)
        emit(&log, PGXR_EV_BREAK, break_reason, 0, 0, 0.0, 0.0);

    int64_t ret = log.n;
    if (cur.status != 0) ret = cur.status;
    else if (log.overflow || (int64_t)models.size > (int64_t)max_models) {
        snprintf(g_err, sizeof g_err, "capacity: %lld events (max %lld), %zu models (max %d)", (long long)log.n, (long long)max_events,
                 models.size, max_models);
        ret = -5;
    } else {
        for (size_t k = 0; k < models.size; ++k)
            memcpy(models_out + k * (size_t)p, models.data[k].descriptor, (size_t)p * sizeof(double));
        *models_n = (int32_t)models.size;
    }
    if (consumed) { consumed[0] = cur.next_proposal; consumed[1] = cur.next_refit; }
    while (models.size) mv_erase(&models, models.size - 1);
    free(models.data);
    pearl_clear_instances(&P);
    free(P.engine_labels);
    free(compound_preference_vector);
    return ret;
}
