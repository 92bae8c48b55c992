/*
 * pgx_oracle.c — CPU oracle for the Progressive-X hot path (see pgx_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into, or called from, the product path.
 * PARITY UNPINNED (no reference tests / golden vectors / buildable reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * Reference paths below are relative to /root/reference/src/pyprogressivex/.
 */
#include "pgx_oracle.h"
#include "bk_maxflow.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* OpenCV's macros, which the reference picks up through <opencv2/core/core.hpp>
 * (progx_model.h:36): MIN(a,b) ((a) > (b) ? (b) : (a)),  MAX(a,b) ((a) < (b) ? (b) : (a)). */
#define CV_MIN(a, b) ((a) > (b) ? (b) : (a))
#define CV_MAX(a, b) ((a) < (b) ? (b) : (a))

int pgxo_model_dims(int model_type, int *point_dim, int *param_dim)
{
    static const int pd[6] = {2, 4, 4, 5, 4, 4};
    static const int md[6] = {3, 9, 9, 12, 3, 18};
    if (model_type < 0 || model_type > 5) return -1;
    if (point_dim) *point_dim = pd[model_type];
    if (param_dim) *param_dim = md[model_type];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Residuals
 * ---------------------------------------------------------------------------------------- */

/* U-4 [UPSTREAM-MEMORY]: Default2DLineEstimator (progressivex_python.cpp:489); model (a,b,c)
 * with a^2+b^2=1 as returned at progressivex_python.cpp:529-531.  r = |a x + b y + c|. */
static double line_residual(const double *p, const double *m)
{
    const double x = p[0], y = p[1];
    return fabs(m[0] * x + m[1] * y + m[2]);
}

/* U-1 [UPSTREAM-MEMORY]: DefaultHomographyEstimator (progressivex_python.cpp:252); H row-major as
 * flattened at progressivex_python.cpp:292-300.  One-way forward transfer error. */
static double homography_sq(const double *p, const double *h)
{
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const double t1 = h[0] * x1 + h[1] * y1 + h[2];
    const double t2 = h[3] * x1 + h[4] * y1 + h[5];
    const double t3 = h[6] * x1 + h[7] * y1 + h[8];
    const double d1 = x2 - (t1 / t3);
    const double d2 = y2 - (t2 / t3);
    return d1 * d1 + d2 * d2;
}

/* Symmetric transfer error (north-star wording); model = [H | H^-1]; forward + backward. */
static double homography_sym_sq(const double *p, const double *h)
{
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const double *g = h + 9;
    const double t1 = h[0] * x1 + h[1] * y1 + h[2];
    const double t2 = h[3] * x1 + h[4] * y1 + h[5];
    const double t3 = h[6] * x1 + h[7] * y1 + h[8];
    const double d1 = x2 - (t1 / t3);
    const double d2 = y2 - (t2 / t3);
    const double s1 = g[0] * x2 + g[1] * y2 + g[2];
    const double s2 = g[3] * x2 + g[4] * y2 + g[5];
    const double s3 = g[6] * x2 + g[7] * y2 + g[8];
    const double e1 = x1 - (s1 / s3);
    const double e2 = y1 - (s2 / s3);
    return (d1 * d1 + d2 * d2) + (e1 * e1 + e2 * e2);
}

/* U-2 [UPSTREAM-MEMORY]: DefaultFundamentalMatrixEstimator (progressivex_python.cpp:616);
 * F row-major (progressivex_python.cpp:654-662); squared Sampson distance. */
static double fundamental_sq(const double *p, const double *f)
{
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    /* F^T x2 */
    const double rxc = f[0] * x2 + f[3] * y2 + f[6];
    const double ryc = f[1] * x2 + f[4] * y2 + f[7];
    const double rwc = f[2] * x2 + f[5] * y2 + f[8];
    const double r = x1 * rxc + y1 * ryc + rwc;
    /* F x1 */
    const double rx = f[0] * x1 + f[1] * y1 + f[2];
    const double ry = f[3] * x1 + f[4] * y1 + f[5];
    return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
}

/* U-3 [UPSTREAM-MEMORY]: DefaultPnPEstimator (progressivex_python.cpp:119); P=[R|t] row-major 3x4
 * (progressivex_python.cpp:156-167); data row (u_n, v_n, X, Y, Z) (progressivex_python.cpp:88-92).
 * Squared reprojection error in normalised image coordinates. */
static double pnp_sq(const double *p, const double *P)
{
    const double u = p[0], v = p[1], X = p[2], Y = p[3], Z = p[4];
    const double px = P[0] * X + P[1] * Y + P[2] * Z + P[3];
    const double py = P[4] * X + P[5] * Y + P[6] * Z + P[7];
    const double pz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
    const double du = u - (px / pz);
    const double dv = v - (py / pz);
    return du * du + dv * dv;
}

/* vanishing_point_estimator.h:166-189 (in-tree, exact operation order). */
static double vp_residual(const double *p, const double *d)
{
    const double xs = p[0], ys = p[1], xe = p[2], ye = p[3];
    double lx, ly, lz;
    const double mx = (xs + xe) / 2.0, my = (ys + ye) / 2.0;
    lx = my * d[2] - d[1];
    ly = -(mx * d[2] - d[0]);
    lz = mx * d[1] - my * d[0];
    return fabs(lx * xs + ly * ys + lz) / sqrt(lx * lx + ly * ly);
}

/* U-14: the F estimator's symmetric-epipolar support (restated from the literature; see include/pgx.h pgx_epipolar_support).
 * out[0] = Sampson inliers (fundamental_sq < T2, strict as the scorer), out[1] = those with r^2 (1/|F x1|^2 + 1/|F^T x2|^2) < S2. */
void pgxo_epipolar_support(const double *pts, int64_t n, const double *f, double T2, double S2, int64_t *out)
{
    int64_t inl = 0, sup = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = pts + i * 4;
        const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
        const double rxc = f[0] * x2 + f[3] * y2 + f[6];
        const double ryc = f[1] * x2 + f[4] * y2 + f[7];
        const double rwc = f[2] * x2 + f[5] * y2 + f[8];
        const double r = x1 * rxc + y1 * ryc + rwc;
        const double rx = f[0] * x1 + f[1] * y1 + f[2];
        const double ry = f[3] * x1 + f[4] * y1 + f[5];
        const double sym = (r * r) * (1.0 / (rx * rx + ry * ry) + 1.0 / (rxc * rxc + ryc * ryc));
        if (fundamental_sq(p, f) < T2) {
            ++inl;
            if (sym < S2) ++sup;
        }
    }
    out[0] = inl;
    out[1] = sup;
}

double pgxo_squared_residual(int model_type, const double *pt, const double *model)
{
    double r;
    switch (model_type) {
    case PGXO_LINE2D: r = line_residual(pt, model); return r * r;
    case PGXO_HOMOGRAPHY: return homography_sq(pt, model);
    case PGXO_FUNDAMENTAL: return fundamental_sq(pt, model);
    case PGXO_PNP: return pnp_sq(pt, model);
    case PGXO_VANISHING_POINT: /* vanishing_point_estimator.h:134-140 */
        r = vp_residual(pt, model); return r * r;
    case PGXO_HOMOGRAPHY_SYM: return homography_sym_sq(pt, model);
    default: return NAN;
    }
}

/* Unsquared residual used by PEARL::parameterEstimation (PEARL.h:371,390).  For the estimators whose
 * source is absent it is restated as sqrt(squaredResidual) [UPSTREAM-MEMORY]. */
double pgxo_residual(int model_type, const double *pt, const double *model)
{
    switch (model_type) {
    case PGXO_LINE2D: return line_residual(pt, model);
    case PGXO_VANISHING_POINT: return vp_residual(pt, model);
    default: return sqrt(pgxo_squared_residual(model_type, pt, model));
    }
}

void pgxo_squared_residuals(int model_type, const double *pts, int64_t n, const double *model, double *out)
{
    int d = 0;
    pgxo_model_dims(model_type, &d, NULL);
    for (int64_t i = 0; i < n; ++i) out[i] = pgxo_squared_residual(model_type, pts + i * d, model);
}

/* ------------------------------------------------------------------------------------------
 * a1  scoring_function_with_compound_model.h:61-125
 * ---------------------------------------------------------------------------------------- */
void pgxo_score(int model_type, const double *pts, int64_t n, const double *models, int M,
                double T2, const double *compound, int has_compound, int exponent,
                const int64_t *best_inlier_number,
                int64_t *counts, double *values, double *shared, double *scores, uint64_t *masks)
{
    int d = 0, pdim = 0;
    pgxo_model_dims(model_type, &d, &pdim);
    const int64_t words = (n + 63) / 64;
    double *pref = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int m = 0; m < M; ++m) {
        const double *model = models + (int64_t)m * pdim;
        const uint64_t best = best_inlier_number ? (uint64_t)best_inlier_number[m] : 0;
        uint64_t inl = 0;
        double value = 0.0;
        int interrupted = 0;
        if (masks) memset(masks + (int64_t)m * words, 0, sizeof(uint64_t) * (size_t)words);
        memset(pref, 0, sizeof(double) * (size_t)n); /* :75 */
        for (int64_t i = 0; i < n; ++i) {            /* :78 */
            const double sq = pgxo_squared_residual(model_type, pts + i * d, model); /* :81 */
            if (sq < T2) {                            /* :85 strict */
                if (masks) masks[(int64_t)m * words + (i >> 6)] |= (uint64_t)1 << (i & 63); /* :88 */
                ++inl;                                /* :91 */
                const double sv = CV_MAX(0.0, 1.0 - sq / T2); /* :94 */
                value += sv;                          /* :97 */
                pref[i] = sv;                         /* :100 */
            }
            /* :105  point_number - point_idx + inlier_number < best.inlier_number (size_t arithmetic) */
            if ((uint64_t)n - (uint64_t)i + inl < best) { interrupted = 1; break; }
        }
        if (interrupted) { /* :106 returns Score() */
            counts[m] = 0; values[m] = 0.0; shared[m] = 0.0; scores[m] = 0.0;
            if (masks) memset(masks + (int64_t)m * words, 0, sizeof(uint64_t) * (size_t)words);
            continue;
        }
        double sh = 0.0;
        double score = value;
        if (has_compound) {                           /* :110 compound_model->size() > 0 */
            for (int64_t i = 0; i < n; ++i)           /* :115-117 */
                sh += CV_MIN(compound[i], pref[i]);
            score -= pow(sh, (double)exponent);       /* :120  std::pow(double,int) */
        }
        counts[m] = (int64_t)inl; values[m] = value; shared[m] = sh; scores[m] = score;
    }
    free(pref);
}

/* ------------------------------------------------------------------------------------------
 * a2  progx_model.h:70-87 ; a3 progressive_x.h:583-588 ; a4 progressive_x.h:604-623
 * ---------------------------------------------------------------------------------------- */
void pgxo_preference(int model_type, const double *pts, int64_t n, const double *model, double T2,
                     double *pref)
{
    int d = 0;
    pgxo_model_dims(model_type, &d, NULL);
    for (int64_t i = 0; i < n; ++i) {
        const double sq = pgxo_squared_residual(model_type, pts + i * d, model);
        const double v = 1.0 - sq / T2;
        pref[i] = CV_MAX(0, v); /* progx_model.h:85  MAX(0, double) */
    }
}

void pgxo_tanimoto_terms(const double *pref, const double *compound, int64_t n,
                         double *dot, double *pref_sqnorm, double *comp_sqnorm)
{
    double d = 0.0, a = 0.0, b = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        d += pref[i] * compound[i];
        a += pref[i] * pref[i];
        b += compound[i] * compound[i];
    }
    *dot = d; *pref_sqnorm = a; *comp_sqnorm = b;
}

int pgxo_is_valid_tanimoto(double dot, double pref_sqnorm, double comp_sqnorm, double max_tanimoto,
                           double *tanimoto)
{
    const double t = dot / (pref_sqnorm + comp_sqnorm - dot); /* progressive_x.h:584-585 */
    if (tanimoto) *tanimoto = t;
    if (max_tanimoto < t) return 0;                            /* :587  (NaN => valid) */
    return 1;
}

void pgxo_compound_max(const double *prefs, int K, int64_t n, double *compound)
{
    for (int64_t i = 0; i < n; ++i) compound[i] = 0.0;       /* :604 */
    for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n; ++i)
            compound[i] = CV_MAX(compound[i], prefs[(int64_t)k * n + i]); /* :620-621 */
}

/* a5  progressive_x.h:495-513 */
uint64_t pgxo_predicted_unseen_inliers(double one_minus_conf, uint64_t sample_size,
                                       uint64_t iteration_number, uint64_t covered, uint64_t point_number)
{
    const uint64_t unseen = point_number - covered;
    const double one_over_it = 1.0 / (double)iteration_number;
    const double one_over_s = 1.0 / (double)sample_size;
    const double ratio = pow(1.0 - pow(one_minus_conf, one_over_it), one_over_s);
    return (uint64_t)round((double)unseen * ratio);
}

/* ------------------------------------------------------------------------------------------
 * a6  PEARL.h:82-128 (dataEnergyFunctor), thresholds PEARL.h:48-51
 * ---------------------------------------------------------------------------------------- */
void pgxo_unary(int model_type, const double *pts, int64_t n, const double *models, int K,
                double threshold, double lambda, double *D)
{
    int d = 0, pdim = 0;
    pgxo_model_dims(model_type, &d, &pdim);
    const double T2 = 9.0 / 4.0 * threshold * threshold; /* :51 */
    const double oml = 1.0 - lambda;                      /* :48 */
    const int L = K + 1;
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < K; ++k) {
            const double sq = pgxo_squared_residual(model_type, pts + i * d, models + (int64_t)k * pdim);
            double c;
            if (sq > T2) c = 2.0 * oml;                   /* :123-124 */
            else c = oml * sq / T2;                       /* :126-127 */
            D[i * L + k] = c;
        }
        D[i * L + K] = oml;                               /* :100-101 */
    }
}

/* Fixed-point quantisation used by the min-cut: round-to-nearest-even multiple of
 * 2^-32; NaN (degenerate model) is priced like "beyond the threshold" by the caller passing 2(1-l). */
int64_t pgxo_quantize(double x)
{
    return (int64_t)nearbyint(x * 4294967296.0);
}

void pgxo_unary_q(int model_type, const double *pts, int64_t n, const double *models, int K,
                  double threshold, double lambda, int64_t *Dq)
{
    const int L = K + 1;
    double *D = (double *)malloc(sizeof(double) * (size_t)(n * L > 0 ? n * L : 1));
    pgxo_unary(model_type, pts, n, models, K, threshold, lambda, D);
    const double far = 2.0 * (1.0 - lambda);
    for (int64_t j = 0; j < n * L; ++j) {
        double c = D[j];
        if (c != c) c = far;
        Dq[j] = pgxo_quantize(c);
    }
    free(D);
}

/* ------------------------------------------------------------------------------------------
 * Max-flow (Dinic, int64 capacities) — stands in for the BK solver inside the absent GCoptimization
 * library.  With integer capacities the *minimal sink side* (nodes that can still reach t in the
 * residual graph of ANY maximum flow) is unique, which is what BK's what_segment(default=SOURCE)
 * returns [U-5]; any correct max-flow therefore reproduces the labelling bit for bit.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int nn;
    int64_t na, cap_arcs;
    int32_t *head, *nxt, *to, *level, *it;
    int64_t *cap;
} dinic_t;

static void dn_init(dinic_t *g, int nn, int64_t max_arcs)
{
    g->nn = nn; g->na = 0; g->cap_arcs = max_arcs;
    g->head = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    g->level = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    g->it = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    g->nxt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_arcs > 0 ? max_arcs : 1));
    g->to = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_arcs > 0 ? max_arcs : 1));
    g->cap = (int64_t *)malloc(sizeof(int64_t) * (size_t)(max_arcs > 0 ? max_arcs : 1));
    for (int i = 0; i < nn; ++i) g->head[i] = -1;
}
static void dn_free(dinic_t *g)
{
    free(g->head); free(g->level); free(g->it); free(g->nxt); free(g->to); free(g->cap);
}
/* adds arc u->v (cap c) and v->u (cap rc) as a pair (a, a^1) */
static void dn_add(dinic_t *g, int u, int v, int64_t c, int64_t rc)
{
    int64_t a = g->na;
    g->to[a] = v; g->cap[a] = c; g->nxt[a] = g->head[u]; g->head[u] = (int32_t)a;
    g->to[a + 1] = u; g->cap[a + 1] = rc; g->nxt[a + 1] = g->head[v]; g->head[v] = (int32_t)(a + 1);
    g->na += 2;
}
static int dn_bfs(dinic_t *g, int s, int t, int32_t *queue)
{
    for (int i = 0; i < g->nn; ++i) g->level[i] = -1;
    int qh = 0, qt = 0;
    queue[qt++] = s; g->level[s] = 0;
    while (qh < qt) {
        int u = queue[qh++];
        for (int32_t a = g->head[u]; a != -1; a = g->nxt[a]) {
            int v = g->to[a];
            if (g->cap[a] > 0 && g->level[v] < 0) { g->level[v] = g->level[u] + 1; queue[qt++] = v; }
        }
    }
    return g->level[t] >= 0;
}
static int64_t dn_maxflow(dinic_t *g, int s, int t)
{
    int64_t flow = 0;
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)g->nn);
    int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)g->nn); /* arc stack */
    while (dn_bfs(g, s, t, queue)) {
        memcpy(g->it, g->head, sizeof(int32_t) * (size_t)g->nn);
        int depth = 0;
        int u = s;
        for (;;) {
            if (u == t) {
                int64_t bott = INT64_MAX;
                for (int k = 0; k < depth; ++k) if (g->cap[path[k]] < bott) bott = g->cap[path[k]];
                int first_sat = depth;
                for (int k = 0; k < depth; ++k) {
                    g->cap[path[k]] -= bott; g->cap[path[k] ^ 1] += bott;
                    if (g->cap[path[k]] == 0 && k < first_sat) first_sat = k;
                }
                flow += bott;
                depth = first_sat;
                u = (depth == 0) ? s : g->to[path[depth - 1]];
                continue;
            }
            int advanced = 0;
            while (g->it[u] != -1) {
                int32_t a = g->it[u];
                int v = g->to[a];
                if (g->cap[a] > 0 && g->level[v] == g->level[u] + 1) {
                    path[depth++] = a; u = v; advanced = 1; break;
                }
                g->it[u] = g->nxt[a];
            }
            if (advanced) continue;
            /* retreat */
            g->level[u] = -1;
            if (depth == 0) break;
            int32_t a = path[--depth];
            u = g->to[a ^ 1];
            g->it[u] = g->nxt[a];
        }
    }
    free(queue); free(path);
    return flow;
}
/* reach[v]=1 iff v can reach t in the residual graph (reverse BFS from t) */
static void dn_sink_side(dinic_t *g, int t, uint8_t *reach)
{
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)g->nn);
    memset(reach, 0, (size_t)g->nn);
    int qh = 0, qt = 0;
    queue[qt++] = t; reach[t] = 1;
    while (qh < qt) {
        int v = queue[qh++];
        for (int32_t a = g->head[v]; a != -1; a = g->nxt[a]) {
            /* arc a is v->w ; its mate a^1 is w->v : w reaches v if cap[a^1] > 0 */
            int w = g->to[a];
            if (!reach[w] && g->cap[a ^ 1] > 0) { reach[w] = 1; queue[qt++] = w; }
        }
    }
    free(queue);
}

int64_t pgxo_maxflow(int nnodes, int64_t narcs, const int32_t *from, const int32_t *to,
                     const int64_t *cap, int s, int t, uint8_t *sink_side)
{
    dinic_t g;
    dn_init(&g, nnodes, 2 * narcs);
    for (int64_t a = 0; a < narcs; ++a) dn_add(&g, from[a], to[a], cap[a], 0);
    int64_t f = dn_maxflow(&g, s, t);
    if (sink_side) dn_sink_side(&g, t, sink_side);
    dn_free(&g);
    return f;
}

/* ------------------------------------------------------------------------------------------
 * a8 / a19  alpha-expansion [U-5]: energy = sum_i D[i][l_i] + sum_{pairs} w_ij [l_i != l_j]
 *           + h * #labels in use  (PEARL.h:519-529, 76-78; label cost incl. the outlier label).
 * ---------------------------------------------------------------------------------------- */
int64_t pgxo_energy(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                    const int32_t *mult, int64_t lambda_q, int64_t h_q, const int32_t *labels)
{
    int64_t e = 0;
    int64_t *cnt = (int64_t *)calloc((size_t)L, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        e += Dq[i * L + labels[i]];
        cnt[labels[i]]++;
        if (off && lambda_q > 0)
            for (int32_t a = off[i]; a < off[i + 1]; ++a) {
                int32_t j = idx[a];
                if (j < i && labels[j] != labels[i]) e += lambda_q * (int64_t)mult[a];
            }
    }
    for (int l = 0; l < L; ++l) if (cnt[l] > 0) e += h_q;
    free(cnt);
    return e;
}

#define PGXO_INF ((int64_t)1 << 60)

/* One expansion move on label alpha: optimal binary move, ties resolved towards alpha
 * (BK: free nodes default to SOURCE = take alpha) [U-5].  Label costs via one auxiliary node per
 * label (Delong et al., IJCV 2012): labels in use other than alpha pay h unless all their sites move;
 * alpha pays h if it is unused and any site moves. */
static int expand_alpha_impl(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                             const int32_t *mult, int64_t lambda_q, int64_t h_q, int alpha, int32_t *labels,
                             int64_t *flow_value, int use_bk)
{
    int64_t *cnt = (int64_t *)calloc((size_t)L, sizeof(int64_t));
    int32_t *var = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t na = 0;
    for (int64_t i = 0; i < n; ++i) {
        cnt[labels[i]]++;
        var[i] = (labels[i] != alpha) ? (int32_t)na++ : -1;
    }
    if (flow_value) *flow_value = 0;
    if (na == 0) { free(cnt); free(var); return 0; }

    const int use_pair = (off != NULL && lambda_q > 0);
    const int use_lc = (h_q > 0);
    /* aux nodes */
    int32_t *hub_of_label = (int32_t *)malloc(sizeof(int32_t) * (size_t)L);
    int nhub = 0;
    for (int l = 0; l < L; ++l) {
        hub_of_label[l] = -1;
        if (!use_lc) continue;
        if (l == alpha) { if (cnt[l] == 0) hub_of_label[l] = (int32_t)(na + nhub++); }
        else if (cnt[l] > 0) hub_of_label[l] = (int32_t)(na + nhub++);
    }
    const int S = (int)(na + nhub), T = S + 1, nn = T + 1;
    int64_t narcs_est = 2 * na /* t-links */ + 2 * (int64_t)nhub;
    if (use_pair) narcs_est += off[n];
    if (use_lc) narcs_est += 2 * na * 2;
    /* the same network for either solver: EDGE(u, v, c, rc) between variable / hub nodes, FROM_S(v, c), TO_T(v, c) */
    dinic_t g;
    bk_graph *bk = NULL;
    if (use_bk) bk = bk_create((int)(na + nhub), narcs_est + 8);
    else dn_init(&g, nn, 2 * narcs_est + 16);
#define EDGE(u, v, c, rc) do { if (use_bk) bk_add_edge(bk, (u), (v), (c), (rc)); else dn_add(&g, (u), (v), (c), (rc)); } while (0)
#define FROM_S(v, c) do { if (use_bk) bk_add_tweights(bk, (v), (c), 0); else dn_add(&g, S, (v), (c), 0); } while (0)
#define TO_T(v, c) do { if (use_bk) bk_add_tweights(bk, (v), 0, (c)); else dn_add(&g, (v), T, (c), 0); } while (0)

    for (int64_t i = 0; i < n; ++i) {
        if (var[i] < 0) continue;
        const int li = labels[i];
        int64_t keep = Dq[i * L + li];      /* paid when x_i = 1 (sink side, keeps its label) */
        const int64_t take = Dq[i * L + alpha]; /* paid when x_i = 0 (source side, takes alpha) */
        if (use_pair)
            for (int32_t a = off[i]; a < off[i + 1]; ++a) {
                const int32_t j = idx[a];
                const int64_t w = lambda_q * (int64_t)mult[a];
                if (var[j] < 0) keep += w;                   /* neighbour already alpha */
                else if (i < j) {
                    if (labels[j] == li) EDGE(var[i], var[j], w, w);
                    else { keep += w; EDGE(var[i], var[j], w, 0); } /* Kolmogorov-Zabih form */
                }
            }
        FROM_S(var[i], keep);
        TO_T(var[i], take);
        if (use_lc) {
            if (hub_of_label[li] >= 0) EDGE(hub_of_label[li], var[i], PGXO_INF, 0);
            if (hub_of_label[alpha] >= 0) EDGE(var[i], hub_of_label[alpha], PGXO_INF, 0);
        }
    }
    if (use_lc)
        for (int l = 0; l < L; ++l) {
            if (hub_of_label[l] < 0) continue;
            if (l == alpha) TO_T(hub_of_label[l], h_q);
            else FROM_S(hub_of_label[l], h_q);
        }
#undef EDGE
#undef FROM_S
#undef TO_T
    int changed = 0;
    if (use_bk) {
        /* BK: what_segment(i, default = SOURCE): sites in the sink tree keep their label, everything else takes alpha [U-5] */
        const int64_t f = bk_maxflow(bk);
        if (flow_value) *flow_value = f;
        for (int64_t i = 0; i < n; ++i)
            if (var[i] >= 0 && !bk_in_sink_tree(bk, var[i])) { labels[i] = alpha; ++changed; }
        bk_destroy(bk);
    } else {
        const int64_t f = dn_maxflow(&g, S, T);
        if (flow_value) *flow_value = f;
        uint8_t *reach = (uint8_t *)malloc((size_t)nn);
        dn_sink_side(&g, T, reach);
        for (int64_t i = 0; i < n; ++i)
            if (var[i] >= 0 && !reach[var[i]]) { labels[i] = alpha; ++changed; }
        free(reach); dn_free(&g);
    }
    free(hub_of_label); free(var); free(cnt);
    return changed;
}

int pgxo_expand_alpha(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                      const int32_t *mult, int64_t lambda_q, int64_t h_q, int alpha, int32_t *labels,
                      int64_t *flow_value)
{
    return expand_alpha_impl(n, L, Dq, off, idx, mult, lambda_q, h_q, alpha, labels, flow_value, 0);
}

/* The same move solved by Boykov-Kolmogorov (bk_maxflow.c): the solver family GCoptimization uses behind PEARL.h:550.  Same
 * labels as the Dinic form by uniqueness of the minimal sink side; exists for the CPU labelling baseline (bench.py). */
int pgxo_expand_alpha_bk(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                         const int32_t *mult, int64_t lambda_q, int64_t h_q, int alpha, int32_t *labels,
                         int64_t *flow_value)
{
    return expand_alpha_impl(n, L, Dq, off, idx, mult, lambda_q, h_q, alpha, labels, flow_value, 1);
}

/* GCO-v3 "standard cycles" loop [U-5] as driven by PEARL.h:550-551 (max 1000 cycles). */
int pgxo_expansion(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                   const int32_t *mult, int64_t lambda_q, int64_t h_q, int32_t *labels, int max_cycles,
                   int64_t *energy_q, int *cycles)
{
    int64_t new_e = pgxo_energy(n, L, Dq, off, idx, mult, lambda_q, h_q, labels);
    int64_t old_e = new_e + 1;
    int c = 0;
    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        if (new_e == old_e) break;
        old_e = new_e;
        for (int alpha = 0; alpha < L; ++alpha)
            pgxo_expand_alpha(n, L, Dq, off, idx, mult, lambda_q, h_q, alpha, labels, NULL);
        new_e = pgxo_energy(n, L, Dq, off, idx, mult, lambda_q, h_q, labels);
        c = cycle;
    }
    if (energy_q) *energy_q = new_e;
    if (cycles) *cycles = c;
    return 0;
}

/* pgxo_expansion with every move solved by BK; mincuts (may be NULL) counts the moves that built a network */
int pgxo_expansion_bk(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                      const int32_t *mult, int64_t lambda_q, int64_t h_q, int32_t *labels, int max_cycles,
                      int64_t *energy_q, int *cycles, int64_t *mincuts)
{
    int64_t new_e = pgxo_energy(n, L, Dq, off, idx, mult, lambda_q, h_q, labels);
    int64_t old_e = new_e + 1, cuts = 0;
    int c = 0;
    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        if (new_e == old_e) break;
        old_e = new_e;
        for (int alpha = 0; alpha < L; ++alpha) {
            pgxo_expand_alpha_bk(n, L, Dq, off, idx, mult, lambda_q, h_q, alpha, labels, NULL);
            ++cuts;
        }
        new_e = pgxo_energy(n, L, Dq, off, idx, mult, lambda_q, h_q, labels);
        c = cycle;
    }
    if (energy_q) *energy_q = new_e;
    if (cycles) *cycles = c;
    if (mincuts) *mincuts = cuts;
    return 0;
}

/* plain s-t max-flow by BK on an explicit arc list (cross-check against pgxo_maxflow / scipy) */
int64_t pgxo_maxflow_bk(int nnodes, int64_t narcs, const int32_t *from, const int32_t *to,
                        const int64_t *cap, int s, int t, uint8_t *sink_side)
{
    bk_graph *g = bk_create(nnodes, narcs + 1);
    for (int64_t a = 0; a < narcs; ++a) {
        if (from[a] == s && to[a] != t) bk_add_tweights(g, to[a], cap[a], 0);
        else if (to[a] == t && from[a] != s) bk_add_tweights(g, from[a], 0, cap[a]);
        else if (from[a] != s && from[a] != t && to[a] != s && to[a] != t) bk_add_edge(g, from[a], to[a], cap[a], 0);
    }
    int64_t f = bk_maxflow(g);
    for (int64_t a = 0; a < narcs; ++a) if (from[a] == s && to[a] == t) f += cap[a];
    if (sink_side) {
        for (int i = 0; i < nnodes; ++i) sink_side[i] = (uint8_t)bk_in_sink_tree(g, i);
        sink_side[t] = 1; sink_side[s] = 0;
    }
    bk_destroy(g);
    return f;
}

/* ------------------------------------------------------------------------------------------
 * U-8  GCO-v3's special cases for an energy WITHOUT smooth costs (PEARL never calls setSmoothCost / setNeighbors when
 * spatial_coherence_weight == 0: /root/reference/src/pyprogressivex/include/PEARL.h:523-536, and then calls
 * expansion(): :550-551).  GCoptimization::expansion() starts with solveSpecialCases(), which for "data costs only"
 * assigns every site its cheapest label and for "data costs + per-label costs" runs solveGreedy(): the greedy
 * uncapacitated-facility-location heuristic.  The GCO sources are absent from the snapshot; restated from the
 * published algorithm [UPSTREAM-MEMORY]:
 *   every site starts unassigned at a prohibitive cost BIG; repeat: for every label not opened yet,
 *   delta(l) = h + sum_i min(0, D[i][l] - e_i); open the label with the most negative delta (ties: lowest index) and
 *   move to it every site it serves strictly cheaper; stop when no delta is negative.  The given labelling is ignored.
 * Integers throughout (D, h are 2^-32 fixed point), so the decisions are exact and CPU == GPU bit for bit.
 * h == 0: per-site argmin, first minimum.  Returns the number of labels opened; energy as pgxo_energy computes it.
 * ---------------------------------------------------------------------------------------- */
int pgxo_greedy_labeling(int64_t n, int L, const int64_t *Dq, int64_t h_q, int32_t *labels, int64_t *energy_q)
{
    const int64_t BIG = (int64_t)1 << 35; /* > every unary cost (<= 2^33 from pgxo_unary_q); n * BIG < 2^62 for n < 2^27 */
    int opened = 0;
    if (h_q <= 0) {
        for (int64_t i = 0; i < n; ++i) {
            int best = 0;
            for (int l = 1; l < L; ++l) if (Dq[i * L + l] < Dq[i * L + best]) best = l;
            labels[i] = best;
        }
        uint8_t *seen = (uint8_t *)calloc((size_t)L, 1);
        for (int64_t i = 0; i < n; ++i) if (!seen[labels[i]]) { seen[labels[i]] = 1; ++opened; }
        free(seen);
    } else {
        int64_t *e = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        uint8_t *open = (uint8_t *)calloc((size_t)L, 1);
        for (int64_t i = 0; i < n; ++i) { e[i] = BIG; labels[i] = 0; }
        for (;;) {
            int best = -1;
            int64_t best_delta = 0;
            for (int l = 0; l < L; ++l) {
                if (open[l]) continue;
                int64_t delta = h_q;
                for (int64_t i = 0; i < n; ++i) {
                    const int64_t d = Dq[i * L + l] - e[i];
                    if (d < 0) delta += d;
                }
                if (delta < best_delta) { best_delta = delta; best = l; }
            }
            if (best < 0) break;
            open[best] = 1;
            ++opened;
            for (int64_t i = 0; i < n; ++i)
                if (Dq[i * L + best] < e[i]) { e[i] = Dq[i * L + best]; labels[i] = best; }
        }
        free(e); free(open);
    }
    if (energy_q) *energy_q = pgxo_energy(n, L, Dq, NULL, NULL, NULL, 0, h_q, labels);
    return opened;
}

/* ------------------------------------------------------------------------------------------
 * GC-RANSAC's inlier/outlier labelling (gcransac::GCRANSAC::labeling; the graph-cut-ransac sources are
 * absent from the snapshot - call site /root/reference/src/pyprogressivex/include/progressive_x.h:294-299 - restated
 * from memory of upstream [U-12]).  The graph is built the way upstream builds it with Kolmogorov's Energy class:
 *   add_term1(i, E0, E1)         -> t-links  s->i += E1, i->t += E0          (0 = SOURCE = outlier, 1 = SINK = inlier)
 *   add_term2(i, j, A, B, C, D)  -> s->i += D, i->t += A, edge i->j = B - A, j->i = C - D
 * with e = clamp(r^2/T2, 0, 1), E0 = (1-lambda)(1-e), E1 = 0 for r^2 <= T2, else E0 = 0, E1 = (1-lambda) e;
 * A = lambda (e_i+e_j)/2, B = C = lambda, D = 0; every undirected pair once (i < j), self loops skipped.
 * Terms are quantised to 2^-32 (A as 2 * Q(lambda (e_i+e_j)/4), lambda as pgxo_quantize_lambda) so that the device's
 * re-parameterised graph represents the same integer energy.  Inliers = sites that reach t.
 * ---------------------------------------------------------------------------------------- */
int64_t pgxo_gc_labeling(int model_type, const double *pts, int64_t n, const double *model, double T2, double lambda,
                         const int32_t *off, const int32_t *idx, int32_t *flags)
{
    int d = 0;
    pgxo_model_dims(model_type, &d, NULL);
    double *e = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    uint8_t *inl = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    int64_t *to_t = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    int64_t *from_s = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    const double oml = 1.0 - lambda;
    const int64_t lambda_q = 2 * (int64_t)nearbyint(lambda * 2147483648.0);
    for (int64_t i = 0; i < n; ++i) {
        const double sq = pgxo_squared_residual(model_type, pts + i * d, model);
        inl[i] = (sq <= T2) ? 1 : 0; /* NaN -> beyond the threshold */
        double c = inl[i] ? sq / T2 : 1.0;
        if (c < 0.0) c = 0.0;
        e[i] = c;
        if (inl[i]) to_t[i] += pgxo_quantize(oml * (1.0 - e[i]));
        else from_s[i] += pgxo_quantize(oml * e[i]);
    }
    const int S = (int)n, T = (int)n + 1;
    dinic_t g;
    dn_init(&g, (int)n + 2, 2 * (2 * n + (off ? off[n] : 0)) + 16);
    if (off)
        for (int64_t i = 0; i < n; ++i)
            for (int32_t a = off[i]; a < off[i + 1]; ++a) {
                const int32_t j = idx[a];
                if (j <= i) continue; /* each undirected pair once, no self loops */
                const int64_t A = 2 * pgxo_quantize(lambda * 0.25 * (e[i] + e[j]));
                to_t[i] += A;
                dn_add(&g, (int)i, (int)j, lambda_q - A, lambda_q);
            }
    for (int64_t i = 0; i < n; ++i) {
        dn_add(&g, S, (int)i, from_s[i], 0);
        dn_add(&g, (int)i, T, to_t[i], 0);
    }
    dn_maxflow(&g, S, T);
    uint8_t *reach = (uint8_t *)malloc((size_t)n + 2);
    dn_sink_side(&g, T, reach);
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i) { flags[i] = reach[i] ? 1 : 0; count += flags[i]; }
    free(reach); dn_free(&g); free(from_s); free(to_t); free(inl); free(e);
    return count;
}

/* ------------------------------------------------------------------------------------------
 * a9  PEARL.h:342-352 (bucket by label, ascending point index) ; PEARL.h:369-371 (residual sums)
 * ---------------------------------------------------------------------------------------- */
void pgxo_bucket(const int32_t *labels, int64_t n, int L, int64_t *counts, int32_t *order)
{
    int64_t *start = (int64_t *)calloc((size_t)L + 1, sizeof(int64_t));
    for (int l = 0; l < L; ++l) counts[l] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int l = labels[i];
        if (l >= L - 1) l = L - 1; /* label >= instance_number => outlier bucket (PEARL.h:348-351) */
        counts[l]++;
    }
    for (int l = 0; l < L; ++l) start[l + 1] = start[l] + counts[l];
    if (order)
        for (int64_t i = 0; i < n; ++i) {
            int l = labels[i];
            if (l >= L - 1) l = L - 1;
            order[start[l]++] = (int32_t)i;
        }
    free(start);
}

double pgxo_residual_sum(int model_type, const double *pts, int64_t n, const double *model,
                         const int32_t *labels, int label)
{
    int d = 0;
    pgxo_model_dims(model_type, &d, NULL);
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i)
        if (labels[i] == label) s += pgxo_residual(model_type, pts + i * d, model);
    return s;
}

/* ------------------------------------------------------------------------------------------
 * Minimal solvers (first slice).  models_out[s*3 .. s*3+2]; NaN for a degenerate sample.
 *   vanishing point from two segments: /root/reference/src/pyprogressivex/include/solver_vanishing_point_two_lines.h
 *     :174-179 l_i = endpoint_0 x endpoint_1 (homogeneous, w = 1), :180-182 v = l_0 x l_1, :123-131 vec_norm,
 *     cross product formula :106-121.
 *   2D line through two points: Default2DLineEstimator's solver is absent upstream [U-4]; restated as the unit normal
 *     (-dy, dx)/|d| with offset c = -(n . a).
 * ---------------------------------------------------------------------------------------- */
static void cross3(double a1, double b1, double c1, double a2, double b2, double c2, double *o)
{
    o[0] = b1 * c2 - c1 * b2;
    o[1] = -(a1 * c2 - c1 * a2);
    o[2] = a1 * b2 - b1 * a2;
}

/* 7-point fundamental matrix (DefaultFundamentalMatrixEstimator's minimal solver is absent upstream; restated from the
 * literature): rows (x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1) of the correspondences divided by `scale`; null space by
 * Gauss-Jordan elimination with full pivoting (first maximum in row-major order; rank test 1e-12); det(F2 + l (F1-F2)) as
 * a cubic; one real root by 200 bisection steps inside the Cauchy bound, the other two from the quadratic factor; every
 * root gives F = l F1 + (1-l) F2, un-scaled and normalised to unit Frobenius norm.  out = 3 slots x 9, NaN = no model. */
static double det3(const double *a)
{
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

static void solve_f7(const double *pts, int64_t n, const int32_t *smp, double scale, double *out)
{
    for (int k = 0; k < 27; ++k) out[k] = NAN;
    double M[7][9];
    for (int r = 0; r < 7; ++r) {
        const int32_t i = smp[r];
        if (i < 0 || i >= n) return;
        const double x1 = pts[(size_t)i * 4] / scale, y1 = pts[(size_t)i * 4 + 1] / scale;
        const double x2 = pts[(size_t)i * 4 + 2] / scale, y2 = pts[(size_t)i * 4 + 3] / scale;
        M[r][0] = x2 * x1; M[r][1] = x2 * y1; M[r][2] = x2;
        M[r][3] = y2 * x1; M[r][4] = y2 * y1; M[r][5] = y2;
        M[r][6] = x1; M[r][7] = y1; M[r][8] = 1.0;
    }
    int col[9];
    for (int j = 0; j < 9; ++j) col[j] = j;
    for (int r = 0; r < 7; ++r) {
        int pr = r, pc = r;
        double best = -1.0;
        for (int i = r; i < 7; ++i)
            for (int j = r; j < 9; ++j) {
                const double a = fabs(M[i][j]);
                if (a > best) { best = a; pr = i; pc = j; }
            }
        if (!(best >= 1e-12)) return;
        if (pr != r) for (int j = 0; j < 9; ++j) { const double t = M[r][j]; M[r][j] = M[pr][j]; M[pr][j] = t; }
        if (pc != r) {
            for (int i = 0; i < 7; ++i) { const double t = M[i][r]; M[i][r] = M[i][pc]; M[i][pc] = t; }
            const int t = col[r]; col[r] = col[pc]; col[pc] = t;
        }
        const double piv = M[r][r];
        for (int j = r; j < 9; ++j) M[r][j] = M[r][j] / piv;
        for (int i = 0; i < 7; ++i) {
            if (i == r) continue;
            const double f = M[i][r];
            for (int j = r; j < 9; ++j) M[i][j] = M[i][j] - f * M[r][j];
        }
    }
    double F1[9], F2[9], D[9], T[9];
    for (int k = 0; k < 7; ++k) { F1[col[k]] = -M[k][7]; F2[col[k]] = -M[k][8]; }
    F1[col[7]] = 1.0; F1[col[8]] = 0.0;
    F2[col[7]] = 0.0; F2[col[8]] = 1.0;
    for (int k = 0; k < 9; ++k) D[k] = F1[k] - F2[k];
    const double c0 = det3(F2), c3 = det3(D);
    double c1 = 0.0, c2 = 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 9; ++k) T[k] = F2[k];
        for (int k = 0; k < 3; ++k) T[3 * r + k] = D[3 * r + k];
        c1 = c1 + det3(T);
        for (int k = 0; k < 9; ++k) T[k] = D[k];
        for (int k = 0; k < 3; ++k) T[3 * r + k] = F2[3 * r + k];
        c2 = c2 + det3(T);
    }
    double roots[3] = {NAN, NAN, NAN};
    const double cm = fmax(fmax(fabs(c0), fabs(c1)), fmax(fabs(c2), fabs(c3)));
    if (!(cm > 0.0) || !(cm < 1e300)) return;
    if (fabs(c3) > 1e-14 * cm) {
        const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
        double lo = -(1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)))), hi = -lo;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            const double pv = ((mid + a) * mid + b) * mid + c;
            if (pv < 0.0) lo = mid; else hi = mid;
        }
        const double l0 = 0.5 * (lo + hi);
        roots[0] = l0;
        const double qb = a + l0, qc = b + qb * l0;
        const double disc = qb * qb - 4.0 * qc;
        if (disc >= 0.0) {
            const double sq = sqrt(disc);
            roots[1] = (-qb - sq) / 2.0;
            roots[2] = (-qb + sq) / 2.0;
        }
    } else if (fabs(c2) > 1e-14 * cm) {
        const double disc = c1 * c1 - 4.0 * c2 * c0;
        if (disc >= 0.0) {
            const double sq = sqrt(disc);
            roots[0] = (-c1 - sq) / (2.0 * c2);
            roots[1] = (-c1 + sq) / (2.0 * c2);
        }
    } else if (fabs(c1) > 1e-14 * cm) {
        roots[0] = -c0 / c1;
    }
    const double s1 = scale, s2 = scale * scale;
    for (int q = 0; q < 3; ++q) {
        const double l = roots[q];
        if (!(l == l)) continue;
        double F[9];
        for (int k = 0; k < 9; ++k) F[k] = l * F1[k] + (1.0 - l) * F2[k];
        F[0] = F[0] / s2; F[1] = F[1] / s2; F[2] = F[2] / s1;
        F[3] = F[3] / s2; F[4] = F[4] / s2; F[5] = F[5] / s1;
        F[6] = F[6] / s1; F[7] = F[7] / s1;
        double nn = 0.0;
        for (int k = 0; k < 9; ++k) nn = nn + F[k] * F[k];
        const double nrm = sqrt(nn);
        if (!(nrm > 0.0) || !(nrm < 1e300)) continue;
        for (int k = 0; k < 9; ++k) out[9 * q + k] = F[k] / nrm;
    }
}

/* monic cubic x^3 + a x^2 + b x + c: one real root by 200 bisection steps inside the Cauchy bound, the other two from the
 * quadratic factor (NaN when complex).  Only + - * / sqrt. */
static void cubic_roots(double a, double b, double c, double *r)
{
    r[0] = r[1] = r[2] = NAN;
    double lo = -(1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)))), hi = -lo;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double pv = ((mid + a) * mid + b) * mid + c;
        if (pv < 0.0) lo = mid; else hi = mid;
    }
    const double l0 = 0.5 * (lo + hi);
    r[0] = l0;
    const double qb = a + l0, qc = b + qb * l0;
    const double disc = qb * qb - 4.0 * qc;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        r[1] = (-qb - sq) / 2.0;
        r[2] = (-qb + sq) / 2.0;
    }
}

static double quartic_eval(const double *m, double x) { return (((x + m[3]) * x + m[2]) * x + m[1]) * x + m[0]; }

/* P3P (Grunert's quartic in v = s3/s1 as in Fischler-Bolles; DefaultPnPEstimator's minimal solver is absent upstream):
 * rows (u, v, X, Y, Z) in normalised image coordinates; positive real roots of the quartic by bisection between the
 * critical points (roots of the derivative cubic) inside (0, Cauchy bound); depths from the two remaining constraints
 * (consistency 1e-6); pose from the orthonormal frames of the two congruent triangles.  out = 4 slots x 12 ([R|t]
 * row-major), NaN = no solution. */
static void solve_p3p(const double *pts, int64_t n, const int32_t *smp, double *out)
{
    for (int k = 0; k < 48; ++k) out[k] = NAN;
    double f[3][3], X[3][3];
    for (int r = 0; r < 3; ++r) {
        const int32_t i = smp[r];
        if (i < 0 || i >= n) return;
        const double u = pts[(size_t)i * 5], v = pts[(size_t)i * 5 + 1];
        const double ln = sqrt(u * u + v * v + 1.0);
        f[r][0] = u / ln; f[r][1] = v / ln; f[r][2] = 1.0 / ln;
        X[r][0] = pts[(size_t)i * 5 + 2]; X[r][1] = pts[(size_t)i * 5 + 3]; X[r][2] = pts[(size_t)i * 5 + 4];
    }
    double d12[3], d02[3], d01[3];
    for (int k = 0; k < 3; ++k) { d12[k] = X[1][k] - X[2][k]; d02[k] = X[0][k] - X[2][k]; d01[k] = X[0][k] - X[1][k]; }
    const double a2 = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    const double b2 = d02[0] * d02[0] + d02[1] * d02[1] + d02[2] * d02[2];
    const double c2 = d01[0] * d01[0] + d01[1] * d01[1] + d01[2] * d01[2];
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return;
    const double ca = f[1][0] * f[2][0] + f[1][1] * f[2][1] + f[1][2] * f[2][2];
    const double cb = f[0][0] * f[2][0] + f[0][1] * f[2][1] + f[0][2] * f[2][2];
    const double cg = f[0][0] * f[1][0] + f[0][1] * f[1][1] + f[0][2] * f[1][2];
    const double q = (a2 - c2) / b2, rr = (a2 + c2) / b2;
    const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - rr) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca - 4.0 * rr * ca * cb * cg +
                             2.0 * (b2 - a2) / b2 * cg * cg);
    const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - rr) * ca * cg);
    const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
    if (!(fabs(A4) > 1e-14) || !(fabs(A4) < 1e300)) return;
    double m[4] = {A0 / A4, A1 / A4, A2 / A4, A3 / A4};  /* monic: x^4 + m3 x^3 + m2 x^2 + m1 x + m0 */
    for (int k = 0; k < 4; ++k) if (!(fabs(m[k]) < 1e300)) return;
    const double B = 1.0 + fmax(fmax(fabs(m[0]), fabs(m[1])), fmax(fabs(m[2]), fabs(m[3])));
    double crit[3];
    cubic_roots(0.75 * m[3], 0.5 * m[2], 0.25 * m[1], crit);
    /* break points of (0, B): the positive critical points in ascending order */
    double bp[5];
    int nb = 0;
    bp[nb++] = 0.0;
    for (int pass = 0; pass < 3; ++pass) {          /* selection in ascending order (at most three values) */
        double best = B;
        for (int k = 0; k < 3; ++k)
            if (crit[k] == crit[k] && crit[k] > bp[nb - 1] && crit[k] < best) best = crit[k];
        if (best < B) bp[nb++] = best; else break;
    }
    bp[nb++] = B;
    int slot = 0;
    for (int s = 0; s + 1 < nb && slot < 4; ++s) {
        double lo = bp[s], hi = bp[s + 1];
        const double plo = quartic_eval(m, lo), phi = quartic_eval(m, hi);
        if (!((plo < 0.0 && phi >= 0.0) || (plo >= 0.0 && phi < 0.0))) continue;
        const int rising = plo < 0.0;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            const double pv = quartic_eval(m, mid);
            if ((pv < 0.0) == rising) lo = mid; else hi = mid;
        }
        const double v = 0.5 * (lo + hi);
        if (!(v > 0.0)) continue;
        const double den = 2.0 * (cg - v * ca);
        const double u = ((-1.0 + q) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
        const double s1 = sqrt(b2 / (1.0 + v * v - 2.0 * v * cb));
        const double s2 = u * s1, s3 = v * s1;
        if (!(s1 > 0.0) || !(s2 > 0.0) || !(s3 > 0.0) || !(s1 < 1e300) || !(s2 < 1e300) || !(s3 < 1e300)) continue;
        const double e1 = s1 * s1 + s2 * s2 - 2.0 * s1 * s2 * cg - c2;
        const double e2 = s2 * s2 + s3 * s3 - 2.0 * s2 * s3 * ca - a2;
        if (!(fabs(e1) < 1e-6 * c2) || !(fabs(e2) < 1e-6 * a2)) continue;
        /* camera-frame points and the two orthonormal frames */
        const double dep[3] = {s1, s2, s3};
        double Y[3][3];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Y[r][k] = f[r][k] * dep[r];
        double EX[3][3], EY[3][3];
        int okf = 1;
        for (int w = 0; w < 2; ++w) {
            double (*P)[3] = w == 0 ? X : Y;
            double (*E)[3] = w == 0 ? EX : EY;
            double p1[3], p2[3];
            for (int k = 0; k < 3; ++k) { p1[k] = P[1][k] - P[0][k]; p2[k] = P[2][k] - P[0][k]; }
            const double n1 = sqrt(p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2]);
            if (!(n1 > 0.0)) { okf = 0; break; }
            for (int k = 0; k < 3; ++k) E[0][k] = p1[k] / n1;
            double cr[3] = {E[0][1] * p2[2] - E[0][2] * p2[1], E[0][2] * p2[0] - E[0][0] * p2[2], E[0][0] * p2[1] - E[0][1] * p2[0]};
            const double n3 = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
            if (!(n3 > 0.0)) { okf = 0; break; }
            for (int k = 0; k < 3; ++k) E[2][k] = cr[k] / n3;
            E[1][0] = E[2][1] * E[0][2] - E[2][2] * E[0][1];
            E[1][1] = E[2][2] * E[0][0] - E[2][0] * E[0][2];
            E[1][2] = E[2][0] * E[0][1] - E[2][1] * E[0][0];
        }
        if (!okf) continue;
        double *P = out + 12 * slot;
        double R[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = EY[0][i] * EX[0][j] + EY[1][i] * EX[1][j] + EY[2][i] * EX[2][j];
        int fin = 1;
        for (int i = 0; i < 3; ++i) {
            const double t = Y[0][i] - (R[i][0] * X[0][0] + R[i][1] * X[0][1] + R[i][2] * X[0][2]);
            if (!(fabs(t) < 1e300)) fin = 0;
            for (int j = 0; j < 3; ++j) { if (!(fabs(R[i][j]) < 1e300)) fin = 0; P[4 * i + j] = R[i][j]; }
            P[4 * i + 3] = t;
        }
        if (!fin) { for (int k = 0; k < 12; ++k) P[k] = NAN; continue; }
        ++slot;
    }
}

/* 4-point homography (DefaultHomographyEstimator's minimal solver is absent upstream; restated): h33 = 1, the 8x8 DLT
 * system (x-equations of the four points, then their y-equations) of the points divided by `scale`, Gaussian
 * elimination with partial pivoting (first maximum; rank test 1e-12), back substitution, scaling undone. */
static void solve_h4(const double *pts, int64_t n, const int32_t *smp, double scale, double *out)
{
    for (int k = 0; k < 9; ++k) out[k] = NAN;
    double M[8][9];
    for (int r = 0; r < 4; ++r) {
        const int32_t i = smp[r];
        if (i < 0 || i >= n) return;
        const double x1 = pts[(size_t)i * 4] / scale, y1 = pts[(size_t)i * 4 + 1] / scale;
        const double x2 = pts[(size_t)i * 4 + 2] / scale, y2 = pts[(size_t)i * 4 + 3] / scale;
        double *a = M[r], *b = M[r + 4];
        a[0] = -x1; a[1] = -y1; a[2] = -1.0; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; a[6] = x2 * x1; a[7] = x2 * y1; a[8] = -x2;
        b[0] = 0.0; b[1] = 0.0; b[2] = 0.0; b[3] = -x1; b[4] = -y1; b[5] = -1.0; b[6] = y2 * x1; b[7] = y2 * y1; b[8] = -y2;
    }
    for (int c = 0; c < 8; ++c) {
        int pr = c;
        double best = fabs(M[c][c]);
        for (int i = c + 1; i < 8; ++i) { const double a = fabs(M[i][c]); if (a > best) { best = a; pr = i; } }
        if (!(best >= 1e-12)) return;
        if (pr != c) for (int j = c; j < 9; ++j) { const double t = M[c][j]; M[c][j] = M[pr][j]; M[pr][j] = t; }
        for (int i = c + 1; i < 8; ++i) {
            const double f = M[i][c] / M[c][c];
            for (int j = c; j < 9; ++j) M[i][j] = M[i][j] - f * M[c][j];
        }
    }
    double h[9];
    for (int c = 7; c >= 0; --c) {
        double acc = M[c][8];
        for (int j = c + 1; j < 8; ++j) acc = acc - M[c][j] * h[j];
        h[c] = acc / M[c][c];
    }
    h[8] = 1.0;
    h[2] = h[2] * scale; h[5] = h[5] * scale; h[6] = h[6] / scale; h[7] = h[7] / scale;
    for (int k = 0; k < 9; ++k) if (!(fabs(h[k]) < 1e300)) return;
    for (int k = 0; k < 9; ++k) out[k] = h[k];
}

int pgxo_solve_minimal(int model_type, const double *pts, int64_t n, const int32_t *samples, int S, double *models_out)
{
    if (model_type == PGXO_PNP) {
        for (int s = 0; s < S; ++s) solve_p3p(pts, n, samples + (size_t)s * 3, models_out + (size_t)s * 48);
        return 0;
    }
    if (model_type == PGXO_HOMOGRAPHY) {
        double scale = 1.0;
        for (int64_t i = 0; i < n * 4; ++i) { const double a = fabs(pts[i]); if (a > scale) scale = a; }
        for (int s = 0; s < S; ++s) solve_h4(pts, n, samples + (size_t)s * 4, scale, models_out + (size_t)s * 9);
        return 0;
    }
    if (model_type == PGXO_FUNDAMENTAL) {
        double scale = 1.0;
        for (int64_t i = 0; i < n * 4; ++i) { const double a = fabs(pts[i]); if (a > scale) scale = a; }
        for (int s = 0; s < S; ++s) solve_f7(pts, n, samples + (size_t)s * 7, scale, models_out + (size_t)s * 27);
        return 0;
    }
    if (model_type != PGXO_LINE2D && model_type != PGXO_VANISHING_POINT) return -1;
    for (int s = 0; s < S; ++s) {
        const int32_t i0 = samples[2 * s], i1 = samples[2 * s + 1];
        double *m = models_out + (size_t)s * 3;
        m[0] = m[1] = m[2] = NAN;
        if (i0 < 0 || i1 < 0 || i0 >= n || i1 >= n) continue;
        if (model_type == PGXO_LINE2D) {
            const double ax = pts[(size_t)i0 * 2], ay = pts[(size_t)i0 * 2 + 1];
            const double dx = pts[(size_t)i1 * 2] - ax, dy = pts[(size_t)i1 * 2 + 1] - ay;
            const double ln = sqrt(dx * dx + dy * dy);
            if (ln > 0.0) {
                m[0] = -dy / ln;
                m[1] = dx / ln;
                m[2] = -(m[0] * ax + m[1] * ay);
            }
        } else {
            const double *a = pts + (size_t)i0 * 4, *b = pts + (size_t)i1 * 4;
            double l0[3], l1[3], v[3];
            cross3(a[0], a[1], 1.0, a[2], a[3], 1.0, l0);
            cross3(b[0], b[1], 1.0, b[2], b[3], 1.0, l1);
            cross3(l0[0], l0[1], l0[2], l1[0], l1[1], l1[2], v);
            const double ln = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (ln > 0.0) { m[0] = v[0] / ln; m[1] = v[1] / ln; m[2] = v[2] / ln; }
        }
    }
    return 0;
}


/* ---- in-repo counter-based generator (see pgx_oracle.h) ------------------------------------------------------------------ */
void pgxo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

void pgxo_sample_uniform(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, int m, int32_t* samples)
{
    const uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    for (int64_t t = 0; t < count; ++t) {
        const uint64_t s = (uint64_t)(first + t);
        int64_t taken[8];
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < m; ++j) {
            if (j % 4 == 0) {
                const uint32_t ctr[4] = {(uint32_t)s, (uint32_t)(s >> 32), batch, (uint32_t)(j / 4)};
                pgxo_philox4x32(ctr, k, w);
            }
            int64_t r = (int64_t)(((uint64_t)w[j % 4] * (uint64_t)(n - j)) >> 32);
            int pos = 0;
            while (pos < j && taken[pos] <= r) { ++r; ++pos; }      /* the r-th index not taken yet */
            for (int q = j; q > pos; --q) taken[q] = taken[q - 1];
            taken[pos] = r;
            samples[t * m + j] = (int32_t)r;
        }
    }
}

/* PROSAC on the same generator (gcransac::sampler::ProsacSampler, progressivex_python.cpp:222; absent upstream [UPSTREAM-MEMORY: the USAC
 * formulation]): sample first + t uses tops[t] = n_k - m - 1 distinct indices of the best n_k - 1 points plus point n_k - 1;
 * tops[t] == 0: uniform over all n; tops[t] < m or > n: no sample (-1). */
void pgxo_sample_prosac(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, const int32_t* tops, int m, int32_t* samples)
{
    for (int64_t t = 0; t < count; ++t) {
        int32_t* row = samples + t * m;
        const int64_t top = tops[t];
        if (top == 0) {
            pgxo_sample_uniform(key, batch, first + t, 1, n, m, row);
        } else if (top < m || top > n) {
            for (int j = 0; j < m; ++j) row[j] = -1;
        } else {
            if (m > 1) {
                int32_t head[8];
                pgxo_sample_uniform(key, batch, first + t, 1, top - 1, m - 1, head);
                for (int j = 0; j < m - 1; ++j) row[j] = head[j];
            }
            row[m - 1] = (int32_t)(top - 1);
        }
    }
}

void pgxo_sample_napsac(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, const int32_t* off, const int32_t* idx, int m,
                        int32_t* samples)
{
    const uint32_t k[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    for (int64_t t = 0; t < count; ++t) {
        const uint64_t s = (uint64_t)(first + t);
        int32_t* row = samples + t * m;
        uint32_t w[4];
        const uint32_t ctr0[4] = {(uint32_t)s, (uint32_t)(s >> 32), batch, 0u};
        pgxo_philox4x32(ctr0, k, w);
        const int64_t c = (int64_t)(((uint64_t)w[0] * (uint64_t)n) >> 32);
        const int64_t a0 = off[c], deg = off[c + 1] - off[c];
        if (deg < m - 1) {
            for (int j = 0; j < m; ++j) row[j] = -1;
            continue;
        }
        int64_t taken[8];
        row[0] = (int32_t)c;
        for (int j = 1; j < m; ++j) {
            if (j % 4 == 0) {
                const uint32_t ctr[4] = {(uint32_t)s, (uint32_t)(s >> 32), batch, (uint32_t)(j / 4)};
                pgxo_philox4x32(ctr, k, w);
            }
            int64_t r = (int64_t)(((uint64_t)w[j % 4] * (uint64_t)(deg - (j - 1))) >> 32);
            int pos = 0;
            while (pos < j - 1 && taken[pos] <= r) { ++r; ++pos; }
            for (int q = j - 1; q > pos; --q) taken[q] = taken[q - 1];
            taken[pos] = r;
            row[j] = idx[a0 + r];
        }
    }
}

/* Progressive NAPSAC on the same generator (gcransac::sampler::ProgressiveNapsacSampler<4>, progressivex_python.cpp:229-238; absent
 * upstream [UPSTREAM-MEMORY]): the statement of pyprogressivex/_rng.py pnapsac_samples in C - the checker of libpgx.so's host code
 * (csrc/sampler_host.hip).  pts [n][d] in quality order, grid layers over the first min(d, 4) coordinates. */
typedef struct { int64_t cid; int32_t idx; } pnap_pair;
static int pnap_cmp(const void *a, const void *b)
{
    const pnap_pair *x = (const pnap_pair *)a, *y = (const pnap_pair *)b;
    if (x->cid != y->cid) return x->cid < y->cid ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);   /* members in ascending index = quality order */
}

int pgxo_sample_pnapsac(const double *pts, int64_t n, int d, const double *sizes, const int32_t *layers, int n_layers, int m,
                        uint64_t key, uint32_t batch, int32_t count, const int32_t *tops, const int64_t *growth_local, int64_t max_local,
                        int32_t *samples)
{
    if (n < m || m < 2 || m > 8 || n_layers < 1 || n_layers > 16) return -1;
    const int dims = d < 4 ? d : 4;
    const uint32_t k2[2] = {(uint32_t)key, (uint32_t)(key >> 32)};
    /* per layer: members sorted by (cell, index), the start of every point's cell run and its length */
    pnap_pair *pairs = (pnap_pair *)malloc((size_t)n_layers * (size_t)n * sizeof(pnap_pair));
    int32_t *run0 = (int32_t *)malloc((size_t)n_layers * (size_t)n * sizeof(int32_t));
    int32_t *runlen = (int32_t *)malloc((size_t)n_layers * (size_t)n * sizeof(int32_t));
    int64_t *hits = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int32_t *subset = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *layer = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    int32_t *others = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!pairs || !run0 || !runlen || !hits || !subset || !layer || !others) { free(pairs); free(run0); free(runlen); free(hits); free(subset); free(layer); free(others); return -2; }
    for (int l = 0; l < n_layers; ++l) {
        pnap_pair *pl = pairs + (size_t)l * (size_t)n;
        const int div = layers[l];
        for (int64_t i = 0; i < n; ++i) {
            int64_t id = 0;
            for (int q = 0; q < dims; ++q) {
                double c = floor(pts[i * d + q] / (sizes[q] / (double)div));
                c = c > 0.0 ? (c > (double)(div - 1) ? (double)(div - 1) : c) : 0.0;
                id = id * div + (int64_t)c;
            }
            pl[i].cid = id;
            pl[i].idx = (int32_t)i;
        }
        qsort(pl, (size_t)n, sizeof(pnap_pair), pnap_cmp);
        for (int64_t a = 0; a < n;) {
            int64_t b = a;
            while (b < n && pl[b].cid == pl[a].cid) ++b;
            for (int64_t q = a; q < b; ++q) { run0[(size_t)l * (size_t)n + (size_t)pl[q].idx] = (int32_t)a; runlen[(size_t)l * (size_t)n + (size_t)pl[q].idx] = (int32_t)(b - a); }
            a = b;
        }
    }
    for (int64_t i = 0; i < n; ++i) subset[i] = m;
    const int64_t n_local = (int64_t)count < max_local ? (int64_t)count : max_local;
    for (int64_t k = 0; k < count; ++k) {
        int32_t *row = samples + k * m;
        int global = k >= n_local;
        if (!global) {
            uint32_t w[4];
            uint32_t ctr[4] = {(uint32_t)k, (uint32_t)((uint64_t)k >> 32), batch, 0u};
            pgxo_philox4x32(ctr, k2, w);
            const int64_t p = k < n ? k : (int64_t)(((uint64_t)w[0] * (uint64_t)n) >> 32);
            hits[p] += 1;
            int64_t sp = subset[p];
            while (sp < n && hits[p] > growth_local[sp - 1]) ++sp;
            subset[p] = (int32_t)sp;
            int lay = layer[p];
            const pnap_pair *nb = NULL;
            for (; lay < n_layers; ++lay)
                if ((int64_t)runlen[(size_t)lay * (size_t)n + (size_t)p] >= sp) { nb = pairs + (size_t)lay * (size_t)n + (size_t)run0[(size_t)lay * (size_t)n + (size_t)p]; break; }
            layer[p] = lay;
            int cnt = 0;
            if (nb) for (int64_t q = 0; q < sp; ++q) if (nb[q].idx != (int32_t)p) others[cnt++] = nb[q].idx;
            if (!nb || cnt < m - 1) global = 1;
            else {
                int32_t taken[8];
                for (int j = 0; j < m - 2; ++j) {
                    if (((1 + j) & 3) == 0) { ctr[3] = (uint32_t)((1 + j) >> 2); pgxo_philox4x32(ctr, k2, w); }
                    int64_t r = (int64_t)(((uint64_t)w[(1 + j) & 3] * (uint64_t)(cnt - 1 - j)) >> 32);
                    int pos = 0;
                    for (; pos < j && taken[pos] <= r; ++pos) ++r;
                    for (int q = j; q > pos; --q) taken[q] = taken[q - 1];
                    taken[pos] = (int32_t)r;
                    row[j] = others[r];
                    hits[others[r]] += 1;
                }
                row[m - 2] = others[cnt - 1];
                hits[others[cnt - 1]] += 1;
                row[m - 1] = (int32_t)p;
            }
        }
        if (global) pgxo_sample_prosac(key, batch, k, 1, n, tops + k, m, row);
    }
    free(pairs); free(run0); free(runlen); free(hits); free(subset); free(layer); free(others);
    return 0;
}


/* ---- smallest eigenpair by cyclic Jacobi (pgx_oracle.h; same operation order as csrc/fit.hip eigh_smallest_kernel) ---------------- */
void pgxo_eigh_smallest(const double* A, int q, int64_t B, double* vec, double* val, int32_t* sweeps)
{
    for (int64_t b = 0; b < B; ++b) {
        double a[9][9] = {{0}}, v[9][9] = {{0}};
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < q; ++j) {
                a[i][j] = A[(size_t)b * q * q + (size_t)i * q + j];
                v[i][j] = i == j ? 1.0 : 0.0;
            }
        int sweep = 0;
        for (; sweep < 50; ++sweep) {
            double off = 0.0, dg = 0.0;
            for (int p = 0; p < q; ++p) {
                dg = dg + a[p][p] * a[p][p];
                for (int r = p + 1; r < q; ++r) off = off + a[p][r] * a[p][r];
            }
            if (!(off > 4.930380657631324e-32 * dg)) break;      /* eps^2; also leaves on NaN */
            for (int p = 0; p < q - 1; ++p)
                for (int r = p + 1; r < q; ++r) {
                    const double apr = a[p][r];
                    if (apr == 0.0) continue;
                    const double theta = (a[r][r] - a[p][p]) / (2.0 * apr);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                    for (int k = 0; k < q; ++k) {
                        const double akp = a[k][p], akr = a[k][r];
                        a[k][p] = c * akp - sn * akr;
                        a[k][r] = sn * akp + c * akr;
                    }
                    for (int k = 0; k < q; ++k) {
                        const double apk = a[p][k], ark = a[r][k];
                        a[p][k] = c * apk - sn * ark;
                        a[r][k] = sn * apk + c * ark;
                    }
                    for (int k = 0; k < q; ++k) {
                        const double vkp = v[k][p], vkr = v[k][r];
                        v[k][p] = c * vkp - sn * vkr;
                        v[k][r] = sn * vkp + c * vkr;
                    }
                }
        }
        int best = 0;
        for (int p = 1; p < q; ++p)
            if (a[p][p] < a[best][best]) best = p;
        for (int k = 0; k < q; ++k) vec[(size_t)b * q + k] = v[k][best];
        val[b] = a[best][best];
        if (sweeps) sweeps[b] = sweep;
    }
}
