/*
 * pgx_oracle.h — CPU oracle for the Progressive-X hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (progressive-x_amd/)
 * may include, link or call this.  Allowed callers: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (danini/progressive-x) has no tests, no golden
 * vectors, cannot be built here (needs Eigen3, OpenCV, glog, gflags and the
 * un-vendored, unpinned `graph-cut-ransac` submodule — see DESIGN.md §3), and is
 * stochastic.  Functions that follow in-tree reference code cite file:line (paths
 * relative to /root/reference/src/pyprogressivex/).  Functions whose upstream
 * source is absent are restated from the published mathematics and tagged U-n
 * (same numbering as SURVEY.md §8a).
 *
 * Everything is `double`, compiled with -ffp-contract=off (the reference builds
 * for baseline x86-64: no FMA).  Single-threaded, like the reference.
 */
#ifndef PGX_ORACLE_H
#define PGX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* model types (same numbering as include/pgx.h) */
enum {
    PGXO_LINE2D = 0,          /* pt (x,y)            model (a,b,c), a^2+b^2=1            U-4 */
    PGXO_HOMOGRAPHY = 1,      /* pt (x1,y1,x2,y2)    model H row-major 3x3 (one-way)     U-1 */
    PGXO_FUNDAMENTAL = 2,     /* pt (x1,y1,x2,y2)    model F row-major 3x3 (Sampson)     U-2 */
    PGXO_PNP = 3,             /* pt (u,v,X,Y,Z)      model P=[R|t] row-major 3x4         U-3 */
    PGXO_VANISHING_POINT = 4, /* pt (xs,ys,xe,ye)    model (v0,v1,v2)  in-tree              */
    PGXO_HOMOGRAPHY_SYM = 5   /* pt (x1,y1,x2,y2)    model [H | H^-1] 18 doubles (symmetric transfer) */
};

#define PGXO_FIXED_SHIFT 32   /* energies are quantised to multiples of 2^-32 for the min-cut */

int pgxo_model_dims(int model_type, int *point_dim, int *param_dim);

/* per-pair residuals */
double pgxo_squared_residual(int model_type, const double *pt, const double *model);
double pgxo_residual(int model_type, const double *pt, const double *model);
void pgxo_squared_residuals(int model_type, const double *pts, int64_t n, const double *model, double *out);

/* a1: MSACScoringFunctionWithCompoundModel::getScore, batched over M hypotheses.
 * best_inlier_number[m] (may be NULL => 0) reproduces the early exit.
 * masks (may be NULL): M x ceil(n/64) uint64 words, bit i of row m = inlier. */
void pgxo_score(int model_type, const double *pts, int64_t n, const double *models, int M,
                double T2, const double *compound, int has_compound, int exponent,
                const int64_t *best_inlier_number,
                int64_t *counts, double *values, double *shared, double *scores, uint64_t *masks);

/* a2/a3: preference vector + Tanimoto terms */
void pgxo_preference(int model_type, const double *pts, int64_t n, const double *model, double T2,
                     double *pref);
void pgxo_tanimoto_terms(const double *pref, const double *compound, int64_t n,
                         double *dot, double *pref_sqnorm, double *comp_sqnorm);
int pgxo_is_valid_tanimoto(double dot, double pref_sqnorm, double comp_sqnorm, double max_tanimoto,
                           double *tanimoto);
/* a4 */
void pgxo_compound_max(const double *prefs, int K, int64_t n, double *compound);
/* a5 */
uint64_t pgxo_predicted_unseen_inliers(double one_minus_conf, uint64_t sample_size,
                                       uint64_t iteration_number, uint64_t covered, uint64_t point_number);

/* a6: unary cost table (double) and its fixed-point quantisation. D is n x (K+1), label K = outlier. */
void pgxo_unary(int model_type, const double *pts, int64_t n, const double *models, int K,
                double threshold, double lambda, double *D);
int64_t pgxo_quantize(double x);
void pgxo_unary_q(int model_type, const double *pts, int64_t n, const double *models, int K,
                  double threshold, double lambda, int64_t *Dq);

/* a8/a19: alpha-expansion with Potts pairwise + uniform per-label cost on a symmetric CSR graph.
 * off[n+1], idx[off[n]], mult[off[n]] : every undirected pair appears in both lists with the same
 * multiplicity (= number of directed neighbour entries of the raw lists, U-6).
 * lambda_q is the quantised weight of ONE directed entry (must be even), h_q the quantised label cost. */
int64_t pgxo_energy(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                    const int32_t *mult, int64_t lambda_q, int64_t h_q, const int32_t *labels);
int pgxo_expand_alpha(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                      const int32_t *mult, int64_t lambda_q, int64_t h_q, int alpha, int32_t *labels,
                      int64_t *flow_value);
int pgxo_expansion(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                   const int32_t *mult, int64_t lambda_q, int64_t h_q, int32_t *labels, int max_cycles,
                   int64_t *energy_q, int *cycles);
/* the same move / loop with Boykov-Kolmogorov as the solver (bk_maxflow.c: the algorithm behind GCoptimization, PEARL.h:550);
 * identical labels by uniqueness of the minimal sink side - used for the CPU labelling baseline of bench.py */
int pgxo_expand_alpha_bk(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                         const int32_t *mult, int64_t lambda_q, int64_t h_q, int alpha, int32_t *labels,
                         int64_t *flow_value);
int pgxo_expansion_bk(int64_t n, int L, const int64_t *Dq, const int32_t *off, const int32_t *idx,
                      const int32_t *mult, int64_t lambda_q, int64_t h_q, int32_t *labels, int max_cycles,
                      int64_t *energy_q, int *cycles, int64_t *mincuts);
int64_t pgxo_maxflow_bk(int nnodes, int64_t narcs, const int32_t *from, const int32_t *to,
                        const int64_t *cap, int s, int t, uint8_t *sink_side);
/* U-8: GCO-v3's no-smooth-cost special cases (greedy facility location with per-label costs; argmin without) */
int pgxo_greedy_labeling(int64_t n, int L, const int64_t *Dq, int64_t h_q, int32_t *labels, int64_t *energy_q);
/* 8f rank 4: GC-RANSAC's inlier/outlier graph cut, graph built as upstream's Energy::add_term1/add_term2 would [U-12].
 * flags[n]: 1 = inlier; returns the inlier count.  off/idx: symmetric CSR (multiplicities are ignored: pairs count once). */
int64_t pgxo_gc_labeling(int model_type, const double *pts, int64_t n, const double *model, double T2, double lambda,
                         const int32_t *off, const int32_t *idx, int32_t *flags);

/* plain s-t max-flow on an explicit arc list (used to cross-check the solver against scipy) */
int64_t pgxo_maxflow(int nnodes, int64_t narcs, const int32_t *from, const int32_t *to,
                     const int64_t *cap, int s, int t, uint8_t *sink_side);

/* a9: label bucketing + residual sums */
void pgxo_bucket(const int32_t *labels, int64_t n, int L, int64_t *counts, int32_t *order);
void pgxo_epipolar_support(const double *pts, int64_t n, const double *f, double T2, double S2, int64_t *out);
double pgxo_residual_sum(int model_type, const double *pts, int64_t n, const double *model,
                         const int32_t *labels, int label);

/* minimal solvers: 2-point line, 2-segment vanishing point; NaN model for a degenerate sample */
int pgxo_solve_minimal(int model_type, const double *pts, int64_t n, const int32_t *samples, int S, double *models_out);

/* ---- the in-repo counter-based generator: Philox4x32-10 (Salmon et al., SC'11) and the uniform
 * minimal-sample sampler on it - an independent C restatement of csrc/rng.hip.h / pyprogressivex/_rng.py for the tests.
 * Replaces gcransac::sampler::UniformSampler (progressivex_python.cpp:121, 215-245; source absent). */
void pgxo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void pgxo_sample_uniform(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, int m, int32_t* samples);
/* NAPSAC on the same generator: a uniform centre + m - 1 distinct entries of its CSR row; rows of -1 where the centre has fewer */
void pgxo_sample_napsac(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, const int32_t* off, const int32_t* idx, int m,
                        int32_t* samples);

/* PROSAC on the same generator: tops[t] = the hypothesis-generation set size n_k of sample first + t (0 = uniform over all n) */
void pgxo_sample_prosac(uint64_t key, uint32_t batch, int64_t first, int64_t count, int64_t n, const int32_t* tops, int m, int32_t* samples);
/* Progressive NAPSAC, one draw (a fresh sampler state), count x m indices; 0 on success */
int pgxo_sample_pnapsac(const double *pts, int64_t n, int d, const double *sizes, const int32_t *layers, int n_layers, int m,
                        uint64_t key, uint32_t batch, int32_t count, const int32_t *tops, const int64_t *growth_local, int64_t max_local,
                        int32_t *samples);

#ifdef __cplusplus
}
#endif
/* Smallest eigenpair of B symmetric q x q matrices (q <= 9) by cyclic Jacobi rotations in FP64 - the small dense solve of the
 * non-minimal refits (the reference reaches Eigen::SelfAdjointEigenSolver: solver_vanishing_point_two_lines.h:227; the DLT / 8-point
 * solvers of the absent submodule do the same on A^T A).  Restated in the SAME operation order as csrc/fit.hip eigh_smallest_kernel:
 * sweeps over the pairs (p, r), p < r, row-cyclic; rotation t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (a_rr - a_pp) /
 * (2 a_pr); stop when the off-diagonal sum of squares is <= eps^2 times the diagonal's, or after 50 sweeps; the eigenvector of the
 * smallest diagonal entry (first on ties).  A [B][q*q] row-major; vec [B][q]; val [B]; sweeps [B] (may be NULL). */
void pgxo_eigh_smallest(const double* A, int q, int64_t B, double* vec, double* val, int32_t* sweeps);
#endif
