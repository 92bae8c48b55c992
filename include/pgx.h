/*
 * pgx.h — C ABI of libpgx.so, the MI355X (gfx950) implementation of Progressive-X's data-parallel hot path:
 * batched hypothesis scoring -> preference / Tanimoto validation -> PEARL unary costs + alpha-expansion labelling.
 *
 * The reference has no C ABI for this path: its boundary is a pybind11 module
 * (/root/reference/src/pyprogressivex/src/bindings.cpp:394-494) whose five entry points call straight into C++
 * templates.  The functions below are what a reference-side FFI for the hot loops would bind; each cites the
 * reference interface it replaces (paths relative to /root/reference/src/pyprogressivex/).  INTEGRATION.md shows the
 * ctypes / pybind11 stubs a maintainer would add.
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative pgx_status on failure
 * (pgx_last_error gives the message); the caller owns every host buffer, the library copies in and out; one pgx_ctx
 * per host thread and per GPU; all device work of a ctx is issued on one HIP stream owned by the ctx.
 */
#ifndef PGX_H
#define PGX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgx_ctx pgx_ctx;

enum pgx_status {
    PGX_OK = 0,
    PGX_ERR_INVALID = -1,   /* bad argument / call order */
    PGX_ERR_HIP = -2,       /* HIP runtime error */
    PGX_ERR_NOMEM = -3,
    PGX_ERR_NOCONVERGE = -4,/* max-flow sweep cap hit */
    PGX_ERR_COMM = -5,      /* RCCL error */
    PGX_ERR_RANGE = -6      /* fixed-point range check failed */
};

/* problem types; the estimator template argument of progx::ProgressiveX (progressivex_python.cpp:119,252,343,489,616) */
enum pgx_model_type {
    PGX_LINE2D = 0,           /* point (x,y);            model (a,b,c)                        findLines            */
    PGX_HOMOGRAPHY = 1,       /* point (x1,y1,x2,y2);    model H 3x3 row-major, one-way error findHomographies     */
    PGX_FUNDAMENTAL = 2,      /* point (x1,y1,x2,y2);    model F 3x3 row-major, Sampson       findTwoViewMotions   */
    PGX_PNP = 3,              /* point (u,v,X,Y,Z);      model [R|t] 3x4 row-major            find6DPoses          */
    PGX_VANISHING_POINT = 4,  /* point (xs,ys,xe,ye);    model (v0,v1,v2)                     findVanishingPoints  */
    PGX_HOMOGRAPHY_SYM = 5    /* point (x1,y1,x2,y2);    model [H | H^-1], symmetric transfer error               */
};

#define PGX_FIXED_SHIFT 32    /* min-cut energies are multiples of 2^-32 */
#define PGX_UNIQUE_ID_BYTES 128

/* ---- library / context ------------------------------------------------------------------------------------ */
int pgx_version(void);
int pgx_device_count(int *count);
const char *pgx_global_error(void);                       /* message of the last failure that had no ctx */
int pgx_create(int device_id, pgx_ctx **out);
void pgx_destroy(pgx_ctx *ctx);
const char *pgx_last_error(const pgx_ctx *ctx);
int pgx_model_dims(int model_type, int *point_dim, int *param_dim);
int pgx_sync(pgx_ctx *ctx);                               /* hipStreamSynchronize on the ctx stream */
int pgx_timer_start(pgx_ctx *ctx);                        /* hipEventRecord on the ctx stream */
int pgx_timer_stop(pgx_ctx *ctx, float *milliseconds);    /* record + synchronize + elapsed */
int pgx_timer_mark(pgx_ctx *ctx);                         /* record the stop event only (no host wait) */
int pgx_timer_elapsed(pgx_ctx *ctx, float *milliseconds); /* elapsed between start and mark (waits for the mark) */
int pgx_device_info(pgx_ctx *ctx, char *name, int name_len, int *cu_count, int64_t *hbm_bytes);

/* ---- resident data ------------------------------------------------------------------------------------------
 * replaces: cv::Mat points(N,d,CV_64F,...) (progressivex_python.cpp:75-93,203,335,454,567) — row-major N x d doubles. */
int pgx_set_points(pgx_ctx *ctx, int model_type, const double *points, int64_t n);
/* compound preference vector (progressive_x.h:141,524); NULL => all zeros */
int pgx_set_compound(pgx_ctx *ctx, const double *compound);
int pgx_get_compound(pgx_ctx *ctx, double *compound);

/* ---- a1: MSACScoringFunctionWithCompoundModel::getScore, batched (scoring_function_with_compound_model.h:61-125)
 * pgx_score = upload + launch + fetch.  models: M x param_dim row-major.  T2 = squared truncated threshold
 * (9/4 thr^2, progressive_x.h:523).  has_compound = compound_model->size() > 0 (:110).  exponent as :39/:120.
 * counts[m] = inlier_number (strict r^2 < T2, :85), values[m] = sum max(0, 1 - r^2/T2) (:94-97),
 * shared[m] = sum min(compound, pref) (:115-117), scores[m] = values - pow(shared, exponent) (:120).
 * masks (optional): M x ceil(n/64) uint64, bit i of row m = point i is an inlier of hypothesis m (:88).
 * The early exit at :105-106 is a pure function of (count, best): callers apply
 *   count + 1 < best_inlier_number  =>  Score()   in hypothesis order (see pyprogressivex/_proposal.py). */
int pgx_score(pgx_ctx *ctx, const double *models, int M, double T2, int has_compound, int exponent,
              int64_t *counts, double *values, double *shared, double *scores, uint64_t *masks);
/* pgx_score_upload: `models` is consumed before the call returns (copied, in locality order, into a pinned staging buffer of
 * the context); the transfer itself is asynchronous on the context's stream, as is everything that follows it there. */
int pgx_score_upload(pgx_ctx *ctx, const double *models, int M);
/* ---- SURVEY 8f "next", rank 1 (first slice): batched minimal solvers on the resident points.  samples[S][2] are point
 * indices; hypothesis s is generated straight into the resident hypothesis buffer (as after pgx_score_upload, in the
 * caller's order), so pgx_score_launch can follow without a model upload; models_out (may be NULL) copies them out.
 * Built: the 2-segment vanishing point solver (solver_vanishing_point_two_lines.h:147-185) and the 2-point line solver
 * [U-4] (samples[S][2], S x 3 models), the 4-point homography solver (samples[S][4], S x 9, h33 = 1,
 * DefaultHomographyEstimator progressivex_python.cpp:252, absent upstream), the 7-point fundamental matrix solver (samples[S][7], THREE model slots per
 * sample: 3S x 9, DefaultFundamentalMatrixEstimator progressivex_python.cpp:616, absent upstream), P3P (samples[S][3],
 * FOUR slots per sample: 4S x 12 [R|t], DefaultPnPEstimator progressivex_python.cpp:119, absent upstream); a degenerate sample
 * or an absent root yields a NaN model (never an inlier).  Other model types: PGX_ERR_INVALID.  `samples` is consumed before the
 * call returns; with models_out == NULL the call does not wait for the solver (the batch stays on the device). */
int pgx_solve_minimal(pgx_ctx *ctx, const int32_t *samples, int S, double *models_out);
/* The same with the samples DRAWN ON THE DEVICE by the in-repo counter-based generator (csrc/rng.hip.h: Philox4x32-10; sample s
 * of batch `batch` under `key` is a pure function of (key, batch, s); m = the resident model type's minimal sample size).
 * sampler 0: gcransac::sampler::UniformSampler - m distinct indices of range(n) (progressivex_python.cpp:121, 215-245);
 * sampler 1: NapsacSampler (sampler id 3 there, the default of findHomographies / findTwoViewMotions) - a uniform centre and m - 1
 * distinct entries of its row of the resident neighbourhood graph; a centre with fewer neighbours yields the row -1 .. -1 and a
 * NaN model (the iteration is spent);
 * sampler 2: ProsacSampler (sampler id 1 there; points ordered by quality) - sample s draws m - 1 distinct indices of the best
 * n_k - 1 points plus point n_k - 1, k = s + 1, n_k = entry s of the table pgx_sampler_prosac_set left on the device.
 * All absent upstream and seeded from std::random_device there.  No host RNG, no index upload; pyprogressivex/_rng.py and the
 * oracle produce the same rows.  samples_out (may be NULL): S x m indices. */
enum { PGX_SAMPLER_UNIFORM = 0, PGX_SAMPLER_NAPSAC = 1, PGX_SAMPLER_PROSAC = 2 };
int pgx_solve_minimal_sampled(pgx_ctx *ctx, int sampler, uint64_t key, uint32_t batch, int S, int32_t *samples_out, double *models_out);
/* PROSAC's hypothesis-generation set sizes for the resident points: subset_sizes[k - 1] = n_k of sample number k = 1 .. count
 * (Chum & Matas' growth function T'_n, a sequential floating-point recurrence: tabulated by the host once per sampler - the sample
 * numbers restart at 1 with every proposal, progressive_x.h:290, so one table of max-iterations entries serves every batch);
 * 0 = past the convergence bound (GC-RANSAC's 100 000 samples): uniform over all points.  Entries outside 0 .. n: PGX_ERR_INVALID.
 * pgx_set_points invalidates the table. */
int pgx_sampler_prosac_set(pgx_ctx *ctx, const int32_t *subset_sizes, int count);
/* Progressive NAPSAC on the same generator (gcransac::sampler::ProgressiveNapsacSampler<4>(&points, {16, 8, 4, 2}, sample_size,
 * {w1, h1, w2, h2}, 0.5), progressivex_python.cpp:229-238, sampler id 2 - absent upstream, restated after Barath et al.).  HOST
 * code, no context and no GPU: every sample updates the hit counters and neighbourhood sizes the next one reads, so a draw is one
 * sequential chain (csrc/sampler_host.hip).  create: the grid layers over the first four coordinates of pts [n][d] (points in
 * quality order; sizes = the four image extents; layers = cells per dimension, finest first; m = the minimal sample size, 2 .. 8).
 * draw: the `count` samples of one proposal (the sampler state restarts, progressive_x.h:290) as count x m indices - sample k <
 * min(count, max_local) local (centre k, m - 2 distinct members of its growing cell neighbourhood, the neighbourhood's last member),
 * the others and the points without a large enough cell through PROSAC with subset size tops[k] (0 = uniform); growth_local [n] =
 * PROSAC's growth function for m - 1 points and max_local samples.  A pure function of (key, batch) and the data:
 * pyprogressivex/_rng.py pnapsac_samples returns the same rows.  Errors: pgx_global_error. */
typedef struct pgx_pnapsac pgx_pnapsac;
int pgx_pnapsac_create(const double *pts, int64_t n, int d, const double *sizes, const int32_t *layers, int n_layers, int m,
                       pgx_pnapsac **out);
int pgx_pnapsac_draw(pgx_pnapsac *sampler, uint64_t key, uint32_t batch, int32_t count, const int32_t *tops,
                     const int64_t *growth_local, int64_t max_local, int32_t *out);
void pgx_pnapsac_destroy(pgx_pnapsac *sampler);
/* host helpers of the samplers that draw from the CALLER's generator (the default numpy-stream samplers of pyprogressivex/_proposal.py: the
 * uniform / NAPSAC / PROSAC samplers of progressivex_python.cpp:215-245 return distinct indices by construction): no GPU, no context.
 * rows_with_duplicates: out_bad[r] = 1 iff row r of s [count][m] holds a repeated value.  fisher_yates_rows: row rows[t] of s [.][m]
 * becomes the first m entries of a Fisher-Yates shuffle of its range whose step j swaps positions j and j + draws[t][j]. */
int pgx_host_rows_with_duplicates(const int64_t *s, int64_t count, int m, uint8_t *out_bad);
int pgx_host_fisher_yates_rows(const int64_t *draws, const int64_t *rows, int64_t k, int m, int64_t *s);
int pgx_score_launch(pgx_ctx *ctx, double T2, int has_compound, int want_masks);   /* asynchronous */
int pgx_score_fetch(pgx_ctx *ctx, int exponent, int64_t *counts, double *values, double *shared,
                    double *scores, uint64_t *masks);
/* The inlier set of hypothesis `row` of the last launch that produced masks (pgx_score with masks / pgx_score_launch with
 * want_masks), as ascending point indices - the `inliers` vector getScore fills (scoring_function_with_compound_model.h:88) -
 * compacted on the device.  index: room for n entries. */
int pgx_score_inliers(pgx_ctx *ctx, int row, int32_t *index, int64_t *count);
/* what one pgx_score_launch reads+writes at minimum (points + models + compound + results), for rooflines */
int pgx_score_algorithmic_bytes(pgx_ctx *ctx, int want_masks, int64_t *bytes, int64_t *pairs);
/* Work counters of one scoring launch of the resident batch (a separate, untimed launch of the same kernels with counting
 * switched on): [0] (point, hypothesis) pairs, [1] (hypothesis, 64-point group) pairs, [2] group steps that survived the
 * bound test (64 f32 filter evaluations each), [3] exact FP64 residual evaluations (-1: not counted on this path),
 * [4] inlier pairs, [5] with PGX_VERIFY=1 in the environment at pgx_create: inlier pairs - by the exact residual over EVERY pair of
 * the batch - that the group bound or the f32 filter discarded (must be 0: a hole in a filter proof otherwise), else -1,
 * [6] path (1 = every pair visited, 2 = cull + group-major), [7] filter (0 none, 1 f64, 2 f32). */
int pgx_score_stats(pgx_ctx *ctx, double T2, int has_compound, int64_t stats[8]);
/* Per-kernel HIP-event timing of the scoring launches on the context's stream (bench.py's roofline block).  on = 1: every
 * pgx_score_launch records two events around its DOMINANT kernel (group-major scoring, or the chunked kernel); on = 2:
 * events around every kernel of the launch (an event costs ~5 us on the stream, so the breakdown is taken outside timed
 * regions).  pgx_score_kernel_times returns the durations of the last launch in ms (0 where not recorded): [0] cull (or the chunked kernel), [1] group-major scoring (the dominant kernel; 0 on the chunked path),
 * [2] finish / reduce, [3] exact evaluation of the queued candidates (0 when it runs inside the group-major kernel). */
int pgx_score_profile(pgx_ctx *ctx, int on);
/* Diagnostic read-back of what pgx_set_points derived for the score path (tests compare the device preprocessing with the
 * host version bit for bit): what = 0 sorted order (n int32), 1 group + super-group rows ((groups + supers) x 12 f32),
 * 2 / 3 the group-blocked f64 / f32 row copies, 4 the sorted f32 rows; 5 the integer accumulators of the last launch, replicas
 * summed, in the caller's hypothesis order (3 x Mpad u64: count | value | shared in 2^-q fixed point - what pgx_score_allreduce
 * adds across ranks).  bytes must match exactly. */
int pgx_score_debug_fetch(pgx_ctx *ctx, int what, void *out, int64_t bytes);
/* Test hook: launch geometry of the group-major score path (results must not depend on it - integer accumulation of per-pair
 * fixed point; tests/test_gpu_parity.py).  what = 0 waves per 64-point group (0 = automatic), 1 a group's waves on one XCD
 * (-1 automatic, 0, 1), 2 replicas of the accumulators (0 = automatic, else a multiple of 8), 3 candidates of 64 from which a
 * step is evaluated in place instead of queued (1..65), 4 segments of groups per hypothesis word in the cull kernel.  Not an
 * environment switch: nothing in the product path calls it. */
int pgx_score_debug_geometry(pgx_ctx *ctx, int what, int value);
int pgx_score_kernel_times(pgx_ctx *ctx, float ms[4]);

/* ---- a2/a3: Model::setPreferenceVector (progx_model.h:70-87) + the three reductions of isPutativeModelValid
 * (progressive_x.h:583-585).  The preference vector is kept on the device in `slot` (>=0) for a4; pref_out optional. */
int pgx_preference(pgx_ctx *ctx, const double *model, double T2, int slot, double *pref_out,
                   double *dot, double *pref_sqnorm, double *comp_sqnorm);
int pgx_get_preference(pgx_ctx *ctx, int slot, double *pref_out);
/* ---- a4: updateCompoundModel (progressive_x.h:597-624): compound[i] = max_k stored pref_k[i] (stale vectors, as the
 * reference does).  K == 0 leaves the compound vector untouched (:600-601). */
int pgx_compound_update(pgx_ctx *ctx, const int32_t *slots, int K, double *compound_out);

/* ---- a6: dataEnergyFunctor / EnergyDataStructure (PEARL.h:18-56,82-128): unary table N x (K+1), label K = outlier,
 * quantised to multiples of 2^-32 for the min-cut.  Kept resident; Dq_out optional. */
int pgx_pearl_unary(pgx_ctx *ctx, const double *models, int K, double threshold, double lambda, int64_t *Dq_out);
int pgx_set_unary_q(pgx_ctx *ctx, const int64_t *Dq, int64_t n, int L);   /* tests: inject a table (no points needed) */

/* ---- a20 consumer: neighbourhood graph as symmetric CSR (PEARL.h:532-536 setNeighbors loop).
 * off[n+1], idx[off[n]], mult[off[n]]: each undirected pair appears in both rows with the same multiplicity
 * (= number of directed entries in the raw getNeighbors lists; a symmetric raw list gives 2, U-6). */
int pgx_set_graph(pgx_ctx *ctx, int64_t n, const int32_t *off, const int32_t *idx, const int32_t *mult);

/* ---- a20 (SURVEY 8f "next", rank 2): the graph itself.  Replaces FlannNeighborhoodGraph(&points, radius) +
 * getNeighbors(i) (progressivex_python.cpp:104,207,339,458,571) and the setNeighbors loop (PEARL.h:532-536); the FLANN
 * code is absent from the snapshot [U-7], so the lists are restated deterministically:
 *   PGX_GRAPH_KNN_IN_BALL  the k nearest neighbours with squared distance <= radius^2 (drop-in default, k = 5)
 *   PGX_GRAPH_KNN          the k nearest neighbours (radius ignored)
 *   PGX_GRAPH_BALL         every point with squared distance <= radius^2 (k ignored; symmetric lists: multiplicity 2)
 * points: n x d doubles (d = 2..5, the data space the reference hands to FLANN), ranking by (squared distance, index),
 * squared distance summed in dimension order without contraction.  The symmetric CSR (multiplicity = number of
 * directed list entries of the pair, U-6) stays resident exactly as after pgx_set_graph; pgx_graph_fetch copies it
 * out (off[n+1]; idx/mult[arcs], may be NULL). */
enum { PGX_GRAPH_KNN_IN_BALL = 0, PGX_GRAPH_BALL = 1, PGX_GRAPH_KNN = 2 };
int pgx_graph_build(pgx_ctx *ctx, const double *points, int64_t n, int d, int kind, double radius, int k, int64_t *arcs);
int pgx_graph_fetch(pgx_ctx *ctx, int32_t *off, int32_t *idx, int32_t *mult);
/* sites and directed arcs of the graph resident NOW (after pgx_graph_build or pgx_set_graph; 0, 0 when none): what the buffers
 * of pgx_graph_fetch must hold (a caller that cached the sizes of an earlier graph would be overrun) */
int pgx_graph_size(pgx_ctx *ctx, int64_t *n, int64_t *arcs);

/* ---- a8/a19: GCoptimizationGeneralGraph::{setLabel, expansion, whatLabel} as used by PEARL::labeling
 * (PEARL.h:507-551).  lambda = spatial coherence weight of ONE directed neighbour entry (PEARL.h:76-78),
 * label_cost = model_complexity_weight (PEARL.h:144,529).  Energies are returned both as the exact fixed-point
 * integer and as double (= energy_q / 2^32).
 * PEARL builds a fresh GCO engine - i.e. starts from the all-zero labelling - for every labelling it cannot warm-start
 * (PEARL.h:507-508, 541-547), mostly with the same instances plus one.  pgx_expansion therefore keeps the labels after each
 * move of the FIRST cycle of an expansion that starts from the all-zero labelling uploaded by pgx_set_labels; when the next
 * such expansion has the same leading unary columns (same model bytes, threshold, lambda, points, graph, label cost) it
 * restores the state behind those moves instead of solving their min-cuts again.  A move is a deterministic function of
 * (labelling, alpha, the columns of the labels present): results are bit-identical with and without the memo (PGX_MF_MEMO=0). */
int pgx_set_labels(pgx_ctx *ctx, const int32_t *labels, int64_t n);
int pgx_get_labels(pgx_ctx *ctx, int32_t *labels);
int pgx_energy(pgx_ctx *ctx, double lambda, double label_cost, int64_t *energy_q, double *energy);
int pgx_expand_alpha(pgx_ctx *ctx, double lambda, double label_cost, int alpha, int64_t *changed);
int pgx_expansion(pgx_ctx *ctx, double lambda, double label_cost, int max_cycles,
                  int64_t *energy_q, double *energy, int *cycles);
/* U-8: what GCO-v3's expansion() does for an energy WITHOUT smooth costs, which is what PEARL hands it when
 * spatial_coherence_weight == 0 (PEARL.h:523-536: no setSmoothCost, no setNeighbors; :550-551 expansion()): its
 * solveSpecialCases() labels "data costs only" by per-site argmin and "data costs + per-label costs" by the greedy
 * facility-location heuristic solveGreedy() instead of running alpha-expansion [UPSTREAM-MEMORY, restated in
 * oracle/pgx_oracle.c].  Works on the resident unary table; the resident labelling is OVERWRITTEN (the heuristic
 * ignores the starting labels).  opened = labels in use. */
int pgx_greedy_labeling(pgx_ctx *ctx, double label_cost, int64_t *energy_q, double *energy, int *opened);

/* counters of the last pgx_expansion / pgx_expand_alpha: [0]=min-cuts solved, [1]=push-relabel sweeps,
 * [2]=global relabels (BFS passes), [3]=BFS levels, [4]=sites relabelled by moves, [5]=wave passes,
 * [6]=work-list sweeps, [7]=moves skipped because the labelling had not changed since that label's last move, which relabelled nothing */
int pgx_expansion_stats(pgx_ctx *ctx, int64_t stats[8]);
/* which min-cut solver finished the moves since pgx_create (the cut is the same under all of them; tests use this to make
 * sure the solver under test is the one that ran): [0]=one workgroup on the whole graph (<= 8192 sites), [1]=first-cycle
 * moves restored from the memo of the previous expansion from the all-zero labelling (same unary columns: not solved again), [2]=one workgroup on the compacted region of open sites, [3]=level-synchronous launches
 * (maxflow.hip), [4]=region moves declined (too many open sites / a sink that could not be promoted), [5]=tile moves handed back */
int pgx_expansion_paths(pgx_ctx *ctx, int64_t paths[6]);
/* how the level-synchronous solver (maxflow.hip; the max-flow behind PEARL.h:549-551) spent its dependent steps since pgx_create:
 * [0]=launches of the persistent one-XCD round kernel (maxflow_xcd.hip.h), [1]=rounds (search + list sweeps) run inside them,
 * [2]=of those launches, the ones that ended with no listed site reaching t, [3]=global relabels run as one launch (level loop
 * inside), [4]=sites visited by the list sweeps inside those launches (a multiple of 16), [5]=sites visited by the list sweeps
 * launched one by one (the labelling roofline charges a list sweep by its list, not by the graph), [6]=BFS levels and [7]=list
 * sweeps that ran inside those persistent launches (dependent steps that were not launches) */
int pgx_expansion_schedule(pgx_ctx *ctx, int64_t out[8]);
/* whole-graph one-workgroup moves (pgx_expansion_paths [0]) ENQUEUED since pgx_create, by kernel: [0]=the LDS-resident solver
 * (graphs of <= 1024 sites and <= 8192 arcs: arc capacities, excesses, heights and hub words in LDS, the site's own state in the
 * registers of its thread - csrc/maxflow_tile.hip t_mini_kernel; PGX_TILE_MINI=0 switches it off), [1]=the solver that keeps the
 * capacities in memory (<= 8192 sites, t_move_kernel).  Same binary problem and cut as every other schedule; tests use the counts to
 * know which kernel they compared with the oracle. */
int pgx_one_workgroup_launches(pgx_ctx *ctx, int64_t out[2]);

/* ---- a9 (SURVEY 8f "next", rank 3): the data pass of estimator.estimateModelNonminimal(...) as called by
 * PEARL::parameterEstimation (PEARL.h:374-380) and by the proposal engine's local optimisation.  The device accumulates
 *   out = sum_i W_i * sum_{rows a of point i} a a^T     (upper triangle of the q x q matrix, row-major, q(q+1)/2 values)
 * over the selected resident points; the small dense solve stays with the caller.  W_i = weights[i]^weight_power when
 * use_weights != 0, else 1.  The per-point weights are RESIDENT: pgx_set_weights uploads them once per point set and checks
 * len == n (a host pointer without a length used to be read for n doubles on every call; the reference has the same
 * unchecked read at solver_vanishing_point_two_lines.h:204-207 via progressivex_python.cpp:381).  pgx_set_weights(NULL, 0)
 * clears them; pgx_set_points invalidates them.  Rows by kind (points as given to pgx_set_points):
 *   PGX_GRAM_AFFINE   a = (1, p_0 .. p_{d-1})                                         q = d+1   (means, covariances)
 *   PGX_GRAM_DLT_H    the two DLT rows of a correspondence, params = (s1,cx1,cy1,s2,cx2,cy2)    q = 9
 *   PGX_GRAM_EPI_F    the epipolar row (x2x1,x2y1,x2,y2x1,y2y1,y2,x1,y1,1), same params         q = 9
 *   PGX_GRAM_VP       solver_vanishing_point_two_lines.h:212-217                               q = 3
 *   PGX_GRAM_PNP_GN   Gauss-Newton rows (J_u, r_u), (J_v, r_v) at the pose params = [R|t] 3x4  q = 7
 * Selection: PGX_SEL_INDEX (index[m], host) or PGX_SEL_LABEL (label == `label` on the resident labelling).
 * count = selected points, bad = points skipped because the row is undefined (PnP: depth ~ 0). */
enum { PGX_GRAM_AFFINE = 0, PGX_GRAM_DLT_H = 1, PGX_GRAM_EPI_F = 2, PGX_GRAM_VP = 3, PGX_GRAM_PNP_GN = 4 };
enum { PGX_SEL_INDEX = 0, PGX_SEL_LABEL = 1 };
int pgx_set_weights(pgx_ctx *ctx, const double *weights, int64_t len);
int pgx_gram(pgx_ctx *ctx, int kind, const double *params, int nparams, int sel, const int32_t *index, int64_t m,
             int label, int use_weights, int weight_power, double *out, int64_t *count, int64_t *bad);

/* All instances of one PEARL iteration at once (PEARL.h:369-390 runs these per instance): Gram matrices of the points with
 * label k under parameter block k (params[K][nparams]), k = 0..K-1, and the residual sums of model k over label k.  One
 * launch each; out[k] / sums[k] are bit-identical to the single-label calls pgx_gram(PGX_SEL_LABEL, k) / pgx_residual_sum. */
int pgx_gram_labels(pgx_ctx *ctx, int kind, const double *params, int nparams, int K, int use_weights,
                    int weight_power, double *out, int64_t *count, int64_t *bad);
int pgx_residual_sums(pgx_ctx *ctx, const double *models, int K, double *sums);

/* Batched form for the inner RANSAC of the local optimisation: B index selections of m points each (index[B][m]), one
 * parameter block per selection (params[B][nparams]), optional per-entry weights already gathered by the caller
 * (weights_sel[B][m]).  out[B][q(q+1)/2], bad[B] (optional).  One launch, one wave per selection. */
int pgx_gram_batch(pgx_ctx *ctx, int kind, const double *params, int nparams, const int32_t *index, int B, int m,
                   const double *weights_sel, int weight_power, double *out, int32_t *bad);

/* U-14 (DefaultFundamentalMatrixEstimator's model validity, progressivex_python.cpp:616; sources absent, restated from the
 * literature): support of a fundamental matrix over all resident correspondences.  counts[0] = points whose squared
 * Sampson distance is < T2 (the scorer's inliers), counts[1] = those of them whose symmetric epipolar distance
 * r^2 (1 / |F x1|^2 + 1 / |F^T x2|^2) is < S2.  The points must be those of a fundamental-matrix problem. */
int pgx_epipolar_support(pgx_ctx *ctx, const double *F, double T2, double S2, int64_t counts[2]);

/* The whole Gauss-Newton refit of B pose hypotheses in one launch (one wave per selection): from inits[B][12] (row-major
 * [R | t]), `iterations` steps of  normal equations over the selection (the PGX_GRAM_PNP_GN rows) -> delta = pinv(J^T J)
 * (-J^T r) with numpy.linalg.pinv's cut-off 6 eps -> R <- exp([delta_omega]_x) R, t += delta_t, stopping at |delta| < 1e-12.
 * This is the non-minimal solver the local optimisation and PEARL call through estimator.estimateModelNonminimal
 * (PEARL.h:374-380; GC-RANSAC's inner RANSAC [UPSTREAM-MEMORY U-12]); the reference's own solver for this model is OpenCV's
 * iterative PnP, restated as plain Gauss-Newton on the reprojection error.  out[B][12]; status[b] = 1 if
 * selection b produced a finite pose (0: a point behind / on the camera plane, non-finite sums, fewer than 4 points). */
int pgx_pnp_refine_batch(pgx_ctx *ctx, const double *inits, const int32_t *index, int B, int m, const double *weights_sel,
                         int weight_power, int iterations, double *out, int32_t *status);

/* The small dense solve of the non-minimal refits behind the C ABI (round 6; first step towards one call per local-optimisation
 * round): the eigenvector of the SMALLEST eigenvalue of B symmetric q x q matrices (q <= 9: the 9 x 9 A^T A of the normalised DLT /
 * 8-point rows, the 3 x 3 of the vanishing-point solver).  Replaces Eigen::SelfAdjointEigenSolver as the refit solvers use it
 * (solver_vanishing_point_two_lines.h:227 in-tree; estimateModelNonminimal of the absent submodule, PEARL.h:374-380), until now
 * numpy's LAPACK on the host.  Cyclic Jacobi in FP64, one lane per matrix, in the operation order of the oracle's
 * pgxo_eigh_smallest: device and oracle return the same bits; both agree with LAPACK to ~1e-13 on the (sign-normalised)
 * eigenvector for well-separated eigenvalues.  A [B][q*q] row-major (host); vec [B][q], val [B] (host).  The drop-in calls use it
 * when refit_solver="jacobi" (default "lapack": unchanged results). */
int pgx_eigh_smallest_batch(pgx_ctx *ctx, const double *A, int q, int64_t B, double *vec, double *val);

/* ---- the inlier/outlier graph cut of GC-RANSAC's local optimisation.
 * Replaces gcransac::GCRANSAC::labeling as reached from proposal_engine->run (progressive_x.h:294-299; settings
 * spatial_coherence_weight / threshold at :541-545).  The graph-cut-ransac sources are absent from the snapshot, so the
 * energy is restated from memory of upstream [U-12, DESIGN.md 5.8]: e_i = clamp(r_i^2 / T2, 0, 1);
 *   unary: r_i^2 <= T2 ? (outlier: (1-lambda)(1-e_i), inlier: 0) : (outlier: 0, inlier: (1-lambda) e_i);
 *   pairwise, every undirected neighbour pair of the resident graph once: both outliers lambda (e_i+e_j)/2, labels differ
 *   lambda, both inliers 0.  One exact s-t cut on the device (terms in 2^-32 fixed point); inliers = sink segment.
 * flags[n] (host): 1 = inlier, 0 = outlier; count = number of inliers.  Needs pgx_set_points + a graph over the points. */
int pgx_gc_labeling(pgx_ctx *ctx, const double *model, double T2, double lambda, int32_t *flags, int64_t *count);
/* The same cut, returning the inliers' indices in ascending order (index: capacity n; *count of them are written) instead of
 * n flags: what GCRANSAC::labeling hands to the inner sampler.  Same cut bit for bit; compaction on the device. */
int pgx_gc_inliers(pgx_ctx *ctx, const double *model, double T2, double lambda, int32_t *index, int64_t *count);

/* ---- a9: PEARL::parameterEstimation bookkeeping (PEARL.h:342-352, 369-371, 388-390) */
int pgx_bucket(pgx_ctx *ctx, int L, int64_t *counts, int32_t *order);     /* order optional: stable, ascending index */
int pgx_residual_sum(pgx_ctx *ctx, const double *model, int label, double *sum);

/* ---- multi-GPU (no reference counterpart): either the hypotheses are sharded over ranks and every rank holds all
 * points (RCCL all-gather of the per-hypothesis (count, value, shared) triples), or the points are sharded and every rank scores all
 * hypotheses (RCCL all-reduce of the integer accumulators); all-reduce(max) of the compound vector */
int pgx_comm_unique_id(uint8_t id[PGX_UNIQUE_ID_BYTES]);
int pgx_comm_init(pgx_ctx *ctx, int nranks, int rank, const uint8_t id[PGX_UNIQUE_ID_BYTES]);
int pgx_comm_destroy(pgx_ctx *ctx);
int pgx_comm_barrier(pgx_ctx *ctx);
int pgx_comm_allreduce_max_f64(pgx_ctx *ctx, double *value);                /* host scalar in/out, via device */
int pgx_score_allgather(pgx_ctx *ctx);                                      /* after pgx_score_launch, asynchronous */
int pgx_score_fetch_all(pgx_ctx *ctx, int exponent, int64_t *counts, double *values, double *shared,
                        double *scores);                                    /* nranks*M entries, rank-major */
/* The same exchange, overlapped with the scoring of the NEXT batch (two batches in flight, slot = 0 / 1): _begin right behind
 * pgx_score_launch copies the launch's result block aside and runs all-gather + copy to pinned memory on a second stream; _end
 * waits for that slot and unpacks it like pgx_score_fetch_all (M = the launch's batch size).  Bitwise the serial results.
 * While a slot is in flight every other collective of the context is refused (one communicator, two streams). */
int pgx_score_allgather_begin(pgx_ctx *ctx, int slot);
int pgx_score_allgather_end(pgx_ctx *ctx, int slot, int exponent, int64_t *counts, double *values, double *shared, double *scores);
/* Point-sharded scoring (north_star: "RCCL all-reduce of per-model inlier counts"; the batched form of getScore,
 * scoring_function_with_compound_model.h:78-121, with the loop over the points split across ranks): every rank holds a SLICE
 * of the points and scores all M hypotheses against it; the integer accumulators of the launch (count, 2^-q fixed-point value
 * and shared support) are summed over the ranks with ncclAllReduce(sum, uint64) - exact in any order, so the reduced table is
 * bitwise the table of one GPU holding all the points.  pgx_score_set_global_n(total points of the job; 0 = this context's own)
 * makes the ranks agree on q.  pgx_score_allreduce: after pgx_score_launch, asynchronous; pgx_score_fetch then returns the
 * reduced table.  _begin / _end: the overlapped form (M rows, one table). */
int pgx_score_set_global_n(pgx_ctx *ctx, int64_t n_total);
int pgx_score_allreduce(pgx_ctx *ctx);
int pgx_score_allreduce_begin(pgx_ctx *ctx, int slot);
int pgx_score_allreduce_end(pgx_ctx *ctx, int slot, int exponent, int64_t *counts, double *values, double *shared, double *scores);
int pgx_compound_allreduce_max(pgx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PGX_H */
