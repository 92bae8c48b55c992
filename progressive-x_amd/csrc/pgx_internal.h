// pgx_internal.h — context layout and helpers shared by the translation units of libpgx.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pgx.h"
#include "residuals.hip.h"

namespace pgx {

// growable device buffer
struct StagedCopy { void* dst; size_t off, bytes; };
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct MaxflowState;  // maxflow.hip
struct TileState;     // maxflow_tile.hip
struct CommState;     // comm.cpp

}  // namespace pgx

struct pgx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t kev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // pgx_score_profile: [0] start, [1] after cull, [2] after group-major, [4] after exact, [3] after finish
    int score_profile = 0;
    std::string err;
    int cu_count = 0;

    // resident problem data
    int model_type = -1, D = 0, P = 0;
    int64_t n = 0;
    pgx::DevBuf pts, comp;
    pgx::DevBuf pmax;        // per point max(|coords the filter scales by|, 1)  (score filter)
    pgx::DevBuf pts32;       // N x 8 f32: coordinates + filter scale (FP32 pre-filter)
    double umax = 0.0;       // max |observed image coordinate| over all points
    double fscale = 0.0;     // max(1, max |coordinate|) over all points: isotropic pre-scaling of the minimal solvers
    int filter_enabled = 1;  // PGX_NO_FILTER: 1 = no rejection filter, 2 = FP64 filter only (A/B, debugging)
    int last_score_filtered = 0;
    int last_score_path = 0;       // 1 = chunked kernel (every pair visited), 2 = cull + group-major
    int score_stats = 0;           // set by pgx_score_stats for one launch: work counters in stats_buf
    int verify = 0;                // PGX_VERIFY=1: pgx_score_stats also re-decides every pair exactly and counts contradictions (score.hip)
    pgx::DevBuf stats_buf;
    // spatially sorted copies for the score kernel (group-level rejection); aliases of the originals
    // when point_sort is off
    int point_sort = 0;          // 1: pts_s / pts32_s / pmax_s / comp_s hold the points in Morton order, pperm maps back
    int group_filter = 1;        // PGX_NO_GROUP=1 disables the sorted copies and the group test (A/B)
    int comp_dirty = 0;          // comp changed since comp_s was gathered
    pgx::DevBuf pts_s, pts32_s, pmax_s, comp_s, pperm, gbounds, masks_s;
    pgx::DevBuf pts_g, p32_g;    // group-blocked SoA copies of the sorted rows: [group][coordinate][64] (group-major kernel)
    int setpoints_host = 0;      // PGX_SETPOINTS_HOST=1: round 1's host preprocessing in pgx_set_points (A/B, cross-check)
    int gc_flip = 1;             // PGX_GC_FLIP=0: the inlier / outlier cut in its original orientation (pointwise.hip gc_labeling_launch)
    int score_dense_min = 32;    // steps with at least this many candidates of 64 are evaluated in place, not queued (PGX_SCORE_DENSE; 65 = never)
    int score_cull_segs = 256;   // segments of groups per hypothesis word in the cull kernel (PGX_SCORE_CULL_SEGS; 8192 waves at M = 2048)
    int score_nrep = 0;         // replicas of the integer accumulators (PGX_SCORE_NREP, multiple of 8); 0 = automatic: 8 when a group's waves share an XCD, else 1
    int score_cull = 1;          // cull + survivor kernels instead of in-kernel group skipping (PGX_SCORE_NO_CULL=1: A/B)
    double sp_kd_weight = 0.25;  // PGX_SP_KD_W: weight of the 3-D part against the observed pair in the k-d order (1 = box normalisation)
    int sp_kd = 1;               // PGX_SP_KD=0: Morton order of the points of a pose problem instead of the k-d order (setpoints.hip)
    pgx::DevBuf weights_scratch; // scratch of the k-d build (64-bit keys, sort workspace, per-node extents)
    int score_group_xcd = -1;    // PGX_SCORE_GROUP_XCD: 1 = a group's workgroups on one XCD (8x less row fetch, per-XCD accumulator replicas), 0 = part p on XCD p;
                                 // -1 (default) = 1 for a locality-ordered batch (few hypothesis words per group have survivors), 0 otherwise (score.hip)
    int score_split = 0;         // waves per 64-point group in the group-major kernel (PGX_SCORE_SPLIT); 0 = 8 with the spread mapping, 5 co-located (score.hip)
    pgx::DevBuf cull_lists, cull_counts;
    pgx::DevBuf gc;          // inlier/outlier graph cut: e[n] | dq[2][n] | wq[E] | labels[n]
    pgx::DevBuf gc_sel;      // ... the inliers' indices (pgx_gc_inliers): index[n] | count | select scratch
    int score_xcd_map = 1;       // XCD-aware block mapping of the score kernel (PGX_SCORE_NO_XCD=1 disables)
    int score_blocks_per_cu = 64;  // grid over-decomposition of the score kernel (PGX_SCORE_BLOCKS_PER_CU)

    // scoring
    int M = 0, Mpad = 0, chunks = 0;
    int64_t chunk = 0, words = 0;
    bool have_masks = false;
    int score_has_compound = 0;
    int64_t score_global_n = 0;  // pgx_score_set_global_n: the fixed-point scale of the sums is taken from max(n, this) - the ranks of a
                                 // point-sharded job (pgx_score_allreduce) then add integers of the SAME scale: bitwise the unsharded sums
    unsigned long long* last_acc = nullptr;   // integer accumulators [nrep][3][Mpad] of the last group-major launch (device order), or nullptr
    int last_nrep = 0;
    double last_qscale = 0.0;
    int last_acc_M = 0, last_acc_Mpad = 0;    // the batch the accumulators belong to (an upload / solve after the launch makes them stale)
    pgx::DevBuf models, pcnt, pval, psh, counts, values, shared, masks;
    pgx::DevBuf perm;        // perm[sorted position] = caller's hypothesis index (locality ordering, capi.hip)
    int score_sort = 1;      // PGX_NO_SORT=1 keeps the caller's order (A/B)
    pgx::DevBuf g_counts, g_values, g_shared;  // all-gathered results (multi-GPU)

    // preference slots + reductions
    std::vector<pgx::DevBuf> slots;
    pgx::DevBuf red_partials, red_out;

    // PEARL
    int L = 0;          // labels of the resident unary table
    int64_t dq_n = 0;   // sites of the resident unary table
    pgx::DevBuf dq;     // label-major [L][n] int64
    int64_t dq_max = 0; // upper bound of the table's entries (pgx_pearl_unary: 2^33; pgx_set_unary_q: the actual maximum)
    pgx::DevBuf kmodels;
    pgx::DevBuf labels; // int32 [n]
    int64_t labels_n = 0;
    // First-cycle memo of pgx_expansion (capi.hip): PEARL starts every labelling it cannot warm-start from the all-zero labelling
    // (PEARL.h:507-508, 541-547), so the first cycle's move alpha sees labels < alpha only and is a deterministic function of the
    // unary columns 0..alpha, lambda, h and the graph.  The labels after each first-cycle move are kept; a later expansion from
    // zeros whose leading unary columns are THE SAME (exact identity of what pgx_pearl_unary computed them from) restores the
    // state behind that prefix instead of solving its min-cuts again.
    struct ExpansionMemo {
        std::vector<std::string> ident;   // identity of the unary column each kept move was solved with
        std::vector<int64_t> changed;     // sites the move relabelled
        int valid = 0;                    // moves 0 .. valid-1 of the last from-zeros expansion are kept
        pgx::DevBuf snaps;                // [cap][n] int32: labels after move alpha
        int64_t n = 0, lq = 0, hq = 0, graph_version = -1;
        int cap = 0;
    } memo;
    // The last pgx_expansion that ended on a FIXED POINT (its final cycle relabelled nothing), with what it was computed from: a call
    // with the same columns, weights, graph and the labels it left is the same deterministic computation - one cycle of no-ops, the same
    // energy - and is answered from here (capi.hip pgx_expansion).  PEARL ends every run with exactly such a call: the iteration that
    // finds "nothing changed" labels once more with the models and the warm start of the one before (PEARL.h:429-467).
    struct ExpansionDone {
        int valid = 0;
        std::vector<std::string> ident;
        int64_t lq = 0, hq = 0, n = 0, graph_version = -1, energy_q = 0;
        int64_t labels_version = -1;   // pgx_ctx::labels_version when the fixed point was recorded (the unary table's identity is `ident`)
    } last_done;
    int64_t labels_version = 0;           // bumped by EVERY writer of `labels` (pgx_set_labels, every expansion move, the greedy labelling): the
                                          // identical-call shortcut compares it instead of trusting each writer to clear last_done
    int mf_done_verify = 0;               // PGX_MF_DONE_VERIFY=1: on a shortcut hit run the real verifying cycle and check 0 changes + equal energy
    std::vector<std::string> unary_ident; // per label: what its unary column was computed from (pgx_pearl_unary); empty = unknown (injected table)
    int64_t points_version = 0;           // bumped by pgx_set_points
    int labels_all_zero = 0;              // the resident labelling is the all-zero one pgx_set_labels uploaded (no move has run since)
    int mf_memo = 1;                      // PGX_MF_MEMO=0: no first-cycle memo (A/B)
    int64_t memo_hits = 0;                // moves restored from the memo (pgx_expansion_paths[1])
    int labels_max = 0;          // largest label pgx_set_labels uploaded (the moves index per-label tables with the labels: checked against L)
    // graph (symmetric CSR)
    int64_t gn = 0, gE = 0;
    int max_degree = 0;
    int64_t max_row_mult = 0;
    pgx::DevBuf goff, gidx, gmult, grev;
    pgx::MaxflowState* mf = nullptr;
    // tile-resident min-cut (maxflow_tile.hip), the default path of an expansion move
    pgx::TileState* tile = nullptr;
    int64_t graph_version = 0;   // bumped whenever the resident graph changes (graph_build_reverse)
    pgx::DevBuf gorder;          // sites in the Morton order of the coordinates the graph was built on (graph.hip); gorder_n == gn when valid
    int64_t gorder_n = 0;
    int mf_tile_batch = 1;       // PGX_MF_TILE_BATCH=0: one host round trip per one-workgroup move (A/B)
    int mf_tile = 1;             // PGX_MF_TILE=0: level-synchronous schedule of maxflow.hip for every move (A/B)
    int tile_order = 1;          // sites of the tile path in the Morton order of the graph's coordinates (0: the caller's order)
    int tile_single_max = 8192;  // graphs up to this many sites: the whole move in one launch of one workgroup
    int tile_expansion_max = 1024;   // PGX_TILE_EXPANSION_MAX: expansion moves on larger graphs (up to tile_single_max) try the region path first (maxflow.hip expand_alpha_on);
                                     // = the LDS-resident whole-graph kernel's limit: beyond it the region path with ITS LDS-resident solver is as fast or faster
                                     // (2 000 sites 1.99 vs 2.03 ms per expansion, 5 000 sites 6.2 vs 7.7; unihouse, 2 084 points: 74 -> 67 ms per call)
    int mf_xcd = 1;              // PGX_MF_XCD=0: no persistent one-XCD rounds (maxflow_xcd.hip.h); read at pgx_create like the switches above
    int mf_xcd_search = 1;       // PGX_MF_XCD_SEARCH=0: no one-launch global relabels
    int mf_xcd_min_depth = 24;   // PGX_MF_XCD_MIN_DEPTH: a search runs as one launch when the previous search of its kind was deeper than this
    long long mf_xcd_max_n = 300000;   // PGX_MF_XCD_MAXN: graphs beyond this stay on level launches (measured: maxflow.hip)
    int mf_sweeps = 0;           // PGX_MF_SWEEPS: sweeps per round, list mode and all-sites alike (0 = the measured defaults; tests shorten the rounds)
    int mf_region = 1;           // PGX_MF_REGION=0: no region moves (maxflow_tile.hip expand_alpha_region)
    int tile_sweeps = 24;        // push-relabel sweeps per discharge launch
    // a caller of a single whole-graph move (the inlier / outlier cut) may leave work here that expand_alpha_tile enqueues BEHIND the move and
    // BEFORE its one synchronisation (the compaction of the cut's flags and its copy back): one host round trip for both.  Consumed
    // (cleared) by the move; tile_pre_sync_ran says it ran and the move was solved (not handed back)
    std::function<int()> tile_pre_sync;
    bool tile_pre_sync_ran = false;
    int tile_mini = 1;           // PGX_TILE_MINI=0: graphs of <= 1024 sites and <= 8192 arcs go through t_move_kernel too (A/B; default: the LDS-resident t_mini_kernel)
    int64_t tile_launches[2] = {0, 0};   // pgx_one_workgroup_launches: whole-graph moves enqueued on t_mini_kernel / on t_move_kernel
    int tile_mini_sweeps = 24;   // PGX_TILE_MINI_SWEEPS: sweeps between two exact searches of t_mini_kernel
    int64_t paths[6] = {0, 0, 0, 0, 0, 0};   // pgx_expansion_paths
    int region_defer = 0;        // region moves are enqueued without a host round trip (pgx_expansion's batches): slot / skip rule below
    int region_slot = 0;
    int region_skip_rel = -1;
    int64_t tile_fallbacks = 0;  // moves the tile path handed back to maxflow.hip
    int tile_debug = 0;          // PGX_MF_DEBUG: one stderr line per global relabel
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    pgx::DevBuf scratch;  // misc small device scratch (bucket, energy, ...)
    pgx::DevBuf prosac_tops;              // pgx_sampler_prosac_set: subset size per sample number (int32 x prosac_count)
    int prosac_count = 0;
    int64_t prosac_points_version = -1;   // the table belongs to these points
    void* h_rb = nullptr;       // pinned ring of d2h() / sync_deliver()
    size_t h_rb_used = 0;
    std::vector<pgx::StagedCopy> rb;
    void* h_res = nullptr;      // pinned host staging for result read-backs (pageable targets make the copies synchronous)
    size_t h_res_cap = 0;
    // Host mirror of the score triples: score_finish_kernel also writes (count, value, shared) in the batch's device order
    // straight into this pinned, device-mapped allocation (coalesced 512 B runs over PCIe), so pgx_score_fetch needs no
    // copy command on the stream - it waits for the kernel and un-permutes on the host (h_perm; empty = identity).
    void* h_models = nullptr;    // pinned staging of pgx_score_upload's (reordered) batch + permutation
    size_t h_models_cap = 0;
    hipEvent_t ev_models = nullptr;
    int h_models_busy = 0;
    void* h_samples = nullptr;   // pinned staging of pgx_solve_minimal's sample indices (solve.hip upload_samples)
    size_t h_samples_cap = 0;
    hipEvent_t ev_samples = nullptr;
    int h_samples_busy = 0;
    void* h_mirror = nullptr;
    size_t h_mirror_cap = 0;
    int mirror_valid = 0;        // the last launch wrote the mirror
    int score_mirror = 1;        // PGX_SCORE_MIRROR=0: read the triples back with a copy instead
    std::vector<int> h_perm;     // host copy of `perm` for an uploaded, locality-sorted batch
    pgx::DevBuf fit_scratch;  // pgx_gram: partials | result | counters | index list
    pgx::DevBuf weights;      // resident per-point weights of the weighted refits (pgx_set_weights), weights_n == n when valid
    int64_t weights_n = 0;

    pgx::CommState* comm = nullptr;
};

namespace pgx {

int fail(pgx_ctx* ctx, int code, const char* fmt, ...);
int ensure(pgx_ctx* ctx, DevBuf& b, size_t bytes);
// Small read-backs go through pinned memory: a copy into a pageable target is a BLOCKING command (14 us against 4 for one into pinned
// memory, scripts/micro/small_copy_bench.hip), and an entry point that returns three small arrays paid it three times.  d2h() enqueues the
// copy into a slice of a pinned ring and remembers where the bytes belong; sync_deliver() synchronises the stream once and hands them
// over.  (Copies beyond 32 KB, or when the ring is full, go straight to the target as before.)  Pairs live inside one function body;
// every entry point starts with an empty list (CTX_GUARD).
int d2h(pgx_ctx* ctx, void* dst, const void* src, size_t bytes);
int sync_deliver(pgx_ctx* ctx);
int host_staging(pgx_ctx* ctx, size_t bytes, void** p);   // ctx->h_res grown to at least `bytes` (pinned: a copy into it is one asynchronous command; a pageable target makes every copy a blocking one)
void release(DevBuf& b);

#define PGX_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return pgx::fail(ctx, PGX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                 \
    } while (0)

#define PGX_TRY(call)               \
    do {                            \
        int r_ = (call);            \
        if (r_ != PGX_OK) return r_; \
    } while (0)

inline int64_t quantize(double x) { return (int64_t)__builtin_nearbyint(x * 4294967296.0); }
// weight of one directed neighbour entry, forced even so that w/2 is exact in the expansion graph
inline int64_t quantize_lambda(double lambda) { return 2 * (int64_t)__builtin_nearbyint(lambda * 2147483648.0); }

// group bounds of the score path (score.hip consumes them, setpoints.hip / score_sort_points build them)
constexpr double kGroupInflate = 1.00001;
constexpr int kGroupRow = 12;  // floats per group: c[3], rho, ub, vb, ru, rv, scale, pad[3]
constexpr int kSuper = 8;      // groups per super-group (512 points): first level of the cull kernel

// launchers implemented in the .hip translation units
int set_points_device(pgx_ctx* ctx, int model_type, const double* points, int64_t n);  // setpoints.hip: upload + all preprocessing
int score_launch(pgx_ctx* ctx, double T2, int has_compound, int want_masks);
int score_inliers_launch(pgx_ctx* ctx, int row, int32_t* index, int64_t* count);   // pointwise.hip
// point-sharded exchange (comm.hip): the last launch's integer accumulators, replicas summed, in the caller's hypothesis order
// ([3][Mpad] words) / counts | values | shared from such a block
int score_acc_export(pgx_ctx* ctx, unsigned long long* out, hipStream_t stream);
int score_acc_import(pgx_ctx* ctx, const unsigned long long* in, int M, int Mpad, double qscale, long long* counts, double* values,
                     double* shared, hipStream_t stream);
int score_sort_points(pgx_ctx* ctx, const double* points, const float* p32, const double* pmax);  // builds the sorted copies
int preference_launch(pgx_ctx* ctx, const double* model, double T2, double* d_pref, double out3[3]);
int compound_launch(pgx_ctx* ctx, const int32_t* slots, int K);
int unary_launch(pgx_ctx* ctx, int K, double threshold, double lambda);
int residual_sum_launch(pgx_ctx* ctx, const double* model, int label, double* sum);
int bucket_launch(pgx_ctx* ctx, int L, int64_t* counts, int32_t* order);
int energy_launch(pgx_ctx* ctx, int64_t lambda_q, int64_t h_q, int64_t* energy_q);
int greedy_labeling_launch(pgx_ctx* ctx, int64_t h_q, int64_t* energy_q, int* opened);
int epipolar_support_launch(pgx_ctx* ctx, const double* F, double T2, double S2, int64_t counts[2]);
int gc_labeling_launch(pgx_ctx* ctx, const double* model, double T2, double lambda, int32_t* flags, int64_t* count, bool want_index = false);
int graph_build_reverse(pgx_ctx* ctx);
int graph_build_launch(pgx_ctx* ctx, const double* pts, int64_t n, int d, int kind, double radius, int k, int64_t* arcs);
int graph_fetch_launch(pgx_ctx* ctx, int32_t* off, int32_t* idx, int32_t* mult);
int solve_minimal_launch(pgx_ctx* ctx, const int32_t* samples, int S, double* models_out, bool resident = false);   // resident: the samples are in ctx->scratch already
int solve_minimal_sampled_launch(pgx_ctx* ctx, int sampler, uint64_t key, uint32_t batch, int S, int32_t* samples_out, double* models_out);
int sampler_prosac_set(pgx_ctx* ctx, const int32_t* tops, int count);
int gram_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, int sel, const int32_t* index, int64_t m,
                int label, int use_weights, int wpow, double* out, int64_t* count, int64_t* bad);
int gram_labels_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, int K, int use_weights, int wpow,
                       double* out, int64_t* count, int64_t* bad);
int residual_sums_launch(pgx_ctx* ctx, const double* models, int K, double* sums);
int gram_batch_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, const int32_t* index, int B, int m,
                      const double* wsel, int wpow, double* out, int32_t* bad);
int pnp_refine_batch_launch(pgx_ctx* ctx, const double* inits, const int32_t* index, int B, int m, const double* wsel, int wpow,
                            int iterations, double* out, int32_t* status);
int expand_alpha_launch(pgx_ctx* ctx, int64_t lambda_q, int64_t h_q, int alpha, int64_t* changed);
int expand_cycle_l0(pgx_ctx* ctx, int64_t h_q, int64_t* changed, int* evaluated);  // lambda = 0: all labels, one read-back
int expand_alpha_on(pgx_ctx* ctx, int64_t n, int L, const long long* dq, int* labels, const long long* wq, int64_t lambda_q,
                    int64_t h_q, int alpha, int64_t* changed, bool source_reach = false);
void maxflow_free(pgx_ctx* ctx);
int maxflow_schedule_stats(pgx_ctx* ctx, int64_t out[8]);   // maxflow.hip
int eigh_smallest_launch(pgx_ctx* ctx, const double* A, int q, int64_t B, double* vec, double* val);   // fit.hip
constexpr int PGX_TILE_FALLBACK = 1000;   // expand_alpha_tile: not handled, run the level-synchronous path (labels untouched)
int expand_alpha_tile(pgx_ctx* ctx, int64_t n, int L, const long long* dq, int* labels, int64_t lambda_q, int64_t h_q, int alpha,
                      int64_t* changed, const long long* wq = nullptr);
void tile_free(pgx_ctx* ctx);
struct MfView;
constexpr int PGX_REGION_PENDING = 1001;  // expand_alpha_region with ctx->region_defer: enqueued, result by region_result after a synchronisation
int region_batch_begin(pgx_ctx* ctx);
int region_batch_fetch(pgx_ctx* ctx, int slots);   // the batch's results to the host mirror (one copy), before the synchronisation
int region_result(pgx_ctx* ctx, int slot, int alpha, int* status, int64_t* changed);
bool region_moves_apply(const pgx_ctx* ctx);   // maxflow.hip: pgx_expansion's moves on the resident problem go through expand_alpha_region
int expand_alpha_region(pgx_ctx* ctx, const MfView& mv, int64_t* changed);   // maxflow_tile.hip: a move with few open sites, one workgroup
void comm_free(pgx_ctx* ctx);

}  // namespace pgx
