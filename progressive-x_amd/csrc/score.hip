// score.hip — batched MSAC-with-compound-model scoring of M hypotheses over N points (gfx950).
//
// Replaces: progx::MSACScoringFunctionWithCompoundModel::getScore,
//           /root/reference/src/pyprogressivex/include/scoring_function_with_compound_model.h:61-125,
//           called once per hypothesis by the GC-RANSAC proposal loop (progressive_x.h:294).
//
// Mapping: ONE HYPOTHESIS PER LANE, points streamed through the scalar unit.
//   * a lane keeps its hypothesis (3..18 doubles) in VGPRs for the whole kernel;
//   * the point row (2..5 doubles) is wave-uniform, so it is fetched with s_load into SGPRs and used as the
//     scalar operand of v_mul_f64/v_add_f64 — no LDS, no cross-lane traffic, no per-pair reduction;
//   * inlier count / truncated-quadratic score / shared support accumulate in registers of the owning lane;
//   * grid = (ceil(M/256) hypothesis groups) x (point chunks); every block writes one partial triple per
//     hypothesis, a second tiny kernel adds the chunk partials in a fixed order (bit-reproducible run to run,
//     which keeps multi-GPU replicas identical without communication).
// The kernel is FP64-VALU bound (≈0.02 algorithmic bytes per pair); MFMA is deliberately unused: there is no
// dense contraction, every pair is a projective map + divide + compare.
#include <cmath>
#include <cstdlib>

#include "pgx_internal.h"
#include "score_filters.hip.h"

namespace pgx {

constexpr int kScoreBlock = 256;

// FILT: 0 = no filter, 1 = FP64 filter, 2 = FP32 pre-filter
template <int MT, bool MASK, int FILT>
__global__ __launch_bounds__(kScoreBlock) void score_kernel(
    const double* __restrict__ pts, int64_t n, const double* __restrict__ models, int M, int Mpad,
    double T2, const double* __restrict__ comp, int has_comp, int64_t chunk,
    const double* __restrict__ pmax, double guard, const float* __restrict__ pts32, double guard32,
    unsigned* __restrict__ pcnt, double* __restrict__ pval, double* __restrict__ psh,
    unsigned long long* __restrict__ masks, int64_t words, const int* __restrict__ perm, int chunks, int xcd_map,
    const float* __restrict__ gbounds)
{
    using R = Residual<MT>;
    // (hypothesis group, point chunk) of this block.  Workgroups are handed to the 8 XCDs round-robin by linear id and
    // every XCD has its own L2: with the plain 2-D grid (group fastest, 8 groups) XCD x ran group x over ALL chunks,
    // i.e. every XCD streamed the whole point set from HBM (PMC: ~1.3 GB fetched per launch for ~0.09 GB of inputs).
    // The 1-D mapping below gives each XCD its own chunks and runs the groups of one chunk back to back on it.
    int gx = (int)blockIdx.x, gy = (int)blockIdx.y;
    if (xcd_map) {
        const int groups = Mpad / kScoreBlock;
        const int slot = (int)(blockIdx.x >> 3);
        gx = slot % groups;
        gy = (int)(blockIdx.x & 7u) + 8 * (slot / groups);
        if (gy >= chunks) return;
    }
    const int m = gx * kScoreBlock + threadIdx.x;
    const bool live = m < M;
    const int64_t i0 = (int64_t)gy * chunk;
    const int64_t i1 = (i0 + chunk < n) ? (i0 + chunk) : n;

    double mdl[R::P];
#pragma unroll
    for (int k = 0; k < R::P; ++k)
        mdl[k] = live ? models[(int64_t)m * R::P + k] : __builtin_nan("");  // NaN model: never an inlier

    using F = Filter<MT>;
    using F32 = Filter32<MT>;
    const typename F::Lane flane = F::prep(mdl, guard);
    const typename F32::Lane flane32 = F32::prep(mdl, guard32, T2);
    const double T2d = T2 * (1.0 + kFilterDelta);
    const float T2d32 = f32_up(T2 * (1.0 + kFilter32Delta));
    const float Tup32 = f32_up(sqrt(T2) * (1.0 + 1.0 / 64.0));  // group test

    unsigned cnt = 0;
    double val = 0.0, sh = 0.0;
    unsigned long long word = 0;

    // One point per step; the body is written once and instantiated for a group of kUnroll points whose scalar loads
    // are all issued before the first use, so one s_waitcnt covers kUnroll points (SMEM returns out of order: the
    // only usable wait is lgkmcnt(0), which makes per-point prefetching impossible).
    auto step = [&](int64_t i, const double (&pt)[R::D], double pm, const float* p32) {
        bool inl = false;
        bool rejected = false;
        if (FILT == 1 && F::enabled) rejected = F::reject(pt, mdl, flane, pm, T2d);
        if (FILT == 2 && F32::enabled) rejected = F32::reject(p32, flane32, T2d32);
        if (live && !rejected) {  // exact path: oracle operation order, no contraction
            const double sq = R::squared(pt, mdl);
            inl = sq < T2;  // strict, scoring_function_with_compound_model.h:85
            if (inl) {
                ++cnt;                                        // :91
                const double s = cv_max(0.0, 1.0 - sq / T2);  // :94
                val += s;                                     // :97
                if (has_comp) sh += cv_min(comp[i], s);       // :115-117 (pref = 0 for non-inliers adds +0)
            }
        }
        if (MASK) {
            word |= (unsigned long long)(inl ? 1 : 0) << (i & 63);
            if ((i & 63) == 63 || i == i1 - 1) {
                if (live) masks[(int64_t)perm[m] * words + (i >> 6)] = word;  // :88 inlier list as a bit mask (row = caller's index)
                word = 0;
            }
        }
    };
    constexpr int kUnroll = 4;
    int64_t i = i0;
    while (i < i1) {
        // one 64-point group at a time (chunks start at multiples of 64)
        const int64_t gend = ((i | 63) + 1 < i1) ? (i | 63) + 1 : i1;
        if (FILT == 2 && F32::enabled && gbounds != nullptr) {
            const float* __restrict__ g = gbounds + (i >> 6) * kGroupRow;  // wave-uniform -> scalar loads
            float gr[kGroupRow];
#pragma unroll
            for (int k = 0; k < 9; ++k) gr[k] = g[k];
            const bool keep = live && !F32::group_reject(gr, flane32, Tup32);
            if (__ballot(keep) == 0) {  // none of this wave's hypotheses can have an inlier in the group
                if (MASK && live) masks[(int64_t)perm[m] * words + (i >> 6)] = 0;
                i = gend;
                continue;
            }
        }
        for (; i + kUnroll <= gend; i += kUnroll) {
            const double* __restrict__ prow = pts + i * R::D;  // wave-uniform address -> scalar loads
            double pt[kUnroll][R::D];
            double pm[kUnroll];
            float p32[kUnroll][8];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
                for (int k = 0; k < R::D; ++k) pt[u][k] = prow[u * R::D + k];
                pm[u] = (FILT == 1 && F::enabled) ? pmax[i + u] : 1.0;
                if (FILT == 2 && F32::enabled) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) p32[u][k] = pts32[(i + u) * 8 + k];
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) step(i + u, pt[u], pm[u], p32[u]);
        }
        for (; i < gend; ++i) {
            const double* __restrict__ prow = pts + i * R::D;
            double pt[R::D];
            float p32[8];
#pragma unroll
            for (int k = 0; k < R::D; ++k) pt[k] = prow[k];
            if (FILT == 2 && F32::enabled) {
#pragma unroll
                for (int k = 0; k < 8; ++k) p32[k] = pts32[i * 8 + k];
            }
            step(i, pt, (FILT == 1 && F::enabled) ? pmax[i] : 1.0, p32);
        }
    }
    const int64_t o = (int64_t)gy * Mpad + m;
    pcnt[o] = cnt;
    pval[o] = val;
    psh[o] = sh;
}

// ---- cull, then score group-major ------------------------------------------------------------------
// Skipping rejected groups inside the chunked kernel needs a whole wave of 64 hypotheses to agree (74 % of the (wave,
// group) pairs) although 93 % of the (hypothesis, group) pairs are rejected, and leaves the surviving work badly
// distributed (instructions 39 %, time 65 %); scoring the survivors hypothesis-major still runs the pre-filter for 64
// hypotheses at a time and the FP64 exact path with ~3 of 64 lanes busy (the union of the wave's candidates).  So the
// surviving pairs are scored the other way round:
//   score_cull_kernel   one wave per (64 hypotheses, segment of groups): the group test; bit h of keep[g][w] = hypothesis
//                       64 w + h may have inliers in group g.  Also stores every hypothesis' f32 filter constants.
//   score_group_kernel  one wave per GROUP, one point per lane (rows loaded once, coalesced); the surviving hypotheses
//                       stream through the scalar unit: 17-op pre-filter for 64 points, then the exact FP64 path with
//                       the group's actual candidates as active lanes; per (hypothesis, group): count = popcount, the
//                       two sums by a fixed shuffle tree, added to the hypothesis' accumulators as integers (count) and
//                       as 2^-q fixed point (sums) with integer atomics: exact and order-free, so results are
//                       bit-reproducible although the accumulation order is not fixed.  q = 62 - ceil(log2 n): the
//                       quantisation error of a sum is < (#groups with inliers) * 2^-(q+1), ~1e-13 relative.
//   score_finish_kernel accumulators -> counts / values / shared in the caller's hypothesis order.
constexpr int kHypRow = 20;    // floats per hypothesis: Filter32<MT>::Lane, padded

template <class LaneT>
__device__ __forceinline__ void lane_store(const LaneT& ln, float* __restrict__ row)
{
    static_assert(sizeof(LaneT) % 4 == 0 && sizeof(LaneT) / 4 <= kHypRow, "Filter32 lane constants must fit a row");
    const float* src = reinterpret_cast<const float*>(&ln);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(LaneT) / 4); ++k) row[k] = src[k];
}
template <class LaneT>
__device__ __forceinline__ LaneT lane_load(const float* __restrict__ row)
{
    LaneT ln;
    float* dst = reinterpret_cast<float*>(&ln);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(LaneT) / 4); ++k) dst[k] = row[k];
    return ln;
}

constexpr int kCullWaves = 4;   // hypothesis words (waves) per workgroup of the cull kernel
constexpr int kCullTile = 64;   // group bounds staged in LDS per step

// The group bounds of a segment are staged in LDS by the whole workgroup (coalesced vector loads, one round trip per 64
// groups) and read back as broadcasts: with scalar loads a wave paid one ~1 us scalar-cache miss per four groups, and the
// kernel — a single generation of waves — took as long as that latency chain (43 us for 27 VALU instructions per test).
template <int MT>
__global__ __launch_bounds__(64 * kCullWaves) void score_cull_kernel(
    const double* __restrict__ models, int M, double T2, double guard32, const float* __restrict__ gbounds, int groups,
    int gps /* groups per segment */, int W, unsigned long long* __restrict__ keep, float* __restrict__ hyp32,
    double* __restrict__ models_t /* [P][W * 64]: component-major copy for the gathers of the group kernel */,
    unsigned long long* __restrict__ zero /* the accumulators of the group kernel, zeroed here (one fill command less) */, int64_t zero_words)
{
    using R = Residual<MT>;
    using F32 = Filter32<MT>;
    __shared__ __attribute__((aligned(16))) float s_gb[kCullTile + kCullTile / kSuper][kGroupRow];  // groups | their super-groups
    {
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * (64 * kCullWaves);
        for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (64 * kCullWaves) + threadIdx.x; i < zero_words; i += nthreads) zero[i] = 0ull;
    }
    const int lane = (int)(threadIdx.x & 63);
    const int w = (int)blockIdx.x * kCullWaves + (int)(threadIdx.x >> 6), seg = (int)blockIdx.y;
    const int m = w * 64 + lane;
    const bool live = w < W && m < M;
    double mdl[R::P];
#pragma unroll
    for (int k = 0; k < R::P; ++k) mdl[k] = live ? models[(int64_t)m * R::P + k] : __builtin_nan("");
    const typename F32::Lane flane32 = F32::prep(mdl, guard32, T2);
    if (seg == 0 && w < W) {
        lane_store(flane32, hyp32 + (int64_t)m * kHypRow);
#pragma unroll
        for (int k = 0; k < R::P; ++k) models_t[(int64_t)k * W * 64 + m] = mdl[k];
    }
    const float Tup32 = f32_up(sqrt(T2) * (1.0 + 1.0 / 64.0));
    const int g0 = seg * gps, g1 = g0 + gps < groups ? g0 + gps : groups;
    for (int t0 = g0; t0 < g1; t0 += kCullTile) {
        const int cnt = g1 - t0 < kCullTile ? g1 - t0 : kCullTile;
        __syncthreads();  // the previous tile has been read
        const int scnt = (cnt + kSuper - 1) / kSuper;  // t0 is a multiple of kSuper (gps is)
        for (int e = (int)threadIdx.x; e < cnt * kGroupRow; e += 64 * kCullWaves)
            (&s_gb[0][0])[e] = gbounds[(int64_t)t0 * kGroupRow + e];
        for (int e = (int)threadIdx.x; e < scnt * kGroupRow; e += 64 * kCullWaves)
            (&s_gb[kCullTile][0])[e] = gbounds[((int64_t)groups + t0 / kSuper) * kGroupRow + e];
        __syncthreads();
        if (w >= W) continue;
        for (int sgi = 0; sgi < scnt; ++sgi) {
            const int i0 = sgi * kSuper, i1 = i0 + kSuper < cnt ? i0 + kSuper : cnt;
            float sr[kGroupRow];
#pragma unroll
            for (int k = 0; k < F32::kGroupVals; ++k) sr[k] = s_gb[kCullTile + sgi][k];  // same address in every lane: LDS broadcast
            if (__ballot(live && !F32::group_reject(sr, flane32, Tup32)) == 0) {  // no hypothesis of the word reaches these 512 points
                if (lane < i1 - i0) keep[(int64_t)(t0 + i0 + lane) * W + w] = 0;
                continue;
            }
            for (int i = i0; i < i1; ++i) {
                float gr[kGroupRow];
#pragma unroll
                for (int k = 0; k < F32::kGroupVals; ++k) gr[k] = s_gb[i][k];
                const unsigned long long bm = __ballot(live && !F32::group_reject(gr, flane32, Tup32));
                if (lane == 0) keep[(int64_t)(t0 + i) * W + w] = bm;
            }
        }
    }
}

// round-to-nearest-even of 0 <= x < 2^51 as an integer: one FP64 add against 1.5 * 2^52 and an integer subtraction (the
// generic double -> int64 conversion is ~20 instructions and ran twice per evaluated pair)
__device__ __forceinline__ long long to_fixed(double x)
{
    const double magic = 6755399441055744.0;
    return __double_as_longlong(x + magic) - __double_as_longlong(magic);
}

constexpr int kGroupWaves = 1;  // waves per workgroup of the group-major kernel

template <int MT, bool MASK, bool STATS = false>
__global__ __launch_bounds__(64 * kGroupWaves) void score_group_kernel(
    const double* __restrict__ pts, const float* __restrict__ pts32, const double* __restrict__ comp, int64_t n, int groups,
    const double* __restrict__ models, int W, double T2, int has_comp, const unsigned long long* __restrict__ keep,
    const float* __restrict__ hyp32, double qscale, unsigned long long* __restrict__ acc /* [3][Mpad]: count, value, shared */,
    int Mpad, unsigned long long* __restrict__ masks, int64_t words, const int* __restrict__ perm, int split, int xcd_local, const double* __restrict__ models_t,
    unsigned long long* __restrict__ stats /* STATS: [0] surviving (hypothesis, group) steps, [1] exact evaluations, [2] inlier pairs */,
    const double* __restrict__ pts_g /* [groups][D][64]: group-blocked SoA copy of the rows (nullptr: AoS) */,
    const float* __restrict__ p32_g /* [groups][8][64] */, int nrep, int dense_min)
{
    // split: waves per group, each takes every split-th word of 64 hypotheses (shorter waves: better tail)
    using R = Residual<MT>;
    using F32 = Filter32<MT>;
    using LaneT = typename F32::Lane;
    const int lane = (int)(threadIdx.x & 63);
    // One item (group, part) per wave, one wave per workgroup.  Measured alternatives at M = 2048, N = 1e6: 4-wave workgroups
    // (a workgroup retires with its slowest wave; 0.32 ms), 2-wave (0.30), persistent waves striding over the items
    // (0.31-0.38), workgroups that hand out 8-64 items to their 4 waves from an LDS counter (0.29-0.32), placing a
    // group's workgroups on one XCD (FETCH_SIZE 165 -> 45 MiB but 0.51 ms); one wave per workgroup with 8 parts per
    // group: 0.28 ms (16 parts: 250 k workgroups, the dispatcher limits at ~1.3 ns per workgroup).
    const int wv = 0;
    int g, part;
    if (xcd_local) {  // workgroup ids go round-robin over the 8 XCDs: all parts of a group on one XCD (its L2 fetches the rows once)
        const int slot = (int)(blockIdx.x >> 3);
        g = (slot / split) * 8 + (int)(blockIdx.x & 7u);
        part = slot % split;
        if (g >= groups) return;
    } else {
        g = (int)blockIdx.x / split;
        part = (int)blockIdx.x % split;
    }
    g = __builtin_amdgcn_readfirstlane(g);
    part = __builtin_amdgcn_readfirstlane(part);
    {   // nothing survived the cull for this wave's hypotheses: leave before the point rows are requested
        unsigned long long any = 0;
        for (int w = part; w < W; w += split) any |= keep[(int64_t)g * W + w];
        if (any == 0) return;
    }
    const int64_t j = (int64_t)g * 64 + lane;
    const bool valid = j < n;
    const int64_t jj = valid ? j : n - 1;
    double pt[R::D];
    float p32[8];
    if (pts_g != nullptr) {
        // group-blocked SoA copies: every load instruction of the wave reads one contiguous 512 B (256 B) run instead of 64
        // rows 40 B (32 B) apart - 13 loads touch 17 cache lines instead of ~100 (the tail group is padded with its last row)
#pragma unroll
        for (int q = 0; q < R::D; ++q) pt[q] = pts_g[((int64_t)g * R::D + q) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) p32[q] = 0.0f;
        if constexpr (RowFromPoint<MT>::in1 >= RowFromPoint<MT>::in0) {
            // the generic f32 row (sp_prep_kernel) recomputed from the f64 row just loaded - the same operations, the same bits -
            // instead of a second 24-byte row per point from memory (24 MB of 89 fetched per launch at 10^6 points)
#pragma unroll
            for (int q = 0; q < R::D; ++q) p32[q] = (float)pt[q];
            double pm = 1.0;
#pragma unroll
            for (int q = RowFromPoint<MT>::in0; q <= RowFromPoint<MT>::in1; ++q) { const double a = fabs(pt[q]); if (!(a <= pm)) pm = a; }
            p32[5] = (float)(pm * 1.000001);
        } else {
#pragma unroll
            for (int q = 0; q < F32::kRowVals; ++q) p32[q] = p32_g[((int64_t)g * 8 + q) * 64 + lane];
        }
    } else {
#pragma unroll
        for (int q = 0; q < R::D; ++q) pt[q] = pts[jj * R::D + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) p32[q] = pts32[jj * 8 + q];
    }
    const double cmp = has_comp ? comp[jj] : 0.0;
    // nrep replicas of the integer accumulators, chosen by workgroup id: workgroup ids go round-robin over the XCDs, so with
    // nrep a multiple of 8 a replica is only ever updated from one XCD (its atomics stay in that L2), and the hot
    // hypotheses (thousands of updates) do not serialise on three addresses.  score_finish_kernel adds the replicas (exact).
    acc += (size_t)(blockIdx.x % (unsigned)nrep) * 3 * (size_t)Mpad;
    const float T2d32 = f32_up(T2 * (1.0 + kFilter32Delta));
    // The f32 constants of a word's surviving hypotheses are staged in LDS by the lanes that own them (one vector-load
    // round trip per 64 hypotheses) and read back as broadcasts: one scalar-memory round trip per hypothesis (~1 us
    // when the scalar cache misses) left the kernel latency-bound (37 % VALU utilisation).  The f64 model is fetched
    // only by pairs that have candidates (staging it as well costs occupancy or compaction work: measured slower).
    __shared__ float s_h32[kGroupWaves][64][kHypRow];
    __shared__ unsigned s_queue[kGroupWaves][128];
    int qn = 0;  // queued candidate pairs of this wave (wave-uniform)
    unsigned long long st_steps = 0, st_exact = 0, st_inl = 0;  // STATS only (wave-uniform / per-lane partials)
    // Exact evaluation of up to 64 queued (hypothesis, point) pairs, one per lane.  A pair's point lives in the registers
    // of lane `src` of this wave (shuffles), its model is gathered from global memory.  Every contribution is converted to
    // 2^-q fixed point BEFORE any summation, so the accumulated integers do not depend on how pairs were batched:
    // results are bit-reproducible and independent of the launch geometry.  Equal hypotheses are adjacent in the queue
    // (pairs are appended hypothesis by hypothesis): a segmented shuffle reduction leaves one atomic set per run.
    auto drain = [&](int c) __attribute__((always_inline)) {
        const bool act = lane < c;
        const unsigned e = act ? s_queue[wv][lane] : 0u;
        const int m = act ? (int)(e >> 6) : -1 - lane;
        const int src = act ? (int)(e & 63u) : lane;
        double q_pt[R::D];
#pragma unroll
        for (int k = 0; k < R::D; ++k) q_pt[k] = __shfl(pt[k], src, 64);
        const double q_cmp = has_comp ? __shfl(cmp, src, 64) : 0.0;
        long long cnt = 0, val = 0, shq = 0;
        if (act) {  // exact path: oracle operation order, no contraction
            double mdl[R::P];
#pragma unroll
            for (int k = 0; k < R::P; ++k) mdl[k] = models_t[(int64_t)k * Mpad + m];  // neighbours in m share cache lines
            const double sq = R::squared(q_pt, mdl);
            if (STATS) ++st_exact;
            if (sq < T2) {  // strict, scoring_function_with_compound_model.h:85
                const double sc = cv_max(0.0, 1.0 - sq / T2);                       // :94
                cnt = 1;
                if (STATS) ++st_inl;
                val = to_fixed(sc * qscale);
                if (has_comp) shq = to_fixed(cv_min(q_cmp, sc) * qscale);           // :115-117
            }
        }
        for (int off = 1; off < 64; off <<= 1) {  // segmented sums: lane i ends with the total of i .. end of its run
            const int mo = __shfl_down(m, off, 64);
            const bool same = lane + off < 64 && mo == m;
            if (__ballot(same) == 0) break;  // runs are contiguous: none reaches `off` lanes, none reaches further
            const long long c2 = __shfl_down(cnt, off, 64), v2 = __shfl_down(val, off, 64), s2 = __shfl_down(shq, off, 64);
            if (same) { cnt += c2; val += v2; shq += s2; }
        }
        const int mp = __shfl_up(m, 1, 64);
        if (act && (lane == 0 || mp != m) && cnt > 0) {
            atomicAdd(&acc[m], (unsigned long long)cnt);
            atomicAdd(&acc[(int64_t)Mpad + m], (unsigned long long)val);
            if (has_comp) atomicAdd(&acc[2 * (int64_t)Mpad + m], (unsigned long long)shq);
        }
    };
    // Exact evaluation IN PLACE for one hypothesis: the candidates of this group are the active lanes (point in registers,
    // the f64 model at a wave-uniform address), per-lane fixed point, one integer shuffle tree, one set of atomics.  Used by
    // the mask-producing variant for every step and by the queued variant for DENSE steps (>= dense_min candidates of 64):
    // through the queue such a step would pay the shuffles, the model gather and the segmented reduction per pair for
    // nothing - its 64 pairs already sit in 64 lanes.  The same per-pair integers as the queued path: bitwise equal sums.
    auto direct = [&](int m, bool cand) __attribute__((always_inline)) {
        double sc = 0.0, shv = 0.0;
        bool inl = false;
        if (cand) {  // exact path: oracle operation order, no contraction
            double mdl[R::P];
#pragma unroll
            for (int k = 0; k < R::P; ++k) mdl[k] = models[(int64_t)m * R::P + k];
            const double sq = R::squared(pt, mdl);
            inl = sq < T2;  // strict, scoring_function_with_compound_model.h:85
            if (STATS) { ++st_exact; if (inl) ++st_inl; }
            if (inl) {
                sc = cv_max(0.0, 1.0 - sq / T2);      // :94
                if (has_comp) shv = cv_min(cmp, sc);  // :115-117
            }
        }
        const unsigned long long bm = __ballot(inl);
        if (bm == 0) return;
        // per-lane fixed point first (the same integers the queued path adds up), then an exact integer tree
        long long val = inl ? to_fixed(sc * qscale) : 0, shq = (inl && has_comp) ? to_fixed(shv * qscale) : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            val += __shfl_down(val, off, 64);
            shq += __shfl_down(shq, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&acc[m], (unsigned long long)__popcll(bm));
            atomicAdd(&acc[(int64_t)Mpad + m], (unsigned long long)val);
            if (has_comp) atomicAdd(&acc[2 * (int64_t)Mpad + m], (unsigned long long)shq);
            if (MASK) masks[(int64_t)perm[m] * words + g] = bm;  // rows start zeroed
        }
    };
    for (int w = part; w < W; w += split) {
        unsigned long long todo = keep[(int64_t)g * W + w];  // wave-uniform -> scalar load
        if (todo == 0) continue;
        if (STATS) st_steps += (unsigned long long)__popcll(todo);
        __builtin_amdgcn_wave_barrier();  // the previous word's reads are done (LDS ops of a wave execute in order)
        if ((todo >> lane) & 1ull) {
            const int64_t ml = (int64_t)w * 64 + lane;
#pragma unroll
            for (int k = 0; k < kHypRow; ++k) s_h32[wv][lane][k] = hyp32[ml * kHypRow + k];
        }
        __builtin_amdgcn_wave_barrier();
        while (todo != 0) {
            const int h = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int m = w * 64 + h;
            const LaneT ln = lane_load<LaneT>(&s_h32[wv][h][0]);  // same address in every lane: LDS broadcast
            const bool cand = valid && !F32::reject(p32, ln, T2d32);
            const unsigned long long cm = __ballot(cand);
            if (cm == 0) continue;
            if (!MASK && __popcll(cm) < dense_min) {
                // the exact path with ~3 of 64 lanes busy per (hypothesis, group) was most of this kernel: candidates
                // are queued instead and evaluated 64 at a time
                if (cand) s_queue[wv][qn + __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u))] =
                    ((unsigned)m << 6) | (unsigned)lane;
                qn += __popcll(cm);
                if (qn >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    drain(64);
                    __builtin_amdgcn_wave_barrier();
                    const unsigned mv = s_queue[wv][64 + lane];
                    __builtin_amdgcn_wave_barrier();
                    s_queue[wv][lane] = mv;
                    qn -= 64;
                }
                continue;
            }
            direct(m, cand);
        }
    }
    if (!MASK && qn > 0) {
        __builtin_amdgcn_wave_barrier();
        drain(qn);
    }
    if (STATS) {  // one set of atomics per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            st_exact += __shfl_down(st_exact, off, 64);
            st_inl += __shfl_down(st_inl, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&stats[0], st_steps);
            atomicAdd(&stats[1], st_exact);
            atomicAdd(&stats[2], st_inl);
        }
    }
}

// ---- PGX_VERIFY=1: every decision of the bound / filter chain against the exact residual -----------------------------------
// The correctness of counts and masks rests on inequalities with hand-derived error budgets (score_filters.hip.h): a group bound
// that removes (hypothesis, 64-point group) pairs and an f32 filter that removes single pairs may only ever remove outliers.
// This kernel re-decides EVERY pair of the batch with the exact FP64 residual (the reference's operation order) and counts the
// pairs the chain discarded although they are inliers: out[0] removed by the group bound, out[1] by the f32 filter; out[2] =
// inlier pairs.  Any non-zero out[0] / out[1] is a hole in a proof.  Run inside pgx_score_stats only (never timed).
template <int MT>
__global__ __launch_bounds__(64) void score_verify_kernel(const double* __restrict__ pts, const float* __restrict__ pts32, int64_t n,
                                                          const double* __restrict__ models, int M, int W, double T2,
                                                          const unsigned long long* __restrict__ keep, const float* __restrict__ hyp32,
                                                          unsigned long long* __restrict__ out)
{
    using R = Residual<MT>;
    using F32 = Filter32<MT>;
    using LaneT = typename F32::Lane;
    const int lane = (int)threadIdx.x, g = (int)blockIdx.x, w = (int)blockIdx.y;
    const int64_t j = (int64_t)g * 64 + lane;
    const bool valid = j < n;
    const int64_t jj = valid ? j : n - 1;
    double pt[R::D];
    float p32[8];
#pragma unroll
    for (int q = 0; q < R::D; ++q) pt[q] = pts[jj * R::D + q];
#pragma unroll
    for (int q = 0; q < 8; ++q) p32[q] = pts32[jj * 8 + q];
    const float T2d32 = f32_up(T2 * (1.0 + kFilter32Delta));
    const unsigned long long todo = keep[(int64_t)g * W + w];
    unsigned long long c_cull = 0, c_filt = 0, c_inl = 0;
    for (int h = 0; h < 64; ++h) {
        const int m = w * 64 + h;
        if (m >= M) break;
        double mdl[R::P];
#pragma unroll
        for (int k = 0; k < R::P; ++k) mdl[k] = models[(int64_t)m * R::P + k];
        const bool inl = valid && R::squared(pt, mdl) < T2;
        const bool kept = (todo >> h) & 1ull;
        const LaneT ln = lane_load<LaneT>(hyp32 + (int64_t)m * kHypRow);
        const bool rej = F32::reject(p32, ln, T2d32);
        c_inl += inl;
        c_cull += inl && !kept;
        c_filt += inl && kept && rej;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c_cull += __shfl_down(c_cull, off, 64);
        c_filt += __shfl_down(c_filt, off, 64);
        c_inl += __shfl_down(c_inl, off, 64);
    }
    if (lane == 0) {
        if (c_cull) atomicAdd(&out[0], c_cull);
        if (c_filt) atomicAdd(&out[1], c_filt);
        if (c_inl) atomicAdd(&out[2], c_inl);
    }
}

__global__ __launch_bounds__(256) void score_finish_kernel(const unsigned long long* __restrict__ acc, int M, int Mpad, double qscale,
                                                           const int* __restrict__ perm, long long* __restrict__ counts,
                                                           double* __restrict__ values, double* __restrict__ shared, int nrep,
                                                           long long* __restrict__ mirror /* pinned host memory or nullptr */)
{
    const int m = (int)(blockIdx.x * 256 + threadIdx.x);
    if (m >= M) return;
    const int o = perm[m];  // hypotheses were scored in locality order: results go back to the caller's order
    // nrep per-XCD replicas of the integer accumulators (co-located groups): integer sums, exact in any order
    unsigned long long c = 0, v = 0, sh = 0;
    for (int r = 0; r < nrep; ++r) {
        const unsigned long long* a = acc + (size_t)r * 3 * (size_t)Mpad;
        c += a[m];
        v += a[(int64_t)Mpad + m];
        sh += a[2 * (int64_t)Mpad + m];
    }
    const double val = (double)(long long)v / qscale, shv = (double)(long long)sh / qscale;
    counts[o] = (long long)c;
    values[o] = val;
    shared[o] = shv;
    if (mirror != nullptr) {  // the same triples in device order, contiguous per wave: pgx_score_fetch reads them without a copy
        mirror[m] = (long long)c;
        reinterpret_cast<double*>(mirror)[(int64_t)Mpad + m] = val;
        reinterpret_cast<double*>(mirror)[2 * (int64_t)Mpad + m] = shv;
    }
}

// ---- point-sharded jobs (comm.hip pgx_score_allreduce): the integer accumulators leave / enter here ---------------------------
// Counts are integers and both sums are 2^-q fixed point, so the accumulators of ranks that scored DIFFERENT points against the
// same hypotheses add exactly in any order: ncclAllReduce(sum, uint64) of 3 x Mpad words gives bitwise the accumulators of one
// GPU scoring all the points (given one q: pgx_score_set_global_n).  Hypotheses are in locality order on the device and the
// order may differ between ranks (it depends on nothing but the models today, but the exchange must not rely on that): the
// block travels in the CALLER's order.
__global__ __launch_bounds__(256) void score_acc_export_kernel(const unsigned long long* __restrict__ acc, int M, int Mpad, int nrep,
                                                               const int* __restrict__ perm, unsigned long long* __restrict__ out)
{
    const int m = (int)(blockIdx.x * 256 + threadIdx.x);
    if (m >= Mpad) return;
    if (m >= M) {   // the padding is nobody's target: zero
        out[m] = 0; out[(size_t)Mpad + m] = 0; out[2 * (size_t)Mpad + m] = 0;
        return;
    }
    unsigned long long c = 0, v = 0, sh = 0;
    for (int r = 0; r < nrep; ++r) {
        const unsigned long long* a = acc + (size_t)r * 3 * (size_t)Mpad;
        c += a[m];
        v += a[(size_t)Mpad + m];
        sh += a[2 * (size_t)Mpad + m];
    }
    const int o = perm[m];
    out[o] = c; out[(size_t)Mpad + o] = v; out[2 * (size_t)Mpad + o] = sh;
}

__global__ __launch_bounds__(256) void score_acc_import_kernel(const unsigned long long* __restrict__ in, int M, int Mpad, double qscale,
                                                               long long* __restrict__ counts, double* __restrict__ values,
                                                               double* __restrict__ shared)
{
    const int m = (int)(blockIdx.x * 256 + threadIdx.x);
    if (m >= M) return;
    counts[m] = (long long)in[m];                                                // as score_finish_kernel
    values[m] = (double)(long long)in[(size_t)Mpad + m] / qscale;
    shared[m] = (double)(long long)in[2 * (size_t)Mpad + m] / qscale;
}

int score_acc_export(pgx_ctx* ctx, unsigned long long* out, hipStream_t stream)
{
    if (ctx->last_acc != nullptr && (ctx->last_acc_M != ctx->M || ctx->last_acc_Mpad != ctx->Mpad))
        return fail(ctx, PGX_ERR_INVALID, "integer accumulators are stale: the batch changed (upload / solve) since the last launch (%d of %d then, %d of %d now)",
                    ctx->last_acc_M, ctx->last_acc_Mpad, ctx->M, ctx->Mpad);
    if (ctx->last_acc == nullptr || ctx->last_score_path != 2)
        return fail(ctx, PGX_ERR_INVALID, "point-sharded exchange: the last launch did not run the group-major path (integer accumulators); "
                                          "it needs sorted points, the f32 filter and the cull (the defaults)");
    hipLaunchKernelGGL(score_acc_export_kernel, dim3((unsigned)((ctx->Mpad + 255) / 256)), dim3(256), 0, stream, ctx->last_acc, ctx->M, ctx->Mpad,
                       ctx->last_nrep, ctx->perm.as<int>(), out);
    PGX_HIP(ctx, hipGetLastError());
    return PGX_OK;
}

int score_acc_import(pgx_ctx* ctx, const unsigned long long* in, int M, int Mpad, double qscale, long long* counts, double* values,
                     double* shared, hipStream_t stream)
{
    hipLaunchKernelGGL(score_acc_import_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, in, M, Mpad, qscale, counts, values, shared);
    PGX_HIP(ctx, hipGetLastError());
    return PGX_OK;
}

// Adds the chunk partials of each hypothesis in a FIXED order (bit-reproducible): 64 hypotheses per block, 16 waves
// each summing the chunks k = wave, wave+16, ... sequentially, then the 16 wave sums are added in wave order.
constexpr int kReduceWaves = 16;
__global__ __launch_bounds__(64 * kReduceWaves) void score_reduce_kernel(
    const unsigned* __restrict__ pcnt, const double* __restrict__ pval, const double* __restrict__ psh,
    int chunks, int Mpad, int M, const int* __restrict__ perm, long long* __restrict__ counts,
    double* __restrict__ values, double* __restrict__ shared)
{
    __shared__ long long lc[kReduceWaves][64];
    __shared__ double lv[kReduceWaves][64], ls[kReduceWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    long long c = 0;
    double v = 0.0, s = 0.0;
    if (m < M)
        for (int k = wave; k < chunks; k += kReduceWaves) {
            const int64_t o = (int64_t)k * Mpad + m;
            c += pcnt[o];
            v += pval[o];
            s += psh[o];
        }
    lc[wave][lane] = c; lv[wave][lane] = v; ls[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && m < M) {
        for (int w = 1; w < kReduceWaves; ++w) { c += lc[w][lane]; v += lv[w][lane]; s += ls[w][lane]; }
        const int o = perm[m];  // hypotheses were scored in locality order: results go back to the caller's order
        counts[o] = c;
        values[o] = v;
        shared[o] = s;
    }
}

template <int MT, bool MASK, int FILT>
static void score_launch_one(pgx_ctx* ctx, double T2, int has_compound, double guard, double guard32 = 0.0)
{
    const unsigned groups = (unsigned)(ctx->Mpad / kScoreBlock);
    dim3 grid(groups, (unsigned)ctx->chunks);
    if (ctx->score_xcd_map) grid = dim3(groups * (((unsigned)ctx->chunks + 7u) / 8u * 8u), 1);
    const bool srt = ctx->point_sort != 0;  // spatially sorted copies (score_sort_points); masks come out in sorted bit order
    hipLaunchKernelGGL((score_kernel<MT, MASK, FILT>), grid, dim3(kScoreBlock), 0, ctx->stream,
                       (srt ? ctx->pts_s : ctx->pts).as<double>(), ctx->n, ctx->models.as<double>(), ctx->M, ctx->Mpad, T2,
                       (srt ? ctx->comp_s : ctx->comp).as<double>(), has_compound, ctx->chunk,
                       (srt ? ctx->pmax_s : ctx->pmax).as<double>(), guard, (srt ? ctx->pts32_s : ctx->pts32).as<float>(), guard32,
                       ctx->pcnt.as<unsigned>(), ctx->pval.as<double>(), ctx->psh.as<double>(),
                       MASK ? (srt ? ctx->masks_s : ctx->masks).as<unsigned long long>() : (unsigned long long*)nullptr,
                       ctx->words, ctx->perm.as<int>(), ctx->chunks, ctx->score_xcd_map,
                       (srt && FILT == 2) ? ctx->gbounds.as<float>() : (const float*)nullptr);
}

template <int MT>
static int score_dispatch(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    // filter guard (see Filter<> above): usable iff Umax / T <= 2^28 and everything is finite
    const double T = std::sqrt(T2);
    double guard = 0.0;
    bool filt = Filter<MT>::enabled && ctx->filter_enabled && T > 0.0 && std::isfinite(T) && std::isfinite(ctx->umax) &&
                ctx->umax <= T * 268435456.0;
    if (filt) {
        guard = 4.5 * 1.1102230246251565e-16 * (1.0 + ctx->umax + T) * 16777216.0 / T;
        filt = std::isfinite(guard);
    }
    // FP32 pre-filter: tau = 2^-10, needs Umax / T <= tau * 2^24 = 2^14
    double guard32 = 0.0;
    bool filt32 = filt && ctx->filter_enabled == 1 && ctx->umax <= T * 16384.0;
    if (filt32) {
        guard32 = 5.5 * 5.9604644775390625e-8 * (1.0 + ctx->umax + T) * 1024.0 / T;
        filt32 = std::isfinite(guard32) && guard32 < 1e30;
    }
    if constexpr (MT == kVanishingPoint)   // its own trust test per pair, no global guard (Filter32<kVanishingPoint>)
        filt32 = ctx->filter_enabled == 1 && T > 0.0 && std::isfinite(T) && T2 < 1e30;
    if constexpr (MT == kHomography || MT == kHomographySym)   // explicit per-pair error terms, no global guard (Filter32<kHomography>)
        filt32 = ctx->filter_enabled == 1 && T2 > 1e-24 && T2 < 1e24;
    if constexpr (MT == kLine2D) {         // per-pair error term, no global guard; T'' must be an ordinary f32
        filt32 = ctx->filter_enabled == 1 && T2 > 1e-24 && T2 < 1e24 && std::isfinite(ctx->fscale);
        guard32 = ctx->fscale;             // Filter32<kLine2D>::prep: overflow guard (fscale >= 1)
    }
    if constexpr (MT == kFundamental) {    // likewise; the bounds on T keep T2 * D~^2 (D~ >= 1e-12) inside the f32 normal range
        filt32 = ctx->filter_enabled == 1 && T2 > 1e-12 && T2 < 1e12 && std::isfinite(ctx->fscale);
        guard32 = ctx->fscale * ctx->fscale;   // Filter32<kFundamental>::prep: overflow guard of the f32 terms (fscale >= 1)
    }
    // every filter's proof takes the exact path's f64 arithmetic as overflow-free: coordinates up to 1e30 with the
    // per-hypothesis band of pow2_normaliser keep it so
    if (!(ctx->fscale <= 1e30)) filt = filt32 = false;
    ctx->last_score_filtered = filt32 ? 2 : (filt ? 1 : 0);
    if constexpr (Filter32<MT>::enabled) {
        if (filt32 && ctx->point_sort && ctx->score_cull) {
            // ---- cull, then score group-major
            const int groups = (int)((ctx->n + 63) / 64);
            const int kCullSegs = ctx->score_cull_segs;
            const int gps = ((groups + kCullSegs - 1) / kCullSegs + kSuper - 1) / kSuper * kSuper;  // whole super-groups per segment
            const int W = ctx->Mpad / 64;
            PGX_TRY(ensure(ctx, ctx->cull_lists, (size_t)groups * W * sizeof(unsigned long long)));             // keep[g][w]
            // Where the waves of a group run.  Part p of every group on XCD p spreads a group's work over the chip but makes every
            // XCD fetch every row; all parts of a group on one XCD fetches a row once (FETCH_SIZE 8x lower) and needs a replica of
            // the accumulators per XCD.  Measured on the final code: the co-located mapping is 9 % faster (group kernel 215 -> 196 us)
            // on a locality-ordered batch, where only a few of a group's hypothesis words have survivors, and 19 % slower (step
            // 0.42 -> 0.50 ms) on a batch in arbitrary order, where all of them do - so the order of the batch decides.
            const int group_xcd = ctx->score_group_xcd >= 0 ? ctx->score_group_xcd : (ctx->h_perm.empty() ? 0 : 1);
            const int nrep = ctx->score_nrep > 0 ? ctx->score_nrep : (group_xcd ? 8 : 1);
            const int xcd_local = group_xcd ? 1 : 0;   // per-XCD replicas of the accumulators when a group's waves share an XCD
            PGX_TRY(ensure(ctx, ctx->cull_counts, (size_t)ctx->Mpad * (kHypRow * sizeof(float) + (size_t)nrep * 3 * sizeof(long long) + Residual<MT>::P * sizeof(double))));  // hyp32 | acc[nrep] | models_t
            float* hyp32 = ctx->cull_counts.as<float>();
            unsigned long long* acc = (unsigned long long*)(ctx->cull_counts.as<char>() + (size_t)ctx->Mpad * kHypRow * sizeof(float));
            double* models_t = (double*)(acc + (size_t)nrep * 3 * (size_t)ctx->Mpad);
            const double* pts_g = ctx->pts_g.p ? ctx->pts_g.as<double>() : (const double*)nullptr;   // group-blocked SoA copies of the rows
            const float* p32_g = ctx->pts_g.p ? ctx->p32_g.as<float>() : (const float*)nullptr;
            int lg = 0;
            const int64_t n_scale = ctx->score_global_n > ctx->n ? ctx->score_global_n : ctx->n;   // pgx_score_set_global_n
            while (((int64_t)1 << lg) < n_scale + 1) ++lg;
            const double qscale = std::ldexp(1.0, 62 - lg < 50 ? 62 - lg : 50);  // every sum is <= n < 2^lg; terms < 2^51 (to_fixed)
            // the accumulators are zeroed by the cull kernel, which runs before their first use
            const int64_t zero_words = (int64_t)nrep * ctx->Mpad * 3;
            if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventRecord(ctx->kev[0], ctx->stream));
            hipLaunchKernelGGL((score_cull_kernel<MT>), dim3((unsigned)((W + kCullWaves - 1) / kCullWaves), kCullSegs), dim3(64 * kCullWaves), 0,
                               ctx->stream, ctx->models.as<double>(), ctx->M, T2, guard32, ctx->gbounds.as<float>(), groups, gps, W,
                               ctx->cull_lists.as<unsigned long long>(), hyp32, models_t, acc, zero_words);
            PGX_HIP(ctx, hipGetLastError());
            if (ctx->score_profile) PGX_HIP(ctx, hipEventRecord(ctx->kev[1], ctx->stream));
            // waves per group.  Spread mapping: 8 (part p = XCD p).  Co-located mapping: 5 - fewer, longer waves load a group's rows
            // less often, and an ODD count keeps the heavy workgroups (a locality-ordered batch puts a group's survivors into two or
            // three neighbouring hypothesis words) from falling into a period of the dispatch order: group kernel 169 (8), 157 (4),
            // 146 (6), 140 (2) against 135-140 us (1, 3, 5, 7) on the metric batch.
            // (pose problems take 5 with the spread mapping as well: RANSAC-like batch 0.371 -> 0.353 ms; Sampson and vanishing-point
            // batches lose 15-20 % there and keep 8)
            // Round 6 (scripts/sweep_vp_geometry.py, after the Hough ordering of the segments): vanishing-point and Sampson batches take 16 -
            // group kernel 301 -> 280 us and 123 -> 114 us against 8 (12: 284 / 118, 24: 284 / 115, 32: 296 / 119).
            const int split_cfg = ctx->score_split > 0 ? ctx->score_split
                                  : ((group_xcd || MT == kPnP) ? 5 : ((MT == kVanishingPoint || MT == kFundamental) ? 16 : 8));
            const int split = split_cfg < W ? split_cfg : W;
            const unsigned gblocks = (xcd_local & 1) ? (unsigned)((int64_t)((groups + 7) / 8) * 8 * split) : (unsigned)((int64_t)groups * split);
            if (want_masks) {
                PGX_HIP(ctx, hipMemsetAsync(ctx->masks_s.p, 0, (size_t)ctx->M * (size_t)ctx->words * sizeof(uint64_t), ctx->stream));
                hipLaunchKernelGGL((score_group_kernel<MT, true>), dim3(gblocks), dim3(64 * kGroupWaves), 0, ctx->stream,
                                   ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->comp_s.as<double>(), ctx->n, groups,
                                   ctx->models.as<double>(), W, T2, has_compound, ctx->cull_lists.as<unsigned long long>(), hyp32,
                                   qscale, acc, ctx->Mpad, ctx->masks_s.as<unsigned long long>(), ctx->words, ctx->perm.as<int>(), split, xcd_local, models_t,
                                   (unsigned long long*)nullptr, pts_g, p32_g, nrep, 65);
            } else if (ctx->score_stats) {  // pgx_score_stats: the same launch with work counters (never timed)
                PGX_TRY(ensure(ctx, ctx->stats_buf, 8 * sizeof(unsigned long long)));
                PGX_HIP(ctx, hipMemsetAsync(ctx->stats_buf.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
                hipLaunchKernelGGL((score_group_kernel<MT, false, true>), dim3(gblocks), dim3(64 * kGroupWaves), 0, ctx->stream,
                                   ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->comp_s.as<double>(), ctx->n, groups,
                                   ctx->models.as<double>(), W, T2, has_compound, ctx->cull_lists.as<unsigned long long>(), hyp32,
                                   qscale, acc, ctx->Mpad, (unsigned long long*)nullptr, ctx->words, ctx->perm.as<int>(), split, xcd_local, models_t,
                                   ctx->stats_buf.as<unsigned long long>(), pts_g, p32_g, nrep, ctx->score_dense_min);
                if (ctx->verify)   // PGX_VERIFY=1: stats_buf[4..6] = inliers the group bound removed / the f32 filter removed / all inlier pairs
                    hipLaunchKernelGGL((score_verify_kernel<MT>), dim3((unsigned)groups, (unsigned)W), dim3(64), 0, ctx->stream,
                                       ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->n, ctx->models.as<double>(), ctx->M, W, T2,
                                       ctx->cull_lists.as<unsigned long long>(), hyp32, ctx->stats_buf.as<unsigned long long>() + 4);
            } else {
                hipLaunchKernelGGL((score_group_kernel<MT, false>), dim3(gblocks), dim3(64 * kGroupWaves), 0, ctx->stream,
                                   ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->comp_s.as<double>(), ctx->n, groups,
                                   ctx->models.as<double>(), W, T2, has_compound, ctx->cull_lists.as<unsigned long long>(), hyp32,
                                   qscale, acc, ctx->Mpad, (unsigned long long*)nullptr, ctx->words, ctx->perm.as<int>(), split, xcd_local, models_t,
                                   (unsigned long long*)nullptr, pts_g, p32_g, nrep, ctx->score_dense_min);
            }
            PGX_HIP(ctx, hipGetLastError());
            if (ctx->score_profile) PGX_HIP(ctx, hipEventRecord(ctx->kev[2], ctx->stream));
            if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventRecord(ctx->kev[4], ctx->stream));
            long long* mirror = nullptr;
            if (ctx->score_mirror && !want_masks) {
                const size_t need = (size_t)ctx->Mpad * 24;
                if (ctx->h_mirror_cap < need) {
                    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // an earlier launch may still be writing the old one
                    if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
                    ctx->h_mirror = nullptr; ctx->h_mirror_cap = 0;
                    PGX_HIP(ctx, hipHostMalloc(&ctx->h_mirror, need * 2, hipHostMallocMapped | hipHostMallocCoherent));
                    ctx->h_mirror_cap = need * 2;
                }
                mirror = (long long*)ctx->h_mirror;
            }
            hipLaunchKernelGGL(score_finish_kernel, dim3((unsigned)((ctx->M + 255) / 256)), dim3(256), 0, ctx->stream, acc, ctx->M,
                               ctx->Mpad, qscale, ctx->perm.as<int>(), ctx->counts.as<long long>(), ctx->values.as<double>(),
                               ctx->shared.as<double>(), nrep, mirror);
            PGX_HIP(ctx, hipGetLastError());
            ctx->mirror_valid = mirror != nullptr;
            ctx->last_acc = acc; ctx->last_nrep = nrep; ctx->last_qscale = qscale; ctx->last_acc_M = ctx->M; ctx->last_acc_Mpad = ctx->Mpad;
            if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventRecord(ctx->kev[3], ctx->stream));
            ctx->last_score_path = 2;
            return PGX_OK;
        }
    }
    ctx->last_score_path = 1;
    ctx->last_acc = nullptr;
    if (ctx->score_profile) PGX_HIP(ctx, hipEventRecord(ctx->kev[0], ctx->stream));
    if constexpr (Filter<MT>::enabled) {
        if (want_masks) {
            if (filt32) score_launch_one<MT, true, 2>(ctx, T2, has_compound, guard, guard32);
            else if (filt) score_launch_one<MT, true, 1>(ctx, T2, has_compound, guard);
            else score_launch_one<MT, true, 0>(ctx, T2, has_compound, guard);
        } else {
            if (filt32) score_launch_one<MT, false, 2>(ctx, T2, has_compound, guard, guard32);
            else if (filt) score_launch_one<MT, false, 1>(ctx, T2, has_compound, guard);
            else score_launch_one<MT, false, 0>(ctx, T2, has_compound, guard);
        }
    } else {
        if (want_masks) score_launch_one<MT, true, 0>(ctx, T2, has_compound, guard);
        else score_launch_one<MT, false, 0>(ctx, T2, has_compound, guard);
    }
    PGX_HIP(ctx, hipGetLastError());
    if (ctx->score_profile) PGX_HIP(ctx, hipEventRecord(ctx->kev[1], ctx->stream));
    if (ctx->score_profile >= 2) { PGX_HIP(ctx, hipEventRecord(ctx->kev[2], ctx->stream)); PGX_HIP(ctx, hipEventRecord(ctx->kev[4], ctx->stream)); }
    hipLaunchKernelGGL(score_reduce_kernel, dim3((unsigned)((ctx->M + 63) / 64)), dim3(64 * kReduceWaves), 0,
                       ctx->stream, ctx->pcnt.as<unsigned>(), ctx->pval.as<double>(), ctx->psh.as<double>(),
                       ctx->chunks, ctx->Mpad, ctx->M, ctx->perm.as<int>(), ctx->counts.as<long long>(),
                       ctx->values.as<double>(), ctx->shared.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventRecord(ctx->kev[3], ctx->stream));
    return PGX_OK;
}

// ---- spatially sorted copies of the point data (group test) ----------------------------------------------------------
__global__ __launch_bounds__(256) void gather_f64_kernel(const double* __restrict__ src, const int* __restrict__ pperm, int64_t n,
                                                         double* __restrict__ dst)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n) dst[j] = src[pperm[j]];
}

// masks written in sorted bit order -> the caller's point order: one thread per (row, sorted word), set bits scattered
__global__ __launch_bounds__(256) void mask_unpermute_kernel(const unsigned long long* __restrict__ ms, const int* __restrict__ pperm,
                                                             int64_t n, int64_t words, int M, unsigned long long* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * words) return;
    const int64_t row = t / words, w = t % words;
    unsigned long long bits = ms[t];
    while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const int64_t j = w * 64 + b;
        if (j < n) {
            const int i = pperm[j];
            atomicOr(&out[row * words + (i >> 6)], 1ull << (i & 63));
        }
    }
}

// Host: Morton order of all coordinates, sorted copies, per-group bounds (rows of kGroupRow floats, see above).
int score_sort_points(pgx_ctx* ctx, const double* points, const float* p32, const double* pmax)
{
    const int64_t n = ctx->n;
    const int d = ctx->D;
    const int bits = 30 / d;
    std::vector<double> lo((size_t)d), inv((size_t)d);
    for (int k = 0; k < d; ++k) {
        double a = points[k], b = points[k];
        for (int64_t i = 1; i < n; ++i) { const double v = points[i * d + k]; if (v < a) a = v; if (v > b) b = v; }
        if (!std::isfinite(a) || !std::isfinite(b)) return PGX_OK;  // non-finite data: no sorted copies, no group test
        lo[(size_t)k] = a;
        inv[(size_t)k] = b > a ? (double)(1u << bits) / (b - a) : 0.0;
    }
    std::vector<uint64_t> kv((size_t)n), tmp((size_t)n);
    const uint32_t qmax = (1u << bits) - 1u;
    // spread[k][v]: the bits of v placed where coordinate k's bits sit in the interleaved key (bit b -> b*d + d-1-k)
    std::vector<std::vector<uint32_t>> spread((size_t)d, std::vector<uint32_t>((size_t)qmax + 1));
    for (int k = 0; k < d; ++k)
        for (uint32_t v = 0; v <= qmax; ++v) {
            uint32_t out = 0;
            for (int b = 0; b < bits; ++b) out |= ((v >> b) & 1u) << (b * d + d - 1 - k);
            spread[(size_t)k][v] = out;
        }
    for (int64_t i = 0; i < n; ++i) {
        uint32_t key = 0;
        for (int k = 0; k < d; ++k) {
            const double t = (points[i * d + k] - lo[(size_t)k]) * inv[(size_t)k];
            uint32_t v = t > 0.0 ? (uint32_t)t : 0u;
            key |= spread[(size_t)k][v > qmax ? qmax : v];
        }
        kv[(size_t)i] = ((uint64_t)key << 32) | (uint32_t)i;
    }
    for (int pass = 0; pass < 4; ++pass) {  // LSD radix sort on the key's four bytes (stable: ties keep index order)
        size_t cnt[257] = {0};
        const int sh = 32 + 8 * pass;
        for (int64_t i = 0; i < n; ++i) ++cnt[((kv[(size_t)i] >> sh) & 0xff) + 1];
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) tmp[cnt[(kv[(size_t)i] >> sh) & 0xff]++] = kv[(size_t)i];
        kv.swap(tmp);
    }
    const int64_t groups = (n + 63) / 64;
    std::vector<int> pperm((size_t)n);
    std::vector<double> sp((size_t)n * d), spm((size_t)n);
    const int64_t supers = (groups + kSuper - 1) / kSuper;
    std::vector<float> sp32((size_t)n * 8), gb((size_t)(groups + supers) * kGroupRow, 0.0f);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t i = (int64_t)(uint32_t)kv[(size_t)j];
        pperm[(size_t)j] = (int)i;
        for (int k = 0; k < d; ++k) sp[(size_t)j * d + k] = points[i * d + k];
        for (int k = 0; k < 8; ++k) sp32[(size_t)j * 8 + k] = p32[i * 8 + k];
        spm[(size_t)j] = pmax[i];
    }
    // which coordinates the projective map multiplies / which are observed (as in pgx_set_points)
    int in0, in1, ob0;
    if (ctx->model_type == kPnP) { in0 = 2; in1 = 4; ob0 = 0; }
    else { in0 = 0; in1 = 1; ob0 = 2; }
    auto bounds = [&](int64_t a, int64_t b, float* row) {  // bounds of the sorted points [a, b)
        // centre = box centre of the f32 rows (the values the kernel's per-point filter sees are not needed here: the
        // bound is about the exact f64 points; extents are inflated below)
        double cmin[3] = {0, 0, 0}, cmax[3] = {0, 0, 0}, omin[2], omax[2];
        for (int k = in0; k <= in1; ++k) { cmin[k - in0] = cmax[k - in0] = sp[(size_t)a * d + k]; }
        for (int k = 0; k < 2; ++k) omin[k] = omax[k] = sp[(size_t)a * d + ob0 + k];
        for (int64_t j = a; j < b; ++j) {
            for (int k = in0; k <= in1; ++k) { const double v = sp[(size_t)j * d + k]; if (v < cmin[k - in0]) cmin[k - in0] = v; if (v > cmax[k - in0]) cmax[k - in0] = v; }
            for (int k = 0; k < 2; ++k) { const double v = sp[(size_t)j * d + ob0 + k]; if (v < omin[k]) omin[k] = v; if (v > omax[k]) omax[k] = v; }
        }
        float cf[3] = {0, 0, 0};
        double scale = 1.0;
        for (int k = 0; k <= in1 - in0; ++k) { cf[k] = (float)(0.5 * (cmin[k] + cmax[k])); if (std::fabs((double)cf[k]) > scale) scale = std::fabs((double)cf[k]); }
        double rho2 = 0.0;  // radius about the f32 centre actually stored
        for (int64_t j = a; j < b; ++j) {
            double s2 = 0.0;
            for (int k = in0; k <= in1; ++k) { const double df = sp[(size_t)j * d + k] - (double)cf[k - in0]; s2 += df * df; }
            if (s2 > rho2) rho2 = s2;
        }
        const float ub = (float)(0.5 * (omin[0] + omax[0])), vb = (float)(0.5 * (omin[1] + omax[1]));
        double ru = 0.0, rv = 0.0;
        for (int64_t j = a; j < b; ++j) {
            const double du = std::fabs(sp[(size_t)j * d + ob0] - (double)ub), dv = std::fabs(sp[(size_t)j * d + ob0 + 1] - (double)vb);
            if (du > ru) ru = du;
            if (dv > rv) rv = dv;
        }
        row[0] = cf[0]; row[1] = cf[1]; row[2] = cf[2];
        row[3] = (float)(std::sqrt(rho2) * kGroupInflate + 1e-30);
        row[4] = ub; row[5] = vb;
        row[6] = (float)(ru * kGroupInflate + 1e-30);
        row[7] = (float)(rv * kGroupInflate + 1e-30);
        row[8] = (float)(scale * 1.000001);
    };
    for (int64_t g = 0; g < groups; ++g) bounds(g * 64, g * 64 + 64 < n ? g * 64 + 64 : n, gb.data() + (size_t)g * kGroupRow);
    // super-groups of kSuper consecutive groups (512 points): the cull kernel tests them first; rows behind the group rows
    for (int64_t sg = 0; sg < supers; ++sg)
        bounds(sg * kSuper * 64, (sg + 1) * kSuper * 64 < n ? (sg + 1) * kSuper * 64 : n, gb.data() + (size_t)(groups + sg) * kGroupRow);
    {   // group-blocked SoA copies for the group-major kernel: [group][coordinate][64]; the tail group repeats its last row
        std::vector<double> pg((size_t)groups * d * 64);
        std::vector<float> p32g((size_t)groups * 8 * 64, 0.0f);
        for (int64_t g = 0; g < groups; ++g)
            for (int l = 0; l < 64; ++l) {
                const int64_t j = g * 64 + l < n ? g * 64 + l : n - 1;
                for (int k = 0; k < d; ++k) pg[((size_t)g * d + k) * 64 + l] = sp[(size_t)j * d + k];
                for (int k = 0; k < 8; ++k) p32g[((size_t)g * 8 + k) * 64 + l] = sp32[(size_t)j * 8 + k];
            }
        PGX_TRY(ensure(ctx, ctx->pts_g, pg.size() * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->p32_g, p32g.size() * sizeof(float)));
        PGX_HIP(ctx, hipMemcpyAsync(ctx->pts_g.p, pg.data(), pg.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        PGX_HIP(ctx, hipMemcpyAsync(ctx->p32_g.p, p32g.data(), p32g.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging vectors die at the end of this block
    }
    PGX_TRY(ensure(ctx, ctx->pts_s, (size_t)n * d * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pts32_s, (size_t)n * 8 * sizeof(float)));
    PGX_TRY(ensure(ctx, ctx->pmax_s, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->comp_s, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pperm, (size_t)n * sizeof(int)));
    PGX_TRY(ensure(ctx, ctx->gbounds, (size_t)(groups + supers) * kGroupRow * sizeof(float)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pts_s.p, sp.data(), (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pts32_s.p, sp32.data(), (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pmax_s.p, spm.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pperm.p, pperm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->gbounds.p, gb.data(), (size_t)(groups + supers) * kGroupRow * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemsetAsync(ctx->comp_s.p, 0, (size_t)n * sizeof(double), ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->point_sort = 1;
    ctx->comp_dirty = 1;
    return PGX_OK;
}

static int score_launch_typed(pgx_ctx* ctx, double T2, int has_compound, int want_masks);

// Chooses the point chunking so that the grid has >= ~8 blocks per CU (all 32 wave slots of every CU filled)
// while chunks stay multiples of 64 points (mask words never straddle blocks).
int score_launch(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    ctx->mirror_valid = 0;   // set by the path whose last kernel writes the host mirror
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score: points not set");
    if (ctx->M <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score: no hypotheses uploaded");
    const int groups = ctx->Mpad / kScoreBlock;
    // Work per (hypothesis, point) pair is data dependent (exact path only for candidates), so the grid is over-
    // decomposed: ~blocks_per_cu blocks per CU keep the tail behind the slowest block short.
    const int target_blocks = (ctx->cu_count > 0 ? ctx->cu_count : 256) * ctx->score_blocks_per_cu;
    int64_t chunks = (target_blocks + groups - 1) / groups;
    int64_t chunk = (ctx->n + chunks - 1) / chunks;
    chunk = ((chunk + 63) / 64) * 64;
    if (chunk < 64) chunk = 64;
    if (chunk > 65472) chunk = 65472;  // queue entries of the deferred kernel are 16-bit offsets into the chunk
    chunks = (ctx->n + chunk - 1) / chunk;
    if (chunks > 65535) {  // gridDim.y limit
        chunks = 65535;
        chunk = (((ctx->n + chunks - 1) / chunks + 63) / 64) * 64;
        chunks = (ctx->n + chunk - 1) / chunk;
    }
    ctx->chunk = chunk;
    ctx->chunks = (int)chunks;
    ctx->words = (ctx->n + 63) / 64;
    const size_t np = (size_t)chunks * (size_t)ctx->Mpad;
    PGX_TRY(ensure(ctx, ctx->pcnt, np * sizeof(unsigned)));
    PGX_TRY(ensure(ctx, ctx->pval, np * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->psh, np * sizeof(double)));
    // counts | values | shared live in ONE allocation (values / shared are views into it): pgx_score_fetch brings all three
    // back with a single copy (three small copies cost ~5 us each on the critical path of a 0.3 ms step)
    PGX_TRY(ensure(ctx, ctx->counts, (size_t)3 * ctx->Mpad * sizeof(long long)));
    ctx->values.p = ctx->counts.as<char>() + (size_t)ctx->Mpad * 8;
    ctx->shared.p = ctx->counts.as<char>() + (size_t)2 * ctx->Mpad * 8;
    ctx->values.cap = ctx->shared.cap = 0;  // not owned
    if (want_masks) PGX_TRY(ensure(ctx, ctx->masks, (size_t)ctx->M * (size_t)ctx->words * sizeof(uint64_t)));
    ctx->have_masks = want_masks != 0;
    if (ctx->point_sort) {
        if (has_compound && ctx->comp_dirty) {  // the kernel reads the compound vector in sorted point order
            hipLaunchKernelGGL(gather_f64_kernel, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->comp.as<double>(), ctx->pperm.as<int>(), ctx->n, ctx->comp_s.as<double>());
            PGX_HIP(ctx, hipGetLastError());
            ctx->comp_dirty = 0;
        }
        if (want_masks) PGX_TRY(ensure(ctx, ctx->masks_s, (size_t)ctx->M * (size_t)ctx->words * sizeof(uint64_t)));
    }
    const int rc = score_launch_typed(ctx, T2, has_compound, want_masks);
    if (rc == PGX_OK && ctx->point_sort && want_masks) {
        const int64_t total = (int64_t)ctx->M * ctx->words;
        PGX_HIP(ctx, hipMemsetAsync(ctx->masks.p, 0, (size_t)total * sizeof(uint64_t), ctx->stream));
        hipLaunchKernelGGL(mask_unpermute_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->masks_s.as<unsigned long long>(), ctx->pperm.as<int>(), ctx->n, ctx->words, ctx->M,
                           ctx->masks.as<unsigned long long>());
        PGX_HIP(ctx, hipGetLastError());
    }
    return rc;
}

static int score_launch_typed(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    switch (ctx->model_type) {
    case kLine2D: return score_dispatch<kLine2D>(ctx, T2, has_compound, want_masks);
    case kHomography: return score_dispatch<kHomography>(ctx, T2, has_compound, want_masks);
    case kFundamental: return score_dispatch<kFundamental>(ctx, T2, has_compound, want_masks);
    case kPnP: return score_dispatch<kPnP>(ctx, T2, has_compound, want_masks);
    case kVanishingPoint: return score_dispatch<kVanishingPoint>(ctx, T2, has_compound, want_masks);
    case kHomographySym: return score_dispatch<kHomographySym>(ctx, T2, has_compound, want_masks);
    default: return fail(ctx, PGX_ERR_INVALID, "pgx_score: bad model type %d", ctx->model_type);
    }
}

}  // namespace pgx
