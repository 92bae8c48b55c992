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
#include "pgx_internal.h"

namespace pgx {

constexpr int kScoreBlock = 256;

template <int MT, bool MASK>
__global__ __launch_bounds__(kScoreBlock) void score_kernel(
    const double* __restrict__ pts, int64_t n, const double* __restrict__ models, int M, int Mpad,
    double T2, const double* __restrict__ comp, int has_comp, int64_t chunk,
    unsigned* __restrict__ pcnt, double* __restrict__ pval, double* __restrict__ psh,
    unsigned long long* __restrict__ masks, int64_t words)
{
    using R = Residual<MT>;
    const int m = blockIdx.x * kScoreBlock + threadIdx.x;
    const bool live = m < M;
    const int64_t i0 = (int64_t)blockIdx.y * chunk;
    const int64_t i1 = (i0 + chunk < n) ? (i0 + chunk) : n;

    double mdl[R::P];
#pragma unroll
    for (int k = 0; k < R::P; ++k)
        mdl[k] = live ? models[(int64_t)m * R::P + k] : __builtin_nan("");  // NaN model: never an inlier

    unsigned cnt = 0;
    double val = 0.0, sh = 0.0;
    unsigned long long word = 0;

    for (int64_t i = i0; i < i1; ++i) {
        const double* __restrict__ prow = pts + i * R::D;  // wave-uniform address -> scalar loads
        double pt[R::D];
#pragma unroll
        for (int k = 0; k < R::D; ++k) pt[k] = prow[k];
        const double sq = R::squared(pt, mdl);
        const bool inl = sq < T2;  // strict, scoring_function_with_compound_model.h:85
        if (inl) {
            ++cnt;                                        // :91
            const double s = cv_max(0.0, 1.0 - sq / T2);  // :94
            val += s;                                     // :97
            if (has_comp) sh += cv_min(comp[i], s);       // :115-117 (pref = 0 for non-inliers adds +0)
        }
        if (MASK) {
            word |= (unsigned long long)(inl ? 1 : 0) << (i & 63);
            if ((i & 63) == 63 || i == i1 - 1) {
                if (live) masks[(int64_t)m * words + (i >> 6)] = word;  // :88 inlier list as a bit mask
                word = 0;
            }
        }
    }
    const int64_t o = (int64_t)blockIdx.y * Mpad + m;
    pcnt[o] = cnt;
    pval[o] = val;
    psh[o] = sh;
}

// Adds the chunk partials of each hypothesis in chunk order (fixed => bit-reproducible).
__global__ __launch_bounds__(256) void score_reduce_kernel(
    const unsigned* __restrict__ pcnt, const double* __restrict__ pval, const double* __restrict__ psh,
    int chunks, int Mpad, int M, long long* __restrict__ counts, double* __restrict__ values,
    double* __restrict__ shared)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    long long c = 0;
    double v = 0.0, s = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const int64_t o = (int64_t)k * Mpad + m;
        c += pcnt[o];
        v += pval[o];
        s += psh[o];
    }
    counts[m] = c;
    values[m] = v;
    shared[m] = s;
}

template <int MT>
static int score_dispatch(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    dim3 grid((unsigned)(ctx->Mpad / kScoreBlock), (unsigned)ctx->chunks);
    dim3 block(kScoreBlock);
    if (want_masks)
        hipLaunchKernelGGL((score_kernel<MT, true>), grid, block, 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->models.as<double>(), ctx->M, ctx->Mpad, T2, ctx->comp.as<double>(), has_compound,
                           ctx->chunk, ctx->pcnt.as<unsigned>(), ctx->pval.as<double>(), ctx->psh.as<double>(),
                           ctx->masks.as<unsigned long long>(), ctx->words);
    else
        hipLaunchKernelGGL((score_kernel<MT, false>), grid, block, 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->models.as<double>(), ctx->M, ctx->Mpad, T2, ctx->comp.as<double>(), has_compound,
                           ctx->chunk, ctx->pcnt.as<unsigned>(), ctx->pval.as<double>(), ctx->psh.as<double>(),
                           (unsigned long long*)nullptr, ctx->words);
    PGX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(score_reduce_kernel, dim3((unsigned)((ctx->M + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->pcnt.as<unsigned>(), ctx->pval.as<double>(), ctx->psh.as<double>(), ctx->chunks,
                       ctx->Mpad, ctx->M, ctx->counts.as<long long>(), ctx->values.as<double>(),
                       ctx->shared.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    return PGX_OK;
}

// Chooses the point chunking so that the grid has >= ~8 blocks per CU (all 32 wave slots of every CU filled)
// while chunks stay multiples of 64 points (mask words never straddle blocks).
int score_launch(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score: points not set");
    if (ctx->M <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score: no hypotheses uploaded");
    const int groups = ctx->Mpad / kScoreBlock;
    const int target_blocks = (ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    int64_t chunks = (target_blocks + groups - 1) / groups;
    int64_t chunk = (ctx->n + chunks - 1) / chunks;
    chunk = ((chunk + 63) / 64) * 64;
    if (chunk < 64) chunk = 64;
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
    PGX_TRY(ensure(ctx, ctx->counts, (size_t)ctx->Mpad * sizeof(long long)));
    PGX_TRY(ensure(ctx, ctx->values, (size_t)ctx->Mpad * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->shared, (size_t)ctx->Mpad * sizeof(double)));
    if (want_masks) PGX_TRY(ensure(ctx, ctx->masks, (size_t)ctx->M * (size_t)ctx->words * sizeof(uint64_t)));
    ctx->have_masks = want_masks != 0;
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
