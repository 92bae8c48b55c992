// pointwise.hip — the HBM-bound, one-point-per-lane kernels of the hot path (gfx950).
//
//   preference_kernel   progx::Model::setPreferenceVector (progx_model.h:70-87) fused with the three reductions of
//                       ProgressiveX::isPutativeModelValid (progressive_x.h:583-585)
//   compound_kernel     ProgressiveX::updateCompoundModel (progressive_x.h:597-624)
//   unary_kernel        pearl::dataEnergyFunctor (PEARL.h:82-128), all (point, label) pairs at once, quantised
//   residual_sum_kernel PEARL::parameterEstimation's before/after sums (PEARL.h:369-371, 388-390)
//   bucket_*            PEARL::parameterEstimation's label bucketing (PEARL.h:342-352): histogram + stable compaction
//                       with wave ballot / prefix sums
//   energy_kernel       GCoptimization::compute_energy (absent upstream, U-5): data + Potts + label costs, exact int64
//
// Model parameters are wave-uniform (kernel arguments or scalar loads); points are read once, coalesced per wave.
// All floating-point reductions use a fixed tree (wave shuffle -> LDS -> per-block partial -> one-block final pass), so
// results are bit-reproducible run to run.  Paths are relative to /root/reference/src/pyprogressivex/.
#include <cstring>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {

struct ModelArg {
    double v[18];
};

constexpr int kPwBlock = 256;

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// block sum of up to 3 values; result valid in thread 0. Fixed order: lanes tree, then waves 0..3.
template <int NV>
__device__ __forceinline__ void block_sum(double (&x)[NV], double* lds /* >= 4*NV */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        x[k] = wave_sum(x[k]);
        if (lane == 0) lds[wave * NV + k] = x[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = lds[k];
            for (int w = 1; w < kPwBlock / 64; ++w) s += lds[w * NV + k];
            x[k] = s;
        }
    }
}

// final pass: one block adds `count` partial NV-tuples in a fixed order
template <int NV>
__global__ __launch_bounds__(kPwBlock) void final_sum_kernel(const double* __restrict__ partials, int count,
                                                             double* __restrict__ out)
{
    __shared__ double lds[4 * NV];
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < count; b += kPwBlock)
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] += partials[(int64_t)b * NV + k];
    block_sum<NV>(acc, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = acc[k];
}

template <int MT>
__device__ __forceinline__ void load_point(const double* __restrict__ pts, int64_t i, double (&pt)[Residual<MT>::D])
{
#pragma unroll
    for (int k = 0; k < Residual<MT>::D; ++k) pt[k] = pts[i * Residual<MT>::D + k];
}

// ---- a2 + a3 -----------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(kPwBlock) void preference_kernel(const double* __restrict__ pts, int64_t n, ModelArg mdl,
                                                              double T2, const double* __restrict__ comp,
                                                              double* __restrict__ pref,
                                                              double* __restrict__ partials)
{
    using R = Residual<MT>;
    __shared__ double lds[12];
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    if (i < n) {
        double pt[R::D];
        load_point<MT>(pts, i, pt);
        const double sq = R::squared(pt, mdl.v);
        const double v = 1.0 - sq / T2;
        const double p = cv_max(0.0, v);  // progx_model.h:85  MAX(0, ...)
        pref[i] = p;
        const double c = comp[i];
        acc[0] = p * c;  // progressive_x.h:583 dot
        acc[1] = p * p;  // :585 squaredNorm
        acc[2] = c * c;
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        partials[(int64_t)blockIdx.x * 3 + 0] = acc[0];
        partials[(int64_t)blockIdx.x * 3 + 1] = acc[1];
        partials[(int64_t)blockIdx.x * 3 + 2] = acc[2];
    }
}

template <int MT>
static int preference_dispatch(pgx_ctx* ctx, const ModelArg& mdl, double T2, double* d_pref, int blocks)
{
    hipLaunchKernelGGL((preference_kernel<MT>), dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream,
                       ctx->pts.as<double>(), ctx->n, mdl, T2, ctx->comp.as<double>(), d_pref,
                       ctx->red_partials.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    return PGX_OK;
}

int preference_launch(pgx_ctx* ctx, const double* model, double T2, double* d_pref, double out3[3])
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_preference: points not set");
    ModelArg mdl;
    for (int k = 0; k < 18; ++k) mdl.v[k] = k < ctx->P ? model[k] : 0.0;
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    PGX_TRY(ensure(ctx, ctx->red_partials, (size_t)blocks * 3 * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->red_out, 8 * sizeof(double)));
    int r;
    switch (ctx->model_type) {
    case kLine2D: r = preference_dispatch<kLine2D>(ctx, mdl, T2, d_pref, blocks); break;
    case kHomography: r = preference_dispatch<kHomography>(ctx, mdl, T2, d_pref, blocks); break;
    case kFundamental: r = preference_dispatch<kFundamental>(ctx, mdl, T2, d_pref, blocks); break;
    case kPnP: r = preference_dispatch<kPnP>(ctx, mdl, T2, d_pref, blocks); break;
    case kVanishingPoint: r = preference_dispatch<kVanishingPoint>(ctx, mdl, T2, d_pref, blocks); break;
    case kHomographySym: r = preference_dispatch<kHomographySym>(ctx, mdl, T2, d_pref, blocks); break;
    default: return fail(ctx, PGX_ERR_INVALID, "bad model type");
    }
    PGX_TRY(r);
    hipLaunchKernelGGL((final_sum_kernel<3>), dim3(1), dim3(kPwBlock), 0, ctx->stream,
                       ctx->red_partials.as<double>(), blocks, ctx->red_out.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, out3, ctx->red_out.p, 3 * sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

// ---- a4 ----------------------------------------------------------------------------------------------------------
struct SlotArg {
    const double* p[32];
};

__global__ __launch_bounds__(kPwBlock) void compound_kernel(SlotArg slots, int K, int64_t n, double* __restrict__ comp, int accumulate)
{
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    double c = accumulate ? comp[i] : 0.0;  // progressive_x.h:604 setConstant(0); later chunks of > 32 models continue the max
    for (int k = 0; k < K; ++k) c = cv_max(c, slots.p[k][i]);  // :620-621
    comp[i] = c;
}

int compound_launch(pgx_ctx* ctx, const int32_t* slots, int K)
{
    for (int k = 0; k < K; ++k)
        if (slots[k] < 0 || slots[k] >= (int)ctx->slots.size() || !ctx->slots[slots[k]].p)
            return fail(ctx, PGX_ERR_INVALID, "pgx_compound_update: slot %d holds no preference vector", slots[k]);
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    // the kernel argument holds 32 pointers; max is associative and exact, so more models run as chunks of 32 that continue
    // from the compound vector written by the chunk before
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int kc = K - k0 < 32 ? K - k0 : 32;
        SlotArg a;
        for (int k = 0; k < 32; ++k) a.p[k] = k < kc ? ctx->slots[slots[k0 + k]].as<double>() : nullptr;
        hipLaunchKernelGGL(compound_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, a, kc, ctx->n,
                           ctx->comp.as<double>(), k0 > 0 ? 1 : 0);
        PGX_HIP(ctx, hipGetLastError());
    }
    ctx->comp_dirty = 1;
    return PGX_OK;
}

// ---- a6 ----------------------------------------------------------------------------------------------------------
// One lane per point, loop over the K active models (scalar loads of the parameters).  The table is stored
// label-major [L][n] so that both this write and the per-label reads of the expansion kernels are coalesced.
template <int MT>
__global__ __launch_bounds__(kPwBlock) void unary_kernel(const double* __restrict__ pts, int64_t n,
                                                         const double* __restrict__ models, int K, double T2,
                                                         double oml, long long* __restrict__ dq)
{
    using R = Residual<MT>;
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    double pt[R::D];
    load_point<MT>(pts, i, pt);
    const double far = 2.0 * oml;
    for (int k = 0; k < K; ++k) {
        double mdl[R::P];
#pragma unroll
        for (int j = 0; j < R::P; ++j) mdl[j] = models[k * R::P + j];
        const double sq = R::squared(pt, mdl);
        double c;
        if (sq > T2) c = far;            // PEARL.h:123-124
        else c = oml * sq / T2;          // PEARL.h:126-127
        if (c != c) c = far;             // NaN (degenerate model): priced as beyond the threshold
        dq[(int64_t)k * n + i] = (long long)__builtin_nearbyint(c * 4294967296.0);
    }
    dq[(int64_t)K * n + i] = (long long)__builtin_nearbyint(oml * 4294967296.0);  // PEARL.h:100-101
}

int unary_launch(pgx_ctx* ctx, int K, double threshold, double lambda)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_pearl_unary: points not set");
    const double T2 = 9.0 / 4.0 * threshold * threshold;  // PEARL.h:51
    const double oml = 1.0 - lambda;                       // PEARL.h:48
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    dim3 g((unsigned)blocks), b(kPwBlock);
    const double* pts = ctx->pts.as<double>();
    const double* mdl = ctx->kmodels.as<double>();
    long long* dq = ctx->dq.as<long long>();
    switch (ctx->model_type) {
    case kLine2D: hipLaunchKernelGGL((unary_kernel<kLine2D>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    case kHomography: hipLaunchKernelGGL((unary_kernel<kHomography>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    case kFundamental: hipLaunchKernelGGL((unary_kernel<kFundamental>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    case kPnP: hipLaunchKernelGGL((unary_kernel<kPnP>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    case kVanishingPoint: hipLaunchKernelGGL((unary_kernel<kVanishingPoint>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    case kHomographySym: hipLaunchKernelGGL((unary_kernel<kHomographySym>), g, b, 0, ctx->stream, pts, ctx->n, mdl, K, T2, oml, dq); break;
    default: return fail(ctx, PGX_ERR_INVALID, "bad model type");
    }
    PGX_HIP(ctx, hipGetLastError());
    return PGX_OK;
}

// ---- GC-RANSAC's inlier/outlier graph cut ---------------------------------------------------------
// gcransac::GCRANSAC::labeling is part of the absent graph-cut-ransac submodule (call site progressive_x.h:294-299,
// settings :541-545); restated from memory of upstream [U-12]: with e_i = clamp(r_i^2 / T^2, 0, 1),
//   unary     r_i^2 <= T^2 : outlier costs (1 - lambda)(1 - e_i), inlier 0      else : outlier 0, inlier (1 - lambda) e_i
//   pairwise  (each undirected neighbour pair once, self loops skipped)
//             both outliers lambda (e_i + e_j) / 2,  labels differ lambda,  both inliers 0
// minimised exactly by one s-t cut; inliers = the SINK segment = the sites that still reach t (ties -> outlier).
// With o = [outlier] the pair term equals  q (o_i + o_j) + (lambda - q) [o_i != o_j],  q = lambda (e_i + e_j) / 4, i.e. a
// unary share per end point plus a Potts edge of weight lambda - q >= lambda / 2: exactly the graph of one expansion move
// (every site "inlier", alpha = "outlier", no label costs) with per-arc weights.  All terms are quantised to 2^-32 first
// (q per pair, so that both end points and both arcs see the same integer), sums are int64: order-free and bit-exact.
template <int MT>
__global__ __launch_bounds__(kPwBlock) void gc_energy_kernel(const double* __restrict__ pts, int64_t n, ModelArg mdl, double T2,
                                                             double* __restrict__ e)
{
    using R = Residual<MT>;
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    double pt[R::D];
    load_point<MT>(pts, i, pt);
    const double sq = R::squared(pt, mdl.v);
    // negative = beyond the threshold (or NaN): e = 1; the sign carries the branch of the unary term
    e[i] = (sq <= T2) ? cv_max(0.0, sq / T2) : -1.0;
}

__global__ __launch_bounds__(kPwBlock) void gc_terms_kernel(const double* __restrict__ e, int64_t n, const int* __restrict__ off,
                                                            const int* __restrict__ idx, double lambda, long long lambda_q,
                                                            long long* __restrict__ dq, long long* __restrict__ wq,
                                                            int* __restrict__ labels, int start_label)
{
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    const double ei_s = e[i];
    const bool inl = ei_s >= 0.0;
    const double ei = inl ? ei_s : 1.0;
    const double oml = 1.0 - lambda;
    long long out = inl ? (long long)__builtin_nearbyint(oml * (1.0 - ei) * 4294967296.0) : 0;
    for (int a = off[i]; a < off[i + 1]; ++a) {
        const int j = idx[a];
        if (j == i) { wq[a] = 0; continue; }
        const double ej_s = e[j];
        const double ej = ej_s >= 0.0 ? ej_s : 1.0;
        const long long q = (long long)__builtin_nearbyint(lambda * 0.25 * (ei + ej) * 4294967296.0);
        out += q;
        wq[a] = lambda_q - q;
    }
    dq[i] = out;                                                                    // label 0 = outlier (alpha)
    dq[n + i] = inl ? 0 : (long long)__builtin_nearbyint(oml * ei * 4294967296.0);  // label 1 = inlier
    labels[i] = start_label;
}

// ---- ascending indices of the set flags of a SMALL array (the reference's own scenes: a few hundred points) ----------------------
// hipcub's select is two launches behind a flags kernel, and the count has to come back before the indices can be copied: two host
// round trips.  Up to kCompactSmall items one workgroup does it in one launch - each thread takes a run of consecutive items, a
// wave scan + 16 wave totals place them - and writes count | indices into ONE buffer that goes back in one copy.
constexpr int64_t kCompactSmall = 8192;

template <bool MASK>
__global__ __launch_bounds__(1024) void compact_small_kernel(const unsigned long long* __restrict__ row, const int* __restrict__ flags, int n,
                                                             int* __restrict__ out)
{
    __shared__ int wsum[16];
    const int tid = (int)threadIdx.x;
    const int per = (n + 1023) / 1024;   // <= 8
    const int i0 = tid * per;
    unsigned m = 0;
    int c = 0;
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        const bool f = i < n && (MASK ? ((row[i >> 6] >> (i & 63)) & 1ull) != 0ull : flags[i] != 0);
        if (f) { m |= 1u << k; ++c; }
    }
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    int pos = base + incl - c;
    for (int k = 0; k < per; ++k)
        if ((m >> k) & 1u) out[1 + pos++] = i0 + k;
    if (tid == 1023) out[0] = base + incl;
}

// enqueue: count | indices of the flagged items into ctx->gc_sel and from there into the pinned staging buffer (the caller synchronises)
template <bool MASK>
static int compact_small_enqueue(pgx_ctx* ctx, const unsigned long long* row, const int* flags, int64_t n)
{
    PGX_TRY(ensure(ctx, ctx->gc_sel, (size_t)(n + 1) * 4));
    void* hs = nullptr;
    PGX_TRY(host_staging(ctx, (size_t)(n + 1) * 4, &hs));
    hipLaunchKernelGGL((compact_small_kernel<MASK>), dim3(1), dim3(1024), 0, ctx->stream, row, flags, (int)n, ctx->gc_sel.as<int>());
    PGX_HIP(ctx, hipGetLastError());
    PGX_HIP(ctx, hipMemcpyAsync(hs, ctx->gc_sel.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    return PGX_OK;
}

int gc_labeling_launch(pgx_ctx* ctx, const double* model, double T2, double lambda, int32_t* flags, int64_t* count, bool want_index)
{
    const int64_t n = ctx->n;
    if (n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_labeling: points not set");
    if (ctx->gn != n) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_labeling: needs a neighbourhood graph over the %lld points", (long long)n);
    if (!(lambda > 0.0) || !(lambda < 1.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_labeling: lambda must lie in (0, 1)");
    if (!(T2 > 0.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_labeling: threshold must be positive");
    const int64_t E = ctx->gE;
    // e[n] f64 | dq[2][n] i64 | wq[E] i64 | labels[n] i32
    PGX_TRY(ensure(ctx, ctx->gc, (size_t)n * 8 + (size_t)2 * n * 8 + (size_t)(E > 0 ? E : 1) * 8 + (size_t)n * 4));
    double* e = ctx->gc.as<double>();
    long long* dq = (long long*)(e + n);
    long long* wq = dq + 2 * n;
    int* labels = (int*)(wq + (E > 0 ? E : 1));
    ModelArg mdl;
    int pd = 0, pp = 0;
    model_dims(ctx->model_type, &pd, &pp);
    for (int j = 0; j < 18; ++j) mdl.v[j] = j < pp ? model[j] : 0.0;
    const int blocks = (int)((n + kPwBlock - 1) / kPwBlock);
    dim3 g((unsigned)blocks), b(kPwBlock);
    const double* pts = ctx->pts.as<double>();
    switch (ctx->model_type) {
    case kLine2D: hipLaunchKernelGGL((gc_energy_kernel<kLine2D>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    case kHomography: hipLaunchKernelGGL((gc_energy_kernel<kHomography>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    case kFundamental: hipLaunchKernelGGL((gc_energy_kernel<kFundamental>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    case kPnP: hipLaunchKernelGGL((gc_energy_kernel<kPnP>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    case kVanishingPoint: hipLaunchKernelGGL((gc_energy_kernel<kVanishingPoint>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    case kHomographySym: hipLaunchKernelGGL((gc_energy_kernel<kHomographySym>), g, b, 0, ctx->stream, pts, n, mdl, T2, e); break;
    default: return fail(ctx, PGX_ERR_INVALID, "bad model type");
    }
    PGX_HIP(ctx, hipGetLastError());
    const long long lambda_q = quantize_lambda(lambda);
    // Orientation.  As stated above every site starts "inlier" and alpha = "outlier": then every site beyond the threshold
    // holds excess (95 % of the sites of a 10^6-point pose problem: 64-level searches, sweeps over all sites).  The same cut with
    // the terminals swapped - every site starts "outlier", alpha = "inlier", and alpha goes to the sites the SOURCE reaches
    // (maxflow.hip mf_k_src_*: the minimal source side, i.e. exactly "reaches t" of the original orientation, ties -> outlier) -
    // has the ~5 % near the model as its excess sites.  Same flags bit for bit (PGX_GC_FLIP=0 for A/B; GPU test).
    // A graph the one-workgroup solver takes (<= 8192 sites: the reference's own scenes) is cut in the original orientation - one
    // launch instead of the level-synchronous schedule's dozens (0.6 ms -> 0.1 ms per cut on a 187-point scene, a fifth of a
    // findTwoViewMotions call there); that solver applies alpha to the sites that cannot reach t, i.e. the same flags.
    const bool flip = ctx->gc_flip != 0 && !(ctx->mf_tile && n <= ctx->tile_single_max && n <= 8192);
    hipLaunchKernelGGL(gc_terms_kernel, g, b, 0, ctx->stream, e, n, ctx->goff.as<int>(), ctx->gidx.as<int>(), lambda, lambda_q, dq,
                       wq, labels, flip ? 0 : 1);
    PGX_HIP(ctx, hipGetLastError());
    int64_t changed = 0;
    // a small scene's cut is one one-workgroup launch: the compaction of its flags and the copy back ride behind it, before the one
    // synchronisation of the move (two round trips before)
    const bool small = want_index && !flip && n <= kCompactSmall;
    if (small) ctx->tile_pre_sync = [ctx, labels, n]() { return compact_small_enqueue<false>(ctx, nullptr, labels, n); };
    ctx->tile_pre_sync_ran = false;
    const int rc = expand_alpha_on(ctx, n, 2, dq, labels, wq, lambda_q, 0, flip ? 1 : 0, &changed, flip);
    ctx->tile_pre_sync = nullptr;
    const bool have_small = ctx->tile_pre_sync_ran;
    ctx->tile_pre_sync_ran = false;
    PGX_TRY(rc);
    const int64_t inliers = flip ? changed : n - changed;
    if (have_small) {
        const int* hs = (const int*)ctx->h_res;
        if (hs[0] != (int)inliers) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_inliers (internal): %d indices for %lld inliers", hs[0], (long long)inliers);
        memcpy(flags, hs + 1, (size_t)inliers * sizeof(int32_t));
        if (count) *count = inliers;
        return PGX_OK;
    }
    if (want_index) {
        // the inliers' indices in ascending order instead of n flags: the caller (the local optimisation's sampler) used to copy
        // 4 n bytes back and scan them with numpy.flatnonzero - 1.9 ms per cut at n = 10^6, as long as the cut itself
        size_t temp_bytes = 0;
        hipcub::CountingInputIterator<int> ids(0);
        PGX_HIP(ctx, hipcub::DeviceSelect::Flagged(nullptr, temp_bytes, ids, labels, (int*)nullptr, (int*)nullptr, (int)n, ctx->stream));
        const size_t sel_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;   // (the select's scratch holds 64-bit words it updates atomically: keep it aligned)
        PGX_TRY(ensure(ctx, ctx->gc_sel, sel_bytes + 256 + temp_bytes));
        int* d_sel = ctx->gc_sel.as<int>();
        int* d_num = (int*)((char*)ctx->gc_sel.p + sel_bytes);
        void* d_temp = (void*)((char*)ctx->gc_sel.p + sel_bytes + 256);
        PGX_HIP(ctx, hipcub::DeviceSelect::Flagged(d_temp, temp_bytes, ids, labels, d_sel, d_num, (int)n, ctx->stream));
        if (inliers > 0) PGX_HIP(ctx, hipMemcpyAsync(flags, d_sel, (size_t)inliers * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        PGX_HIP(ctx, hipMemcpyAsync(flags, labels, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (count) *count = inliers;
    return PGX_OK;
}

// ---- inlier indices of a scored hypothesis (the `inliers` vector getScore fills, scoring_function_with_compound_model.h:88) ----
// pgx_score returns inlier sets as bit masks; the host's proposal loop needs index lists (refit selections, the final inlier
// list of a proposal) and used to unpack a 10^6-bit row with numpy (unpackbits + nonzero: ~1.5 ms, 100+ times per find6DPoses
// call).  Here the row is expanded to flags and compacted on the device (hipcub select): ascending indices, bit for bit the set.
__global__ __launch_bounds__(kPwBlock) void mask_flags_kernel(const unsigned long long* __restrict__ row, int64_t n, int* __restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i < n) flags[i] = (int)((row[i >> 6] >> (i & 63)) & 1ull);
}

int score_inliers_launch(pgx_ctx* ctx, int row, int32_t* index, int64_t* count)
{
    const int64_t n = ctx->n;
    if (!ctx->have_masks || ctx->M <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_inliers: the last launch produced no masks");
    if (row < 0 || row >= ctx->M) return fail(ctx, PGX_ERR_INVALID, "pgx_score_inliers: row %d of %d", row, ctx->M);
    if (n <= kCompactSmall) {   // one launch, one copy, one synchronisation
        PGX_TRY(compact_small_enqueue<true>(ctx, ctx->masks.as<unsigned long long>() + (size_t)row * (size_t)ctx->words, nullptr, n));
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int* hs = (const int*)ctx->h_res;
        memcpy(index, hs + 1, (size_t)hs[0] * sizeof(int32_t));
        *count = hs[0];
        return PGX_OK;
    }
    size_t temp_bytes = 0;
    hipcub::CountingInputIterator<int> ids(0);
    PGX_HIP(ctx, hipcub::DeviceSelect::Flagged(nullptr, temp_bytes, ids, (int*)nullptr, (int*)nullptr, (int*)nullptr, (int)n, ctx->stream));
    const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
    PGX_TRY(ensure(ctx, ctx->gc_sel, 2 * arr + 256 + temp_bytes));
    int* d_flags = ctx->gc_sel.as<int>();
    int* d_sel = (int*)((char*)ctx->gc_sel.p + arr);
    int* d_num = (int*)((char*)ctx->gc_sel.p + 2 * arr);
    void* d_temp = (void*)((char*)ctx->gc_sel.p + 2 * arr + 256);
    hipLaunchKernelGGL(mask_flags_kernel, dim3((unsigned)((n + kPwBlock - 1) / kPwBlock)), dim3(kPwBlock), 0, ctx->stream,
                       ctx->masks.as<unsigned long long>() + (size_t)row * (size_t)ctx->words, n, d_flags);
    PGX_HIP(ctx, hipGetLastError());
    PGX_HIP(ctx, hipcub::DeviceSelect::Flagged(d_temp, temp_bytes, ids, d_flags, d_sel, d_num, (int)n, ctx->stream));
    int num = 0;
    PGX_TRY(d2h(ctx, &num, d_num, sizeof(int)));
    PGX_TRY(sync_deliver(ctx));
    if (num > 0) {
        PGX_TRY(d2h(ctx, index, d_sel, (size_t)num * sizeof(int32_t)));
        PGX_TRY(sync_deliver(ctx));
    }
    *count = num;
    return PGX_OK;
}

// ---- a9: residual sums ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(kPwBlock) void residual_sum_kernel(const double* __restrict__ pts, int64_t n, ModelArg mdl,
                                                                const int* __restrict__ labels, int label,
                                                                double* __restrict__ partials)
{
    using R = Residual<MT>;
    __shared__ double lds[4];
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    double acc[1] = {0.0};
    if (i < n && labels[i] == label) {
        double pt[R::D];
        load_point<MT>(pts, i, pt);
        acc[0] = R::plain(pt, mdl.v);  // PEARL.h:371 / :390  (unsquared residual)
    }
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// All K labels in one launch (blockIdx.y = label, model k for label k): the same per-block tree and the same final pass as
// the single-label kernel, so sums[k] is bit-identical to residual_sum_launch(model k, k).  PEARL::parameterEstimation
// needs 2K such sums per iteration; one host round trip each made its loop latency-bound.
template <int MT>
__global__ __launch_bounds__(kPwBlock) void residual_sums_kernel(const double* __restrict__ pts, int64_t n,
                                                                 const double* __restrict__ models,
                                                                 const int* __restrict__ labels, int blocks,
                                                                 double* __restrict__ partials)
{
    using R = Residual<MT>;
    __shared__ double lds[4];
    const int label = (int)blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    double acc[1] = {0.0};
    if (i < n && labels[i] == label) {
        double pt[R::D], mdl[R::P];
#pragma unroll
        for (int k = 0; k < R::P; ++k) mdl[k] = models[(int64_t)label * R::P + k];
        load_point<MT>(pts, i, pt);
        acc[0] = R::plain(pt, mdl);
    }
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) partials[(int64_t)label * blocks + blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(kPwBlock) void final_sums_kernel(const double* __restrict__ partials, int count, double* __restrict__ out)
{
    __shared__ double lds[4];
    const double* part = partials + (int64_t)blockIdx.x * count;
    double acc[1] = {0.0};
    for (int b = threadIdx.x; b < count; b += kPwBlock) acc[0] += part[b];
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

int residual_sums_launch(pgx_ctx* ctx, const double* models, int K, double* sums)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sums: points not set");
    if (ctx->labels_n != ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sums: labels not set");
    if (K <= 0 || K > 65535) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sums: K = %d out of range", K);
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    PGX_TRY(ensure(ctx, ctx->red_partials, (size_t)blocks * (size_t)(K > 3 ? K : 3) * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->red_out, (size_t)(K > 8 ? K : 8) * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->kmodels, (size_t)K * ctx->P * sizeof(double)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->kmodels.p, models, (size_t)K * ctx->P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dim3 g((unsigned)blocks, (unsigned)K), b(kPwBlock);
    const double* pts = ctx->pts.as<double>();
    const double* mdl = ctx->kmodels.as<double>();
    const int* lab = ctx->labels.as<int>();
    double* part = ctx->red_partials.as<double>();
    switch (ctx->model_type) {
    case kLine2D: hipLaunchKernelGGL((residual_sums_kernel<kLine2D>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    case kHomography: hipLaunchKernelGGL((residual_sums_kernel<kHomography>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    case kFundamental: hipLaunchKernelGGL((residual_sums_kernel<kFundamental>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    case kPnP: hipLaunchKernelGGL((residual_sums_kernel<kPnP>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    case kVanishingPoint: hipLaunchKernelGGL((residual_sums_kernel<kVanishingPoint>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    case kHomographySym: hipLaunchKernelGGL((residual_sums_kernel<kHomographySym>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, blocks, part); break;
    default: return fail(ctx, PGX_ERR_INVALID, "bad model type");
    }
    PGX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(final_sums_kernel, dim3((unsigned)K), dim3(kPwBlock), 0, ctx->stream, part, blocks, ctx->red_out.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, sums, ctx->red_out.p, (size_t)K * sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

int residual_sum_launch(pgx_ctx* ctx, const double* model, int label, double* sum)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sum: points not set");
    if (ctx->labels_n != ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sum: labels not set");
    ModelArg mdl;
    for (int k = 0; k < 18; ++k) mdl.v[k] = k < ctx->P ? model[k] : 0.0;
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    PGX_TRY(ensure(ctx, ctx->red_partials, (size_t)blocks * 3 * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->red_out, 8 * sizeof(double)));
    dim3 g((unsigned)blocks), b(kPwBlock);
    const double* pts = ctx->pts.as<double>();
    const int* lab = ctx->labels.as<int>();
    double* part = ctx->red_partials.as<double>();
    switch (ctx->model_type) {
    case kLine2D: hipLaunchKernelGGL((residual_sum_kernel<kLine2D>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    case kHomography: hipLaunchKernelGGL((residual_sum_kernel<kHomography>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    case kFundamental: hipLaunchKernelGGL((residual_sum_kernel<kFundamental>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    case kPnP: hipLaunchKernelGGL((residual_sum_kernel<kPnP>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    case kVanishingPoint: hipLaunchKernelGGL((residual_sum_kernel<kVanishingPoint>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    case kHomographySym: hipLaunchKernelGGL((residual_sum_kernel<kHomographySym>), g, b, 0, ctx->stream, pts, ctx->n, mdl, lab, label, part); break;
    default: return fail(ctx, PGX_ERR_INVALID, "bad model type");
    }
    PGX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL((final_sum_kernel<1>), dim3(1), dim3(kPwBlock), 0, ctx->stream, part, blocks,
                       ctx->red_out.as<double>());
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, sum, ctx->red_out.p, sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

// ---- U-14: support of a fundamental matrix under the symmetric epipolar distance --------------------------------------------
// The validity stage of the F estimator (pyprogressivex/_estimators.py FundamentalEstimator.valid_best: a so-far-best F must keep
// at least half of its Sampson inliers under the symmetric epipolar distance, thresholds T2 and S2) counted over all points in one
// pass.  The Sampson value is the scorer's own (Residual<kFundamental>::squared: the same inlier set, strict <); the symmetric
// distance r^2 (1 / |F x1|_12^2 + 1 / |F^T x2|_12^2) with the Sampson functor's intermediate values, left to right.
__global__ __launch_bounds__(kPwBlock) void epipolar_support_kernel(const double* __restrict__ pts, int64_t n, ModelArg mdl, double T2,
                                                                     double S2, unsigned long long* __restrict__ out /*[2]*/)
{
    using R = Residual<kFundamental>;
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    bool inl = false, sup = false;
    if (i < n) {
        double pt[R::D];
        load_point<kFundamental>(pts, i, pt);
        const double* f = mdl.v;
        const double x1 = pt[0], y1 = pt[1], x2 = pt[2], y2 = pt[3];
        const double rxc = f[0] * x2 + f[3] * y2 + f[6];
        const double ryc = f[1] * x2 + f[4] * y2 + f[7];
        const double rwc = f[2] * x2 + f[5] * y2 + f[8];
        const double r = x1 * rxc + y1 * ryc + rwc;
        const double rx = f[0] * x1 + f[1] * y1 + f[2];
        const double ry = f[3] * x1 + f[4] * y1 + f[5];
        const double sym = (r * r) * (1.0 / (rx * rx + ry * ry) + 1.0 / (rxc * rxc + ryc * ryc));
        inl = R::squared(pt, mdl.v) < T2;       // false on NaN
        sup = inl && sym < S2;
    }
    const unsigned long long mi = __ballot(inl), ms = __ballot(sup);
    if ((threadIdx.x & 63) == 0) {
        if (mi) atomicAdd(&s_cnt[0], (unsigned)__popcll(mi));
        if (ms) atomicAdd(&s_cnt[1], (unsigned)__popcll(ms));
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

int epipolar_support_launch(pgx_ctx* ctx, const double* F, double T2, double S2, int64_t counts[2])
{
    if (ctx->n <= 0 || ctx->model_type != kFundamental)
        return fail(ctx, PGX_ERR_INVALID, "pgx_epipolar_support: needs the points of a fundamental-matrix problem");
    ModelArg mdl;
    for (int k = 0; k < 18; ++k) mdl.v[k] = k < 9 ? F[k] : 0.0;
    PGX_TRY(ensure(ctx, ctx->red_out, 8 * sizeof(double)));
    unsigned long long* out = (unsigned long long*)ctx->red_out.p;
    PGX_HIP(ctx, hipMemsetAsync(out, 0, 16, ctx->stream));
    const int blocks = (int)((ctx->n + kPwBlock - 1) / kPwBlock);
    hipLaunchKernelGGL(epipolar_support_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n, mdl, T2, S2, out);
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, counts, out, 16));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

// ---- a9: bucket by label (PEARL.h:342-352) -----------------------------------------------------------------------
// Labels >= L-1 fall into the last (outlier) bucket, as `label < instance_number` does at PEARL.h:348.
constexpr int kMaxBucketLabels = 64;

__global__ __launch_bounds__(kPwBlock) void bucket_count_kernel(const int* __restrict__ labels, int64_t n, int L,
                                                                unsigned* __restrict__ block_counts /*[blocks][L]*/,
                                                                unsigned long long* __restrict__ totals /*[L] or nullptr*/)
{
    __shared__ unsigned hist[kMaxBucketLabels];
    if (threadIdx.x < kMaxBucketLabels) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    int l = -1;
    if (i < n) { l = labels[i]; if (l >= L - 1) l = L - 1; }
    // one LDS atomic per (wave, label present): ballot per label value, lowest lane of each group adds the popcount
    for (int k = 0; k < L; ++k) {
        const unsigned long long b = __ballot(l == k);
        if (b != 0 && (threadIdx.x & 63) == (unsigned)__ffsll((long long)b) - 1)
            atomicAdd(&hist[k], (unsigned)__popcll(b));
    }
    __syncthreads();
    if ((int)threadIdx.x < L) {
        // bucket SIZES only (PEARL's refits select the members on the device by label): integer atomics, no scan over the blocks
        if (totals != nullptr) { if (hist[threadIdx.x]) atomicAdd(&totals[threadIdx.x], (unsigned long long)hist[threadIdx.x]); }
        else block_counts[(int64_t)blockIdx.x * L + threadIdx.x] = hist[threadIdx.x];
    }
}

// exclusive scan over blocks for every label (one thread per label; blocks <= a few thousand) + label starts
__global__ void bucket_scan_kernel(unsigned* __restrict__ block_counts, int blocks, int L,
                                   long long* __restrict__ counts, long long* __restrict__ starts)
{
    const int k = threadIdx.x;
    if (k < L) {
        long long run = 0;
        for (int b = 0; b < blocks; ++b) {
            const unsigned c = block_counts[(int64_t)b * L + k];
            block_counts[(int64_t)b * L + k] = (unsigned)run;  // offset of block b inside bucket k (fits: n < 2^31)
            run += c;
        }
        counts[k] = run;
    }
    __syncthreads();
    if (k == 0) {
        long long s = 0;
        for (int j = 0; j < L; ++j) { starts[j] = s; s += counts[j]; }
    }
}

__global__ __launch_bounds__(kPwBlock) void bucket_scatter_kernel(const int* __restrict__ labels, int64_t n, int L,
                                                                  const unsigned* __restrict__ block_offsets,
                                                                  const long long* __restrict__ starts,
                                                                  int* __restrict__ order)
{
    __shared__ unsigned wave_cnt[4][kMaxBucketLabels];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    int l = -1;
    if (i < n) { l = labels[i]; if (l >= L - 1) l = L - 1; }
    unsigned rank_in_wave = 0;
    for (int k = 0; k < L; ++k) {
        const unsigned long long b = __ballot(l == k);
        if (l == k) rank_in_wave = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave][k] = (unsigned)__popcll(b);
    }
    __syncthreads();
    if (l >= 0) {
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w][l];
        const long long pos = starts[l] + block_offsets[(int64_t)blockIdx.x * L + l] + before + rank_in_wave;
        order[pos] = (int)i;
    }
}

int bucket_launch(pgx_ctx* ctx, int L, int64_t* counts, int32_t* order)
{
    if (ctx->labels_n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_bucket: labels not set");
    if (L < 1 || L > kMaxBucketLabels) return fail(ctx, PGX_ERR_INVALID, "pgx_bucket: L must be in [1,%d]", kMaxBucketLabels);
    const int64_t n = ctx->labels_n;
    const int blocks = (int)((n + kPwBlock - 1) / kPwBlock);
    const size_t bc_bytes = (size_t)blocks * L * sizeof(unsigned);
    const size_t need = bc_bytes + 2 * (size_t)L * sizeof(long long) + 64 + (order ? (size_t)n * sizeof(int) : 0);
    PGX_TRY(ensure(ctx, ctx->scratch, need));
    char* base = (char*)ctx->scratch.p;
    unsigned* bc = (unsigned*)base;
    size_t o = (bc_bytes + 15) & ~(size_t)15;
    long long* d_counts = (long long*)(base + o); o += (size_t)L * sizeof(long long);
    long long* d_starts = (long long*)(base + o); o += (size_t)L * sizeof(long long);
    o = (o + 15) & ~(size_t)15;
    int* d_order = (int*)(base + o);
    if (!order) {   // (the one-thread-per-label scan over the blocks was 100 us per PEARL iteration at 2e5 points)
        PGX_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)L * sizeof(long long), ctx->stream));
        hipLaunchKernelGGL(bucket_count_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream,
                           ctx->labels.as<int>(), n, L, bc, (unsigned long long*)d_counts);
    } else {
        hipLaunchKernelGGL(bucket_count_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream,
                           ctx->labels.as<int>(), n, L, bc, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(64), 0, ctx->stream, bc, blocks, L, d_counts, d_starts);
    }
    if (order)
        hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream,
                           ctx->labels.as<int>(), n, L, bc, d_starts, d_order);
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, counts, d_counts, (size_t)L * sizeof(long long)));
    if (order)
        PGX_TRY(d2h(ctx, order, d_order, (size_t)n * sizeof(int)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

// ---- energy (U-5 compute_energy): exact int64 ----------------------------------------------------------------------
__global__ __launch_bounds__(kPwBlock) void energy_kernel(const long long* __restrict__ dq, int64_t n, int L,
                                                          const int* __restrict__ labels,
                                                          const int* __restrict__ off, const int* __restrict__ idx,
                                                          const int* __restrict__ mult, long long lambda_q,
                                                          unsigned long long* __restrict__ energy,
                                                          unsigned* __restrict__ used /*[L]*/)
{
    __shared__ unsigned long long lds[4];
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    unsigned long long e = 0;
    if (i < n) {
        const int li = labels[i];
        e = (unsigned long long)dq[(int64_t)li * n + i];
        used[li] = 1;  // benign race: every writer stores 1
        if (off != nullptr && lambda_q > 0)
            for (int a = off[i]; a < off[i + 1]; ++a) {
                const int j = idx[a];
                if (j < i && labels[j] != li) e += (unsigned long long)(lambda_q * (long long)mult[a]);
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(energy, lds[0] + lds[1] + lds[2] + lds[3]);  // integer: order-independent
}

int energy_launch(pgx_ctx* ctx, int64_t lambda_q, int64_t h_q, int64_t* energy_q)
{
    const int64_t n = ctx->dq_n;
    const int L = ctx->L;
    if (n <= 0 || L <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_energy: unary table not set");
    if (ctx->labels_n != n) return fail(ctx, PGX_ERR_INVALID, "pgx_energy: labels not set");
    if (ctx->labels_max >= L) return fail(ctx, PGX_ERR_INVALID, "pgx_energy: label %d out of range (the unary table has %d labels)", ctx->labels_max, L);
    const bool pair = lambda_q > 0 && ctx->gn == n;
    if (lambda_q > 0 && ctx->gn != n) return fail(ctx, PGX_ERR_INVALID, "pgx_energy: lambda > 0 but no graph set");
    PGX_TRY(ensure(ctx, ctx->scratch, 16 + (size_t)L * sizeof(unsigned)));
    unsigned long long* d_e = (unsigned long long*)ctx->scratch.p;
    unsigned* d_used = (unsigned*)((char*)ctx->scratch.p + 16);
    PGX_HIP(ctx, hipMemsetAsync(ctx->scratch.p, 0, 16 + (size_t)L * sizeof(unsigned), ctx->stream));
    const int blocks = (int)((n + kPwBlock - 1) / kPwBlock);
    hipLaunchKernelGGL(energy_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, ctx->dq.as<long long>(),
                       n, L, ctx->labels.as<int>(), pair ? ctx->goff.as<int>() : (const int*)nullptr,
                       ctx->gidx.as<int>(), ctx->gmult.as<int>(), (long long)lambda_q, d_e, d_used);
    PGX_HIP(ctx, hipGetLastError());
    std::vector<unsigned char> host(16 + (size_t)L * sizeof(unsigned));
    PGX_TRY(d2h(ctx, host.data(), ctx->scratch.p, host.size()));
    PGX_TRY(sync_deliver(ctx));
    int64_t e = (int64_t)(*(unsigned long long*)host.data());
    const unsigned* used = (const unsigned*)(host.data() + 16);
    for (int l = 0; l < L; ++l) if (used[l]) e += h_q;
    *energy_q = e;
    return PGX_OK;
}

// ---- U-8: GCO-v3's labelling of an energy without smooth costs -------------------------------------------------------
// PEARL sets no smooth cost and no neighbours when spatial_coherence_weight == 0 (PEARL.h:523-536) and then calls
// expansion() (:550-551), whose first step in GCO-v3 is solveSpecialCases(): "data costs only" -> every site takes its
// cheapest label; "data costs + per-label costs" -> solveGreedy(), the greedy uncapacitated-facility-location heuristic
// (restated from the published algorithm [UPSTREAM-MEMORY]; the CPU restatement under oracle/ is the checker).  One round = one reduction over all sites for
// every label not opened yet (integer sums: exact and order-free, so the host's choice is the oracle's) + one apply pass;
// at most L rounds.  HBM-bound: a round reads (#closed labels + 1) * n * 8 B of the label-major table.
constexpr long long kGreedyBig = 1ll << 35;  // "unassigned": above every unary cost (dq_max is checked), n * BIG < 2^62

__global__ __launch_bounds__(kPwBlock) void greedy_init_kernel(long long* __restrict__ e, int* __restrict__ labels, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i < n) { e[i] = kGreedyBig; labels[i] = 0; }
}

// Rounds without a host round trip: the choice of the label to open is made on the device (greedy_decide_kernel, one thread,
// the same strict "<" over the labels in index order as the oracle), so the host enqueues all L rounds at once and reads the
// result once - a round used to be three launches, a copy back and a synchronisation (~40 us; findTwoViewMotions at C3 runs
// 244 such labellings).  state: [0] mask of opened labels, [1] label opened in this round, [2] count, [3] done.
constexpr int kGreedyBlocks = 256;   // grid-stride: every workgroup ends with one atomic per label on L addresses

__global__ __launch_bounds__(kPwBlock) void greedy_delta_kernel(const long long* __restrict__ dq, int64_t n, int L,
                                                               const long long* __restrict__ e,
                                                               const unsigned long long* __restrict__ state,
                                                               unsigned long long* __restrict__ delta /*[L]*/)
{
    if (state[3]) return;
    const unsigned long long open_mask = state[0];
    __shared__ long long lds[kPwBlock / 64];
    for (int l = 0; l < L; ++l) {
        if ((open_mask >> l) & 1ull) continue;  // uniform
        long long d = 0;
        for (int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kPwBlock) {
            const long long t = dq[(int64_t)l * n + i] - e[i];
            if (t < 0) d += t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        __syncthreads();  // the previous label's partials have been read
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long s = 0;
            for (int w = 0; w < kPwBlock / 64; ++w) s += lds[w];
            if (s != 0) atomicAdd(&delta[l], (unsigned long long)s);  // two's complement: integer sums are exact
        }
    }
}

__global__ void greedy_decide_kernel(int L, long long h_q, unsigned long long* __restrict__ state, unsigned long long* __restrict__ delta)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || state[3]) return;
    const unsigned long long open_mask = state[0];
    int best = -1;
    long long best_delta = 0;
    for (int l = 0; l < L; ++l) {
        if (!((open_mask >> l) & 1ull)) {
            const long long dl = (long long)delta[l] + h_q;
            if (dl < best_delta) { best_delta = dl; best = l; }
        }
        delta[l] = 0;   // for the next round
    }
    if (best < 0) { state[3] = 1; return; }
    state[0] = open_mask | (1ull << best);
    state[1] = (unsigned long long)best;
    state[2] += 1;
}

__global__ __launch_bounds__(kPwBlock) void greedy_apply_kernel(const long long* __restrict__ dq, int64_t n,
                                                               const unsigned long long* __restrict__ state,
                                                               long long* __restrict__ e, int* __restrict__ labels)
{
    if (state[3]) return;
    const int alpha = (int)state[1];
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    const long long d = dq[(int64_t)alpha * n + i];
    if (d < e[i]) { e[i] = d; labels[i] = alpha; }
}

__global__ __launch_bounds__(kPwBlock) void greedy_argmin_kernel(const long long* __restrict__ dq, int64_t n, int L,
                                                                int* __restrict__ labels)
{
    const int64_t i = (int64_t)blockIdx.x * kPwBlock + threadIdx.x;
    if (i >= n) return;
    int best = 0;
    long long bv = dq[i];
    for (int l = 1; l < L; ++l) {
        const long long v = dq[(int64_t)l * n + i];
        if (v < bv) { bv = v; best = l; }  // first minimum
    }
    labels[i] = best;
}

int greedy_labeling_launch(pgx_ctx* ctx, int64_t h_q, int64_t* energy_q, int* opened)
{
    const int64_t n = ctx->dq_n;
    const int L = ctx->L;
    if (n <= 0 || L <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_greedy_labeling: unary table not set");
    if (L > 64) return fail(ctx, PGX_ERR_INVALID, "pgx_greedy_labeling: at most 64 labels (got %d)", L);
    if (ctx->dq_max >= kGreedyBig || (double)n * (double)kGreedyBig >= 4.6e18)
        return fail(ctx, PGX_ERR_RANGE, "pgx_greedy_labeling: fixed-point range exceeded (max unary %lld, n %lld)",
                    (long long)ctx->dq_max, (long long)n);
    PGX_TRY(ensure(ctx, ctx->labels, (size_t)n * sizeof(int32_t)));
    ctx->labels_n = n;
    ctx->labels_max = L - 1;
    ctx->labels_all_zero = 0;   // (the labelling is no longer the uploaded all-zero one: no first-cycle memo for the next pgx_expansion)
    ctx->last_done.valid = 0;
    ctx->labels_version += 1;
    const int blocks = (int)((n + kPwBlock - 1) / kPwBlock);
    const long long* dq = ctx->dq.as<long long>();
    int* labels = ctx->labels.as<int>();
    int count = 0;
    if (h_q <= 0) {
        hipLaunchKernelGGL(greedy_argmin_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, dq, n, L, labels);
        PGX_HIP(ctx, hipGetLastError());
        count = -1;  // counted from the energy pass below
    } else {
        PGX_TRY(ensure(ctx, ctx->gc, (size_t)n * sizeof(long long) + 72 * sizeof(long long)));  // e[n] | delta[64] | state[4]
        long long* e = ctx->gc.as<long long>();
        unsigned long long* d_delta = (unsigned long long*)(e + n);
        unsigned long long* d_state = d_delta + 64;
        hipLaunchKernelGGL(greedy_init_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, e, labels, n);
        PGX_HIP(ctx, hipGetLastError());
        PGX_HIP(ctx, hipMemsetAsync(d_delta, 0, 72 * sizeof(long long), ctx->stream));
        const unsigned gb = (unsigned)(blocks < kGreedyBlocks ? blocks : kGreedyBlocks);
        for (int round = 0; round < L; ++round) {   // all rounds enqueued at once: finished rounds leave at their first instruction
            hipLaunchKernelGGL(greedy_delta_kernel, dim3(gb), dim3(kPwBlock), 0, ctx->stream, dq, n, L, e, d_state, d_delta);
            hipLaunchKernelGGL(greedy_decide_kernel, dim3(1), dim3(64), 0, ctx->stream, L, (long long)h_q, d_state, d_delta);
            hipLaunchKernelGGL(greedy_apply_kernel, dim3((unsigned)blocks), dim3(kPwBlock), 0, ctx->stream, dq, n, d_state, e, labels);
        }
        PGX_HIP(ctx, hipGetLastError());
        unsigned long long h_state[4];
        PGX_TRY(d2h(ctx, h_state, d_state, sizeof(h_state)));
        PGX_TRY(sync_deliver(ctx));
        count = (int)h_state[2];
    }
    int64_t eq = 0;
    PGX_TRY(energy_launch(ctx, 0, h_q, &eq));
    if (count < 0) {  // argmin path: labels in use = (energy's label-cost share is zero) count them on the host from a bucket pass
        std::vector<int64_t> cnt((size_t)L, 0);
        PGX_TRY(bucket_launch(ctx, L, cnt.data(), nullptr));
        count = 0;
        for (int l = 0; l < L; ++l) if (cnt[(size_t)l] > 0) ++count;
    }
    if (energy_q) *energy_q = eq;
    if (opened) *opened = count;
    return PGX_OK;
}

}  // namespace pgx
