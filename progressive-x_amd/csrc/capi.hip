// capi.hip — the C ABI of libpgx.so (declared in include/pgx.h): context management, host<->device marshalling and
// the thin wrappers around the kernel launchers.  No C++ type, exception or PyTorch object crosses this boundary.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include <cmath>

#include "pgx_internal.h"

namespace pgx {

static thread_local std::string g_err;

int fail(pgx_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

int ensure(pgx_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return PGX_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8;  // a little slack so that growing M/K does not reallocate every call
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, PGX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return PGX_OK;
}

void release(DevBuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

constexpr size_t kStageRing = 256 << 10, kStageMax = 32 << 10;

int d2h(pgx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return PGX_OK;
    if (bytes > kStageMax || ctx->h_rb_used + bytes > kStageRing) {
        PGX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return PGX_OK;
    }
    if (!ctx->h_rb) PGX_HIP(ctx, hipHostMalloc(&ctx->h_rb, kStageRing, hipHostMallocDefault));
    const size_t off = ctx->h_rb_used;
    ctx->h_rb_used += (bytes + 63) & ~(size_t)63;
    PGX_HIP(ctx, hipMemcpyAsync((char*)ctx->h_rb + off, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ctx->rb.push_back(StagedCopy{dst, off, bytes});
    return PGX_OK;
}

int sync_deliver(pgx_ctx* ctx)
{
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess)
        for (const StagedCopy& c : ctx->rb) memcpy(c.dst, (const char*)ctx->h_rb + c.off, c.bytes);
    ctx->rb.clear();
    ctx->h_rb_used = 0;
    if (e != hipSuccess) return fail(ctx, PGX_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    return PGX_OK;
}

int host_staging(pgx_ctx* ctx, size_t bytes, void** p)
{
    if (ctx->h_res_cap < bytes) {
        if (ctx->h_res) (void)hipHostFree(ctx->h_res);
        ctx->h_res = nullptr; ctx->h_res_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&ctx->h_res, bytes * 2, hipHostMallocDefault));
        ctx->h_res_cap = bytes * 2;
    }
    *p = ctx->h_res;
    return PGX_OK;
}

}  // namespace pgx

using namespace pgx;

extern "C" {

int pgx_version(void) { return 100; }

const char* pgx_global_error(void) { return g_err.c_str(); }

int pgx_device_count(int* count)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        if (count) *count = 0;
        return fail(nullptr, PGX_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    if (count) *count = c;
    return PGX_OK;
}

int pgx_model_dims(int model_type, int* point_dim, int* param_dim)
{
    return model_dims(model_type, point_dim, param_dim) == 0 ? PGX_OK : PGX_ERR_INVALID;
}

int pgx_create(int device_id, pgx_ctx** out)
{
    if (!out) return fail(nullptr, PGX_ERR_INVALID, "pgx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, PGX_ERR_HIP, "pgx_create: no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, PGX_ERR_INVALID, "pgx_create: device %d out of range [0,%d)", device_id, count);
    pgx_ctx* ctx = new pgx_ctx();
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
        int r = fail(nullptr, PGX_ERR_HIP, "pgx_create: HIP init failed: %s", hipGetErrorString(e));
        delete ctx;
        return r;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    const char* nf = std::getenv("PGX_NO_FILTER");
    ctx->filter_enabled = (nf && nf[0] == '1') ? 0 : ((nf && nf[0] == '2') ? 2 : 1);
    if (const char* ns = std::getenv("PGX_NO_SORT")) ctx->score_sort = (ns[0] == '1') ? 0 : 1;
    if (const char* b = std::getenv("PGX_SP_KD")) { const int v = std::atoi(b); ctx->sp_kd = v < 0 ? 0 : (v > 2 ? 2 : v); }
    if (const char* b = std::getenv("PGX_SETPOINTS_HOST")) ctx->setpoints_host = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_GC_FLIP")) ctx->gc_flip = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_SCORE_MIRROR")) ctx->score_mirror = std::atoi(b) != 0;
    if (const char* b = std::getenv("PGX_SCORE_NO_CULL")) ctx->score_cull = std::atoi(b) ? 0 : 1;
    if (const char* b = std::getenv("PGX_NO_GROUP")) ctx->group_filter = std::atoi(b) ? 0 : 1;
    if (const char* b = std::getenv("PGX_VERIFY")) ctx->verify = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_TILE")) ctx->mf_tile = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_TILE_BATCH")) ctx->mf_tile_batch = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_REGION")) ctx->mf_region = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_MEMO")) ctx->mf_memo = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_DONE_VERIFY")) ctx->mf_done_verify = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_XCD")) ctx->mf_xcd = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_XCD_SEARCH")) ctx->mf_xcd_search = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_MF_XCD_MIN_DEPTH")) ctx->mf_xcd_min_depth = std::atoi(b);
    if (const char* b = std::getenv("PGX_MF_XCD_MAXN")) ctx->mf_xcd_max_n = std::atoll(b);
    if (const char* b = std::getenv("PGX_MF_SWEEPS")) ctx->mf_sweeps = std::atoi(b);
    if (const char* b = std::getenv("PGX_TILE_MINI")) ctx->tile_mini = std::atoi(b) ? 1 : 0;
    if (const char* b = std::getenv("PGX_TILE_EXPANSION_MAX")) { const int v = std::atoi(b); ctx->tile_expansion_max = v < 0 ? 0 : v; }
    if (const char* b = std::getenv("PGX_TILE_MINI_SWEEPS")) { const int v = std::atoi(b); ctx->tile_mini_sweeps = v < 1 ? 1 : v; }
    if (const char* b = std::getenv("PGX_MF_DEBUG")) ctx->tile_debug = std::atoi(b);
    *out = ctx;
    return PGX_OK;
}

void pgx_destroy(pgx_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    comm_free(ctx);
    maxflow_free(ctx);
    tile_free(ctx);
    release(ctx->gorder);
    DevBuf* bufs[] = {&ctx->pts, &ctx->comp, &ctx->pmax, &ctx->pts32, &ctx->perm, &ctx->models, &ctx->pcnt, &ctx->pval, &ctx->psh, &ctx->counts,
                      &ctx->masks, &ctx->g_counts, &ctx->g_values, &ctx->g_shared,
                      &ctx->red_partials, &ctx->red_out, &ctx->dq, &ctx->kmodels, &ctx->labels, &ctx->goff,
                      &ctx->gidx, &ctx->gmult, &ctx->grev, &ctx->scratch, &ctx->fit_scratch, &ctx->pts_s, &ctx->pts32_s,
                      &ctx->pmax_s, &ctx->comp_s, &ctx->pperm, &ctx->gbounds, &ctx->masks_s, &ctx->cull_lists, &ctx->cull_counts, &ctx->gc, &ctx->gc_sel,
                      &ctx->weights, &ctx->stats_buf, &ctx->pts_g, &ctx->p32_g, &ctx->weights_scratch, &ctx->memo.snaps, &ctx->prosac_tops};
    for (DevBuf* b : bufs) release(*b);
    for (DevBuf& b : ctx->slots) release(b);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    if (ctx->h_rb) (void)hipHostFree(ctx->h_rb);
    if (ctx->h_mirror) (void)hipHostFree(ctx->h_mirror);
    if (ctx->h_samples) (void)hipHostFree(ctx->h_samples);
    if (ctx->h_models) (void)hipHostFree(ctx->h_models);
    if (ctx->ev_models) (void)hipEventDestroy(ctx->ev_models);
    if (ctx->ev_samples) (void)hipEventDestroy(ctx->ev_samples);
    for (int k = 0; k < 5; ++k) if (ctx->kev[k]) (void)hipEventDestroy(ctx->kev[k]);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* pgx_last_error(const pgx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

#define CTX_GUARD(ctx)                                                            \
    do {                                                                          \
        if (!(ctx)) return fail(nullptr, PGX_ERR_INVALID, "ctx is NULL");        \
        hipError_t e_ = hipSetDevice((ctx)->device);                              \
        if (e_ != hipSuccess) return fail(ctx, PGX_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e_)); \
        (ctx)->rb.clear();                                                        \
        (ctx)->h_rb_used = 0;                                                     \
    } while (0)

int pgx_sync(pgx_ctx* ctx)
{
    CTX_GUARD(ctx);
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

int pgx_timer_start(pgx_ctx* ctx)
{
    CTX_GUARD(ctx);
    PGX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return PGX_OK;
}

int pgx_timer_stop(pgx_ctx* ctx, float* ms)
{
    CTX_GUARD(ctx);
    PGX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PGX_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float t = 0.f;
    PGX_HIP(ctx, hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    if (ms) *ms = t;
    return PGX_OK;
}

int pgx_timer_mark(pgx_ctx* ctx)
{
    CTX_GUARD(ctx);
    PGX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    return PGX_OK;
}

int pgx_timer_elapsed(pgx_ctx* ctx, float* ms)
{
    CTX_GUARD(ctx);
    PGX_HIP(ctx, hipEventSynchronize(ctx->ev1));  // returns at once when the stream was synchronised after the mark
    float t = 0.f;
    PGX_HIP(ctx, hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    if (ms) *ms = t;
    return PGX_OK;
}

int pgx_device_info(pgx_ctx* ctx, char* name, int name_len, int* cu_count, int64_t* hbm_bytes)
{
    CTX_GUARD(ctx);
    hipDeviceProp_t prop;
    PGX_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        // some boxes report an empty marketing name (the driver's bench box did: "device": " (gfx950:...)"): say what is known
        char fallback[64];
        snprintf(fallback, sizeof fallback, "AMD GPU, %d CUs", prop.multiProcessorCount);
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name[0] ? prop.name : fallback, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PGX_OK;
}

/* ---- resident data ---------------------------------------------------------------------------------------- */
// The host version of the preprocessing (round 1; PGX_SETPOINTS_HOST=1 keeps it for A/B and as the cross-check of
// setpoints.hip: both produce the same sorted order and the same group rows).
static int set_points_host(pgx_ctx* ctx, int model_type, const double* points, int64_t n, int d)
{
    PGX_TRY(ensure(ctx, ctx->pts, (size_t)n * d * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->comp, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pmax, (size_t)n * sizeof(double)));
    // score-filter scales (score.hip Filter<>): Umax = largest |observed image coordinate|, pmax[i] = max(|the
    // coordinates the projective map multiplies|, 1).  Model types without a filter get pmax = 1, Umax = 0.
    std::vector<double> pmax((size_t)n, 1.0);
    std::vector<float> p32((size_t)n * 8, 0.0f);
    double umax = 0.0;
    int obs0 = -1, obs1 = -1, in0 = 0, in1 = -1;
    if (model_type == kPnP) { obs0 = 0; obs1 = 1; in0 = 2; in1 = 4; }
    else if (model_type == kHomography || model_type == kHomographySym) { obs0 = 2; obs1 = 3; in0 = 0; in1 = 3; }
    if (obs0 >= 0)
        for (int64_t i = 0; i < n; ++i) {
            const double* r = points + i * d;
            const double a = std::fabs(r[obs0]), b = std::fabs(r[obs1]);
            if (!(a <= umax)) umax = a;   // NaN propagates into umax => filter disabled
            if (!(b <= umax)) umax = b;
            double pm = 1.0;
            for (int k = in0; k <= in1; ++k) { const double v = std::fabs(r[k]); if (!(v <= pm)) pm = v; }
            pmax[(size_t)i] = pm;
            float* q = p32.data() + (size_t)i * 8;   // f32 row of the pre-filter: coords, then the scale rounded up
            for (int k = 0; k < d; ++k) q[k] = (float)r[k];
            q[5] = (float)(pm * 1.000001);
        }
    ctx->umax = umax;
    {
        double fs = 1.0;
        for (int64_t i = 0; i < n * d; ++i) { const double a = std::fabs(points[i]); if (a > fs) fs = a; }
        ctx->fscale = fs;   // NaN coordinates leave it at what the finite ones give; the solvers then produce NaN models
    }
    PGX_TRY(ensure(ctx, ctx->pts32, (size_t)n * 8 * sizeof(float)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pts32.p, p32.data(), (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pts.p, points, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pmax.p, pmax.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemsetAsync(ctx->comp.p, 0, (size_t)n * sizeof(double), ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->point_sort = 0;
    ctx->comp_dirty = 1;
    if (obs0 >= 0 && ctx->group_filter && ctx->filter_enabled == 1 && std::isfinite(umax))
        PGX_TRY(score_sort_points(ctx, points, p32.data(), pmax.data()));
    return PGX_OK;
}

int pgx_set_points(pgx_ctx* ctx, int model_type, const double* points, int64_t n)
{
    CTX_GUARD(ctx);
    int d = 0, p = 0;
    if (model_dims(model_type, &d, &p) != 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_points: bad model type %d", model_type);
    if (!points || n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_points: empty input");
    if (n >= ((int64_t)1 << 31)) return fail(ctx, PGX_ERR_INVALID, "pgx_set_points: n must be < 2^31");
    ctx->model_type = model_type; ctx->D = d; ctx->P = p; ctx->n = n;
    ctx->points_version += 1;
    ctx->labels_all_zero = 0;
    ctx->M = 0; ctx->last_acc = nullptr; ctx->dq_n = 0; ctx->L = 0; ctx->labels_n = 0;
    ctx->weights_n = 0;  // weights belong to a point set
    ctx->score_global_n = 0;   // the fixed-point scale of a sharded job belongs to that job's point set
    if (ctx->memo.snaps.p) { (void)hipStreamSynchronize(ctx->stream); release(ctx->memo.snaps); ctx->memo.cap = 0; ctx->memo.valid = 0; ctx->memo.n = 0; }   // first-cycle snapshots of the old point set
    // upload + every derived copy on the device (setpoints.hip): filter scales, f32 rows, Morton order, group bounds
    const int rc = !ctx->setpoints_host ? set_points_device(ctx, model_type, points, n) : set_points_host(ctx, model_type, points, n, d);
    if (rc != PGX_OK) {   // half-built copies (an allocation failed midway): no resident problem, every later call fails cleanly
        ctx->n = 0; ctx->model_type = -1; ctx->D = 0; ctx->P = 0; ctx->point_sort = 0; ctx->mirror_valid = 0;
        return rc;
    }
    for (DevBuf& b : ctx->slots) release(b);
    ctx->slots.clear();
    return PGX_OK;
}

int pgx_set_compound(pgx_ctx* ctx, const double* compound)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_compound: points not set");
    if (compound)
        PGX_HIP(ctx, hipMemcpyAsync(ctx->comp.p, compound, (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    else
        PGX_HIP(ctx, hipMemsetAsync(ctx->comp.p, 0, (size_t)ctx->n * sizeof(double), ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->comp_dirty = 1;
    return PGX_OK;
}

int pgx_get_compound(pgx_ctx* ctx, double* compound)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0 || !compound) return fail(ctx, PGX_ERR_INVALID, "pgx_get_compound: points not set");
    PGX_TRY(d2h(ctx, compound, ctx->comp.p, (size_t)ctx->n * sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

/* ---- scoring ---------------------------------------------------------------------------------------------- */
// Locality key of a hypothesis: where it sends a probe point, quantised to a 16+16 bit Morton code.  A wave scores 64
// hypotheses against the same point and pays for the exact path whenever ANY of them has a candidate there (the union
// over lanes); hypotheses that explain the same image region share their candidates, so ordering the batch by this key
// shrinks the union (metric batch: 26 % -> ~7 % of wave-steps on the exact path).  Pure scheduling: results are
// written back in the caller's order (score_reduce_kernel / mask rows use `perm`).
static uint32_t spread16(uint32_t x)
{
    x &= 0xffff;
    x = (x | (x << 8)) & 0x00ff00ff;
    x = (x | (x << 4)) & 0x0f0f0f0f;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    return x;
}

static bool locality_keys(const pgx_ctx* ctx, const double* models, int M, std::vector<uint64_t>& keys)
{
    const int P = ctx->P;
    std::vector<double> kx((size_t)M), ky((size_t)M);
    for (int m = 0; m < M; ++m) {
        const double* q = models + (size_t)m * P;
        double x, y;
        if (ctx->model_type == kPnP) {            // projection of the object origin
            x = q[3] / q[11]; y = q[7] / q[11];
        } else if (ctx->model_type == kHomography || ctx->model_type == kHomographySym) {  // image of the probe (0,0)
            x = q[2] / q[8]; y = q[5] / q[8];
        } else return false;
        kx[(size_t)m] = x; ky[(size_t)m] = y;
    }
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int m = 0; m < M; ++m) {
        if (std::isfinite(kx[m])) { lo[0] = std::min(lo[0], kx[m]); hi[0] = std::max(hi[0], kx[m]); }
        if (std::isfinite(ky[m])) { lo[1] = std::min(lo[1], ky[m]); hi[1] = std::max(hi[1], ky[m]); }
    }
    keys.resize((size_t)M);
    for (int m = 0; m < M; ++m) {
        if (!std::isfinite(kx[m]) || !std::isfinite(ky[m]) || !(hi[0] > lo[0]) || !(hi[1] > lo[1])) {
            keys[(size_t)m] = 1ull << 32;  // degenerate hypotheses last (above every 32-bit Morton code)
            continue;
        }
        const uint32_t qx = (uint32_t)((kx[m] - lo[0]) / (hi[0] - lo[0]) * 65535.0);
        const uint32_t qy = (uint32_t)((ky[m] - lo[1]) / (hi[1] - lo[1]) * 65535.0);
        keys[(size_t)m] = (uint64_t)(spread16(qx) | (spread16(qy) << 1));
    }
    return true;
}

// Stable LSD radix sort of the batch indices by their 33-bit locality key (three passes of 11 bits): the same order as
// std::stable_sort by key, a third of its time at M = 2048 (the sort is inside every upload-inclusive step).
static void radix_perm(const std::vector<uint64_t>& keys, std::vector<int>& perm)
{
    const int M = (int)keys.size();
    std::vector<int> tmp((size_t)M);
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * 11;
        unsigned cnt[2049] = {0};
        for (int m = 0; m < M; ++m) ++cnt[((keys[(size_t)perm[(size_t)m]] >> shift) & 0x7ffu) + 1];
        for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
        for (int m = 0; m < M; ++m) tmp[cnt[(keys[(size_t)perm[(size_t)m]] >> shift) & 0x7ffu]++] = perm[(size_t)m];
        perm.swap(tmp);
    }
}

int pgx_score_upload(pgx_ctx* ctx, const double* models, int M)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_upload: points not set");
    if (!models || M <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_upload: empty hypothesis batch");
    const int P = ctx->P;
    // the host mirror of an earlier launch is un-permuted with THAT launch's order: from here on h_perm belongs to this batch, so a
    // fetch of the earlier launch's results must come from the device copy (the finish kernel scattered it through the device perm)
    ctx->mirror_valid = 0;
    std::vector<int> perm((size_t)M);
    for (int m = 0; m < M; ++m) perm[(size_t)m] = m;
    std::vector<uint64_t> keys;
    const bool reorder = ctx->score_sort && M > 64 && locality_keys(ctx, models, M, keys);
    if (reorder) radix_perm(keys, perm);   // 33-bit keys (locality_keys)
    if (!reorder) ctx->h_perm.clear();     // identity
    else ctx->h_perm.assign(perm.begin(), perm.end());
    ctx->Mpad = ((M + 255) / 256) * 256;
    // The (reordered) batch and its permutation are assembled in a pinned staging buffer owned by the context: the caller's
    // array is consumed before the call returns, and the copies need no synchronisation - the host goes on to enqueue the
    // scoring kernels behind them.  An event guards the staging buffer against the next upload.
    const size_t mbytes = (size_t)M * P * sizeof(double), pbytes = (size_t)ctx->Mpad * sizeof(int);
    if (ctx->h_models_busy) { PGX_HIP(ctx, hipEventSynchronize(ctx->ev_models)); ctx->h_models_busy = 0; }
    if (ctx->h_models_cap < mbytes + pbytes) {
        if (ctx->h_models) (void)hipHostFree(ctx->h_models);
        ctx->h_models = nullptr; ctx->h_models_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&ctx->h_models, (mbytes + pbytes) * 2, hipHostMallocDefault));
        ctx->h_models_cap = (mbytes + pbytes) * 2;
    }
    if (!ctx->ev_models) PGX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_models, hipEventDisableTiming));
    double* hm = (double*)ctx->h_models;
    int* hp = (int*)((char*)ctx->h_models + mbytes);
    if (reorder)
        for (int m = 0; m < M; ++m) std::memcpy(hm + (size_t)m * P, models + (size_t)perm[(size_t)m] * P, (size_t)P * sizeof(double));
    else
        std::memcpy(hm, models, mbytes);
    std::memcpy(hp, perm.data(), (size_t)M * sizeof(int));
    std::memset(hp + M, 0, pbytes - (size_t)M * sizeof(int));
    PGX_TRY(ensure(ctx, ctx->models, mbytes));
    PGX_TRY(ensure(ctx, ctx->perm, pbytes));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->models.p, hm, mbytes, hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->perm.p, hp, pbytes, hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipEventRecord(ctx->ev_models, ctx->stream));
    ctx->h_models_busy = 1;
    ctx->M = M; ctx->last_acc = nullptr;
    return PGX_OK;
}

int pgx_solve_minimal(pgx_ctx* ctx, const int32_t* samples, int S, double* models_out)
{
    CTX_GUARD(ctx);
    ctx->h_perm.clear();   // device-generated batches stay in the caller's order
    ctx->mirror_valid = 0; // (see pgx_score_upload: the mirror of an earlier launch goes with that launch's order)
    return solve_minimal_launch(ctx, samples, S, models_out);
}

int pgx_solve_minimal_sampled(pgx_ctx* ctx, int sampler, uint64_t key, uint32_t batch, int S, int32_t* samples_out, double* models_out)
{
    CTX_GUARD(ctx);
    return solve_minimal_sampled_launch(ctx, sampler, key, batch, S, samples_out, models_out);
}

int pgx_sampler_prosac_set(pgx_ctx* ctx, const int32_t* subset_sizes, int count)
{
    CTX_GUARD(ctx);
    return sampler_prosac_set(ctx, subset_sizes, count);
}

int pgx_score_set_global_n(pgx_ctx* ctx, int64_t n_total)
{
    CTX_GUARD(ctx);
    if (n_total < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_set_global_n: negative point count");
    ctx->score_global_n = n_total;
    return PGX_OK;
}

int pgx_score_launch(pgx_ctx* ctx, double T2, int has_compound, int want_masks)
{
    CTX_GUARD(ctx);
    ctx->score_has_compound = has_compound != 0;
    return score_launch(ctx, T2, has_compound, want_masks);
}

static void finish_scores(int n, int has_compound, int exponent, const double* values, const double* shared,
                          double* scores)
{
    if (!scores) return;
    for (int m = 0; m < n; ++m) {
        // scoring_function_with_compound_model.h:110,120: subtract pow(shared, exponent) iff the compound instance
        // is non-empty; std::pow(double,int) promotes to pow(double,double).
        // A hypothesis that shares no inlier with the compound instance (most of a batch) has shared == +0: pow(+0, e) = +0 and
        // v - (+0) = v bit for bit, so libm's pow (12 ns a call: 25 us per 2048-hypothesis fetch, inside every proposal step)
        // runs only where it can change the result.
        const double sh = shared[m];
        scores[m] = has_compound && !(sh == 0.0 && !std::signbit(sh) && exponent > 0) ? values[m] - std::pow(sh, (double)exponent)
                                                                                       : values[m];
    }
}

int pgx_score_fetch(pgx_ctx* ctx, int exponent, int64_t* counts, double* values, double* shared, double* scores,
                    uint64_t* masks)
{
    CTX_GUARD(ctx);
    const int M = ctx->M;
    if (M <= 0 || !ctx->counts.p) return fail(ctx, PGX_ERR_INVALID, "pgx_score_fetch: nothing launched");
    const size_t Mp = (size_t)ctx->Mpad;
    const size_t need = Mp * 24;
    if (ctx->h_res_cap < need) {
        if (ctx->h_res) (void)hipHostFree(ctx->h_res);
        ctx->h_res = nullptr; ctx->h_res_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&ctx->h_res, need * 2, hipHostMallocDefault));
        ctx->h_res_cap = need * 2;
    }
    int64_t* c = (int64_t*)ctx->h_res;
    double* v = (double*)ctx->h_res + Mp;
    double* s = (double*)ctx->h_res + 2 * Mp;
    const bool mirrored = ctx->mirror_valid && ctx->h_mirror_cap >= need;
    // counts | values | shared are one allocation of 3 x Mpad words (score_launch): one copy - or none, when the last
    // kernel of the launch has already written them to the host mirror
    if (!mirrored) PGX_HIP(ctx, hipMemcpyAsync(c, ctx->counts.p, need, hipMemcpyDeviceToHost, ctx->stream));
    if (masks) {
        if (!ctx->have_masks) return fail(ctx, PGX_ERR_INVALID, "pgx_score_fetch: masks were not requested at launch");
        PGX_TRY(d2h(ctx, masks, ctx->masks.p, (size_t)M * (size_t)ctx->words * sizeof(uint64_t)));
    }
    PGX_TRY(sync_deliver(ctx));
    if (mirrored) {   // device order -> the caller's order
        const int64_t* mc = (const int64_t*)ctx->h_mirror;
        const double* mv = (const double*)ctx->h_mirror + Mp;
        const double* ms = (const double*)ctx->h_mirror + 2 * Mp;
        if (ctx->h_perm.size() >= (size_t)M) {
            const int* pm = ctx->h_perm.data();
            for (int m = 0; m < M; ++m) { const int o = pm[m]; c[o] = mc[m]; v[o] = mv[m]; s[o] = ms[m]; }
        } else {
            memcpy(c, mc, (size_t)M * sizeof(int64_t));
            memcpy(v, mv, (size_t)M * sizeof(double));
            memcpy(s, ms, (size_t)M * sizeof(double));
        }
    }
    if (counts) memcpy(counts, c, (size_t)M * sizeof(int64_t));
    if (values) memcpy(values, v, (size_t)M * sizeof(double));
    if (shared) memcpy(shared, s, (size_t)M * sizeof(double));
    finish_scores(M, ctx->score_has_compound, exponent, v, s, scores);
    return PGX_OK;
}

int pgx_score(pgx_ctx* ctx, const double* models, int M, double T2, int has_compound, int exponent,
              int64_t* counts, double* values, double* shared, double* scores, uint64_t* masks)
{
    PGX_TRY(pgx_score_upload(ctx, models, M));
    PGX_TRY(pgx_score_launch(ctx, T2, has_compound, masks != nullptr));
    return pgx_score_fetch(ctx, exponent, counts, values, shared, scores, masks);
}

int pgx_score_algorithmic_bytes(pgx_ctx* ctx, int want_masks, int64_t* bytes, int64_t* pairs)
{
    if (!ctx || ctx->n <= 0 || ctx->M <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_algorithmic_bytes: nothing to score");
    // the inputs once and the outputs once - N d 8 (points) + M p 8 (hypotheses) + M 16 (count, score), plus
    // the compound preference vector (N 8) when the compound term is on.  Derived copies this library keeps (f32 shadow rows,
    // sorted copies, group bounds) are NOT algorithmic bytes.
    int64_t b = ctx->n * ctx->D * 8 + (int64_t)ctx->M * ctx->P * 8 + (int64_t)ctx->M * 16;
    if (ctx->score_has_compound) b += ctx->n * 8;
    if (want_masks) b += (int64_t)ctx->M * ((ctx->n + 63) / 64) * 8;
    if (bytes) *bytes = b;
    if (pairs) *pairs = ctx->n * (int64_t)ctx->M;
    return PGX_OK;
}

int pgx_score_debug_geometry(pgx_ctx* ctx, int what, int value)
{
    CTX_GUARD(ctx);
    switch (what) {
    case 0: if (value < 0 || value > 1024) break; ctx->score_split = value; return PGX_OK;
    case 1: if (value < -1 || value > 1) break; ctx->score_group_xcd = value; return PGX_OK;
    case 2: if (value < 0 || value > 1024 || value % 8) break; ctx->score_nrep = value; return PGX_OK;
    case 3: if (value < 1 || value > 65) break; ctx->score_dense_min = value; return PGX_OK;
    case 4: if (value < 1 || value > 65535) break; ctx->score_cull_segs = value; return PGX_OK;
    }
    return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_geometry: what = %d, value = %d out of range", what, value);
}

int pgx_score_debug_fetch(pgx_ctx* ctx, int what, void* out, int64_t bytes)
{
    CTX_GUARD(ctx);
    if (!out || bytes <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_fetch: empty destination");
    if (!ctx->point_sort) return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_fetch: no sorted copies for the resident points");
    const int64_t groups = (ctx->n + 63) / 64, supers = (groups + kSuper - 1) / kSuper;
    if (what == 5) {   // the last launch's integer accumulators, replicas summed, caller's order: [3][Mpad] u64 (score.hip score_acc_export)
        const int64_t want = (int64_t)3 * ctx->Mpad * 8;
        if (bytes != want) return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_fetch: buffer 5 holds %lld bytes, asked for %lld", (long long)want, (long long)bytes);
        PGX_TRY(ensure(ctx, ctx->g_counts, (size_t)want));
        PGX_TRY(score_acc_export(ctx, (unsigned long long*)ctx->g_counts.p, ctx->stream));
        PGX_TRY(d2h(ctx, out, ctx->g_counts.p, (size_t)bytes));
        PGX_TRY(sync_deliver(ctx));
        return PGX_OK;
    }
    const DevBuf* b = nullptr;
    int64_t have = 0;
    switch (what) {
    case 0: b = &ctx->pperm; have = ctx->n * 4; break;
    case 1: b = &ctx->gbounds; have = (groups + supers) * kGroupRow * 4; break;
    case 2: b = &ctx->pts_g; have = groups * 64 * ctx->D * 8; break;
    case 3: b = &ctx->p32_g; have = groups * 64 * 8 * 4; break;
    case 4: b = &ctx->pts32_s; have = ctx->n * 8 * 4; break;
    default: return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_fetch: unknown buffer %d", what);
    }
    if (bytes != have) return fail(ctx, PGX_ERR_INVALID, "pgx_score_debug_fetch: buffer %d holds %lld bytes, asked for %lld", what, (long long)have, (long long)bytes);
    PGX_TRY(d2h(ctx, out, b->p, (size_t)bytes));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

int pgx_score_profile(pgx_ctx* ctx, int on)
{
    CTX_GUARD(ctx);
    if (on && !ctx->kev[0])
        for (int k = 0; k < 5; ++k) PGX_HIP(ctx, hipEventCreate(&ctx->kev[k]));
    ctx->score_profile = on < 0 ? 0 : (on > 2 ? 2 : on);
    return PGX_OK;
}

int pgx_score_kernel_times(pgx_ctx* ctx, float ms[4])
{
    CTX_GUARD(ctx);
    if (!ms || !ctx->kev[0] || ctx->last_score_path == 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_kernel_times: no profiled launch");
    ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    if (ctx->score_profile == 0) return fail(ctx, PGX_ERR_INVALID, "pgx_score_kernel_times: profiling is off");
    if (ctx->last_score_path == 2) {   // cull + group-major
        PGX_HIP(ctx, hipEventSynchronize(ctx->score_profile >= 2 ? ctx->kev[3] : ctx->kev[2]));
        PGX_HIP(ctx, hipEventElapsedTime(&ms[1], ctx->kev[1], ctx->kev[2]));
        if (ctx->score_profile >= 2) {
            PGX_HIP(ctx, hipEventElapsedTime(&ms[0], ctx->kev[0], ctx->kev[1]));
            PGX_HIP(ctx, hipEventElapsedTime(&ms[2], ctx->kev[4], ctx->kev[3]));
            PGX_HIP(ctx, hipEventElapsedTime(&ms[3], ctx->kev[2], ctx->kev[4]));
        }
    } else {                           // chunked kernel + reduce
        PGX_HIP(ctx, hipEventSynchronize(ctx->score_profile >= 2 ? ctx->kev[3] : ctx->kev[1]));
        PGX_HIP(ctx, hipEventElapsedTime(&ms[0], ctx->kev[0], ctx->kev[1]));
        if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventElapsedTime(&ms[2], ctx->kev[4], ctx->kev[3]));
    }
    return PGX_OK;
}

int pgx_score_stats(pgx_ctx* ctx, double T2, int has_compound, int64_t stats[8])
{
    CTX_GUARD(ctx);
    if (!stats) return fail(ctx, PGX_ERR_INVALID, "pgx_score_stats: stats is NULL");
    for (int k = 0; k < 8; ++k) stats[k] = 0;
    ctx->score_stats = 1;
    ctx->score_has_compound = has_compound != 0;
    const int rc = score_launch(ctx, T2, has_compound, 0);
    ctx->score_stats = 0;
    if (rc != PGX_OK) return rc;
    const int64_t groups = (ctx->n + 63) / 64;
    stats[0] = ctx->n * (int64_t)ctx->M;   // (point, hypothesis) pairs of the batch
    stats[6] = ctx->last_score_path;
    stats[7] = ctx->last_score_filtered;
    if (ctx->last_score_path == 2) {
        unsigned long long h[8];
        PGX_TRY(d2h(ctx, h, ctx->stats_buf.p, sizeof(h)));
        PGX_TRY(sync_deliver(ctx));
        stats[1] = groups * (int64_t)ctx->M;  // (hypothesis, group) bound tests an un-hierarchical cull would run
        stats[2] = (int64_t)h[0];             // surviving (hypothesis, group) steps: 64 f32 filter evaluations each
        stats[3] = (int64_t)h[1];             // exact FP64 residual evaluations
        stats[4] = (int64_t)h[2];             // inlier pairs
        // PGX_VERIFY=1: inlier pairs (by the exact residual over EVERY pair) that the group bound or the f32 filter discarded
        // (+ the difference between the exhaustive pass's inlier count and the scoring path's: equal unless something else is wrong)
        stats[5] = ctx->verify ? (int64_t)(h[4] + h[5]) + ((int64_t)h[6] > stats[4] ? (int64_t)h[6] - stats[4] : stats[4] - (int64_t)h[6]) : -1;
    } else {
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        stats[3] = ctx->last_score_filtered ? -1 : stats[0];  // unfiltered chunked kernel: every pair is evaluated exactly
    }
    return PGX_OK;
}

/* ---- preference / compound -------------------------------------------------------------------------------- */
static int slot_buffer(pgx_ctx* ctx, int slot, double** out)
{
    if (slot < 0 || slot > 4096) return fail(ctx, PGX_ERR_INVALID, "preference slot %d out of range", slot);
    if ((int)ctx->slots.size() <= slot) ctx->slots.resize((size_t)slot + 1);
    PGX_TRY(ensure(ctx, ctx->slots[slot], (size_t)ctx->n * sizeof(double)));
    *out = ctx->slots[slot].as<double>();
    return PGX_OK;
}

int pgx_preference(pgx_ctx* ctx, const double* model, double T2, int slot, double* pref_out, double* dot,
                   double* pref_sqnorm, double* comp_sqnorm)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0 || !model) return fail(ctx, PGX_ERR_INVALID, "pgx_preference: points/model not set");
    double* d_pref = nullptr;
    PGX_TRY(slot_buffer(ctx, slot, &d_pref));
    double out3[3] = {0, 0, 0};
    PGX_TRY(preference_launch(ctx, model, T2, d_pref, out3));
    if (dot) *dot = out3[0];
    if (pref_sqnorm) *pref_sqnorm = out3[1];
    if (comp_sqnorm) *comp_sqnorm = out3[2];
    if (pref_out) {
        PGX_TRY(d2h(ctx, pref_out, d_pref, (size_t)ctx->n * sizeof(double)));
        PGX_TRY(sync_deliver(ctx));
    }
    return PGX_OK;
}

int pgx_get_preference(pgx_ctx* ctx, int slot, double* pref_out)
{
    CTX_GUARD(ctx);
    if (slot < 0 || slot >= (int)ctx->slots.size() || !ctx->slots[slot].p || !pref_out)
        return fail(ctx, PGX_ERR_INVALID, "pgx_get_preference: slot %d is empty", slot);
    PGX_TRY(d2h(ctx, pref_out, ctx->slots[slot].p, (size_t)ctx->n * sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

int pgx_compound_update(pgx_ctx* ctx, const int32_t* slots, int K, double* compound_out)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_compound_update: points not set");
    if (K > 0) {  // progressive_x.h:600-601: nothing happens for an empty compound instance
        if (!slots) return fail(ctx, PGX_ERR_INVALID, "pgx_compound_update: slots is NULL");
        PGX_TRY(compound_launch(ctx, slots, K));
    }
    if (compound_out) return pgx_get_compound(ctx, compound_out);
    return PGX_OK;
}

/* ---- PEARL: unary table, labels, graph --------------------------------------------------------------------- */
int pgx_pearl_unary(pgx_ctx* ctx, const double* models, int K, double threshold, double lambda, int64_t* Dq_out)
{
    CTX_GUARD(ctx);
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_pearl_unary: points not set");
    if (K < 0 || (K > 0 && !models)) return fail(ctx, PGX_ERR_INVALID, "pgx_pearl_unary: bad model list");
    const int L = K + 1;
    PGX_TRY(ensure(ctx, ctx->kmodels, (size_t)(K > 0 ? K : 1) * ctx->P * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->dq, (size_t)L * (size_t)ctx->n * sizeof(int64_t)));
    if (K > 0)
        PGX_HIP(ctx, hipMemcpyAsync(ctx->kmodels.p, models, (size_t)K * ctx->P * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_TRY(unary_launch(ctx, K, threshold, lambda));
    {   // what each column was computed from, byte for byte (the first-cycle memo of pgx_expansion compares these)
        ctx->unary_ident.assign((size_t)L, std::string());
        char head[64];
        const int hl = std::snprintf(head, sizeof head, "%d|%lld|%a|%a|", ctx->model_type, (long long)ctx->points_version, threshold, lambda);
        for (int l = 0; l < K; ++l)
            ctx->unary_ident[(size_t)l] = std::string(head, (size_t)hl) + std::string((const char*)(models + (size_t)l * ctx->P), (size_t)ctx->P * sizeof(double));
        ctx->unary_ident[(size_t)K] = std::string(head, (size_t)hl) + "outlier";
    }
    ctx->L = L;
    ctx->dq_n = ctx->n;
    ctx->dq_max = (int64_t)1 << 33;  // 2 (1 - lambda) <= 2 in 2^-32 fixed point (PEARL.h:123)
    if (Dq_out) {  // ABI layout is point-major N x L (as the reference's per-point functor); device is label-major
        std::vector<int64_t> tmp((size_t)L * (size_t)ctx->n);
        PGX_TRY(d2h(ctx, tmp.data(), ctx->dq.p, tmp.size() * sizeof(int64_t)));
        PGX_TRY(sync_deliver(ctx));
        for (int l = 0; l < L; ++l)
            for (int64_t i = 0; i < ctx->n; ++i) Dq_out[i * L + l] = tmp[(size_t)l * ctx->n + i];
    } else {
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `models` may be freed by the caller
    }
    return PGX_OK;
}

int pgx_set_unary_q(pgx_ctx* ctx, const int64_t* Dq, int64_t n, int L)
{
    CTX_GUARD(ctx);
    if (!Dq || n <= 0 || L <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_unary_q: empty table");
    if (n >= ((int64_t)1 << 31)) return fail(ctx, PGX_ERR_INVALID, "pgx_set_unary_q: n must be < 2^31");
    std::vector<int64_t> tmp((size_t)L * (size_t)n);
    int64_t mx = 0;
    for (int l = 0; l < L; ++l)
        for (int64_t i = 0; i < n; ++i) {
            const int64_t v = Dq[i * L + l];
            tmp[(size_t)l * n + i] = v;
            if (v < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_unary_q: negative cost at site %lld, label %d", (long long)i, l);
            if (v > mx) mx = v;
        }
    ctx->dq_max = mx;
    ctx->unary_ident.clear();   // an injected table has no identity: no first-cycle memo, no identical-call shortcut
    PGX_TRY(ensure(ctx, ctx->dq, tmp.size() * sizeof(int64_t)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->dq.p, tmp.data(), tmp.size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->L = L;
    ctx->dq_n = n;
    return PGX_OK;
}

int pgx_set_labels(pgx_ctx* ctx, const int32_t* labels, int64_t n)
{
    CTX_GUARD(ctx);
    if (!labels || n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_labels: empty labels");
    int32_t lo = labels[0], hi = labels[0];
    for (int64_t i = 1; i < n; ++i) { lo = labels[i] < lo ? labels[i] : lo; hi = labels[i] > hi ? labels[i] : hi; }
    if (lo < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_labels: negative label %d", (int)lo);
    ctx->labels_max = hi;
    ctx->labels_all_zero = hi == 0 ? 1 : 0;
    ctx->last_done.valid = 0;   // (the labels are the caller's now)
    ctx->labels_version += 1;
    PGX_TRY(ensure(ctx, ctx->labels, (size_t)n * sizeof(int32_t)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->labels.p, labels, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->labels_n = n;
    return PGX_OK;
}

int pgx_get_labels(pgx_ctx* ctx, int32_t* labels)
{
    CTX_GUARD(ctx);
    if (!labels || ctx->labels_n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_get_labels: labels not set");
    PGX_TRY(d2h(ctx, labels, ctx->labels.p, (size_t)ctx->labels_n * sizeof(int32_t)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

int pgx_set_graph(pgx_ctx* ctx, int64_t n, const int32_t* off, const int32_t* idx, const int32_t* mult)
{
    CTX_GUARD(ctx);
    if (n <= 0 || !off) return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: empty graph");
    const int64_t E = off[n];
    if (E < 0 || off[0] != 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: malformed offsets");
    if (E > 0 && (!idx || !mult)) return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: idx/mult missing");
    int maxdeg = 0;
    int64_t max_row = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int deg = off[i + 1] - off[i];
        if (deg < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: offsets not monotone at row %lld", (long long)i);
        if (deg > maxdeg) maxdeg = deg;
        int64_t row = 0;
        for (int32_t a = off[i]; a < off[i + 1]; ++a) {
            if (idx[a] < 0 || idx[a] >= n || idx[a] == i)
                return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: bad neighbour %d in row %lld", idx[a], (long long)i);
            if (mult[a] <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: multiplicity must be positive");
            row += mult[a];
        }
        if (row > max_row) max_row = row;
    }
    PGX_TRY(ensure(ctx, ctx->goff, (size_t)(n + 1) * sizeof(int32_t)));
    PGX_TRY(ensure(ctx, ctx->gidx, (size_t)(E > 0 ? E : 1) * sizeof(int32_t)));
    PGX_TRY(ensure(ctx, ctx->gmult, (size_t)(E > 0 ? E : 1) * sizeof(int32_t)));
    PGX_TRY(ensure(ctx, ctx->grev, (size_t)(E > 0 ? E : 1) * sizeof(int32_t)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->goff.p, off, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (E > 0) {
        PGX_HIP(ctx, hipMemcpyAsync(ctx->gidx.p, idx, (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        PGX_HIP(ctx, hipMemcpyAsync(ctx->gmult.p, mult, (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gn = n; ctx->gE = E; ctx->max_degree = maxdeg; ctx->max_row_mult = max_row;
    ctx->gorder_n = 0;   // no coordinates: the tile path orders the sites like the resident points (or not at all)
    return graph_build_reverse(ctx);
}

int pgx_set_weights(pgx_ctx* ctx, const double* weights, int64_t len)
{
    CTX_GUARD(ctx);
    if (!weights || len == 0) { ctx->weights_n = 0; return PGX_OK; }  // clears
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_set_weights: points not set");
    if (len != ctx->n)
        return fail(ctx, PGX_ERR_INVALID, "pgx_set_weights: %lld weights for %lld points", (long long)len, (long long)ctx->n);
    PGX_TRY(ensure(ctx, ctx->weights, (size_t)len * sizeof(double)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->weights.p, weights, (size_t)len * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->weights_n = len;
    return PGX_OK;
}

int pgx_gram(pgx_ctx* ctx, int kind, const double* params, int nparams, int sel, const int32_t* index, int64_t m, int label,
             int use_weights, int weight_power, double* out, int64_t* count, int64_t* bad)
{
    CTX_GUARD(ctx);
    return gram_launch(ctx, kind, params, nparams, sel, index, m, label, use_weights, weight_power, out, count, bad);
}

int pgx_graph_build(pgx_ctx* ctx, const double* points, int64_t n, int d, int kind, double radius, int k, int64_t* arcs)
{
    CTX_GUARD(ctx);
    return graph_build_launch(ctx, points, n, d, kind, radius, k, arcs);
}

int pgx_graph_fetch(pgx_ctx* ctx, int32_t* off, int32_t* idx, int32_t* mult)
{
    CTX_GUARD(ctx);
    return graph_fetch_launch(ctx, off, idx, mult);
}

int pgx_graph_size(pgx_ctx* ctx, int64_t* n, int64_t* arcs)
{
    if (!ctx || !n || !arcs) return fail(ctx, PGX_ERR_INVALID, "pgx_graph_size: NULL argument");
    *n = ctx->gn > 0 ? ctx->gn : 0;
    *arcs = ctx->gn > 0 ? ctx->gE : 0;
    return PGX_OK;
}

static int flow_params(pgx_ctx* ctx, double lambda, double label_cost, int64_t* lambda_q, int64_t* h_q)
{
    if (!(lambda >= 0.0) || !(label_cost >= 0.0))
        return fail(ctx, PGX_ERR_INVALID, "lambda and label_cost must be >= 0");
    *lambda_q = quantize_lambda(lambda);
    *h_q = quantize(label_cost);
    // Range check: every excess / capacity sum must stay far below 2^63.
    // (an injected table - pgx_set_unary_q - may hold costs beyond PEARL's 2 (1 - lambda) <= 2: dq_max is what counts)
    const double dmax = (double)ctx->dq_max / 4294967296.0;
    const double per_site = 2.0 * (dmax > 2.0 ? dmax : 2.0) + 2.0 * lambda * (double)(ctx->gn > 0 ? ctx->max_row_mult : 0);
    const double total = per_site * (double)(ctx->dq_n > 0 ? ctx->dq_n : 1) + label_cost * (double)(ctx->L + 1);
    if (total >= 1073741824.0)  // 2^30 * 2^32 = 2^62
        return fail(ctx, PGX_ERR_RANGE, "fixed-point range exceeded: n*(2*max_cost+2*lambda*row_mult)+L*h = %.3g >= 2^30", total);
    return PGX_OK;
}

int pgx_energy(pgx_ctx* ctx, double lambda, double label_cost, int64_t* energy_q, double* energy)
{
    CTX_GUARD(ctx);
    int64_t lq, hq, e = 0;
    PGX_TRY(flow_params(ctx, lambda, label_cost, &lq, &hq));
    PGX_TRY(energy_launch(ctx, lq, hq, &e));
    if (energy_q) *energy_q = e;
    if (energy) *energy = (double)e / 4294967296.0;
    return PGX_OK;
}

int pgx_expand_alpha(pgx_ctx* ctx, double lambda, double label_cost, int alpha, int64_t* changed)
{
    CTX_GUARD(ctx);
    int64_t lq, hq, ch = 0;
    PGX_TRY(flow_params(ctx, lambda, label_cost, &lq, &hq));
    for (int k = 0; k < 8; ++k) ctx->stats[k] = 0;
    ctx->labels_all_zero = 0;
    ctx->last_done.valid = 0;
    ctx->labels_version += 1;
    PGX_TRY(expand_alpha_launch(ctx, lq, hq, alpha, &ch));
    if (changed) *changed = ch;
    return PGX_OK;
}

// GCO-v3 "standard cycles" loop [U-5] as PEARL drives it (PEARL.h:550-551): cycle over the labels in index order until
// a whole cycle leaves the energy unchanged, at most max_cycles cycles.
static thread_local bool done_verify_running = false;   // (PGX_MF_DONE_VERIFY: the nested call must run the real cycle)

int pgx_expansion(pgx_ctx* ctx, double lambda, double label_cost, int max_cycles, int64_t* energy_q, double* energy,
                  int* cycles)
{
    CTX_GUARD(ctx);
    int64_t lq, hq;
    PGX_TRY(flow_params(ctx, lambda, label_cost, &lq, &hq));
    for (int k = 0; k < 8; ++k) ctx->stats[k] = 0;
    // ---- identical call (pgx_internal.h ExpansionDone): the labels are those the last expansion left at a fixed point, and columns,
    // weights and graph are what it was computed from => one verifying cycle of no-ops and the same energy, without running it
    pgx_ctx::ExpansionDone& last = ctx->last_done;
    const bool ident_known = ctx->L > 0 && (int)ctx->unary_ident.size() == ctx->L && ctx->labels_n == ctx->dq_n;
    if (ctx->mf_memo && last.valid && ident_known && max_cycles >= 1 && last.lq == lq && last.hq == hq && last.n == ctx->dq_n &&
        (lq <= 0 || last.graph_version == ctx->graph_version) && last.ident == ctx->unary_ident &&
        last.labels_version == ctx->labels_version && !done_verify_running) {
        if (ctx->labels_max >= ctx->L) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: label %d out of range (the unary table has %d labels)", ctx->labels_max, ctx->L);
        ctx->memo_hits += ctx->L;
        ctx->stats[7] += ctx->L;   // "moves skipped because the labelling had not changed since that label's last move, which relabelled nothing"
        if (ctx->mf_done_verify) {   // debug mode: run the cycle the shortcut stands for and insist on what it promises
            int64_t eq2 = 0;
            int cyc2 = 0;
            const int64_t want_e = last.energy_q;
            done_verify_running = true;
            const int rc = pgx_expansion(ctx, lambda, label_cost, max_cycles, &eq2, nullptr, &cyc2);
            done_verify_running = false;
            if (rc != PGX_OK) return rc;
            if (eq2 != want_e || cyc2 != 1 || ctx->stats[4] != 0)
                return fail(ctx, PGX_ERR_INVALID, "pgx_expansion (PGX_MF_DONE_VERIFY): the identical-call shortcut would have answered energy %lld, 1 cycle, 0 changes; "
                                                  "the real cycle gave energy %lld, %d cycle(s), %lld change(s)", (long long)want_e, (long long)eq2, cyc2, (long long)ctx->stats[4]);
        }
        if (energy_q) *energy_q = last.energy_q;
        if (energy) *energy = (double)last.energy_q / 4294967296.0;
        if (cycles) *cycles = 1;
        return PGX_OK;
    }
    last.valid = 0;
    ctx->labels_version += 1;   // (the moves below write the labels)
    int64_t new_e = 0;
    PGX_TRY(energy_launch(ctx, lq, hq, &new_e));
    int64_t old_e = new_e + 1;
    int done = 0;
    int64_t last_cycle_changed = -1;
    int64_t version = 0;
    std::vector<int64_t> noop_at((size_t)(ctx->L > 0 ? ctx->L : 1), -1);
    // ---- first-cycle memo (pgx_internal.h ExpansionMemo): usable when this expansion starts from the all-zero labelling on columns
    // of known identity; `keep` = first-cycle moves recorded by this call, `memo_p` = leading moves restored instead of solved
    pgx_ctx::ExpansionMemo& memo = ctx->memo;
    const int64_t n_sites = ctx->dq_n;
    const bool memo_on = ctx->mf_memo && lq > 0 && ctx->labels_all_zero && ctx->L > 1 && (int)ctx->unary_ident.size() == ctx->L &&
                         ctx->labels_n == n_sites && (size_t)ctx->L * (size_t)n_sites * 4 <= ((size_t)2 << 30);
    ctx->labels_all_zero = 0;
    int memo_p = 0;
    if (memo_on) {
        if (memo.n != n_sites || memo.lq != lq || memo.hq != hq || memo.graph_version != ctx->graph_version) memo.valid = 0;
        while (memo_p < memo.valid && memo_p < ctx->L && memo.ident[(size_t)memo_p] == ctx->unary_ident[(size_t)memo_p]) ++memo_p;
        if (memo.cap < ctx->L || memo.n != n_sites) {   // room for this call's snapshots (the kept prefix moves along)
            pgx::DevBuf grown;
            PGX_TRY(ensure(ctx, grown, (size_t)ctx->L * (size_t)n_sites * 4));
            if (memo_p > 0)
                PGX_HIP(ctx, hipMemcpyAsync(grown.p, memo.snaps.p, (size_t)memo_p * (size_t)n_sites * 4, hipMemcpyDeviceToDevice, ctx->stream));
            PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            release(memo.snaps);
            memo.snaps = grown;
            memo.cap = ctx->L;
        }
        memo.ident.resize((size_t)ctx->L);
        memo.changed.resize((size_t)ctx->L, 0);
        memo.valid = memo_p;
        memo.n = n_sites; memo.lq = lq; memo.hq = hq; memo.graph_version = ctx->graph_version;
    }
    auto memo_snapshot = [&](int a) -> int {     // enqueued right behind move a of the first cycle: the labels it left
        PGX_HIP(ctx, hipMemcpyAsync((char*)memo.snaps.p + (size_t)a * (size_t)n_sites * 4, ctx->labels.p, (size_t)n_sites * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return PGX_OK;
    };
    auto memo_record = [&](int a, int64_t ch) {  // once move a's outcome is known (moves are recorded in order: valid stays a prefix)
        if (memo.valid == a) { memo.ident[(size_t)a] = ctx->unary_ident[(size_t)a]; memo.changed[(size_t)a] = ch; memo.valid = a + 1; }
    };
    for (int cycle = 1; cycle <= max_cycles; ++cycle) {
        if (new_e == old_e) break;
        old_e = new_e;
        int64_t changed_total = 0;
        int alpha0 = 0;
        const bool keep = memo_on && cycle == 1;
        if (keep && memo_p > 0) {   // the leading moves as they went last time: their label changes, the skip rule's bookkeeping, the labels
            for (int a = 0; a < memo_p; ++a) {
                const int64_t ch = memo.changed[(size_t)a];
                changed_total += ch;
                if (ch > 0) { ++version; noop_at[(size_t)a] = -1; }
                else noop_at[(size_t)a] = version;
            }
            PGX_HIP(ctx, hipMemcpyAsync(ctx->labels.p, (char*)memo.snaps.p + (size_t)(memo_p - 1) * (size_t)n_sites * 4, (size_t)n_sites * 4,
                                        hipMemcpyDeviceToDevice, ctx->stream));
            ctx->memo_hits += memo_p;
            alpha0 = memo_p;
        }
        if (lq <= 0) {  // no pairwise term: closed-form moves, the whole cycle enqueued at once (one host round trip)
            std::vector<int64_t> ch((size_t)(ctx->L > 0 ? ctx->L : 1), 0);
            std::vector<int> ev((size_t)(ctx->L > 0 ? ctx->L : 1), 0);
            PGX_TRY(expand_cycle_l0(ctx, hq, ch.data(), ev.data()));
            for (int alpha = 0; alpha < ctx->L; ++alpha) changed_total += ch[(size_t)alpha];
        } else if ((ctx->mf_tile && ctx->mf_tile_batch && ctx->dq_n <= ctx->tile_single_max && ctx->dq_n <= 8192 && ctx->L <= 64) ||
                   (region_moves_apply(ctx) && !(ctx->mf_tile && ctx->dq_n <= ctx->tile_single_max && ctx->dq_n <= ctx->tile_expansion_max))) {
            // (graphs of <= 8192 sites - the reference's own scenes: every move is ONE launch of the one-workgroup solver,
            //  maxflow_tile.hip expand_alpha_tile; batched the same way, a read-back per move was two thirds of a move's time)
            // Region moves (maxflow_tile.hip): the moves of the cycle are enqueued back to back and resolved together - one
            // host round trip per batch instead of one per move.  A move that declines poisons the rest of its batch on the device
            // (they return untouched); it is solved by the general path and the cycle resumes behind it.  The skip rule below is
            // applied on the device for moves behind others of the same batch (their outcome is not known when they are enqueued).
            int alpha = alpha0;
            std::vector<int> batch;
            while (alpha < ctx->L) {
                batch.clear();
                PGX_TRY(region_batch_begin(ctx));
                const int64_t version0 = version;
                int rc = PGX_OK;
                for (int a = alpha; a < ctx->L && rc == PGX_OK; ++a) {
                    if (batch.empty() && noop_at[a] == version) { ctx->stats[7]++; alpha = a + 1; continue; }   // (known now: nothing is in flight)
                    ctx->region_defer = 1;
                    ctx->region_slot = (int)batch.size();
                    ctx->region_skip_rel = noop_at[a] >= version0 ? (int)(noop_at[a] - version0) : -1;
                    int64_t ch = 0;
                    rc = expand_alpha_launch(ctx, lq, hq, a, &ch);
                    ctx->region_defer = 0;
                    if (rc == PGX_REGION_PENDING) { batch.push_back(a); rc = keep ? memo_snapshot(a) : PGX_OK; }
                    else if (rc == PGX_OK) rc = fail(ctx, PGX_ERR_INVALID, "pgx_expansion: a batched move was not enqueued");
                }
                if (rc != PGX_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
                if (batch.empty()) break;
                PGX_TRY(region_batch_fetch(ctx, (int)batch.size()));
                PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
                int next = ctx->L;
                for (size_t k = 0; k < batch.size(); ++k) {
                    const int a = batch[k];
                    int status = 0;
                    int64_t ch = 0;
                    PGX_TRY(region_result(ctx, (int)k, a, &status, &ch));
                    if (status == 2) {   // the device applied the skip rule
                        if (noop_at[a] != version) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: move %d of a batch did not run", a);
                        ctx->stats[7]++;
                        continue;
                    }
                    if (status == 1) {   // declined: the general path, then a new batch behind it
                        const int region_on = ctx->mf_region;
                        ctx->mf_region = 0;
                        const int r1 = expand_alpha_launch(ctx, lq, hq, a, &ch);
                        ctx->mf_region = region_on;
                        PGX_TRY(r1);
                        if (cycle == 1 && memo_on) PGX_TRY(memo_snapshot(a));   // (the snapshot taken behind the declined move holds the untouched labels)
                        next = a + 1;
                    }
                    if (cycle == 1 && memo_on) memo_record(a, ch);
                    changed_total += ch;
                    if (ch > 0) { ++version; noop_at[a] = -1; }
                    else noop_at[a] = version;
                    if (status == 1) break;
                }
                alpha = next;
            }
        } else
        for (int alpha = alpha0; alpha < ctx->L; ++alpha) {
            // a move is a deterministic function of (labelling, alpha): one that relabelled nothing and has seen no
            // label change since would relabel nothing again — skipped (typically the tail of the verifying cycle)
            if (noop_at[alpha] == version) { ctx->stats[7]++; continue; }
            int64_t ch = 0;
            PGX_TRY(expand_alpha_launch(ctx, lq, hq, alpha, &ch));
            if (keep) { PGX_TRY(memo_snapshot(alpha)); memo_record(alpha, ch); }
            changed_total += ch;
            if (ch > 0) { ++version; noop_at[alpha] = -1; }
            else noop_at[alpha] = version;
        }
        // a cycle that relabels nothing leaves the energy unchanged by construction: skip the recomputation
        if (changed_total > 0) PGX_TRY(energy_launch(ctx, lq, hq, &new_e));
        done = cycle;
        last_cycle_changed = changed_total;
    }
    if (ident_known && done >= 1 && last_cycle_changed == 0) {   // a fixed point of the whole cycle (a final cycle that moved ties at equal energy is not)
        last.valid = 1;
        last.ident = ctx->unary_ident;
        last.lq = lq; last.hq = hq; last.n = ctx->dq_n; last.graph_version = ctx->graph_version; last.energy_q = new_e;
        last.labels_version = ctx->labels_version;
    }
    if (energy_q) *energy_q = new_e;
    if (energy) *energy = (double)new_e / 4294967296.0;
    if (cycles) *cycles = done;
    return PGX_OK;
}

int pgx_greedy_labeling(pgx_ctx* ctx, double label_cost, int64_t* energy_q, double* energy, int* opened)
{
    CTX_GUARD(ctx);
    int64_t lq, hq, e = 0;
    PGX_TRY(flow_params(ctx, 0.0, label_cost, &lq, &hq));
    int op = 0;
    PGX_TRY(greedy_labeling_launch(ctx, hq, &e, &op));
    if (energy_q) *energy_q = e;
    if (energy) *energy = (double)e / 4294967296.0;
    if (opened) *opened = op;
    return PGX_OK;
}

int pgx_expansion_stats(pgx_ctx* ctx, int64_t stats[8])
{
    if (!ctx || !stats) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion_stats: NULL argument");
    for (int k = 0; k < 8; ++k) stats[k] = ctx->stats[k];
    return PGX_OK;
}

int pgx_expansion_paths(pgx_ctx* ctx, int64_t paths[6])
{
    if (!ctx || !paths) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion_paths: NULL argument");
    for (int k = 0; k < 6; ++k) paths[k] = ctx->paths[k];
    paths[1] = ctx->memo_hits;
    paths[5] = ctx->tile_fallbacks;
    return PGX_OK;
}

int pgx_expansion_schedule(pgx_ctx* ctx, int64_t out[8])
{
    if (!ctx || !out) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion_schedule: NULL argument");
    return maxflow_schedule_stats(ctx, out);
}

int pgx_one_workgroup_launches(pgx_ctx* ctx, int64_t out[2])
{
    if (!ctx || !out) return fail(ctx, PGX_ERR_INVALID, "pgx_one_workgroup_launches: NULL argument");
    out[0] = ctx->tile_launches[0];
    out[1] = ctx->tile_launches[1];
    return PGX_OK;
}

int pgx_bucket(pgx_ctx* ctx, int L, int64_t* counts, int32_t* order)
{
    CTX_GUARD(ctx);
    if (!counts) return fail(ctx, PGX_ERR_INVALID, "pgx_bucket: counts is NULL");
    return bucket_launch(ctx, L, counts, order);
}

int pgx_residual_sum(pgx_ctx* ctx, const double* model, int label, double* sum)
{
    CTX_GUARD(ctx);
    if (!model || !sum) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sum: NULL argument");
    return residual_sum_launch(ctx, model, label, sum);
}

int pgx_gram_labels(pgx_ctx* ctx, int kind, const double* params, int nparams, int K, int use_weights, int weight_power,
                    double* out, int64_t* count, int64_t* bad)
{
    CTX_GUARD(ctx);
    return gram_labels_launch(ctx, kind, params, nparams, K, use_weights, weight_power, out, count, bad);
}

int pgx_residual_sums(pgx_ctx* ctx, const double* models, int K, double* sums)
{
    CTX_GUARD(ctx);
    if (!models || !sums) return fail(ctx, PGX_ERR_INVALID, "pgx_residual_sums: NULL argument");
    return residual_sums_launch(ctx, models, K, sums);
}

int pgx_gram_batch(pgx_ctx* ctx, int kind, const double* params, int nparams, const int32_t* index, int B, int m,
                   const double* weights_sel, int weight_power, double* out, int32_t* bad)
{
    CTX_GUARD(ctx);
    return gram_batch_launch(ctx, kind, params, nparams, index, B, m, weights_sel, weight_power, out, bad);
}

int pgx_pnp_refine_batch(pgx_ctx* ctx, const double* inits, const int32_t* index, int B, int m, const double* weights_sel,
                         int weight_power, int iterations, double* out, int32_t* status)
{
    CTX_GUARD(ctx);
    return pnp_refine_batch_launch(ctx, inits, index, B, m, weights_sel, weight_power, iterations, out, status);
}

int pgx_eigh_smallest_batch(pgx_ctx* ctx, const double* A, int q, int64_t B, double* vec, double* val)
{
    CTX_GUARD(ctx);
    return eigh_smallest_launch(ctx, A, q, B, vec, val);
}

int pgx_gc_labeling(pgx_ctx* ctx, const double* model, double T2, double lambda, int32_t* flags, int64_t* count)
{
    CTX_GUARD(ctx);
    if (!model || !flags) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_labeling: NULL argument");
    return gc_labeling_launch(ctx, model, T2, lambda, flags, count);
}

int pgx_epipolar_support(pgx_ctx* ctx, const double* F, double T2, double S2, int64_t counts[2])
{
    CTX_GUARD(ctx);
    if (!F || !counts) return fail(ctx, PGX_ERR_INVALID, "pgx_epipolar_support: NULL argument");
    return epipolar_support_launch(ctx, F, T2, S2, counts);
}

int pgx_score_inliers(pgx_ctx* ctx, int row, int32_t* index, int64_t* count)
{
    CTX_GUARD(ctx);
    if (!index || !count) return fail(ctx, PGX_ERR_INVALID, "pgx_score_inliers: NULL argument");
    return score_inliers_launch(ctx, row, index, count);
}

int pgx_gc_inliers(pgx_ctx* ctx, const double* model, double T2, double lambda, int32_t* index, int64_t* count)
{
    CTX_GUARD(ctx);
    if (!model || !index || !count) return fail(ctx, PGX_ERR_INVALID, "pgx_gc_inliers: NULL argument");
    return gc_labeling_launch(ctx, model, T2, lambda, index, count, true);
}

}  // extern "C"
