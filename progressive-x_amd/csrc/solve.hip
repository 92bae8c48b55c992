// solve.hip — batched minimal solvers on the resident points (first slice): the two solvers with
// an exact in-tree or closed-form specification.  The hypotheses are generated straight into the resident hypothesis
// buffer, so scoring them needs no host -> device model upload.
//
// Replaces, per sample of the proposal engine's main loop (gcransac::GCRANSAC::run, graph-cut-ransac submodule, absent):
//   vanishing point from two segments   /root/reference/src/pyprogressivex/include/solver_vanishing_point_two_lines.h:147-185
//                                       (l_i = e_i0 x e_i1 :174-179, v = l_0 x l_1 :180-182, vec_norm :123-131)
//   2D line through two points          Default2DLineEstimator's minimal solver (progressivex_python.cpp:489), absent
//                                       upstream [U-4]: unit normal (-dy, dx) / |d|, offset c = -(n . a)
// Operation order is the contract (bit-exact against the oracle's C restatement, no contraction, IEEE sqrt and divide).
// A degenerate sample (coincident points / parallel or identical lines) yields a NaN model, which can never have an
// inlier; the caller drops it (the reference's solvers return "no model").
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pgx_internal.h"

namespace pgx {

namespace {

constexpr int kSolveBlock = 256;

__device__ __forceinline__ void cross3(double a1, double b1, double c1, double a2, double b2, double c2, double* o)
{
    o[0] = b1 * c2 - c1 * b2;      // solver_vanishing_point_two_lines.h:106-121
    o[1] = -(a1 * c2 - c1 * a2);
    o[2] = a1 * b2 - b1 * a2;
}

template <int MT>
__global__ __launch_bounds__(kSolveBlock) void solve_kernel(const double* __restrict__ pts, int64_t n, const int* __restrict__ samples,
                                                            int S, double* __restrict__ models, int* __restrict__ perm, int Mpad)
{
    const int s = (int)(blockIdx.x * kSolveBlock + threadIdx.x);
    if (s < Mpad) perm[s] = s < S ? s : 0;  // generated in the caller's order: no locality permutation
    if (s >= S) return;
    const int i0 = samples[2 * s], i1 = samples[2 * s + 1];
    double m[3];
    const double nan = __builtin_nan("");
    if (i0 < 0 || i1 < 0 || i0 >= n || i1 >= n) {
        m[0] = m[1] = m[2] = nan;
    } else if (MT == kLine2D) {
        const double ax = pts[(int64_t)i0 * 2], ay = pts[(int64_t)i0 * 2 + 1];
        const double dx = pts[(int64_t)i1 * 2] - ax, dy = pts[(int64_t)i1 * 2 + 1] - ay;
        const double ln = sqrt(dx * dx + dy * dy);
        if (ln > 0.0) {
            m[0] = -dy / ln;
            m[1] = dx / ln;
            m[2] = -(m[0] * ax + m[1] * ay);
        } else {
            m[0] = m[1] = m[2] = nan;
        }
    } else {
        const double* a = pts + (int64_t)i0 * 4;
        const double* b = pts + (int64_t)i1 * 4;
        double l0[3], l1[3], v[3];
        cross3(a[0], a[1], 1.0, a[2], a[3], 1.0, l0);
        cross3(b[0], b[1], 1.0, b[2], b[3], 1.0, l1);
        cross3(l0[0], l0[1], l0[2], l1[0], l1[1], l1[2], v);
        const double ln = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (ln > 0.0) { m[0] = v[0] / ln; m[1] = v[1] / ln; m[2] = v[2] / ln; }
        else { m[0] = m[1] = m[2] = nan; }
    }
    models[(int64_t)s * 3] = m[0];
    models[(int64_t)s * 3 + 1] = m[1];
    models[(int64_t)s * 3 + 2] = m[2];
}

}  // namespace

int solve_minimal_launch(pgx_ctx* ctx, const int32_t* samples, int S, double* models_out)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: points not set");
    if (!samples || S <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: empty sample batch");
    if (ctx->model_type != kLine2D && ctx->model_type != kVanishingPoint)
        return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: only the 2-point line and the 2-segment vanishing point solvers run on the device (model type %d)", ctx->model_type);
    ctx->Mpad = ((S + 255) / 256) * 256;
    PGX_TRY(ensure(ctx, ctx->models, (size_t)S * 3 * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
    PGX_TRY(ensure(ctx, ctx->scratch, (size_t)S * 2 * sizeof(int32_t)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, samples, (size_t)S * 2 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)((ctx->Mpad + kSolveBlock - 1) / kSolveBlock);
    if (ctx->model_type == kLine2D)
        hipLaunchKernelGGL((solve_kernel<kLine2D>), dim3(blocks), dim3(kSolveBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
    else
        hipLaunchKernelGGL((solve_kernel<kVanishingPoint>), dim3(blocks), dim3(kSolveBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
    PGX_HIP(ctx, hipGetLastError());
    if (models_out)
        PGX_HIP(ctx, hipMemcpyAsync(models_out, ctx->models.p, (size_t)S * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->M = S;
    return PGX_OK;
}

}  // namespace pgx
