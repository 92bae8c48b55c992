// solve.hip — batched minimal solvers on the resident points (first slice): the two solvers with
// an exact in-tree or closed-form specification.  The hypotheses are generated straight into the resident hypothesis
// buffer, so scoring them needs no host -> device model upload.
//
// Replaces, per sample of the proposal engine's main loop (gcransac::GCRANSAC::run, graph-cut-ransac submodule, absent):
//   vanishing point from two segments   /root/reference/src/pyprogressivex/include/solver_vanishing_point_two_lines.h:147-185
//                                       (l_i = e_i0 x e_i1 :174-179, v = l_0 x l_1 :180-182, vec_norm :123-131)
//   2D line through two points          Default2DLineEstimator's minimal solver (progressivex_python.cpp:489), absent
//                                       upstream [U-4]: unit normal (-dy, dx) / |d|, offset c = -(n . a)
//   fundamental matrix from 7 points    DefaultFundamentalMatrixEstimator's minimal solver (progressivex_python.cpp:616), absent
//                                       upstream: null space of the 7x9 epipolar system by Gauss-Jordan elimination with
//                                       full pivoting, det(l F1 + (1-l) F2) = 0 as a cubic, real roots by bisection of
//                                       one root + the quadratic factor (only + - * / sqrt: identical on CPU and GPU),
//                                       up to three models per sample (three slots, NaN = no root)
//   homography from 4 correspondences   DefaultHomographyEstimator's minimal solver (progressivex_python.cpp:252), absent
//                                       upstream: h33 = 1, the 8x8 DLT system of the isotropically scaled points by
//                                       Gaussian elimination with partial pivoting (rank test 1e-12), scaling undone
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

// ---- 4-point homography -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void solve_h4_kernel(const double* __restrict__ pts, int64_t n, const int* __restrict__ samples,
                                                      int S, double scale, double* __restrict__ models, int* __restrict__ perm,
                                                      int Mpad)
{
    const int s = (int)(blockIdx.x * 64 + threadIdx.x);
    for (int t = s; t < Mpad; t += (int)(gridDim.x * 64)) perm[t] = t < S ? t : 0;
    if (s >= S) return;
    const double nan = __builtin_nan("");
    double* out = models + (int64_t)s * 9;
    for (int k = 0; k < 9; ++k) out[k] = nan;
    double M[8][9];  // rows 0..3: x-equations of the four points, rows 4..7: y-equations; column 8 = right-hand side
    for (int r = 0; r < 4; ++r) {
        const int i = samples[4 * s + r];
        if (i < 0 || i >= n) return;
        const double x1 = pts[(int64_t)i * 4] / scale, y1 = pts[(int64_t)i * 4 + 1] / scale;
        const double x2 = pts[(int64_t)i * 4 + 2] / scale, y2 = pts[(int64_t)i * 4 + 3] / scale;
        double* a = M[r];
        double* b = M[r + 4];
        a[0] = -x1; a[1] = -y1; a[2] = -1.0; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; a[6] = x2 * x1; a[7] = x2 * y1; a[8] = -x2;
        b[0] = 0.0; b[1] = 0.0; b[2] = 0.0; b[3] = -x1; b[4] = -y1; b[5] = -1.0; b[6] = y2 * x1; b[7] = y2 * y1; b[8] = -y2;
    }
    for (int c = 0; c < 8; ++c) {
        int pr = c;
        double best = fabs(M[c][c]);
        for (int i = c + 1; i < 8; ++i) { const double a = fabs(M[i][c]); if (a > best) { best = a; pr = i; } }
        if (!(best >= 1e-12)) return;  // degenerate sample (three collinear points, repeated point) or NaN
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
    h[2] = h[2] * scale; h[5] = h[5] * scale; h[6] = h[6] / scale; h[7] = h[7] / scale;  // H = S^-1 Hn S, S = diag(1/s, 1/s, 1)
    for (int k = 0; k < 9; ++k) if (!(fabs(h[k]) < 1e300)) return;
    for (int k = 0; k < 9; ++k) out[k] = h[k];
}

// ---- 7-point fundamental matrix -------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double* a)
{
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

__global__ __launch_bounds__(64) void solve_f7_kernel(const double* __restrict__ pts, int64_t n, const int* __restrict__ samples,
                                                      int S, double scale, double* __restrict__ models, int* __restrict__ perm,
                                                      int Mpad)
{
    const int s = (int)(blockIdx.x * 64 + threadIdx.x);
    for (int t = s; t < Mpad; t += (int)(gridDim.x * 64)) perm[t] = t < 3 * S ? t : 0;
    if (s >= S) return;
    const double nan = __builtin_nan("");
    double* out = models + (int64_t)s * 27;
    for (int k = 0; k < 27; ++k) out[k] = nan;
    double M[7][9];
    for (int r = 0; r < 7; ++r) {
        const int i = samples[7 * s + r];
        if (i < 0 || i >= n) return;
        const double x1 = pts[(int64_t)i * 4] / scale, y1 = pts[(int64_t)i * 4 + 1] / scale;
        const double x2 = pts[(int64_t)i * 4 + 2] / scale, y2 = pts[(int64_t)i * 4 + 3] / scale;
        M[r][0] = x2 * x1; M[r][1] = x2 * y1; M[r][2] = x2;
        M[r][3] = y2 * x1; M[r][4] = y2 * y1; M[r][5] = y2;
        M[r][6] = x1; M[r][7] = y1; M[r][8] = 1.0;
    }
    int col[9];
    for (int j = 0; j < 9; ++j) col[j] = j;
    for (int r = 0; r < 7; ++r) {  // Gauss-Jordan with full pivoting (first maximum in row-major order)
        int pr = r, pc = r;
        double best = -1.0;
        for (int i = r; i < 7; ++i)
            for (int j = r; j < 9; ++j) {
                const double a = fabs(M[i][j]);
                if (a > best) { best = a; pr = i; pc = j; }
            }
        if (!(best >= 1e-12)) return;  // rank deficient sample (or NaN)
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
    double F1[9], F2[9];  // null vectors: free variable col[7] (resp. col[8]) = 1
    for (int k = 0; k < 7; ++k) { F1[col[k]] = -M[k][7]; F2[col[k]] = -M[k][8]; }
    F1[col[7]] = 1.0; F1[col[8]] = 0.0;
    F2[col[7]] = 0.0; F2[col[8]] = 1.0;
    // det(F2 + l D), D = F1 - F2:  c3 l^3 + c2 l^2 + c1 l + c0
    double D[9], T[9];
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
    double roots[3] = {nan, nan, nan};
    const double cm = fmax(fmax(fabs(c0), fabs(c1)), fmax(fabs(c2), fabs(c3)));
    if (!(cm > 0.0) || !(cm < 1e300)) return;
    if (fabs(c3) > 1e-14 * cm) {
        const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
        double lo = -(1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)))), hi = -lo;  // Cauchy bound: p(lo) < 0 < p(hi)
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            const double pv = ((mid + a) * mid + b) * mid + c;
            if (pv < 0.0) lo = mid; else hi = mid;
        }
        const double l0 = 0.5 * (lo + hi);
        roots[0] = l0;
        const double qb = a + l0, qc = b + qb * l0;  // p(l) = (l - l0)(l^2 + qb l + qc)
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
        // undo the isotropic scaling: F <- diag(1/s, 1/s, 1) F diag(1/s, 1/s, 1)
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

}  // namespace

int solve_minimal_launch(pgx_ctx* ctx, const int32_t* samples, int S, double* models_out)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: points not set");
    if (!samples || S <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: empty sample batch");
    if (ctx->model_type == kFundamental) {
        // three model slots per sample; isotropic pre-scaling by the largest coordinate magnitude (pmax of set_points is 1
        // for this model type, so it is recomputed here from the caller-visible data: umax is not kept either)
        if (!(ctx->fscale >= 1.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: coordinate scale not available");
        const int Mtot = 3 * S;
        ctx->Mpad = ((Mtot + 255) / 256) * 256;
        PGX_TRY(ensure(ctx, ctx->models, (size_t)Mtot * 9 * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
        PGX_TRY(ensure(ctx, ctx->scratch, (size_t)S * 7 * sizeof(int32_t)));
        PGX_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, samples, (size_t)S * 7 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        const unsigned blocks = (unsigned)((S + 63) / 64);
        hipLaunchKernelGGL(solve_f7_kernel, dim3(blocks), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->fscale, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
        PGX_HIP(ctx, hipGetLastError());
        if (models_out)
            PGX_HIP(ctx, hipMemcpyAsync(models_out, ctx->models.p, (size_t)Mtot * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->M = Mtot;
        return PGX_OK;
    }
    if (ctx->model_type == kHomography) {
        if (!(ctx->fscale >= 1.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: coordinate scale not available");
        ctx->Mpad = ((S + 255) / 256) * 256;
        PGX_TRY(ensure(ctx, ctx->models, (size_t)S * 9 * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
        PGX_TRY(ensure(ctx, ctx->scratch, (size_t)S * 4 * sizeof(int32_t)));
        PGX_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, samples, (size_t)S * 4 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(solve_h4_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->fscale, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
        PGX_HIP(ctx, hipGetLastError());
        if (models_out)
            PGX_HIP(ctx, hipMemcpyAsync(models_out, ctx->models.p, (size_t)S * 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->M = S;
        return PGX_OK;
    }
    if (ctx->model_type != kLine2D && ctx->model_type != kVanishingPoint)
        return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: no device solver for model type %d yet (built: 2-point line, 2-segment vanishing point, 4-point homography, 7-point fundamental matrix)", ctx->model_type);
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
