// fit.hip — accumulation pass of the non-minimal refits: weighted Gram matrices of per-point
// design rows over a subset of the resident points.  The small dense solve (2x2 .. 9x9 eigen / 6x6 linear) stays on the
// host, as SURVEY a9 prescribes ("GPU does bucketing + sums, CPU does the solve").
//
// Replaces the data pass of estimator.estimateModelNonminimal(...) as called by
//   pearl::PEARL::parameterEstimation   /root/reference/src/pyprogressivex/include/PEARL.h:374-380
//   GC-RANSAC's local optimisation      (graph-cut-ransac submodule, absent from the snapshot)
// for the five estimators of progressivex_python.cpp:119,252,343,489,616.  Only the vanishing-point rows have an in-tree
// specification (solver_vanishing_point_two_lines.h:212-218: A = [y0*mz-my, mx-x0*mz, x0*my-y0*mx] * w); the other
// solvers are absent upstream and restated from the literature (normalised DLT, normalised 8-point, total least squares
// line, Gauss-Newton on the reprojection error) — DESIGN.md §3.
//
//   out = sum over selected points i of  W_i * sum over the rows a of point i of  a a^T      (upper triangle, row-major)
//   W_i = w_i^wpow (weights optional).  Selection: an uploaded index list, or label == k on the resident labelling.
// Fixed reduction tree (lanes -> waves -> per-block partials -> one final block): bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "pgx_internal.h"

namespace pgx {

namespace {

constexpr int kFitBlock = 256;

struct FitParams {
    double v[12];
};

template <int Q>
struct Acc {
    double s[Q * (Q + 1) / 2];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int k = 0; k < Q * (Q + 1) / 2; ++k) s[k] = 0.0;
    }
    __device__ __forceinline__ void add(const double (&a)[Q], double w)
    {
        int k = 0;
#pragma unroll
        for (int r = 0; r < Q; ++r) {
            const double wa = w * a[r];
#pragma unroll
            for (int c = r; c < Q; ++c) s[k++] += wa * a[c];
        }
    }
};

// row generators: Q = row length, emit(pt, prm, acc, w, bad)
struct GenAffine2 { static constexpr int Q = 3, D = 2; };
struct GenAffine4 { static constexpr int Q = 5, D = 4; };
struct GenAffine5 { static constexpr int Q = 6, D = 5; };
struct GenDltH { static constexpr int Q = 9, D = 4; };
struct GenEpiF { static constexpr int Q = 9, D = 4; };
struct GenVp { static constexpr int Q = 3, D = 4; };
struct GenPnpGn { static constexpr int Q = 7, D = 5; };

template <class G>
__device__ __forceinline__ void emit(const double* pt, const FitParams& prm, Acc<G::Q>& acc, double w, int& bad);

template <>
__device__ __forceinline__ void emit<GenAffine2>(const double* pt, const FitParams&, Acc<3>& acc, double w, int&)
{
    const double a[3] = {1.0, pt[0], pt[1]};
    acc.add(a, w);
}
template <>
__device__ __forceinline__ void emit<GenAffine4>(const double* pt, const FitParams&, Acc<5>& acc, double w, int&)
{
    const double a[5] = {1.0, pt[0], pt[1], pt[2], pt[3]};
    acc.add(a, w);
}
template <>
__device__ __forceinline__ void emit<GenAffine5>(const double* pt, const FitParams&, Acc<6>& acc, double w, int&)
{
    const double a[6] = {1.0, pt[0], pt[1], pt[2], pt[3], pt[4]};
    acc.add(a, w);
}
// prm = (s1, cx1, cy1, s2, cx2, cy2): Hartley-normalised coordinates
template <>
__device__ __forceinline__ void emit<GenDltH>(const double* pt, const FitParams& prm, Acc<9>& acc, double w, int&)
{
    const double x1 = (pt[0] - prm.v[1]) * prm.v[0], y1 = (pt[1] - prm.v[2]) * prm.v[0];
    const double x2 = (pt[2] - prm.v[4]) * prm.v[3], y2 = (pt[3] - prm.v[5]) * prm.v[3];
    const double r1[9] = {-x1, -y1, -1.0, 0.0, 0.0, 0.0, x2 * x1, x2 * y1, x2};
    const double r2[9] = {0.0, 0.0, 0.0, -x1, -y1, -1.0, y2 * x1, y2 * y1, y2};
    acc.add(r1, w);
    acc.add(r2, w);
}
template <>
__device__ __forceinline__ void emit<GenEpiF>(const double* pt, const FitParams& prm, Acc<9>& acc, double w, int&)
{
    const double x1 = (pt[0] - prm.v[1]) * prm.v[0], y1 = (pt[1] - prm.v[2]) * prm.v[0];
    const double x2 = (pt[2] - prm.v[4]) * prm.v[3], y2 = (pt[3] - prm.v[5]) * prm.v[3];
    const double a[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    acc.add(a, w);
}
// solver_vanishing_point_two_lines.h:212-217
template <>
__device__ __forceinline__ void emit<GenVp>(const double* pt, const FitParams&, Acc<3>& acc, double w, int&)
{
    const double x0 = pt[0], y0 = pt[1], x1 = pt[2], y1 = pt[3];
    const double mx = (x0 + x1) / 2.0, my = (y0 + y1) / 2.0, mz = 1.0;
    const double a[3] = {y0 * mz - my, mx - x0 * mz, x0 * my - y0 * mx};
    acc.add(a, w);
}
// prm = P = [R | t] row-major 3x4; rows (J_u, r_u), (J_v, r_v) with J = d proj / d (omega, t) at omega = 0
template <>
__device__ __forceinline__ void emit<GenPnpGn>(const double* pt, const FitParams& prm, Acc<7>& acc, double w, int& bad)
{
    const double* P = prm.v;
    const double X = pt[2], Y = pt[3], Z = pt[4];
    const double rx = P[0] * X + P[1] * Y + P[2] * Z, ry = P[4] * X + P[5] * Y + P[6] * Z, rz = P[8] * X + P[9] * Y + P[10] * Z;
    const double xc = rx + P[3], yc = ry + P[7], zc = rz + P[11];
    if (!(fabs(zc) >= 1e-12)) { bad = 1; return; }
    const double inv = 1.0 / zc;
    const double du = xc * inv - pt[0], dv = yc * inv - pt[1];
    // d proj / d Xc = [[inv, 0, -xc inv^2], [0, inv, -yc inv^2]];  d Xc / d omega = -[R X]_x, d Xc / d t = I
    const double a = inv, b = -xc * inv * inv, c = -yc * inv * inv;
    // skew(RX) as used by the host restatement: rows (0, rz, -ry), (-rz, 0, rx), (ry, -rx, 0)
    const double ju[7] = {b * ry, a * rz + b * (-rx), a * (-ry), a, 0.0, b, du};
    const double jv[7] = {a * (-rz) + c * ry, c * (-rx), a * rx, 0.0, a, c, dv};
    acc.add(ju, w);
    acc.add(jv, w);
}

template <class G>
__global__ __launch_bounds__(kFitBlock) void gram_kernel(const double* __restrict__ pts, int64_t n, FitParams prm,
                                                         const int* __restrict__ index, int64_t m,
                                                         const int* __restrict__ labels, int label,
                                                         const double* __restrict__ weights, int wpow,
                                                         double* __restrict__ partials, int* __restrict__ counters)
{
    constexpr int Q = G::Q, NV = Q * (Q + 1) / 2;
    __shared__ double lds[(kFitBlock / 64) * NV];
    __shared__ int s_cnt, s_bad;
    if (threadIdx.x == 0) { s_cnt = 0; s_bad = 0; }
    __syncthreads();
    Acc<Q> acc;
    acc.zero();
    const int64_t t = (int64_t)blockIdx.x * kFitBlock + threadIdx.x;
    int64_t i = -1;
    if (index != nullptr) { if (t < m) i = index[t]; }
    else if (t < n && labels[t] == label) i = t;
    int bad = 0;
    if (i >= 0) {
        double pt[G::D];
#pragma unroll
        for (int k = 0; k < G::D; ++k) pt[k] = pts[i * G::D + k];
        double w = 1.0;
        if (weights != nullptr) { w = weights[i]; if (wpow == 2) w = w * w; }
        emit<G>(pt, prm, acc, w, bad);
        atomicAdd(&s_cnt, 1);
        if (bad) atomicAdd(&s_bad, 1);
    }
    // fixed tree: lanes (shuffle), then waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double x = acc.s[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) lds[wave * NV + k] = x;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += kFitBlock) {
        double s = lds[k];
        for (int w2 = 1; w2 < kFitBlock / 64; ++w2) s += lds[w2 * NV + k];
        partials[(int64_t)blockIdx.x * NV + k] = s;
    }
    if (threadIdx.x == 0) {
        if (s_cnt) atomicAdd(&counters[0], s_cnt);
        if (s_bad) atomicAdd(&counters[1], s_bad);
    }
}

// one block: value k is summed by 16 lanes (lane j takes the blocks b = j mod 16 in order), then a fixed xor tree
__global__ __launch_bounds__(1024) void gram_final_kernel(const double* __restrict__ partials, int blocks, int nv,
                                                          double* __restrict__ out)
{
    const int k = (int)threadIdx.x >> 4, j = (int)threadIdx.x & 15;
    double s = 0.0;
    if (k < nv)
        for (int b = j; b < blocks; b += 16) s += partials[(int64_t)b * nv + k];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (k < nv && j == 0) out[k] = s;
}

template <class G>
void launch(pgx_ctx* ctx, const FitParams& prm, const int* index, int64_t m, int label, const double* weights, int wpow,
            int blocks, double* partials, int* counters)
{
    hipLaunchKernelGGL((gram_kernel<G>), dim3((unsigned)blocks), dim3(kFitBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                       prm, index, m, ctx->labels.as<int>(), label, weights, wpow, partials, counters);
}

// All labels in one launch (PEARL::parameterEstimation refits every instance per iteration): blockIdx.y = label, one
// parameter block per label.  Per-block tree and final pass are those of gram_kernel / gram_final_kernel, so out[k] is
// bit-identical to the single-label call with label k.
template <class G>
__global__ __launch_bounds__(kFitBlock) void gram_labels_kernel(const double* __restrict__ pts, int64_t n, const double* __restrict__ prm_k,
                                                                const int* __restrict__ labels, const double* __restrict__ weights,
                                                                int wpow, int blocks, double* __restrict__ partials,
                                                                int* __restrict__ counters)
{
    constexpr int Q = G::Q, NV = Q * (Q + 1) / 2;
    __shared__ double lds[(kFitBlock / 64) * NV];
    __shared__ int s_cnt, s_bad;
    if (threadIdx.x == 0) { s_cnt = 0; s_bad = 0; }
    __syncthreads();
    const int label = (int)blockIdx.y;
    FitParams prm;
#pragma unroll
    for (int k = 0; k < 12; ++k) prm.v[k] = prm_k[(int64_t)label * 12 + k];
    Acc<Q> acc;
    acc.zero();
    const int64_t t = (int64_t)blockIdx.x * kFitBlock + threadIdx.x;
    const int64_t i = (t < n && labels[t] == label) ? t : -1;
    int bad = 0;
    if (i >= 0) {
        double pt[G::D];
#pragma unroll
        for (int k = 0; k < G::D; ++k) pt[k] = pts[i * G::D + k];
        double w = 1.0;
        if (weights != nullptr) { w = weights[i]; if (wpow == 2) w = w * w; }
        emit<G>(pt, prm, acc, w, bad);
        atomicAdd(&s_cnt, 1);
        if (bad) atomicAdd(&s_bad, 1);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double x = acc.s[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) lds[wave * NV + k] = x;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += kFitBlock) {
        double s = lds[k];
        for (int w2 = 1; w2 < kFitBlock / 64; ++w2) s += lds[w2 * NV + k];
        partials[((int64_t)label * blocks + blockIdx.x) * NV + k] = s;
    }
    if (threadIdx.x == 0) {
        if (s_cnt) atomicAdd(&counters[2 * label], s_cnt);
        if (s_bad) atomicAdd(&counters[2 * label + 1], s_bad);
    }
}

__global__ __launch_bounds__(1024) void gram_final_labels_kernel(const double* __restrict__ partials, int blocks, int nv,
                                                                 double* __restrict__ out)
{
    const double* part = partials + (int64_t)blockIdx.x * blocks * nv;
    const int k = (int)threadIdx.x >> 4, j = (int)threadIdx.x & 15;
    double s = 0.0;
    if (k < nv)
        for (int b = j; b < blocks; b += 16) s += part[(int64_t)b * nv + k];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (k < nv && j == 0) out[(int64_t)blockIdx.x * nv + k] = s;
}

template <class G>
void launch_labels(pgx_ctx* ctx, int K, const double* prm, const double* weights, int wpow, int blocks, double* partials, int* counters)
{
    hipLaunchKernelGGL((gram_labels_kernel<G>), dim3((unsigned)blocks, (unsigned)K), dim3(kFitBlock), 0, ctx->stream,
                       ctx->pts.as<double>(), ctx->n, prm, ctx->labels.as<int>(), weights, wpow, blocks, partials, counters);
}

// Batched variant for the inner RANSAC of the local optimisation: B small index selections of m points
// each, one wave per selection (m <= 64 in practice: 7 x the minimal sample size), per-selection parameter blocks.
// Lanes take the points t = lane, lane + 64, ..; fixed shuffle tree: bit-reproducible.
template <class G>
__global__ __launch_bounds__(64) void gram_batch_kernel(const double* __restrict__ pts, const double* __restrict__ prm,
                                                        const int* __restrict__ index, int m,
                                                        const double* __restrict__ wsel, int wpow,
                                                        double* __restrict__ out, int* __restrict__ bad)
{
    constexpr int Q = G::Q, NV = Q * (Q + 1) / 2;
    const int b = blockIdx.x;
    FitParams p;
#pragma unroll
    for (int k = 0; k < 12; ++k) p.v[k] = prm[(int64_t)b * 12 + k];
    Acc<Q> acc;
    acc.zero();
    int nbad = 0;
    for (int t = threadIdx.x; t < m; t += 64) {
        const int64_t i = index[(int64_t)b * m + t];
        double pt[G::D];
#pragma unroll
        for (int k = 0; k < G::D; ++k) pt[k] = pts[i * G::D + k];
        double w = 1.0;
        if (wsel != nullptr) { w = wsel[(int64_t)b * m + t]; if (wpow == 2) w = w * w; }
        int bd = 0;
        emit<G>(pt, p, acc, w, bd);
        nbad += bd;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double x = acc.s[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (threadIdx.x == 0) out[(int64_t)b * NV + k] = x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nbad += __shfl_down(nbad, off, 64);
    if (threadIdx.x == 0) bad[b] = nbad;
}

template <class G>
void launch_batch(pgx_ctx* ctx, int B, const double* prm, const int* index, int m, const double* wsel, int wpow, double* out,
                  int* bad)
{
    hipLaunchKernelGGL((gram_batch_kernel<G>), dim3((unsigned)B), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), prm, index, m,
                       wsel, wpow, out, bad);
}

// ---- Gauss-Newton pose refits of a whole batch in ONE launch --------------------------------------------------------------
// The local optimisation refits ~50 selections of 21 points per graph-cut round; each Gauss-Newton step used to be one
// gram_batch launch, a copy back, a stacked 6x6 pseudo-inverse on the host and a copy up (10 steps per round: a third of
// find6DPoses' proposal time at C4).  Here one wave owns one selection for all its steps: the normal equations by the same
// rows and the same shuffle tree as gram_batch_kernel (bitwise the same sums), broadcast to every lane, and each lane
// redundantly runs the small dense part - pseudo-inverse of the symmetric 6x6 through a cyclic Jacobi eigen-decomposition with
// numpy.linalg.pinv's cut-off (|lambda| <= rcond max|lambda| dropped), Rodrigues update of R, t += dt - exactly the iteration
// of pyprogressivex/_estimators.py PnPEstimator._fit_many, whose iterates it reproduces up to rounding (tests: 1e-9).

// x = pinv(A) b for symmetric A (6x6).  Cyclic Jacobi, fixed rotation order, at most 12 sweeps (it converges quadratically:
// 5-7 sweeps reach the rounding floor).
__device__ __forceinline__ void sym6_pinv_apply(const double (&A0)[6][6], const double (&b)[6], double rcond, double (&x)[6])
{
    double A[6][6], V[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) { A[r][c] = A0[r][c]; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            dia += A[r][r] * A[r][r];
#pragma unroll
            for (int c = r + 1; c < 6; ++c) off += A[r][c] * A[r][c];
        }
        if (!(off > 1e-36 * dia)) break;   // (also leaves on NaN)
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                for (int k = 0; k < 6; ++k) {   // columns p, q
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {   // rows p, q
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    double smax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) smax = fmax(smax, fabs(A[i][i]));
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double lam = A[i][i];
        if (!(fabs(lam) > rcond * smax)) continue;
        double vb = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) vb += V[k][i] * b[k];
        const double f = vb / lam;
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] += f * V[k][i];
    }
}

__global__ __launch_bounds__(64) void pnp_refine_batch_kernel(const double* __restrict__ pts, const double* __restrict__ inits,
                                                              const int* __restrict__ index, int m, const double* __restrict__ wsel, int wpow,
                                                              int iterations, double* __restrict__ out, int* __restrict__ status)
{
    using G = GenPnpGn;
    constexpr int Q = 7, NV = 28;
    const int b = blockIdx.x;
    FitParams p;
#pragma unroll
    for (int k = 0; k < 12; ++k) p.v[k] = inits[(int64_t)b * 12 + k];
    bool failed = m < 4;
    for (int it = 0; it < iterations && !failed; ++it) {
        Acc<Q> acc;
        acc.zero();
        int nbad = 0;
        for (int t = threadIdx.x; t < m; t += 64) {
            const int64_t i = index[(int64_t)b * m + t];
            double pt[G::D];
#pragma unroll
            for (int k = 0; k < G::D; ++k) pt[k] = pts[i * G::D + k];
            double w = 1.0;
            if (wsel != nullptr) { w = wsel[(int64_t)b * m + t]; if (wpow == 2) w = w * w; }
            int bd = 0;
            emit<G>(pt, p, acc, w, bd);
            nbad += bd;
        }
        double g[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double x = acc.s[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            g[k] = __shfl(x, 0, 64);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nbad += __shfl_down(nbad, off, 64);
        nbad = __shfl(nbad, 0, 64);
        if (nbad > 0) { failed = true; break; }
        // upper triangle, row-major: (r, c) at r * 7 - r (r - 1) / 2 + (c - r)
        double A[6][6], rhs[6];
        bool fin = true;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) {
                const double v = g[r * 7 - r * (r - 1) / 2 + (c - r)];
                A[r][c] = v; A[c][r] = v;
                fin = fin && isfinite(v);
            }
            rhs[r] = -g[r * 7 - r * (r - 1) / 2 + (6 - r)];
            fin = fin && isfinite(rhs[r]);
        }
        if (!fin) { failed = true; break; }
        double d[6];
        sym6_pinv_apply(A, rhs, 6.0 * 2.220446049250313e-16, d);
        const double ang = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (ang > 0.0) {   // R <- exp([omega]_x) R
            const double kx = d[0] / ang, ky = d[1] / ang, kz = d[2] / ang;
            const double K[3][3] = {{0.0, -kz, ky}, {kz, 0.0, -kx}, {-ky, kx, 0.0}};
            const double sn = sin(ang), cs = 1.0 - cos(ang);
            double rot[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double kk = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) kk += K[r][j] * K[j][c];
                    rot[r][c] = (r == c ? 1.0 : 0.0) + sn * K[r][c] + cs * kk;
                }
            double Rn[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) v += rot[r][j] * p.v[j * 4 + c];
                    Rn[r][c] = v;
                }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) p.v[r * 4 + c] = Rn[r][c];
        }
        p.v[3] += d[3]; p.v[7] += d[4]; p.v[11] += d[5];
        double nd = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) nd += d[k] * d[k];
        if (sqrt(nd) < 1e-12) break;
    }
    if (threadIdx.x == 0) {
        bool fin = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) { out[(int64_t)b * 12 + k] = p.v[k]; fin = fin && isfinite(p.v[k]); }
        status[b] = (!failed && fin) ? 1 : 0;
    }
}

}  // namespace

int pnp_refine_batch_launch(pgx_ctx* ctx, const double* inits, const int32_t* index, int B, int m, const double* wsel, int wpow,
                            int iterations, double* out, int32_t* status)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_pnp_refine_batch: points not set");
    if (ctx->D != 5) return fail(ctx, PGX_ERR_INVALID, "pgx_pnp_refine_batch: needs 5-D 2D-3D rows");
    if (wpow != 1 && wpow != 2) return fail(ctx, PGX_ERR_INVALID, "pgx_pnp_refine_batch: weight power must be 1 or 2");
    if (!inits || !out || !status || B < 0 || m < 0 || iterations < 0 || ((int64_t)B * m > 0 && !index))
        return fail(ctx, PGX_ERR_INVALID, "pgx_pnp_refine_batch: bad argument");
    if (B == 0) return PGX_OK;
    const int64_t tot = (int64_t)B * m;
    for (int64_t t = 0; t < tot; ++t)
        if (index[t] < 0 || index[t] >= ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_pnp_refine_batch: index %d out of range", index[t]);
    // scratch: inits[B][12] | out[B][12] | wsel[B][m] | index[B][m] | status[B]
    const size_t prm_bytes = (size_t)B * 12 * 8, w_bytes = wsel ? (size_t)tot * 8 : 0;
    const size_t idx_bytes = ((size_t)tot * 4 + 7) & ~(size_t)7, st_bytes = (size_t)B * 4;
    PGX_TRY(ensure(ctx, ctx->fit_scratch, 2 * prm_bytes + w_bytes + idx_bytes + st_bytes + 64));
    char* base = (char*)ctx->fit_scratch.p;
    double* d_in = (double*)base;
    double* d_out = (double*)(base + prm_bytes);
    double* d_w = (double*)(base + 2 * prm_bytes);
    int* d_idx = (int*)(base + 2 * prm_bytes + w_bytes);
    int* d_st = (int*)(base + 2 * prm_bytes + w_bytes + idx_bytes);
    PGX_HIP(ctx, hipMemcpyAsync(d_in, inits, prm_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (tot > 0) PGX_HIP(ctx, hipMemcpyAsync(d_idx, index, (size_t)tot * 4, hipMemcpyHostToDevice, ctx->stream));
    if (w_bytes) PGX_HIP(ctx, hipMemcpyAsync(d_w, wsel, w_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pnp_refine_batch_kernel, dim3((unsigned)B), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), d_in, d_idx, m,
                       wsel ? d_w : nullptr, wpow, iterations, d_out, d_st);
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, out, d_out, prm_bytes));
    PGX_TRY(d2h(ctx, status, d_st, st_bytes));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

static int gram_row_length(pgx_ctx* ctx, const char* who, int kind, int nparams, int* q)
{
    const int D = ctx->D;
    switch (kind) {
    case PGX_GRAM_AFFINE: *q = D + 1; if (D != 2 && D != 4 && D != 5) return fail(ctx, PGX_ERR_INVALID, "%s: affine rows need 2-, 4- or 5-D points", who); break;
    case PGX_GRAM_DLT_H: case PGX_GRAM_EPI_F: *q = 9; if (D != 4 || nparams != 6) return fail(ctx, PGX_ERR_INVALID, "%s: needs 4-D correspondences and 6 normalisation parameters", who); break;
    case PGX_GRAM_VP: *q = 3; if (D != 4) return fail(ctx, PGX_ERR_INVALID, "%s: needs 4-D segments", who); break;
    case PGX_GRAM_PNP_GN: *q = 7; if (D != 5 || nparams != 12) return fail(ctx, PGX_ERR_INVALID, "%s: needs 5-D 2D-3D rows and a 3x4 pose", who); break;
    default: return fail(ctx, PGX_ERR_INVALID, "%s: unknown row kind %d", who, kind);
    }
    return PGX_OK;
}

int gram_batch_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, const int32_t* index, int B, int m,
                      const double* wsel, int wpow, double* out, int32_t* bad)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_batch: points not set");
    if (wpow != 1 && wpow != 2) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_batch: weight power must be 1 or 2");
    if (nparams < 0 || nparams > 12 || (nparams > 0 && !params)) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_batch: bad parameter block");
    if (!out || B < 0 || m < 0 || ((int64_t)B * m > 0 && !index)) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_batch: bad argument");
    int q = 0;
    PGX_TRY(gram_row_length(ctx, "pgx_gram_batch", kind, nparams, &q));
    const int nv = q * (q + 1) / 2;
    if (B == 0) return PGX_OK;
    const int64_t tot = (int64_t)B * m;
    for (int64_t t = 0; t < tot; ++t)
        if (index[t] < 0 || index[t] >= ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_batch: index %d out of range", index[t]);
    // scratch: prm[B][12] | out[B][nv] | wsel[B][m] | index[B][m] | bad[B]
    const size_t prm_bytes = (size_t)B * 12 * 8, out_bytes = (size_t)B * nv * 8, w_bytes = wsel ? (size_t)tot * 8 : 0;
    const size_t idx_bytes = ((size_t)tot * 4 + 7) & ~(size_t)7, bad_bytes = (size_t)B * 4;
    PGX_TRY(ensure(ctx, ctx->fit_scratch, prm_bytes + out_bytes + w_bytes + idx_bytes + bad_bytes + 64));
    char* base = (char*)ctx->fit_scratch.p;
    double* d_prm = (double*)base;
    double* d_out = (double*)(base + prm_bytes);
    double* d_w = (double*)(base + prm_bytes + out_bytes);
    int* d_idx = (int*)(base + prm_bytes + out_bytes + w_bytes);
    int* d_bad = (int*)(base + prm_bytes + out_bytes + w_bytes + idx_bytes);
    std::vector<double> hp((size_t)B * 12, 0.0);
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < nparams; ++k) hp[(size_t)b * 12 + k] = params[(size_t)b * nparams + k];
    PGX_HIP(ctx, hipMemcpyAsync(d_prm, hp.data(), prm_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (tot > 0) PGX_HIP(ctx, hipMemcpyAsync(d_idx, index, (size_t)tot * 4, hipMemcpyHostToDevice, ctx->stream));
    if (w_bytes) PGX_HIP(ctx, hipMemcpyAsync(d_w, wsel, w_bytes, hipMemcpyHostToDevice, ctx->stream));
    const double* ww = wsel ? d_w : nullptr;
    const int D = ctx->D;
    switch (kind) {
    case PGX_GRAM_AFFINE:
        if (D == 2) launch_batch<GenAffine2>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad);
        else if (D == 4) launch_batch<GenAffine4>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad);
        else launch_batch<GenAffine5>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad);
        break;
    case PGX_GRAM_DLT_H: launch_batch<GenDltH>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad); break;
    case PGX_GRAM_EPI_F: launch_batch<GenEpiF>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad); break;
    case PGX_GRAM_VP: launch_batch<GenVp>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad); break;
    default: launch_batch<GenPnpGn>(ctx, B, d_prm, d_idx, m, ww, wpow, d_out, d_bad); break;
    }
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, out, d_out, out_bytes));
    if (bad) PGX_TRY(d2h(ctx, bad, d_bad, bad_bytes));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

// Per-point weights are RESIDENT (pgx_set_weights checks their length against n and uploads them once): the Gram calls only
// say whether to use them.  (They used to take a host pointer without a length and copied n doubles from it on every call.)
static int resident_weights(pgx_ctx* ctx, const char* who, int use_weights, const double** ww)
{
    *ww = nullptr;
    if (!use_weights) return PGX_OK;
    if (ctx->weights_n != ctx->n || !ctx->weights.p)
        return fail(ctx, PGX_ERR_INVALID, "%s: use_weights set but no weights are resident for the current points (pgx_set_weights)", who);
    *ww = ctx->weights.as<double>();
    return PGX_OK;
}

int gram_labels_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, int K, int use_weights, int wpow,
                       double* out, int64_t* count, int64_t* bad)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_labels: points not set");
    if (ctx->labels_n != ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_labels: labels not set");
    if (wpow != 1 && wpow != 2) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_labels: weight power must be 1 or 2");
    if (nparams < 0 || nparams > 12 || (nparams > 0 && !params)) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_labels: bad parameter block");
    if (!out || K <= 0 || K > 4096) return fail(ctx, PGX_ERR_INVALID, "pgx_gram_labels: bad argument");
    int q = 0;
    PGX_TRY(gram_row_length(ctx, "pgx_gram_labels", kind, nparams, &q));
    const int nv = q * (q + 1) / 2;
    const int blocks = (int)((ctx->n + kFitBlock - 1) / kFitBlock);
    const double* ww = nullptr;
    PGX_TRY(resident_weights(ctx, "pgx_gram_labels", use_weights, &ww));
    // scratch: partials[K][blocks][nv] | out[K][nv] | counters[2K] | prm[K][12]
    const size_t part_bytes = (size_t)K * blocks * nv * 8, out_bytes = (size_t)K * nv * 8, cnt_bytes = ((size_t)K * 8 + 15) & ~(size_t)15;
    const size_t prm_bytes = (size_t)K * 12 * 8;
    PGX_TRY(ensure(ctx, ctx->fit_scratch, part_bytes + out_bytes + cnt_bytes + prm_bytes + 64));
    char* base = (char*)ctx->fit_scratch.p;
    double* d_part = (double*)base;
    double* d_out = (double*)(base + part_bytes);
    int* d_cnt = (int*)(base + part_bytes + out_bytes);
    double* d_prm = (double*)(base + part_bytes + out_bytes + cnt_bytes);
    // counters (zero) | parameter blocks are adjacent: ONE upload clears the first and fills the second; result | counters are
    // adjacent too: ONE copy back, into pinned memory.  (A fill, two uploads' worth of commands and two blocking copies into pageable
    // memory before: a third of the call on a 300-point scene, scripts/bench_small_calls.py.)
    std::vector<double> hp(cnt_bytes / 8 + (size_t)K * 12, 0.0);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < nparams; ++j) hp[cnt_bytes / 8 + (size_t)k * 12 + j] = params[(size_t)k * nparams + j];
    PGX_HIP(ctx, hipMemcpyAsync(d_cnt, hp.data(), cnt_bytes + prm_bytes, hipMemcpyHostToDevice, ctx->stream));
    const int D = ctx->D;
    switch (kind) {
    case PGX_GRAM_AFFINE:
        if (D == 2) launch_labels<GenAffine2>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt);
        else if (D == 4) launch_labels<GenAffine4>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt);
        else launch_labels<GenAffine5>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt);
        break;
    case PGX_GRAM_DLT_H: launch_labels<GenDltH>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt); break;
    case PGX_GRAM_EPI_F: launch_labels<GenEpiF>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt); break;
    case PGX_GRAM_VP: launch_labels<GenVp>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt); break;
    default: launch_labels<GenPnpGn>(ctx, K, d_prm, ww, wpow, blocks, d_part, d_cnt); break;
    }
    PGX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(gram_final_labels_kernel, dim3((unsigned)K), dim3(1024), 0, ctx->stream, d_part, blocks, nv, d_out);
    PGX_HIP(ctx, hipGetLastError());
    void* hs = nullptr;
    PGX_TRY(host_staging(ctx, out_bytes + cnt_bytes, &hs));
    PGX_TRY(d2h(ctx, hs, d_out, out_bytes + (size_t)2 * K * sizeof(int)));
    PGX_TRY(sync_deliver(ctx));
    memcpy(out, hs, out_bytes);
    const int* cnt = (const int*)((const char*)hs + out_bytes);
    for (int k = 0; k < K; ++k) {
        if (count) count[k] = cnt[2 * k];
        if (bad) bad[k] = cnt[2 * k + 1];
    }
    return PGX_OK;
}

int gram_launch(pgx_ctx* ctx, int kind, const double* params, int nparams, int sel, const int32_t* index, int64_t m,
                int label, int use_weights, int wpow, double* out, int64_t* count, int64_t* bad)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: points not set");
    if (wpow != 1 && wpow != 2) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: weight power must be 1 or 2");
    if (nparams < 0 || nparams > 12 || (nparams > 0 && !params)) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: bad parameter block");
    if (!out) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: out is NULL");
    int q = 0;
    const int D = ctx->D;
    switch (kind) {
    case PGX_GRAM_AFFINE: q = D + 1; if (D != 2 && D != 4 && D != 5) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: affine rows need 2-, 4- or 5-D points"); break;
    case PGX_GRAM_DLT_H: case PGX_GRAM_EPI_F: q = 9; if (D != 4 || nparams != 6) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: needs 4-D correspondences and 6 normalisation parameters"); break;
    case PGX_GRAM_VP: q = 3; if (D != 4) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: needs 4-D segments"); break;
    case PGX_GRAM_PNP_GN: q = 7; if (D != 5 || nparams != 12) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: needs 5-D 2D-3D rows and a 3x4 pose"); break;
    default: return fail(ctx, PGX_ERR_INVALID, "pgx_gram: unknown row kind %d", kind);
    }
    const int nv = q * (q + 1) / 2;
    int64_t work = 0;
    if (sel == PGX_SEL_INDEX) {
        if (m < 0 || (m > 0 && !index)) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: index list missing");
        for (int64_t t = 0; t < m; ++t)
            if (index[t] < 0 || index[t] >= ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: index %d out of range", index[t]);
        work = m;
    } else if (sel == PGX_SEL_LABEL) {
        if (ctx->labels_n != ctx->n) return fail(ctx, PGX_ERR_INVALID, "pgx_gram: labels not set");
        work = ctx->n;
    } else {
        return fail(ctx, PGX_ERR_INVALID, "pgx_gram: unknown selection %d", sel);
    }
    const double* ww = nullptr;
    PGX_TRY(resident_weights(ctx, "pgx_gram", use_weights, &ww));
    FitParams prm;
    for (int k = 0; k < 12; ++k) prm.v[k] = k < nparams ? params[k] : 0.0;
    const int blocks = (int)((work + kFitBlock - 1) / kFitBlock);
    if (blocks == 0) {
        for (int k = 0; k < nv; ++k) out[k] = 0.0;
        if (count) *count = 0;
        if (bad) *bad = 0;
        return PGX_OK;
    }
    // scratch: partials | out | counters | index
    const size_t part_bytes = (size_t)blocks * nv * sizeof(double);
    const size_t idx_bytes = sel == PGX_SEL_INDEX ? (size_t)m * sizeof(int32_t) : 0;
    const size_t total = part_bytes + 64 * sizeof(double) + 64 + ((idx_bytes + 7) & ~(size_t)7);
    PGX_TRY(ensure(ctx, ctx->fit_scratch, total));
    char* base = (char*)ctx->fit_scratch.p;
    double* d_part = (double*)base;
    double* d_out = (double*)(base + part_bytes);
    int* d_cnt = (int*)(base + part_bytes + 64 * sizeof(double));
    int* d_idx = (int*)(base + part_bytes + 64 * sizeof(double) + 64);
    // counters (64 bytes, zero) | index list are adjacent: a short list is uploaded together with the zeros (one command instead of two)
    std::vector<int32_t> up;   // (alive until the stream has been synchronised below)
    if (idx_bytes && m <= 65536) {
        up.assign(16 + (size_t)m, 0);
        memcpy(up.data() + 16, index, idx_bytes);
        PGX_HIP(ctx, hipMemcpyAsync(d_cnt, up.data(), 64 + idx_bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        PGX_HIP(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
        if (idx_bytes) PGX_HIP(ctx, hipMemcpyAsync(d_idx, index, idx_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    const int* ix = sel == PGX_SEL_INDEX ? d_idx : nullptr;
    switch (kind) {
    case PGX_GRAM_AFFINE:
        if (D == 2) launch<GenAffine2>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt);
        else if (D == 4) launch<GenAffine4>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt);
        else launch<GenAffine5>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt);
        break;
    case PGX_GRAM_DLT_H: launch<GenDltH>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt); break;
    case PGX_GRAM_EPI_F: launch<GenEpiF>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt); break;
    case PGX_GRAM_VP: launch<GenVp>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt); break;
    default: launch<GenPnpGn>(ctx, prm, ix, m, label, ww, wpow, blocks, d_part, d_cnt); break;
    }
    PGX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(gram_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_part, blocks, nv, d_out);
    PGX_HIP(ctx, hipGetLastError());
    // result (64 doubles) | counters are adjacent: one copy back, into pinned memory
    void* hs = nullptr;
    PGX_TRY(host_staging(ctx, 64 * sizeof(double) + 8, &hs));
    PGX_TRY(d2h(ctx, hs, d_out, 64 * sizeof(double) + 8));
    PGX_TRY(sync_deliver(ctx));
    memcpy(out, hs, (size_t)nv * sizeof(double));
    const int* cnt = (const int*)((const char*)hs + 64 * sizeof(double));
    if (count) *count = cnt[0];
    if (bad) *bad = cnt[1];
    return PGX_OK;
}


// ---- smallest eigenpair of small symmetric matrices: the dense solve of the non-minimal refits behind the C ABI (round 6) -------
// Replaces: Eigen::SelfAdjointEigenSolver as the refit solvers use it (solver_vanishing_point_two_lines.h:227 in-tree; the DLT /
// 8-point solvers of the absent submodule on A^T A), so far numpy's LAPACK on the host.  Cyclic Jacobi in FP64, ONE LANE PER
// MATRIX in the operation order of the CPU restatement the tests hold (no contraction, IEEE division and square root): the
// device and that restatement return the same bits (tests), and both agree with LAPACK to ~1e-13 on the eigenvector (tests, tolerance
// stated there).  B is tens of matrices per call (one local-optimisation round): latency, not throughput.
__global__ __launch_bounds__(64) void eigh_smallest_kernel(const double* __restrict__ A, int q, int64_t B, double* __restrict__ vec,
                                                            double* __restrict__ val)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double a[9][9], v[9][9];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            a[i][j] = (i < q && j < q) ? A[b * q * q + (int64_t)i * q + j] : 0.0;
            v[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int p = 0; p < q; ++p) {
            dg = dg + a[p][p] * a[p][p];
            for (int r = p + 1; r < q; ++r) off = off + a[p][r] * a[p][r];
        }
        if (!(off > 4.930380657631324e-32 * dg)) break;
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
    for (int k = 0; k < q; ++k) vec[b * q + k] = v[k][best];
    val[b] = a[best][best];
}

int eigh_smallest_launch(pgx_ctx* ctx, const double* A, int q, int64_t B, double* vec, double* val)
{
    if (!A || !vec || !val) return fail(ctx, PGX_ERR_INVALID, "pgx_eigh_smallest_batch: NULL argument");
    if (q < 1 || q > 9) return fail(ctx, PGX_ERR_INVALID, "pgx_eigh_smallest_batch: q = %d (1..9)", q);
    if (B <= 0) return PGX_OK;
    if (B > (1 << 20)) return fail(ctx, PGX_ERR_INVALID, "pgx_eigh_smallest_batch: at most 2^20 matrices per call");
    const size_t in_bytes = (size_t)B * q * q * sizeof(double), out_bytes = (size_t)B * (q + 1) * sizeof(double);
    PGX_TRY(ensure(ctx, ctx->fit_scratch, in_bytes + out_bytes + 256));
    double* d_A = (double*)ctx->fit_scratch.p;
    double* d_vec = (double*)((char*)ctx->fit_scratch.p + ((in_bytes + 255) & ~(size_t)255));
    double* d_val = d_vec + (size_t)B * q;
    PGX_HIP(ctx, hipMemcpyAsync(d_A, A, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(eigh_smallest_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, d_A, q, B, d_vec, d_val);
    PGX_HIP(ctx, hipGetLastError());
    PGX_TRY(d2h(ctx, vec, d_vec, (size_t)B * q * sizeof(double)));
    PGX_TRY(d2h(ctx, val, d_val, (size_t)B * sizeof(double)));
    PGX_TRY(sync_deliver(ctx));
    return PGX_OK;
}

}  // namespace pgx
