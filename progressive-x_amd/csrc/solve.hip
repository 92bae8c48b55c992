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
//   absolute pose from 3 points (P3P)   DefaultPnPEstimator's minimal solver (progressivex_python.cpp:119), absent upstream:
//                                       Grunert's quartic, positive real roots by bisection between the critical points,
//                                       pose from the orthonormal frames of the two congruent triangles; four slots
// Operation order is the contract (bit-exact against the oracle's C restatement, no contraction, IEEE sqrt and divide).
// A degenerate sample (coincident points / parallel or identical lines) yields a NaN model, which can never have an
// inlier; the caller drops it (the reference's solvers return "no model").
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "pgx_internal.h"
#include "rng.hip.h"

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

// ---- P3P -------------------------------------------------------------------------------------------------------------
/* monic cubic x^3 + a x^2 + b x + c: one real root by 200 bisection steps inside the Cauchy bound, the other two from the
 * quadratic factor (NaN when complex).  Only + - * / sqrt. */
__device__ void cubic_roots(double a, double b, double c, double *r)
{
    r[0] = r[1] = r[2] = __builtin_nan("");
    double lo = -(1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)))), hi = -lo;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        // once the midpoint rounds onto an end point this update either changes nothing or collapses the interval onto it; either
        // way every later step reproduces the state: stopping after it gives the bits of all 200 steps (~60 are needed)
        const bool last = mid == lo || mid == hi;
        const double pv = ((mid + a) * mid + b) * mid + c;
        if (pv < 0.0) lo = mid; else hi = mid;
        if (last) break;
    }
    const double l0 = 0.5 * (lo + hi);
    r[0] = l0;
    const double qb = a + l0, qc = b + qb * l0;
    const double disc = qb * qb - 4.0 * qc;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        r[1] = (-qb - sq) / 2.0;
        r[2] = (-qb + sq) / 2.0;
    }
}

__device__ double quartic_eval(const double *m, double x) { return (((x + m[3]) * x + m[2]) * x + m[1]) * x + m[0]; }

/* P3P (Grunert's quartic in v = s3/s1 as in Fischler-Bolles; DefaultPnPEstimator's minimal solver is absent upstream):
 * rows (u, v, X, Y, Z) in normalised image coordinates; positive real roots of the quartic by bisection between the
 * critical points (roots of the derivative cubic) inside (0, Cauchy bound); depths from the two remaining constraints
 * (consistency 1e-6); pose from the orthonormal frames of the two congruent triangles.  out = 4 slots x 12 ([R|t]
 * row-major), NaN = no solution. */
__device__ void solve_p3p(const double *pts, int64_t n, const int32_t *smp, double *out)
{
    for (int k = 0; k < 48; ++k) out[k] = __builtin_nan("");
    double f[3][3], X[3][3];
    for (int r = 0; r < 3; ++r) {
        const int32_t i = smp[r];
        if (i < 0 || i >= n) return;
        const double u = pts[(int64_t)i * 5], v = pts[(int64_t)i * 5 + 1];
        const double ln = sqrt(u * u + v * v + 1.0);
        f[r][0] = u / ln; f[r][1] = v / ln; f[r][2] = 1.0 / ln;
        X[r][0] = pts[(int64_t)i * 5 + 2]; X[r][1] = pts[(int64_t)i * 5 + 3]; X[r][2] = pts[(int64_t)i * 5 + 4];
    }
    double d12[3], d02[3], d01[3];
    for (int k = 0; k < 3; ++k) { d12[k] = X[1][k] - X[2][k]; d02[k] = X[0][k] - X[2][k]; d01[k] = X[0][k] - X[1][k]; }
    const double a2 = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    const double b2 = d02[0] * d02[0] + d02[1] * d02[1] + d02[2] * d02[2];
    const double c2 = d01[0] * d01[0] + d01[1] * d01[1] + d01[2] * d01[2];
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return;
    const double ca = f[1][0] * f[2][0] + f[1][1] * f[2][1] + f[1][2] * f[2][2];
    const double cb = f[0][0] * f[2][0] + f[0][1] * f[2][1] + f[0][2] * f[2][2];
    const double cg = f[0][0] * f[1][0] + f[0][1] * f[1][1] + f[0][2] * f[1][2];
    const double q = (a2 - c2) / b2, rr = (a2 + c2) / b2;
    const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - rr) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca - 4.0 * rr * ca * cb * cg +
                             2.0 * (b2 - a2) / b2 * cg * cg);
    const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - rr) * ca * cg);
    const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
    if (!(fabs(A4) > 1e-14) || !(fabs(A4) < 1e300)) return;
    double m[4] = {A0 / A4, A1 / A4, A2 / A4, A3 / A4};  /* monic: x^4 + m3 x^3 + m2 x^2 + m1 x + m0 */
    for (int k = 0; k < 4; ++k) if (!(fabs(m[k]) < 1e300)) return;
    const double B = 1.0 + fmax(fmax(fabs(m[0]), fabs(m[1])), fmax(fabs(m[2]), fabs(m[3])));
    double crit[3];
    cubic_roots(0.75 * m[3], 0.5 * m[2], 0.25 * m[1], crit);
    /* break points of (0, B): the positive critical points in ascending order */
    double bp[5];
    int nb = 0;
    bp[nb++] = 0.0;
    for (int pass = 0; pass < 3; ++pass) {          /* selection in ascending order (at most three values) */
        double best = B;
        for (int k = 0; k < 3; ++k)
            if (crit[k] == crit[k] && crit[k] > bp[nb - 1] && crit[k] < best) best = crit[k];
        if (best < B) bp[nb++] = best; else break;
    }
    bp[nb++] = B;
    int slot = 0;
    for (int s = 0; s + 1 < nb && slot < 4; ++s) {
        double lo = bp[s], hi = bp[s + 1];
        const double plo = quartic_eval(m, lo), phi = quartic_eval(m, hi);
        if (!((plo < 0.0 && phi >= 0.0) || (plo >= 0.0 && phi < 0.0))) continue;
        const int rising = plo < 0.0;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            const bool last = mid == lo || mid == hi;   // (see cubic_roots: the state is a fixed point after this update)
            const double pv = quartic_eval(m, mid);
            if ((pv < 0.0) == rising) lo = mid; else hi = mid;
            if (last) break;
        }
        const double v = 0.5 * (lo + hi);
        if (!(v > 0.0)) continue;
        const double den = 2.0 * (cg - v * ca);
        const double u = ((-1.0 + q) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
        const double s1 = sqrt(b2 / (1.0 + v * v - 2.0 * v * cb));
        const double s2 = u * s1, s3 = v * s1;
        if (!(s1 > 0.0) || !(s2 > 0.0) || !(s3 > 0.0) || !(s1 < 1e300) || !(s2 < 1e300) || !(s3 < 1e300)) continue;
        const double e1 = s1 * s1 + s2 * s2 - 2.0 * s1 * s2 * cg - c2;
        const double e2 = s2 * s2 + s3 * s3 - 2.0 * s2 * s3 * ca - a2;
        if (!(fabs(e1) < 1e-6 * c2) || !(fabs(e2) < 1e-6 * a2)) continue;
        /* camera-frame points and the two orthonormal frames */
        const double dep[3] = {s1, s2, s3};
        double Y[3][3];
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Y[r][k] = f[r][k] * dep[r];
        double EX[3][3], EY[3][3];
        int okf = 1;
        for (int w = 0; w < 2; ++w) {
            double (*P)[3] = w == 0 ? X : Y;
            double (*E)[3] = w == 0 ? EX : EY;
            double p1[3], p2[3];
            for (int k = 0; k < 3; ++k) { p1[k] = P[1][k] - P[0][k]; p2[k] = P[2][k] - P[0][k]; }
            const double n1 = sqrt(p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2]);
            if (!(n1 > 0.0)) { okf = 0; break; }
            for (int k = 0; k < 3; ++k) E[0][k] = p1[k] / n1;
            double cr[3] = {E[0][1] * p2[2] - E[0][2] * p2[1], E[0][2] * p2[0] - E[0][0] * p2[2], E[0][0] * p2[1] - E[0][1] * p2[0]};
            const double n3 = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
            if (!(n3 > 0.0)) { okf = 0; break; }
            for (int k = 0; k < 3; ++k) E[2][k] = cr[k] / n3;
            E[1][0] = E[2][1] * E[0][2] - E[2][2] * E[0][1];
            E[1][1] = E[2][2] * E[0][0] - E[2][0] * E[0][2];
            E[1][2] = E[2][0] * E[0][1] - E[2][1] * E[0][0];
        }
        if (!okf) continue;
        double *P = out + 12 * slot;
        double R[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = EY[0][i] * EX[0][j] + EY[1][i] * EX[1][j] + EY[2][i] * EX[2][j];
        int fin = 1;
        for (int i = 0; i < 3; ++i) {
            const double t = Y[0][i] - (R[i][0] * X[0][0] + R[i][1] * X[0][1] + R[i][2] * X[0][2]);
            if (!(fabs(t) < 1e300)) fin = 0;
            for (int j = 0; j < 3; ++j) { if (!(fabs(R[i][j]) < 1e300)) fin = 0; P[4 * i + j] = R[i][j]; }
            P[4 * i + 3] = t;
        }
        if (!fin) { for (int k = 0; k < 12; ++k) P[k] = __builtin_nan(""); continue; }
        ++slot;
    }
}

__global__ __launch_bounds__(64) void solve_p3p_kernel(const double* __restrict__ pts, int64_t n, const int* __restrict__ samples,
                                                       int S, double* __restrict__ models, int* __restrict__ perm, int Mpad)
{
    const int s = (int)(blockIdx.x * 64 + threadIdx.x);
    for (int t = s; t < Mpad; t += (int)(gridDim.x * 64)) perm[t] = t < 4 * S ? t : 0;
    if (s >= S) return;
    solve_p3p(pts, n, samples + (int64_t)s * 3, models + (int64_t)s * 48);
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
            const bool last = mid == lo || mid == hi;   // (see cubic_roots: the state is a fixed point after this update)
            const double pv = ((mid + a) * mid + b) * mid + c;
            if (pv < 0.0) lo = mid; else hi = mid;
            if (last) break;
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

// The sample indices go through a pinned staging buffer owned by the context: the caller's array is consumed before the
// call returns, so a launch that hands nothing back (models_out == NULL: the batch stays resident for pgx_score_launch) needs
// no synchronisation at all - the host goes straight on to enqueue the scoring kernels behind the solver.  An event guards
// the staging buffer against the next call.
static int upload_samples(pgx_ctx* ctx, const int32_t* samples, size_t bytes)
{
    if (ctx->h_samples_busy) { PGX_HIP(ctx, hipEventSynchronize(ctx->ev_samples)); ctx->h_samples_busy = 0; }
    if (ctx->h_samples_cap < bytes) {
        if (ctx->h_samples) (void)hipHostFree(ctx->h_samples);
        ctx->h_samples = nullptr; ctx->h_samples_cap = 0;
        PGX_HIP(ctx, hipHostMalloc(&ctx->h_samples, bytes * 2, hipHostMallocDefault));
        ctx->h_samples_cap = bytes * 2;
    }
    if (!ctx->ev_samples) PGX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_samples, hipEventDisableTiming));
    std::memcpy(ctx->h_samples, samples, bytes);
    PGX_TRY(ensure(ctx, ctx->scratch, bytes));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, ctx->h_samples, bytes, hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipEventRecord(ctx->ev_samples, ctx->stream));
    ctx->h_samples_busy = 1;
    return PGX_OK;
}

// ---- samples drawn on the device (rng.hip.h): one lane per sample, straight into the buffer the solvers read -----------------
__global__ __launch_bounds__(256) void sample_uniform_kernel(unsigned long long key, unsigned batch, int S, int64_t n, int m, int* __restrict__ samples)
{
    const int s = (int)(blockIdx.x * 256 + threadIdx.x);
    if (s >= S) return;
    int32_t row[kMaxSampleSize];
    sample_distinct(key, batch, (uint64_t)s, n, m, row);
    for (int j = 0; j < m; ++j) samples[(int64_t)s * m + j] = row[j];
}

__global__ __launch_bounds__(256) void sample_napsac_kernel(unsigned long long key, unsigned batch, int S, int64_t n, const int* __restrict__ off,
                                                            const int* __restrict__ idx, int m, int* __restrict__ samples)
{
    const int s = (int)(blockIdx.x * 256 + threadIdx.x);
    if (s >= S) return;
    int32_t row[kMaxSampleSize];
    sample_napsac(key, batch, (uint64_t)s, n, off, idx, m, row);
    for (int j = 0; j < m; ++j) samples[(int64_t)s * m + j] = row[j];
}

__global__ __launch_bounds__(256) void sample_prosac_kernel(unsigned long long key, unsigned batch, int S, int64_t n, const int* __restrict__ tops, int m,
                                                            int* __restrict__ samples)
{
    const int s = (int)(blockIdx.x * 256 + threadIdx.x);
    if (s >= S) return;
    int32_t row[kMaxSampleSize];
    sample_prosac(key, batch, (uint64_t)s, n, tops[s], m, row);
    for (int j = 0; j < m; ++j) samples[(int64_t)s * m + j] = row[j];
}

// pgx_sampler_prosac_set: the subset size n_k of PROSAC's sample number k = 1 .. count (0 = uniform over all points)
int sampler_prosac_set(pgx_ctx* ctx, const int32_t* tops, int count)
{
    if (ctx->n <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_sampler_prosac_set: points not set");
    if (!tops || count <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_sampler_prosac_set: empty table");
    for (int k = 0; k < count; ++k)
        if (tops[k] < 0 || tops[k] > ctx->n)
            return fail(ctx, PGX_ERR_INVALID, "pgx_sampler_prosac_set: subset size %d of sample %d is outside 0 .. %lld", tops[k], k + 1, (long long)ctx->n);
    PGX_TRY(ensure(ctx, ctx->prosac_tops, (size_t)count * sizeof(int32_t)));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->prosac_tops.p, tops, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the caller's buffer is free on return)
    ctx->prosac_count = count;
    ctx->prosac_points_version = ctx->points_version;
    return PGX_OK;
}

int solve_minimal_launch(pgx_ctx* ctx, const int32_t* samples, int S, double* models_out, bool resident)
{
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: points not set");
    if ((!samples && !resident) || S <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: empty sample batch");
    if (ctx->model_type == kFundamental) {
        // three model slots per sample; isotropic pre-scaling by the largest coordinate magnitude (pmax of set_points is 1
        // for this model type, so it is recomputed here from the caller-visible data: umax is not kept either)
        if (!(ctx->fscale >= 1.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: coordinate scale not available");
        const int Mtot = 3 * S;
        ctx->Mpad = ((Mtot + 255) / 256) * 256;
        PGX_TRY(ensure(ctx, ctx->models, (size_t)Mtot * 9 * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
        if (!resident) PGX_TRY(upload_samples(ctx, samples, (size_t)S * 7 * sizeof(int32_t)));
        const unsigned blocks = (unsigned)((S + 63) / 64);
        hipLaunchKernelGGL(solve_f7_kernel, dim3(blocks), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->fscale, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
        PGX_HIP(ctx, hipGetLastError());
        if (models_out) {
            PGX_TRY(d2h(ctx, models_out, ctx->models.p, (size_t)Mtot * 9 * sizeof(double)));
            PGX_TRY(sync_deliver(ctx));
        }
        ctx->M = Mtot; ctx->last_acc = nullptr;
        return PGX_OK;
    }
    if (ctx->model_type == kPnP) {
        const int Mtot = 4 * S;  // four solution slots per sample
        ctx->Mpad = ((Mtot + 255) / 256) * 256;
        PGX_TRY(ensure(ctx, ctx->models, (size_t)Mtot * 12 * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
        if (!resident) PGX_TRY(upload_samples(ctx, samples, (size_t)S * 3 * sizeof(int32_t)));
        hipLaunchKernelGGL(solve_p3p_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
        PGX_HIP(ctx, hipGetLastError());
        if (models_out) {
            PGX_TRY(d2h(ctx, models_out, ctx->models.p, (size_t)Mtot * 12 * sizeof(double)));
            PGX_TRY(sync_deliver(ctx));
        }
        ctx->M = Mtot; ctx->last_acc = nullptr;
        return PGX_OK;
    }
    if (ctx->model_type == kHomography) {
        if (!(ctx->fscale >= 1.0)) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: coordinate scale not available");
        ctx->Mpad = ((S + 255) / 256) * 256;
        PGX_TRY(ensure(ctx, ctx->models, (size_t)S * 9 * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
        if (!resident) PGX_TRY(upload_samples(ctx, samples, (size_t)S * 4 * sizeof(int32_t)));
        hipLaunchKernelGGL(solve_h4_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->fscale, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
        PGX_HIP(ctx, hipGetLastError());
        if (models_out) {
            PGX_TRY(d2h(ctx, models_out, ctx->models.p, (size_t)S * 9 * sizeof(double)));
            PGX_TRY(sync_deliver(ctx));
        }
        ctx->M = S; ctx->last_acc = nullptr;
        return PGX_OK;
    }
    if (ctx->model_type != kLine2D && ctx->model_type != kVanishingPoint)
        return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal: no device solver for model type %d yet (built: 2-point line, 2-segment vanishing point, 4-point homography, 7-point fundamental matrix, P3P)", ctx->model_type);
    ctx->Mpad = ((S + 255) / 256) * 256;
    PGX_TRY(ensure(ctx, ctx->models, (size_t)S * 3 * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->perm, (size_t)ctx->Mpad * sizeof(int)));
    if (!resident) PGX_TRY(upload_samples(ctx, samples, (size_t)S * 2 * sizeof(int32_t)));
    const unsigned blocks = (unsigned)((ctx->Mpad + kSolveBlock - 1) / kSolveBlock);
    if (ctx->model_type == kLine2D)
        hipLaunchKernelGGL((solve_kernel<kLine2D>), dim3(blocks), dim3(kSolveBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
    else
        hipLaunchKernelGGL((solve_kernel<kVanishingPoint>), dim3(blocks), dim3(kSolveBlock), 0, ctx->stream, ctx->pts.as<double>(), ctx->n,
                           ctx->scratch.as<int>(), S, ctx->models.as<double>(), ctx->perm.as<int>(), ctx->Mpad);
    PGX_HIP(ctx, hipGetLastError());
    if (models_out) {
        PGX_TRY(d2h(ctx, models_out, ctx->models.p, (size_t)S * 3 * sizeof(double)));
        PGX_TRY(sync_deliver(ctx));
    }
    ctx->M = S; ctx->last_acc = nullptr;
    return PGX_OK;
}

// pgx_solve_minimal_sampled: S minimal samples (uniform, NAPSAC on the resident graph, or PROSAC with the resident subset-size table) from the in-repo
// generator (key, batch), drawn by the device into the solvers' sample buffer - no host RNG, no index upload - then the solver of the resident model type as in pgx_solve_minimal.
int solve_minimal_sampled_launch(pgx_ctx* ctx, int sampler, uint64_t key, uint32_t batch, int S, int32_t* samples_out, double* models_out)
{
    if (sampler < 0 || sampler > 2) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: sampler %d (0 uniform, 1 NAPSAC, 2 PROSAC)", sampler);
    if (sampler == 2 && (ctx->prosac_count < S || ctx->prosac_points_version != ctx->points_version))
        return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: PROSAC needs pgx_sampler_prosac_set for the resident points with at least %d entries (has %d)", S,
                    ctx->prosac_points_version == ctx->points_version ? ctx->prosac_count : 0);
    if (sampler == 1 && (ctx->gn != ctx->n || ctx->gE <= 0)) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: NAPSAC needs the neighbourhood graph of the resident points (pgx_graph_build / pgx_set_graph)");
    if (ctx->n <= 0 || ctx->model_type < 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: points not set");
    if (S <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: empty sample batch");
    int m = 0;
    switch (ctx->model_type) {
    case kLine2D: case kVanishingPoint: m = 2; break;
    case kPnP: m = 3; break;
    case kHomography: m = 4; break;
    case kFundamental: m = 7; break;
    default: return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: no device solver for model type %d", ctx->model_type);
    }
    if (ctx->n < m) return fail(ctx, PGX_ERR_INVALID, "pgx_solve_minimal_sampled: %lld points, the minimal sample needs %d", (long long)ctx->n, m);
    PGX_TRY(ensure(ctx, ctx->scratch, (size_t)S * m * sizeof(int32_t)));
    if (sampler == 0)
        hipLaunchKernelGGL(sample_uniform_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long)key, batch, S, ctx->n, m,
                           ctx->scratch.as<int>());
    else if (sampler == 1)
        hipLaunchKernelGGL(sample_napsac_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long)key, batch, S, ctx->n,
                           ctx->goff.as<int>(), ctx->gidx.as<int>(), m, ctx->scratch.as<int>());
    else
        hipLaunchKernelGGL(sample_prosac_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned long long)key, batch, S, ctx->n,
                           ctx->prosac_tops.as<int>(), m, ctx->scratch.as<int>());
    PGX_HIP(ctx, hipGetLastError());
    if (samples_out) {
        PGX_TRY(d2h(ctx, samples_out, ctx->scratch.p, (size_t)S * m * sizeof(int32_t)));
        PGX_TRY(sync_deliver(ctx));
    }
    return solve_minimal_launch(ctx, nullptr, S, models_out, true);
}

}  // namespace pgx
