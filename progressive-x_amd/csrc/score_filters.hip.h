// score_filters.hip.h — the conservative rejection filters of the score path: Filter<MT> (FP64, division-free), Filter32<MT>
// (FP32 pre-filter per pair + the group bound of the cull kernel).  A filter may only say "certainly an outlier"; every pair it
// does not reject goes through the exact FP64 residual in the reference's operation order, so counts, masks and scores are
// those of evaluating every pair.  The error budgets are derived in docs/lab-notebook.md (5.2) and machine-checked by
// scripts/verify_filters.py (exact rational arithmetic) and by the PGX_VERIFY=1 mode of the group-major kernel.
//
// Replaces nothing in the reference (its getScore evaluates every pair: scoring_function_with_compound_model.h:78-121).
#pragma once
#include <cmath>

#include "pgx_internal.h"

namespace pgx {

// ---- conservative rejection filter ------------------------------------------------------------
// Most (point, hypothesis) pairs are far from the threshold.  For those the two IEEE divisions of the reprojection /
// transfer residual (~110 of ~230 VALU cycles per wave-iteration) are wasted: the pair only has to be PROVEN an outlier.
// reject() evaluates the division-free form  (u pz - px)^2 + (v pz - py)^2 > T2 (1 + 2^-20) pz^2  with FMA arithmetic
// and returns true only when, additionally, pz is large enough against the rounding error E <= 4.5 eps L_h P_i of the
// projection (L_h = largest row 1-norm of the hypothesis, P_i = max(|coords of point i|, 1), both precomputed) that the
// inequality cannot be flipped by rounding in either arithmetic:  E (1 + Umax + T) 2^24 / T <= |pz|, with the host
// guaranteeing Umax / T <= 2^28 (otherwise the unfiltered instance is launched).  Every pair that is not rejected —
// including everything involving NaN/Inf, for which all comparisons are false — goes through the exact, oracle-order,
// no-FMA path below, so counts, masks and scores are bit-identical to the unfiltered kernel; the proof that no true
// inlier (exact r^2 < T2) can be rejected is in DESIGN.md §5.2.
constexpr double kFilterDelta = 1.0 / 1048576.0;  // 2^-20 > 12 * 2^-24

template <int MT> struct Filter {
    static constexpr bool enabled = false;
    struct Lane {};
    template <class MD> static __device__ __forceinline__ Lane prep(const MD&, double) { return {}; }
    template <class PT, class MD>
    static __device__ __forceinline__ bool reject(const PT&, const MD&, const Lane&, double, double) { return false; }
};

template <> struct Filter<kPnP> {
    static constexpr bool enabled = true;
    struct Lane { double c; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& m, double guard) {
        const double l0 = fabs(m[0]) + fabs(m[1]) + fabs(m[2]) + fabs(m[3]);
        const double l1 = fabs(m[4]) + fabs(m[5]) + fabs(m[6]) + fabs(m[7]);
        const double l2 = fabs(m[8]) + fabs(m[9]) + fabs(m[10]) + fabs(m[11]);
        const double l = fmax(l0, fmax(l1, l2));
        // the test squares pz: outside this range of scales it is not trusted at all (trust = c * pmax <= |pz| fails for c = inf)
        return {(l > 1e-100 && l < 1e100) ? guard * l : 1.0 / 0.0};
    }
    template <class PT, class MD>
    static __device__ __forceinline__ bool reject(const PT& p, const MD& m, const Lane& ln, double pmax, double T2d) {
        const double px = __builtin_fma(m[0], p[2], __builtin_fma(m[1], p[3], __builtin_fma(m[2], p[4], m[3])));
        const double py = __builtin_fma(m[4], p[2], __builtin_fma(m[5], p[3], __builtin_fma(m[6], p[4], m[7])));
        const double pz = __builtin_fma(m[8], p[2], __builtin_fma(m[9], p[3], __builtin_fma(m[10], p[4], m[11])));
        const double a = __builtin_fma(p[0], pz, -px);
        const double b = __builtin_fma(p[1], pz, -py);
        const double lhs = __builtin_fma(b, b, a * a);
        const double rhs = (pz * pz) * T2d;
        const bool trust = ln.c * pmax <= fabs(pz);  // false on NaN
        return trust && (lhs > rhs);                 // false on NaN
    }
};

template <> struct Filter<kHomography> {
    static constexpr bool enabled = true;
    struct Lane { double c; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& h, double guard) {
        const double l0 = fabs(h[0]) + fabs(h[1]) + fabs(h[2]);
        const double l1 = fabs(h[3]) + fabs(h[4]) + fabs(h[5]);
        const double l2 = fabs(h[6]) + fabs(h[7]) + fabs(h[8]);
        const double l = fmax(l0, fmax(l1, l2));
        return {(l > 1e-100 && l < 1e100) ? guard * l : 1.0 / 0.0};   // (see Filter<kPnP>)
    }
    template <class PT, class MD>
    static __device__ __forceinline__ bool reject(const PT& p, const MD& h, const Lane& ln, double pmax, double T2d) {
        const double t1 = __builtin_fma(h[0], p[0], __builtin_fma(h[1], p[1], h[2]));
        const double t2 = __builtin_fma(h[3], p[0], __builtin_fma(h[4], p[1], h[5]));
        const double t3 = __builtin_fma(h[6], p[0], __builtin_fma(h[7], p[1], h[8]));
        const double a = __builtin_fma(p[2], t3, -t1);
        const double b = __builtin_fma(p[3], t3, -t2);
        const double lhs = __builtin_fma(b, b, a * a);
        const double rhs = (t3 * t3) * T2d;
        const bool trust = ln.c * pmax <= fabs(t3);
        return trust && (lhs > rhs);
    }
};

// Symmetric transfer error, model [H | H^-1]: r^2 = fl(forward + backward) >= the forward term, which is computed in exactly
// the operation order of Residual<kHomography> (residuals.hip.h) - rounding is monotone and the backward term is >= 0 or NaN -
// so every pair the forward filters prove "not an inlier" is not an inlier of the symmetric residual either: the
// homography filters are reused on the first nine entries.  (The backward term is not filtered.)
template <> struct Filter<kHomographySym> : Filter<kHomography> {};

// ---- FP32 pre-filter -------------------------------------------------------------------------------
// Same inequality evaluated in single precision on f32 copies of the point (one 32-byte row: coords, scale) and of the
// hypothesis: 18 VALU ops at the f32 rate instead of 17 at the f64 rate.  Error budget: inputs rounded to f32 and three
// chained FMAs give |p~ - p*| <= E32 = 5.5 * 2^-24 * (L3_h P_i + t_h) (L3_h = largest 1-norm of the multiplying part of a
// row, t_h = largest |constant term|); with tau = 2^-10 the trust test E32 (1 + U + T) / (tau T) <= |p~z| (one FMA +
// compare per pair; constants rounded up) and the host guard U / T <= tau * 2^24 bound every error term by tau T |p~z|,
// and the chain of inequalities of §5.2 gives a~^2 + b~^2 <= p~z^2 T^2 (1 + 7.7 tau + O(tau^2) + 6 * 2^-24)
// < p~z^2 T^2 (1 + 2^-7) for every pair the exact path accepts.  Candidates go straight to the exact FP64 path.
constexpr double kFilter32Delta = 1.0 / 128.0;  // 2^-7
constexpr double kInflate = 1.000001;           // (float)(x * kInflate) >= x for every finite double x > 0

__device__ __forceinline__ float f32_up(double x) { return (float)(x * kInflate); }

// The f32 tests below are homogeneous in the hypothesis (a residual does not change when its model is multiplied by a constant -
// for lines the threshold scales along), but their intermediate SQUARES are not representable for every scale: a hypothesis
// 1e-24 times a perfectly good one (tests/soak_scoring.py found it) made pz^2 underflow to 0 before the multiplication by T2,
// and "lhs > 0" rejected true inliers.  Every f32 copy is therefore made from the hypothesis scaled by a power of two (exact)
// that brings its largest entry into [0.5, 1): returns that factor (1 when the largest entry is 0, Inf or NaN).
// *off: the hypothesis is outside the band of scales (largest entry in [1e-75, 1e75]) in which the EXACT path's own f64
// arithmetic neither overflows nor underflows for coordinates up to 1e30 (the dispatch checks that): outside it the oracle's
// residual is whatever IEEE makes of it (a vanishing point 1e158 away divides by inf and every segment becomes an "inlier"
// with residual 0), and only the exact path reproduces that - such a hypothesis is never rejected or culled.
template <int P, class MD> __device__ __forceinline__ double pow2_normaliser(const MD& m, bool* off)
{
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k) { const double a = fabs(m[k]); if (a > mx) mx = a; }   // NaN entries are skipped by the comparison
    *off = !(mx >= 1e-75) || !(mx <= 1e75);
    if (!(mx > 0.0) || !(mx < 1.7976931348623157e308)) return 1.0;
    int e;
    (void)frexp(mx, &e);
    return ldexp(1.0, -e);
}

template <int MT> struct Filter32 {
    static constexpr bool enabled = false;
    static constexpr int kRowVals = 6;    // floats of a point's f32 row the filter reads
    static constexpr int kGroupVals = 9;  // floats of a group row the bound test reads
    struct Lane {};
    template <class MD> static __device__ __forceinline__ Lane prep(const MD&, double, double) { return {}; }
    static __device__ __forceinline__ bool reject(const float*, const Lane&, float) { return false; }
    static __device__ __forceinline__ bool group_reject(const float*, const Lane&, float) { return false; }
};

// ---- group-level rejection ---------------------------------------------------------------------------
// The points are kept in Morton order of all their coordinates, so 64 consecutive points form a compact group: centre c
// and radius rho of the part the projective map multiplies, centre (ub, vb) and half extents (ru, rv) of the observed
// part.  For every point of the group |z_i - z_c| <= ||p_z|| rho =: dz (likewise dx, dy), hence
//   |u_i z_i - x_i| >= |ub z_c - x_c| - (ru (|z_c| + dz) + |ub| dz + dx)      and      T |z_i| <= T (|z_c| + dz),
// and an inlier needs |u_i z_i - x_i| < T |z_i| in both coordinates: if either lower bound exceeds the upper bound no
// point of the group is an inlier of this hypothesis.  Evaluated in f32 at the centre under the point filter's trust
// test (centre errors <= a few tau T |z_c|), radii and norms stored rounded up by 1e-5, T inflated by 2^-6.  A wave
// skips the 64 points when all of its 64 hypotheses reject the group (the batch is in locality order, so a wave's
// hypotheses look at the same image region): 74 % of the (wave, group) pairs of the metric batch.
// kGroupInflate, kGroupRow, kSuper: pgx_internal.h (shared with setpoints.hip, which builds the rows on the device)

// Model types whose f32 row is the generic one of setpoints.hip sp_prep_kernel - (float) of every coordinate, then the scale
// max(1, |coordinates in0 .. in1|) * 1.000001 in slot 5: the group kernel derives it from the f64 row instead of loading it.
template <int MT> struct RowFromPoint { static constexpr int in0 = 0, in1 = -1; };           // in1 < in0: no (the row is loaded)
template <> struct RowFromPoint<kPnP> { static constexpr int in0 = 2, in1 = 4; };            // setpoints.hip: obs0 = 0, in0 = 2, in1 = 4
template <> struct RowFromPoint<kHomography> { static constexpr int in0 = 0, in1 = 3; };     // obs0 = 2, in0 = 0, in1 = 3
template <> struct RowFromPoint<kHomographySym> { static constexpr int in0 = 0, in1 = 3; };

template <> struct Filter32<kPnP> {
    static constexpr bool enabled = true;
    static constexpr int kRowVals = 6, kGroupVals = 9;
    struct Lane { float m[12]; float c1, c0; float n0, n1, n2; float nanh; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& m0, double guard32, double) {
        Lane ln;
        bool nan = false;
        bool off;
        const double sc = pow2_normaliser<12>(m0, &off);
        double m[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { m[k] = m0[k] * sc; ln.m[k] = (float)m[k]; nan |= !(m0[k] == m0[k]); }
        ln.nanh = nan ? 1.0f : 0.0f;  // a NaN entry makes every residual NaN (all 12 enter it): never an inlier
        ln.n0 = (float)(sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) * kGroupInflate);
        ln.n1 = (float)(sqrt(m[4] * m[4] + m[5] * m[5] + m[6] * m[6]) * kGroupInflate);
        ln.n2 = (float)(sqrt(m[8] * m[8] + m[9] * m[9] + m[10] * m[10]) * kGroupInflate);
        const double l3 = fmax(fabs(m[0]) + fabs(m[1]) + fabs(m[2]),
                               fmax(fabs(m[4]) + fabs(m[5]) + fabs(m[6]), fabs(m[8]) + fabs(m[9]) + fabs(m[10])));
        const double t = fmax(fabs(m[3]), fmax(fabs(m[7]), fabs(m[11])));
        ln.c1 = f32_up(guard32 * l3);
        ln.c0 = off ? __builtin_inff() : fmaxf(f32_up(guard32 * t), 1e-30f);   // inf: the trust test never holds
        return ln;
    }
    // p = (u, v, X, Y, Z, scale, -, -) in f32
    static __device__ __forceinline__ bool reject(const float* p, const Lane& ln, float T2d) {
        const float* m = ln.m;
        const float px = __builtin_fmaf(m[0], p[2], __builtin_fmaf(m[1], p[3], __builtin_fmaf(m[2], p[4], m[3])));
        const float py = __builtin_fmaf(m[4], p[2], __builtin_fmaf(m[5], p[3], __builtin_fmaf(m[6], p[4], m[7])));
        const float pz = __builtin_fmaf(m[8], p[2], __builtin_fmaf(m[9], p[3], __builtin_fmaf(m[10], p[4], m[11])));
        const float a = __builtin_fmaf(p[0], pz, -px);
        const float b = __builtin_fmaf(p[1], pz, -py);
        const float lhs = __builtin_fmaf(b, b, a * a);
        const float rhs = (pz * pz) * T2d;
        const bool trust = __builtin_fmaf(ln.c1, p[5], ln.c0) <= fabsf(pz);  // false on NaN
        return trust && (lhs > rhs);                                         // false on NaN
    }
    // g = (cX, cY, cZ, rho, ub, vb, ru, rv, scale, -, -, -)
    static __device__ __forceinline__ bool group_reject(const float* g, const Lane& ln, float Tup) {
        const float* m = ln.m;
        const float cx = __builtin_fmaf(m[0], g[0], __builtin_fmaf(m[1], g[1], __builtin_fmaf(m[2], g[2], m[3])));
        const float cy = __builtin_fmaf(m[4], g[0], __builtin_fmaf(m[5], g[1], __builtin_fmaf(m[6], g[2], m[7])));
        const float cz = __builtin_fmaf(m[8], g[0], __builtin_fmaf(m[9], g[1], __builtin_fmaf(m[10], g[2], m[11])));
        // NaN hypothesis: its residuals are NaN, never an inlier.  (Tested on the hypothesis itself: an entry that merely
        // overflows f32 gives inf - inf = NaN HERE although its f64 residuals may be perfectly good - such a group is kept,
        // every comparison below being false on NaN.)
        if (ln.nanh != 0.0f) return true;
        const bool trust = __builtin_fmaf(ln.c1, g[8], ln.c0) <= fabsf(cz);
        const float dz = ln.n2 * g[3], dx = ln.n0 * g[3], dy = ln.n1 * g[3];
        const float zs = fabsf(cz) + dz;
        const float ex = fabsf(__builtin_fmaf(g[4], cz, -cx)), ey = fabsf(__builtin_fmaf(g[5], cz, -cy));
        const float mx = __builtin_fmaf(g[6], zs, __builtin_fmaf(fabsf(g[4]), dz, dx));
        const float my = __builtin_fmaf(g[7], zs, __builtin_fmaf(fabsf(g[5]), dz, dy));
        const float tol = Tup * zs;
        return trust && ((ex - mx > tol) || (ey - my > tol));  // false on NaN/inf arithmetic
    }
};

// Homographies are scored in PIXEL coordinates (|coordinates| ~ 1e3): there the trust test of the PnP filter above (errors
// <= tau T |t3| with tau = 2^-10) fails for almost every pair - E32 (1 + U) ~ 0.5 px against tau T ~ 0.004 px - and the
// filter would pass everything on.  So this filter carries its error terms explicitly instead (like the vanishing-point and
// Sampson filters below): with E_t = 5.5 u (L2 P + t) + 4 eta bounding the f32 error of t1, t2, t3 AND the exact path's own
// f64 rounding (L2 = largest |h_a| + |h_b| of a row, t = largest |h_c|, P = max(|coordinates|, 1) rounded up),
//   |a~ - a*| <= E_a = E_t (1 + P) + 1.01 u (P |t3~| + |a~|)        (a = x2 t3 - t1: the product, the rounded x2, the FMA)
//   reject  <=>  max(|a~| - E_a, 0)^2 + max(|b~| - E_b, 0)^2  >  T2 (1 + 2^-6) (|t3~| + E_t)^2
// which implies (a*^2 + b*^2) / t3*^2 > T2 (1 + 2^-6)(1 - 8 u) and the computed r_c^2 >= that (1 - 10 eps) > T2.  Overflow of
// an f32 product makes E_a infinite (it contains P |t3~| and |a~|): inf - inf = NaN, not rejected.  No global guard.
// Group test: as for PnP with the same explicit terms at the group's scales - |u_i z_i - x_i| >= ex - mx - E_g and
// |z_i| <= |cz| + dz + E_t, E_g = E_t (1 + |ub|) + 1.01 u (|ub| |cz| + ex).
template <> struct Filter32<kHomography> {
    static constexpr bool enabled = true;
    static constexpr int kRowVals = 6, kGroupVals = 9;
    struct Lane { float m[9]; float e1, e0; float n0, n1, n2; float nanh; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& h0, double, double) {
        Lane ln;
        bool nan = false;
        bool off;
        const double sc = pow2_normaliser<9>(h0, &off);
        double h[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { h[k] = h0[k] * sc; ln.m[k] = (float)h[k]; nan |= !(h0[k] == h0[k]); }
        ln.nanh = nan ? 1.0f : 0.0f;
        ln.n0 = (float)(sqrt(h[0] * h[0] + h[1] * h[1]) * kGroupInflate);
        ln.n1 = (float)(sqrt(h[3] * h[3] + h[4] * h[4]) * kGroupInflate);
        ln.n2 = (float)(sqrt(h[6] * h[6] + h[7] * h[7]) * kGroupInflate);
        const double l2 = fmax(fabs(h[0]) + fabs(h[1]), fmax(fabs(h[3]) + fabs(h[4]), fabs(h[6]) + fabs(h[7])));
        const double t = fmax(fabs(h[2]), fmax(fabs(h[5]), fabs(h[8])));
        const double u = 5.9604644775390625e-8, eta = 1.1754943508222875e-38;
        ln.e1 = f32_up(5.5 * u * l2 + 4.0 * eta);
        ln.e0 = off ? __builtin_inff() : f32_up(5.5 * u * t + eta);   // inf: every error term is infinite, nothing is rejected
        return ln;
    }
    // p = (x1, y1, x2, y2, -, P, -, -) in f32, P = max(|all four coordinates|, 1) rounded up
    static __device__ __forceinline__ bool reject(const float* p, const Lane& ln, float T2d) {
        const float* h = ln.m;
        const float t1 = __builtin_fmaf(h[0], p[0], __builtin_fmaf(h[1], p[1], h[2]));
        const float t2 = __builtin_fmaf(h[3], p[0], __builtin_fmaf(h[4], p[1], h[5]));
        const float t3 = __builtin_fmaf(h[6], p[0], __builtin_fmaf(h[7], p[1], h[8]));
        const float a = fabsf(__builtin_fmaf(p[2], t3, -t1));
        const float b = fabsf(__builtin_fmaf(p[3], t3, -t2));
        const float Et = __builtin_fmaf(ln.e1, p[5], ln.e0);
        const float at3 = fabsf(t3);
        const float base = __builtin_fmaf(6.0202e-8f /* 1.01 u */ * p[5], at3, __builtin_fmaf(Et, p[5], Et));
        const float ma = fmaxf(a - __builtin_fmaf(6.0202e-8f, a, base), 0.0f);   // fmaxf(NaN, 0) = 0: never rejects on its own
        const float mb = fmaxf(b - __builtin_fmaf(6.0202e-8f, b, base), 0.0f);
        const float den = at3 + Et;
        return __builtin_fmaf(mb, mb, ma * ma) > (den * den) * T2d && (base == base);   // NaN / inf error terms: not rejected
    }
    // g = (c1x, c1y, 0, rho, x2b, y2b, r2x, r2y, scale, -, -, -)
    static __device__ __forceinline__ bool group_reject(const float* g, const Lane& ln, float Tup) {
        const float* h = ln.m;
        const float cx = __builtin_fmaf(h[0], g[0], __builtin_fmaf(h[1], g[1], h[2]));
        const float cy = __builtin_fmaf(h[3], g[0], __builtin_fmaf(h[4], g[1], h[5]));
        const float cz = __builtin_fmaf(h[6], g[0], __builtin_fmaf(h[7], g[1], h[8]));
        if (ln.nanh != 0.0f) return true;  // NaN entry: every residual is NaN
        const float Et = __builtin_fmaf(ln.e1, g[8], ln.e0);
        const float dz = ln.n2 * g[3], dx = ln.n0 * g[3], dy = ln.n1 * g[3];
        const float acz = fabsf(cz);
        const float zs = acz + dz + Et;
        const float ex = fabsf(__builtin_fmaf(g[4], cz, -cx)), ey = fabsf(__builtin_fmaf(g[5], cz, -cy));
        const float aub = fabsf(g[4]), avb = fabsf(g[5]);
        const float Egx = __builtin_fmaf(Et, aub, Et) + 6.0202e-8f * __builtin_fmaf(aub, acz, ex);
        const float Egy = __builtin_fmaf(Et, avb, Et) + 6.0202e-8f * __builtin_fmaf(avb, acz, ey);
        const float mx = __builtin_fmaf(g[6], zs, __builtin_fmaf(aub, dz, dx)) + Egx;
        const float my = __builtin_fmaf(g[7], zs, __builtin_fmaf(avb, dz, dy)) + Egy;
        const float tol = Tup * zs;
        return (ex - mx * 1.0001f > tol) || (ey - my * 1.0001f > tol);  // false on NaN / inf arithmetic
    }
};

// ---- vanishing points (vanishing_point_estimator.h:166-189) ---------------------------------------------------------------
// r = |N| / D with N = lx xs + ly ys + lz and D = ||(lx, ly)||, l = m x v (m = the segment's midpoint).  Expanding l gives
//   N = v0 a + v1 b + v2 c,   a = (ys - ye) / 2,  b = (xe - xs) / 2,  c = (xs ye - xe ys) / 2      (exact identity)
//   lx = my v2 - v1,  ly = v0 - mx v2
// so a segment's f32 row holds (a, b, c, mx, my, P, P^2), P = max(|coordinates|, 1) rounded up, and the filter is 17 f32
// operations without a division or a root.  Error budget (u = 2^-24, eps = 2^-53, tau = 2^-10):
//   |N~ - N*| <= E_N = 8 u (|v0||a| + |v1||b| + |v2||c|) + 2^-50 |v2| P^2      (inputs rounded to f32, three FMAs; c itself is
//                                                                              a difference of two f64 products)
//   ||(lx~, ly~) - (lx*, ly*)|| <= 6 u (P |v2| + |v0| + |v1|) =: E_D
//   the exact path's own f64 evaluation: |N_c - N*| <= 8 eps (2 P^2 |v2| + 3 P (|v0| + |v1|)) =: E64
// trust test  D~ >= t(P) = e2 P^2 + e1 P + e0  (constants below) makes E_D <= tau D and E64 <= tau T D; then
//   reject  <=>  trust  and  m := |N~| - E_N > 0  and  m^2 > T2 (1 + 2^-6) D~^2
// implies r* = |N*| / D* > T (1 + tau)^3 and the computed r_c >= r* (1 - tau) / (1 + tau) > T: the exact path would not
// have accepted.  NaN / Inf anywhere make a comparison false: not rejected.
// Group test: the same quantities on the group's box of ORIENTED, LENGTH-NORMALISED features (a, b, c) / h, h = half the
// segment length (the residual is h |sin angle(segment, direction to the vanishing point)|): with centre (A, B, C, MX, MY),
// radii (rA, rB, rC), rM = radius of the midpoints, hmin = the shortest half length,
//   |N^_i| >= |N^(centre)| - (|v0| rA + |v1| rB + |v2| rC),   D_i <= D(centre) + |v2| rM,
// and no member is an inlier when hmin (|N^c| - R_N) > T'' (Dc + R_D), under the trust test at Dc - R_D with the group's
// largest P.  Groups are 64 consecutive segments of the Hough order of their lines (theta, rho, length) - setpoints.hip sp_vp_rows_kernel.
template <> struct Filter32<kVanishingPoint> {
    static constexpr bool enabled = true;
    static constexpr int kRowVals = 7, kGroupVals = 12;
    struct Lane { float v[3]; float e2, e1, e0, e50, t2pp, invT, nanh; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& v0, double, double T2) {
        Lane ln;
        bool off;
        const double sc = pow2_normaliser<3>(v0, &off);
        const double v[3] = {v0[0] * sc, v0[1] * sc, v0[2] * sc};
        ln.v[0] = (float)v[0]; ln.v[1] = (float)v[1]; ln.v[2] = (float)v[2];
        ln.nanh = (v0[0] == v0[0] && v0[1] == v0[1] && v0[2] == v0[2]) ? 0.0f : 1.0f;
        const double V1 = fabs(v[0]) + fabs(v[1]), V2 = fabs(v[2]), T = sqrt(T2);
        const double k38 = 3.637978807091713e-12 /* 2^-38 */, k11 = 4.8828125e-4 /* 2^-11 */;
        ln.e2 = f32_up(k38 * V2 / T * 1.001);
        ln.e1 = f32_up((k11 * V2 + k38 * V1 / T) * 1.001);
        ln.e0 = off ? __builtin_inff() : fmaxf(f32_up(k11 * V1 * 1.001), 1e-37f);   // inf: the trust test never holds
        ln.e50 = f32_up(8.881784197001252e-16 /* 2^-50 */ * V2 * 1.001);
        ln.t2pp = f32_up(T2 * (1.0 + 1.0 / 64.0));
        ln.invT = (float)(1.0 / (T * (1.0 + 1.0 / 64.0)) * 0.99999);  // rounded DOWN: 1 / T''
        return ln;
    }
    // p = (a, b, c, mx, my, P, P^2, -) in f32
    static __device__ __forceinline__ bool reject(const float* p, const Lane& ln, float) {
        const float* v = ln.v;
        const float N = __builtin_fmaf(v[0], p[0], __builtin_fmaf(v[1], p[1], v[2] * p[2]));
        const float S = __builtin_fmaf(fabsf(v[0]), fabsf(p[0]), __builtin_fmaf(fabsf(v[1]), fabsf(p[1]), fabsf(v[2]) * fabsf(p[2])));
        const float lx = __builtin_fmaf(p[4], v[2], -v[1]);
        const float ly = __builtin_fmaf(-p[3], v[2], v[0]);
        const float D2 = __builtin_fmaf(lx, lx, ly * ly);
        const float tt = __builtin_fmaf(ln.e2, p[6], __builtin_fmaf(ln.e1, p[5], ln.e0));
        const float m = fabsf(N) - __builtin_fmaf(4.76837158203125e-7f /* 8 u */, S, ln.e50 * p[6]);
        return (D2 >= tt * tt) && (m > 0.0f) && (m * m > ln.t2pp * D2);  // every comparison is false on NaN
    }
    // g = (A, B, C, MX, MY, rA, rB, rC, rM, hmin, Pmax, P2max)
    static __device__ __forceinline__ bool group_reject(const float* g, const Lane& ln, float) {
        const float* v = ln.v;
        if (ln.nanh != 0.0f) return true;  // NaN entry in the hypothesis: every residual is NaN, never an inlier
        const float N = __builtin_fmaf(v[0], g[0], __builtin_fmaf(v[1], g[1], v[2] * g[2]));
        const float S = __builtin_fmaf(fabsf(v[0]), fabsf(g[0]), __builtin_fmaf(fabsf(v[1]), fabsf(g[1]), fabsf(v[2]) * fabsf(g[2])));
        const float RN = __builtin_fmaf(fabsf(v[0]), g[5], __builtin_fmaf(fabsf(v[1]), g[6], fabsf(v[2]) * g[7]));
        const float lx = __builtin_fmaf(g[4], v[2], -v[1]);
        const float ly = __builtin_fmaf(-g[3], v[2], v[0]);
        const float D2 = __builtin_fmaf(lx, lx, ly * ly);
        const float RD = fabsf(v[2]) * g[8] * 1.001f;
        const float tt0 = __builtin_fmaf(ln.e2, g[11], __builtin_fmaf(ln.e1, g[10], ln.e0));
        const float tt = tt0 + RD;  // trust at Dc - R_D
        const float L = fabsf(N) - RN * 1.001f - 1.9073486328125e-6f /* 32 u */ * S;
        const float G = L * g[9] * ln.invT - RD;
        // Trust: every member's D must stay above t(Pmax).  Two lower bounds on D_i: Dc - R_D (midpoints within rM of the centre), and
        // the member's own |N^_i| >= L - the distance of the vanishing point from the segment's LINE never exceeds its distance from the
        // midpoint ON that line (D_i = |v2| |m_i - vp| >= |v2| dist(vp, line_i) = |N^_i|).  The second one is what lets groups of
        // collinear segments spread over the whole image (Hough order, setpoints.hip) be culled: their rM is hundreds of pixels.
        const bool trust = (D2 >= tt * tt * 1.001f) || (L >= tt0 * 1.01f);
        return trust && (G > 0.0f) && (G * G > D2 * 1.004f);
    }
};

// ---- fundamental matrices: Sampson distance [U-2] (residuals.hip.h Residual<kFundamental>) -----------------------------------
// n(p) = x_b^T F x_a (x_a = (p0, p1, 1), x_b = (p2, p3, 1)) is BILINEAR in the four image coordinates and the Sampson
// denominator is the squared norm of its gradient: (rxc, ryc, rx, ry) = (dn/dp0, dn/dp1, dn/dp2, dn/dp3).  So the squared
// residual is n^2 / |grad n|^2 and the filter needs neither the division nor a root: 21 f32 operations per pair.
// With A4 = |f0| + |f1| + |f3| + |f4|, B4 = |f2| + |f5| + |f6| + |f7|, P = max(|coordinates|, 1) rounded up, u = 2^-24,
// eta = 2^-126 (an operation that underflows may be flushed), tau = 2^-10:
//   |n~ - n*| and |n_c - n*| (the exact path's own f64 evaluation) together  <= E_n = 8.5 u (A4 P^2 + B4 P + |f8|) + 16 eta P^2
//                                   (inputs rounded to f32, at most seven roundings on any of the nine terms)
//   ||grad~ - grad*|| + ||grad_c - grad*||  <= E_D = 4.1 u (2 A4 P + B4) + 8 eta P      (four components, two FMAs each)
// trust test  D~ >= t(P) = E_D / tau  gives D_c <= D~ (1 + 1.01 tau)^2 (D~ = ||grad~||), and then
//   reject  <=>  trust  and  m := |n~| - E_n > 0  and  m^2 > T2 (1 + 2^-6) D~^2
// implies the computed r_c^2 = fl(fl(n_c^2) / D_c) >= m^2 / (D~^2 (1 + 1.01 tau)^2) (1 - 3 eps) > T2: the exact path would not have
// accepted.  NaN / Inf make a comparison false: not rejected; a hypothesis with |f_k| Pmax^2 > 1e36 for some entry (a term of
// n~ could overflow f32 and not the one that cancels it) gets t(P) = inf and is never rejected; one with a NaN entry has NaN
// residuals everywhere (all nine entries enter n) and is culled outright.
// Group test: for a member c + d of the group (d1, d2 = the offsets in the two images, |d1| <= r1, |d2| <= r2, |d| <= R),
//   n(c + d) = n(c) + grad n(c) . d + d2^T A d1,   A = (f0 f1; f3 f4),        grad n(c + d) = grad n(c) + (A^T d2, A d1),
// so |n| >= |n(c)| - ||grad n(c)|| R - ||A|| r1 r2 and ||grad n|| <= ||grad n(c)|| + ||A|| R: no member is an inlier when
//   |n~(c)| - E_n - (G + E_D) R - ||A||_F r1 r2  >  T'' (G + E_D + ||A||_F R),   G = ||grad~(c)||,
// the error terms taken at the group's largest P (the centre lies inside the box), every subtracted term inflated.
// Groups are 64 consecutive points of the Morton order of all four coordinates (setpoints.hip): 51 % of the (hypothesis,
// group) pairs of the C3 set are culled (scripts/analysis_sampson_bound.py; per-coordinate extents would give 53 %).
template <> struct Filter32<kFundamental> {
    static constexpr bool enabled = true;
    static constexpr int kRowVals = 7, kGroupVals = 9;
    struct Lane { float f[9]; float e1, e0, n2, n1, n0, t2pp, nA, nanh; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& f0, double pscale2 /* max(|coordinate|, 1)^2 over the point set */, double T2) {
        Lane ln;
        bool nan = false, big = false;
        bool off;
        const double sc = pow2_normaliser<9>(f0, &off);
        big = off;
        double f[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { f[k] = f0[k] * sc; ln.f[k] = (float)f[k]; nan |= !(f0[k] == f0[k]); big |= !(fabs(f[k]) * pscale2 <= 1e36); }
        ln.nanh = nan ? 1.0f : 0.0f;
        const double A4 = fabs(f[0]) + fabs(f[1]) + fabs(f[3]) + fabs(f[4]), B4 = fabs(f[2]) + fabs(f[5]) + fabs(f[6]) + fabs(f[7]);
        const double u = 5.9604644775390625e-8, eta = 1.1754943508222875e-38, itau = 1024.0;
        ln.e1 = f32_up((4.1 * u * 2.0 * A4 + 8.0 * eta) * itau);
        ln.e0 = big ? __builtin_inff() : fmaxf(f32_up(4.1 * u * B4 * itau), 1e-12f);
        ln.n2 = f32_up(8.5 * u * A4 + 16.0 * eta);
        ln.n1 = f32_up(8.5 * u * B4);
        ln.n0 = f32_up(8.5 * u * fabs(f[8]));
        ln.t2pp = f32_up(T2 * (1.0 + 1.0 / 64.0));
        ln.nA = f32_up(sqrt(f[0] * f[0] + f[1] * f[1] + f[3] * f[3] + f[4] * f[4]) * 1.001);
        return ln;
    }
    // p = (x_a, y_a, x_b, y_b, -, P, P^2, -) in f32
    static __device__ __forceinline__ bool reject(const float* p, const Lane& ln, float) {
        const float* f = ln.f;
        const float rxc = __builtin_fmaf(f[0], p[2], __builtin_fmaf(f[3], p[3], f[6]));
        const float ryc = __builtin_fmaf(f[1], p[2], __builtin_fmaf(f[4], p[3], f[7]));
        const float rwc = __builtin_fmaf(f[2], p[2], __builtin_fmaf(f[5], p[3], f[8]));
        const float n = __builtin_fmaf(p[0], rxc, __builtin_fmaf(p[1], ryc, rwc));
        const float rx = __builtin_fmaf(f[0], p[0], __builtin_fmaf(f[1], p[1], f[2]));
        const float ry = __builtin_fmaf(f[3], p[0], __builtin_fmaf(f[4], p[1], f[5]));
        const float D2 = __builtin_fmaf(rxc, rxc, __builtin_fmaf(ryc, ryc, __builtin_fmaf(rx, rx, ry * ry)));
        const float tt = __builtin_fmaf(ln.e1, p[5], ln.e0);
        const float m = fabsf(n) - __builtin_fmaf(ln.n2, p[6], __builtin_fmaf(ln.n1, p[5], ln.n0));
        return (D2 >= tt * tt) && (m > 0.0f) && (m * m > ln.t2pp * D2);  // every comparison is false on NaN
    }
    // g = (ca_x, ca_y, cb_x, cb_y, r1, r2, R, Pmax, P2max, -, -, -)
    static __device__ __forceinline__ bool group_reject(const float* g, const Lane& ln, float Tup) {
        const float* f = ln.f;
        if (ln.nanh != 0.0f) return true;  // NaN entry in the hypothesis: every residual is NaN, never an inlier
        const float rxc = __builtin_fmaf(f[0], g[2], __builtin_fmaf(f[3], g[3], f[6]));
        const float ryc = __builtin_fmaf(f[1], g[2], __builtin_fmaf(f[4], g[3], f[7]));
        const float rwc = __builtin_fmaf(f[2], g[2], __builtin_fmaf(f[5], g[3], f[8]));
        const float n = __builtin_fmaf(g[0], rxc, __builtin_fmaf(g[1], ryc, rwc));
        const float rx = __builtin_fmaf(f[0], g[0], __builtin_fmaf(f[1], g[1], f[2]));
        const float ry = __builtin_fmaf(f[3], g[0], __builtin_fmaf(f[4], g[1], f[5]));
        const float D2 = __builtin_fmaf(rxc, rxc, __builtin_fmaf(ryc, ryc, __builtin_fmaf(rx, rx, ry * ry)));
        const float ED = __builtin_fmaf(ln.e1, g[7], ln.e0) * 9.765625e-4f /* tau */;   // inf for a hypothesis beyond f32: never culled
        const float G = __builtin_sqrtf(D2) * 1.001f + ED;
        const float En = __builtin_fmaf(ln.n2, g[8], __builtin_fmaf(ln.n1, g[7], ln.n0));
        const float L = fabsf(n) - En - (G * g[6] + ln.nA * g[4] * g[5]) * 1.001f;
        const float U = __builtin_fmaf(ln.nA, g[6], G) * 1.001f;
        return L > Tup * U;  // false on NaN / Inf arithmetic
    }
};

// ---- symmetric transfer error: the homography filter and group bound on the forward part (see Filter<kHomographySym>) --------
template <> struct Filter32<kHomographySym> : Filter32<kHomography> {
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& h, double guard32, double T2) {
        Lane ln = Filter32<kHomography>::prep(h, guard32, T2);
        bool nan = false;
#pragma unroll
        for (int k = 0; k < 18; ++k) nan |= !(h[k] == h[k]);   // a NaN anywhere (also in the inverse) makes every residual NaN
        ln.nanh = nan ? 1.0f : 0.0f;
        return ln;
    }
};

// ---- 2-D lines [U-4]: r = |a x + b y + c| (Residual<kLine2D>), inlier iff r^2 < T2 ------------------------------------------------
// Three f32 FMAs; with P = max(|x|, |y|, 1) rounded up, u = 2^-24, eta = 2^-126:
//   |n~ - n*| + |n_c - n*| (the exact path's own rounding)  <= E = 4.1 u ((|a| + |b|) P + |c|) + 4 eta P
//   reject  <=>  m := |n~| - E > T'' = T (1 + 2^-6):  then |n_c| > T (1 + 2^-6) and the computed r_c^2 = fl(n_c^2) > T2.
// A hypothesis with an entry times the largest P of the set beyond 1e36 gets E = inf (never rejected), one with a NaN entry
// is culled outright (all three entries enter every residual).  Group test on the 2-D box of 64 Morton-consecutive points
// (centre, radius R): |n(x)| >= |n(centre)| - ||(a, b)|| R, the error term at the group's largest P.  The dense kernel costs
// ~10 instructions per pair, so the filter itself buys nothing - the cull does: a line's inliers are a strip of width 2T.
template <> struct Filter32<kLine2D> {
    static constexpr bool enabled = true;
    static constexpr int kRowVals = 6, kGroupVals = 4;
    struct Lane { float a, b, c, e1, e0, nrm, tpp, nanh; };
    template <class MD> static __device__ __forceinline__ Lane prep(const MD& m0, double pscale /* max(|coordinate|, 1) over the set */, double T2) {
        Lane ln;
        // r = |a x + b y + c| scales with the model: the scaled copy is tested against the scaled threshold (sc is a power of two)
        bool off;
        const double sc = pow2_normaliser<3>(m0, &off);
        const double m[3] = {m0[0] * sc, m0[1] * sc, m0[2] * sc};
        ln.a = (float)m[0]; ln.b = (float)m[1]; ln.c = (float)m[2];
        ln.nanh = (m0[0] == m0[0] && m0[1] == m0[1] && m0[2] == m0[2]) ? 0.0f : 1.0f;
        const double Ts = sqrt(T2) * sc;   // the threshold in the scaled model's units: must be an ordinary f32 as well
        const bool big = !(fabs(m[0]) * pscale <= 1e36) || !(fabs(m[1]) * pscale <= 1e36) || !(fabs(m[2]) <= 1e36) || !(Ts > 1e-30) || !(Ts < 1e30) || off;
        const double u = 5.9604644775390625e-8, eta = 1.1754943508222875e-38;
        ln.e1 = f32_up(4.1 * u * (fabs(m[0]) + fabs(m[1])) + 4.0 * eta);
        ln.e0 = big ? __builtin_inff() : f32_up(4.1 * u * fabs(m[2]) + eta);
        ln.nrm = f32_up(sqrt(m[0] * m[0] + m[1] * m[1]) * 1.001);
        ln.tpp = f32_up(Ts * (1.0 + 1.0 / 64.0));
        return ln;
    }
    // p = (x, y, -, -, -, P, -, -) in f32
    static __device__ __forceinline__ bool reject(const float* p, const Lane& ln, float) {
        const float n = __builtin_fmaf(ln.a, p[0], __builtin_fmaf(ln.b, p[1], ln.c));
        const float m = fabsf(n) - __builtin_fmaf(ln.e1, p[5], ln.e0);
        return m > ln.tpp;  // false on NaN / inf - inf
    }
    // g = (cx, cy, R, Pmax)
    static __device__ __forceinline__ bool group_reject(const float* g, const Lane& ln, float) {
        if (ln.nanh != 0.0f) return true;  // NaN entry in the hypothesis: every residual is NaN, never an inlier
        const float n = __builtin_fmaf(ln.a, g[0], __builtin_fmaf(ln.b, g[1], ln.c));
        const float m = fabsf(n) - __builtin_fmaf(ln.e1, g[3], ln.e0) - ln.nrm * g[2] * 1.001f;
        return m > ln.tpp * 1.001f;
    }
};

}  // namespace pgx
