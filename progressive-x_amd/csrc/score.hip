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

namespace pgx {

constexpr int kScoreBlock = 256;

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
// the operation order of Residual<kHomography> (residuals.cuh) - rounding is monotone and the backward term is >= 0 or NaN -
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
// largest P.  Groups are built from the Morton order of (mx, my, orientation) - setpoints.hip.
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
        const float tt = __builtin_fmaf(ln.e2, g[11], __builtin_fmaf(ln.e1, g[10], ln.e0)) + RD;  // trust at Dc - R_D
        const float L = fabsf(N) - RN * 1.001f - 1.9073486328125e-6f /* 32 u */ * S;
        const float G = L * g[9] * ln.invT - RD;
        return (D2 >= tt * tt * 1.001f) && (G > 0.0f) && (G * G > D2 * 1.004f);
    }
};

// ---- fundamental matrices: Sampson distance [U-2] (residuals.cuh Residual<kFundamental>) -----------------------------------
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
#pragma unroll
        for (int q = 0; q < F32::kRowVals; ++q) p32[q] = p32_g[((int64_t)g * 8 + q) * 64 + lane];
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
            while (((int64_t)1 << lg) < ctx->n + 1) ++lg;
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
            const int split_cfg = ctx->score_split > 0 ? ctx->score_split : ((group_xcd || MT == kPnP) ? 5 : 8);
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
            if (ctx->score_profile >= 2) PGX_HIP(ctx, hipEventRecord(ctx->kev[3], ctx->stream));
            ctx->last_score_path = 2;
            return PGX_OK;
        }
    }
    ctx->last_score_path = 1;
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
