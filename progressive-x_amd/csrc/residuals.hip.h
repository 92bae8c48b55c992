// residuals.hip.h — per-(point, model) residuals for the five Progressive-X problem types, gfx950 device code.
//
// FP64 throughout, compiled with -ffp-contract=off: the reference is built for baseline x86-64 (no FMA,
// /root/reference/CMakeLists.txt:23) and parity of inlier masks is bit-exact, so every product and sum is
// rounded separately; `/` and sqrt() lower to the correctly-rounded AMDGPU expansions.
//
// Each functor documents the reference interface it replaces (paths relative to
// /root/reference/src/pyprogressivex/).  U-n = upstream source absent from the snapshot (graph-cut-ransac
// submodule is empty), restated from the published definition; see DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>

namespace pgx {

enum ModelType : int {
    kLine2D = 0, kHomography = 1, kFundamental = 2, kPnP = 3, kVanishingPoint = 4, kHomographySym = 5,
    kNumModelTypes = 6
};

// OpenCV's MIN/MAX macros (the reference sees them via progx_model.h:36): MAX(a,b) ((a) < (b) ? (b) : (a)).
__device__ __forceinline__ double cv_max(double a, double b) { return a < b ? b : a; }
__device__ __forceinline__ double cv_min(double a, double b) { return a > b ? b : a; }

template <int MT> struct Residual;

// Default2DLineEstimator (progressivex_python.cpp:489) [U-4]: model (a,b,c), r = |a x + b y + c|.
template <> struct Residual<kLine2D> {
    static constexpr int D = 2, P = 3;
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& m) {
        return fabs(m[0] * p[0] + m[1] * p[1] + m[2]);
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& m) {
        const double r = plain(p, m);
        return r * r;
    }
};

// DefaultHomographyEstimator (progressivex_python.cpp:252) [U-1]: one-way forward transfer error,
// H row-major 3x3 (progressivex_python.cpp:292-300).
template <> struct Residual<kHomography> {
    static constexpr int D = 4, P = 9;
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& h) {
        const double t1 = h[0] * p[0] + h[1] * p[1] + h[2];
        const double t2 = h[3] * p[0] + h[4] * p[1] + h[5];
        const double t3 = h[6] * p[0] + h[7] * p[1] + h[8];
        const double d1 = p[2] - (t1 / t3);
        const double d2 = p[3] - (t2 / t3);
        return d1 * d1 + d2 * d2;
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& m) { return sqrt(squared(p, m)); }
};

// Symmetric transfer error (north-star wording): model = [H | H^-1].
template <> struct Residual<kHomographySym> {
    static constexpr int D = 4, P = 18;
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& h) {
        const double t1 = h[0] * p[0] + h[1] * p[1] + h[2];
        const double t2 = h[3] * p[0] + h[4] * p[1] + h[5];
        const double t3 = h[6] * p[0] + h[7] * p[1] + h[8];
        const double d1 = p[2] - (t1 / t3);
        const double d2 = p[3] - (t2 / t3);
        const double s1 = h[9] * p[2] + h[10] * p[3] + h[11];
        const double s2 = h[12] * p[2] + h[13] * p[3] + h[14];
        const double s3 = h[15] * p[2] + h[16] * p[3] + h[17];
        const double e1 = p[0] - (s1 / s3);
        const double e2 = p[1] - (s2 / s3);
        return (d1 * d1 + d2 * d2) + (e1 * e1 + e2 * e2);
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& m) { return sqrt(squared(p, m)); }
};

// DefaultFundamentalMatrixEstimator (progressivex_python.cpp:616) [U-2]: squared Sampson distance,
// F row-major (progressivex_python.cpp:654-662).
template <> struct Residual<kFundamental> {
    static constexpr int D = 4, P = 9;
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& f) {
        const double rxc = f[0] * p[2] + f[3] * p[3] + f[6];
        const double ryc = f[1] * p[2] + f[4] * p[3] + f[7];
        const double rwc = f[2] * p[2] + f[5] * p[3] + f[8];
        const double r = p[0] * rxc + p[1] * ryc + rwc;
        const double rx = f[0] * p[0] + f[1] * p[1] + f[2];
        const double ry = f[3] * p[0] + f[4] * p[1] + f[5];
        return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& m) { return sqrt(squared(p, m)); }
};

// DefaultPnPEstimator (progressivex_python.cpp:119) [U-3]: squared reprojection error in normalised image
// coordinates; P=[R|t] row-major 3x4 (progressivex_python.cpp:156-167); row (u,v,X,Y,Z) (:88-92).
template <> struct Residual<kPnP> {
    static constexpr int D = 5, P = 12;
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& m) {
        const double px = m[0] * p[2] + m[1] * p[3] + m[2] * p[4] + m[3];
        const double py = m[4] * p[2] + m[5] * p[3] + m[6] * p[4] + m[7];
        const double pz = m[8] * p[2] + m[9] * p[3] + m[10] * p[4] + m[11];
        const double du = p[0] - (px / pz);
        const double dv = p[1] - (py / pz);
        return du * du + dv * dv;
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& m) { return sqrt(squared(p, m)); }
};

// VanishingPointEstimator::residual, vanishing_point_estimator.h:166-189 (squared at :134-140), in-tree.
template <> struct Residual<kVanishingPoint> {
    static constexpr int D = 4, P = 3;
    template <class PT, class MD>
    static __device__ __forceinline__ double plain(const PT& p, const MD& v) {
        const double mx = (p[0] + p[2]) / 2.0, my = (p[1] + p[3]) / 2.0;
        const double lx = my * v[2] - v[1];
        const double ly = -(mx * v[2] - v[0]);
        const double lz = mx * v[1] - my * v[0];
        return fabs(lx * p[0] + ly * p[1] + lz) / sqrt(lx * lx + ly * ly);
    }
    template <class PT, class MD>
    static __device__ __forceinline__ double squared(const PT& p, const MD& m) {
        const double r = plain(p, m);
        return r * r;
    }
};

// Host-side dims table (same numbers as the functors above).
inline int model_dims(int mt, int* d, int* p) {
    static const int D[kNumModelTypes] = {2, 4, 4, 5, 4, 4};
    static const int P[kNumModelTypes] = {3, 9, 9, 12, 3, 18};
    if (mt < 0 || mt >= kNumModelTypes) return -1;
    if (d) *d = D[mt];
    if (p) *p = P[mt];
    return 0;
}

}  // namespace pgx
