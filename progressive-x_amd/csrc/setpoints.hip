// setpoints.hip — pgx_set_points on the device: everything the score path derives from the points once per problem.
//
// Replaces the host loops of round 1 (58-89 ms at N = 1e6: filter scales, f32 rows, Morton keys, a 4-pass radix sort,
// sorted copies, per-group bounds) by kernels on the context's stream; the caller's buffer is read exactly once (the
// upload).  What is computed is unchanged - same keys, a stable sort, the same f64 arithmetic for the bounds - so the
// sorted order, the group rows and therefore every culling decision are those of the host version (PGX_SETPOINTS_HOST=1
// keeps it for A/B; a GPU test compares the two bit for bit).
//
// Serves: the resident data of MSACScoringFunctionWithCompoundModel::getScore
// (/root/reference/src/pyprogressivex/include/scoring_function_with_compound_model.h:61-125), which upstream re-reads from a
// cv::Mat per hypothesis; the reference has no preprocessing step of its own.
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "pgx_internal.h"

namespace pgx {

namespace {

constexpr int kSpBlock = 256;

// order-preserving map double -> u64 (for atomicMin / atomicMax on doubles)
__device__ __forceinline__ unsigned long long f64_key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
inline double key_f64(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    std::memcpy(&x, &b, 8);
    return x;
}

// stats: [0..4] min key per dim, [5..9] max key per dim, [10] max |observed| key, [11] max |coord| key (NaN ignored),
//        [12] flags: bit 0 = a non-finite coordinate, bit 1 = a NaN among the observed coordinates
__global__ __launch_bounds__(kSpBlock) void sp_prep_kernel(const double* __restrict__ pts, int64_t n, int d, int obs0, int in0, int in1,
                                                           double* __restrict__ pmax, float* __restrict__ p32,
                                                           unsigned long long* __restrict__ stats)
{
    __shared__ unsigned long long s_min[5], s_max[5], s_um, s_fs;
    __shared__ unsigned s_flag;
    if (threadIdx.x < 5) { s_min[threadIdx.x] = ~0ull; s_max[threadIdx.x] = 0ull; }
    if (threadIdx.x == 0) { s_um = 0ull; s_fs = 0ull; s_flag = 0u; }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (i < n) {
        double r[5];
        unsigned flag = 0;
        for (int k = 0; k < d; ++k) {
            r[k] = pts[i * d + k];
            if (!(fabs(r[k]) <= 1.7976931348623157e308)) flag |= 1u;  // NaN or Inf
            atomicMin(&s_min[k], f64_key(r[k]));   // NaN keys sort above +Inf / below -Inf: harmless, the flag decides
            atomicMax(&s_max[k], f64_key(r[k]));
            const double a = fabs(r[k]);
            if (a == a) atomicMax(&s_fs, f64_key(a));
        }
        double pm = 1.0;
        if (obs0 >= 0) {
            const double a = fabs(r[obs0]), b = fabs(r[obs0 + 1]);
            if (!(a == a) || !(b == b)) flag |= 2u;
            else { atomicMax(&s_um, f64_key(a)); atomicMax(&s_um, f64_key(b)); }
            for (int k = in0; k <= in1; ++k) { const double v = fabs(r[k]); if (!(v <= pm)) pm = v; }
            float* q = p32 + i * 8;   // f32 row of the pre-filter: coordinates, then the scale rounded up
            for (int k = 0; k < 8; ++k) q[k] = 0.0f;
            for (int k = 0; k < d; ++k) q[k] = (float)r[k];
            q[5] = (float)(pm * 1.000001);
        } else {
            float* q = p32 + i * 8;
            for (int k = 0; k < 8; ++k) q[k] = 0.0f;
        }
        pmax[i] = pm;
        if (flag) atomicOr(&s_flag, flag);
    }
    __syncthreads();
    if ((int)threadIdx.x < d) {
        atomicMin(&stats[threadIdx.x], s_min[threadIdx.x]);
        atomicMax(&stats[5 + threadIdx.x], s_max[threadIdx.x]);
    }
    if (threadIdx.x == 0) {
        atomicMax(&stats[10], s_um);
        atomicMax(&stats[11], s_fs);
        if (s_flag) atomicOr(&stats[12], (unsigned long long)s_flag);
    }
}

struct MortonArg {
    double lo[5], inv[5];
    int d, bits;
};

__global__ __launch_bounds__(kSpBlock) void sp_keys_kernel(const double* __restrict__ pts, int64_t n, MortonArg m,
                                                           unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned qmax = (1u << m.bits) - 1u;
    unsigned key = 0;
    for (int k = 0; k < m.d; ++k) {
        const double t = (pts[i * m.d + k] - m.lo[k]) * m.inv[k];
        unsigned v = t > 0.0 ? (unsigned)t : 0u;
        if (v > qmax) v = qmax;
        for (int b = 0; b < m.bits; ++b) key |= ((v >> b) & 1u) << (b * m.d + m.d - 1 - k);  // bit b of coordinate k
    }
    keys[i] = key;
    vals[i] = (unsigned)i;
}


// ---- k-d order (PGX_SP_KD, pose problems): median splits of the widest dimension instead of a Morton curve ----------------
// The group bound's slack is (radius of the 3-D part) x (projection scale) + (half extent of the observed part): a Morton cell
// is as wide in every normalised coordinate, a k-d leaf is balanced by construction (every split halves the widest extent,
// whole groups of 64 on either side) and adapts to the density - 89-99 surviving hypotheses per group against 135
// (scripts/analysis_cull_bound.py).  Built level by level: the segment boundaries depend on n alone (host), a level is a
// segmented min/max, a key (node << 32 | coordinate along the node's widest dimension) and one stable radix sort of the pairs.
__device__ __forceinline__ unsigned f32_ord(float x)
{
    const unsigned b = __float_as_uint(x);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ int kd_node_of(const int* __restrict__ seg, int nseg, int pos)   // last s with seg[s] <= pos
{
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
struct KdScale { double s[5]; };

__global__ __launch_bounds__(kSpBlock) void sp_kd_extent_kernel(const double* __restrict__ pts, int64_t n, int d, KdScale sc,
                                                                const unsigned* __restrict__ order, const int* __restrict__ seg, int nseg,
                                                                unsigned* __restrict__ mn /* [nseg][5] */, unsigned* __restrict__ mx)
{
    // Every segment starts at a multiple of 64 positions (the splits send whole groups to the left), so the 64 positions of a
    // wave lie in ONE node: a shuffle reduction per wave, then one atomic per (wave, dimension) instead of one per point.
    const int64_t pos = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    const int64_t wave0 = pos & ~(int64_t)63;
    if (wave0 >= n) return;
    const bool live = pos < n;
    const int node = kd_node_of(seg, nseg, (int)wave0);
    unsigned lo[5], hi[5];
    for (int k = 0; k < 5; ++k) { lo[k] = 0xffffffffu; hi[k] = 0u; }
    if (live) {
        const unsigned i = order[pos];
        for (int k = 0; k < d; ++k) lo[k] = hi[k] = f32_ord((float)(pts[(int64_t)i * d + k] * sc.s[k]));
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < d; ++k) {
            const unsigned a = __shfl_xor(lo[k], off, 64), b = __shfl_xor(hi[k], off, 64);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
    const int lane = (int)(threadIdx.x & 63);
    if (lane < d) {
        unsigned vlo = lo[0], vhi = hi[0];
        for (int k = 1; k < d; ++k) if (lane == k) { vlo = lo[k]; vhi = hi[k]; }
        atomicMin(&mn[node * 5 + lane], vlo);
        atomicMax(&mx[node * 5 + lane], vhi);
    }
}

__device__ __forceinline__ float ord_f32(unsigned k)
{
    const unsigned b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

__global__ __launch_bounds__(kSpBlock) void sp_kd_keys_kernel(const double* __restrict__ pts, int64_t n, int d, KdScale sc,
                                                              const unsigned* __restrict__ order, const int* __restrict__ seg, int nseg,
                                                              const unsigned* __restrict__ mn, const unsigned* __restrict__ mx,
                                                              unsigned* __restrict__ keys, int qbits)
{
    const int64_t pos = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (pos >= n) return;
    const int node = kd_node_of(seg, nseg, (int)pos);
    int best = 0;
    float ext = -1.0f, lo = 0.0f;
    for (int k = 0; k < d; ++k) {   // widest extent (first among equals); a NaN extent never wins
        const float a = ord_f32(mn[node * 5 + k]), e = ord_f32(mx[node * 5 + k]) - a;
        if (e > ext) { ext = e; best = k; lo = a; }
    }
    // position along that dimension, quantised to qbits inside the node's own extent: (node, position) fits a 32-bit key and
    // the sort needs half the passes of a 64-bit one; equal positions keep their order (stable sort)
    const unsigned i = order[pos];
    const float x = (float)(pts[(int64_t)i * d + best] * sc.s[best]);
    const float qmax = (float)((1u << qbits) - 1u);
    float t = ext > 0.0f ? (x - lo) / ext * qmax : 0.0f;
    t = t > 0.0f ? (t < qmax ? t : qmax) : 0.0f;   // (NaN -> 0)
    keys[pos] = ((unsigned)node << qbits) | (unsigned)t;
}

__global__ __launch_bounds__(kSpBlock) void sp_iota_kernel(unsigned* __restrict__ v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}

// sorted copies: AoS rows (chunked kernel, exact kernel), f32 rows, scales, and the group-blocked SoA copies
__global__ __launch_bounds__(kSpBlock) void sp_gather_kernel(const double* __restrict__ pts, const float* __restrict__ p32,
                                                             const double* __restrict__ pmax, const unsigned* __restrict__ order,
                                                             int64_t n, int d, int64_t padded, int* __restrict__ pperm,
                                                             double* __restrict__ pts_s, float* __restrict__ p32_s,
                                                             double* __restrict__ pmax_s, double* __restrict__ pts_g,
                                                             float* __restrict__ p32_g)
{
    const int64_t j = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (j >= padded) return;
    const int64_t jj = j < n ? j : n - 1;   // the tail group repeats its last row
    const int64_t i = (int64_t)order[jj];
    const int64_t g = j >> 6;
    const int l = (int)(j & 63);
    for (int k = 0; k < d; ++k) {
        const double v = pts[i * d + k];
        if (j < n) pts_s[j * d + k] = v;
        pts_g[(g * d + k) * 64 + l] = v;
    }
    for (int k = 0; k < 8; ++k) {
        const float v = p32[i * 8 + k];
        if (j < n) p32_s[j * 8 + k] = v;
        p32_g[(g * 8 + k) * 64 + l] = v;
    }
    if (j < n) { pperm[j] = (int)i; pmax_s[j] = pmax[i]; }
}

// bounds of `span` consecutive sorted points per workgroup (64 = a group, 512 = a super-group); the arithmetic of the
// host version (score_sort_points): box centres in f64 -> f32, radius about the STORED f32 centre, inflated extents
template <int SPAN>
__global__ __launch_bounds__(SPAN) void sp_bounds_kernel(const double* __restrict__ sp, int64_t n, int d, int in0, int in1, int ob0,
                                                         float* __restrict__ rows /* first row of this level */)
{
    __shared__ double s_lo[5], s_hi[5];
    __shared__ unsigned long long s_red[3];
    __shared__ float s_c[3], s_ob[2];
    const int64_t a = (int64_t)blockIdx.x * SPAN, j = a + threadIdx.x;
    const bool valid = j < n;
    if (threadIdx.x < 5) { s_lo[threadIdx.x] = 0.0; s_hi[threadIdx.x] = 0.0; }
    if (threadIdx.x < 3) s_red[threadIdx.x] = 0ull;
    __syncthreads();
    double r[5] = {0, 0, 0, 0, 0};
    if (valid)
        for (int k = 0; k < d; ++k) r[k] = sp[j * d + k];
    // min / max of every coordinate over the valid points (wave shuffles, then LDS across the waves)
    __shared__ double s_wlo[SPAN / 64][5], s_whi[SPAN / 64][5];
    for (int k = 0; k < d; ++k) {
        double lo = valid ? r[k] : 1.7976931348623157e308, hi = valid ? r[k] : -1.7976931348623157e308;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            if (l2 < lo) lo = l2;
            if (h2 > hi) hi = h2;
        }
        if ((threadIdx.x & 63) == 0) { s_wlo[threadIdx.x >> 6][k] = lo; s_whi[threadIdx.x >> 6][k] = hi; }
    }
    __syncthreads();
    if ((int)threadIdx.x < d) {
        double lo = s_wlo[0][threadIdx.x], hi = s_whi[0][threadIdx.x];
        for (int w = 1; w < SPAN / 64; ++w) {
            if (s_wlo[w][threadIdx.x] < lo) lo = s_wlo[w][threadIdx.x];
            if (s_whi[w][threadIdx.x] > hi) hi = s_whi[w][threadIdx.x];
        }
        s_lo[threadIdx.x] = lo;
        s_hi[threadIdx.x] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k <= in1 - in0; ++k) s_c[k] = (float)(0.5 * (s_lo[in0 + k] + s_hi[in0 + k]));
        for (int k = in1 - in0 + 1; k < 3; ++k) s_c[k] = 0.0f;
        s_ob[0] = (float)(0.5 * (s_lo[ob0] + s_hi[ob0]));
        s_ob[1] = (float)(0.5 * (s_lo[ob0 + 1] + s_hi[ob0 + 1]));
    }
    __syncthreads();
    double s2 = 0.0, du = 0.0, dv = 0.0;
    if (valid) {
        for (int k = in0; k <= in1; ++k) { const double df = r[k] - (double)s_c[k - in0]; s2 += df * df; }
        du = fabs(r[ob0] - (double)s_ob[0]);
        dv = fabs(r[ob0 + 1] - (double)s_ob[1]);
    }
    // maxima of non-negative doubles: their bit patterns order like unsigned integers
    atomicMax(&s_red[0], (unsigned long long)__double_as_longlong(s2));
    atomicMax(&s_red[1], (unsigned long long)__double_as_longlong(du));
    atomicMax(&s_red[2], (unsigned long long)__double_as_longlong(dv));
    __syncthreads();
    if (threadIdx.x == 0) {
        float* row = rows + (int64_t)blockIdx.x * kGroupRow;
        const double rho2 = __longlong_as_double((long long)s_red[0]), ru = __longlong_as_double((long long)s_red[1]),
                     rv = __longlong_as_double((long long)s_red[2]);
        double scale = 1.0;
        for (int k = 0; k <= in1 - in0; ++k) if (fabs((double)s_c[k]) > scale) scale = fabs((double)s_c[k]);
        row[0] = s_c[0]; row[1] = s_c[1]; row[2] = s_c[2];
        row[3] = (float)(sqrt(rho2) * kGroupInflate + 1e-30);
        row[4] = s_ob[0]; row[5] = s_ob[1];
        row[6] = (float)(ru * kGroupInflate + 1e-30);
        row[7] = (float)(rv * kGroupInflate + 1e-30);
        row[8] = (float)(scale * 1.000001);
        row[9] = row[10] = row[11] = 0.0f;
    }
}

// ---- vanishing points: oriented, length-normalised segment features (score.hip Filter32<kVanishingPoint>) ---------------
struct VpFeat {
    double a, b, c, mx, my, h, P;
};

__device__ __forceinline__ VpFeat vp_features(const double* __restrict__ r)
{
    VpFeat f;
    const double dx = r[2] - r[0], dy = r[3] - r[1];
    const bool flip = dx < 0.0 || (dx == 0.0 && dy < 0.0);   // canonical orientation: N only enters through |N|
    f.a = (r[1] - r[3]) / 2.0;
    f.b = (r[2] - r[0]) / 2.0;
    f.c = (r[0] * r[3] - r[2] * r[1]) / 2.0;
    if (flip) { f.a = -f.a; f.b = -f.b; f.c = -f.c; }
    f.mx = (r[0] + r[2]) / 2.0;
    f.my = (r[1] + r[3]) / 2.0;
    f.h = 0.5 * sqrt(dx * dx + dy * dy);
    double P = 1.0;
    for (int k = 0; k < 4; ++k) { const double v = fabs(r[k]); if (!(v <= P)) P = v; }
    f.P = P;
    return f;
}

// f32 row (a, b, c, mx, my, P, P^2, 0) and the sort key.
// Order (round 6): the group bound works on the LINE of a segment - N^ = v . (a, b, c) / h is the distance of the vanishing point from
// the line through the segment, in Hough terms the point (theta, rho) against the sinusoid of the vanishing point - so groups must be
// compact in (theta, rho), not in (midpoint, orientation): segments of one line anywhere in the image share a row, segments that
// merely lie close to each other do not.  Key = bit-interleaved (theta, rho') with 10 bits each, rho' = the signed distance of the
// line from the CENTRE of the bounding box (all 10 bits in use wherever the origin is), and 3 bits of log2(half length) under the
// top three rounds (the bound scales with the shortest member, hmin).  The bound is valid for any grouping (its radii are those of
// the actual members), so the order changes work only: 37 % -> 19 % surviving (hypothesis, group) pairs on the C5 set against a
// floor of 14 % that hold an inlier (scripts/analysis_vp_order.py); rounds 2-5 sorted by Morton(mx, my, orientation).
__global__ __launch_bounds__(kSpBlock) void sp_vp_rows_kernel(const double* __restrict__ pts, int64_t n, double cx, double cy, double rh,
                                                              float* __restrict__ p32, unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (i >= n) return;
    const VpFeat f = vp_features(pts + i * 4);
    float* q = p32 + i * 8;
    q[0] = (float)f.a; q[1] = (float)f.b; q[2] = (float)f.c; q[3] = (float)f.mx; q[4] = (float)f.my;
    q[5] = (float)(f.P * 1.000001);
    q[6] = (float)(f.P * f.P * 1.000001);
    q[7] = 0.0f;
    // orientation of the canonical direction (b, -a) = (dx, dy) / 2 with dx >= 0: angle in (-pi/2, pi/2]
    const double th = atan2(-f.a, f.b);
    const double rho = (f.a * cx + f.b * cy + f.c) / f.h;           // NaN for a zero-length segment: quantised to 0 below
    const double t[3] = {(th + 1.5707963267948966) * (1024.0 / 3.141592653589793), rh > 0.0 ? (rho + rh) * (512.0 / rh) : 0.0,
                         rh > 0.0 ? (log2(f.h / rh) + 10.0) * 0.8 : 0.0};
    unsigned v[3];
    for (int k = 0; k < 3; ++k) {
        const unsigned lim = k == 2 ? 7u : 1023u;
        v[k] = t[k] > 0.0 ? (t[k] < (double)lim ? (unsigned)t[k] : lim) : 0u;      // (NaN compares false: 0)
    }
    unsigned key = 0;
    for (int level = 0; level < 10; ++level) {     // most significant bits first
        key = (key << 1) | ((v[0] >> (9 - level)) & 1u);
        key = (key << 1) | ((v[1] >> (9 - level)) & 1u);
        if (level < 3) key = (key << 1) | ((v[2] >> (2 - level)) & 1u);
    }
    keys[i] = key;
    vals[i] = (unsigned)i;
}

// group rows (A, B, C, MX, MY, rA, rB, rC, rM, hmin, Pmax, P2max) over SPAN consecutive sorted segments
template <int SPAN>
__global__ __launch_bounds__(SPAN) void sp_vp_bounds_kernel(const double* __restrict__ sp, int64_t n, float* __restrict__ rows)
{
    __shared__ double s_wlo[SPAN / 64][5], s_whi[SPAN / 64][5];
    __shared__ float s_c[5];
    __shared__ unsigned long long s_red[6];   // radii of the 3 features, of the midpoints (squared), max P, max 1/h (as min h)
    const int64_t j = (int64_t)blockIdx.x * SPAN + threadIdx.x;
    const bool valid = j < n;
    if (threadIdx.x < 6) s_red[threadIdx.x] = 0ull;
    VpFeat f = {0, 0, 0, 0, 0, 1, 1};
    double feat[5] = {0, 0, 0, 0, 0};
    if (valid) {
        f = vp_features(sp + j * 4);
        feat[0] = f.a / f.h; feat[1] = f.b / f.h; feat[2] = f.c / f.h; feat[3] = f.mx; feat[4] = f.my;   // h = 0: NaN row, never culled
    }
    for (int k = 0; k < 5; ++k) {
        // NaN features must poison the row: min / max by comparisons would skip them, so they are mapped to +-inf first
        const bool bad = valid && !(feat[k] == feat[k]);
        double lo = valid ? (bad ? -1.0 / 0.0 : feat[k]) : 1.7976931348623157e308;
        double hi = valid ? (bad ? 1.0 / 0.0 : feat[k]) : -1.7976931348623157e308;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            if (l2 < lo) lo = l2;
            if (h2 > hi) hi = h2;
        }
        if ((threadIdx.x & 63) == 0) { s_wlo[threadIdx.x >> 6][k] = lo; s_whi[threadIdx.x >> 6][k] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double lo = s_wlo[0][threadIdx.x], hi = s_whi[0][threadIdx.x];
        for (int w = 1; w < SPAN / 64; ++w) {
            if (s_wlo[w][threadIdx.x] < lo) lo = s_wlo[w][threadIdx.x];
            if (s_whi[w][threadIdx.x] > hi) hi = s_whi[w][threadIdx.x];
        }
        s_c[threadIdx.x] = (float)(0.5 * (lo + hi));   // -inf + inf = NaN: a degenerate member poisons the centre
    }
    __syncthreads();
    if (valid) {
        double rm2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double dfk = fabs(feat[k] - (double)s_c[k]);
            atomicMax(&s_red[k], (unsigned long long)__double_as_longlong(dfk == dfk ? dfk : 1.0 / 0.0));
        }
        const double ddx = f.mx - (double)s_c[3], ddy = f.my - (double)s_c[4];
        rm2 = ddx * ddx + ddy * ddy;
        atomicMax(&s_red[3], (unsigned long long)__double_as_longlong(rm2 == rm2 ? rm2 : 1.0 / 0.0));
        atomicMax(&s_red[4], (unsigned long long)__double_as_longlong(f.P == f.P ? f.P : 1.0 / 0.0));
        const double ih = 1.0 / f.h;   // max of 1 / h = 1 / min h (h = 0 -> inf -> hmin = 0: the test can never fire)
        atomicMax(&s_red[5], (unsigned long long)__double_as_longlong(ih == ih ? ih : 1.0 / 0.0));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* row = rows + (int64_t)blockIdx.x * kGroupRow;
        for (int k = 0; k < 5; ++k) row[k] = s_c[k];
        for (int k = 0; k < 3; ++k) row[5 + k] = (float)(__longlong_as_double((long long)s_red[k]) * kGroupInflate + 1e-30);
        row[8] = (float)(sqrt(__longlong_as_double((long long)s_red[3])) * kGroupInflate + 1e-30);
        const double ihmax = __longlong_as_double((long long)s_red[5]);
        row[9] = ihmax > 0.0 ? (float)(1.0 / ihmax * 0.99999) : 0.0f;   // hmin rounded DOWN
        const double P = __longlong_as_double((long long)s_red[4]);
        row[10] = (float)(P * 1.000001);
        row[11] = (float)(P * P * 1.000001);
    }
}

// ---- fundamental matrices: f32 rows (x_a, y_a, x_b, y_b, 0, P, P^2, 0) and group rows of the 4-D boxes ---------------------
// (score.hip Filter32<kFundamental>: P = max(|coordinates|, 1) rounded up)
__global__ __launch_bounds__(kSpBlock) void sp_fund_rows_kernel(const double* __restrict__ pts, int64_t n, int d /* 4, or 2 for lines */,
                                                                float* __restrict__ p32, double* __restrict__ pmax)
{
    const int64_t i = (int64_t)blockIdx.x * kSpBlock + threadIdx.x;
    if (i >= n) return;
    double P = 1.0;
    float* q = p32 + i * 8;
    for (int k = 0; k < 4; ++k) q[k] = 0.0f;
    for (int k = 0; k < d; ++k) {
        const double v = pts[i * d + k];
        q[k] = (float)v;
        if (!(fabs(v) <= P)) P = fabs(v);
    }
    q[4] = 0.0f;
    q[5] = (float)(P * 1.000001);
    q[6] = (float)(P * P * 1.000001);
    q[7] = 0.0f;
    pmax[i] = P;
}

// group rows (ca_x, ca_y, cb_x, cb_y, r1, r2, R, Pmax, P2max, 0, 0, 0) over SPAN consecutive sorted correspondences: box centre
// in f64 -> f32, radii about the STORED f32 centre (r1 / r2: the two images, R: all four coordinates), inflated
template <int SPAN>
__global__ __launch_bounds__(SPAN) void sp_fund_bounds_kernel(const double* __restrict__ sp, int64_t n, float* __restrict__ rows)
{
    __shared__ double s_wlo[SPAN / 64][4], s_whi[SPAN / 64][4];
    __shared__ float s_c[4];
    __shared__ unsigned long long s_red[4];   // r1^2, r2^2, R^2, P
    const int64_t j = (int64_t)blockIdx.x * SPAN + threadIdx.x;
    const bool valid = j < n;
    if (threadIdx.x < 4) s_red[threadIdx.x] = 0ull;
    double r[4] = {0, 0, 0, 0};
    if (valid)
        for (int k = 0; k < 4; ++k) r[k] = sp[j * 4 + k];
    for (int k = 0; k < 4; ++k) {
        double lo = valid ? r[k] : 1.7976931348623157e308, hi = valid ? r[k] : -1.7976931348623157e308;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            if (l2 < lo) lo = l2;
            if (h2 > hi) hi = h2;
        }
        if ((threadIdx.x & 63) == 0) { s_wlo[threadIdx.x >> 6][k] = lo; s_whi[threadIdx.x >> 6][k] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double lo = s_wlo[0][threadIdx.x], hi = s_whi[0][threadIdx.x];
        for (int w = 1; w < SPAN / 64; ++w) {
            if (s_wlo[w][threadIdx.x] < lo) lo = s_wlo[w][threadIdx.x];
            if (s_whi[w][threadIdx.x] > hi) hi = s_whi[w][threadIdx.x];
        }
        s_c[threadIdx.x] = (float)(0.5 * (lo + hi));
    }
    __syncthreads();
    if (valid) {
        double d2[4], P = 1.0;
        for (int k = 0; k < 4; ++k) {
            const double df = r[k] - (double)s_c[k];
            d2[k] = df * df;
            if (fabs(r[k]) > P) P = fabs(r[k]);
        }
        const double a = d2[0] + d2[1], b = d2[2] + d2[3];
        atomicMax(&s_red[0], (unsigned long long)__double_as_longlong(a));
        atomicMax(&s_red[1], (unsigned long long)__double_as_longlong(b));
        atomicMax(&s_red[2], (unsigned long long)__double_as_longlong(a + b));
        atomicMax(&s_red[3], (unsigned long long)__double_as_longlong(P));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* row = rows + (int64_t)blockIdx.x * kGroupRow;
        for (int k = 0; k < 4; ++k) row[k] = s_c[k];
        for (int k = 0; k < 3; ++k) row[4 + k] = (float)(sqrt(__longlong_as_double((long long)s_red[k])) * kGroupInflate + 1e-30);
        const double P = __longlong_as_double((long long)s_red[3]);
        row[7] = (float)(P * 1.000002);    // >= |stored centre| too: the centre lies in the box up to its f32 rounding
        row[8] = (float)(P * P * 1.000004);
        row[9] = row[10] = row[11] = 0.0f;
    }
}

// 2-D lines: group rows (cx, cy, R, Pmax, 0 ...) - score.hip Filter32<kLine2D>
template <int SPAN>
__global__ __launch_bounds__(SPAN) void sp_line_bounds_kernel(const double* __restrict__ sp, int64_t n, float* __restrict__ rows)
{
    __shared__ double s_wlo[SPAN / 64][2], s_whi[SPAN / 64][2];
    __shared__ float s_c[2];
    __shared__ unsigned long long s_red[2];   // R^2, P
    const int64_t j = (int64_t)blockIdx.x * SPAN + threadIdx.x;
    const bool valid = j < n;
    if (threadIdx.x < 2) s_red[threadIdx.x] = 0ull;
    double r[2] = {0, 0};
    if (valid) { r[0] = sp[j * 2]; r[1] = sp[j * 2 + 1]; }
    for (int k = 0; k < 2; ++k) {
        double lo = valid ? r[k] : 1.7976931348623157e308, hi = valid ? r[k] : -1.7976931348623157e308;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            if (l2 < lo) lo = l2;
            if (h2 > hi) hi = h2;
        }
        if ((threadIdx.x & 63) == 0) { s_wlo[threadIdx.x >> 6][k] = lo; s_whi[threadIdx.x >> 6][k] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double lo = s_wlo[0][threadIdx.x], hi = s_whi[0][threadIdx.x];
        for (int w = 1; w < SPAN / 64; ++w) {
            if (s_wlo[w][threadIdx.x] < lo) lo = s_wlo[w][threadIdx.x];
            if (s_whi[w][threadIdx.x] > hi) hi = s_whi[w][threadIdx.x];
        }
        s_c[threadIdx.x] = (float)(0.5 * (lo + hi));
    }
    __syncthreads();
    if (valid) {
        const double dx = r[0] - (double)s_c[0], dy = r[1] - (double)s_c[1];
        double P = 1.0;
        if (fabs(r[0]) > P) P = fabs(r[0]);
        if (fabs(r[1]) > P) P = fabs(r[1]);
        atomicMax(&s_red[0], (unsigned long long)__double_as_longlong(dx * dx + dy * dy));
        atomicMax(&s_red[1], (unsigned long long)__double_as_longlong(P));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* row = rows + (int64_t)blockIdx.x * kGroupRow;
        row[0] = s_c[0]; row[1] = s_c[1];
        row[2] = (float)(sqrt(__longlong_as_double((long long)s_red[0])) * kGroupInflate + 1e-30);
        row[3] = (float)(__longlong_as_double((long long)s_red[1]) * 1.000002);   // >= |stored centre| too
        for (int k = 4; k < kGroupRow; ++k) row[k] = 0.0f;
    }
}

}  // namespace

int set_points_device(pgx_ctx* ctx, int model_type, const double* points, int64_t n)
{
    const int d = ctx->D;
    int obs0 = -1, in0 = 0, in1 = -1;
    if (model_type == kPnP) { obs0 = 0; in0 = 2; in1 = 4; }
    else if (model_type == kHomography || model_type == kHomographySym) { obs0 = 2; in0 = 0; in1 = 3; }  // scale over all four coordinates (Filter32<kHomography>); Sym: the forward part
    PGX_TRY(ensure(ctx, ctx->pts, (size_t)n * d * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->comp, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pmax, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pts32, (size_t)n * 8 * sizeof(float)));
    PGX_TRY(ensure(ctx, ctx->scratch, 16 * sizeof(unsigned long long)));
    unsigned long long init[16];
    for (int k = 0; k < 16; ++k) init[k] = 0ull;
    for (int k = 0; k < 5; ++k) init[k] = ~0ull;
    PGX_HIP(ctx, hipMemcpyAsync(ctx->scratch.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemcpyAsync(ctx->pts.p, points, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PGX_HIP(ctx, hipMemsetAsync(ctx->comp.p, 0, (size_t)n * sizeof(double), ctx->stream));
    const unsigned blocks = (unsigned)((n + kSpBlock - 1) / kSpBlock);
    hipLaunchKernelGGL(sp_prep_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, d, obs0, in0, in1,
                       ctx->pmax.as<double>(), ctx->pts32.as<float>(), ctx->scratch.as<unsigned long long>());
    PGX_HIP(ctx, hipGetLastError());
    unsigned long long st[16];
    PGX_HIP(ctx, hipMemcpyAsync(st, ctx->scratch.p, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also: the caller's buffer has been consumed
    const unsigned flags = (unsigned)st[12];
    ctx->umax = obs0 >= 0 ? ((flags & 2u) ? std::nan("") : (st[10] ? key_f64(st[10]) : 0.0)) : 0.0;
    const double fs = st[11] ? key_f64(st[11]) : 0.0;
    ctx->fscale = fs > 1.0 ? fs : 1.0;   // NaN coordinates leave it at what the finite ones give; the solvers then produce NaN models
    ctx->point_sort = 0;
    ctx->comp_dirty = 1;
    if (model_type == kVanishingPoint && ctx->group_filter && ctx->filter_enabled == 1 && !(flags & 1u) && n >= 1) {
        // ---- segments: f32 feature rows, Hough order of the segments' lines (sp_vp_rows_kernel), group rows of the normalised features
        const double xa = std::fmin(key_f64(st[0]), key_f64(st[2])), xb = std::fmax(key_f64(st[5]), key_f64(st[7]));
        const double ya = std::fmin(key_f64(st[1]), key_f64(st[3])), yb = std::fmax(key_f64(st[6]), key_f64(st[8]));
        const int64_t groups = (n + 63) / 64, supers = (groups + kSuper - 1) / kSuper, padded = groups * 64;
        size_t tmp_bytes = 0;
        unsigned* nullu = nullptr;
        PGX_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, nullu, nullu, nullu, nullu, (size_t)n, 0, 30u, ctx->stream));
        const size_t arr = ((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255;
        PGX_TRY(ensure(ctx, ctx->fit_scratch, 4 * arr + tmp_bytes + 256));
        unsigned* k_in = (unsigned*)ctx->fit_scratch.p;
        unsigned* k_out = (unsigned*)((char*)ctx->fit_scratch.p + arr);
        unsigned* v_in = (unsigned*)((char*)ctx->fit_scratch.p + 2 * arr);
        unsigned* v_out = (unsigned*)((char*)ctx->fit_scratch.p + 3 * arr);
        void* tmp = (char*)ctx->fit_scratch.p + 4 * arr;
        const double rh = 0.5 * std::hypot(xb - xa, yb - ya);    // every line through the box passes within rh of its centre
        hipLaunchKernelGGL(sp_vp_rows_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, 0.5 * (xa + xb), 0.5 * (ya + yb),
                           std::isfinite(rh) ? rh : 0.0, ctx->pts32.as<float>(), k_in, v_in);
        PGX_HIP(ctx, hipGetLastError());
        PGX_HIP(ctx, rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, 30u, ctx->stream));
        PGX_TRY(ensure(ctx, ctx->pts_s, (size_t)n * d * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->pts32_s, (size_t)n * 8 * sizeof(float)));
        PGX_TRY(ensure(ctx, ctx->pmax_s, (size_t)n * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->comp_s, (size_t)n * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->pperm, (size_t)n * sizeof(int)));
        PGX_TRY(ensure(ctx, ctx->pts_g, (size_t)padded * d * sizeof(double)));
        PGX_TRY(ensure(ctx, ctx->p32_g, (size_t)padded * 8 * sizeof(float)));
        PGX_TRY(ensure(ctx, ctx->gbounds, (size_t)(groups + supers) * kGroupRow * sizeof(float)));
        hipLaunchKernelGGL(sp_gather_kernel, dim3((unsigned)((padded + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, ctx->stream,
                           ctx->pts.as<double>(), ctx->pts32.as<float>(), ctx->pmax.as<double>(), v_out, n, d, padded, ctx->pperm.as<int>(),
                           ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->pmax_s.as<double>(), ctx->pts_g.as<double>(),
                           ctx->p32_g.as<float>());
        hipLaunchKernelGGL((sp_vp_bounds_kernel<64>), dim3((unsigned)groups), dim3(64), 0, ctx->stream, ctx->pts_s.as<double>(), n,
                           ctx->gbounds.as<float>());
        hipLaunchKernelGGL((sp_vp_bounds_kernel<64 * kSuper>), dim3((unsigned)supers), dim3(64 * kSuper), 0, ctx->stream,
                           ctx->pts_s.as<double>(), n, ctx->gbounds.as<float>() + groups * kGroupRow);
        PGX_HIP(ctx, hipGetLastError());
        PGX_HIP(ctx, hipMemsetAsync(ctx->comp_s.p, 0, (size_t)n * sizeof(double), ctx->stream));
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->point_sort = 1;
        return PGX_OK;
    }
    const bool line = model_type == kLine2D;
    const bool fund = model_type == kFundamental || line;   // model types whose f32 rows / boxes are built from ALL coordinates
    if (!((obs0 >= 0 || fund) && ctx->group_filter && ctx->filter_enabled == 1 && std::isfinite(ctx->umax)) || (flags & 1u)) return PGX_OK;
    if (fund) {   // f32 rows of the Sampson / line filter (the prep kernel left them zero) + the scales
        hipLaunchKernelGGL(sp_fund_rows_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, d, ctx->pts32.as<float>(),
                           ctx->pmax.as<double>());
        PGX_HIP(ctx, hipGetLastError());
    }

    // ---- Morton order of all coordinates (stable: ties keep index order), sorted copies, group bounds
    MortonArg m;
    m.d = d;
    m.bits = 30 / d;
    for (int k = 0; k < 5; ++k) { m.lo[k] = 0.0; m.inv[k] = 0.0; }
    for (int k = 0; k < d; ++k) {
        const double a = key_f64(st[k]), b = key_f64(st[5 + k]);
        m.lo[k] = a;
        m.inv[k] = b > a ? (double)(1u << m.bits) / (b - a) : 0.0;
    }
    const int64_t groups = (n + 63) / 64, supers = (groups + kSuper - 1) / kSuper, padded = groups * 64;
    size_t tmp_bytes = 0;
    unsigned* nullu = nullptr;
    PGX_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, nullu, nullu, nullu, nullu, (size_t)n, 0, (unsigned)(m.bits * d), ctx->stream));
    const size_t arr = ((size_t)n * sizeof(unsigned) + 255) & ~(size_t)255;
    PGX_TRY(ensure(ctx, ctx->fit_scratch, 4 * arr + tmp_bytes + 256));
    unsigned* k_in = (unsigned*)ctx->fit_scratch.p;
    unsigned* k_out = (unsigned*)((char*)ctx->fit_scratch.p + arr);
    unsigned* v_in = (unsigned*)((char*)ctx->fit_scratch.p + 2 * arr);
    unsigned* v_out = (unsigned*)((char*)ctx->fit_scratch.p + 3 * arr);
    void* tmp = (char*)ctx->fit_scratch.p + 4 * arr;
    bool kd_done = false;
    if (ctx->sp_kd && (model_type == kPnP || (ctx->sp_kd >= 2 && !line)) && n > 128 && n < (1ll << 31)) {   // (2: experimental, every model type on this path)
        // segment boundaries of every level (they depend on n alone): a segment of m > 64 points sends the first
        // 64 * ((m / 64) / 2) (at least 64) to the left
        std::vector<std::vector<int>> levels;
        std::vector<int> cur = {0, (int)n};
        for (;;) {
            std::vector<int> nxt;
            bool split = false;
            for (size_t q = 0; q + 1 < cur.size(); ++q) {
                const int a = cur[q], mseg = cur[q + 1] - a;
                nxt.push_back(a);
                if (mseg > 64) {
                    int nl = ((mseg / 64) / 2) * 64;
                    if (nl <= 0) nl = 64;
                    if (nl < mseg) { nxt.push_back(a + nl); split = true; }
                }
            }
            nxt.push_back((int)n);
            levels.push_back(cur);
            if (!split) break;
            cur.swap(nxt);
        }
        KdScale sc;
        {   // one scale for the observed pair, one for the 3-D part: a quarter of what box normalisation would give the latter
            double eo = 0.0, ei = 0.0;
            for (int k = 0; k < d; ++k) {
                const double e = key_f64(st[5 + k]) - key_f64(st[k]);
                if (k < 2) eo = std::fmax(eo, e); else ei = std::fmax(ei, e);
            }
            if (model_type != kPnP) eo = ei = std::fmax(eo, ei) * 4.0 > 0.0 ? std::fmax(eo, ei) : 0.0;   // correspondences: one scale for all four pixel coordinates
            for (int k = 0; k < d; ++k) sc.s[k] = k < 2 ? (eo > 0.0 ? 1.0 / eo : 0.0) : (ei > 0.0 ? (model_type == kPnP ? ctx->sp_kd_weight : 1.0) / ei : 0.0);
            for (int k = d; k < 5; ++k) sc.s[k] = 0.0;
        }
        const size_t maxseg = levels.back().size();
        int nodebits = 1;
        while ((1ll << nodebits) < (long long)maxseg) ++nodebits;
        const int qbits = 32 - nodebits < 16 ? 32 - nodebits : 16;
        if (qbits >= 8) {
            const size_t segbytes = (maxseg * sizeof(int) + 255) & ~(size_t)255, mmbytes = (maxseg * 5 * sizeof(unsigned) + 255) & ~(size_t)255;
            PGX_TRY(ensure(ctx, ctx->weights_scratch, segbytes + 2 * mmbytes + 256));
            int* d_seg = (int*)ctx->weights_scratch.p;
            unsigned* d_mn = (unsigned*)((char*)d_seg + segbytes);
            unsigned* d_mx = (unsigned*)((char*)d_mn + mmbytes);
            hipLaunchKernelGGL(sp_iota_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, v_out, n);
            unsigned* ord_cur = v_out;
            unsigned* ord_nxt = v_in;
            for (size_t lv = 0; lv + 1 < levels.size(); ++lv) {   // (`levels` outlives the copies: the stream is synchronised below)
                const std::vector<int>& seg = levels[lv];
                const int nseg = (int)seg.size() - 1;
                PGX_HIP(ctx, hipMemcpyAsync(d_seg, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
                PGX_HIP(ctx, hipMemsetAsync(d_mn, 0xff, (size_t)nseg * 5 * sizeof(unsigned), ctx->stream));
                PGX_HIP(ctx, hipMemsetAsync(d_mx, 0, (size_t)nseg * 5 * sizeof(unsigned), ctx->stream));
                hipLaunchKernelGGL(sp_kd_extent_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, d, sc, ord_cur, d_seg,
                                   nseg, d_mn, d_mx);
                hipLaunchKernelGGL(sp_kd_keys_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, d, sc, ord_cur, d_seg,
                                   nseg, d_mn, d_mx, k_in, qbits);
                PGX_HIP(ctx, hipGetLastError());
                int nb = 1;
                while ((1ll << nb) < nseg) ++nb;
                PGX_HIP(ctx, rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, ord_cur, ord_nxt, (size_t)n, 0, (unsigned)(qbits + nb), ctx->stream));
                std::swap(ord_cur, ord_nxt);
            }
            PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ord_cur != v_out)
                PGX_HIP(ctx, hipMemcpyAsync(v_out, ord_cur, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream));
            kd_done = true;
        }
    }
    if (!kd_done) {
        hipLaunchKernelGGL(sp_keys_kernel, dim3(blocks), dim3(kSpBlock), 0, ctx->stream, ctx->pts.as<double>(), n, m, k_in, v_in);
        PGX_HIP(ctx, hipGetLastError());
        PGX_HIP(ctx, rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, (unsigned)(m.bits * d), ctx->stream));
    }
    PGX_TRY(ensure(ctx, ctx->pts_s, (size_t)n * d * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pts32_s, (size_t)n * 8 * sizeof(float)));
    PGX_TRY(ensure(ctx, ctx->pmax_s, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->comp_s, (size_t)n * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->pperm, (size_t)n * sizeof(int)));
    PGX_TRY(ensure(ctx, ctx->pts_g, (size_t)padded * d * sizeof(double)));
    PGX_TRY(ensure(ctx, ctx->p32_g, (size_t)padded * 8 * sizeof(float)));
    PGX_TRY(ensure(ctx, ctx->gbounds, (size_t)(groups + supers) * kGroupRow * sizeof(float)));
    hipLaunchKernelGGL(sp_gather_kernel, dim3((unsigned)((padded + kSpBlock - 1) / kSpBlock)), dim3(kSpBlock), 0, ctx->stream,
                       ctx->pts.as<double>(), ctx->pts32.as<float>(), ctx->pmax.as<double>(), v_out, n, d, padded, ctx->pperm.as<int>(),
                       ctx->pts_s.as<double>(), ctx->pts32_s.as<float>(), ctx->pmax_s.as<double>(), ctx->pts_g.as<double>(),
                       ctx->p32_g.as<float>());
    PGX_HIP(ctx, hipGetLastError());
    const int ib0 = model_type == kPnP ? 2 : 0, ib1 = model_type == kPnP ? 4 : 1, ob0 = model_type == kPnP ? 0 : 2;
    if (line) {
        hipLaunchKernelGGL((sp_line_bounds_kernel<64>), dim3((unsigned)groups), dim3(64), 0, ctx->stream, ctx->pts_s.as<double>(), n,
                           ctx->gbounds.as<float>());
        hipLaunchKernelGGL((sp_line_bounds_kernel<64 * kSuper>), dim3((unsigned)supers), dim3(64 * kSuper), 0, ctx->stream,
                           ctx->pts_s.as<double>(), n, ctx->gbounds.as<float>() + groups * kGroupRow);
    } else if (fund) {
        hipLaunchKernelGGL((sp_fund_bounds_kernel<64>), dim3((unsigned)groups), dim3(64), 0, ctx->stream, ctx->pts_s.as<double>(), n,
                           ctx->gbounds.as<float>());
        hipLaunchKernelGGL((sp_fund_bounds_kernel<64 * kSuper>), dim3((unsigned)supers), dim3(64 * kSuper), 0, ctx->stream,
                           ctx->pts_s.as<double>(), n, ctx->gbounds.as<float>() + groups * kGroupRow);
    } else {
        hipLaunchKernelGGL((sp_bounds_kernel<64>), dim3((unsigned)groups), dim3(64), 0, ctx->stream, ctx->pts_s.as<double>(), n, d, ib0, ib1, ob0,
                           ctx->gbounds.as<float>());
        hipLaunchKernelGGL((sp_bounds_kernel<64 * kSuper>), dim3((unsigned)supers), dim3(64 * kSuper), 0, ctx->stream, ctx->pts_s.as<double>(), n, d,
                           ib0, ib1, ob0, ctx->gbounds.as<float>() + groups * kGroupRow);
    }
    PGX_HIP(ctx, hipGetLastError());
    PGX_HIP(ctx, hipMemsetAsync(ctx->comp_s.p, 0, (size_t)n * sizeof(double), ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->point_sort = 1;
    ctx->comp_dirty = 1;
    return PGX_OK;
}

}  // namespace pgx
