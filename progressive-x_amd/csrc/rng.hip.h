// rng.hip.h — the in-repo counter-based random number generator and the uniform minimal-sample sampler built on it.
//
// Replaces: gcransac::sampler::UniformSampler (constructed at /root/reference/src/pyprogressivex/src/progressivex_python.cpp:
//           215-245 and :121 for find6DPoses; its source is in the absent graph-cut-ransac submodule and seeds itself from
//           std::random_device, so the reference is not reproducible run to run - SURVEY.md §7 step 0 / §8(f1) ask for a
//           counter-based generator that is THE SAME in Python and on the device instead).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): four 32-bit words from a
// 128-bit counter and a 64-bit key, no state - sample s of batch b is a pure function of (key, b, s), so a lane generates its
// own sample without any ordering between lanes.  The same arithmetic in pyprogressivex/_rng.py (numpy) and in the tests'
// CPU checker (C); the published known-answer vectors are in tests/test_rng.py.
//
// A sample = m DISTINCT indices of range(n), every ordered m-tuple equally likely (up to the 2^-32 granularity of the
// multiply-shift range reduction): position j draws r in [0, n - j) and takes the r-th index not taken yet (the taken ones are
// kept sorted: r is stepped past every taken index <= r).  No rejection loop: exactly m words per sample.
#pragma once
#include <cstdint>

namespace pgx {

constexpr int kMaxSampleSize = 8;

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sample `s` of batch `batch` under `key`: m distinct indices of range(n) into out[0 .. m)   (1 <= m <= kMaxSampleSize <= n)
__host__ __device__ inline void sample_distinct(uint64_t key, uint32_t batch, uint64_t s, int64_t n, int m, int32_t* out)
{
    int32_t taken[kMaxSampleSize];   // ascending
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < m; ++j) {
        if ((j & 3) == 0) philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), batch, (uint32_t)(j >> 2), (uint32_t)key, (uint32_t)(key >> 32), w);
        int64_t r = (int64_t)(((uint64_t)w[j & 3] * (uint64_t)(n - j)) >> 32);   // uniform over range(n - j)
        int pos = 0;
        for (; pos < j && taken[pos] <= r; ++pos) ++r;                           // the r-th index not taken yet
        for (int q = j; q > pos; --q) taken[q] = taken[q - 1];
        taken[pos] = (int32_t)r;
        out[j] = (int32_t)r;
    }
}

// NAPSAC on the same generator (gcransac::sampler::NapsacSampler, progressivex_python.cpp:241 sampler id 3; absent upstream): word 0
// draws the centre uniformly, words 1 .. m-1 draw m - 1 DISTINCT entries of the centre's neighbour list (the resident graph's CSR
// row), by the same "r-th entry not taken yet" rule.  A centre with fewer than m - 1 neighbours gives no sample: the row is all -1,
// the solvers turn it into a NaN model, the RANSAC iteration is spent (what the reference's failed sample costs).
__host__ __device__ inline void sample_napsac(uint64_t key, uint32_t batch, uint64_t s, int64_t n, const int* off, const int* idx, int m, int32_t* out)
{
    int32_t taken[kMaxSampleSize];
    uint32_t w[4];
    philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), batch, 0u, (uint32_t)key, (uint32_t)(key >> 32), w);
    const int64_t c = (int64_t)(((uint64_t)w[0] * (uint64_t)n) >> 32);
    const int a0 = off[c], deg = off[c + 1] - a0;
    if (deg < m - 1) {
        for (int j = 0; j < m; ++j) out[j] = -1;
        return;
    }
    out[0] = (int32_t)c;
    for (int j = 1; j < m; ++j) {
        if ((j & 3) == 0) philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), batch, (uint32_t)(j >> 2), (uint32_t)key, (uint32_t)(key >> 32), w);
        int64_t r = (int64_t)(((uint64_t)w[j & 3] * (uint64_t)(deg - (j - 1))) >> 32);
        int pos = 0;
        for (; pos < j - 1 && taken[pos] <= r; ++pos) ++r;
        for (int q = j - 1; q > pos; --q) taken[q] = taken[q - 1];
        taken[pos] = (int32_t)r;
        out[j] = idx[a0 + r];
    }
}

// PROSAC on the same generator (gcransac::sampler::ProsacSampler, progressivex_python.cpp:222 sampler id 1; absent upstream, the
// USAC formulation): the points are ordered by quality; sample number k draws from the best `top` = n_k points - m - 1 DISTINCT
// ones of the first top - 1 and point top - 1 itself.  n_k comes from Chum & Matas' growth function, a sequential floating-point
// recurrence the host tabulates once per sampler (the sample numbers restart at 1 with every proposal, progressive_x.h:290, so one
// table of max-iterations entries serves every batch).  top == 0: past the convergence bound the sampler is uniform over all n;
// top < m (never produced by the table): no sample, row -1.
__host__ __device__ inline void sample_prosac(uint64_t key, uint32_t batch, uint64_t s, int64_t n, int top, int m, int32_t* out)
{
    if (top == 0) {
        sample_distinct(key, batch, s, n, m, out);
        return;
    }
    if (top < m || top > n) {
        for (int j = 0; j < m; ++j) out[j] = -1;
        return;
    }
    if (m > 1) sample_distinct(key, batch, s, top - 1, m - 1, out);
    out[m - 1] = top - 1;
}

}  // namespace pgx
