// sampler_host.hip — Progressive NAPSAC on the in-repo counter-based generator: host code (C++), no kernel.
//
// Replaces: gcransac::sampler::ProgressiveNapsacSampler<4>(&points, {16, 8, 4, 2}, sample_size, {w1, h1, w2, h2}, 0.5), constructed
//           at /root/reference/src/pyprogressivex/src/progressivex_python.cpp:229-238 (sampler id 2; source in the absent
//           graph-cut-ransac submodule, seeded from std::random_device there) - restated after Barath et al., "MAGSAC++ ...
//           Progressive NAPSAC" [UPSTREAM-MEMORY], the statement of pyprogressivex/_proposal.py ProgressiveNapsacSampler.
//
// Why the host: every sample updates the hit counter and the neighbourhood size of its points, and the next sample reads them - the
// draw is one sequential chain (unlike the uniform / NAPSAC / PROSAC samplers of rng.hip.h, which the solver's launch draws per
// lane).  On the device it would be a chain of dependent global accesses per sample; here it is ~0.1 us per sample against the
// ~25 us of the interpreted loop it replaces (a cliff at large N: 10^4 local samples per proposal).  All randomness comes from
// (key, batch, sample number) through Philox4x32-10, so pyprogressivex/_rng.py pnapsac_samples gives the same rows (tests).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "pgx_internal.h"
#include "rng.hip.h"

namespace pgx {

struct PnapsacLayer {
    std::vector<int32_t> cell_of;   // [n] dense cell number of every point
    std::vector<int32_t> start;     // [cells + 1] CSR over `member`
    std::vector<int32_t> member;    // point indices, ascending (= quality order) inside a cell
};

struct PnapsacState {
    int64_t n = 0;
    int m = 0;
    std::vector<PnapsacLayer> layers;
    std::vector<int64_t> hits;
    std::vector<int32_t> subset, layer;
};

}  // namespace pgx

using namespace pgx;

extern "C" {

int pgx_pnapsac_create(const double* pts, int64_t n, int d, const double* sizes, const int32_t* layers, int n_layers, int m, pgx_pnapsac** out)
{
    if (!out) return fail(nullptr, PGX_ERR_INVALID, "pgx_pnapsac_create: out is NULL");
    *out = nullptr;
    if (!pts || !sizes || !layers || n <= 0 || n >= ((int64_t)1 << 31) || d < 1 || n_layers < 1 || n_layers > 16 || m < 2 || m > kMaxSampleSize)
        return fail(nullptr, PGX_ERR_INVALID, "pgx_pnapsac_create: bad argument (n = %lld, d = %d, layers = %d, m = %d)", (long long)n, d, n_layers, m);
    const int dims = d < 4 ? d : 4;   // the grid lives on the first four coordinates (x1, y1, x2, y2)
    PnapsacState* st = new PnapsacState();
    st->n = n;
    st->m = m;
    st->layers.resize((size_t)n_layers);
    std::vector<int64_t> cid((size_t)n);
    std::vector<int32_t> order((size_t)n);
    for (int l = 0; l < n_layers; ++l) {
        const int div = layers[l];
        if (div < 1 || div > 255) { delete st; return fail(nullptr, PGX_ERR_INVALID, "pgx_pnapsac_create: layer %d has %d cells per dimension", l, div); }
        double cell[4];
        for (int k = 0; k < dims; ++k) cell[k] = sizes[k] / (double)div;
        for (int64_t i = 0; i < n; ++i) {
            int64_t id = 0;
            for (int k = 0; k < dims; ++k) {
                double c = std::floor(pts[i * d + k] / cell[k]);
                c = c > 0.0 ? (c > (double)(div - 1) ? (double)(div - 1) : c) : 0.0;   // clip to 0 .. div - 1 (a NaN coordinate lands in cell 0)
                id = id * div + (int64_t)c;
            }
            cid[(size_t)i] = id;
        }
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cid[(size_t)a] < cid[(size_t)b]; });
        PnapsacLayer& L = st->layers[(size_t)l];
        L.cell_of.resize((size_t)n);
        L.member = order;
        L.start.clear();
        int32_t cells = 0;
        for (int64_t q = 0; q < n; ++q) {
            if (q == 0 || cid[(size_t)order[(size_t)q]] != cid[(size_t)order[(size_t)(q - 1)]]) { L.start.push_back((int32_t)q); ++cells; }
            L.cell_of[(size_t)order[(size_t)q]] = cells - 1;
        }
        L.start.push_back((int32_t)n);
    }
    st->hits.resize((size_t)n);
    st->subset.resize((size_t)n);
    st->layer.resize((size_t)n);
    *out = reinterpret_cast<pgx_pnapsac*>(st);
    return PGX_OK;
}

void pgx_pnapsac_destroy(pgx_pnapsac* h)
{
    delete reinterpret_cast<PnapsacState*>(h);
}

int pgx_pnapsac_draw(pgx_pnapsac* h, uint64_t key, uint32_t batch, int32_t count, const int32_t* tops, const int64_t* growth_local,
                     int64_t max_local, int32_t* out)
{
    PnapsacState* st = reinterpret_cast<PnapsacState*>(h);
    if (!st || !tops || !growth_local || !out || count < 0) return fail(nullptr, PGX_ERR_INVALID, "pgx_pnapsac_draw: bad argument");
    const int64_t n = st->n;
    const int m = st->m;
    if (n < m) return fail(nullptr, PGX_ERR_INVALID, "pgx_pnapsac_draw: %lld points for samples of %d", (long long)n, m);
    // a fresh sampler state per draw: ProgressiveX::run resets the sampler before every proposal (progressive_x.h:290)
    std::fill(st->hits.begin(), st->hits.end(), (int64_t)0);
    std::fill(st->subset.begin(), st->subset.end(), (int32_t)m);
    std::fill(st->layer.begin(), st->layer.end(), (int32_t)0);
    const int n_layers = (int)st->layers.size();
    const int64_t n_local = (int64_t)count < max_local ? (int64_t)count : max_local;
    std::vector<int32_t> others;
    for (int64_t k = 0; k < count; ++k) {
        int32_t* row = out + k * m;
        bool global = k >= n_local;
        if (!global) {
            uint32_t w[4];
            philox4x32_10((uint32_t)k, (uint32_t)((uint64_t)k >> 32), batch, 0u, (uint32_t)key, (uint32_t)(key >> 32), w);
            const int64_t p = k < n ? k : (int64_t)(((uint64_t)w[0] * (uint64_t)n) >> 32);
            const int64_t hp = ++st->hits[(size_t)p];
            int64_t sp = st->subset[(size_t)p];
            while (sp < n && hp > growth_local[sp - 1]) ++sp;
            st->subset[(size_t)p] = (int32_t)sp;
            int lay = st->layer[(size_t)p];
            const int32_t* nb = nullptr;
            for (; lay < n_layers; ++lay) {
                const PnapsacLayer& L = st->layers[(size_t)lay];
                const int32_t c = L.cell_of[(size_t)p];
                if ((int64_t)(L.start[(size_t)c + 1] - L.start[(size_t)c]) >= sp) { nb = L.member.data() + L.start[(size_t)c]; break; }
            }
            st->layer[(size_t)p] = (int32_t)lay;
            if (!nb) global = true;
            else {
                others.clear();
                for (int64_t q = 0; q < sp; ++q)
                    if (nb[q] != (int32_t)p) others.push_back(nb[q]);   // the centre is part of its own cell
                const int cnt = (int)others.size();
                if (cnt < m - 1) global = true;
                else {
                    int32_t taken[kMaxSampleSize];
                    for (int j = 0; j < m - 2; ++j) {
                        if (((1 + j) & 3) == 0)
                            philox4x32_10((uint32_t)k, (uint32_t)((uint64_t)k >> 32), batch, (uint32_t)((1 + j) >> 2), (uint32_t)key, (uint32_t)(key >> 32), w);
                        int64_t r = (int64_t)(((uint64_t)w[(1 + j) & 3] * (uint64_t)(cnt - 1 - j)) >> 32);
                        int pos = 0;
                        for (; pos < j && taken[pos] <= r; ++pos) ++r;
                        for (int q = j; q > pos; --q) taken[q] = taken[q - 1];
                        taken[pos] = (int32_t)r;
                        row[j] = others[(size_t)r];
                        st->hits[(size_t)others[(size_t)r]] += 1;
                    }
                    row[m - 2] = others[(size_t)(cnt - 1)];   // "the farthest one" in PROSAC order: always part of the sample
                    st->hits[(size_t)others[(size_t)(cnt - 1)]] += 1;
                    row[m - 1] = (int32_t)p;
                }
            }
        }
        if (global) {   // the global PROSAC sampler, sample number k + 1 (rng.hip.h sample_prosac)
            sample_prosac(key, batch, (uint64_t)k, n, tops[k], m, row);   // (a subset size outside m .. n: the row -1 .. -1)
        }
    }
    return PGX_OK;
}

// ---- row helpers of the numpy-stream samplers (pyprogressivex/_proposal.py _distinct_rows / _fisher_yates_rows) ---------------------
// The random numbers of those samplers come from the caller's numpy generator (its stream is part of every seeded result); what follows
// the draws - which rows hold a repeated index, and the partial Fisher-Yates shuffle of the rows that get an exact draw - was a dozen
// numpy calls on [1 000, m] arrays per proposal (0.3 ms, a tenth of a call on the reference's own scenes).  Same results, bit for bit
// (tests/test_host_logic.py).
int pgx_host_rows_with_duplicates(const int64_t* s, int64_t count, int m, uint8_t* out_bad)
{
    if (!s || !out_bad || count < 0 || m < 0) return PGX_ERR_INVALID;
    for (int64_t r = 0; r < count; ++r) {
        const int64_t* row = s + r * m;
        uint8_t bad = 0;
        for (int a = 1; a < m && !bad; ++a)
            for (int b = 0; b < a; ++b)
                if (row[a] == row[b]) { bad = 1; break; }
        out_bad[r] = bad;
    }
    return PGX_OK;
}

// rows[t] of s [.][m] <- m steps of Fisher-Yates on range(top) with a sparse swap table: step j swaps positions j and j + draws[t][j]
// (draws[t][j] uniform over range(top - j), drawn by the caller); the row is what lands in positions 0 .. m-1.
int pgx_host_fisher_yates_rows(const int64_t* draws, const int64_t* rows, int64_t k, int m, int64_t* s)
{
    if (!draws || !rows || !s || k < 0 || m < 0 || m > 64) return PGX_ERR_INVALID;
    int64_t pos[128], val[128];
    for (int64_t t = 0; t < k; ++t) {
        int used = 0;
        auto get = [&](int64_t p) { for (int i = 0; i < used; ++i) if (pos[i] == p) return val[i]; return p; };
        auto put = [&](int64_t p, int64_t v) { for (int i = 0; i < used; ++i) if (pos[i] == p) { val[i] = v; return; } pos[used] = p; val[used] = v; ++used; };
        int64_t* row = s + rows[t] * m;
        for (int j = 0; j < m; ++j) {
            const int64_t kk = j + draws[t * m + j];
            const int64_t vj = get(j), vk = get(kk);
            put(j, vk);
            put(kk, vj);
            row[j] = vk;
        }
    }
    return PGX_OK;
}

}  // extern "C"
