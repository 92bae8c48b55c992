// pnapsac_driver.cpp — drives csrc/sampler_host.hip (Progressive NAPSAC, host code of libpgx.so) under AddressSanitizer +
// UndefinedBehaviorSanitizer on random problems (TEST INFRASTRUCTURE, scripts/sanitize.sh).  The rows themselves are checked
// against the numpy restatement by tests/test_rng.py; this run only has to come back clean.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "../../include/pgx.h"

struct pgx_ctx;
namespace pgx {
int fail(pgx_ctx*, int code, const char* fmt, ...)   // (capi.hip's error sink, not linked here)
{
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
    return code;
}
}  // namespace pgx

int main()
{
    std::mt19937_64 rng(7);
    long long rows = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const int64_t n = 2 + (int64_t)(rng() % (trial % 10 == 0 ? 20000 : 600));
        const int m = 2 + (int)(rng() % 7);
        const int d = 4 + (int)(rng() % 2);
        std::vector<double> pts((size_t)(n * d));
        for (auto& x : pts) x = (double)(rng() % 100000) / 100.0 - (trial % 7 == 0 ? 200.0 : 0.0);
        if (trial % 5 == 0) for (int64_t i = 0; i < n / 2; ++i) for (int k = 0; k < d; ++k) pts[(size_t)(i * d + k)] = pts[(size_t)k];   // duplicates
        const double sizes[4] = {1000.0, 800.0, 1000.0, 800.0};
        const int32_t layers[4] = {16, 8, 4, 2};
        pgx_pnapsac* h = nullptr;
        const int rc = pgx_pnapsac_create(pts.data(), n, d, sizes, layers, 4, m, &h);
        if (rc != 0) return 1;
        const int32_t count = (int32_t)(rng() % 5000);
        std::vector<int32_t> tops((size_t)count + 1);
        for (int32_t k = 0; k < count; ++k) tops[(size_t)k] = (int32_t)(rng() % 8 == 0 ? 0 : (int64_t)m + (int64_t)(rng() % (uint64_t)(n - m + 1 > 0 ? n - m + 1 : 1)));
        std::vector<int64_t> growth((size_t)n);
        int64_t g = 1;
        for (int64_t i = 0; i < n; ++i) { if (i >= m - 1) g += (int64_t)(rng() % 3); growth[(size_t)i] = g; }
        std::vector<int32_t> out((size_t)count * (size_t)m + 1);
        if (n >= m) {
            if (pgx_pnapsac_draw(h, rng(), (uint32_t)trial, count, tops.data(), growth.data(), (int64_t)(n / 2), out.data()) != 0) return 2;
            for (int64_t k = 0; k < (int64_t)count * m; ++k)
                if (out[(size_t)k] < -1 || out[(size_t)k] >= n) { std::fprintf(stderr, "index out of range\n"); return 3; }
            rows += count;
        }
        pgx_pnapsac_destroy(h);
    }
    std::printf("pnapsac sanitizer run: %lld rows drawn, clean\n", rows);
    return 0;
}
