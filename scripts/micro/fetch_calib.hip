// fetch_calib.hip - known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS repo's access patterns
// (VERDICT r4 item 8).  MI355X_MICROARCH.md: FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced streaming read and
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The score
// path's group kernel loads 8 B per lane from the group-blocked SoA copy of the points (64 consecutive doubles per row per group),
// the cull kernel reads 4 B per lane, and the accumulators are 8-byte atomics scattered over a 48 KB table.
// Every kernel touches `bytes` of a 1 GiB buffer exactly once (well beyond the 256 MiB Infinity Cache) and folds what it read into
// one word so that the loads cannot be dropped.   build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void read16_coalesced(const float4* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *out = acc;
}
__global__ __launch_bounds__(256) void read8_coalesced(const double* __restrict__ p, size_t n, float* out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678) *out = (float)acc;
}
// the group kernel's pattern: one single-wave workgroup per 64-point group, d = 5 rows of 64 doubles each, rows 512 B apart
__global__ __launch_bounds__(64) void read8_group_blocked(const double* __restrict__ p, size_t groups, float* out)
{
    double acc = 0.0;
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const double* row = p + g * 5 * 64;
#pragma unroll
        for (int r = 0; r < 5; ++r) acc += row[r * 64 + threadIdx.x];
    }
    if (acc == 12345.678) *out = (float)acc;
}
__global__ __launch_bounds__(256) void read4_coalesced(const float* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678f) *out = acc;
}
__global__ __launch_bounds__(256) void write8_coalesced(double* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (double)i;
}
// the accumulators' pattern: `count` 8-byte integer atomics spread over a 48 KB table (3 x 2048 words), one per lane
__global__ __launch_bounds__(256) void atomic8_table(unsigned long long* __restrict__ table, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        atomicAdd(&table[(i * 2654435761u) % 6144], 1ull);
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)1 << 30;
    void* buf = nullptr;
    float* out = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const int reps = argc > 1 ? std::atoi(argv[1]) : 3;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(read16_coalesced, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(read8_coalesced, dim3(4096), dim3(256), 0, 0, (const double*)buf, bytes / 8, out);
        hipLaunchKernelGGL(read8_group_blocked, dim3(65536), dim3(64), 0, 0, (const double*)buf, bytes / (5 * 64 * 8), out);
        hipLaunchKernelGGL(read4_coalesced, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
        hipLaunchKernelGGL(write8_coalesced, dim3(4096), dim3(256), 0, 0, (double*)buf, bytes / 8);
        hipLaunchKernelGGL(atomic8_table, dim3(4096), dim3(256), 0, 0, (unsigned long long*)buf, (size_t)1 << 24);
        CK(hipDeviceSynchronize());
    }
    std::printf("known bytes per launch: read16 / read8 / read4 %zu, read8_group_blocked %zu, write8 %zu, atomic8_table %zu atomics x 8 B = %zu\n",
                bytes, (bytes / (5 * 64 * 8)) * 5 * 64 * 8, bytes, (size_t)1 << 24, ((size_t)1 << 24) * 8);
    return 0;
}
