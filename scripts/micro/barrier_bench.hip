// Micro-benchmark (round 4): latency of a hand-written grid barrier on MI355X - the number that decides whether the
// level-synchronous max-flow schedule (one launch per BFS level / sweep, ~13 us each at N = 1e6) can move into one
// persistent kernel.  hipcc --offload-arch=gfx950 -O3 barrier_bench.hip -o barrier_bench && ./barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// mode 0: all workgroups take part; mode 1: only workgroups with (blockIdx.x & 7) == 0 (one XCD)
__global__ void k_barrier(unsigned* ctr, int rounds, int mode, int* data, int nwrite, unsigned long long* out)
{
    unsigned part = gridDim.x, me = blockIdx.x;
    if (mode == 1) {
        if (blockIdx.x & 7u) return;
        part = (gridDim.x + 7u) / 8u;
        me = blockIdx.x >> 3;
    }
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        // a little dependent global traffic per round, as a BFS level would have: write own slice, read a neighbour's
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x) data[me * nwrite + i] = r + i;
        grid_barrier(ctr, (unsigned)(r + 1) * part);
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x)
            acc += __hip_atomic_load(&data[((me + 1) % part) * nwrite + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (r + i);
    }
    if (threadIdx.x == 0 && me == 0) { out[0] = wall_clock64() - t0; out[1] = (unsigned long long)acc; }
}

int main()
{
    unsigned* ctr; int* data; unsigned long long* out;
    hipMalloc(&ctr, 4); hipMalloc(&data, 4096 * 1024 * 4); hipMalloc(&out, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 2000;
    struct Cfg { int grid, block, mode, nwrite; };
    std::vector<Cfg> cfgs = {{256, 256, 0, 0}, {256, 256, 0, 256}, {256, 1024, 0, 0}, {256, 1024, 0, 1024}, {512, 256, 0, 0}, {512, 512, 0, 512},
                             {1024, 256, 0, 0}, {1024, 256, 0, 256},
                             {256, 256, 1, 0}, {256, 1024, 1, 0}, {256, 1024, 1, 1024}, {512, 1024, 1, 1024}, {64, 1024, 0, 1024}, {128, 1024, 0, 1024}};
    for (auto c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(ctr, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_barrier, dim3(c.grid), dim3(c.block), 0, 0, ctr, rounds, c.mode, data, c.nwrite, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            if (rep) printf("grid %4d block %4d mode %d nwrite %4d: %.2f us per barrier round (acc %llu, err %s)\n", c.grid, c.block, c.mode, c.nwrite,
                            ms * 1000.0 / rounds, h[1], hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
