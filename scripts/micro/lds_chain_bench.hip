// Micro-benchmark: what a dependent LDS round trip, an LDS atomic with return and a workgroup barrier cost inside ONE workgroup on an
// otherwise idle GPU (the situation of the one-workgroup min-cut kernels), in nanoseconds and in shader clocks.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_chain_bench lds_chain_bench.hip ; run: ./lds_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NT>
__global__ __launch_bounds__(NT) void k(unsigned long long* out, int iters)
{
    __shared__ int chain[4096];
    __shared__ int word;
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += NT) chain[i] = (i * 97 + 13) & 4095;
    if (tid == 0) word = 0;
    __syncthreads();
    unsigned long long w0 = wall_clock64(), c0 = clock64();
    int p = tid;
    for (int i = 0; i < iters; ++i) p = chain[p];
    unsigned long long w1 = wall_clock64(), c1 = clock64();
    int q = p;
    for (int i = 0; i < iters; ++i) q = atomicAdd(&chain[q & 4095], 0) ;
    unsigned long long w2 = wall_clock64(), c2 = clock64();
    for (int i = 0; i < iters; ++i) __syncthreads();
    unsigned long long w3 = wall_clock64(), c3 = clock64();
    int r = q;
    for (int i = 0; i < iters; ++i) { atomicOr(&word, r & 1); __syncthreads(); r += word; }
    unsigned long long w4 = wall_clock64(), c4 = clock64();
    if (tid == 0) {
        out[0] = w1 - w0; out[1] = c1 - c0; out[2] = w2 - w1; out[3] = c2 - c1; out[4] = w3 - w2; out[5] = c3 - c2; out[6] = w4 - w3; out[7] = c4 - c3;
        out[8] = (unsigned long long)(p + q + r);
    }
}

template <int NT>
void run(unsigned long long* d, int iters)
{
    unsigned long long h[9];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<NT>, dim3(1), dim3(NT), 0, 0, d, iters);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    const char* names[4] = {"dependent LDS read", "LDS atomic with return", "workgroup barrier", "atomicOr + barrier + read (a vote)"};
    for (int j = 0; j < 4; ++j)
        std::printf("threads %4d  %-36s %7.1f ns  %7.1f clocks  (%.0f MHz)\n", NT, names[j], 10.0 * h[2 * j] / iters, (double)h[2 * j + 1] / iters,
                    h[2 * j] ? 100.0 * h[2 * j + 1] / h[2 * j] : 0.0);
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 9 * 8);
    const int iters = 2000;
    run<64>(d, iters);
    run<256>(d, iters);
    run<512>(d, iters);
    run<1024>(d, iters);
    // the same right after a long idle gap (clock ramp)
    hipDeviceSynchronize();
    return 0;
}
