// Dispatch-rate probe: how long does a grid of trivial workgroups take (per workgroup size / LDS size)?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void k_empty(int* out) {
    __shared__ int s[LDS / 4];
    if (threadIdx.x == 0) s[0] = blockIdx.x;
    __syncthreads();
    if (out && s[0] < 0) out[0] = 1;
}
template <int LDS>
void run(const char* name, int blocks, int threads) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(blocks), dim3(threads), 0, 0, (int*)nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(blocks), dim3(threads), 0, 0, (int*)nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s blocks=%d threads=%d lds=%d: %.1f us per launch, %.2f ns per workgroup\n", name, blocks, threads, LDS, ms * 50, ms * 50e3 / blocks);
}
int main() {
    run<64>("tiny", 125000, 64); run<64>("tiny", 250000, 64); run<64>("tiny", 62500, 256); run<64>("tiny", 15625, 256);
    run<5632>("lds5k", 125000, 64); run<22528>("lds22k", 62500, 256); run<64>("tiny", 500000, 64);
    return 0;
}
