// Micro-benchmark: host time of small copies and fills as the C ABI issues them (one stream), microseconds per command.
// build: hipcc --offload-arch=gfx950 -O3 -o small_copy_bench small_copy_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <class F>
double us(int reps, hipStream_t s, F f)
{
    for (int i = 0; i < 50; ++i) f();
    (void)hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    char *d, *pin;
    CK(hipMalloc(&d, 1 << 20));
    CK(hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault));
    std::vector<char> page(1 << 20, 1);
    const int reps = 2000;
    for (size_t bytes : {64ul, 1024ul, 16384ul, 65536ul}) {
        std::printf("%6zu B  H2D pageable %6.1f  H2D pinned %6.1f  D2H pageable %6.1f  D2H pinned %6.1f  D2H pinned + sync %6.1f  fill %6.1f\n", bytes,
                    us(reps, s, [&] { (void)hipMemcpyAsync(d, page.data(), bytes, hipMemcpyHostToDevice, s); }),
                    us(reps, s, [&] { (void)hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s); }),
                    us(reps, s, [&] { (void)hipMemcpyAsync(page.data(), d, bytes, hipMemcpyDeviceToHost, s); }),
                    us(reps, s, [&] { (void)hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s); }),
                    us(reps, s, [&] { (void)hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }),
                    us(reps, s, [&] { (void)hipMemsetAsync(d, 0, bytes, s); }));
    }
    std::printf("kernel launch (empty) %6.1f   launch + sync %6.1f   launch + D2H pinned 64 B + sync %6.1f   launch + D2H pageable 64 B + sync %6.1f\n",
                us(reps, s, [&] { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); }),
                us(reps, s, [&] { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipStreamSynchronize(s); }),
                us(reps, s, [&] { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipMemcpyAsync(pin, d, 64, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }),
                us(reps, s, [&] { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipMemcpyAsync(page.data(), d, 64, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }));
    // a kernel writing straight into pinned host memory instead of a copy
    int* hp;
    CK(hipHostMalloc(&hp, 4096, hipHostMallocDefault));
    std::printf("launch (kernel writes pinned host memory) + sync %6.1f\n",
                us(reps, s, [&] { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, hp); (void)hipStreamSynchronize(s); }));
    return 0;
}
