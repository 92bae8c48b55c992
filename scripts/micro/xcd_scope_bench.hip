// Micro-benchmark (round 6): what a dependent memory access costs INSIDE one XCD when nothing forces it past the XCD's L2.
// The level-synchronous min-cut pays ~10 us per sweep / level for a chain of ~6-10 dependent accesses, all agent scope (atomics
// that execute memory-side and drop the line from L2, loads that then miss).  If every participant sits on ONE XCD, the L2 is a
// coherent point for them: atomics WITHOUT sc1 (workgroup scope in the source) execute in that L2 and loads need only bypass
// the CU's L1 (sc1).  Measures: (1) a pointer chase through words last written by plain stores / agent atomics / workgroup
// atomics of ANOTHER CU on the same XCD; (2) ping-pong between two workgroups of one XCD; (3) a barrier over the 32
// workgroups of one XCD in three flavours.   hipcc --offload-arch=gfx950 -O3 xcd_scope_bench.hip -o /tmp/xsb && /tmp/xsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define LOAD_SC1(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// ---- (1) chase: workgroup W (blockIdx 8) writes next[] with flavour `wr`, then workgroup R (blockIdx 0) chases it
__global__ void k_chase(int* next, int n, int wr, int steps, unsigned* flag, unsigned long long* out, int rd)
{
    if (blockIdx.x != 0 && blockIdx.x != 8) return;
    if (blockIdx.x == 8) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int v = (int)(((long long)i * 7919 + 13) % n);
            if (wr == 0) next[i] = v;
            else if (wr == 1) __hip_atomic_exchange(&next[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (wr == 2) __hip_atomic_exchange(&next[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else { __hip_atomic_store(&next[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&next[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[3] = xcc_id();
        }
        return;
    }
    if (threadIdx.x != 0) return;
    { unsigned spins = 0; while (LOAD_SC1(flag) == 0) { __builtin_amdgcn_s_sleep(1); if (++spins > 20000000u) break; } }
    int p = 0;
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        if (rd == 0) p = LOAD_SC1(&next[p]);
        else if (rd == 1) p = __hip_atomic_fetch_add(&next[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else p = __hip_atomic_fetch_add(&next[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[0] = wall_clock64() - t0;
    out[1] = (unsigned long long)p;
    out[2] = xcc_id();
}

// ---- (2) ping-pong between blockIdx 0 and blockIdx `other` (8: same XCD, 1: the next XCD)
__global__ void k_pingpong(unsigned* word, int rounds, int flavour, int other, unsigned long long* out)
{
    if ((int)blockIdx.x != 0 && (int)blockIdx.x != other) return;
    if (threadIdx.x != 0) return;
    const unsigned me = blockIdx.x == 0 ? 0u : 1u;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        const unsigned want = 2u * r + me;
        unsigned spins = 0;
        while (LOAD_SC1(word) != want) { if (++spins > 20000000u) { out[1] = 1; return; } }     // (never hang the box: report instead)
        if (flavour == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (flavour == 1) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else *(volatile unsigned*)word = want + 1u;     // plain store (write-through L1, stays in L2)
    }
    if (me == 0) out[0] = wall_clock64() - t0;
}

// ---- (3) barrier over the workgroups with (blockIdx & 7) == 0, with a dependent write + read of a neighbour's slice per round
__device__ __forceinline__ void xcd_barrier(unsigned* ctr, unsigned target, int flavour)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        if (flavour == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > 20000000u) break; }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, flavour == 1 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned spins = 0;
            while (LOAD_SC1(ctr) < target) { if (++spins > 20000000u) break; }
        }
    }
    __syncthreads();
}
__global__ void k_xbar(unsigned* ctr, int rounds, int flavour, int* data, int nwrite, unsigned long long* out, int wgs)
{
    if (blockIdx.x & 7u) return;
    const unsigned me = blockIdx.x >> 3;
    if ((int)me >= wgs) return;
    const unsigned part = (unsigned)wgs;
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x) {
            if (flavour == 0) __hip_atomic_store(&data[me * nwrite + i], r + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else data[me * nwrite + i] = r + i;
        }
        xcd_barrier(ctr, (unsigned)(r + 1) * part, flavour);
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x) acc += LOAD_SC1(&data[((me + 1) % part) * nwrite + i]) - (r + i);
    }
    if (threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)(acc != 0));
    if (threadIdx.x == 0 && me == 0) out[0] = wall_clock64() - t0;
}

// ---- (4) all 8 XCDs, no fences: per-XCD arrival counter -> the XCD's last arriver bumps a top counter -> the top's last arriver
// bumps a generation word every workgroup polls; data exchanged with agent-scope (sc1) stores and loads only
struct HBar { unsigned xc[8 * 16]; unsigned top[16]; unsigned gen[16]; };
__device__ __forceinline__ void hier_barrier(HBar* b, unsigned round, unsigned per_xcd, unsigned xcds, unsigned xcc)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = __hip_atomic_fetch_add(&b->xc[xcc * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == (round + 1) * per_xcd) {
            const unsigned t = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (round + 1) * xcds) __hip_atomic_store(&b->gen[0], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (LOAD_SC1(&b->gen[0]) < round + 1) { if (++spins > 20000000u) break; }
    }
    __syncthreads();
}
__global__ void k_hbar(HBar* b, int rounds, int* data, int nwrite, unsigned long long* out, int flat)
{
    const unsigned xcc = xcc_id() & 7u, me = blockIdx.x, part = gridDim.x;
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x) __hip_atomic_store(&data[me * nwrite + i], r + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flat) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (LOAD_SC1(&b->top[0]) < (unsigned)(r + 1) * part) { if (++spins > 20000000u) break; }
            }
            __syncthreads();
        } else hier_barrier(b, (unsigned)r, part / 8, 8, xcc);
        for (int i = threadIdx.x; i < nwrite; i += blockDim.x) acc += LOAD_SC1(&data[((me + 9) % part) * nwrite + i]) - (r + i);   // a workgroup of another XCD
    }
    if (threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)(acc != 0));
    if (threadIdx.x == 0 && me == 0) out[0] = wall_clock64() - t0;
}

int main()
{
    int* next; unsigned* flag; unsigned long long* out; int* data;
    const int n = 1 << 16;
    hipMalloc(&next, n * 4); hipMalloc(&flag, 64); hipMalloc(&out, 64); hipMalloc(&data, 64 * 4096 * 4);
    const double tick_ns = 10.0;   // wall_clock64: 100 MHz
    const char* wrn[] = {"plain stores", "agent-scope atomics", "workgroup-scope atomic exchange", "workgroup-scope atomic add"};
    const char* rdn[] = {"sc1 load", "workgroup-scope atomic (returning)", "agent-scope atomic (returning)"};
    for (int wr = 0; wr < 4; ++wr)
        for (int rd = 0; rd < 3; ++rd) {
            unsigned long long h[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(flag, 0, 64); hipMemset(next, 0, n * 4);
                hipLaunchKernelGGL(k_chase, dim3(16), dim3(256), 0, 0, next, n, wr, 2000, flag, out, rd);
                hipDeviceSynchronize();
                hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
            }
            printf("chase  written by %-34s read by %-36s: %7.1f ns per dependent access  (xcc reader %llu writer %llu, end %llu)\n", wrn[wr], rdn[rd],
                   h[0] * tick_ns / 2000, h[2], h[3], h[1]);
        }
    const char* fl[] = {"agent-scope atomic add", "workgroup-scope atomic add", "plain store"};
    for (int other : {8, 1})
        for (int f = 0; f < (other == 8 ? 3 : 1); ++f) {     // (across XCDs only the agent-scope form is visible at all)
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(flag, 0, 64);
                hipMemset(out, 0, 64);
                hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(64), 0, 0, flag, 2000, f, other, out);
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
                hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            }
            printf("ping-pong with block %d (%s XCD), %-28s: %7.1f ns per hand-off%s\n", other, other == 8 ? "same" : "next", fl[f], h[0] * tick_ns / 4000,
                   h[1] ? "  (GAVE UP: never became visible)" : "");
        }
    const char* bn[] = {"release/acquire agent (round 4's form)", "relaxed agent atomic + sc1 poll, plain data", "workgroup-scope atomic + sc1 poll, plain data"};
    for (int wgs : {32, 8, 2})
        for (int nwrite : {0, 1024})
            for (int f = 0; f < 3; ++f) {
                unsigned long long h[2] = {0, 0};
                for (int rep = 0; rep < 2; ++rep) {
                    hipMemset(flag, 0, 64); hipMemset(out, 0, 64);
                    hipLaunchKernelGGL(k_xbar, dim3(256), dim3(1024), 0, 0, flag, 2000, f, data, nwrite, out, wgs);
                    hipError_t e = hipDeviceSynchronize();
                    if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
                    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
                }
                printf("barrier %2d workgroups of one XCD, %4d ints exchanged, %-46s: %6.2f us per round (stale reads in %llu workgroups)\n", wgs, nwrite, bn[f],
                       h[0] * tick_ns / 2000 / 1000, h[1]);
            }
    HBar* hb; hipMalloc(&hb, sizeof(HBar));
    for (int block : {256, 1024})
        for (int nwrite : {0, 1024})
            for (int flat : {1, 0}) {
                unsigned long long h[2] = {0, 0};
                for (int rep = 0; rep < 2; ++rep) {
                    hipMemset(hb, 0, sizeof(HBar)); hipMemset(out, 0, 64);
                    hipLaunchKernelGGL(k_hbar, dim3(256), dim3(block), 0, 0, hb, 2000, data, nwrite, out, flat);
                    hipError_t e = hipDeviceSynchronize();
                    if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
                    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
                }
                printf("barrier 256 workgroups x %4d threads on all 8 XCDs, no fences, sc1 data, %4d ints exchanged, %-12s: %6.2f us per round (stale reads in %llu workgroups)\n", block, nwrite,
                       flat ? "one counter" : "hierarchical", h[0] * tick_ns / 2000 / 1000, h[1]);
            }
    return 0;
}
