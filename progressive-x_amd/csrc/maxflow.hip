// maxflow.hip — HIP backend of the alpha-expansion move: kernels wrapping the per-site bodies of maxflow_body.hip.h and
// the launch plumbing for maxflow_driver.inl.
//
// Replaces: GCoptimizationGeneralGraph::alpha_expansion + BK max-flow behind pearl::PEARL::labeling
//           (/root/reference/src/pyprogressivex/include/PEARL.h:507-551); upstream source absent [U-5].
//
// Layout: one lane per site, CSR neighbour rows (off/idx/mult/rev, symmetric), residual capacities per directed arc,
// per-site int64 excess / sink residual / hub flows, int32 heights.  The kernels are latency/HBM bound irregular
// gathers; cross-workgroup communication is through device-scope atomics only, and every kernel
// boundary is a full synchronisation point, so no in-launch release/acquire protocol is needed.
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "maxflow_driver.inl"
#include "maxflow_l0.hip.h"
#include "pgx_internal.h"

namespace pgx {

constexpr int kMfBlock = 256;

struct MaxflowState {
    DevBuf cap, tot, ex, rt, d, f, g, small, front;
    int* h_flags = nullptr;  // pinned host mirror for flag read-backs
    int* h_pub = nullptr;    // device-mapped host words the last kernel before a read-back publishes the flags into (mf_publish):
                             // [0 .. kMfFlags) flags, [kMfFlags] cnt[alpha], [15] sequence number the host spins on
    int pub_seq = 0;
    DevBuf lists;            // act[2][n] | mark[n]
    DevBuf bar;              // word 8: arrival ticket of the kernels that publish the flags to the host (zeroed once)
    int next_stamp = 1;
    int64_t mark_n = 0;
    DevBuf xcd;              // XcdCtl of the persistent one-XCD rounds (maxflow_xcd.hip.h)
    int* h_xcd = nullptr;    // pinned host copy of its result words
    bool xcd_broken = false; // a persistent launch came back without participants (its workgroups never became co-resident, e.g. a partitioned GPU):
                             // never tried again on this context - its registration waits seconds before it gives up
    int64_t xcd_launches = 0, xcd_rounds = 0, xcd_done = 0, xcd_swept16 = 0, xcd_searches = 0, xcd_levels = 0, xcd_sweeps = 0;
    int bfs_hint[2] = {0, 0};  // last labelled level of the previous first / later search of a move (MfTuning::bfs_hint)
};

constexpr int kMfMaxLabels = 64;

// What a workgroup reports when it retires - a minimum per label, a "something happened" flag - lands on a handful of
// addresses, and same-address atomics serialise at ~20 ns each: 512 workgroups reporting the same hub height cost 10 us at
// the tail of a sweep over all sites.  Nearly all of them report what is already there: look first (a load is not serialised).
__device__ __forceinline__ void mf_report_min(int* p, int val)
{
    if (val < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(p, val);
}
__device__ __forceinline__ void mf_report_flag(int* p, int val)
{
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != val) __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kMfBlock) void mf_k_count(MfView v, int, int)
{
    // per-block LDS histogram, one global atomic per (block, label present); grid-stride: same-address atomics cost ~20 ns
    // each, serialised, so the number of workgroups is capped by the launcher (kAggBlocks)
    __shared__ int hist[kMfMaxLabels];
    if (threadIdx.x < kMfMaxLabels) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x; u < v.n; u += (int64_t)gridDim.x * kMfBlock)
        atomicAdd(&hist[v.labels[u]], 1);
    __syncthreads();
    if ((int)threadIdx.x < v.L && hist[threadIdx.x] > 0) atomicAdd(&v.cnt[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(kMfBlock) void mf_k_init(MfView v, int, int)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    if (u < v.n) mf_body_init_site(v, u);
}

// Workgroup-local staging of BFS appends (see mf_bfs_label): an LDS list, moved to the level's segment of `order` with one
// atomic on the level counter per flush.  All threads of the workgroup call stage_flush.
constexpr int kStageCap = 4096;

struct Stage {
    int list[kStageCap];
    int count;
    int base;
};

__device__ __forceinline__ void stage_flush(const MfView& v, Stage& st, int k, int level_base = -1)
{
    __syncthreads();
    const int n = st.count;
    if (n > 0) {
        if (threadIdx.x == 0) st.base = (level_base >= 0 ? level_base : mf_level_base(v, k)) + atomicAdd(&v.fcount[k % 3], n);
        __syncthreads();
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x)
            __hip_atomic_store(&v.order[st.base + i], st.list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) st.count = 0;
    }
    __syncthreads();
}

// Kernels whose per-site body reports a boolean and/or accumulates per-label minima: both are aggregated per block in
// LDS and flushed with at most (L + 1) global operations per block.
enum { kCountActive = 2, kApply = 4 };

// A read-back used to be a 48-byte copy command plus hipStreamSynchronize behind the kernel that finished the flags (~9 us on
// top of that kernel, 12 000 times per findVanishingPoints call).  Instead that kernel's last wave stores the flags straight
// into device-mapped host memory, fences, and stores a sequence number the host spins on: ~3 us.  Called by one whole wave.
__device__ __forceinline__ void mf_publish(const MfView& v, int* pub, int seq)
{
    const int lane = (int)(threadIdx.x & 63);
    if (lane < kMfFlags) __hip_atomic_store(&pub[lane], __hip_atomic_load(&v.flags[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (lane == kMfFlags) __hip_atomic_store(&pub[lane], __hip_atomic_load(&v.cnt[v.alpha], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&pub[15], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int WHAT>
__global__ __launch_bounds__(kMfBlock) void mf_k_agg(MfView v, int a0, int a1, int* pub, int* ticket, int seq)
{
    __shared__ int s_min[kMfMaxLabels];
    if (threadIdx.x < kMfMaxLabels) s_min[threadIdx.x] = kMfInf;
    __syncthreads();
    int mine = 0;   // grid-stride (kAggBlocks workgroups): one set of global atomics per workgroup, not per 256 sites
    for (int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x; u < v.n; u += (int64_t)gridDim.x * kMfBlock) {
        if (WHAT == kCountActive) mine += mf_body_count_active(v, u) ? 1 : 0;
        else if (WHAT == kApply) mine += mf_body_apply(v, u) ? 1 : 0;
    }
    __shared__ int s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    if (mine > 0) atomicAdd(&s_count, mine);
    __syncthreads();
    const int count = s_count;
    if (WHAT == kCountActive) {
        if (threadIdx.x == 0 && count > 0) {
            mf_report_flag(&v.flags[1], 1);
            atomicAdd(&v.flags[3], count);  // number of active sites after this global relabel (diagnostics)
        }
    } else {
        if (threadIdx.x == 0 && count > 0) atomicAdd(&v.flags[2], count);
    }
    if (pub != nullptr) {   // the last workgroup to get here publishes the finished flags to the host (workgroup-uniform branch)
        __shared__ int s_last;
        if (threadIdx.x == 0) {
            __threadfence();
            const int t = atomicAdd(ticket, 1);
            s_last = t == (int)gridDim.x - 1;
            if (s_last) atomicExch(ticket, 0);
        }
        __syncthreads();
        if (s_last && threadIdx.x < 64) mf_publish(v, pub, seq);
    }
}

// One push-relabel step of a workgroup's sites: (1) hub pull requests, summed per hub in LDS (slot kMfMaxLabels is the
// alpha hub); (2) one reservation per (workgroup, hub) on the hub's excess word; (3) the grant is split in LDS arrival
// order and every site runs its discharge step.  `u` < 0 marks an idle lane.  All threads of the workgroup call it.
struct SweepLds {
    int min[kMfMaxLabels];
    unsigned long long want[kMfMaxLabels + 1];
    long long got[kMfMaxLabels + 1];
    unsigned long long pushA;
    int moved;
};

__device__ __forceinline__ void sweep_lds_init(SweepLds& s)
{
    if (threadIdx.x < kMfMaxLabels) s.min[threadIdx.x] = kMfInf;
    if (threadIdx.x <= kMfMaxLabels) { s.want[threadIdx.x] = 0; s.got[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { s.pushA = 0; s.moved = 0; }
    __syncthreads();
}

__device__ __forceinline__ bool mf_sweep_step(const MfView& v, int64_t u, int prev, int cur, bool list_mode, SweepLds& s, int* pushed_to,
                                              bool* listed = nullptr)
{
    MfSweepIo io;
    io.list_mode = list_mode;
    long long want = 0, before = 0;
    int slot = 0;
    // Work-list rounds start with no beta hub able to deliver (the driver's precondition) and end when one comes back into
    // play (flags[6]), so without an alpha hub nobody can pull: the pull phase - three dependent gathers and two workgroup
    // barriers per pass - is skipped (workgroup-uniform condition).
    const bool pulls = !list_mode || v.has_alpha_hub[0] != 0;
    if (pulls) {
        if (u >= 0) {
            want = mf_body_pull_want(v, u, prev, &io.which);
            if (io.which != 0) {
                slot = io.which == 1 ? v.labels[u] : kMfMaxLabels;
                before = (long long)atomicAdd(&s.want[slot], (unsigned long long)want);
            }
        }
        __syncthreads();
        if ((int)threadIdx.x <= kMfMaxLabels && s.want[threadIdx.x] > 0) {
            const long long w = (long long)s.want[threadIdx.x];
            if ((int)threadIdx.x == kMfMaxLabels) {
                s.got[threadIdx.x] = mf_reserve(v.hubA_e, w);
                atomicAdd((unsigned long long*)&v.hubA_want[cur], (unsigned long long)w);
            } else {
                s.got[threadIdx.x] = mf_reserve(&v.hub_e[threadIdx.x], w);
            }
            s.want[threadIdx.x] = 0;
        }
        __syncthreads();
    }
    bool r = false;
    if (u >= 0) {
        if (io.which != 0) {
            const long long g = s.got[slot] - before;
            io.granted = g < 0 ? 0 : (g > want ? want : g);
        }
        r = mf_body_sweep(v, u, prev, cur, s.min, &io);
        if (io.pushedA > 0) atomicAdd(&s.pushA, (unsigned long long)io.pushedA);
        if (io.moved) s.moved = 1;
        if (pushed_to) *pushed_to = io.pushed_to;
        if (listed) *listed = io.listed;
    }
    return r;
}

__device__ __forceinline__ void mf_sweep_flush(const MfView& v, int cur, bool list_mode, SweepLds& s, bool any)
{
    const int count = __syncthreads_count(any ? 1 : 0);
    if (!list_mode && (int)threadIdx.x < v.L && s.min[threadIdx.x] != kMfInf)
        mf_report_min(&v.hub_min[cur * v.L + threadIdx.x], s.min[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (s.pushA > 0) atomicAdd((unsigned long long*)v.hubA_e, s.pushA);
        if (count > 0) mf_report_flag(&v.flags[1], 1);
        if (s.moved) mf_report_flag(&v.flags[8], 1);
    }
}

constexpr int kSweepList = 8 * kMfBlock;   // live sites a workgroup collects before it runs the step over them

// sweep over all sites
__global__ __launch_bounds__(kMfBlock) void mf_k_sweep(MfView v, int prev, int cur)
{
    if (mf_sweep_idle(v)) return;
    __shared__ SweepLds s;
    sweep_lds_init(s);
    // grid-stride over 256-site chunks: a workgroup ends with up to L atomicMin on the hub heights (~20 ns each, serialised
    // per address) - with one workgroup per chunk that was 3 906 of them per label at N = 1e6, most of the 70 us a sweep
    // over all sites took there (12 k such sweeps per find6DPoses call)
    // A site takes part only if it holds excess or its hub is in play (holds excess: members pull and report their heights);
    // everything else returns from the per-site body without side effects - after three dependent gathers and two workgroup
    // barriers.  Here label and excess are loaded up front (two independent coalesced loads), the "hub in play" flags sit in
    // LDS, and a chunk without a live site costs one barrier.  (The gates are the same plain, possibly stale reads the body
    // makes; a push arriving later in this sweep is picked up by the next one, as before.)
    __shared__ int s_hub_live[kMfMaxLabels];
    if (threadIdx.x < kMfMaxLabels)
        s_hub_live[threadIdx.x] = (int)threadIdx.x < v.L && v.hub_exists[threadIdx.x] && v.hub_e[threadIdx.x] > 0;
    const bool hub_a = v.has_alpha_hub[0] != 0;
    __syncthreads();
    // Two phases per workgroup: its chunks are scanned back to back (no barrier in between: the loads pipeline) and the live
    // sites collected in LDS; then the push-relabel step runs over that compact list, 256 sites per pass.  In-kernel timers on
    // a find6DPoses call: 23 of 256 sites of a chunk are live, and a pass over a chunk with ONE live site costs the same 4 us
    // (two barriers, the gather chain) as a full one - 1.8 to 8 such passes per workgroup became one.
    __shared__ int s_list[kSweepList];
    __shared__ int s_nlist;
    if (threadIdx.x == 0) s_nlist = 0;
    __syncthreads();
    bool r = false;
    auto run_list = [&]() {
        __syncthreads();
        const int nl = s_nlist;
        for (int i = (int)threadIdx.x; i < (nl + kMfBlock - 1) / kMfBlock * kMfBlock; i += kMfBlock)
            r |= mf_sweep_step(v, i < nl ? s_list[i] : -1, prev, cur, false, s, nullptr);
        __syncthreads();
        if (threadIdx.x == 0) s_nlist = 0;
        __syncthreads();
    };
    const int64_t chunks = (v.n + kMfBlock - 1) / kMfBlock;
    int pending = 0;   // chunks scanned since the list was last emptied (workgroup-uniform)
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const int64_t u = c * kMfBlock + threadIdx.x;
        if (u < v.n) {
            const int lu = v.labels[u];
            const long long e = v.ex[u];
            if (lu != v.alpha && (hub_a || s_hub_live[lu] || e > 0)) s_list[atomicAdd(&s_nlist, 1)] = (int)u;
        }
        if (++pending == kSweepList / kMfBlock) { run_list(); pending = 0; }   // the list cannot overflow
    }
    if (pending > 0) run_list();
    mf_sweep_flush(v, cur, false, s, r);
}

// Two conditional appends per lane (the site itself, the site it pushed to) with ONE reservation per wave: the list
// maintenance of a work-list sweep - membership test, two claims, two wave-aggregated appends - took as long as the
// push-relabel step itself (6.3 vs 7.6 us per pass, wall_clock64 in the kernel).
__device__ __forceinline__ void mf_append2(int* counter, int* list, int a, bool want_a, int b, bool want_b)
{
    const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
    const int na = __popcll(ma), nb = __popcll(mb);
    if (na + nb == 0) return;
    const int lane = __lane_id();
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, na + nb);
    base = __shfl(base, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (want_a) list[base + __popcll(ma & below)] = a;
    if (want_b) list[base + na + __popcll(mb & below)] = b;
}

// ---- list-mode sweeps (maxflow_driver.inl): only the sites that can act are visited ------------------------------------
__global__ __launch_bounds__(kMfBlock) void mf_k_build_list(MfView v, int stamp, int)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    const bool want = u < v.n && mf_listed(v, u);
    if (want) v.mark[u] = stamp;
    mf_list_append(v, 0, (int)u, want);
}

__global__ __launch_bounds__(kMfBlock) void mf_k_sweep_list(MfView v, int prev, int cur, int parity, int stamp)
{
    if (mf_sweep_idle(v)) return;
    const int cnt = v.acnt[parity];
    // workgroups beyond the list leave at once; the others (at least workgroup 0) take part in the retirement protocol
    unsigned nb = (unsigned)((cnt + kMfBlock - 1) / kMfBlock);
    if (nb > gridDim.x) nb = gridDim.x;
    if (nb == 0) nb = 1;
    if (blockIdx.x >= nb) return;
    __shared__ SweepLds s;
    sweep_lds_init(s);
    const int* __restrict__ in = v.act[parity];
    bool any = false;
    const int stride = (int)(gridDim.x * kMfBlock);
    const int rounded = (cnt + kMfBlock - 1) / kMfBlock * kMfBlock;  // whole workgroups iterate together (barriers, appends)
    for (int i = (int)(blockIdx.x * kMfBlock + threadIdx.x); i < rounded; i += stride) {
        const int u = i < cnt ? in[i] : -1;
        int pushed = -1;
        bool listed = false;   // mf_listed(v, u) with the values the step ended on (no second round of gathers)
        any |= mf_sweep_step(v, u, prev, cur, true, s, &pushed, &listed);
        const bool again = u >= 0 && listed && mf_list_claim(v, u, stamp);
        const bool fresh = pushed >= 0 && mf_list_claim(v, pushed, stamp);
        mf_append2(&v.acnt[1 - parity], v.act[1 - parity], u, again, pushed, fresh);
    }
    mf_sweep_flush(v, cur, true, s, any);
}

// ---- lambda = 0: closed-form move (maxflow_l0.hip.h) -------------------------------------------------------------------
__global__ __launch_bounds__(kMfBlock) void mf_k_l0_reduce(const long long* __restrict__ dq, const int* __restrict__ labels,
                                                           int64_t n, int L, int alpha, long long* __restrict__ sums)
{
    __shared__ unsigned long long s_sum[2 * kMfMaxLabels];
    if (threadIdx.x < 2 * kMfMaxLabels) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    if (u < n) {
        const int lu = labels[u];
        if (lu != alpha) {
            long long rt, ex;
            l0_site_terms(dq, n, u, lu, alpha, &rt, &ex);
            if (rt) atomicAdd(&s_sum[lu * 2], (unsigned long long)rt);
            if (ex) atomicAdd(&s_sum[lu * 2 + 1], (unsigned long long)ex);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * L && s_sum[threadIdx.x] != 0)
        atomicAdd((unsigned long long*)&sums[threadIdx.x], s_sum[threadIdx.x]);
}

struct L0Arg {
    int switch_any;
    unsigned char all[kMfMaxLabels];
};

__global__ __launch_bounds__(kMfBlock) void mf_k_l0_apply(const long long* __restrict__ dq, int* __restrict__ labels,
                                                          int64_t n, int alpha, L0Arg dec, int* __restrict__ changed)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    bool sw = false;
    if (u < n) {
        const int lu = labels[u];
        if (lu != alpha && l0_site_switches(dq, n, u, lu, alpha, dec.switch_any, dec.all[lu])) {
            labels[u] = alpha;
            sw = true;
        }
    }
    const int c = __syncthreads_count(sw ? 1 : 0);
    if (threadIdx.x == 0 && c > 0) atomicAdd(changed, c);
}

// The same move with the decision taken on the device (every workgroup evaluates the O(L) rule from the finished sums), so
// that a whole cycle of L moves is enqueued without a host round trip: at lambda = 0 a PEARL iteration is ~20 moves and
// two synchronisations per move were most of its time (C3: 324 iterations).
__global__ __launch_bounds__(kMfBlock) void mf_k_l0_apply_dev(const long long* __restrict__ dq, int* __restrict__ labels, int64_t n,
                                                              int L, int alpha, long long h_q, const long long* __restrict__ sums,
                                                              const int* __restrict__ cnt, int* __restrict__ changed,
                                                              int* __restrict__ evaluated)
{
    __shared__ L0Decision s_dec;
    __shared__ int s_live;
    if (threadIdx.x == 0) {
        s_live = (int64_t)cnt[alpha] != n;  // every site already carries alpha: nothing to evaluate
        if (s_live) l0_decide(L, alpha, h_q, sums, cnt, &s_dec);
        if (blockIdx.x == 0 && s_live) *evaluated = 1;
    }
    __syncthreads();
    if (!s_live || !s_dec.switch_any) return;
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    bool sw = false;
    if (u < n) {
        const int lu = labels[u];
        if (lu != alpha && l0_site_switches(dq, n, u, lu, alpha, 1, s_dec.all[lu])) {
            labels[u] = alpha;
            sw = true;
        }
    }
    const int c = __syncthreads_count(sw ? 1 : 0);
    if (threadIdx.x == 0 && c > 0) atomicAdd(changed, c);
}

// ---- one BFS level: the sites labelled k-1 label their residual in-neighbours k; then the hub part if a hub received
// distance k-1.  A fixed, small grid with grid-stride loops: deep searches run hundreds of levels of a few thousand
// sites each (the relay sites of a new-instance move at N = 1e6), and a grid sized for all sites cost ~54 us per level.
// Eight lanes share one frontier site and stride over its arcs ("virtual warp"): a lane-per-site loop is a chain of
// ~5 dependent gathers per arc (measured ~45 us per level for frontiers of a few thousand sites).
constexpr int kBfsLevelBlocks = 256;
// lanes that share one frontier site: 4 (two arcs each at the usual degree of ~6) instead of 8 doubles the sites per pass - the
// frontiers of a find6DPoses call average 19 000 sites, i.e. 2.3 passes of 8 192 - and measured 14.2 vs 17.1 us per large level
constexpr int kBfsLanesLog = 2;
constexpr int kBfsLanes = 1 << kBfsLanesLog;

// level 1: every site with residual capacity to t
__global__ __launch_bounds__(kMfBlock) void mf_k_bfs_init(MfView v, int, int)
{
    __shared__ int s_min[kMfMaxLabels];
    __shared__ Stage s_stage;
    if (threadIdx.x < kMfMaxLabels) s_min[threadIdx.x] = kMfInf;
    if (threadIdx.x == 0) s_stage.count = 0;
    __syncthreads();
    // grid-stride over 256-site chunks: a workgroup flushes its stage only when it is nearly full, so the level counter
    // sees a few hundred atomics instead of one per chunk (3906 serialised ~20 ns atomics were most of the 75 us this
    // kernel took at N = 1e6)
    bool r = false;
    const int64_t chunks = (v.n + kMfBlock - 1) / kMfBlock;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const int64_t u = c * kMfBlock + threadIdx.x;
        if (u < v.n) r |= mf_body_bfs_init(v, u, s_min, &s_stage.count, s_stage.list);
        __syncthreads();
        if (s_stage.count > kStageCap - kMfBlock) stage_flush(v, s_stage, 1);
    }
    stage_flush(v, s_stage, 1);
    const int count = __syncthreads_count(r ? 1 : 0);
    if ((int)threadIdx.x < v.L && s_min[threadIdx.x] != kMfInf) mf_report_min(&v.bfs_hub_d[threadIdx.x], s_min[threadIdx.x]);
    if (threadIdx.x == 0 && count > 0) mf_report_flag(&v.flags[0], 1);
}

// One BFS level for the calling workgroup (all of its threads): the frontier part (F sites of level k-1 at `fin`), then the
// hub part when a hub received distance k-1 (`ev`).  level_base = where level k starts in `order`.  Returns "labelled".
// stage_margin: the most one pass can append (256 threads x arcs per thread); 0 = append straight to `order`
__device__ __forceinline__ bool bfs_level_body(const MfView& v, int k, int F, const int* __restrict__ fin, int level_base, int ev,
                                               int stage_margin, int* s_min, Stage& s_stage, bool hubs)
{
    bool r = false;
    const int sub = (int)(threadIdx.x & (kBfsLanes - 1));
    // no hub at all in this move (label cost 0: the inlier / outlier cut of the local optimisation; or every label but alpha
    // unused): labelling a site then needs neither its label nor the hub table - one dependent gather less per pass
    const int64_t nthreads = (int64_t)gridDim.x * kMfBlock;
    const int64_t gtid = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    int* const scnt = stage_margin > 0 ? &s_stage.count : nullptr;
    int* const slist = stage_margin > 0 ? s_stage.list : nullptr;
    const int full = kStageCap - (stage_margin > 256 ? stage_margin : 256);
    // Level 2, bottom-up.  Level 1 is every site with residual capacity to t - 94 % of the sites in the steady-state moves of
    // PEARL - and expanding it top-down walks all their arcs to find the few unlabelled neighbours: 250-300 us at N = 1e6,
    // 56 us at 2e5, once per search (a quarter of all BFS time of a find6DPoses call).  Here the UNLABELLED sites look for a
    // level-1 neighbour they have a residual arc to instead (u -> w residual = cap of u's own arc: the same test the
    // top-down step makes through tot - cap of w's arc): one pass over d[], arcs only for the few that are unlabelled.
    if (k == 2 && ev == 0 && v.off != nullptr && (int64_t)F * 4 > v.n) {
        const int64_t rounded = (v.n + kMfBlock - 1) / kMfBlock * kMfBlock;
        for (int64_t u0 = (int64_t)blockIdx.x * kMfBlock; u0 < rounded; u0 += nthreads) {
            const int64_t u = u0 + threadIdx.x;
            bool want = false;
            if (u < v.n && v.d[u] == kMfInf) {   // active and unlabelled (inactive sites carry kMfDead)
                for (int a = v.off[u]; a < v.off[u + 1] && !want; ++a)
                    want = v.cap[a] > 0 && v.d[v.idx[a]] == 1;
            }
            r |= mf_bfs_label(v, u < v.n ? u : 0, k, s_min, want, level_base, scnt, slist, hubs);
            if (__syncthreads_or(s_stage.count > full)) stage_flush(v, s_stage, k, level_base);
        }
        stage_flush(v, s_stage, k, level_base);
        return r;
    }
    // workgroup-uniform loops (stage_flush has barriers): every pass handles nthreads / kBfsLanes frontier sites
    for (int64_t q0 = 0; v.off != nullptr && q0 < F; q0 += nthreads >> kBfsLanesLog) {
        const int64_t q = q0 + (gtid >> kBfsLanesLog);
        if (q < F) {
            const int w = __hip_atomic_load(&fin[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int a = v.off[w] + sub; a < v.off[w + 1]; a += kBfsLanes) {
                const int uu = v.idx[a];
                // residual of the reverse arc uu -> w without the gather through rev[a]; label test folded into d (kMfDead).
                // The compare-and-swap itself tests "still unlabelled" (no load of d first: one round trip less per pass;
                // a failed CAS on an already labelled neighbour costs what the load did)
                const bool want = v.tot[a] - __hip_atomic_load(&v.cap[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
                r |= mf_bfs_label(v, uu, k, s_min, want, level_base, scnt, slist, hubs);
            }
        }
        if (__syncthreads_or(s_stage.count > full)) stage_flush(v, s_stage, k, level_base);
    }
    if (ev != 0) {
        const int64_t rounded = (v.n + kMfBlock - 1) / kMfBlock * kMfBlock;  // whole workgroups iterate together
        for (int64_t u0 = (int64_t)blockIdx.x * kMfBlock; u0 < rounded; u0 += nthreads) {
            const int64_t u = u0 + threadIdx.x;
            r |= u < v.n ? mf_body_bfs_hubpass(v, u, k, (ev & 1) != 0, s_min, scnt, slist)
                         : mf_bfs_label(v, 0, k, s_min, false, level_base, scnt, slist);
            if (__syncthreads_or(s_stage.count > full)) stage_flush(v, s_stage, k, level_base);
        }
    }
    stage_flush(v, s_stage, k, level_base);
    return r;
}

// "which hub events fire at level k" (mf_bfs_hub_events) and "does any hub exist" with one lane per label: the scalar loops over
// the labels were a chain of ~2 L dependent loads at the head of every level (1.7 us of the ~11 a level over 10^4 sites takes).
// s_flag[0] = events, s_flag[1] = any hub.  All threads call it; ends with a barrier.
__device__ __forceinline__ void bfs_hub_flags(const MfView& v, int k, int* s_flag, bool coherent)
{
    if (threadIdx.x == 0) {
        const int ha = v.has_alpha_hub[0];
        const int da = ha ? (coherent ? __hip_atomic_load(&v.bfs_hubA_d[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : v.bfs_hubA_d[0]) : kMfInf;
        s_flag[0] = (ha && da == k - 1) ? 1 : 0;
        s_flag[1] = ha != 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < v.L) {
        const int ex = v.hub_exists[threadIdx.x];
        if (ex) {
            atomicOr(&s_flag[1], 1);
            const int hd = coherent ? __hip_atomic_load(&v.bfs_hub_d[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : v.bfs_hub_d[threadIdx.x];
            if (ex == 2 && hd == k - 1) atomicOr(&s_flag[0], 2);   // == 2: some member holds f > 0 (mf_bfs_hub_events)
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kMfBlock) void mf_k_bfs_level(MfView v, int k, int stage_margin)
{
    __shared__ int s_min[kMfMaxLabels];
    __shared__ Stage s_stage;
    if (threadIdx.x < kMfMaxLabels) s_min[threadIdx.x] = kMfInf;
    __shared__ int s_flag[2];
    if (threadIdx.x == 0) s_stage.count = 0;
    bfs_hub_flags(v, k, s_flag, false);   // (plain reads: a hub's distance k-1 was written by an earlier kernel, see mf_bfs_hub_events)
    const bool r = bfs_level_body(v, k, v.fcount[(k - 1) % 3], v.order + v.lvl[k - 1], mf_level_base(v, k), s_flag[0],
                                  stage_margin, s_min, s_stage, s_flag[1] != 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        v.lvl[k] = mf_level_base(v, k);
        v.fcount[(k + 1) % 3] = 0;  // slot of the level after this one
    }
    const int count = __syncthreads_count(r ? 1 : 0);
    if ((int)threadIdx.x < v.L && s_min[threadIdx.x] != kMfInf) mf_report_min(&v.bfs_hub_d[threadIdx.x], s_min[threadIdx.x]);
    if (threadIdx.x == 0 && count > 0) mf_report_flag(&v.flags[0], k);
}

// ---- minimal SOURCE side (source_reach mode) ---------------------------------------------------------------------------
// apply() hands alpha to every site that cannot reach t - the complement of the minimal sink side, which is what BK's
// what_segment(default = SOURCE) yields for an expansion move.  The inlier / outlier cut of the local optimisation is stated
// the other way round ("inliers = the sites that reach t", ties -> outlier), and solving it in that orientation makes 95 % of
// the sites hold excess.  Solved with the terminals swapped (every site "outlier", alpha = "inlier") the same answer is the
// set the SOURCE reaches in the residual graph of the maximum flow: exactly the sites reachable from the excess that stayed
// stranded (returning it to s only frees arcs back towards its origins, which are reachable from where it sits), while a site
// that neither reaches t nor is reached from s stays "outlier" - as in the original orientation, where it does not reach t.
// Jacobi-style growth over the source-side sites (d == kMfInf): a handful of rounds, the stranded sites are dense in the set.
__global__ __launch_bounds__(kMfBlock) void mf_k_src_seed(MfView v, int stamp, int)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    if (u >= v.n || v.labels[u] == v.alpha) return;
    if (v.d[u] == kMfInf && v.ex[u] > 0) v.mark[u] = stamp;
}

__global__ __launch_bounds__(kMfBlock) void mf_k_src_grow(MfView v, int stamp, int)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    bool grew = false;
    if (u < v.n && v.labels[u] != v.alpha && v.d[u] == kMfInf && __hip_atomic_load(&v.mark[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp) {
        for (int a = v.off[u]; a < v.off[u + 1]; ++a) {
            const int w = v.idx[a];
            // residual of w -> u = cap[rev[a]] = tot[a] - cap[a]; w is on the list iff marked (inactive sites never are)
            if (v.tot[a] - v.cap[a] > 0 && __hip_atomic_load(&v.mark[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == stamp) { grew = true; break; }
        }
        if (grew) __hip_atomic_store(&v.mark[u], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__syncthreads_or(grew) && threadIdx.x == 0) __hip_atomic_store(&v.flags[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kMfBlock) void mf_k_src_finish(MfView v, int stamp, int)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    if (u >= v.n || v.labels[u] == v.alpha) return;
    if (v.d[u] == kMfInf && v.mark[u] != stamp) v.d[u] = kMfInf - 1;   // not reached from s: keeps its label
}

// stranded excess of the sites (maxflow_body.hip.h mf_body_stuck_excess): per-workgroup sum, one atomic per workgroup
__global__ __launch_bounds__(kMfBlock) void mf_k_stuck(MfView v, unsigned long long* __restrict__ out, int)
{
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    long long e = u < v.n ? mf_body_stuck_excess(v, u) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
    if ((threadIdx.x & 63) == 0 && e > 0) atomicAdd(&s_sum, (unsigned long long)e);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum > 0) atomicAdd(out, s_sum);
}

// ---- wave pass: the sites of BFS level k push into level k-1 (maxflow_body.hip.h mf_body_wave) -------------------------
constexpr int kWaveBlocks = 256;

__global__ __launch_bounds__(kMfBlock) void mf_k_wave(MfView v, int k)
{
    const int lo = v.lvl[k], hi = v.lvl[k + 1];
    for (int i = lo + (int)(blockIdx.x * kMfBlock + threadIdx.x); i < hi; i += kWaveBlocks * kMfBlock)
        mf_body_wave(v, v.order[i], k);
}

// The sweep epilogue (maxflow_body.hip.h mf_body_sweep_epilogue) with one LANE per label: the one-thread version walks the
// labels through a chain of dependent loads (~5.3 us; at C5 a call ran 80 k of them, 19 % of its GPU time).  Same result.
__device__ __forceinline__ void mf_sweep_epilogue_wave(const MfView& v, int cur, int next, int consumed)
{
    auto ld32 = [](const int* p) { return *p; };          // (everything read here was written by earlier kernels)
    auto ld64 = [](const long long* p) { return *p; };
    const int l = (int)threadIdx.x;   // one wave, L <= 64
    const int prev = (cur + 2) % 3;
    // every load up front (one round trip): the kernel is nothing but its dependent-load chain
    const bool in = l < v.L;
    const int exists = in ? v.hub_exists[l] : 0;
    const long long he = in ? ld64(&v.hub_e[l]) : 0;
    int mc = in ? ld32(&v.hub_min[cur * v.L + l]) : kMfInf;
    const int mp = in ? v.hub_min[prev * v.L + l] : kMfInf;
    const int work = ld32(&v.flags[1]);
    const int moved = ld32(&v.flags[8]), stall = v.flags[11];
    const int hub_a = v.has_alpha_hub[0];
    const long long hae = ld64(v.hubA_e);
    const unsigned long long ham = v.hubA_min[cur];
    bool hub_act = false;
    if (in) {
        if (exists && (consumed >= 0 || he <= 0) && mc == kMfInf) {   // no scan was requested: keep the last known height
            mc = mp;
            v.hub_min[cur * v.L + l] = mc;
        }
        v.hub_min[next * v.L + l] = kMfInf;
        hub_act = exists && he > 0 && mc != kMfInf;
    }
    const bool any = __ballot(hub_act) != 0;
    if (l == 0) {
        int act = work;
        if (any) act = 1;
        if (hub_a && hae > 0 && ham != ~0ull) act = 1;
        v.hubA_min[next] = ~0ull;
        v.hubA_want[next] = 0;
        v.flags[4] = act;
        v.flags[1] = 0;
        v.flags[11] = moved ? 0 : stall + 1;   // as mf_body_sweep_epilogue
        v.flags[8] = 0;
        if (consumed >= 0) {
            if (v.swept) v.swept[0] += v.acnt[consumed];   // (as mf_body_sweep_epilogue: what the list sweep visited, for the labelling roofline)
            v.acnt[consumed] = 0;
        }
    }
}

__global__ void mf_k_single(MfView v, int what, int a0, int a1, int a2, int* pub, int seq)
{
    if (blockIdx.x != 0) return;
    if (what == 3) {   // all 64 lanes
        if (!mf_sweep_idle(v)) mf_sweep_epilogue_wave(v, a0, a1, a2);
        if (pub != nullptr) mf_publish(v, pub, seq);   // a read-back follows this sweep
        return;
    }
    if (threadIdx.x != 0) return;
    switch (what) {
    case 0: mf_body_hub_setup(v); break;
    case 1: mf_body_bfs_reset(v); break;
    case 2: mf_body_bfs_finish(v, a0, a1); break;
    }
}

__global__ __launch_bounds__(kMfBlock) void graph_reverse_kernel(int64_t n, const int* __restrict__ off,
                                                                 const int* __restrict__ idx, int* __restrict__ rev,
                                                                 int* __restrict__ bad)
{
    const int64_t u = (int64_t)blockIdx.x * kMfBlock + threadIdx.x;
    if (u >= n) return;
    for (int a = off[u]; a < off[u + 1]; ++a) {
        const int q = idx[a];
        int r = -1;
        for (int b = off[q]; b < off[q + 1]; ++b)
            if (idx[b] == (int)u) { r = b; break; }
        if (r < 0) { *bad = 1; r = a; }
        rev[a] = r;
    }
}

int graph_build_reverse(pgx_ctx* ctx)
{
    ctx->graph_version += 1;
    if (ctx->gE == 0) return PGX_OK;
    PGX_TRY(ensure(ctx, ctx->scratch, 64));
    PGX_HIP(ctx, hipMemsetAsync(ctx->scratch.p, 0, 4, ctx->stream));
    const int blocks = (int)((ctx->gn + kMfBlock - 1) / kMfBlock);
    hipLaunchKernelGGL(graph_reverse_kernel, dim3((unsigned)blocks), dim3(kMfBlock), 0, ctx->stream, ctx->gn,
                       ctx->goff.as<int>(), ctx->gidx.as<int>(), ctx->grev.as<int>(), (int*)ctx->scratch.p);
    PGX_HIP(ctx, hipGetLastError());
    int bad = 0;
    PGX_TRY(d2h(ctx, &bad, ctx->scratch.p, 4));
    PGX_TRY(sync_deliver(ctx));
    if (bad) {
        ctx->gn = 0;
        ctx->gE = 0;
        return fail(ctx, PGX_ERR_INVALID, "pgx_set_graph: neighbour lists are not symmetric (u in N(v) but v not in N(u))");
    }
    return PGX_OK;
}

namespace {


}  // namespace
}  // namespace pgx
#include "maxflow_xcd.hip.h"
namespace pgx {
namespace {

struct HipBackend {
    pgx_ctx* ctx;
    MaxflowState* st;
    unsigned blocks;
    bool v_has_graph;
    unsigned list_blocks;
    hipError_t err = hipSuccess;

    int stage_margin() const
    {
        // staging needs room for everything one pass can append: 256 threads x ceil(max degree / kBfsLanes) arcs each
        const int per_pass = kMfBlock * ((ctx->max_degree + kBfsLanes - 1) / kBfsLanes);
        return per_pass <= kStageCap / 2 ? (per_pass > 0 ? per_pass : kMfBlock) : 0;
    }

    void check() { if (err == hipSuccess) err = hipGetLastError(); }
    template <class K> void site(K k, const MfView& v, int a0 = 0, int a1 = 0)
    {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, v, a0, a1);
        check();
    }
    void single(const MfView& v, int what, int a0 = 0, int a1 = 0, int a2 = 0, int* pub = nullptr, int seq = 0)
    {
        hipLaunchKernelGGL(mf_k_single, dim3(1), dim3(64), 0, ctx->stream, v, what, a0, a1, a2, pub, seq);
        check();
    }
    int publish = 1;             // PGX_MF_PUBLISH=0: read the flags back with a copy + hipStreamSynchronize instead
    bool pub_pending = false;    // the last kernel enqueued publishes the flags under sequence number st->pub_seq
    // waits for the published flags; false if nothing was published or the wait gave up (the caller falls back to a copy)
    bool take_published(int out[kMfFlags], int* cnt_alpha)
    {
        if (!pub_pending) return false;
        pub_pending = false;
        volatile int* p = st->h_pub;
        const int want = st->pub_seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; p[15] != want; ++spin) {
            if ((spin & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                // a kernel that faulted never publishes: let the runtime report it
                hipError_t e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess && err == hipSuccess) err = e;
                if (p[15] != want) return false;
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int k = 0; k < kMfFlags; ++k) out[k] = p[k];
        if (cnt_alpha) *cnt_alpha = p[kMfFlags];
        return true;
    }
    int read_int(const int* dptr)
    {
        hipError_t e = hipMemcpyAsync(st->h_flags, dptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        return st->h_flags[0];
    }
    static constexpr unsigned kAggBlocks = 512;   // kernels that end with per-workgroup atomics on a few addresses
    template <class K, class... A> void agg(K k, const MfView& v, A... extra)
    {
        hipLaunchKernelGGL(k, dim3(blocks < kAggBlocks ? blocks : kAggBlocks), dim3(kMfBlock), 0, ctx->stream, v, 0, 0, extra...);
        check();
    }
    void count_and_setup(const MfView& v)
    {
        hipError_t e = hipMemsetAsync(v.cnt, 0, sizeof(int) * (size_t)v.L, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        agg(mf_k_count, v);
        single(v, 0);
    }
    int read_count(const MfView& v, int l) { return read_int(v.cnt + l); }
    void init_sites(const MfView& v) { site(mf_k_init, v); }
    void bfs_reset(const MfView& v) { single(v, 1); }
    void bfs_init(const MfView& v)
    {
        const unsigned g = blocks < bfs_init_blocks ? blocks : bfs_init_blocks;
        hipLaunchKernelGGL(mf_k_bfs_init, dim3(g), dim3(kMfBlock), 0, ctx->stream, v, 0, 0);
        check();
    }
    void bfs_level(const MfView& v, int k)
    {
        const unsigned g = blocks < bfs_level_blocks ? blocks : bfs_level_blocks;
        hipLaunchKernelGGL(mf_k_bfs_level, dim3(g), dim3(kMfBlock), 0, ctx->stream, v, k, stage_margin());
        check();
    }
    int read_flag(const MfView& v, int i)
    {
        int fl[kMfFlags];
        if (take_published(fl, nullptr)) return fl[i];
        return read_int(v.flags + i);
    }
    void read_flags_and_count(const MfView& v, int out[kMfFlags], int* cnt_alpha)
    {
        if (take_published(out, cnt_alpha)) return;
        hipError_t e = hipMemcpyAsync(st->h_flags + kMfFlags, v.cnt + v.alpha, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        read_flags(v, out);
        *cnt_alpha = st->h_flags[kMfFlags];
    }
    void read_flags(const MfView& v, int out[kMfFlags])
    {
        if (take_published(out, nullptr)) return;
        hipError_t e = hipMemcpyAsync(st->h_flags, v.flags, kMfFlags * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        for (int k = 0; k < kMfFlags; ++k) out[k] = st->h_flags[k];
    }
    void wave(const MfView& v, int k)
    {
        hipLaunchKernelGGL(mf_k_wave, dim3(kWaveBlocks), dim3(kMfBlock), 0, ctx->stream, v, k);
        check();
    }
    template <class T> T peek(const T* dptr)
    {
        T x{};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(&x, dptr, sizeof(T), hipMemcpyDeviceToHost);
        return x;
    }
    void debug_dump(const MfView& v, int nact)
    {
        for (int l = 0; l < v.L; ++l)
            if (peek(v.hub_exists + l))
                std::fprintf(stderr, "    hub %d: e=%lld d=%d cnt=%d\n", l, peek(v.hub_e + l), peek(v.bfs_hub_d + l), peek(v.cnt + l));
        if (peek(v.has_alpha_hub)) std::fprintf(stderr, "    hubA: rt=%lld d=%d\n", peek(v.hubA_rt), peek(v.bfs_hubA_d));
        if (ctx->tile_debug == 4) {  // level sizes of the BFS that just ran (nact carries the last level)
            std::vector<int> lv((size_t)nact + 2);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(lv.data(), v.lvl, sizeof(int) * lv.size(), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "    level sizes:");
            for (int k = 1; k <= nact; ++k) std::fprintf(stderr, " %d", lv[(size_t)k + 1] - lv[(size_t)k]);
            std::fprintf(stderr, "\n");
            return;
        }
        if (ctx->tile_debug < 3) return;
        std::vector<long long> ex((size_t)v.n), rt((size_t)v.n);
        std::vector<int> d((size_t)v.n), lab((size_t)v.n);
        (void)hipMemcpy(ex.data(), v.ex, sizeof(long long) * (size_t)v.n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(rt.data(), v.rt, sizeof(long long) * (size_t)v.n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(d.data(), v.d, sizeof(int) * (size_t)v.n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(lab.data(), v.labels, sizeof(int) * (size_t)v.n, hipMemcpyDeviceToHost);
        long long n_act = 0, n_stuck = 0, n_exit = 0, n_relay = 0;
        double s_act = 0, s_stuck = 0, s_rt = 0;
        for (int64_t u = 0; u < v.n; ++u) {
            if (lab[u] == v.alpha) continue;
            if (ex[u] > 0 && d[u] != kMfInf) { ++n_act; s_act += (double)ex[u]; }
            if (ex[u] > 0 && d[u] == kMfInf) { ++n_stuck; s_stuck += (double)ex[u]; }
            if (rt[u] > 0) { ++n_exit; s_rt += (double)rt[u]; }
            if (ex[u] <= 0 && rt[u] <= 0 && d[u] != kMfInf) ++n_relay;
        }
        std::fprintf(stderr, "    active %lld (sum %.4g)  stuck %lld (sum %.4g)  exits %lld (sum rt %.6g)  relays %lld\n", n_act, s_act / 4294967296.0,
                     n_stuck, s_stuck / 4294967296.0, n_exit, s_rt / 4294967296.0, n_relay);
    }
    void bfs_finish(const MfView& v, int slot, int last_level) { single(v, 2, slot, last_level); }
    void count_active(const MfView& v)   // always followed by a read-back (maxflow_driver.inl): its last workgroup publishes
    {
        if (!publish || !st->h_pub) { agg(mf_k_agg<kCountActive>, v, (int*)nullptr, (int*)nullptr, 0); return; }
        hipLaunchKernelGGL(mf_k_agg<kCountActive>, dim3(blocks < kAggBlocks ? blocks : kAggBlocks), dim3(kMfBlock), 0, ctx->stream, v, 0, 0,
                           st->h_pub, st->bar.as<int>() + 8, ++st->pub_seq);
        check();
        pub_pending = true;
    }
    unsigned sweep_blocks = 512;     // workgroups of a sweep over all sites (PGX_MF_SWEEP_BLOCKS)
    void sweep(const MfView& v, int prev, int cur)
    {
        hipLaunchKernelGGL(mf_k_sweep, dim3(blocks < sweep_blocks ? blocks : sweep_blocks), dim3(kMfBlock), 0, ctx->stream, v, prev, cur);
        check();
    }
    void sweep_epilogue(const MfView& v, int cur, int next, int consumed, bool read_follows = false)
    {
        if (read_follows && publish && st->h_pub) { single(v, 3, cur, next, consumed, st->h_pub, ++st->pub_seq); pub_pending = true; }
        else single(v, 3, cur, next, consumed);
    }
    int take_stamps(const MfView& v, int count)
    {
        if (st->next_stamp > 0x3fff0000 - count) {  // stamps only grow: start over with a clean mark array
            hipError_t e = hipMemsetAsync(v.mark, 0, sizeof(int) * (size_t)v.n, ctx->stream);
            if (e != hipSuccess && err == hipSuccess) err = e;
            st->next_stamp = 1;
        }
        const int s = st->next_stamp;
        st->next_stamp += count;
        return s;
    }
    void build_list(const MfView& v, int stamp) { site(mf_k_build_list, v, stamp); }
    unsigned bfs_level_blocks = (unsigned)kBfsLevelBlocks;   // PGX_MF_LEVEL_BLOCKS
    unsigned bfs_init_blocks = 256;  // every workgroup ends with one atomic on the level counter and up to L on the hub distances:
                                     // ~20 ns each, serialised per address (PGX_MF_INIT_BLOCKS)
    void sweep_list(const MfView& v, int prev, int cur, int parity, int stamp)
    {
        hipLaunchKernelGGL(mf_k_sweep_list, dim3(list_blocks), dim3(kMfBlock), 0, ctx->stream, v, prev, cur, parity, stamp);
        check();
    }
    long long stuck_excess(const MfView& v)
    {
        // sites: device reduction into the first word of the (free after the BFS) histogram-sized scratch in `lists`
        unsigned long long* d_sum = (unsigned long long*)v.hubA_want;  // 3 x 8 B, unused while the alpha hub is gated off
        hipError_t e = hipMemsetAsync(d_sum, 0, 8, ctx->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        hipLaunchKernelGGL(mf_k_stuck, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, v, d_sum, 0);
        check();
        long long total = (long long)peek((const long long*)d_sum);
        for (int l = 0; l < v.L; ++l)
            if (peek(v.hub_exists + l)) { const long long he = peek(v.hub_e + l); if (he > 0) total += he; }
        return total;
    }
    void apply(const MfView& v)   // followed by the read-back of the number of relabelled sites (flags[2]): published as well
    {
        if (!publish || !st->h_pub) { agg(mf_k_agg<kApply>, v, (int*)nullptr, (int*)nullptr, 0); return; }
        agg(mf_k_agg<kApply>, v, st->h_pub, st->bar.as<int>() + 8, ++st->pub_seq);
        pub_pending = true;
    }
    // ---- persistent launches on one XCD (maxflow_xcd.hip.h); false = not available / declined (the move goes on with level launches)
    int xcd_enabled = 1;
    int64_t xcd_max_n = 300000;   // (pgx_ctx::mf_xcd_max_n) measured (scripts/ab_expansion.py): C3 (1e5 sites) 32.8 -> 29.2 ms, C5 (2e5) 78.0 -> 69.0, C4 (1e6) 286 -> 325: frontiers of
                                  // ~1e4 sites and a 180 MB working set want the whole GPU's memory parallelism, not one XCD's (PGX_MF_XCD_MAXN)
    int xcd_spp() const
    {
        const int deg = ctx->max_degree > 0 ? ctx->max_degree : 1;
        int spp = (kXcdStage / 2) / deg;
        if (spp > kXcdBlock / 8) spp = kXcdBlock / 8;
        return spp;
    }
    XcdCtl* xcd_prepare()
    {
        if (ensure(ctx, st->xcd, 512) != PGX_OK) return nullptr;
        if (!st->h_xcd && hipHostMalloc((void**)&st->h_xcd, 128, hipHostMallocDefault) != hipSuccess) return nullptr;
        hipError_t e = hipMemsetAsync(st->xcd.p, 0, 512, ctx->stream);
        if (e != hipSuccess) { if (err == hipSuccess) err = e; return nullptr; }
        return (XcdCtl*)st->xcd.p;
    }
    void xcd_prof(XcdCtl* c, const char* what)
    {
        unsigned long long pr[8];
        (void)hipMemcpy(pr, c->prof, sizeof(pr), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "    xcd %s us: first list %.0f, level 1 %.0f, level 2 %.0f, levels %.0f, list passes %.0f; of all that in barriers %.0f\n", what,
                     pr[0] * 0.01, pr[1] * 0.01, pr[2] * 0.01, pr[3] * 0.01, pr[4] * 0.01, pr[5] * 0.01);
    }
    // the further hub-free rounds of a hard move
    bool xcd_rounds(const MfView& v, const MfTuning& tune, int out[8])
    {
        const int spp = xcd_spp();
        if (!xcd_enabled || st->xcd_broken || spp < 4 || v.n > xcd_max_n) return false;   // (spp < 4: rows too long for the staging buffer)
        XcdCtl* c = xcd_prepare();
        if (!c) return false;
        const int stamp0 = take_stamps(v, tune.xcd_max_rounds * (tune.sweeps_list + 3) + 8);
        hipLaunchKernelGGL(mf_k_xcd_rounds, dim3(kXcdGrid), dim3(kXcdBlock), 0, ctx->stream, v, c, tune.xcd_max_rounds, tune.sweeps_list,
                           tune.stall_sweeps, stamp0, spp);
        check();
        hipError_t e = hipMemcpyAsync(st->h_xcd, c->out, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { if (err == hipSuccess) err = e; return false; }
        for (int k = 0; k < 8; ++k) out[k] = st->h_xcd[k];
        if (tune.debug) xcd_prof(c, "rounds");
        if (out[4] <= 0) { st->xcd_broken = true; return false; }
        st->xcd_launches += 1;
        st->xcd_rounds += out[0];
        st->xcd_done += out[3] == 1 ? 1 : 0;
        st->xcd_swept16 += out[6];
        st->xcd_levels += out[1];
        st->xcd_sweeps += out[2];
        return out[4] > 0;
    }
    // one global relabel with the level loop inside the launch; fills the flags the driver reads after a search
    bool xcd_search(const MfView& v, int slot, int fl[kMfFlags], int* cnt_alpha, int* levels)
    {
        const int spp = xcd_spp();
        if (!xcd_enabled || st->xcd_broken || spp < 4 || v.n > xcd_max_n) return false;
        XcdCtl* c = xcd_prepare();
        if (!c) return false;
        hipLaunchKernelGGL(mf_k_xcd_search, dim3(kXcdGrid), dim3(kXcdBlock), 0, ctx->stream, v, c, slot, spp);
        check();
        static_assert(sizeof(int) * (8 + 16) <= 128, "pinned read-back buffer");
        hipError_t e = hipMemcpyAsync(st->h_xcd, c->out, (8 + 16) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);   // out[8] | flags[16]
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { if (err == hipSuccess) err = e; return false; }
        if (st->h_xcd[4] <= 0) st->xcd_broken = true;   // nobody took part: see MaxflowState::xcd_broken
        if (st->h_xcd[3] != 1) return false;   // declined (a materialised alpha hub) or nobody on XCC 0
        for (int k = 0; k < kMfFlags; ++k) fl[k] = st->h_xcd[8 + k];
        *cnt_alpha = st->h_xcd[8 + kMfFlags];
        *levels = st->h_xcd[1];
        if (ctx->tile_debug) xcd_prof(c, "search");
        st->xcd_searches += 1;
        st->xcd_levels += *levels;
        return true;
    }
    void keep_source_reachable_only(const MfView& v)
    {
        const int stamp = take_stamps(v, 1);
        site(mf_k_src_seed, v, stamp);
        for (int round = 0; round < 1 << 20; ++round) {
            hipError_t e = hipMemsetAsync(v.flags + 5, 0, sizeof(int), ctx->stream);
            if (e != hipSuccess && err == hipSuccess) err = e;
            for (int r = 0; r < 4; ++r) site(mf_k_src_grow, v, stamp);   // four rounds per read-back
            if (read_int(v.flags + 5) == 0 || err != hipSuccess) break;
        }
        site(mf_k_src_finish, v, stamp);
    }
};

}  // namespace

void maxflow_free(pgx_ctx* ctx)
{
    if (!ctx->mf) return;
    MaxflowState* st = ctx->mf;
    release(st->cap); release(st->tot); release(st->ex); release(st->rt); release(st->d); release(st->f); release(st->g);
    release(st->small); release(st->front); release(st->lists); release(st->bar); release(st->xcd);
    if (st->h_xcd) (void)hipHostFree(st->h_xcd);
    if (st->h_flags) (void)hipHostFree(st->h_flags);
    if (st->h_pub) (void)hipHostFree(st->h_pub);
    delete st;
    ctx->mf = nullptr;
}

// pgx_expansion_schedule: how the level-synchronous solver spent its dependent steps since pgx_create
int maxflow_schedule_stats(pgx_ctx* ctx, int64_t out[8])
{
    for (int k = 0; k < 8; ++k) out[k] = 0;
    MaxflowState* st = ctx->mf;
    if (!st) return PGX_OK;
    out[0] = st->xcd_launches;
    out[1] = st->xcd_rounds;
    out[2] = st->xcd_done;
    out[3] = st->xcd_searches;
    out[4] = st->xcd_swept16 * 16;
    if (st->bar.p) {
        long long sw = 0;
        PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        PGX_HIP(ctx, hipMemcpy(&sw, (const long long*)st->bar.p + 6, sizeof(sw), hipMemcpyDeviceToHost));
        out[5] = sw;
    }
    out[6] = st->xcd_levels;
    out[7] = st->xcd_sweeps;
    return PGX_OK;
}

// lambda = 0: one reduction pass, a host decision over <= 64 labels, one apply pass (maxflow_l0.hip.h)
static int expand_alpha_l0(pgx_ctx* ctx, int64_t h_q, int alpha, int64_t* changed)
{
    const int64_t n = ctx->dq_n;
    const int L = ctx->L;
    *changed = 0;
    if (!ctx->mf) {
        ctx->mf = new MaxflowState();
        PGX_HIP(ctx, hipHostMalloc((void**)&ctx->mf->h_flags, 64, hipHostMallocDefault));
    }
    MaxflowState* st = ctx->mf;
    // small state: sums[2L] i64 | cnt[L] i32 | changed i32
    const size_t bytes = (size_t)2 * L * 8 + (size_t)L * 4 + 8;
    PGX_TRY(ensure(ctx, st->small, bytes + 4096));
    long long* d_sums = (long long*)st->small.p;
    int* d_cnt = (int*)((char*)st->small.p + (size_t)2 * L * 8);
    int* d_changed = d_cnt + L;
    PGX_HIP(ctx, hipMemsetAsync(st->small.p, 0, bytes, ctx->stream));
    const unsigned blocks = (unsigned)((n + kMfBlock - 1) / kMfBlock);
    MfView v{};
    v.n = n; v.L = L; v.alpha = alpha; v.labels = ctx->labels.as<int>(); v.cnt = d_cnt;
    hipLaunchKernelGGL(mf_k_count, dim3(blocks < 512u ? blocks : 512u), dim3(kMfBlock), 0, ctx->stream, v, 0, 0);
    hipLaunchKernelGGL(mf_k_l0_reduce, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, ctx->dq.as<long long>(),
                       ctx->labels.as<int>(), n, L, alpha, d_sums);
    PGX_HIP(ctx, hipGetLastError());
    std::vector<unsigned char> host(bytes);
    PGX_TRY(d2h(ctx, host.data(), st->small.p, bytes));
    PGX_TRY(sync_deliver(ctx));
    const long long* sums = (const long long*)host.data();
    const int* cnt = (const int*)(host.data() + (size_t)2 * L * 8);
    if ((int64_t)cnt[alpha] == n) return PGX_OK;
    ctx->stats[0] += 1;
    L0Decision dec;
    l0_decide(L, alpha, (long long)h_q, sums, cnt, &dec);
    if (!dec.switch_any) return PGX_OK;
    L0Arg arg;
    arg.switch_any = 1;
    for (int l = 0; l < kMfMaxLabels; ++l) arg.all[l] = (l < L && dec.all[l]) ? 1 : 0;
    hipLaunchKernelGGL(mf_k_l0_apply, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, ctx->dq.as<long long>(),
                       ctx->labels.as<int>(), n, alpha, arg, d_changed);
    PGX_HIP(ctx, hipGetLastError());
    int ch = 0;
    PGX_TRY(d2h(ctx, &ch, d_changed, sizeof(int)));
    PGX_TRY(sync_deliver(ctx));
    *changed = ch;
    ctx->stats[4] += ch;
    return PGX_OK;
}

// One cycle over all labels at lambda = 0, enqueued back to back; one read-back at the end.  changed[alpha] = sites that
// took alpha in move alpha, evaluated[alpha] = 0 when every site already carried alpha.
int expand_cycle_l0(pgx_ctx* ctx, int64_t h_q, int64_t* changed, int* evaluated)
{
    const int64_t n = ctx->dq_n;
    const int L = ctx->L;
    if (n <= 0 || L <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: unary table not set");
    if (ctx->labels_n != n) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: labels not set (or wrong length)");
    if (ctx->labels_max >= L) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: label %d out of range (the unary table has %d labels)", ctx->labels_max, L);
    if (L > kMfMaxLabels) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: at most %d labels (got %d)", kMfMaxLabels, L);
    if (L - 1 > ctx->labels_max) ctx->labels_max = L - 1;   // the cycle may write every label
    if (!ctx->mf) {
        ctx->mf = new MaxflowState();
        PGX_HIP(ctx, hipHostMalloc((void**)&ctx->mf->h_flags, 64, hipHostMallocDefault));
    }
    MaxflowState* st = ctx->mf;
    // small state: sums[2L] i64 | cnt[L] i32 (cleared before every move) | changed[L] i32 | evaluated[L] i32
    const size_t move_bytes = (size_t)2 * L * 8 + (size_t)L * 4, res_bytes = (size_t)2 * L * 4;
    PGX_TRY(ensure(ctx, st->small, move_bytes + res_bytes + 4096));
    long long* d_sums = (long long*)st->small.p;
    int* d_cnt = (int*)((char*)st->small.p + (size_t)2 * L * 8);
    int* d_changed = d_cnt + L;
    int* d_eval = d_changed + L;
    PGX_HIP(ctx, hipMemsetAsync(d_changed, 0, res_bytes, ctx->stream));
    const unsigned blocks = (unsigned)((n + kMfBlock - 1) / kMfBlock);
    for (int alpha = 0; alpha < L; ++alpha) {
        PGX_HIP(ctx, hipMemsetAsync(st->small.p, 0, move_bytes, ctx->stream));
        MfView v{};
        v.n = n; v.L = L; v.alpha = alpha; v.labels = ctx->labels.as<int>(); v.cnt = d_cnt;
        hipLaunchKernelGGL(mf_k_count, dim3(blocks < 512u ? blocks : 512u), dim3(kMfBlock), 0, ctx->stream, v, 0, 0);
        hipLaunchKernelGGL(mf_k_l0_reduce, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, ctx->dq.as<long long>(),
                           ctx->labels.as<int>(), n, L, alpha, d_sums);
        hipLaunchKernelGGL(mf_k_l0_apply_dev, dim3(blocks), dim3(kMfBlock), 0, ctx->stream, ctx->dq.as<long long>(),
                           ctx->labels.as<int>(), n, L, alpha, (long long)h_q, d_sums, d_cnt, d_changed + alpha, d_eval + alpha);
    }
    PGX_HIP(ctx, hipGetLastError());
    std::vector<int> host((size_t)2 * L);
    PGX_TRY(d2h(ctx, host.data(), d_changed, res_bytes));
    PGX_TRY(sync_deliver(ctx));
    for (int alpha = 0; alpha < L; ++alpha) {
        changed[alpha] = host[(size_t)alpha];
        evaluated[alpha] = host[(size_t)L + alpha];
        ctx->stats[0] += evaluated[alpha] ? 1 : 0;
        ctx->stats[4] += changed[alpha];
    }
    return PGX_OK;
}

int expand_alpha_launch(pgx_ctx* ctx, int64_t lambda_q, int64_t h_q, int alpha, int64_t* changed)
{
    const int64_t n = ctx->dq_n;
    const int L = ctx->L;
    if (n <= 0 || L <= 0) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: unary table not set");
    if (ctx->labels_n != n) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: labels not set (or wrong length)");
    if (ctx->labels_max >= L) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: label %d out of range (the unary table has %d labels)", ctx->labels_max, L);
    if (alpha < 0 || alpha >= L) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: alpha %d out of range", alpha);
    if (L > kMfMaxLabels) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: at most %d labels (got %d)", kMfMaxLabels, L);
    if (alpha > ctx->labels_max) ctx->labels_max = alpha;   // the move may write alpha: a later, smaller table must see it
    if (lambda_q <= 0) return expand_alpha_l0(ctx, h_q, alpha, changed);
    return expand_alpha_on(ctx, n, L, ctx->dq.as<long long>(), ctx->labels.as<int>(), nullptr, lambda_q, h_q, alpha, changed);
}

// Region moves (maxflow_tile.hip expand_alpha_region) apply to the resident problem: a graph whose rows fit the compact arrays,
// beyond the one-workgroup size (smaller graphs are solved whole), at most 64 labels.
bool region_moves_apply(const pgx_ctx* ctx)
{
    return ctx->mf_region && ctx->max_degree >= 1 && ctx->max_degree <= 32 && ctx->L <= 64 && ctx->gn < ((int64_t)1 << 30);
}

// One expansion move on caller-chosen tables: dq [L][n] label-major, labels [n], and optionally per-arc weights wq [E]
// (used by the binary inlier/outlier cut of gclo.hip, whose pairwise weights depend on both end points).
int expand_alpha_on(pgx_ctx* ctx, int64_t n, int L, const long long* dq, int* labels, const long long* wq,
                    int64_t lambda_q, int64_t h_q, int alpha, int64_t* changed, bool source_reach)
{
    if (source_reach && h_q != 0) return fail(ctx, PGX_ERR_INVALID, "expansion move: the source-side variant takes no label cost");
    const bool pair = true;
    if (ctx->gn != n) return fail(ctx, PGX_ERR_INVALID, "pgx_expansion: lambda > 0 needs a graph over %lld sites", (long long)n);
    // Expansion moves on graphs of tile_expansion_max (1 024) .. 8192 sites try the region path FIRST (measured at 5 000 sites, C2: 6.2 ms per
    // expansion against 7.7 with every move on the whole-graph kernel, which works out of memory beyond the LDS-resident kernel's 1 024 sites;
    // 2 000 sites: 1.99 against 2.03); a move the region path declines goes to the whole-graph kernel (the caller re-runs it with the region
    // path off; unbatched: below).
    const bool region_first = !source_reach && wq == nullptr && L <= 64 && n > ctx->tile_expansion_max && region_moves_apply(ctx);
    if (ctx->mf_tile && !source_reach && n <= ctx->tile_single_max && n <= 8192 && L <= 64 && !region_first) {   // (the sizes expand_alpha_tile takes)
        const int r = expand_alpha_tile(ctx, n, L, dq, labels, lambda_q, h_q, alpha, changed, wq);
        if (r != PGX_TILE_FALLBACK) return r;
        ctx->tile_fallbacks += 1;   // the one-workgroup solver ran and gave the move back
    }
    if (!ctx->mf) {
        ctx->mf = new MaxflowState();
        PGX_HIP(ctx, hipHostMalloc((void**)&ctx->mf->h_flags, 64, hipHostMallocDefault));
    }
    MaxflowState* st = ctx->mf;
    const int64_t E = pair ? ctx->gE : 0;
    PGX_TRY(ensure(ctx, st->cap, (size_t)(E > 0 ? E : 1) * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->tot, (size_t)(E > 0 ? E : 1) * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->ex, (size_t)n * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->rt, (size_t)n * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->f, (size_t)n * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->g, (size_t)n * sizeof(long long)));
    PGX_TRY(ensure(ctx, st->d, (size_t)n * sizeof(int)));
    // small state: hub_e[L] i64 | hubA_rt i64 | hubA_min[3] u64 | cnt[L] | hub_exists[L] | bfs_hub_d[L] | hub_min[3L] |
    //              has_alpha_hub | bfs_hubA_d | flags[8]
    PGX_TRY(ensure(ctx, st->front, (size_t)(2 * n + L + 160) * sizeof(int)));  // order[n] | lvl[n + L + 160]
    if (st->mark_n != n) {  // act[2][n] | mark[n]; stamps restart with a zeroed mark array
        PGX_TRY(ensure(ctx, st->lists, (size_t)3 * n * sizeof(int)));
        if (hipMemsetAsync(st->lists.as<int>() + 2 * n, 0, sizeof(int) * (size_t)n, ctx->stream) != hipSuccess)
            return fail(ctx, PGX_ERR_HIP, "expansion move: clearing the list marks failed");
        st->mark_n = n;
        st->next_stamp = 1;
    }
    const size_t small_bytes = (size_t)(L + 1 + 3 + 4) * 8 + (size_t)(9 * L + 2 + 3 + kMfFlags + 2) * 4 + 64;
    PGX_TRY(ensure(ctx, st->small, small_bytes));
    char* sp = (char*)st->small.p;
    MfView v;
    v.n = n; v.L = L; v.alpha = alpha; v.lambda_q = lambda_q; v.h_q = h_q;
    v.dq = dq;
    v.labels = labels;
    v.wq = wq;
    v.off = pair ? ctx->goff.as<int>() : nullptr;
    v.idx = ctx->gidx.as<int>(); v.mult = ctx->gmult.as<int>(); v.rev = ctx->grev.as<int>();
    v.cap = st->cap.as<long long>(); v.tot = st->tot.as<long long>(); v.ex = st->ex.as<long long>(); v.rt = st->rt.as<long long>();
    v.d = st->d.as<int>(); v.f = st->f.as<long long>(); v.g = st->g.as<long long>();
    v.hub_e = (long long*)sp; sp += (size_t)L * 8;
    v.hubA_rt = (long long*)sp; sp += 8;
    v.hubA_min = (unsigned long long*)sp; sp += 24;
    v.hubA_e = (long long*)sp; sp += 8;
    v.hubA_want = (long long*)sp; sp += 24;
    v.cnt = (int*)sp; sp += (size_t)L * 4;
    v.hub_exists = (int*)sp; sp += (size_t)L * 4;
    v.bfs_hub_d = (int*)sp; sp += (size_t)L * 4;
    v.hub_min = (int*)sp; sp += (size_t)3 * L * 4;
    v.fcount = (int*)sp; sp += 12;
    v.acnt = (int*)sp; sp += 8;
    v.act[0] = st->lists.as<int>(); v.act[1] = st->lists.as<int>() + n; v.mark = st->lists.as<int>() + 2 * n;
    v.order = st->front.as<int>();
    v.lvl = st->front.as<int>() + n;
    v.has_alpha_hub = (int*)sp; sp += 4;
    v.bfs_hubA_d = (int*)sp; sp += 4;
    v.flags = (int*)sp;
    v.hmax = (int)((n + L + 3 < (int64_t)kMfInf) ? (n + L + 3) : (int64_t)kMfInf - 1);
    v.swept = nullptr;
    v.gate = 1;   // an unused alpha is handled by the stranded-excess test (maxflow_body.hip.h); the materialised hub stays in the bodies for the CPU emulation

    HipBackend be{ctx, st, (unsigned)((n + kMfBlock - 1) / kMfBlock), pair, 1};
    if (!st->bar.p) {
        PGX_TRY(ensure(ctx, st->bar, 64));
        PGX_HIP(ctx, hipMemsetAsync(st->bar.p, 0, 64, ctx->stream));
    }
    v.swept = (long long*)st->bar.p + 6;   // bytes 48..55 of the once-zeroed block
    if (!st->h_pub) {
        PGX_HIP(ctx, hipHostMalloc((void**)&st->h_pub, 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(st->h_pub, 0, 64);
    }
    MfTuning tune;   // the schedule constants of maxflow_driver.inl (measured: DESIGN.md 5.4)
    if (tune.list_div > 0) be.list_blocks = (unsigned)((n / tune.list_div + kMfBlock - 1) / kMfBlock + 1);
    tune.debug = ctx->tile_debug;
    tune.bfs_hint = st->bfs_hint;
    tune.source_reach = source_reach ? 1 : 0;
    {   // schedule knobs for A/B runs (any schedule gives the same cut; defaults = the measured ones in maxflow_driver.inl)
        static const int e_wave_max = std::getenv("PGX_MF_WAVE_MAX") ? std::atoi(std::getenv("PGX_MF_WAVE_MAX")) : -1;
        static const int e_wave_from = std::getenv("PGX_MF_WAVE_FROM") ? std::atoi(std::getenv("PGX_MF_WAVE_FROM")) : -1;
        static const int e_sweeps_list = std::getenv("PGX_MF_SWEEPS_LIST") ? std::atoi(std::getenv("PGX_MF_SWEEPS_LIST")) : -1;
        static const int e_stall = std::getenv("PGX_MF_STALL") ? std::atoi(std::getenv("PGX_MF_STALL")) : -1;
        if (e_wave_max >= 0) tune.wave_max = e_wave_max;
        if (e_wave_from >= 0) tune.wave_from = e_wave_from;
        if (e_sweeps_list > 0) tune.sweeps_list = e_sweeps_list;
        if (e_stall >= 0) tune.stall_sweeps = e_stall;
        tune.xcd = ctx->mf_xcd;
        tune.xcd_search = ctx->mf_xcd_search;
        tune.xcd_search_min = ctx->mf_xcd_min_depth;
        be.xcd_enabled = ctx->mf_xcd || ctx->mf_xcd_search;
        be.xcd_max_n = ctx->mf_xcd_max_n;
        if (ctx->mf_sweeps > 0) { tune.sweeps_per_relabel = tune.sweeps_list = ctx->mf_sweeps; if (tune.sweep_check > ctx->mf_sweeps) tune.sweep_check = ctx->mf_sweeps; }
    }
    // A move with few OPEN sites (no t-link: excess or relay) is solved by one workgroup on their compacted sub-graph
    // (maxflow_tile.hip expand_alpha_region); it needs the t-links and arcs set up here first and leaves them intact when it declines.
    if (!source_reach && wq == nullptr && pair && L <= 64 && region_moves_apply(ctx)) {   // (then expand_alpha_region runs its first kernel = the per-site initialisation)
        const int rr = expand_alpha_region(ctx, v, changed);   // (its first kernel is init_sites + the label count fused with the search for open sites)
        if (rr != PGX_TILE_FALLBACK) return rr;   // solved, enqueued (PGX_REGION_PENDING) or an error
        if (region_first && ctx->mf_tile && n <= ctx->tile_single_max && n <= 8192) {   // declined, and the graph fits the whole-graph kernel
            const int r = expand_alpha_tile(ctx, n, L, dq, labels, lambda_q, h_q, alpha, changed, wq);
            if (r != PGX_TILE_FALLBACK) return r;
            ctx->tile_fallbacks += 1;
        }
        tune.preinit = 1;   // the sites are initialised; the hub set-up (which does not touch them) is still to run
    }
    const int r = mf_expand_alpha(be, v, tune, changed, ctx->stats);
    if (be.err != hipSuccess) return fail(ctx, PGX_ERR_HIP, "expansion move failed: %s", hipGetErrorString(be.err));
    if (r != 0)
        return fail(ctx, PGX_ERR_NOCONVERGE, "push-relabel did not converge within %d global relabels (alpha=%d)",
                    tune.max_relabels, alpha);
    ctx->paths[3] += 1;
    return PGX_OK;
}

}  // namespace pgx
