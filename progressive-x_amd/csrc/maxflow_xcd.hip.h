// maxflow_xcd.hip.h — the dependent-step chains of an expansion move (global relabels, rounds of list sweeps) in ONE persistent
// launch on ONE XCD (round 6; included by maxflow.hip).
//
// Replaces (part of): the BK max-flow behind GCoptimizationGeneralGraph::expansion as driven by pearl::PEARL::labeling,
//                     /root/reference/src/pyprogressivex/include/PEARL.h:507-551 (sources absent upstream: U-5).
//
// Why one XCD: per-XCD L2s are not coherent with each other, so a grid-wide persistent loop needs a release / acquire pair
// per barrier (buffer_wbl2 + buffer_inv: 4-7 us, what rounds 2 and 4 measured and why the level loop stayed one launch per
// level).  Workgroups of the SAME XCD share one L2: a plain store is write-through to it, an sc1 load bypasses the CU's L1 and
// is served by it, atomics execute in or beyond it.  No fence at all; a barrier over the XCD's 32 workgroups is one relaxed
// atomic + an sc1 poll: 0.75 us bare, 1.0 us with a dependent exchange (scripts/micro/xcd_scope_bench.hip, profiles/round6_xcd_scope.txt),
// against ~12 us for a level launch and ~18 us for a list sweep + its epilogue.  32 workgroups x 1024 threads are plenty for
// frontiers and work lists of a few thousand sites.
//
// Two kernels:
//  * mf_k_xcd_search: one global relabel = the level-synchronous reverse BFS of maxflow_driver.inl (bfs_reset, bfs_init, bfs_level
//    x depth, bfs_finish, count_active) with the level loop inside the launch, beta hubs included: a hub's distance is 1 + its
//    nearest labelled member, published level by level, and a hub that some member has pulled from hands it on to its members
//    with f > 0 one level later (the hub pass).  Declines only with a materialised alpha hub (never under the product's gate).
//  * mf_k_xcd_rounds: the later rounds of a hard move - { search | drop the listed sites that no longer reach t | list sweeps
//    until the list is empty, the budget is spent or nothing has reached t for longer than the search was deep } - hub-free
//    (maxflow_body.hip.h mf_body_tail_step), until no listed site reaches t.
// Placement is not promised by HIP: every workgroup reads HW_REG_XCC_ID, the ones on XCC 0 take part (32 of 256 with the observed
// round-robin), the others register and leave; the participants learn their number once all workgroups have registered.
// Exactness does not rest on any of this for the rounds (they leave a valid preflow whatever they read - capacities and excesses
// move by atomics - and the move is declared finished only by a search that follows); the search kernel's labels are checked
// like every other path: against the oracle's cut (tests/test_fullsize_pins.py, the expansion soaks).
#pragma once

namespace pgx {
namespace {

constexpr int kXcdBlock = 1024;
constexpr int kXcdGrid = 256;          // one workgroup per CU; the 32 on XCC 0 do the work
constexpr int kXcdStage = 6144;        // LDS staging of appends per workgroup (ints)
constexpr int kXcdHalf = kXcdStage / 2;
constexpr unsigned kXcdSpinLimit = 40000000u;   // a barrier that long (~10 s) means a participant died: give up, never hang the box

struct XcdCtl {            // device words, zeroed by the host before the launch
    unsigned arrived;      // workgroups that have started (all of the grid)
    unsigned joined;       // ... of which on XCC 0
    unsigned bar;          // barrier arrivals (monotonic)
    unsigned dead;         // a participant gave up waiting
    int fcount[3];         // frontier sizes by level % 3
    int fbase[3];          // where each of them starts in `order`
    int lcnt[3];           // work-list sizes by pass % 3 (list q lives in act[q & 1]); slot (q + 2) % 3 is cleared during pass q
    int moved[3];          // flow reached t during pass q, same rotation
    int active;            // search kernel: sites that hold excess and reach t
    int level1;            // level 1 of the search is not empty
    unsigned long long prof[8];   // wall_clock64 ticks (10 ns) of workgroup 0: 0 first list, 1 level 1, 2 level 2, 3 levels 3.., 4 list passes, 5 barriers
    int out[8];            // 0 rounds, 1 BFS levels, 2 sweeps, 3 status (rounds: 1 = no listed site reaches t any more, 2 = round budget spent;
                           // search: 1 = done, 3 = declined), 4 participants, 5 sites on the first list, 6 sum of the list lengths swept (>> 4)
    int flags[16];         // search kernel: copy of MfView::flags + cnt[alpha] (what the driver's read-back wants)
};

__device__ __forceinline__ unsigned xcd_xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}
__device__ __forceinline__ int xcd_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcd_loadu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct XcdStage {
    int list[kXcdStage];
    int count;
    int base;
    int ok, me, part;
    int hub[64];           // per-label minimum of (distance + 1) over this workgroup's labelled members (search kernel)
    int hubsent[64];       // ... what of it this workgroup has already published to bfs_hub_d
    unsigned long long evmask;   // labels whose hub hands its distance on at the level being run
};

struct XcdRt {             // per-thread view of the launch
    XcdCtl* c;
    unsigned epoch, part;
    int me;
    int64_t T, tid;
    unsigned long long t_bar, t_mark;
    unsigned long long t_prof[6];
};
#define XCD_PROF(rt, slot) { const unsigned long long t_now = wall_clock64(); (rt).t_prof[slot] += t_now - (rt).t_mark; (rt).t_mark = t_now; }

// registration: false for the workgroups that do not take part
__device__ __forceinline__ bool xcd_join(XcdCtl* c, XcdStage& st, XcdRt& rt)
{
    if (threadIdx.x == 0) {
        const bool mine = xcd_xcc_id() == 0;
        st.me = mine ? (int)__hip_atomic_fetch_add(&c->joined, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        st.count = 0;
    }
    if (threadIdx.x < 64) { st.hub[threadIdx.x] = kMfInf; st.hubsent[threadIdx.x] = kMfInf; }
    __syncthreads();
    if (st.me < 0) return false;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (xcd_loadu(&c->arrived) < gridDim.x && ++spins < kXcdSpinLimit) __builtin_amdgcn_s_sleep(2);
        st.part = xcd_loadu(&c->arrived) >= gridDim.x ? (int)xcd_loadu(&c->joined) : 0;
    }
    __syncthreads();
    if (st.part == 0) return false;   // (some workgroup never started: leave everything as it is)
    rt.c = c;
    rt.epoch = 0;
    rt.part = (unsigned)st.part;
    rt.me = st.me;
    rt.T = (int64_t)st.part * kXcdBlock;
    rt.tid = (int64_t)st.me * kXcdBlock + threadIdx.x;
    rt.t_bar = 0;
    rt.t_mark = wall_clock64();
    for (int k = 0; k < 6; ++k) rt.t_prof[k] = 0;
    return true;
}

// all threads of all participating workgroups; stores issued before it are in the XCD's L2 when it returns
__device__ __forceinline__ bool xcd_sync(XcdStage& st, XcdRt& rt)
{
    const unsigned long long t0 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    rt.epoch += rt.part;
    if (threadIdx.x == 0) {
        XcdCtl* c = rt.c;
        __hip_atomic_fetch_add(&c->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while ((int)(xcd_loadu(&c->bar) - rt.epoch) < 0) {
            if (++spins > kXcdSpinLimit || ((spins & 0xffffu) == 0 && xcd_loadu(&c->dead) != 0)) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_store(&c->dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st.ok = ok;
    }
    __syncthreads();
    rt.t_bar += wall_clock64() - t0;
    return st.ok != 0;
}

// every thread of the workgroup calls it; moves the staged entries behind *counter in dst
__device__ __forceinline__ void xcd_flush(XcdStage& st, int* counter, int* dst)
{
    __syncthreads();
    const int n = st.count;
    if (n > 0) {
        if (threadIdx.x == 0) st.base = atomicAdd(counter, n);
        __syncthreads();
        for (int i = (int)threadIdx.x; i < n; i += kXcdBlock) dst[st.base + i] = st.list[i];
        __syncthreads();
        if (threadIdx.x == 0) st.count = 0;
    }
    __syncthreads();
}
__device__ __forceinline__ void xcd_stage(XcdStage& st, int value, bool want)
{
    if (want) st.list[atomicAdd(&st.count, 1)] = value;
}
// (called by every thread of the workgroup)
__device__ __forceinline__ void xcd_flush_if_half(XcdStage& st, int* counter, int* dst)
{
    __syncthreads();
    if (st.count > kXcdHalf) xcd_flush(st, counter, dst);
    __syncthreads();
}

// this workgroup's per-label minima of (member distance + 1) -> bfs_hub_d, one atomic per label whose minimum improved
__device__ __forceinline__ void xcd_publish_hubs(const MfView& v, XcdStage& st)
{
    __syncthreads();
    if ((int)threadIdx.x < v.L && st.hub[threadIdx.x] < st.hubsent[threadIdx.x]) {
        atomicMin(&v.bfs_hub_d[threadIdx.x], st.hub[threadIdx.x]);
        st.hubsent[threadIdx.x] = st.hub[threadIdx.x];
    }
}

// One reverse BFS from the sites with residual capacity to t over the n-links.  HUBS: also record, per label with a hub, the
// distance of its nearest labelled member + 1 in st.hub (maxflow_body.hip.h mf_bfs_label: y_beta -> u has infinite capacity).
// Level 1 is not listed (it is most of the graph in a steady-state move and a whole cluster in a hard one): level 2 runs
// bottom-up over the unlabelled sites, levels 3.. top-down from the frontier lists in `order`, eight lanes per frontier site.
// returns the last level that labelled a site (0 = nobody has residual capacity to t), or -1 if a barrier gave up.
template <bool HUBS>
__device__ __forceinline__ int xcd_bfs(const MfView& v, XcdStage& st, XcdRt& rt, int spp, int* levels)
{
    XcdCtl* c = rt.c;
    const int64_t n = v.n, T = rt.T;
    const int64_t n_round = (n + kXcdBlock - 1) / kXcdBlock * kXcdBlock;   // whole workgroups iterate together (flushes)
    bool any1 = false;
    for (int64_t u = rt.tid; u < n; u += T) {
        const int lu = v.labels[u];
        if (lu == v.alpha) continue;
        const bool first = mf_hd_load64(&v.rt[u]) > 0;
        mf_plain_store32(&v.d[u], first ? 1 : kMfInf);
        if (first) {
            any1 = true;
            if (HUBS && v.hub_exists[lu]) atomicMin(&st.hub[lu], 2);
        }
    }
    if (rt.tid == 0) { c->fcount[0] = c->fcount[1] = c->fcount[2] = 0; c->fbase[2] = 0; }
    {
        const unsigned long long m1 = __ballot(any1);   // one store per wave that labelled somebody
        if (m1 != 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m1) - 1) __hip_atomic_store(&c->level1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (HUBS) xcd_publish_hubs(v, st);
    if (!xcd_sync(st, rt)) return -1;
    XCD_PROF(rt, 1)
    for (int64_t u0 = (int64_t)rt.me * kXcdBlock; u0 < n_round; u0 += T) {
        const int64_t u = u0 + threadIdx.x;
        bool want = false;
        if (u < n && v.labels[u] != v.alpha && mf_hd_load32(&v.d[u]) == kMfInf && mf_body_tail_level2(v, u)) {
            mf_plain_store32(&v.d[u], 2);   // (only this thread looks at site u in this pass; neighbours test for d == 1)
            if (HUBS && v.hub_exists[v.labels[u]]) atomicMin(&st.hub[v.labels[u]], 3);
            want = true;
        }
        xcd_stage(st, (int)u, want);
        xcd_flush_if_half(st, &c->fcount[2], v.order);
    }
    xcd_flush(st, &c->fcount[2], v.order);
    if (HUBS) xcd_publish_hubs(v, st);
    if (!xcd_sync(st, rt)) return -1;
    XCD_PROF(rt, 2)
    int k = 3;
    int depth = xcd_load(&c->level1) != 0 ? 1 : 0;
    for (;; ++k) {
        const int fprev = xcd_load(&c->fcount[(k - 1) % 3]);
        // hub events of level k (mf_bfs_hub_events): a beta hub some member has pulled from (hub_exists == 2: only then can a member
        // hold f > 0, i.e. residual u -> y_beta) received distance k - 1 from a member labelled at level k - 2: its members with f > 0
        // that are still unlabelled get k.  The distances were published before the barrier that ended level k - 1.
        unsigned long long ev = 0;
        if (HUBS) {
            if (threadIdx.x < 64) {
                const int l = (int)threadIdx.x;
                const bool e = l < v.L && v.hub_exists[l] == 2 && xcd_load(&v.bfs_hub_d[l]) == k - 1;
                const unsigned long long m = __ballot(e);
                if (l == 0) st.evmask = m;
            }
            __syncthreads();
            ev = st.evmask;
            __syncthreads();
        }
        if (fprev == 0 && ev == 0) break;
        if (fprev > 0) depth = k - 1;
        const int pbase = xcd_load(&c->fbase[(k - 1) % 3]);
        const int kbase = pbase + fprev;
        if (rt.tid == 0) { c->fbase[k % 3] = kbase; c->fcount[(k + 1) % 3] = 0; }
        const int f_round = (fprev + spp - 1) / spp * spp;
        for (int i0 = rt.me * spp; i0 < f_round; i0 += (int)rt.part * spp) {
            const int slot = (int)(threadIdx.x >> 3), lane = (int)(threadIdx.x & 7);
            const int i = i0 + slot;
            if (slot < spp && i < fprev) {
                const int w = xcd_load(&v.order[pbase + i]);
                const int a_hi = v.off[w + 1];
                for (int a = v.off[w] + lane; a < a_hi; a += 8) {
                    const int u = v.idx[a];
                    // residual u -> w = tot - cap of the arc w -> u; sites that are already alpha carry d = kMfDead
                    if (v.tot[a] - mf_hd_load64(&v.cap[a]) > 0 && mf_hd_load32(&v.d[u]) == kMfInf && mf_hd_cas32(&v.d[u], kMfInf, k)) {
                        xcd_stage(st, u, true);
                        if (HUBS && v.hub_exists[v.labels[u]]) atomicMin(&st.hub[v.labels[u]], k + 1);
                    }
                }
            }
            xcd_flush_if_half(st, &c->fcount[k % 3], v.order + kbase);
        }
        if (HUBS && ev != 0) {   // the hub pass: every site once (rare: at most one event per hub and search)
            for (int64_t u0 = (int64_t)rt.me * kXcdBlock; u0 < n_round; u0 += T) {
                const int64_t u = u0 + threadIdx.x;
                bool want = false;
                if (u < n) {
                    const int lu = v.labels[u];
                    if (lu != v.alpha && ((ev >> lu) & 1ull) && v.f[u] > 0 && mf_hd_load32(&v.d[u]) == kMfInf && mf_hd_cas32(&v.d[u], kMfInf, k)) want = true;
                }
                xcd_stage(st, (int)u, want);
                xcd_flush_if_half(st, &c->fcount[k % 3], v.order + kbase);
            }
        }
        xcd_flush(st, &c->fcount[k % 3], v.order + kbase);
        if (HUBS) xcd_publish_hubs(v, st);
        if (!xcd_sync(st, rt)) return -1;
        if (HUBS && ev != 0 && xcd_load(&c->fcount[k % 3]) > 0) depth = k;   // (a level that only the hub pass filled)
    }
    *levels += k;
    XCD_PROF(rt, 3)
    return depth;
}

__device__ __forceinline__ void xcd_write_prof(XcdRt& rt)
{
    for (int k = 0; k < 5; ++k) rt.c->prof[k] = rt.t_prof[k];
    rt.c->prof[5] = rt.t_bar;
}

// ---- one global relabel (see the head of the file).  slot: the hub-height slot the sweeps read next (maxflow_driver.inl)
__global__ __launch_bounds__(kXcdBlock) void mf_k_xcd_search(MfView v, XcdCtl* c, int slot, int spp)
{
    __shared__ XcdStage st;
    XcdRt rt;
    if (!xcd_join(c, st, rt)) return;
    // a materialised alpha hub (never with the stranded-excess gate the product runs with) is not handled here
    const bool passive = v.has_alpha_hub[0] == 0;
    if (!passive) {
        if (rt.tid == 0) { c->out[3] = 3; c->out[4] = (int)rt.part; }
        return;
    }
    if (rt.tid == 0) mf_body_bfs_reset(v);
    int levels = 0;
    const int depth = xcd_bfs<true>(v, st, rt, spp, &levels);
    if (depth < 0) return;
    // (hub distances: published level by level - bfs_reset stored "unreached" before the first barrier)
    if (rt.tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the bodies below read with plain loads what other CUs wrote by atomics
        v.flags[0] = depth;
        mf_body_bfs_finish(v, slot, depth + 1 > 2 ? depth + 1 : 2);
    }
    // sites that hold excess and reach t (mf_body_count_active with loads that bypass the L1)
    int mine = 0;
    for (int64_t u = rt.tid; u < v.n; u += rt.T)
        mine += (v.labels[u] != v.alpha && mf_hd_load64(&v.ex[u]) > 0 && mf_hd_load32(&v.d[u]) != kMfInf) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine > 0) atomicAdd(&c->active, mine);
    if (!xcd_sync(st, rt)) return;
    if (rt.tid == 0) {
        const int act = xcd_load(&c->active);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        v.flags[3] = act;
        if (act > 0) v.flags[1] = 1;
        for (int k = 0; k < kMfFlags; ++k) c->flags[k] = v.flags[k];
        c->flags[kMfFlags] = v.cnt[v.alpha];
        c->out[0] = 1; c->out[1] = levels; c->out[3] = 1; c->out[4] = (int)rt.part;
        xcd_write_prof(rt);
    }
}

// ---- the later rounds of a hard move (see the head of the file).  max_rounds / sweeps / stall_min: the schedule of
// maxflow_driver.inl (sweeps_list, stall_sweeps); stamp0: first of the list stamps reserved for this launch (one per pass over a
// list); spp: frontier sites a workgroup expands per pass (8 lanes each), chosen by the host so that spp x max_degree appends fit
// half the staging buffer
__global__ __launch_bounds__(kXcdBlock) void mf_k_xcd_rounds(MfView v, XcdCtl* c, int max_rounds, int sweeps, int stall_min, int stamp0, int spp)
{
    __shared__ XcdStage st;
    XcdRt rt;
    if (!xcd_join(c, st, rt)) return;
    const int64_t n = v.n, T = rt.T;
    const int64_t n_round = (n + kXcdBlock - 1) / kXcdBlock * kXcdBlock;
    int q = 0;     // passes over a work list so far: list q is act[q & 1] with its size in lcnt[q % 3]; stamp0 + q marks its appends

    // ---- list 0: every site that holds excess
    for (int64_t u0 = (int64_t)rt.me * kXcdBlock; u0 < n_round; u0 += T) {
        const int64_t u = u0 + threadIdx.x;
        xcd_stage(st, (int)u, u < n && v.labels[u] != v.alpha && mf_hd_load64(&v.ex[u]) > 0);
        xcd_flush_if_half(st, &c->lcnt[0], v.act[0]);
    }
    xcd_flush(st, &c->lcnt[0], v.act[0]);
    if (!xcd_sync(st, rt)) return;
    if (rt.tid == 0) c->out[5] = xcd_load(&c->lcnt[0]);
    XCD_PROF(rt, 0)

    int rounds = 0, levels = 0, nsweeps = 0, status = 2;
    long long swept = 0;
    while (rounds < max_rounds) {
        ++rounds;
        const int depth = xcd_bfs<false>(v, st, rt, spp, &levels);
        if (depth < 0) return;
        // ---- passes over the work list: pass 0 of a round only drops the sites that no longer hold excess or reach t
        const int stall_limit = stall_min > depth + 2 ? stall_min : depth + 2;
        int stall = 0;
        bool none_left = false;
        for (int s = 0; s <= sweeps; ++s, ++q) {
            const int cnt = xcd_load(&c->lcnt[q % 3]);
            if (cnt == 0) { none_left = true; break; }   // every site that holds excess and may reach t is listed (its own step or the push that
                                                          // fed it claims it): an empty list means nothing that holds excess reaches t
            if (rt.tid == 0) { c->lcnt[(q + 2) % 3] = 0; c->moved[(q + 2) % 3] = 0; }
            const int c_round = (cnt + kXcdBlock - 1) / kXcdBlock * kXcdBlock;
            const int* in = v.act[q & 1];
            int* out = v.act[(q + 1) & 1];
            int* out_cnt = &c->lcnt[(q + 1) % 3];
            const int stamp = stamp0 + q;
            bool moved = false;
            if (s > 0) swept += cnt;
            for (int i0 = rt.me * kXcdBlock; i0 < c_round; i0 += (int)rt.part * kXcdBlock) {
                const int i = i0 + (int)threadIdx.x;
                if (i < cnt) {
                    const int u = xcd_load(&in[i]);
                    if (s == 0) {
                        if (mf_hd_load64(&v.ex[u]) > 0 && mf_hd_load32(&v.d[u]) != kMfInf && mf_hd_claim(&v.mark[u], stamp)) xcd_stage(st, u, true);
                    } else {
                        MfTailOut o;
                        mf_body_tail_step(v, u, &o);
                        moved |= o.moved;
                        if (o.listed && mf_hd_claim(&v.mark[u], stamp)) xcd_stage(st, u, true);
                        if (o.pushed_to >= 0 && mf_hd_claim(&v.mark[o.pushed_to], stamp)) xcd_stage(st, o.pushed_to, true);
                    }
                }
                xcd_flush_if_half(st, out_cnt, out);
            }
            if (moved) __hip_atomic_store(&c->moved[q % 3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xcd_flush(st, out_cnt, out);
            if (!xcd_sync(st, rt)) return;
            if (s > 0) {
                ++nsweeps;
                stall = xcd_load(&c->moved[q % 3]) != 0 ? 0 : stall + 1;
                if (stall >= stall_limit) { ++q; break; }
            }
        }
        XCD_PROF(rt, 4)
        if (none_left) { status = 1; break; }
    }
    if (rt.tid == 0) {
        c->out[0] = rounds; c->out[1] = levels; c->out[2] = nsweeps; c->out[3] = status; c->out[4] = (int)rt.part;
        c->out[6] = (int)(swept >> 4);
        xcd_write_prof(rt);
    }
}
#undef XCD_PROF

}  // namespace
}  // namespace pgx
