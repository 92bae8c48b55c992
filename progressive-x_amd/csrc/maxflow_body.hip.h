// maxflow_body.hip.h — per-site bodies of the alpha-expansion min-cut (lock-free push-relabel + BFS global relabel).
//
// Replaces: GCoptimizationGeneralGraph::expansion -> alpha_expansion -> Energy::minimize (BK max-flow) as driven by
//           pearl::PEARL::labeling, /root/reference/src/pyprogressivex/include/PEARL.h:507-551.  The GCoptimization
//           sources are absent from the reference snapshot (empty graph-cut-ransac submodule): semantics restated in
//           DESIGN.md §5.4 [U-5, U-6].
//
// The binary problem of one expansion move on label alpha (site p active iff label[p] != alpha;
// x_p = 0 "take alpha" = SOURCE side, x_p = 1 "keep" = SINK side):
//   t-links   keep_p = D[l_p][p] + sum_{q: l_q = alpha} w_pq + sum_{q active, l_q != l_p} w_pq / 2      (s -> p)
//             take_p = D[alpha][p]                                                                     (p -> t)
//   n-links   p <-> q, both active:  capacity w_pq (same label) or w_pq / 2 (different labels), both directions
//   label costs (Delong et al., IJCV 2012; one "hub" node per label):
//             beta in use, beta != alpha :  s -> y_beta (h),  y_beta -> p (inf) for every p with l_p = beta
//             alpha unused               :  p -> y_alpha (inf) for every site, y_alpha -> t (h)
//   The alpha hub is not materialised (gate = 1): a minimum cut either cuts y_alpha -> t (pay h, the sites are free) or
//   keeps y_alpha with t, which drags every site to the sink side (nobody switches).  So the move is solved WITHOUT the
//   hub and applied iff the excess left stranded at convergence (= source capacity - max flow = what switching gains)
//   is >= h; ties switch, as the minimal sink side demands (with exactly h stranded the hub is saturated by sites that
//   cannot reach t otherwise).  With the hub every site is adjacent to every other through it, and the moves that hand a
//   new instance its points (the label is unused until then) needed 10-14 global relabels of ~55 levels each.
// All capacities are int64 multiples of 2^-32, so the maximum flow is exact and the minimal sink side
// {v : v reaches t in the residual graph} is unique => labels are bit-identical to the CPU oracle's Dinic solver and to
// BK's what_segment(default = SOURCE), independent of push order, atomics and scheduling.
//
// Only the max-PREFLOW phase is run: when no site/hub with excess can reach t, the reverse BFS from t already yields
// the minimal sink side (returning stranded excess to s never touches nodes that reach t).
//
// The bodies are host/device so that tests/emu can run the identical algorithm sequentially on the CPU (test
// infrastructure for the host logic; the product path always runs the HIP kernels in maxflow.hip).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PGX_HD __host__ __device__ __forceinline__
#else
#define PGX_HD inline
#endif

namespace pgx {

constexpr int kMfInf = 0x3fffffff;
constexpr int kMfDead = -1;  // height of a site that is already alpha (not part of the move's graph)
constexpr int kMfFlags = 12;  // ints in MfView::flags

struct MfView {
    int64_t n;
    int L;
    int alpha;
    long long lambda_q, h_q;
    const long long* dq;  // [L][n] label-major unary table
    int* labels;          // [n]
    const int* off;       // [n+1]   (nullptr => no pairwise term)
    const int* idx;       // [E]
    const int* mult;      // [E]
    const long long* wq;  // [E] per-arc weight (symmetric: wq[a] == wq[rev[a]]) replacing lambda_q * mult[a], or nullptr
    const int* rev;       // [E] index of the reverse arc
    long long* cap;       // [E] residual capacity of arc a (row owner -> idx[a])
    long long* tot;       // [E] cap[a] + cap[rev[a]], invariant under pushes: the BFS reads the reverse residual as
                          //     tot[a] - cap[a], two sequential reads instead of a gather through rev[a]
    long long* ex;        // [n] excess
    long long* rt;        // [n] residual capacity site -> t
    int* d;               // [n] height / BFS distance to t
    long long* f;         // [n] flow received from the site's beta hub (residual of p -> y_beta)
    long long* g;         // [n] flow sent into the alpha hub (residual of y_alpha -> p)
    // hubs
    int* cnt;                      // [L] label histogram
    int* hub_exists;               // [L] beta hub present: 1, or 2 once a member pulled from it (only then can a member hold f > 0)
    long long* hub_e;              // [L] beta hub excess
    int* has_alpha_hub;            // [1]
    long long* hubA_rt;            // [1] residual y_alpha -> t
    long long* hubA_e;             // [1] excess held by y_alpha
    long long* hubA_want;          // [3] rotating: what the alpha hub's members asked for in a sweep (a hint for pushers)
    int* bfs_hub_d;                // [L] BFS distance of beta hubs
    int* bfs_hubA_d;               // [1]
    int* hub_min;                  // [3][L] rotating: min member height of each beta hub
    unsigned long long* hubA_min;  // [3] rotating: (height << 32 | site) of the lowest member with g > 0
    int* order;                    // [n] sites in BFS order: level k occupies order[lvl[k] .. lvl[k] + fcount[k % 3])
    int* lvl;                      // [hmax + 144] start of each level in `order` (lvl[k + 1] is written by level k + 1)
    int* fcount;                   // [3] level sizes, rotating by level % 3
    int* act[2];                   // [n] each: work lists of the list-mode sweeps (read one, write the other)
    int* acnt;                     // [2] their sizes
    int* mark;                     // [n] stamp of the last list a site was appended to (stamps only grow)
    int* flags;                    // [kMfFlags]: 8 flow reached t in the sweep being run, 11 sweeps since flow last reached t (epilogue);
                                   //      0 last BFS level that labelled a site, 1 work-left (boolean, being
                                   //      accumulated), 2 sites relabelled by apply, 4 work-left of the last finished sweep,
                                   //      6 a list-mode sweep pushed into a beta hub (all members must take part again)
    long long* swept;              // [1] sum of the work-list lengths swept so far (never cleared: pgx_expansion_schedule), or nullptr
    int hmax;                      // heights >= hmax are treated as unreachable
    int gate;                      // 1: an unused alpha is handled by the stuck-excess test instead of a hub (see below)
};

// ---- atomics ---------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void mf_add64(long long* p, long long v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
__device__ __forceinline__ long long mf_load64(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool mf_cas64(long long* p, long long& expected, long long desired)
{
    const unsigned long long old = atomicCAS((unsigned long long*)p, (unsigned long long)expected, (unsigned long long)desired);
    const bool ok = old == (unsigned long long)expected;
    expected = (long long)old;
    return ok;
}
__device__ __forceinline__ void mf_min32(int* p, int v) { atomicMin(p, v); }
__device__ __forceinline__ bool mf_cas32(int* p, int expected, int desired) { return atomicCAS(p, expected, desired) == expected; }
// wave-aggregated append: every ACTIVE lane calls it; lanes with want == true get consecutive slots from one atomic
__device__ __forceinline__ void mf_append(int* counter, int* list, int value, bool want)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}
__device__ __forceinline__ void mf_minu64(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
__device__ __forceinline__ void mf_add32(int* p, int v) { atomicAdd(p, v); }
__device__ __forceinline__ int mf_load32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mf_store32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
inline void mf_add64(long long* p, long long v) { *p += v; }
inline long long mf_load64(const long long* p) { return *p; }
inline bool mf_cas64(long long* p, long long& expected, long long desired)
{
    if (*p == expected) { *p = desired; return true; }
    expected = *p;
    return false;
}
inline void mf_min32(int* p, int v) { if (v < *p) *p = v; }
inline bool mf_cas32(int* p, int expected, int desired)
{
    if (*p == expected) { *p = desired; return true; }
    return false;
}
inline void mf_append(int* counter, int* list, int value, bool want)
{
    if (want) list[(*counter)++] = value;
}
inline void mf_minu64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
inline void mf_add32(int* p, int v) { *p += v; }
inline int mf_load32(const int* p) { return *p; }
inline void mf_store32(int* p, int v) { *p = v; }
#endif

// One lane per wave among those with `cond` (all lanes on the host).  Used to thin out contenders on the few hub budget
// words: without it every member of a hub CASes the same address in the same sweep (measured: ~2e5 serialised L2
// atomics per hub per sweep at N = 2e5); the budget is small, so a few winners per sweep drain it just as fast.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool mf_elect(bool cond)
{
    const unsigned long long mask = __ballot(cond);
    return cond && (__lane_id() == __ffsll((long long)mask) - 1);
}
#else
inline bool mf_elect(bool cond) { return cond; }
#endif

// Take up to `want` out of a shared budget, wait-free: one fetch-add, plus one refund when the budget ran short.  The
// word may be transiently negative (readers treat <= 0 as empty); `old` can only UNDER-state what is available, so the
// sum of grants never exceeds the budget.  (A CAS loop here cost O(k^2) retries with k ~ 3000 contenders per hub word:
// the sweep kernel went from ~30 us to ~250 us whenever a large label cost was being drained.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ long long mf_fetch_add64(long long* p, long long v)
{
    return (long long)atomicAdd((unsigned long long*)p, (unsigned long long)v);
}
#else
inline long long mf_fetch_add64(long long* p, long long v) { const long long o = *p; *p += v; return o; }
#endif

PGX_HD long long mf_reserve(long long* budget, long long want)
{
    if (want <= 0 || mf_load64(budget) <= 0) return 0;
    const long long old = mf_fetch_add64(budget, -want);
    if (old >= want) return want;
    const long long got = old > 0 ? old : 0;
    mf_fetch_add64(budget, want - got);
    return got;
}

PGX_HD unsigned long long mf_pack(int h, int site) { return ((unsigned long long)(unsigned)h << 32) | (unsigned)site; }

// ---- per-move setup -----------------------------------------------------------------------------------------------
PGX_HD void mf_body_count(const MfView& v, int64_t u) { mf_add32(&v.cnt[v.labels[u]], 1); }

// one thread: decide which hubs exist (after mf_body_count ran for every site)
PGX_HD void mf_body_hub_setup(const MfView& v)
{
    const bool lc = v.h_q > 0;
    for (int l = 0; l < v.L; ++l) {
        const bool ex = lc && l != v.alpha && v.cnt[l] > 0;
        v.hub_exists[l] = ex ? 1 : 0;
        v.hub_e[l] = ex ? v.h_q : 0;
        v.bfs_hub_d[l] = kMfInf;
        for (int r = 0; r < 3; ++r) v.hub_min[r * v.L + l] = kMfInf;
    }
    const bool ha = lc && v.cnt[v.alpha] == 0 && !v.gate;
    v.has_alpha_hub[0] = ha ? 1 : 0;
    v.hubA_rt[0] = ha ? v.h_q : 0;
    v.hubA_e[0] = 0;
    v.hubA_want[0] = v.hubA_want[1] = v.hubA_want[2] = 0;
    v.bfs_hubA_d[0] = kMfInf;
    for (int r = 0; r < 3; ++r) v.hubA_min[r] = ~0ull;
    for (int k = 0; k < kMfFlags; ++k) v.flags[k] = 0;
}

PGX_HD void mf_body_init_site(const MfView& v, int64_t u)
{
    const int lu = v.labels[u];
    v.f[u] = 0;
    v.g[u] = 0;
    v.d[u] = kMfInf;
    if (lu == v.alpha) {  // inactive: already alpha
        v.d[u] = kMfDead;  // never "unlabelled": the BFS tests d alone and skips the label gather
        v.ex[u] = 0;
        v.rt[u] = 0;
        if (v.off)
            for (int a = v.off[u]; a < v.off[u + 1]; ++a) { v.cap[a] = 0; v.tot[a] = 0; }
        return;
    }
    long long keep = v.dq[(int64_t)lu * v.n + u];
    const long long take = v.dq[(int64_t)v.alpha * v.n + u];
    if (v.off && v.lambda_q > 0)
        for (int a = v.off[u]; a < v.off[u + 1]; ++a) {
            const int q = v.idx[a];
            const int lq = v.labels[q];
            const long long w = v.wq ? v.wq[a] : v.lambda_q * (long long)v.mult[a];
            if (lq == v.alpha) { keep += w; v.cap[a] = 0; v.tot[a] = 0; }
            else if (lq == lu) { v.cap[a] = w; v.tot[a] = 2 * w; }
            else { keep += w / 2; v.cap[a] = w / 2; v.tot[a] = w; }  // the reverse arc gets w / 2 as well
        }
    if (keep > take) { v.ex[u] = keep - take; v.rt[u] = 0; }
    else { v.ex[u] = 0; v.rt[u] = take - keep; }
}

// ---- global relabel: level-synchronous reverse BFS from t --------------------------------------------------------
// hub_acc: where per-label minima are accumulated — the global array itself in the sequential emulation, a per-block
// LDS array on the device (flushed with one global atomic per label per block; a direct global atomicMin per site
// serialises ~N atomics on L addresses and was measured to dominate the kernels).
PGX_HD void mf_acc_min(int* acc, int val)
{
    if (val < mf_load32(acc)) mf_min32(acc, val);
}

// Where level k starts in `order`: behind level k-1, whose size is final when level k runs (uniform plain reads).
PGX_HD int mf_level_base(const MfView& v, int k) { return k <= 1 ? 0 : v.lvl[k - 1] + v.fcount[(k - 1) % 3]; }

// Labels site u with BFS distance k (if still unlabelled), records what that implies for the hubs and appends u to the
// frontier of level k.  Must be called convergently by all active lanes (wave-aggregated append); `want` selects lanes.
// base: start of level k in `order`, or -1 to look it up (mf_level_base; only valid across kernel boundaries).
// stage_cnt / stage_list: when given, the site is appended to that (workgroup-local) list instead and the caller moves the
// list to `order` with ONE atomic on the level counter per flush: a wave-level append per labelled site group was ~20 ns
// of serialised L2 atomics each on a single address — 331 us for the level-1 pass and most of the ~42 us per level at
// N = 1e6.
PGX_HD bool mf_bfs_label(const MfView& v, int64_t u, int k, int* hub_acc, bool want, int base = -1,
                         int* stage_cnt = nullptr, int* stage_list = nullptr, bool hubs = true, bool preclaimed = false)
{
    bool mine = false;
    if (want) mine = preclaimed ? true : mf_cas32(&v.d[u], kMfInf, k);   // preclaimed: the caller has written d[u] = k itself
    if (mine && hubs) {   // hubs == false: the caller knows that no hub exists in this move (saves the label gather)
        const int lu = v.labels[u];
        if (v.hub_exists[lu]) mf_acc_min(&hub_acc[lu], k + 1);      // y_beta -> u has infinite capacity
        if (v.has_alpha_hub[0] && mf_load64(&v.g[u]) > 0) {          // y_alpha -> u has residual g[u]
            mf_minu64(&v.hubA_min[0], mf_pack(k, (int)u));           // slot 0 is the BFS result slot
            mf_min32(&v.bfs_hubA_d[0], k + 1);
        }
    }
    if (stage_cnt) mf_append(stage_cnt, stage_list, (int)u, mine);
    else mf_append(&v.fcount[k % 3], v.order + (base >= 0 ? base : mf_level_base(v, k)), (int)u, mine);
    return mine;
}

// one thread, before level 1
PGX_HD void mf_body_bfs_reset(const MfView& v)
{
    for (int l = 0; l < v.L; ++l) v.bfs_hub_d[l] = kMfInf;
    v.bfs_hubA_d[0] = (v.has_alpha_hub[0] && v.hubA_rt[0] > 0) ? 1 : kMfInf;
    v.hubA_min[0] = ~0ull;
    v.fcount[0] = v.fcount[1] = v.fcount[2] = 0;
    v.lvl[0] = v.lvl[1] = 0;
    v.flags[0] = 0;
    v.flags[1] = 0;
    v.flags[3] = 0;
    v.flags[4] = 1;  // "work left": the sweeps of this round run until an epilogue clears it (mf_sweep_idle)
    v.flags[8] = 0;
    v.flags[11] = 0;
    v.flags[6] = 0;
    v.flags[7] = 0;
    v.acnt[0] = v.acnt[1] = 0;
}

// The sweeps of a round are enqueued in batches between two flag read-backs.  Once an epilogue has latched "no work left"
// (no site holds excess that reaches t, no hub can deliver) every later sweep of the round would change nothing: sweep
// kernels and epilogues leave at once (a plain read: flags[4] was written by an earlier kernel).  Most rounds of the
// steady-state moves finish within 2-3 sweeps of the 8 a batch issues.
PGX_HD bool mf_sweep_idle(const MfView& v) { return v.flags[4] == 0; }

// level 1: sites with residual capacity to t.  Returns true iff the site was labelled.
PGX_HD bool mf_body_bfs_init(const MfView& v, int64_t u, int* hub_acc, int* stage_cnt = nullptr, int* stage_list = nullptr)
{
    // nobody else touches d[u] in this pass: one store of the final value instead of "unlabelled", then a compare-and-swap
    const bool active = v.labels[u] != v.alpha;
    const bool first = active && v.rt[u] > 0;
    if (active) mf_store32(&v.d[u], first ? 1 : kMfInf);
    return mf_bfs_label(v, u, 1, hub_acc, first, -1, stage_cnt, stage_list, true, true);
}

// level k, frontier part: site w was labelled k-1; every active unlabelled neighbour u with residual u -> w gets k
PGX_HD bool mf_body_bfs_expand(const MfView& v, int64_t w, int k, int* hub_acc)
{
    bool any = false;
    if (!v.off) return false;
    for (int a = v.off[w]; a < v.off[w + 1]; ++a) {
        const int u = v.idx[a];
        const bool want = v.tot[a] - mf_load64(&v.cap[a]) > 0 && mf_load32(&v.d[u]) == kMfInf;  // residual u -> w; inactive sites have d = kMfDead
        any |= mf_bfs_label(v, u, k, hub_acc, want);
    }
    return any;
}

// level k, hub part (only run when some hub received distance k-1): u -> y_alpha (inf) / u -> y_beta (residual f[u])
PGX_HD bool mf_body_bfs_hubpass(const MfView& v, int64_t u, int k, bool alpha_event, int* hub_acc,
                                int* stage_cnt = nullptr, int* stage_list = nullptr)
{
    const int lu = v.labels[u];
    bool want = false;
    if (lu != v.alpha && mf_load32(&v.d[u]) == kMfInf)
        want = alpha_event || (v.hub_exists[lu] && v.f[u] > 0 && mf_load32(&v.bfs_hub_d[lu]) == k - 1);
    return mf_bfs_label(v, u, k, hub_acc, want, -1, stage_cnt, stage_list);
}

// uniform per level: which hub events fire at level k (bit 0: alpha hub, bit 1: some beta hub)
// Plain (cached) reads on purpose: every thread of every level evaluates this, and L2-scope loads of the same few
// words from 2e5 threads measured ~30 us per level.  A hub's distance k-1 was written by an EARLIER kernel (level
// k-2's flush writes k-1); writes racing in THIS kernel store k+1, so a stale read cannot fake or hide an event.
PGX_HD int mf_bfs_hub_events(const MfView& v, int k)
{
    int ev = 0;
    if (v.has_alpha_hub[0] && v.bfs_hubA_d[0] == k - 1) ev |= 1;
    for (int l = 0; l < v.L; ++l)
        if (v.hub_exists[l] == 2 && v.bfs_hub_d[l] == k - 1) ev |= 2;  // == 2: some member holds f > 0 (else the pass over all sites labels nobody)
    return ev;
}

// one thread, after the BFS: publish hub heights for the sweeps (slot `slot`) and count active hubs
PGX_HD void mf_body_bfs_finish(const MfView& v, int slot, int last_level)
{
    v.flags[3] = 0;   // the count of active sites follows (it is redone when the search turns out to need more levels)
    v.lvl[last_level + 1] = mf_level_base(v, last_level + 1);  // closes the level table for the wave pass
    for (int l = 0; l < v.L; ++l) {
        const int hd = v.bfs_hub_d[l];
        for (int r = 0; r < 3; ++r) v.hub_min[r * v.L + l] = kMfInf;
        v.hub_min[slot * v.L + l] = hd == kMfInf ? kMfInf : hd - 1;
        if (v.hub_exists[l] && v.hub_e[l] > 0 && hd != kMfInf) { v.flags[1] = 1; v.flags[7] = 1; }  // [7]: a hub can still deliver
    }
    const unsigned long long pk = v.hubA_min[0];
    for (int r = 0; r < 3; ++r) { v.hubA_min[r] = ~0ull; v.hubA_want[r] = 0; }
    v.hubA_min[slot] = pk;
    if (v.has_alpha_hub[0] && v.hubA_e[0] > 0 && v.bfs_hubA_d[0] != kMfInf) v.flags[1] = 1;  // the hub can still deliver
}

// After a global relabel: does site u hold excess that can reach t?
PGX_HD bool mf_body_count_active(const MfView& v, int64_t u)
{
    // plain loads, issued together: everything read here was written by earlier kernels
    const int lu = v.labels[u], du = v.d[u];
    const long long e = v.ex[u];
    return lu != v.alpha && e > 0 && du != kMfInf;
}

// ---- work lists ---------------------------------------------------------------------------------------------------
// A site belongs on the list while it can still act: it holds excess and reaches t, or it lent flow to the alpha hub
// (its height then takes part in the hub's height).  Appending is idempotent per list through the stamp in mark[].
PGX_HD bool mf_listed(const MfView& v, int64_t u)
{
    if (v.labels[u] == v.alpha || mf_load32(&v.d[u]) == kMfInf) return false;
    return mf_load64(&v.ex[u]) > 0 || (v.has_alpha_hub[0] && mf_load64(&v.g[u]) > 0);
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool mf_claim(int* mark, int stamp) { return atomicExch(mark, stamp) != stamp; }
#else
inline bool mf_claim(int* mark, int stamp) { const bool fresh = *mark != stamp; *mark = stamp; return fresh; }
#endif

// (host/device wrappers: kernels may only call PGX_HD functions, the two variants above are picked inside them)
PGX_HD bool mf_list_claim(const MfView& v, int site, int stamp) { return mf_claim(&v.mark[site], stamp); }
PGX_HD void mf_list_append(const MfView& v, int which, int site, bool want) { mf_append(&v.acnt[which], v.act[which], site, want); }

// ---- wave pass --------------------------------------------------------------------------------------------------
// Right after a global relabel the heights are exact distances and `order` lists the sites level by level.  Visiting
// the levels from the farthest to the nearest and letting every site push along ALL its arcs into the level below moves
// excess over the whole length of a shortest path in one pass (a lock-step sweep moves it one hop: with paths of 100+
// arcs, as in the moves that hand a new instance its ~5e4 points at N = 1e6, a round of 48 sweeps never reached t and a
// move took up to 239 global relabels).  Only n-links and t-links; hub arcs are left to the sweeps.
PGX_HD void mf_body_wave(const MfView& v, int64_t u, int k)
{
    long long e = mf_load64(&v.ex[u]);
    if (e <= 0) return;
    const long long r = v.rt[u];
    if (r > 0) {
        const long long dl = e < r ? e : r;
        v.rt[u] = r - dl;
        mf_add64(&v.ex[u], -dl);
        e -= dl;
    }
    if (e <= 0 || !v.off) return;
    for (int a = v.off[u]; a < v.off[u + 1] && e > 0; ++a) {
        const long long c = mf_load64(&v.cap[a]);
        if (c <= 0) continue;
        const int w = v.idx[a];
        if (mf_load32(&v.d[w]) != k - 1) continue;
        const long long dl = e < c ? e : c;
        mf_add64(&v.cap[a], -dl);
        mf_add64(&v.cap[v.rev[a]], dl);
        mf_add64(&v.ex[u], -dl);
        mf_add64(&v.ex[w], dl);
        e -= dl;
    }
}

// ---- hub pulls -----------------------------------------------------------------------------------------------------
// Flow leaves a hub by being PULLED: the arc hub -> u is admissible when the hub sits one above u, and what u asks for
// is what it can pass on at once (its own residual to t, or else the residual of its admissible arcs).
//  * beta hub (s -> y_beta -> members): measured at N = 1e6, h = 5000: when one elected member per wave pulled and
//    members without a t residual asked for "everything visible", a single stale-height member hoarded the whole label
//    cost and dribbled it back over several global relabels (13-19 ms per steady-state move, ~16 deliveries per sweep).
//  * alpha hub (sites -> y_alpha -> t, and back out through the members that lent it flow, g > 0): pushing "through"
//    the saturated hub into its single lowest member moved one member's g per sweep, ~3 units of flow per global
//    relabel in the moves that hand a new instance its ~5e4 points (75-240 relabels per move).  The hub now holds
//    excess like any node and all members one below it pull concurrently.
// A member that is eligible but can pass nothing on has a stale height: it is lifted instead (relabelling a node without
// an admissible residual arc is valid whether or not it holds excess), which lets the hub rise to members that deliver.
// The requests of a workgroup are summed per hub in LDS and reserved with one atomic per (workgroup, hub); the grant is
// split in LDS arrival order (maxflow.hip mf_sweep_step).  The sequential emulation reserves per site.
PGX_HD int mf_hubA_height(const MfView& v, int prev)
{
    if (v.hubA_rt[0] > 0) return 1;  // plain read (gate)
    const unsigned long long pk = v.hubA_min[prev];
    return pk == ~0ull ? kMfInf : (int)(pk >> 32) + 1;
}

PGX_HD long long mf_passable(const MfView& v, int64_t u, int du, int prev, long long e)
{
    if (v.rt[u] > 0) return v.rt[u] > e ? v.rt[u] - e : 0;
    if (e > 0) return 0;  // already holds excess it has not placed yet
    const int lu = v.labels[u];
    long long adm = 0;
    int best_h = kMfInf;
    if (v.hub_exists[lu] && v.f[u] > 0) {  // residual u -> y_beta
        const int m = v.hub_min[prev * v.L + lu];
        if (m != kMfInf) best_h = m + 1;
        if (best_h < du) adm += v.f[u];
    }
    if (v.has_alpha_hub[0]) {             // u -> y_alpha (inf)
        const int ha = mf_hubA_height(v, prev);
        if (ha < best_h) best_h = ha;
        const long long room = v.hubA_rt[0] > 0 ? v.hubA_rt[0] : v.hubA_want[prev];
        if (ha < du && room > 0) adm += room;
    }
    if (v.off) {
        const int end = v.off[u + 1];
        for (int a0 = v.off[u]; a0 < end; a0 += 8) {  // batched loads, see mf_body_sweep
            long long c[8];
            int w[8], h[8];
            for (int j = 0; j < 8; ++j) {
                const bool in = a0 + j < end;
                c[j] = in ? mf_load64(&v.cap[a0 + j]) : 0;
                w[j] = in ? v.idx[a0 + j] : 0;
            }
            for (int j = 0; j < 8; ++j) h[j] = c[j] > 0 ? mf_load32(&v.d[w[j]]) : kMfInf;
            for (int j = 0; j < 8; ++j) {
                if (h[j] < best_h) best_h = h[j];
                if (h[j] < du) adm += c[j];
            }
        }
    }
    if (adm > 0) return adm;
    int nd = best_h == kMfInf ? kMfInf : best_h + 1;
    if (nd >= v.hmax) nd = kMfInf;
    if (nd > du) mf_store32(&v.d[u], nd);
    return 0;
}

// *which: 0 nothing, 1 pull from the site's beta hub, 2 pull from the alpha hub
PGX_HD long long mf_body_pull_want(const MfView& v, int64_t u, int prev, int* which)
{
    *which = 0;
    const int lu = v.labels[u];
    if (lu == v.alpha) return 0;
    const bool b = v.hub_exists[lu] && v.hub_e[lu] > 0;                           // plain (cached) reads: gates only
    const bool a = v.has_alpha_hub[0] && v.hubA_e[0] > 0 && v.hubA_rt[0] <= 0;
    if (!a && !b) return 0;
    const int du = v.d[u];
    if (du == kMfInf) return 0;
    int from = 0;
    long long cap = 0;
    if (b && v.hub_min[prev * v.L + lu] == du) from = 1;  // hub height m + 1 must be exactly one above u
    else if (a) {
        const unsigned long long pk = v.hubA_min[prev];
        cap = mf_load64(&v.g[u]);
        if (pk != ~0ull && (int)(pk >> 32) == du && cap > 0) from = 2;
    }
    if (from == 0) return 0;
    long long want = mf_passable(v, u, du, prev, mf_load64(&v.ex[u]));
    if (from == 2 && want > cap) want = cap;  // y_alpha -> u has residual g[u]
    if (want > 0) *which = from;
    return want;
}

// ---- one push-relabel step for site u ------------------------------------------------------------------------------
// prev/cur/next: rotating slots of the hub height scans (read prev, accumulate cur, clear next).
// returns true iff this site did or still has work (the caller latches flags[1])
// list_mode: only the sites on a work list are visited (maxflow_driver.inl), so beta-hub height scans are skipped (the
// epilogue carries the published heights forward) and *pushed_to names the site that received flow, if any.
struct MfSweepIo {
    long long granted = 0;   // in: flow granted by the pull phase ...
    int which = 0;           //     ... from the beta hub (1) or the alpha hub (2)
    bool list_mode = false;  // in
    int pushed_to = -1;      // out: site that received flow along an n-link
    long long pushedA = 0;   // out: flow pushed into the alpha hub (the caller adds it to hubA_e)
    bool moved = false;      // out: this site delivered flow to t (its own t-link or the alpha hub's)
    bool listed = false;     // out: the site still belongs on the work list (mf_listed with the values this step ended on)
};

PGX_HD bool mf_body_sweep(const MfView& v, int64_t u, int prev, int cur, int* hub_acc, MfSweepIo* io)
{
    const bool list_mode = io->list_mode;
    // Everything that depends on u alone is requested up front - ONE round trip instead of a chain of six (label -> excess ->
    // height -> t-link -> row start -> row end: measured 7.6 us per pass of a work-list sweep, as much again as its list
    // maintenance) - and the excess is tracked locally from here on: what a neighbour pushes to u meanwhile is that
    // neighbour's "work" and puts u on the next list through `pushed_to`, exactly as when it arrived just after the old reload.
    const int lu = v.labels[u];
    long long e = mf_load64(&v.ex[u]);
    int du = v.d[u];
    const long long rtu = v.rt[u];
    const int a_lo = v.off ? v.off[u] : 0, a_hi = v.off ? v.off[u + 1] : 0;
    if (lu == v.alpha) return false;
    bool work = false;
    const bool hub_b = v.hub_exists[lu] != 0;
    const bool hub_a = v.has_alpha_hub[0] != 0;
    // A hub's height is rescanned only while it holds excess (members pull); otherwise the last known height is carried
    // forward by the epilogue (heights only grow, so a stale value is a valid lower bound: a member may still push back
    // into the hub, which then holds excess and is rescanned).  Sites without excess and without hub business are done.
    if (io->granted > 0) {  // hub -> u (mf_body_pull_want)
        if (io->which == 1) {
            v.f[u] += io->granted;
            if (v.hub_exists[lu] != 2) mf_store32(&v.hub_exists[lu], 2);  // the BFS runs this hub's member pass from now on
        }
        else mf_add64(&v.g[u], -io->granted);
        mf_add64(&v.ex[u], io->granted);
        e += io->granted;
        work = true;
    }
    const bool scan_b = !list_mode && hub_b && v.hub_e[lu] > 0;  // plain (cached) read: a gate, not a synchronisation
    if (!scan_b && e <= 0 && !(hub_a && mf_load64(&v.g[u]) > 0)) return false;
    // hub heights as published by the previous scan
    int hb = kMfInf;
    if (hub_b) { const int m = v.hub_min[prev * v.L + lu]; hb = m == kMfInf ? kMfInf : m + 1; }
    const int ha = hub_a ? mf_hubA_height(v, prev) : kMfInf;
    if (du != kMfInf && e > 0) {
        if (rtu > 0) {  // u -> t
            const long long dl = e < rtu ? e : rtu;
            v.rt[u] = rtu - dl;
            mf_add64(&v.ex[u], -dl);
            e -= dl;
            io->moved = true;
        }
        if (e > 0) {
            int best_h = kMfInf, best_a = -1, kind = 0;  // kind 1 n-link, 2 alpha hub, 3 beta hub
            long long best_c = 0;
            // lowest residual neighbour.  Loads are issued in batches of eight arcs (capacities and heads, then the heads'
            // heights): one arc at a time is a chain of ~2 memory round trips per arc, which is what a sweep over a short
            // work list spends its time on.
            for (int a0 = a_lo; a0 < a_hi; a0 += 8) {
                long long c[8];
                int w[8], h[8];
                for (int j = 0; j < 8; ++j) {
                    const bool in = a0 + j < a_hi;
                    c[j] = in ? mf_load64(&v.cap[a0 + j]) : 0;
                    w[j] = in ? v.idx[a0 + j] : 0;
                }
                for (int j = 0; j < 8; ++j) h[j] = c[j] > 0 ? mf_load32(&v.d[w[j]]) : kMfInf;
                for (int j = 0; j < 8; ++j)
                    if (h[j] < best_h) { best_h = h[j]; best_a = a0 + j; best_c = c[j]; kind = 1; }
            }
            if (hub_a && ha < best_h) { best_h = ha; kind = 2; }
            if (hub_b && v.f[u] > 0 && hb < best_h) { best_h = hb; kind = 3; }
            if (kind == 0 || best_h == kMfInf) {
                du = kMfInf;  // no residual arc leads anywhere that reaches t
                mf_store32(&v.d[u], du);
            } else if (du > best_h) {
                if (kind == 1) {
                    // best_c as loaded above: only this site lowers the capacity of its own arcs, so the value can only have grown
                    const long long dl = e < best_c ? e : best_c;
                    const int to = v.idx[best_a];
                    mf_add64(&v.cap[best_a], -dl);
                    mf_add64(&v.cap[v.rev[best_a]], dl);
                    mf_add64(&v.ex[u], -dl);
                    mf_add64(&v.ex[to], dl);
                    e -= dl;
                    io->pushed_to = to;
                    work = true;
                } else if (kind == 2) {
                    if (ha == 1 && v.hubA_rt[0] > 0) {  // budget y_alpha -> t still open: one contender per wave
                        if (!mf_elect(true)) work = true;
                        else {
                            const long long got = mf_reserve(v.hubA_rt, e);
                            if (got > 0) { mf_add64(&v.g[u], got); mf_add64(&v.ex[u], -got); e -= got; io->moved = true; }
                            work = true;
                        }
                    } else {                            // into the hub; members one below it pull it out again
                        mf_add64(&v.g[u], e);
                        mf_add64(&v.ex[u], -e);
                        io->pushedA += e;
                        e = 0;
                        work = true;
                    }
                } else {  // back into the beta hub
                    const long long dl = e < v.f[u] ? e : v.f[u];
                    v.f[u] -= dl;
                    mf_add64(&v.ex[u], -dl);
                    mf_add64(&v.hub_e[lu], dl);
                    e -= dl;
                    if (list_mode) mf_store32(&v.flags[6], 1);
                    work = true;
                }
            } else {
                du = best_h + 1;
                if (du >= v.hmax) du = kMfInf;
                mf_store32(&v.d[u], du);
            }
        }
    }
    // contribute to the next hub height scan with the final height
    if (scan_b && du != kMfInf) mf_acc_min(&hub_acc[lu], du);
    bool lent = false;
    if (hub_a && du != kMfInf) {
        lent = mf_load64(&v.g[u]) > 0;
        if (lent) {
            const unsigned long long pk = mf_pack(du, (int)u);
            if (pk < v.hubA_min[cur]) mf_minu64(&v.hubA_min[cur], pk);
        }
    }
    if (du != kMfInf && e > 0) work = true;
    io->listed = du != kMfInf && (e > 0 || lent);
    return work;
}

// one thread per sweep: clear the slot the NEXT sweep will accumulate into; latch the work-left flag
// consumed: parity of the work list the sweep just read (list mode), -1 for a sweep over all sites
PGX_HD void mf_body_sweep_epilogue(const MfView& v, int cur, int next, int consumed = -1)
{
    int act = v.flags[1];
    const int prev = (cur + 2) % 3;
    for (int l = 0; l < v.L; ++l) {
        // no scan was requested for this hub in this sweep: keep its last known height
        if (v.hub_exists[l] && (consumed >= 0 || v.hub_e[l] <= 0) && v.hub_min[cur * v.L + l] == kMfInf)
            v.hub_min[cur * v.L + l] = v.hub_min[prev * v.L + l];
        v.hub_min[next * v.L + l] = kMfInf;
        if (v.hub_exists[l] && v.hub_e[l] > 0 && v.hub_min[cur * v.L + l] != kMfInf) act = 1;
    }
    if (v.has_alpha_hub[0] && v.hubA_e[0] > 0 && v.hubA_min[cur] != ~0ull) act = 1;
    v.hubA_min[next] = ~0ull;
    v.hubA_want[next] = 0;
    v.flags[4] = act;
    v.flags[1] = 0;
    v.flags[11] = v.flags[8] ? 0 : v.flags[11] + 1;   // sweeps in a row without any flow reaching t (the driver then searches again)
    v.flags[8] = 0;
    if (consumed >= 0) {
        if (v.swept) v.swept[0] += v.acnt[consumed];   // (one thread per sweep: what a list sweep actually visited, for the labelling roofline)
        v.acnt[consumed] = 0;
    }
}

// excess that cannot reach t any more (valid after the last global relabel): the gain of the move
PGX_HD long long mf_body_stuck_excess(const MfView& v, int64_t u)
{
    if (v.labels[u] == v.alpha || v.d[u] != kMfInf) return 0;
    const long long e = mf_load64(&v.ex[u]);
    return e > 0 ? e : 0;
}

// ---- hub-free rounds inside ONE launch on ONE XCD (maxflow_xcd.hip.h; round 6) -------------------------------------------
// The rounds of a hard move after the first (the excess that escaped its cluster - a hundred units on a few hundred
// sites - squeezing through a sparse sea in Dinic-like phases) touch frontiers and work lists of a few thousand sites: every level
// and every sweep is a chain of dependent accesses, and a launch per step costs 12-18 us for ~2 us of work.  One persistent launch
// whose workgroups all sit on the same XCD shares that XCD's L2 as a coherent point: plain stores (write-through L1) + loads that
// bypass the L1 (sc1) need NO fence, and a barrier over its 32 workgroups costs 0.75-1.0 us (scripts/micro/xcd_scope_bench.hip) against
// 4.2-5.8 with the release / acquire pair round 4 measured.  The step below is mf_body_sweep without hubs (n-links and t-links
// only): ignoring the hub arcs is valid - a subset of the residual arcs - and convergence is only ever declared by the ordinary
// search WITH hubs that follows (maxflow_driver.inl), so these rounds can never change a result, only leave less to do.
// Every load of mutable state bypasses the L1; rt[u] and d[u] are written by the site's own step only (plain stores).
PGX_HD void mf_plain_store64(long long* p, long long v) { *p = v; }
PGX_HD void mf_plain_store32(int* p, int v) { *p = v; }

// (host/device wrappers for the kernel of maxflow_xcd.hip.h: kernels may only call PGX_HD functions)
PGX_HD long long mf_hd_load64(const long long* p) { return mf_load64(p); }
PGX_HD int mf_hd_load32(const int* p) { return mf_load32(p); }
PGX_HD bool mf_hd_cas32(int* p, int expected, int desired) { return mf_cas32(p, expected, desired); }
PGX_HD bool mf_hd_claim(int* mark, int stamp) { return mf_claim(mark, stamp); }

struct MfTailOut {
    int pushed_to;   // site that received flow along an n-link, or -1
    bool moved;      // flow reached t
    bool listed;     // the site still holds excess that reaches t (by the labels it ended on)
};

PGX_HD void mf_body_tail_step(const MfView& v, int64_t u, MfTailOut* o)
{
    o->pushed_to = -1;
    o->moved = false;
    o->listed = false;
    long long e = mf_load64(&v.ex[u]);
    int du = mf_load32(&v.d[u]);
    const long long rtu = mf_load64(&v.rt[u]);
    const int a_lo = v.off[u], a_hi = v.off[u + 1];
    if (e <= 0 || du == kMfInf || du == kMfDead) return;
    if (rtu > 0) {  // u -> t
        const long long dl = e < rtu ? e : rtu;
        mf_plain_store64(&v.rt[u], rtu - dl);
        mf_add64(&v.ex[u], -dl);
        e -= dl;
        o->moved = true;
    }
    if (e > 0) {
        int best_h = kMfInf, best_a = -1;
        long long best_c = 0;
        for (int a0 = a_lo; a0 < a_hi; a0 += 8) {   // batched loads, as in mf_body_sweep
            long long c[8];
            int w[8], h[8];
            for (int j = 0; j < 8; ++j) {
                const bool in = a0 + j < a_hi;
                c[j] = in ? mf_load64(&v.cap[a0 + j]) : 0;
                w[j] = in ? v.idx[a0 + j] : 0;
            }
            for (int j = 0; j < 8; ++j) h[j] = c[j] > 0 ? mf_load32(&v.d[w[j]]) : kMfInf;
            for (int j = 0; j < 8; ++j)
                if (h[j] < best_h) { best_h = h[j]; best_a = a0 + j; best_c = c[j]; }
        }
        if (best_h == kMfInf) {
            du = kMfInf;   // no residual arc leads anywhere that reaches t
            mf_plain_store32(&v.d[u], du);
        } else if (du > best_h) {
            const long long dl = e < best_c ? e : best_c;   // only this site lowers the capacity of its own arcs
            const int to = v.idx[best_a];
            mf_add64(&v.cap[best_a], -dl);
            mf_add64(&v.cap[v.rev[best_a]], dl);
            mf_add64(&v.ex[u], -dl);
            mf_add64(&v.ex[to], dl);
            e -= dl;
            o->pushed_to = to;
        } else {
            du = best_h + 1;
            if (du >= v.hmax) du = kMfInf;
            mf_plain_store32(&v.d[u], du);
        }
    }
    o->listed = du != kMfInf && e > 0;
}

// reverse BFS without hubs: frontier site w (level k - 1) labels its unlabelled neighbours that have residual capacity to it
// (mf_body_bfs_expand with the hub bookkeeping removed); level 2 runs bottom-up: an unlabelled site looks for a level-1 neighbour
PGX_HD bool mf_body_tail_level2(const MfView& v, int64_t u)
{
    for (int a = v.off[u]; a < v.off[u + 1]; ++a)
        if (mf_load64(&v.cap[a]) > 0 && mf_load32(&v.d[v.idx[a]]) == 1) return true;
    return false;
}

// ---- apply the cut ---------------------------------------------------------------------------------------------
PGX_HD bool mf_body_apply(const MfView& v, int64_t u)
{
    if (v.labels[u] == v.alpha) return false;
    if (v.d[u] == kMfInf) {  // cannot reach t => SOURCE side => takes alpha
        v.labels[u] = v.alpha;
        return true;           // the caller counts relabelled sites into flags[2]
    }
    return false;
}

}  // namespace pgx
