// maxflow_tile.hip — LDS-resident push-relabel for one alpha-expansion move: one workgroup, one launch per move.
//
// Replaces: GCoptimizationGeneralGraph::alpha_expansion + BK max-flow behind pearl::PEARL::labeling
//           (/root/reference/src/pyprogressivex/include/PEARL.h:499-551); upstream source absent [U-5].  Same binary
//           problem, same fixed-point capacities and the same answer (the unique minimal sink side) as maxflow.hip /
//           maxflow_body.hip.h, whose header states the construction; this file is a different SCHEDULE of it.
//
// Why: the level-synchronous schedule of maxflow.hip costs one launch per BFS level and two per sweep (~12 us each of
// dependent device-scope round trips): ~820 launches per min-cut at N = 1e6.  A problem of <= 8192 sites fits one workgroup:
// the sites are permuted into a locality order once per graph (Morton order of the graph's own coordinates, or the points'
// order), heights and excesses live in LDS, arc tables in registers, and ONE launch runs the whole move:
//   * relax:     label-correcting reverse search from t to its fixpoint (exact distances; no launch per level),
//   * discharge: push-relabel sweeps over the active sites (pushes are LDS / workgroup-scope atomics), 16 per outer step,
//   * apply:     the sites that do not reach t are the source side = take alpha.
// Heights are only a heuristic in between: any sequence of capacity-respecting pushes is a preflow, and the move ends when a
// search started from scratch shows that no site with excess reaches t.
// Two kinds of problems arrive here: a whole graph of <= 8192 sites (expand_alpha_tile), and the compacted sub-graph of the
// OPEN sites of a move on a larger graph (expand_alpha_region, "region moves", below).
//
// Memory-scope rule (gfx950, 8 XCDs with one L2 each): an address is touched with ONE scope inside a kernel.  Arc capacities:
// workgroup scope; heights in global memory, hub words, flags: agent scope.
//
// Label costs: beta hubs (s -> y_beta (h), y_beta -> members (inf)) are global words; members with a t-link pull the
// hub's excess through one aggregated reservation per hub; a member pushes back through its own counter f.
// An unused alpha is handled by the stranded-excess test (maxflow_body.hip.h, "gate").  Anything else (materialised alpha
// hub, per-arc weights, the source-side variant of the local optimisation's cut, a hub that only reaches t through
// members without t-links) falls back to maxflow.hip: the function returns PGX_TILE_FALLBACK before touching the labels.
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "maxflow_body.hip.h"
#include "pgx_internal.h"

namespace pgx {

namespace {

constexpr int kInf = 0x3fffffff;
constexpr int kMaxL = 64;
constexpr int kRegionCapTotal = 8192;   // open sites a region move takes (expand_alpha_region)


#define SC_WG __HIP_MEMORY_SCOPE_WORKGROUP
#define SC_AG __HIP_MEMORY_SCOPE_AGENT

template <int SC> __device__ __forceinline__ long long ld64(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SC); }
template <int SC> __device__ __forceinline__ int ld32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SC); }
template <int SC> __device__ __forceinline__ void st32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SC); }
template <int SC> __device__ __forceinline__ void add64(long long* p, long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SC); }
template <int SC> __device__ __forceinline__ long long xadd64(long long* p, long long v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SC); }

// what the region kernels leave for the solver (device memory)
struct RegionInfo {
    int count;        // open sites found (may exceed the capacity: then nothing was built)
    int bad;          // a neighbouring sink could be saturated by the region's arcs: not handled here
    int cnt_alpha;    // sites that already carry alpha
    int pad;
    long long pool[kMaxL];      // per label: sink capacity of the sites OUTSIDE the region
    long long needsum[kMaxL];   // per label: capacity of the region's arcs into such sites
    int cnt[kMaxL];             // sites per label (counted by the region's first kernel: a label in use other than alpha has a hub when h > 0)
};

// Region moves of one expansion cycle are enqueued back to back, the host reads all results at the end (pgx_expansion).  Two
// words in device memory keep that exact:  ctl[0] "poison" - a move declined (the general path has to solve it): every later move
// of the batch returns at once, untouched, and is enqueued again after the host has solved the declined one;  ctl[1] - the moves
// of the batch that relabelled sites so far: a move that relabelled nothing last time and has seen no relabelling since would
// relabel nothing again (it is a function of the labelling and alpha) and returns at once (the host's skip rule, capi.hip).
// Both words are only written by the LAST kernel of a move, so all kernels of a move decide alike.
__device__ __forceinline__ bool batch_skips(const int* ctl, int skip_rel)
{
    if (ctl == nullptr) return false;
    if (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    return skip_rel >= 0 && __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == skip_rel;
}

// state of one move in sorted ("tile") space
struct TView {
    int64_t n;
    int L, alpha;
    long long lambda_q, h_q;
    const long long* dq;  // [L][n] original space
    int* labels;          // [n]    original space
    const int* perm;      // sorted -> original
    const int *off, *idx, *rev, *mult;  // sorted-space CSR
    long long *cap, *ex, *rt, *f;
    int *d, *lab;
    // small block
    long long* hub_e;  // [L]
    int* hub_d;        // [L]
    int* hub_exists;   // [L]
    int* flags;        // 0 active sites that reach t, 1 sites relabelled by apply, 2 flow reached t (discharge), 3 work left after discharge
    unsigned long long* dbg;   // [16] PGX_MF_DEBUG: time per phase of tile 0 (100 MHz ticks), or nullptr
    const struct RegionInfo* rg;   // region mode (expand_alpha_region): the problem is PREPARED in the arrays above, n = rg->count
    int* ctl;          // region moves enqueued without a host round trip (BatchCtl below), or nullptr
    int skip_rel;      // ... this move relabels nothing if exactly this many moves of the batch have relabelled sites so far (-1: unknown)
    int alpha_apply;   // the label apply writes (= alpha; region mode: the move's label, alpha itself is the dummy 1)
    const long long* wq = nullptr;   // [E] per-arc weights in the ORIGINAL arc order (the inlier / outlier cut, pointwise.hip), or nullptr
    const int* goff = nullptr;       // ... and the original CSR offsets: rows keep their entry order, so tile arc a0 + j of site s is arc goff[perm[s]] + j
};

// ---- per-move setup ----------------------------------------------------------------------------------------------------
// t-links and arc capacities of site s (maxflow_body.hip.h mf_body_init_site, in tile space)
__device__ __forceinline__ void init_site(const TView& v, const int64_t s)
{
    const int lu = v.lab[s];
    v.f[s] = 0;
    st32<SC_AG>(&v.d[s], kInf);
    const int a0 = v.off[s], a1 = v.off[s + 1];
    if (lu == v.alpha) {
        v.ex[s] = 0;
        v.rt[s] = 0;
        for (int a = a0; a < a1; ++a) v.cap[a] = 0;
        return;
    }
    const int64_t o = v.perm[s];
    long long keep = v.dq[(int64_t)lu * v.n + o];
    const long long take = v.dq[(int64_t)v.alpha * v.n + o];
    const long long* const wrow = v.wq ? v.wq + v.goff[o] - a0 : nullptr;
    for (int a = a0; a < a1; ++a) {
        const int lq = v.lab[v.idx[a]];
        const long long w = wrow ? wrow[a] : v.lambda_q * (long long)v.mult[a];
        if (lq == v.alpha) { keep += w; v.cap[a] = 0; }
        else if (lq == lu) v.cap[a] = w;
        else { keep += w / 2; v.cap[a] = w / 2; }
    }
    if (keep > take) { v.ex[s] = keep - take; v.rt[s] = 0; }
    else { v.ex[s] = 0; v.rt[s] = take - keep; }
}

// ---- LDS of a tile -----------------------------------------------------------------------------------------------------
template <int T>
struct TileLds {
    int d[T];
    long long ex[T];              // discharge: excess
    unsigned short act[T];        // discharge: sites of the current inner sweep
    unsigned short farl[T];       // discharge: sites that wait for the next outer (device-scope) step
    unsigned char wf[T];          // discharge: site is on farl
    int hubmin[kMaxL];
    int hubd[kMaxL];
    long long hube[kMaxL];
    unsigned long long want[kMaxL];
    long long got[kMaxL];
    int nact, nfar;
};


__device__ __forceinline__ void add32_ag(int* p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SC_AG); }

// ---- relax: reverse label-correcting search from t, to the fixpoint -------------------------------------------------------
// phase 0: heights are reset (1 for sites with a t-link, "unreachable" otherwise) and the arcs of the register tables are used;
// phase 2 (only when some row is longer than the register table): continues from the stored heights and also walks the rest of
// those rows in memory.  Values only fall, every value is witnessed by a path, and the caller repeats phase 2 until it changes
// nothing: that state is the exact distance labelling.  Returns (workgroup-uniform) whether anything was lowered.
template <int NT, int SPT, int MAXD>
__device__ __forceinline__ bool tile_relax(const TView& v, const int tile_n, const int phase, TileLds<NT * SPT>& lds, int& any_far)
{
    const int tid = (int)threadIdx.x;
    bool open[SPT], fpos[SPT], slow[SPT];
    int myd[SPT], d0[SPT], mylab[SPT], rmin[SPT];
    unsigned resm[SPT];
    unsigned pk[SPT][MAXD / 2];
    if (tid < kMaxL) { lds.hubmin[tid] = kInf; lds.hubd[tid] = kInf; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int slot = tid + j * NT;
        const bool valid = slot < tile_n;
        const int64_t s = slot;
        const int lab = valid ? v.lab[s] : v.alpha;
        mylab[j] = lab;
        open[j] = false; fpos[j] = false; slow[j] = false;
        resm[j] = 0;
        rmin[j] = kInf;
        int dl = kInf;
        if (valid && lab != v.alpha) {
            const long long r = v.rt[s];
            dl = phase == 0 ? (r > 0 ? 1 : kInf) : ld32<SC_AG>(&v.d[s]);   // (one scope per address and kernel: v.d is read and written at device scope throughout)
            open[j] = r <= 0;
        }
        myd[j] = dl;
        d0[j] = phase == 0 ? -2 : dl;   // phase 0 publishes every height
        lds.d[slot] = dl;
#pragma unroll
        for (int k = 0; k < MAXD / 2; ++k) pk[j][k] = 0;
        if (open[j]) {
            fpos[j] = v.hub_exists[lab] != 0 && v.f[s] > 0;
            const int a0 = v.off[s], a1 = v.off[s + 1];
            if (a1 > a0) {
                // all loads of the row are issued before the first one is used (clamped addresses instead of branches): one
                // round trip per row instead of one per arc
                int q[MAXD];
                long long c[MAXD];
#pragma unroll
                for (int k = 0; k < MAXD; ++k) q[k] = v.idx[a0 + k < a1 ? a0 + k : a1 - 1];
#pragma unroll
                for (int k = 0; k < MAXD; ++k) c[k] = ld64<SC_WG>(&v.cap[a0 + k < a1 ? a0 + k : a1 - 1]);
#pragma unroll
                for (int k = 0; k < MAXD; ++k) {
                    if (a0 + k >= a1 || c[k] <= 0) continue;
                    resm[j] |= 1u << k;
                    pk[j][k >> 1] |= ((unsigned)q[k] & 0xffffu) << ((k & 1) * 16);
                }
            }
            if (a1 - a0 > MAXD) slow[j] = true;   // rows longer than the register table: the rest is walked in memory at every poll
        }
    }
    bool farany = false;
#pragma unroll
    for (int j = 0; j < SPT; ++j) farany |= slow[j];
    const int far_wg = __syncthreads_or(farany ? 1 : 0);
    any_far = far_wg;
    bool changed = false;
    for (;;) {
        // the arcs beyond the register table
        if (phase > 0 && far_wg) {
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (!slow[j]) continue;
                int m = kInf;
                const int64_t s = tid + j * NT;
                const int a0 = v.off[s], a1 = v.off[s + 1];
                for (int a = a0 + MAXD; a < a1; ++a) {
                    if (ld64<SC_WG>(&v.cap[a]) <= 0) continue;
                    const int h = lds.d[v.idx[a]];
                    m = h < m ? h : m;
                }
                rmin[j] = m;
            }
        }
        if (tid < v.L) lds.hubd[tid] = v.hub_exists[tid] ? ld32<SC_AG>(&v.hub_d[tid]) : kInf;
        __syncthreads();
        bool mine = false;
        for (;;) {
            bool ch = false;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (!open[j]) continue;
                int m = rmin[j];
#pragma unroll
                for (int k = 0; k < MAXD; ++k)
                    if (resm[j] & (1u << k)) {
                        const int h = lds.d[(pk[j][k >> 1] >> ((k & 1) * 16)) & 0xffffu];
                        m = h < m ? h : m;
                    }
                if (fpos[j]) { const int h = lds.hubd[mylab[j]]; m = h < m ? h : m; }
                const int nd = m >= kInf ? kInf : m + 1;
                if (nd < myd[j]) { myd[j] = nd; lds.d[tid + j * NT] = nd; ch = true; }
            }
            mine |= ch;
            if (!__syncthreads_or(ch ? 1 : 0)) break;
        }
        // what the members say about their hubs
#pragma unroll
        for (int j = 0; j < SPT; ++j)
            if (mylab[j] != v.alpha && myd[j] < kInf && v.hub_exists[mylab[j]] && myd[j] + 1 < lds.hubd[mylab[j]])
                atomicMin(&lds.hubmin[mylab[j]], myd[j] + 1);
        // store lowered heights
#pragma unroll
        for (int j = 0; j < SPT; ++j)
            if (myd[j] != d0[j] && tid + j * NT < tile_n) { st32<SC_AG>(&v.d[tid + j * NT], myd[j]); d0[j] = myd[j]; }
        __syncthreads();
        bool hub_ch = false;
        if (tid < v.L && lds.hubmin[tid] < lds.hubd[tid]) {
            atomicMin(&v.hub_d[tid], lds.hubmin[tid]);   // device scope
            hub_ch = true;
        }
        const int any_hub = __syncthreads_or(hub_ch ? 1 : 0);
        const int any = __syncthreads_or(mine ? 1 : 0) | any_hub;
        if (!any) break;
        changed = true;
    }
    return changed;
}

// ---- discharge: push-relabel sweeps of one tile ---------------------------------------------------------------------------
// Take up to `want` out of a shared budget, wait-free (maxflow_body.hip.h mf_reserve, device scope).
__device__ __forceinline__ long long reserve_ag(long long* budget, long long want)
{
    if (want <= 0 || ld64<SC_AG>(budget) <= 0) return 0;
    const long long old = xadd64<SC_AG>(budget, -want);
    if (old >= want) return want;
    const long long got = old > 0 ? old : 0;
    add64<SC_AG>(budget, want - got);
    return got;
}

// One push-relabel step of the site in `slot`.  FAR = false (inner sweeps): the t-link and the arcs - LDS heights, LDS excesses,
// workgroup-scope capacities: no device-scope round trip - and a site that still holds excess and has a residual into its hub is
// deferred to the next outer step (returns true) instead of being relabelled.  FAR = true (outer step): the hub residual as well
// (device scope); the site is relabelled if nothing is admissible.
template <bool FAR, class Lds>
__device__ __forceinline__ bool site_step(const TView& v, Lds& lds, const int slot, bool& moved_now)
{
    const int64_t s = slot;
    const int du = lds.d[slot];
    long long e = lds.ex[slot], pushed = 0;
    bool defer = false;
    const long long r = v.rt[s];
    if (r > 0) {
        const long long dl = e < r ? e : r;
        v.rt[s] = r - dl;
        e -= dl;
        pushed += dl;
        moved_now = true;
    }
    if (e > 0) {
        int minh = kInf;
        bool has_far = false;
        const int a0 = v.off[s], a1 = v.off[s + 1];
        for (int ab = a0; ab < a1 && e > 0; ab += 8) {
            long long c[8];
            int w[8], h[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool in = ab + q < a1;
                w[q] = in ? v.idx[ab + q] : 0;
                c[q] = in ? ld64<SC_WG>(&v.cap[ab + q]) : 0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = c[q] <= 0 ? kInf : lds.d[w[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (c[q] <= 0) continue;
                if (h[q] < du && e > 0) {
                    const long long dl = e < c[q] ? e : c[q];
                    const int a = ab + q, ra = v.rev[a];
                    add64<SC_WG>(&v.cap[a], -dl);
                    add64<SC_WG>(&v.cap[ra], dl);
                    atomicAdd((unsigned long long*)&lds.ex[w[q]], (unsigned long long)dl);
                    e -= dl;
                    pushed += dl;
                    c[q] -= dl;
                }
                if (c[q] > 0 && h[q] < minh) minh = h[q];
            }
        }
        if (e > 0) {
            const int l = v.lab[s];
            if (v.hub_exists[l]) {   // residual site -> y_beta: what the site received from its hub
                const long long fu = v.f[s];
                if (fu > 0) {
                    if (!FAR) has_far = true;
                    else {
                        const int hd = lds.hubd[l];
                        long long left = fu;
                        if (hd < du) {
                            const long long dl = e < fu ? e : fu;
                            v.f[s] = fu - dl;
                            add64<SC_AG>(&v.hub_e[l], dl);
                            e -= dl;
                            pushed += dl;
                            left = fu - dl;
                        }
                        if (left > 0 && hd < minh) minh = hd;
                    }
                }
            }
        }
        if (e > 0) {
            if (!FAR && has_far) defer = true;
            else {   // no admissible arc left: relabel (heights only rise here; a relax pass resets them)
                const int nd = minh >= kInf ? kInf : minh + 1;
                if (nd > du) lds.d[slot] = nd;
            }
        }
    }
    if (pushed > 0) atomicAdd((unsigned long long*)&lds.ex[slot], (unsigned long long)(-pushed));
    return defer;
}

// wave-aggregated append to an LDS list (all lanes call it)
__device__ __forceinline__ void lds_append(int* counter, unsigned short* list, int value, bool want)
{
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
    int b = 0;
    if (lane == leader) b = atomicAdd(counter, __popcll(m));
    b = __shfl(b, leader, 64);
    if (want) list[b + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)value;
}

// `sweeps` = budget of inner sweeps; an outer step (everything that needs a device-scope round trip: the hubs) every kInner
template <int NT, int SPT>
__device__ __forceinline__ void tile_discharge(const TView& v, const int tile_n, TileLds<NT * SPT>& lds, const int sweeps, int& out_left, int& out_moved)
{
    constexpr int base = 0;
    constexpr int kInner = 16;
    const int tid = (int)threadIdx.x;
    int mylab[SPT];
    bool any_hub = false;
    for (int l = 0; l < v.L; ++l) any_hub |= v.hub_exists[l] != 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int slot = tid + j * NT;
        const bool valid = slot < tile_n;
        const int64_t s = (int64_t)base + slot;
        mylab[j] = valid ? v.lab[s] : v.alpha;
        lds.d[slot] = (valid && mylab[j] != v.alpha) ? ld32<SC_AG>(&v.d[s]) : kInf;
        lds.ex[slot] = valid ? v.ex[s] : 0;
        lds.wf[slot] = 0;
    }
    if (tid == 0) { lds.nfar = 0; lds.nact = 0; }
    if (tid < kMaxL) { lds.hube[tid] = 0; lds.hubd[tid] = kInf; }
    __syncthreads();
    bool moved = false;
    int stall = 0;      // inner sweeps in a row in which nothing was delivered to t
    int done = 0;
    while (done < sweeps) {
        // ---- outer step: everything that needs a device-scope round trip
        bool worked = false, moved_now = false;
        if (any_hub && tid < kMaxL) {
            lds.want[tid] = 0;
            lds.got[tid] = 0;
            const bool ex = tid < v.L && v.hub_exists[tid];
            lds.hube[tid] = ex ? ld64<SC_AG>(&v.hub_e[tid]) : 0;
            lds.hubd[tid] = ex ? ld32<SC_AG>(&v.hub_d[tid]) : kInf;
        }
        __syncthreads();
        // hubs holding excess: members with a t-link pull it (one reservation per tile and hub)
        bool pulls = false;
        if (any_hub)
            for (int l = 0; l < v.L; ++l) pulls |= lds.hube[l] > 0 && lds.hubd[l] < kInf;   // uniform
        if (pulls) {
            long long want[SPT], before[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                want[j] = 0;
                before[j] = 0;
                const int slot = tid + j * NT;
                const int l = mylab[j];
                if (slot >= tile_n || l == v.alpha || !(lds.hube[l] > 0) || !(lds.d[slot] < lds.hubd[l])) continue;
                const long long r = v.rt[(int64_t)base + slot];
                if (r > 0) { want[j] = r; before[j] = (long long)atomicAdd(&lds.want[l], (unsigned long long)r); }
            }
            __syncthreads();
            if (tid < v.L && lds.want[tid] > 0) lds.got[tid] = reserve_ag(&v.hub_e[tid], (long long)lds.want[tid]);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (want[j] <= 0) continue;
                long long g = lds.got[mylab[j]] - before[j];
                g = g < 0 ? 0 : (g > want[j] ? want[j] : g);
                if (g > 0) {   // hub -> member -> t
                    const int64_t s = (int64_t)base + tid + j * NT;
                    v.rt[s] -= g;
                    v.f[s] += g;
                    moved_now = true;
                    worked = true;
                }
            }
        }
        // the sites the inner sweeps deferred: all their arcs
        const int nfar = lds.nfar;
        for (int i = tid; i < nfar; i += NT) {
            const int slot = lds.farl[i];
            site_step<true>(v, lds, slot, moved_now);
            lds.wf[slot] = 0;
        }
        if (nfar > 0) worked = true;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) lds.nfar = 0;
        // ---- inner sweeps: arcs inside the tile only
        int inner = 0;
        for (; inner < kInner && done < sweeps; ++inner, ++done) {
            if (tid == 0) lds.nact = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int slot = tid + j * NT;
                lds_append(&lds.nact, lds.act, slot, slot < tile_n && lds.ex[slot] > 0 && lds.d[slot] < kInf && lds.wf[slot] == 0);
            }
            __syncthreads();
            const int nact = lds.nact;
            if (nact == 0) {
                const int mv0 = __syncthreads_or(moved_now ? 1 : 0);
                moved |= mv0 != 0;
                moved_now = false;
                break;
            }
            worked = true;
            for (int i = tid; i < nact; i += NT) {
                const int slot = lds.act[i];
                const bool defer = site_step<false>(v, lds, slot, moved_now);
                if (defer) lds.wf[slot] = 1;
                lds_append(&lds.nfar, lds.farl, slot, defer);
            }
            // (the loop above is not wave-uniform in its trip count: lanes beyond nact skip the append - it only ballots active lanes)
            __builtin_amdgcn_s_waitcnt(0);
            const int mv = __syncthreads_or(moved_now ? 1 : 0);
            moved |= mv != 0;
            moved_now = false;
            // nothing has reached t for a while: what still moves is excess that cannot get there any more, climbing a level per
            // sweep - the next exact search settles that at once
            stall = mv ? 0 : stall + 1;
        }
        if (stall >= 6) break;
        if (!__syncthreads_or((worked || lds.nfar > 0) ? 1 : 0)) break;
    }
    __syncthreads();
    // write the tile back
    bool left = false;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int slot = tid + j * NT;
        if (slot >= tile_n) continue;
        const int64_t s = (int64_t)base + slot;
        v.ex[s] = lds.ex[slot];
        if (mylab[j] != v.alpha) {
            st32<SC_AG>(&v.d[s], lds.d[slot]);
            left |= lds.ex[slot] > 0 && lds.d[slot] < kInf;
        }
    }
    out_left = __syncthreads_or(left ? 1 : 0);
    out_moved = __syncthreads_or(moved ? 1 : 0);
}

// ---- the whole move in ONE launch of ONE workgroup ------------------------------------------------------------------------
// labels + histogram | t-links | per round { search from t to its fixpoint | count | discharge } | gate + apply, separated by
// workgroup barriers.  (A variant with one workgroup per 4096-site tile and device-scope hand-overs between them was exact and
// slower than maxflow.hip's schedule on every configuration measured - docs/lab-notebook.md, round 3 - and has been removed.)
// flags out: [1] sites relabelled, [4] rounds, [5] != 0 gave up: 1 round cap / hub stall, 4 region not valid or a hub in play, [6] a region solver took the move
template <int NT, int SPT, int MAXD>
__global__ __launch_bounds__(NT) void t_move_kernel(TView v, int sweeps, int max_rounds)
{
    constexpr int T = NT * SPT;
    __shared__ TileLds<T> lds;
    __shared__ int s_cnt;
    __shared__ unsigned long long s_stuck;
    __shared__ unsigned long long s_pool[kMaxL];
    __shared__ int s_open;
    const int tid = (int)threadIdx.x;
    const bool region = v.rg != nullptr;   // the arrays hold a prepared problem of rg->count sites, no hubs
    if (region) {
        // two launches per region move: a 256-thread workgroup for regions of <= 1024 sites (nearly all of them; barriers over 4
        // waves instead of 16), then this kernel with 1024 threads, which returns at once when the small one took the move
        if (v.flags[6] != 0) return;        // (plain read: written by the previous kernel)
        if (batch_skips(v.ctl, v.skip_rel)) return;
        const int c = v.rg->count;
        if (c > kRegionCapTotal || v.rg->bad) {   // not built / not valid: the caller runs the general path
            if (tid == 0) {
                st32<SC_AG>(&v.flags[5], 4);
                if (v.ctl) st32<SC_AG>(&v.ctl[0], 1);
            }
            return;
        }
        if (c > T) return;                  // too large for this workgroup: the next launch takes it
        if (tid == 0) st32<SC_AG>(&v.flags[6], 1);
        v.n = c;
    } else if (v.ctl) {
        // a whole-graph move enqueued in a batch (pgx_expansion on a graph of <= 8192 sites: the moves of a cycle back to back, one
        // read-back per batch): behind a move that gave up, or under the skip rule, it returns untouched - as the region moves do
        if (batch_skips(v.ctl, v.skip_rel)) return;
        if (tid == 0) st32<SC_AG>(&v.flags[6], 1);
    }
    const int tile_n = (int)v.n;
    int cnt_alpha = 0;
    if (region) {
        cnt_alpha = v.rg->cnt[v.alpha_apply];
        __syncthreads();
    } else {
        // labels in tile space + histogram
        if (tid < kMaxL) lds.hubmin[tid] = 0;
        __syncthreads();
        for (int i = tid; i < tile_n; i += NT) {
            const int l = v.labels[v.perm[i]];
            v.lab[i] = l;
            atomicAdd(&lds.hubmin[l], 1);
        }
        __syncthreads();
        if (tid < v.L) {
            const bool ex = v.h_q > 0 && tid != v.alpha && lds.hubmin[tid] > 0;
            st32<SC_AG>(&v.hub_exists[tid], ex ? 1 : 0);
            __hip_atomic_store(&v.hub_e[tid], ex ? v.h_q : 0ll, __ATOMIC_RELAXED, SC_AG);
            st32<SC_AG>(&v.hub_d[tid], kInf);
        }
        cnt_alpha = lds.hubmin[v.alpha];
        for (int i = tid; i < tile_n; i += NT) init_site(v, i);
        __syncthreads();
        // A move nobody can want: no site without a t-link (only such a site can fail to reach t - and takes alpha then, ties
        // included) and every label's hub drains into its members' t-links with capacity to spare (sum of their rt > h, strictly:
        // at equality every member's t-link may end saturated).  Then every site reaches t whatever is pushed, the cut is empty and
        // nothing is relabelled: decided here, before the first search - most moves of a steady-state cycle end this way, and the
        // search + count they skip is half of a move's time on the reference's scenes.
        if (tid < kMaxL) s_pool[tid] = 0ull;
        if (tid == 0) s_open = 0;
        __syncthreads();
        bool open = false;
        for (int i = tid; i < tile_n; i += NT) {
            const int l = v.lab[i];
            if (l == v.alpha) continue;
            const long long r = v.rt[i];   // (written by this thread in init_site)
            if (r <= 0) open = true;
            else if (v.h_q > 0) atomicAdd(&s_pool[l], (unsigned long long)r);
        }
        if (open) s_open = 1;
        __syncthreads();
        const bool thin = tid < v.L && v.h_q > 0 && tid != v.alpha && lds.hubmin[tid] > 0 && s_pool[tid] <= (unsigned long long)v.h_q;
        const int thin_any = __syncthreads_or(thin ? 1 : 0);
        if (s_open == 0 && thin_any == 0) return;   // flags stay zero: nothing relabelled, no rounds, not given up
    }
    int rounds = 0, gave_up = 0;
    long long hub_left_prev = -1;
    unsigned long long tprev = wall_clock64();
    auto lap = [&](int k) {   // (debug) time since the last lap goes to slot k
        if (v.dbg && tid == 0) { const unsigned long long t = wall_clock64(); v.dbg[k] += t - tprev; tprev = t; }
    };
    lap(0);
    unsigned long long stuck_final = 0;
    for (;; ++rounds) {
        if (tid < v.L) st32<SC_AG>(&v.hub_d[tid], kInf);
        __syncthreads();
        lap(1);
        int any_far = 0;
        tile_relax<NT, SPT, MAXD>(v, tile_n, 0, lds, any_far);
        // arcs of rows longer than the register table read the stored heights, which phase 0 ignores
        if (any_far) while (tile_relax<NT, SPT, MAXD>(v, tile_n, 2, lds, any_far)) {}
        lap(2);
        // sites with excess that reach t / excess that does not (lds.d holds the heights)
        if (tid == 0) { s_cnt = 0; s_stuck = 0; }
        __syncthreads();
        int mine = 0;
        unsigned long long st = 0;
        for (int i = tid; i < tile_n; i += NT) {
            if (v.lab[i] == v.alpha) continue;
            const long long e = v.ex[i];
            if (e <= 0) continue;
            if (lds.d[i] < kInf) ++mine;
            else st += (unsigned long long)e;
        }
        if (mine) atomicAdd(&s_cnt, mine);
        if (st) atomicAdd(&s_stuck, st);
        __syncthreads();
        const int active = s_cnt;
        stuck_final = s_stuck;
        lap(5);
        bool hub_act = false;
        long long hub_left = 0;
        for (int l = 0; l < v.L; ++l)
            if (v.hub_exists[l]) {
                const long long he = ld64<SC_AG>(&v.hub_e[l]);
                hub_left += he > 0 ? he : 0;
                hub_act |= he > 0 && ld32<SC_AG>(&v.hub_d[l]) < kInf;
            }
        if (active == 0 && !hub_act) break;
        if (active == 0) {   // only hubs hold excess that reaches t: members with a t-link pull it.  Nothing pulled since the last
                             // round => the hub reaches t through members WITHOUT a t-link only: not handled here
            if (hub_left == hub_left_prev) { gave_up = 1; break; }
            hub_left_prev = hub_left;
        }
        if (rounds >= max_rounds) { gave_up = 1; break; }
        int left = 0, moved = 0;
        lap(7);
        tile_discharge<NT, SPT>(v, tile_n, lds, sweeps, left, moved);
        lap(8);
    }
    lap(7);
    int changed = 0;
    if (!gave_up && region) {
        // the hubs were left out: a label's hub drains into members outside the region as long as their sink capacity exceeds what
        // the region's arcs can claim of it by MORE than h (then some member keeps a residual t-link whatever happens, and every
        // member the hub pushed into reaches t through the hub); otherwise the hub interacts with this cut: the general path solves it
        const bool in_play = tid < kMaxL && v.h_q > 0 && tid != v.alpha_apply && v.rg->cnt[tid] > 0 && v.rg->pool[tid] - v.rg->needsum[tid] <= v.h_q;
        if (__syncthreads_or(in_play ? 1 : 0)) gave_up = 4;   // (one label per thread: the serial loop over 64 labels was 4 us of every move)
    }
    if (!gave_up) {
        bool apply = true;
        if (cnt_alpha == 0 && v.h_q > 0) {   // alpha is not in use: taking it costs h once (maxflow_body.hip.h, gate)
            long long total = (long long)stuck_final;
            for (int l = 0; l < v.L; ++l)
                if (v.hub_exists[l]) { const long long he = ld64<SC_AG>(&v.hub_e[l]); total += he > 0 ? he : 0; }
            apply = total >= v.h_q;
        }
        if (apply) {
            int mine = 0;
            for (int i = tid; i < tile_n; i += NT)
                if (v.lab[i] != v.alpha && ld32<SC_AG>(&v.d[i]) == kInf) { v.labels[v.perm[i]] = v.alpha_apply; ++mine; }   // cannot reach t => takes alpha
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            if (mine) atomicAdd(&s_cnt, mine);
            __syncthreads();
            changed = s_cnt;
        }
    }
    lap(10);
    if (tid == 0) {
        if (changed) add32_ag(&v.flags[1], changed);
        st32<SC_AG>(&v.flags[4], rounds + 1);
        st32<SC_AG>(&v.flags[5], gave_up);
        if (v.ctl) {
            if (gave_up) st32<SC_AG>(&v.ctl[0], 1);
            else if (changed) add32_ag(&v.ctl[1], 1);
        }
    }
}

// ---- the whole move in LDS: graphs of <= 1024 sites and <= 8192 arcs -------------------------------------------------------------
// t_move_kernel keeps heights and excesses in LDS but arc capacities, t-links and hub words in memory: on the reference's own scenes
// (unionhouse: 332 sites, 2 168 arcs) a move was 118 us - a few hundred dependent trips to the L2 (an atomic on an arc, a hub word, a
// row of capacities), whatever the width of the workgroup.  Here EVERYTHING the move touches lives in LDS or in its site's registers -
// arc tables, capacities, excesses, heights, the hubs' excess and height in LDS; the site's t-link, hub flow, label and row bounds in
// the registers of the ONE thread that owns the site (the workgroup is at least as wide as the graph) - and memory is read once
// (labels, unary costs, the CSR: two dependent round trips) and written once (the labels of the sites that take alpha).  With the
// hubs in LDS a hub is an ordinary node: when no member with a t-link is left below it, a member WITHOUT one pulls the excess and
// passes it on (t_move_kernel hands such a move back to maxflow.hip).
// The move is the same binary problem with the same answer (unique minimal sink side, fixed-point capacities): any schedule of
// capacity-respecting pushes is a preflow, and the move ends when an exact search from scratch shows that no excess reaches t.
// Every row scan loads its arcs eight at a time BEFORE it looks at any of them: two dependent LDS round trips per eight arcs instead
// of three per arc (with one-arc-at-a-time loops a step of the search was 3.2 us on 332 sites and a sweep 3.9 us: nothing but that chain).
constexpr int kMiniSites = 1024, kMiniArcs = 8192;

struct MiniLds {
    long long cap[kMiniArcs];
    long long ex[kMiniSites];
    int d[kMiniSites];
    unsigned short idx[kMiniArcs], rev[kMiniArcs];
    unsigned char lab[kMiniSites];
    long long hube[kMaxL];
    unsigned long long pool[kMaxL];
    int hubd[kMaxL], cnt[kMaxL];
    unsigned char hubx[kMaxL];
    int note[3];               // per step: bit 0 something changed / was pushed, bit 1 flow reached t (three slots in rotation: one barrier per step)
    int s_open;
    unsigned long long s_stuck;
};

// take up to `want` out of an LDS budget (reserve_ag above, workgroup scope)
__device__ __forceinline__ long long mini_reserve(long long* budget, long long want)
{
    if (want <= 0) return 0;
    const long long old = (long long)atomicAdd((unsigned long long*)budget, (unsigned long long)(-want));
    if (old >= want) return want;
    const long long got = old > 0 ? old : 0;
    atomicAdd((unsigned long long*)budget, (unsigned long long)(want - got));
    return got;
}

// one barrier per step: every WAVE ORs its bits into the step's slot and everybody reads it behind the barrier; thread 0 clears the slot
// read one step earlier (nobody reads it again, and it is written next two barriers from now).  The wave's OR is two ballots: an
// atomicOr of every lane on one LDS word is turned by the compiler into a scalar loop over the active lanes (~30 clocks per lane: 2 us
// per step with the workgroup's lanes active - scripts/micro/lds_chain_bench.hip), which was most of a step.
__device__ __forceinline__ int mini_vote(MiniLds& L, int& step, const int bits)
{
    const int slot = step % 3;
    const int wb = (__ballot(bits & 1) ? 1 : 0) | (__ballot(bits & 2) ? 2 : 0);
    if (wb && (threadIdx.x & 63) == 0) atomicOr(&L.note[slot], wb);
    __syncthreads();
    const int r = L.note[slot];
    if (threadIdx.x == 0) L.note[(step + 2) % 3] = 0;
    ++step;
    return r;
}

// per-thread state of the LDS-resident solver: thread i owns site i
struct MiniSite {
    bool valid, part, my_hub;   // a site of the graph | of the move's graph (does not carry alpha) | its label has a hub
    int lu, a0, a1;             // label, row bounds in the LDS arc tables
    long long rt, f;            // t-link and hub flow: only the owner touches them
    int du;                     // its height (the copy other threads read is L.d[i])
    int nb[16];                 // the first 16 neighbours of the row (slots beyond the row name the site itself)
    bool any_long, any_hub;     // some row is longer than 16 arcs | some label has a hub (workgroup-uniform)
};

// rounds of { exact search | who holds excess that reaches t | sweeps } on the problem in LDS until no excess reaches t.  On return
// s.du is the distance of the LAST search (no sweep behind it): kInf = the site does not reach t.
template <int NT, class Lap>
__device__ __forceinline__ void mini_solve(MiniLds& L, const TView& v, MiniSite& s, int sweeps, const int max_rounds, int& rounds, int& gave_up,
                                           int& n_steps, int& n_sweeps, Lap lap)
{
    const int tid = (int)threadIdx.x;
    const bool valid = s.valid, part = s.part, my_hub = s.my_hub, any_long = s.any_long, any_hub = s.any_hub;
    const int lu = s.lu, a0 = s.a0, a1 = s.a1;
    long long rt = s.rt, f = s.f;
    int du = kInf;
    int nb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) nb[q] = s.nb[q];
    int step = 0;
    rounds = 0;
    gave_up = 0;
    du = kInf;
    n_steps = 0;
    n_sweeps = 0;
    for (;; ++rounds) {
        // ---- search: exact distances to t by label correction from "unreachable" (values only fall, each witnessed by a residual path;
        // the fixpoint of d = 1 + min over residual arcs is the distance labelling).  A hub is a node: below every member (inf arc
        // y -> member), above a member that holds hub flow (f > 0).
        du = (part && rt > 0) ? 1 : kInf;
        if (valid) L.d[tid] = du;
        if (tid < kMaxL) L.hubd[tid] = kInf;
        __syncthreads();
        lap(1);
        const bool open = part && !(rt > 0);
        unsigned resm = 0;   // residual arcs among the row's first 16 (capacities do not change during a search)
        if (open) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { const long long c = L.cap[a0 + q < a1 ? a0 + q : a0]; if (a0 + q < a1 && c > 0) resm |= 1u << q; }
        }
        for (bool first = true;; first = false) {
            int bits = 0;
            bool lowered = false;
            if (open) {
                int m = kInf;
                {
                    int h[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) h[q] = L.d[nb[q]];
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (((resm >> q) & 1u) && h[q] < m) m = h[q];
                }
                if (any_long)
                    for (int a = a0 + 16; a < a1; ++a)
                        if (L.cap[a] > 0) { const int h = L.d[L.idx[a]]; m = h < m ? h : m; }
                if (my_hub && f > 0) { const int h = L.hubd[lu]; m = h < m ? h : m; }
                const int nd = m >= kInf ? kInf : m + 1;
                if (nd < du) { du = nd; L.d[tid] = nd; lowered = true; bits = 1; }
            }
            if (my_hub && (first || lowered) && du < kInf && du + 1 < L.hubd[lu])
                if (atomicMin(&L.hubd[lu], du + 1) > du + 1) bits = 1;
            ++n_steps;
            if (!mini_vote(L, step, bits)) break;
        }
        lap(2);
        // ---- who holds excess that reaches t
        {
            int bits = (part && du < kInf && L.ex[tid] > 0) ? 1 : 0;
            if (any_hub && tid < kMaxL && L.hubx[tid] && L.hube[tid] > 0 && L.hubd[tid] < kInf) bits |= 2;
            const int r = mini_vote(L, step, bits);
            lap(5);
            if (r == 0) break;   // no site and no hub holds excess that reaches t
        }
        if (rounds >= max_rounds) { gave_up = 1; break; }
        // ---- discharge: sweeps over all sites, one barrier each
        int stall = 0;
        for (int s = 0; s < sweeps; ++s) {
            int bits = 0;
            if (part && du < kInf) {
                // the hub above this site holds excess: a member with a t-link takes what the link carries (hub -> member -> t), a member
                // without one takes it all and passes it on below
                if (my_hub && du < L.hubd[lu]) {
                    const long long he = L.hube[lu];
                    if (he > 0) {
                        const long long got = mini_reserve(&L.hube[lu], rt > 0 ? rt : he);
                        if (got > 0) {
                            f += got;
                            if (rt > 0) { rt -= got; bits |= 3; }
                            else { atomicAdd((unsigned long long*)&L.ex[tid], (unsigned long long)got); bits |= 1; }
                        }
                    }
                }
                long long e = L.ex[tid];
                if (e > 0) {
                    bits |= 1;
                    long long pushed = 0;
                    if (rt > 0) {
                        const long long dl = e < rt ? e : rt;
                        rt -= dl;
                        e -= dl;
                        pushed += dl;
                        bits |= 2;
                    }
                    if (e > 0) {
                        int minh = kInf;
                        {   // the row's first 16 arcs: neighbours from registers, capacities and heights in two rounds of loads
                            long long c[16];
                            int h[16];
#pragma unroll
                            for (int q = 0; q < 16; ++q) c[q] = L.cap[a0 + q < a1 ? a0 + q : a0];
#pragma unroll
                            for (int q = 0; q < 16; ++q) h[q] = L.d[nb[q]];
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                if (a0 + q >= a1 || c[q] <= 0) continue;
                                if (h[q] < du && e > 0) {
                                    const long long dl = e < c[q] ? e : c[q];
                                    atomicAdd((unsigned long long*)&L.cap[a0 + q], (unsigned long long)(-dl));
                                    atomicAdd((unsigned long long*)&L.cap[L.rev[a0 + q]], (unsigned long long)dl);
                                    atomicAdd((unsigned long long*)&L.ex[nb[q]], (unsigned long long)dl);
                                    e -= dl;
                                    pushed += dl;
                                    c[q] -= dl;
                                }
                                if (c[q] > 0 && h[q] < minh) minh = h[q];
                            }
                        }
                        if (any_long)
                            for (int a = a0 + 16; a < a1; ++a) {
                                long long c = L.cap[a];
                                if (c <= 0) continue;
                                const int w = L.idx[a];
                                const int h = L.d[w];
                                if (h < du && e > 0) {
                                    const long long dl = e < c ? e : c;
                                    atomicAdd((unsigned long long*)&L.cap[a], (unsigned long long)(-dl));
                                    atomicAdd((unsigned long long*)&L.cap[L.rev[a]], (unsigned long long)dl);
                                    atomicAdd((unsigned long long*)&L.ex[w], (unsigned long long)dl);
                                    e -= dl;
                                    pushed += dl;
                                    c -= dl;
                                }
                                if (c > 0 && h < minh) minh = h;
                            }
                        if (e > 0 && my_hub && f > 0) {   // residual site -> hub: what the site received from it
                            const int hd = L.hubd[lu];
                            long long left = f;
                            if (hd < du) {
                                const long long dl = e < f ? e : f;
                                f -= dl;
                                atomicAdd((unsigned long long*)&L.hube[lu], (unsigned long long)dl);
                                e -= dl;
                                pushed += dl;
                                left = f;
                            }
                            if (left > 0 && hd < minh) minh = hd;
                        }
                        if (e > 0) {   // nothing admissible left: relabel (heights only rise here; the next search resets them)
                            const int nd = minh >= kInf ? kInf : minh + 1;
                            if (nd > du) { du = nd; L.d[tid] = nd; }
                        }
                    }
                    if (pushed > 0) atomicAdd((unsigned long long*)&L.ex[tid], (unsigned long long)(-pushed));
                }
            }
            ++n_sweeps;
            const int r = mini_vote(L, step, bits);
            if (!(r & 1)) break;                 // no site holds excess it could move, no hub was pulled
            stall = (r & 2) ? 0 : stall + 1;     // nothing has reached t for a while: the next exact search settles what is left
            if (stall >= 6) break;
        }
        lap(8);
    }
    s.rt = rt;
    s.f = f;
    s.du = du;
}

template <int NT>
__global__ __launch_bounds__(NT) void t_mini_kernel(TView v, int sweeps, int max_rounds)
{
    __shared__ MiniLds L;
    const int tid = (int)threadIdx.x;
    if (v.ctl) {   // a move of a batch (t_move_kernel)
        if (batch_skips(v.ctl, v.skip_rel)) return;
        if (tid == 0) st32<SC_AG>(&v.flags[6], 1);
    }
    const int n = (int)v.n;   // <= NT (the host picks the width): thread i owns site i
    unsigned long long tprev = wall_clock64();
    auto lap = [&](int k) {   // (debug) time since the last lap goes to slot k
        if (v.dbg && tid == 0) { const unsigned long long t = wall_clock64(); v.dbg[k] += t - tprev; tprev = t; }
    };
    // ---- the problem into LDS and registers.  Round trip 1: the site's place and row; 2: its label, costs and (all threads together)
    // the arc tables; behind the barrier the rows turn the arc weights into capacities.
    const bool valid = tid < n;
    const int64_t o = valid ? v.perm[tid] : 0;
    const int a0 = valid ? v.off[tid] : 0, a1 = valid ? v.off[tid + 1] : 0;
    const int E = v.off[n];
    if (tid < kMaxL) { L.cnt[tid] = 0; L.pool[tid] = 0ull; }
    if (tid < 3) L.note[tid] = 0;
    if (tid == 0) { L.s_open = 0; L.s_stuck = 0; }
    __syncthreads();
    const int lu = valid ? v.labels[o] : v.alpha;
    const long long take = valid ? v.dq[(int64_t)v.alpha * v.n + o] : 0;
    for (int a = tid; a < E; a += NT) {
        L.idx[a] = (unsigned short)v.idx[a];
        L.rev[a] = (unsigned short)v.rev[a];
        if (!v.wq) L.cap[a] = v.lambda_q * (long long)v.mult[a];
    }
    const bool mine_alpha = lu == v.alpha;   // (a site that carries alpha takes no part)
    long long keep = (valid && !mine_alpha) ? v.dq[(int64_t)lu * v.n + o] : 0;
    if (valid) { L.lab[tid] = (unsigned char)lu; atomicAdd(&L.cnt[lu], 1); L.d[tid] = kInf; }
    if (v.wq && valid) {   // per-arc weights in the ORIGINAL arc order: rows keep their entry order
        const long long* const wrow = v.wq + v.goff[o] - a0;
        for (int a = a0; a < a1; ++a) L.cap[a] = wrow[a];
    }
    __syncthreads();
    if (tid < kMaxL) {
        const bool ex = tid < v.L && v.h_q > 0 && tid != v.alpha && L.cnt[tid] > 0;
        L.hubx[tid] = ex ? 1 : 0;
        L.hube[tid] = ex ? v.h_q : 0ll;
        L.hubd[tid] = kInf;
    }
    const int cnt_alpha = L.cnt[v.alpha];
    long long rt = 0, f = 0;   // the site's t-link and what it holds of its hub's flow: only its own thread touches them
    if (valid) {
        long long e0 = 0;
        for (int ab = a0; ab < a1; ab += 8) {
            long long w[8];
            int lq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int a = ab + q < a1 ? ab + q : a1 - 1; w[q] = L.cap[a]; lq[q] = L.idx[a]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) lq[q] = L.lab[lq[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (ab + q >= a1) continue;
                long long c;
                if (mine_alpha || lq[q] == v.alpha) { keep += w[q]; c = 0; }
                else if (lq[q] == lu) c = w[q];
                else { keep += w[q] / 2; c = w[q] / 2; }
                L.cap[ab + q] = c;
            }
        }
        if (!mine_alpha) {
            if (keep > take) e0 = keep - take;
            else rt = take - keep;
            if (rt <= 0) L.s_open = 1;
            else if (v.h_q > 0) atomicAdd(&L.pool[lu], (unsigned long long)rt);
        }
        L.ex[tid] = e0;
    }
    __syncthreads();
    // a move nobody can want (t_move_kernel): no site without a t-link and every hub drains into its members' t-links with room to spare
    const bool thin = tid < kMaxL && L.hubx[tid] && L.pool[tid] <= (unsigned long long)v.h_q;
    const int thin_any = __syncthreads_or(thin ? 1 : 0);
    if (L.s_open == 0 && thin_any == 0) return;   // flags stay zero: nothing relabelled, no rounds, not given up
    const bool any_hub = __syncthreads_or((tid < kMaxL && L.hubx[tid]) ? 1 : 0) != 0;
    const bool part = valid && !mine_alpha;           // the site is in the move's graph
    // the row's first 16 neighbours stay in registers (slots beyond the row name the site itself: harmless in a min over heights): the LDS
    // serves every wave of the workgroup one instruction at a time, and a step's cost is the number of LDS instructions its waves issue
    int nb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) nb[q] = (part && a0 + q < a1) ? (int)L.idx[a0 + q] : (valid ? tid : 0);
    const bool long_row = part && a1 - a0 > 16;
    const bool any_long = __syncthreads_or(long_row ? 1 : 0) != 0;
    const bool my_hub = part && L.hubx[lu] != 0;      // its label has a hub
    lap(0);
    MiniSite st;
    st.valid = valid; st.part = part; st.my_hub = my_hub; st.lu = lu; st.a0 = a0; st.a1 = a1; st.rt = rt; st.f = f; st.du = kInf;
#pragma unroll
    for (int q = 0; q < 16; ++q) st.nb[q] = nb[q];
    st.any_long = any_long; st.any_hub = any_hub;
    int rounds = 0, gave_up = 0, n_steps = 0, n_sweeps = 0;
    mini_solve<NT>(L, v, st, sweeps, max_rounds, rounds, gave_up, n_steps, n_sweeps, lap);
    const int du = st.du;
    lap(7);
    int changed = 0;
    if (!gave_up) {
        bool apply = true;
        if (cnt_alpha == 0 && v.h_q > 0) {   // alpha is not in use: taking it costs h once (maxflow_body.hip.h, gate)
            // stranded excess: of the sites that do not reach t (du: the distance of the LAST search) and of the hubs (none reaches t here)
            if (part && du == kInf) { const long long e = L.ex[tid]; if (e > 0) atomicAdd(&L.s_stuck, (unsigned long long)e); }
            if (tid < kMaxL && L.hubx[tid] && L.hube[tid] > 0) atomicAdd(&L.s_stuck, (unsigned long long)L.hube[tid]);
            __syncthreads();
            apply = (long long)L.s_stuck >= v.h_q;
        }
        if (apply) {
            const bool takes = part && du == kInf;   // cannot reach t => takes alpha (du: the distance of the LAST search, no sweep behind it)
            if (takes) v.labels[o] = v.alpha_apply;
            changed = __syncthreads_count(takes ? 1 : 0);
        }
    }
    lap(10);
    if (v.dbg && tid == 0) { v.dbg[11] += n_steps; v.dbg[12] += rounds; v.dbg[13] += n_sweeps; }
    if (tid == 0) {
        if (changed) add32_ag(&v.flags[1], changed);
        st32<SC_AG>(&v.flags[4], rounds + 1);
        st32<SC_AG>(&v.flags[5], gave_up);
        if (v.ctl) {
            if (gave_up) st32<SC_AG>(&v.ctl[0], 1);
            else if (changed) add32_ag(&v.ctl[1], 1);
        }
    }
}

// ---- a region move of <= 1 024 open sites and <= 8 192 arcs between them, solved in LDS --------------------------------------------
// The problem expand_alpha_region prepares (r_build_kernel: rows of uniform stride, absent arcs marked by head == the site itself, no hubs)
// compacted into the LDS tables of t_mini_kernel - a site's real arcs are counted, a workgroup scan places its row, the position of a
// reverse arc is the neighbour's row start + the rank of the arc among the neighbour's real arcs (one 32-bit mask per site) - and solved by the same
// rounds (mini_solve).  Launched ahead of t_move_kernel's region instance, which returns at once when this kernel took the move
// (flags[6]) and solves what it leaves (more sites or arcs).  C5: the solver's share of a region move 45.6 -> see DESIGN.md 4.3.
__global__ __launch_bounds__(1024) void t_region_mini_kernel(TView v, int stride, int sweeps, int max_rounds)
{
    constexpr int NT = 1024;
    __shared__ MiniLds L;
    __shared__ unsigned s_mask[kMiniSites];
    __shared__ int s_off[kMiniSites + 1];
    __shared__ int s_wsum[16];
    const int tid = (int)threadIdx.x;
    if (v.flags[6] != 0) return;        // (plain read: written by an earlier kernel)
    if (batch_skips(v.ctl, v.skip_rel)) return;
    const int c = v.rg->count;
    if (c > kRegionCapTotal || v.rg->bad) {   // not built / not valid: the caller runs the general path (as t_move_kernel)
        if (tid == 0) {
            st32<SC_AG>(&v.flags[5], 4);
            if (v.ctl) st32<SC_AG>(&v.ctl[0], 1);
        }
        return;
    }
    if (c > kMiniSites || stride > 32) return;   // the next launch takes it
    unsigned long long tprev = wall_clock64();
    auto lap = [&](int k) {
        if (v.dbg && tid == 0) { const unsigned long long t = wall_clock64(); v.dbg[k] += t - tprev; tprev = t; }
    };
    const bool valid = tid < c;
    const int base = tid * stride;
    // real arcs of the row: head != the site itself
    unsigned mask = 0;
    if (valid)
        for (int j = 0; j < stride; ++j)
            if (v.idx[base + j] != tid) mask |= 1u << j;
    const int rdeg = __popc(mask);
    if (tid < 3) L.note[tid] = 0;
    if (tid == 0) L.s_stuck = 0;
    if (valid) s_mask[tid] = mask;
    // exclusive scan of the real degrees over the workgroup
    int incl = rdeg;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += t; }
    if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const int x = s_wsum[w]; if (w < (tid >> 6)) wbase += x; total += x; }
    if (total > kMiniArcs) return;   // (workgroup-uniform) too many arcs for the LDS tables: the next launch takes it
    const int a0 = wbase + incl - rdeg, a1 = a0 + rdeg;
    if (valid) s_off[tid] = a0;
    if (tid == 0) st32<SC_AG>(&v.flags[6], 1);   // this kernel takes the move
    __syncthreads();
    MiniSite st;
    st.valid = valid; st.part = valid; st.my_hub = false; st.lu = 0; st.a0 = a0; st.a1 = a1; st.f = 0; st.du = kInf;
    st.rt = valid ? v.rt[tid] : 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) st.nb[q] = valid ? tid : 0;
    if (valid) {
        L.ex[tid] = v.ex[tid];
        L.d[tid] = kInf;
        int k = 0;
        for (int j = 0; j < stride; ++j) {
            if (!((mask >> j) & 1u)) continue;
            const int head = v.idx[base + j], r = v.rev[base + j];
            const int jj = r - head * stride;   // the reverse arc's slot in the head's row
            L.idx[a0 + k] = (unsigned short)head;
            L.rev[a0 + k] = (unsigned short)(s_off[head] + __popc(s_mask[head] & ((1u << jj) - 1u)));
            L.cap[a0 + k] = v.cap[base + j];
            if (k < 16) {
#pragma unroll
                for (int q = 0; q < 16; ++q) if (q == k) st.nb[q] = head;
            }
            ++k;
        }
    }
    st.any_long = __syncthreads_or((valid && rdeg > 16) ? 1 : 0) != 0;
    st.any_hub = false;
    lap(0);
    int rounds = 0, gave_up = 0, n_steps = 0, n_sweeps = 0;
    mini_solve<NT>(L, v, st, sweeps, max_rounds, rounds, gave_up, n_steps, n_sweeps, lap);
    lap(7);
    int changed = 0;
    if (!gave_up) {
        // the hubs were left out (t_move_kernel, region mode): a label's hub must drain outside the region with room to spare
        const bool in_play = tid < kMaxL && v.h_q > 0 && tid != v.alpha_apply && v.rg->cnt[tid] > 0 && v.rg->pool[tid] - v.rg->needsum[tid] <= v.h_q;
        if (__syncthreads_or(in_play ? 1 : 0)) gave_up = 4;
    }
    if (!gave_up) {
        bool apply = true;
        if (v.rg->cnt[v.alpha_apply] == 0 && v.h_q > 0) {   // alpha is not in use: taking it costs h once (the gate; no hub excess in a region)
            if (valid && st.du == kInf) { const long long e = L.ex[tid]; if (e > 0) atomicAdd(&L.s_stuck, (unsigned long long)e); }
            __syncthreads();
            apply = (long long)L.s_stuck >= v.h_q;
        }
        if (apply) {
            const bool takes = valid && st.du == kInf;
            if (takes) v.labels[v.perm[tid]] = v.alpha_apply;
            changed = __syncthreads_count(takes ? 1 : 0);
        }
    }
    lap(10);
    if (tid == 0) {
        if (changed) add32_ag(&v.flags[1], changed);
        st32<SC_AG>(&v.flags[4], rounds + 1);
        st32<SC_AG>(&v.flags[5], gave_up);
        if (v.ctl) {
            if (gave_up) st32<SC_AG>(&v.ctl[0], 1);
            else if (changed) add32_ag(&v.ctl[1], 1);
        }
    }
}

// ---- region moves: the few sites without a t-link, compacted, solved by one workgroup ------------------------------------------
// In a steady-state move almost every active site has residual capacity to t (it prefers its label): such a site is level 1 of
// every search and a sink for its neighbours.  What a search and the pushes actually work on are the OPEN sites - no t-link:
// they hold excess or are relays - a few hundred to a few thousand of 10^5..10^6.  expand_alpha_region (after maxflow.hip has set
// up t-links and arcs as always) numbers the open sites, builds their sub-graph in compact arrays (uniform row stride, reverse
// arcs by position) and folds every arc into a site WITH a t-link into the open site's own t-link.  That is exact as long as such
// a neighbour cannot be saturated by the region - sum of the region's arc capacities into it < its sink capacity, strictly (checked on the
// device for every neighbour) - and as long as every label's hub can drain outside the region (pool - needsum > h, checked by the
// solver): then all of them stay on the sink side whatever the region does.  The compact problem goes through t_move_kernel's
// rounds in ONE workgroup, which also applies the cut; the host makes a single read-back per move.  Anything else - more than
// 8192 open sites (a new instance taking its points), a saturable neighbour, a hub in play - is handed to the general path.
constexpr int kRegionCap = 8192;

// maxflow.hip's per-site initialisation (t-links, arcs) and, in the same pass, the search for the open sites: they are numbered
// (slot / site), need[] is cleared, and the sink capacity outside the region is summed per label (pool)
__global__ __launch_bounds__(256) void r_init_mark_kernel(MfView mv, RegionInfo* __restrict__ rg, int* __restrict__ slot, int* __restrict__ site,
                                                          long long* __restrict__ need, const int* __restrict__ ctl, int skip_rel)
{
    __shared__ unsigned long long s_pool[kMaxL];
    __shared__ int s_lab[kMaxL];   // sites per label in this workgroup (maxflow.hip's count pass, folded in: the labels are read here anyway)
    if (batch_skips(ctl, skip_rel)) return;
    if (threadIdx.x < kMaxL) { s_pool[threadIdx.x] = 0; s_lab[threadIdx.x] = 0; }
    __syncthreads();
    const int64_t n = mv.n;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n + 255 - (n + 255) % 256; u += (int64_t)gridDim.x * 256) {
        bool open = false;
        int lab_here = -1;
        if (u < n) {
            mf_body_init_site(mv, u);
            need[u] = 0;
            const int lu = mv.labels[u];
            lab_here = lu;
            const long long r = mv.rt[u];
            open = lu != mv.alpha && r <= 0;
            if (!open) slot[u] = -1;
            if (lu != mv.alpha && r > 0) atomicAdd(&s_pool[lu], (unsigned long long)r);
        }
        // label histogram: one LDS atomic per (wave, label present)
        for (unsigned long long todo = __ballot(lab_here >= 0); todo != 0;) {
            const int l0 = __shfl(lab_here, __ffsll((long long)todo) - 1, 64);
            const unsigned long long same = __ballot(lab_here == l0);
            if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)same) - 1) && l0 < kMaxL) atomicAdd(&s_lab[l0], __popcll(same));
            todo &= ~same;
        }
        const unsigned long long m = __ballot(open);
        if (m) {
            const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m) - 1;
            int b = 0;
            if (lane == leader) b = atomicAdd(&rg->count, __popcll(m));
            b = __shfl(b, leader, 64);
            if (open) {
                const int i = b + __popcll(m & ((1ull << lane) - 1ull));
                slot[u] = i < kRegionCap ? i : -1;
                if (i < kRegionCap) site[i] = (int)u;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < mv.L && s_pool[threadIdx.x] > 0) atomicAdd((unsigned long long*)&rg->pool[threadIdx.x], s_pool[threadIdx.x]);
    if ((int)threadIdx.x < kMaxL && s_lab[threadIdx.x] > 0) atomicAdd(&rg->cnt[threadIdx.x], s_lab[threadIdx.x]);
}

// Weak sinks join the region.  A neighbour q with a t-link stays outside only if the region's arcs cannot saturate it:
// need[q] = sum of the capacities of the region's arcs into q < rt[q], STRICTLY - a neighbour whose t-link the region can use up
// exactly may end with no residual to t, and whether it then still reaches t (it takes alpha if not: ties go to alpha, the minimal
// sink side) depends on the rest of the graph.  (The soak found the non-strict form on a unary table full of ties.)  One workgroup walks the region in rounds: the members
// of the round add their arcs' capacities to need[]; the add that crosses rt[q] makes q a member (it is appended and walked in
// the next round), until a round promotes nobody.  (A member that is promoted while a neighbour still adds to its need is harmless:
// need[] of a member is never read.)  A region that is still growing after kPromoteRounds is declined (bad).
constexpr int kPromoteRounds = 16;

__global__ __launch_bounds__(1024) void r_promote_kernel(RegionInfo* __restrict__ rg, int alpha, const int* __restrict__ labels,
                                                         const int* __restrict__ off, const int* __restrict__ idx, const long long* __restrict__ cap,
                                                         const long long* __restrict__ rt, int* __restrict__ slot, int* __restrict__ site,
                                                         long long* __restrict__ need, const int* __restrict__ ctl, int skip_rel)
{
    // The kernel is ONE workgroup: the count of region sites and the list of the sites it appends live in LDS while it runs (the global
    // copies are written for the kernels behind it) - a round no longer starts with two dependent device-scope loads (count, then site;
    // typically 4-5 rounds per move at C5, ~5 us each: histogram in the notebook)
    __shared__ int s_count;
    __shared__ int s_site[kRegionCap];
    if (batch_skips(ctl, skip_rel)) return;
    if (threadIdx.x == 0) s_count = ld32<SC_AG>(&rg->count);
    __syncthreads();
    int lo = 0, hi = s_count;
    if (hi > kRegionCap) return;
    for (int i = (int)threadIdx.x; i < hi; i += 1024) s_site[i] = site[i];   // (written by the kernel before this one)
    __syncthreads();
    for (int round = 0; round < kPromoteRounds && lo < hi; ++round) {
        for (int i = lo + (int)threadIdx.x; i < hi; i += 1024) {
            const int u = s_site[i];
            for (int a = off[u]; a < off[u + 1]; ++a) {
                const long long c = cap[a];
                const int q = idx[a];
                if (c <= 0 || labels[q] == alpha || ld32<SC_AG>(&slot[q]) >= 0) continue;
                const long long old = xadd64<SC_AG>(&need[q], c), r = rt[q];
                if (old < r && old + c >= r) {   // this add reached the neighbour's sink capacity: exactly one add does
                    const int j = atomicAdd(&s_count, 1);
                    if (j < kRegionCap) { s_site[j] = q; st32<SC_AG>(&site[j], q); st32<SC_AG>(&slot[q], j); }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        lo = hi;
        hi = s_count;
        if (hi > kRegionCap) break;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st32<SC_AG>(&rg->count, hi);   // (> kRegionCap: the kernels behind this one decline the move)
        // members promoted in the last round have not added their own arcs to need[] yet: the checks of r_build_kernel would be made
        // on incomplete sums - the region is not valid
        if (hi <= kRegionCap && lo < hi) rg->bad = 1;
    }
}

__global__ __launch_bounds__(256) void r_build_kernel(RegionInfo* __restrict__ rg, int alpha, int stride, const int* __restrict__ labels,
                                                      const int* __restrict__ off, const int* __restrict__ idx, const int* __restrict__ rev,
                                                      const long long* __restrict__ cap, const long long* __restrict__ ex, const long long* __restrict__ rt,
                                                      const int* __restrict__ slot, const int* __restrict__ site, const long long* __restrict__ need, TView c)
{
    if (batch_skips(c.ctl, c.skip_rel)) return;
    const int count = rg->count;
    if (count > kRegionCap) return;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= count) return;
    const int u = site[i];
    const int a0 = off[u], deg = off[u + 1] - a0, base = i * stride;
    const long long own = rt[u];
    long long rtc = own;
    bool bad = false;
    for (int j = 0; j < stride; ++j) {
        int head = i, r = base + j;
        long long cc = 0;
        if (j < deg) {
            const int a = a0 + j, q = idx[a];
            const long long ca = cap[a];
            const int sq = slot[q];
            if (sq >= 0) { head = sq; cc = ca; r = sq * stride + (rev[a] - off[q]); }
            else if (ca > 0 && labels[q] != alpha) {   // a neighbour that keeps its t-link whatever the region does: as good as t
                rtc += ca;
                bad |= need[q] >= rt[q];
                atomicAdd((unsigned long long*)&rg->needsum[labels[q]], (unsigned long long)ca);
            }
        }
        const_cast<int*>(c.idx)[base + j] = head;
        const_cast<int*>(c.rev)[base + j] = r;
        c.cap[base + j] = cc;
    }
    if (bad) rg->bad = 1;
    if (own > 0) atomicAdd((unsigned long long*)&rg->pool[labels[u]], (unsigned long long)(-own));   // a promoted sink is not part of the pool
    const_cast<int*>(c.off)[i] = base;
    const_cast<int*>(c.off)[i + 1] = base + stride;
    const long long e = ex[u];
    if (e >= rtc) { c.ex[i] = e - rtc; c.rt[i] = 0; }   // (only an open site has e > 0, and then own = 0)
    else { c.ex[i] = 0; c.rt[i] = rtc - e; }
    c.f[i] = 0;
    c.d[i] = kInf;
    c.lab[i] = 0;
}

// ---- the graph in tile space ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_iota_kernel(int* p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ __launch_bounds__(256) void tg_inv_deg_kernel(const int* __restrict__ perm, const int* __restrict__ goff, int64_t n,
                                                         int* __restrict__ inv, int* __restrict__ deg)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s > n) return;
    if (s == n) { deg[s] = 0; return; }
    const int o = perm[s];
    inv[o] = (int)s;
    deg[s] = goff[o + 1] - goff[o];
}

__global__ __launch_bounds__(256) void tg_fill_kernel(const int* __restrict__ perm, const int* __restrict__ inv, const int* __restrict__ goff,
                                                      const int* __restrict__ gidx, const int* __restrict__ gmult, const int* __restrict__ grev,
                                                      const int* __restrict__ off2, int64_t n, int* __restrict__ idx2, int* __restrict__ mult2,
                                                      int* __restrict__ rev2)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int o = perm[s];
    const int a0 = goff[o], a1 = goff[o + 1], b0 = off2[s];
    for (int a = a0; a < a1; ++a) {
        const int q = gidx[a];
        idx2[b0 + (a - a0)] = inv[q];
        mult2[b0 + (a - a0)] = gmult[a];
        rev2[b0 + (a - a0)] = off2[inv[q]] + (grev[a] - goff[q]);   // rows keep their entry order
    }
}

}  // namespace

struct TileState {
    int64_t n = 0, E = 0;
    int64_t version = -1;     // graph_version this copy was built from
    DevBuf perm, inv, off, idx, rev, mult, tmp;
    bool slot_is_tile[64] = {};   // per slot of the current batch: a whole-graph move (expand_alpha_tile) rather than a region move
    DevBuf cap, ex, rt, f, d, lab, small, dbg;
    DevBuf rg_slot, rg_need, rg_site, rg_off, rg_idx, rg_rev, rg_cap, rg_site_state;   // region moves (expand_alpha_region)
    DevBuf rg_small, rg_ctl;  // kRegionSlots small blocks (one per move in flight) and the batch's control words
    void* h_rg = nullptr;     // pinned: what the host reads of each slot
    long long region_moves = 0, region_rejects = 0;
    unsigned long long dbg_acc[16] = {0};
    long long dbg_moves = 0;
    void* h_small = nullptr;  // pinned mirror of the small block
};

void tile_free(pgx_ctx* ctx)
{
    TileState* ts = ctx->tile;
    if (!ts) return;
    DevBuf* all[] = {&ts->perm, &ts->inv, &ts->off, &ts->idx, &ts->rev, &ts->mult, &ts->tmp, &ts->cap, &ts->ex,
                     &ts->rt, &ts->f, &ts->d, &ts->lab, &ts->small, &ts->dbg,
                     &ts->rg_slot, &ts->rg_need, &ts->rg_site, &ts->rg_off, &ts->rg_idx, &ts->rg_rev, &ts->rg_cap, &ts->rg_site_state,
                     &ts->rg_small, &ts->rg_ctl};
    for (DevBuf* b : all) release(*b);
    if (ts->h_small) (void)hipHostFree(ts->h_small);
    if (ts->h_rg) (void)hipHostFree(ts->h_rg);
    delete ts;
    ctx->tile = nullptr;
}

// Site order of the tiles: the graph's own Morton order when the graph was built on the device (graph.hip leaves it in
// ctx->gorder), else the points' locality order (pperm), else the caller's order.
static int tile_graph_prepare(pgx_ctx* ctx)
{
    if (!ctx->tile) ctx->tile = new TileState();
    TileState* ts = ctx->tile;
    if (ts->version == ctx->graph_version && ts->n == ctx->gn) return PGX_OK;
    const int64_t n = ctx->gn, E = ctx->gE;
    const unsigned nb = (unsigned)((n + 256) / 256);
    PGX_TRY(ensure(ctx, ts->perm, (size_t)n * sizeof(int)));
    PGX_TRY(ensure(ctx, ts->inv, (size_t)n * sizeof(int)));
    PGX_TRY(ensure(ctx, ts->off, (size_t)(n + 1) * sizeof(int)));
    PGX_TRY(ensure(ctx, ts->idx, (size_t)(E > 0 ? E : 1) * sizeof(int)));
    PGX_TRY(ensure(ctx, ts->rev, (size_t)(E > 0 ? E : 1) * sizeof(int)));
    PGX_TRY(ensure(ctx, ts->mult, (size_t)(E > 0 ? E : 1) * sizeof(int)));
    if (ctx->gorder_n == n && ctx->tile_order != 0)
        PGX_HIP(ctx, hipMemcpyAsync(ts->perm.p, ctx->gorder.p, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    else if (ctx->point_sort && ctx->n == n && ctx->tile_order != 0)
        PGX_HIP(ctx, hipMemcpyAsync(ts->perm.p, ctx->pperm.p, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    else
        hipLaunchKernelGGL(tg_iota_kernel, dim3(nb), dim3(256), 0, ctx->stream, ts->perm.as<int>(), n);
    size_t tmp_bytes = 0;
    int* nulli = nullptr;
    PGX_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, nulli, nulli, 0, (size_t)(n + 1), rocprim::plus<int>(), ctx->stream));
    const size_t arr = ((size_t)(n + 1) * sizeof(int) + 255) & ~(size_t)255;
    PGX_TRY(ensure(ctx, ts->tmp, arr + tmp_bytes + 256));
    int* deg = ts->tmp.as<int>();
    void* tmp = (char*)ts->tmp.p + arr;
    hipLaunchKernelGGL(tg_inv_deg_kernel, dim3(nb), dim3(256), 0, ctx->stream, ts->perm.as<int>(), ctx->goff.as<int>(), n, ts->inv.as<int>(), deg);
    PGX_HIP(ctx, rocprim::exclusive_scan(tmp, tmp_bytes, deg, ts->off.as<int>(), 0, (size_t)(n + 1), rocprim::plus<int>(), ctx->stream));
    if (E > 0)
        hipLaunchKernelGGL(tg_fill_kernel, dim3(nb), dim3(256), 0, ctx->stream, ts->perm.as<int>(), ts->inv.as<int>(), ctx->goff.as<int>(),
                           ctx->gidx.as<int>(), ctx->gmult.as<int>(), ctx->grev.as<int>(), ts->off.as<int>(), n, ts->idx.as<int>(),
                           ts->mult.as<int>(), ts->rev.as<int>());
    PGX_HIP(ctx, hipGetLastError());
    ts->n = n;
    ts->E = E;
    ts->version = ctx->graph_version;
    return PGX_OK;
}

namespace {

struct SmallLayout {   // byte offsets inside the small block
    static constexpr size_t hub_e = 0;                       // [64] i64
    static constexpr size_t hub_d = 64 * 8;                  // [64] i32
    static constexpr size_t hub_exists = hub_d + 64 * 4;     // [64]
    static constexpr size_t flags = hub_exists + 64 * 4;     // [8]
    static constexpr size_t bytes = flags + 8 * 4;
};

}  // namespace

constexpr int kRegionSlots = 64;   // moves in flight per batch (one per label: kMaxL)
constexpr size_t kRegionBlock = (SmallLayout::bytes + sizeof(RegionInfo) + 255) / 256 * 256;   // a move's small block + region info
constexpr size_t kRegionHostOff = SmallLayout::flags;   // the host mirror has the device layout: a slot's flags[8] | count, bad, cnt_alpha sit at this offset of its block

// One expansion move on a graph that fits one workgroup (<= 8192 sites): one launch.  Returns PGX_OK (done, *changed set),
// PGX_TILE_FALLBACK (not handled: the caller runs maxflow.hip; labels untouched) or an error.
int expand_alpha_tile(pgx_ctx* ctx, int64_t n, int L, const long long* dq, int* labels, int64_t lambda_q, int64_t h_q, int alpha, int64_t* changed,
                      const long long* wq)
{
    if (n > ctx->tile_single_max || n > 8192 || L > kMaxL) { ctx->tile_pre_sync = nullptr; return PGX_TILE_FALLBACK; }
    PGX_TRY(tile_graph_prepare(ctx));
    TileState* ts = ctx->tile;
    const int64_t E = ts->E;
    PGX_TRY(ensure(ctx, ts->cap, (size_t)(E > 0 ? E : 1) * 8));
    PGX_TRY(ensure(ctx, ts->ex, (size_t)n * 8));
    PGX_TRY(ensure(ctx, ts->rt, (size_t)n * 8));
    PGX_TRY(ensure(ctx, ts->f, (size_t)n * 8));
    PGX_TRY(ensure(ctx, ts->d, (size_t)n * 4));
    PGX_TRY(ensure(ctx, ts->lab, (size_t)n * 4));
    PGX_TRY(ensure(ctx, ts->small, SmallLayout::bytes));
    if (!ts->h_small) PGX_HIP(ctx, hipHostMalloc(&ts->h_small, SmallLayout::bytes, hipHostMallocDefault));
    // enqueued in a batch (pgx_expansion, ctx->region_defer): the move's small block is a slot of the batch - cleared by
    // region_batch_begin, read back by region_batch_fetch, interpreted by region_result - and nothing here waits for the device
    const bool defer = ctx->region_defer != 0 && wq == nullptr;
    if (defer && (ctx->region_slot < 0 || ctx->region_slot >= kRegionSlots || !ts->rg_small.p || !ts->rg_ctl.p))
        return fail(ctx, PGX_ERR_INVALID, "batched move: slot %d not prepared", ctx->region_slot);
    char* sp = defer ? (char*)ts->rg_small.p + (size_t)ctx->region_slot * kRegionBlock : (char*)ts->small.p;
    if (defer) ts->slot_is_tile[ctx->region_slot] = true;
    TView v;
    v.n = n; v.L = L; v.alpha = alpha; v.lambda_q = lambda_q; v.h_q = h_q;
    v.dq = dq; v.labels = labels;
    v.perm = ts->perm.as<int>();
    v.off = ts->off.as<int>(); v.idx = ts->idx.as<int>(); v.rev = ts->rev.as<int>(); v.mult = ts->mult.as<int>();
    v.cap = ts->cap.as<long long>(); v.ex = ts->ex.as<long long>();
    v.rt = ts->rt.as<long long>(); v.f = ts->f.as<long long>();
    v.d = ts->d.as<int>(); v.lab = ts->lab.as<int>();
    v.hub_e = (long long*)(sp + SmallLayout::hub_e);
    v.hub_d = (int*)(sp + SmallLayout::hub_d);
    v.hub_exists = (int*)(sp + SmallLayout::hub_exists);
    v.flags = (int*)(sp + SmallLayout::flags);
    v.rg = nullptr;
    v.ctl = defer ? ts->rg_ctl.as<int>() : nullptr;
    v.skip_rel = defer ? ctx->region_skip_rel : -1;
    v.alpha_apply = alpha;
    v.wq = wq; v.goff = wq ? ctx->goff.as<int>() : nullptr;
    *changed = 0;
    v.dbg = nullptr;
    if (ctx->tile_debug) {
        PGX_TRY(ensure(ctx, ts->dbg, 16 * 8));
        PGX_HIP(ctx, hipMemsetAsync(ts->dbg.p, 0, 16 * 8, ctx->stream));
        v.dbg = ts->dbg.as<unsigned long long>();
    }
    if (!defer) PGX_HIP(ctx, hipMemsetAsync(sp, 0, SmallLayout::bytes, ctx->stream));
    const int sweeps = ctx->tile_sweeps, max_rounds = 4096;
    if (ctx->tile_mini && n <= kMiniSites && E <= kMiniArcs) {   // everything in LDS (t_mini_kernel)
        if (n <= 256) hipLaunchKernelGGL(t_mini_kernel<256>, dim3(1), dim3(256), 0, ctx->stream, v, ctx->tile_mini_sweeps, max_rounds);
        else if (n <= 512) hipLaunchKernelGGL(t_mini_kernel<512>, dim3(1), dim3(512), 0, ctx->stream, v, ctx->tile_mini_sweeps, max_rounds);
        else hipLaunchKernelGGL(t_mini_kernel<1024>, dim3(1), dim3(1024), 0, ctx->stream, v, ctx->tile_mini_sweeps, max_rounds);
        ctx->tile_launches[0] += 1;
    } else if (n <= 4096) hipLaunchKernelGGL((t_move_kernel<1024, 4, 16>), dim3(1), dim3(1024), 0, ctx->stream, v, sweeps, max_rounds);
    else hipLaunchKernelGGL((t_move_kernel<1024, 8, 16>), dim3(1), dim3(1024), 0, ctx->stream, v, sweeps, max_rounds);
    if (!(ctx->tile_mini && n <= kMiniSites && E <= kMiniArcs)) ctx->tile_launches[1] += 1;
    PGX_HIP(ctx, hipGetLastError());
    if (defer) return PGX_REGION_PENDING;
    char* hs = (char*)ts->h_small;
    std::function<int()> hook;
    hook.swap(ctx->tile_pre_sync);
    ctx->tile_pre_sync_ran = false;
    if (hook) PGX_TRY(hook());   // (the caller's work behind the move: shares the synchronisation below)
    PGX_HIP(ctx, hipMemcpyAsync(hs, sp, SmallLayout::bytes, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int* h_flags = (const int*)(hs + SmallLayout::flags);
    ctx->tile_pre_sync_ran = (bool)hook && h_flags[5] == 0;
    if (ctx->tile_debug) {
        unsigned long long t[16];
        (void)hipMemcpy(t, ts->dbg.p, sizeof(t), hipMemcpyDeviceToHost);
        for (int k = 0; k < 16; ++k) ts->dbg_acc[k] += t[k];
        ts->dbg_moves += 1;
        if (ctx->tile_debug >= 2)
            std::fprintf(stderr, "[tile] alpha=%d rounds=%d gave_up=%d changed=%d\n", alpha, h_flags[4], h_flags[5], h_flags[1]);
        if (ts->dbg_moves % 500 == 0 || ctx->tile_debug >= 2) {
            const double m = 100.0 * (double)ts->dbg_moves;
            std::fprintf(stderr, "[tile] %lld moves, us per move: setup %.1f | reset %.1f search %.1f | count %.1f | decide %.1f discharge %.1f | apply %.1f; per move (LDS-resident kernel only): search steps %.1f, discharges %.2f, sweeps %.1f\n",
                         ts->dbg_moves, ts->dbg_acc[0] / m, ts->dbg_acc[1] / m, ts->dbg_acc[2] / m, ts->dbg_acc[5] / m, ts->dbg_acc[7] / m,
                         ts->dbg_acc[8] / m, ts->dbg_acc[10] / m, (double)ts->dbg_acc[11] / (double)ts->dbg_moves, (double)ts->dbg_acc[12] / (double)ts->dbg_moves,
                         (double)ts->dbg_acc[13] / (double)ts->dbg_moves);
        }
    }
    if (h_flags[5] != 0) return PGX_TILE_FALLBACK;
    ctx->stats[0] += 1;
    ctx->paths[0] += 1;
    ctx->stats[2] += h_flags[4];
    *changed = h_flags[1];
    ctx->stats[4] += *changed;
    return PGX_OK;
}

// One expansion move through the region path.  `mv` = the move's view of maxflow.hip; nothing has run for it yet: the per-site
// initialisation (t-links, arcs) and the label count are this function's first kernel, fused with the search for open sites.  PGX_OK: done, *changed set.  PGX_TILE_FALLBACK: declined - the labels are untouched and mv's state is
// initialised and intact, the general path continues from it.  The caller checks the applicability conditions of the first
// line (maxflow.hip): when they fail nothing has been initialised.

int expand_alpha_region(pgx_ctx* ctx, const MfView& mv, int64_t* changed)
{
    const int64_t n = mv.n;
    const int stride = ctx->max_degree;
    if (stride < 1 || stride > 32 || mv.L > kMaxL || n >= ((int64_t)1 << 30)) return PGX_TILE_FALLBACK;
    if (!ctx->tile) ctx->tile = new TileState();
    TileState* ts = ctx->tile;
    const bool defer = ctx->region_defer != 0;
    const int slot = defer ? ctx->region_slot : 0;
    if (slot < 0 || slot >= kRegionSlots) return fail(ctx, PGX_ERR_INVALID, "region move: slot %d out of range", slot);
    const size_t C = kRegionCap, A = C * (size_t)stride;
    PGX_TRY(ensure(ctx, ts->rg_slot, (size_t)n * 4));
    PGX_TRY(ensure(ctx, ts->rg_need, (size_t)n * 8));
    PGX_TRY(ensure(ctx, ts->rg_site, C * 4));
    PGX_TRY(ensure(ctx, ts->rg_off, (C + 1) * 4));
    PGX_TRY(ensure(ctx, ts->rg_idx, A * 4));
    PGX_TRY(ensure(ctx, ts->rg_rev, A * 4));
    PGX_TRY(ensure(ctx, ts->rg_cap, A * 8));
    PGX_TRY(ensure(ctx, ts->rg_site_state, C * (8 * 3 + 4 * 2)));   // ex | rt | f (i64) | d | lab (i32)
    PGX_TRY(ensure(ctx, ts->rg_small, kRegionSlots * kRegionBlock));
    if (!ts->rg_ctl.p) {
        PGX_TRY(ensure(ctx, ts->rg_ctl, 64));
        PGX_HIP(ctx, hipMemsetAsync(ts->rg_ctl.p, 0, 64, ctx->stream));
    }
    if (!ts->h_rg) PGX_HIP(ctx, hipHostMalloc(&ts->h_rg, kRegionSlots * kRegionBlock, hipHostMallocDefault));
    char* sp = (char*)ts->rg_small.p + (size_t)slot * kRegionBlock;
    ts->slot_is_tile[slot] = false;
    TView v;
    v.n = 0; v.L = 2; v.alpha = 1; v.alpha_apply = mv.alpha; v.lambda_q = mv.lambda_q; v.h_q = mv.h_q;
    v.dq = nullptr; v.labels = mv.labels;
    v.perm = ts->rg_site.as<int>();
    v.off = ts->rg_off.as<int>(); v.idx = ts->rg_idx.as<int>(); v.rev = ts->rg_rev.as<int>(); v.mult = nullptr;
    v.cap = ts->rg_cap.as<long long>();
    long long* st8 = ts->rg_site_state.as<long long>();
    v.ex = st8; v.rt = st8 + C; v.f = st8 + 2 * C;
    v.d = (int*)(st8 + 3 * C); v.lab = v.d + C;
    v.hub_e = (long long*)(sp + SmallLayout::hub_e);
    v.hub_d = (int*)(sp + SmallLayout::hub_d);
    v.hub_exists = (int*)(sp + SmallLayout::hub_exists);
    v.flags = (int*)(sp + SmallLayout::flags);
    v.dbg = nullptr;
    if (ctx->tile_debug) {   // phase timers accumulate on the device over the moves (region_result prints them)
        if (!ts->dbg.p) {
            PGX_TRY(ensure(ctx, ts->dbg, 16 * 8));
            PGX_HIP(ctx, hipMemsetAsync(ts->dbg.p, 0, 16 * 8, ctx->stream));
        }
        v.dbg = ts->dbg.as<unsigned long long>();
    }
    RegionInfo* rg = (RegionInfo*)(sp + SmallLayout::bytes);
    v.rg = rg;
    v.ctl = defer ? ts->rg_ctl.as<int>() : nullptr;
    v.skip_rel = defer ? ctx->region_skip_rel : -1;
    *changed = 0;
    // (a batch's slots are cleared together by region_batch_begin and read back together by region_batch_fetch: one fill and one
    //  copy command per batch instead of one each per move - 8 000 + 9 200 commands of ~3.5 us in a findVanishingPoints call at C5)
    if (!defer) PGX_HIP(ctx, hipMemsetAsync(sp, 0, SmallLayout::bytes + sizeof(RegionInfo), ctx->stream));
    const unsigned nb = (unsigned)((n + 255) / 256), agg = nb < 1024u ? nb : 1024u;
    hipLaunchKernelGGL(r_init_mark_kernel, dim3(agg), dim3(256), 0, ctx->stream, mv, rg, ts->rg_slot.as<int>(), ts->rg_site.as<int>(),
                       ts->rg_need.as<long long>(), (const int*)v.ctl, v.skip_rel);
    hipLaunchKernelGGL(r_promote_kernel, dim3(1), dim3(1024), 0, ctx->stream, rg, mv.alpha, mv.labels, mv.off, mv.idx, mv.cap, mv.rt,
                       ts->rg_slot.as<int>(), ts->rg_site.as<int>(), ts->rg_need.as<long long>(), (const int*)v.ctl, v.skip_rel);
    hipLaunchKernelGGL(r_build_kernel, dim3(kRegionCap / 256), dim3(256), 0, ctx->stream, rg, mv.alpha, stride, mv.labels, mv.off, mv.idx, mv.rev,
                       mv.cap, mv.ex, mv.rt, ts->rg_slot.as<int>(), ts->rg_site.as<int>(), ts->rg_need.as<long long>(), v);
    // the solver: the LDS-resident kernel for regions of <= 1 024 sites / 8 192 arcs (nearly all of them), then the memory-resident one,
    // which returns at once when the first took the move (PGX_TILE_MINI=0: the 256-thread memory-resident instance first, as before)
    if (ctx->tile_mini) hipLaunchKernelGGL(t_region_mini_kernel, dim3(1), dim3(1024), 0, ctx->stream, v, stride, ctx->tile_mini_sweeps, 4096);
    else hipLaunchKernelGGL((t_move_kernel<256, 4, 16>), dim3(1), dim3(256), 0, ctx->stream, v, ctx->tile_sweeps, 4096);
    hipLaunchKernelGGL((t_move_kernel<1024, 8, 16>), dim3(1), dim3(1024), 0, ctx->stream, v, ctx->tile_sweeps, 4096);
    PGX_HIP(ctx, hipGetLastError());
    static_assert(SmallLayout::flags + 8 * 4 == SmallLayout::bytes, "the flags end the small block: flags | region info head is one copy");
    if (defer) return PGX_REGION_PENDING;   // (region_batch_fetch copies the batch's slots)
    char* hs = (char*)ts->h_rg + (size_t)slot * kRegionBlock + kRegionHostOff;
    PGX_HIP(ctx, hipMemcpyAsync(hs, sp + SmallLayout::flags, 8 * 4 + 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int status = 0;
    PGX_TRY(region_result(ctx, 0, mv.alpha, &status, changed));
    return status == 0 ? PGX_OK : PGX_TILE_FALLBACK;
}

// the batch's control words back to zero (enqueued: the moves that follow on the stream see them cleared)
int region_batch_begin(pgx_ctx* ctx)
{
    if (!ctx->tile) ctx->tile = new TileState();
    TileState* ts = ctx->tile;
    PGX_TRY(ensure(ctx, ts->rg_ctl, 64));
    PGX_HIP(ctx, hipMemsetAsync(ts->rg_ctl.p, 0, 64, ctx->stream));
    PGX_TRY(ensure(ctx, ts->rg_small, kRegionSlots * kRegionBlock));
    PGX_HIP(ctx, hipMemsetAsync(ts->rg_small.p, 0, kRegionSlots * kRegionBlock, ctx->stream));   // every slot's small block + region info
    return PGX_OK;
}

// the first `slots` moves of the batch: their flags and region heads to the host mirror in one copy (enqueued; the caller synchronises)
int region_batch_fetch(pgx_ctx* ctx, int slots)
{
    TileState* ts = ctx->tile;
    if (!ts || slots <= 0) return PGX_OK;
    if (!ts->h_rg) PGX_HIP(ctx, hipHostMalloc(&ts->h_rg, kRegionSlots * kRegionBlock, hipHostMallocDefault));
    PGX_HIP(ctx, hipMemcpyAsync(ts->h_rg, ts->rg_small.p, (size_t)slots * kRegionBlock, hipMemcpyDeviceToHost, ctx->stream));
    return PGX_OK;
}

// What the move in `slot` did, once the stream has been synchronised.  status 0: solved (*changed = sites relabelled),
// 1: declined (the general path has to solve it; later moves of the batch did not run), 2: did not run (skipped by the
// no-change rule, or behind a declined move)
int region_result(pgx_ctx* ctx, int slot, int alpha, int* status, int64_t* changed)
{
    TileState* ts = ctx->tile;
    const int* h = (const int*)((const char*)ts->h_rg + (size_t)slot * kRegionBlock + kRegionHostOff);   // flags[8] | count, bad, cnt_alpha
    if (ctx->tile_debug >= 2)
        std::fprintf(stderr, "[region] alpha=%d open=%d bad=%d rounds=%d gave_up=%d taken=%d changed=%d\n", alpha, h[8], h[9], h[4], h[5], h[6], h[1]);
    *changed = 0;
    const bool whole = ts->slot_is_tile[slot];   // a batched whole-graph move: accounted like expand_alpha_tile's own
    if (whole && h[5] != 0) { *status = 1; return PGX_OK; }   // (the caller runs the move again, unbatched: counted there)
    if (whole && h[6] != 0) {
        ctx->stats[0] += 1;
        ctx->paths[0] += 1;
        ctx->stats[2] += h[4];
        *changed = h[1];
        ctx->stats[4] += *changed;
        *status = 0;
        return PGX_OK;
    }
    if (h[5] != 0 || h[7] != 0) { ts->region_rejects += 1; ctx->paths[4] += 1; *status = 1; return PGX_OK; }
    if (h[6] == 0) { *status = 2; return PGX_OK; }
    ts->region_moves += 1;
    if (ctx->tile_debug && ts->region_moves % 1000 == 0 && ts->dbg.p) {
        unsigned long long t[16];
        (void)hipMemcpy(t, ts->dbg.p, sizeof(t), hipMemcpyDeviceToHost);
        const double m = 100.0 * (double)ts->region_moves;
        std::fprintf(stderr, "[region] %lld moves, us per move: setup %.1f | reset %.1f phase0 %.1f gb %.1f | count %.1f gb %.1f | decide %.1f discharge %.1f gb %.1f | apply %.1f\n",
                     ts->region_moves, t[0] / m, t[1] / m, t[2] / m, t[3] / m, t[5] / m, t[6] / m, t[7] / m, t[8] / m, t[9] / m, t[10] / m);
    }
    ctx->stats[0] += 1;
    ctx->paths[2] += 1;
    ctx->stats[2] += h[4];
    *changed = h[1];
    ctx->stats[4] += *changed;
    *status = 0;
    return PGX_OK;
}

}  // namespace pgx
