// mf_emu.cpp — sequential CPU emulation of the HIP push-relabel expansion move (TEST INFRASTRUCTURE).
//
// Compiles progressive-x_amd/csrc/maxflow_body.hip.h + maxflow_driver.inl with g++ and runs every "kernel" as a loop
// over sites in a (optionally shuffled) order.  It lets `pytest -m "not gpu"` check the algorithm that the GPU runs
// (graph construction, hub handling, BFS/sweep orchestration, termination) against the oracle's Dinic solver on
// thousands of random instances without a GPU.  It is never loaded by the product package.
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "../../progressive-x_amd/csrc/maxflow_driver.inl"
#include "../../progressive-x_amd/csrc/maxflow_l0.hip.h"

using namespace pgx;

namespace {

struct EmuBackend {
    std::mt19937_64 rng;
    bool shuffle;
    std::vector<int64_t> order;
    explicit EmuBackend(int64_t n, uint64_t seed) : rng(seed), shuffle(seed != 0), order((size_t)n)
    {
        std::iota(order.begin(), order.end(), 0);
    }
    template <class F> void each(F f)
    {
        if (shuffle) std::shuffle(order.begin(), order.end(), rng);
        for (int64_t u : order) f(u);
    }
    void count_and_setup(const MfView& v)
    {
        for (int l = 0; l < v.L; ++l) v.cnt[l] = 0;
        each([&](int64_t u) { mf_body_count(v, u); });
        mf_body_hub_setup(v);
    }
    int read_count(const MfView& v, int l) { return v.cnt[l]; }
    void keep_source_reachable_only(const MfView&) {}   // device-only variant (gc_labeling flipped), never requested here
    void init_sites(const MfView& v) { each([&](int64_t u) { mf_body_init_site(v, u); }); }
    void bfs_reset(const MfView& v) { mf_body_bfs_reset(v); }
    void bfs_init(const MfView& v) { each([&](int64_t u) { if (mf_body_bfs_init(v, u, v.bfs_hub_d)) v.flags[0] = 1; }); }
    void bfs_level(const MfView& v, int k)
    {
        const int F = v.fcount[(k - 1) % 3];
        const int base = mf_level_base(v, k);
        std::vector<int> fr(v.order + v.lvl[k - 1], v.order + v.lvl[k - 1] + F);
        if (shuffle) std::shuffle(fr.begin(), fr.end(), rng);
        for (int w : fr) if (mf_body_bfs_expand(v, w, k, v.bfs_hub_d)) v.flags[0] = k;
        // NB: on the device hub distances accumulated during level k become visible only after the level's kernel,
        // and the event test uses distance k-1, which no label written at level k (>= k+1) can produce: same here.
        const int ev = mf_bfs_hub_events(v, k);
        if (ev != 0) each([&](int64_t u) { if (mf_body_bfs_hubpass(v, u, k, (ev & 1) != 0, v.bfs_hub_d)) v.flags[0] = k; });
        v.lvl[k] = base;
        v.fcount[(k + 1) % 3] = 0;
    }
    int read_flag(const MfView& v, int i) { return v.flags[i]; }
    void read_flags(const MfView& v, int out[kMfFlags]) { for (int k = 0; k < kMfFlags; ++k) out[k] = v.flags[k]; }
    void read_flags_and_count(const MfView& v, int out[kMfFlags], int* cnt_alpha) { read_flags(v, out); *cnt_alpha = v.cnt[v.alpha]; }
    void wave(const MfView& v, int k)
    {
        std::vector<int> lv(v.order + v.lvl[k], v.order + v.lvl[k + 1]);
        if (shuffle) std::shuffle(lv.begin(), lv.end(), rng);
        for (int u : lv) mf_body_wave(v, u, k);
    }
    void debug_dump(const MfView&, int) {}
    void bfs_finish(const MfView& v, int slot, int last_level) { mf_body_bfs_finish(v, slot, last_level); }
    void count_active(const MfView& v)
    {
        each([&](int64_t u) { if (mf_body_count_active(v, u)) { v.flags[1] = 1; v.flags[3] += 1; } });
        if (std::getenv("MF_EMU_TRACE")) {   // scripts/exp_emu_hard_moves.py: what a round starts from
            long long ea = 0, es = 0, r = 0; int64_t reach = 0; int maxd = 0; double sumd = 0; int na = 0;
            for (int64_t u = 0; u < v.n; ++u) {
                if (v.labels[u] == v.alpha) continue;
                if (v.d[u] != kMfInf) { ea += v.ex[u]; ++reach; if (v.ex[u] > 0) { if (v.d[u] > maxd) maxd = v.d[u]; sumd += v.d[u]; ++na; } }
                else es += v.ex[u];
                r += v.rt[u];
            }
            std::fprintf(stderr, "    excess that reaches t %.3f, stranded %.3f, deficit %.3f, sites that reach t %lld; sites with excess: deepest level %d, mean %.1f\n",
                         ea / 4294967296.0, es / 4294967296.0, r / 4294967296.0, (long long)reach, maxd, na ? sumd / na : 0.0);
        }
        dump(v, "after bfs");
    }
    int next_stamp = 1;
    int take_stamps(const MfView&, int count) { const int s = next_stamp; next_stamp += count; return s; }
    void build_list(const MfView& v, int stamp)
    {
        v.acnt[0] = v.acnt[1] = 0;
        each([&](int64_t u) { if (mf_listed(v, u) && mf_claim(&v.mark[u], stamp)) v.act[0][v.acnt[0]++] = (int)u; });
    }
    bool step(const MfView& v, int64_t u, int prev, int cur, bool list_mode, int* pushed)
    {
        MfSweepIo io;
        io.list_mode = list_mode;
        const long long want = mf_body_pull_want(v, u, prev, &io.which);
        if (io.which == 1) io.granted = mf_reserve(&v.hub_e[v.labels[u]], want);
        if (io.which == 2) { io.granted = mf_reserve(v.hubA_e, want); v.hubA_want[cur] += want; }
        const bool r = mf_body_sweep(v, u, prev, cur, v.hub_min + cur * v.L, &io);
        v.hubA_e[0] += io.pushedA;
        if (io.moved) v.flags[8] = 1;
        if (pushed) *pushed = io.pushed_to;
        return r;
    }
    void sweep_list(const MfView& v, int prev, int cur, int parity, int stamp)
    {
        if (mf_sweep_idle(v)) return;
        std::vector<int> lst(v.act[parity], v.act[parity] + v.acnt[parity]);
        if (shuffle) std::shuffle(lst.begin(), lst.end(), rng);
        int* out = v.act[1 - parity];
        int* oc = &v.acnt[1 - parity];
        for (int u : lst) {
            int pushed = -1;
            if (step(v, u, prev, cur, true, &pushed)) v.flags[1] = 1;
            if (mf_listed(v, u) && mf_claim(&v.mark[u], stamp)) out[(*oc)++] = u;
            if (pushed >= 0 && mf_claim(&v.mark[pushed], stamp)) out[(*oc)++] = pushed;
        }
    }
    void sweep(const MfView& v, int prev, int cur)
    {
        if (mf_sweep_idle(v)) return;
        each([&](int64_t u) { if (step(v, u, prev, cur, false, nullptr)) v.flags[1] = 1; });
    }
    void dump(const MfView& v, const char* tag)
    {
        if (!std::getenv("MF_EMU_DEBUG")) return;
        std::fprintf(stderr, "%s flags=[%d %d %d %d %d] hubA_rt=%lld hubA_e=%lld hubA_d=%d\n", tag, v.flags[0], v.flags[1], v.flags[2],
                     v.flags[3], v.flags[4], (long long)v.hubA_rt[0], (long long)v.hubA_e[0], v.bfs_hubA_d[0]);
        for (int l = 0; l < v.L; ++l)
            if (v.hub_exists[l]) std::fprintf(stderr, "  hub %d e=%lld bfs_d=%d min=[%d %d %d]\n", l, (long long)v.hub_e[l], v.bfs_hub_d[l],
                v.hub_min[l], v.hub_min[v.L + l], v.hub_min[2 * v.L + l]);
        for (int64_t u = 0; u < v.n; ++u)
            std::fprintf(stderr, "  u=%lld l=%d d=%d ex=%lld rt=%lld f=%lld g=%lld\n", (long long)u, v.labels[u], v.d[u], (long long)v.ex[u],
                         (long long)v.rt[u], (long long)v.f[u], (long long)v.g[u]);
    }
    void sweep_epilogue(const MfView& v, int cur, int next, int consumed, bool = false)
    {
        if (mf_sweep_idle(v)) return;
        mf_body_sweep_epilogue(v, cur, next, consumed);
        dump(v, "after sweep");
    }
    long long stuck_excess(const MfView& v)
    {
        long long s = 0;
        for (int64_t u = 0; u < v.n; ++u) s += mf_body_stuck_excess(v, u);
        for (int l = 0; l < v.L; ++l) if (v.hub_exists[l] && v.hub_e[l] > 0) s += v.hub_e[l];
        return s;
    }
    // ---- the hub-free rounds of maxflow_xcd.hip.h, sequentially, with the same per-site step (host-logic check of the schedule:
    // list invariant, filter pass, stall rule, what the kernel leaves behind; the memory model of the launch is the GPU tests' job)
    bool xcd_on = false;
    int xcd_searches = 0, xcd_round_launches = 0;
    // mf_k_xcd_search: the whole global relabel in one call, hubs passive (declines otherwise)
    bool xcd_search(const MfView& v, int slot, int fl[kMfFlags], int* cnt_alpha, int* levels)
    {
        if (!xcd_on) return false;
        if (v.has_alpha_hub[0]) return false;
        mf_body_bfs_reset(v);
        std::vector<int> fr, nf;
        bool any1 = false;
        auto hub = [&](int64_t u, int k) { const int lu = v.labels[u]; if (v.hub_exists[lu] && k + 1 < v.bfs_hub_d[lu]) v.bfs_hub_d[lu] = k + 1; };
        for (int64_t u = 0; u < v.n; ++u) {
            if (v.labels[u] == v.alpha) continue;
            v.d[u] = v.rt[u] > 0 ? 1 : kMfInf;
            if (v.d[u] == 1) { any1 = true; hub(u, 1); }
        }
        for (int64_t u = 0; u < v.n; ++u)
            if (v.labels[u] != v.alpha && v.d[u] == kMfInf && mf_body_tail_level2(v, u)) fr.push_back((int)u);
        for (int u : fr) { v.d[u] = 2; hub(u, 2); }
        int k = 3, depth = any1 ? 1 : 0;
        for (;; ++k) {
            unsigned long long ev = 0;   // hubs that hand their distance on at this level (a member pulled from them before)
            for (int l = 0; l < v.L; ++l) if (v.hub_exists[l] == 2 && v.bfs_hub_d[l] == k - 1) ev |= 1ull << l;
            if (fr.empty() && ev == 0) break;
            if (!fr.empty()) depth = k - 1;
            nf.clear();
            if (shuffle) std::shuffle(fr.begin(), fr.end(), rng);
            std::vector<std::pair<int, int>> hubs_seen;   // (published after the level, as the device does)
            for (int w : fr)
                for (int a = v.off[w]; a < v.off[w + 1]; ++a) {
                    const int u = v.idx[a];
                    if (v.tot[a] - v.cap[a] > 0 && v.d[u] == kMfInf) { v.d[u] = k; hubs_seen.push_back({u, k}); nf.push_back(u); }
                }
            if (ev != 0)
                for (int64_t u = 0; u < v.n; ++u) {
                    const int lu = v.labels[u];
                    if (lu != v.alpha && ((ev >> lu) & 1ull) && v.f[u] > 0 && v.d[u] == kMfInf) { v.d[u] = k; hubs_seen.push_back({(int)u, k}); nf.push_back((int)u); }
                }
            for (auto& h2 : hubs_seen) hub(h2.first, h2.second);
            fr.swap(nf);
        }
        v.flags[0] = depth;
        mf_body_bfs_finish(v, slot, depth + 1 > 2 ? depth + 1 : 2);
        int act = 0;
        for (int64_t u = 0; u < v.n; ++u) act += (v.labels[u] != v.alpha && v.ex[u] > 0 && v.d[u] != kMfInf) ? 1 : 0;
        v.flags[3] = act;
        if (act > 0) v.flags[1] = 1;
        for (int i = 0; i < kMfFlags; ++i) fl[i] = v.flags[i];
        *cnt_alpha = v.cnt[v.alpha];
        *levels = k;
        ++xcd_searches;
        return true;
    }
    bool xcd_rounds(const MfView& v, const MfTuning& tune, int out[8])
    {
        if (!xcd_on) return false;
        std::vector<int> list, next;
        for (int64_t u = 0; u < v.n; ++u) if (v.labels[u] != v.alpha && v.ex[u] > 0) list.push_back((int)u);
        int stamp = take_stamps(v, tune.xcd_max_rounds * (tune.sweeps_list + 3) + 4);
        int rounds = 0, levels = 0, nsweeps = 0, status = 2;
        out[5] = (int)list.size();
        while (rounds < tune.xcd_max_rounds) {
            ++rounds;
            std::vector<int> fr, nf;
            for (int64_t u = 0; u < v.n; ++u) if (v.labels[u] != v.alpha) v.d[u] = v.rt[u] > 0 ? 1 : kMfInf;
            for (int64_t u = 0; u < v.n; ++u)
                if (v.labels[u] != v.alpha && v.d[u] == kMfInf && mf_body_tail_level2(v, u)) fr.push_back((int)u);
            for (int u : fr) v.d[u] = 2;   // (after the pass: the device's level-2 pass reads only d == 1)
            int k = 3, depth = 2;
            for (; !fr.empty(); ++k) {
                depth = k - 1;
                nf.clear();
                if (shuffle) std::shuffle(fr.begin(), fr.end(), rng);
                for (int w : fr)
                    for (int a = v.off[w]; a < v.off[w + 1]; ++a) {
                        const int u = v.idx[a];
                        if (v.tot[a] - v.cap[a] > 0 && v.d[u] == kMfInf) { v.d[u] = k; nf.push_back(u); }
                    }
                fr.swap(nf);
            }
            levels += k;
            const int stall_limit = tune.stall_sweeps > depth + 2 ? tune.stall_sweeps : depth + 2;
            int stall = 0;
            bool none_left = false;
            for (int s = 0; s <= tune.sweeps_list; ++s, ++stamp) {
                if (list.empty()) { none_left = true; break; }
                next.clear();
                bool moved = false;
                if (shuffle) std::shuffle(list.begin(), list.end(), rng);
                for (int u : list) {
                    if (s == 0) {
                        if (v.ex[u] > 0 && v.d[u] != kMfInf && mf_claim(&v.mark[u], stamp)) next.push_back(u);
                        continue;
                    }
                    MfTailOut o;
                    mf_body_tail_step(v, u, &o);
                    moved |= o.moved;
                    if (o.listed && mf_claim(&v.mark[u], stamp)) next.push_back(u);
                    if (o.pushed_to >= 0 && mf_claim(&v.mark[o.pushed_to], stamp)) next.push_back(o.pushed_to);
                }
                list.swap(next);
                if (s > 0) {
                    ++nsweeps;
                    stall = moved ? 0 : stall + 1;
                    if (stall >= stall_limit) { ++stamp; break; }
                }
            }
            if (none_left) { status = 1; break; }
        }
        out[0] = rounds; out[1] = levels; out[2] = nsweeps; out[3] = status; out[4] = 1;
        ++xcd_round_launches;
        return true;
    }
    void apply(const MfView& v) { each([&](int64_t u) { if (mf_body_apply(v, u)) v.flags[2] += 1; }); }
};

}  // namespace

extern "C" int emu_expand_alpha(int64_t n, int L, const int64_t* Dq_point_major, const int32_t* off,
                                const int32_t* idx, const int32_t* mult, int64_t lambda_q, int64_t h_q, int alpha,
                                int32_t* labels, uint64_t order_seed, int sweeps_per_relabel, int64_t* changed,
                                int64_t* stats)
{
    std::vector<long long> dq((size_t)n * L);
    for (int64_t i = 0; i < n; ++i)
        for (int l = 0; l < L; ++l) dq[(size_t)l * n + i] = Dq_point_major[i * L + l];
    const bool pair = off != nullptr && lambda_q > 0;
    if (!pair && std::getenv("MF_EMU_FORCE_FLOW") == nullptr) {  // the product's lambda = 0 path: closed form
        std::vector<long long> sums((size_t)2 * L, 0);
        std::vector<int> cnt((size_t)L, 0);
        for (int64_t u = 0; u < n; ++u) {
            const int lu = labels[u];
            cnt[lu]++;
            if (lu == alpha) continue;
            long long rt, ex;
            l0_site_terms(dq.data(), n, u, lu, alpha, &rt, &ex);
            sums[lu * 2] += rt; sums[lu * 2 + 1] += ex;
        }
        *changed = 0;
        if (cnt[alpha] == n) return 0;
        L0Decision dec;
        l0_decide(L, alpha, h_q, sums.data(), cnt.data(), &dec);
        for (int64_t u = 0; u < n; ++u) {
            const int lu = labels[u];
            if (lu != alpha && l0_site_switches(dq.data(), n, u, lu, alpha, dec.switch_any, dec.all[lu])) {
                labels[u] = alpha;
                ++*changed;
            }
        }
        if (stats) for (int k = 0; k < 8; ++k) stats[k] = 0;
        return 0;
    }
    const int64_t E = pair ? off[n] : 0;
    std::vector<int> rev((size_t)(E > 0 ? E : 1));
    if (pair)
        for (int64_t u = 0; u < n; ++u)
            for (int a = off[u]; a < off[u + 1]; ++a) {
                const int q = idx[a];
                int r = -1;
                for (int b = off[q]; b < off[q + 1]; ++b) if (idx[b] == u) { r = b; break; }
                if (r < 0) return -10;  // graph not symmetric
                rev[a] = r;
            }
    std::vector<long long> tot((size_t)(E > 0 ? E : 1));
    std::vector<long long> cap((size_t)(E > 0 ? E : 1)), ex((size_t)n), rt((size_t)n), f((size_t)n), g((size_t)n),
        hub_e((size_t)L), hubA_rt(1), hubA_e(1), hubA_want(3);
    std::vector<int> d((size_t)n), cnt((size_t)L), hub_exists((size_t)L), has_alpha(1), bfs_hub_d((size_t)L),
        bfs_hubA_d(1), hub_min((size_t)3 * L), order((size_t)n), lvl((size_t)(n + L + 160)), fcount(3), flags(kMfFlags), act0((size_t)n), act1((size_t)n), acnt(2), mark((size_t)n, 0);
    std::vector<unsigned long long> hubA_min(3);
    MfView v;
    v.n = n; v.L = L; v.alpha = alpha; v.lambda_q = lambda_q; v.h_q = h_q;
    v.dq = dq.data(); v.labels = labels;
    v.off = pair ? off : nullptr; v.idx = idx; v.mult = mult; v.rev = rev.data(); v.wq = nullptr;
    v.cap = cap.data(); v.tot = tot.data(); v.ex = ex.data(); v.rt = rt.data(); v.d = d.data(); v.f = f.data(); v.g = g.data();
    v.cnt = cnt.data(); v.hub_exists = hub_exists.data(); v.hub_e = hub_e.data();
    v.has_alpha_hub = has_alpha.data(); v.hubA_rt = hubA_rt.data(); v.hubA_e = hubA_e.data(); v.hubA_want = hubA_want.data(); v.bfs_hub_d = bfs_hub_d.data();
    v.bfs_hubA_d = bfs_hubA_d.data(); v.hub_min = hub_min.data(); v.hubA_min = hubA_min.data();
    v.flags = flags.data();
    v.order = order.data(); v.lvl = lvl.data(); v.fcount = fcount.data(); v.act[0] = act0.data(); v.act[1] = act1.data(); v.acnt = acnt.data(); v.mark = mark.data();
    v.swept = nullptr;
    v.hmax = (int)(n + L + 3);
    v.gate = std::getenv("MF_EMU_NO_GATE") ? 0 : 1;
    EmuBackend be(n, order_seed);
    MfTuning tune;
    if (sweeps_per_relabel > 0) { tune.sweeps_per_relabel = tune.sweeps_list = sweeps_per_relabel; tune.sweep_check = 1; }
    if (const char* e = std::getenv("MF_EMU_LIST_DIV")) tune.list_div = std::atoi(e);
    if (const char* e = std::getenv("MF_EMU_SWEEPS_LIST")) tune.sweeps_list = std::atoi(e);
    if (const char* e = std::getenv("MF_EMU_STALL")) tune.stall_sweeps = std::atoi(e);
    if (std::getenv("MF_EMU_TRACE")) tune.debug = 1;
    if (const char* e = std::getenv("MF_EMU_XCD")) {   // bit 0: rounds, bit 1: searches (with the depth threshold at 0 so that small problems take it)
        const int x = std::atoi(e);
        tune.xcd = x & 1;
        tune.xcd_search = (x >> 1) & 1;
        tune.xcd_search_min = 0;
        be.xcd_on = x != 0;
    }
    int hint[2] = {1, 1};
    if (tune.xcd_search) tune.bfs_hint = hint;
    if (const char* e = std::getenv("MF_EMU_WAVE_FROM")) tune.wave_from = std::atoi(e);
    int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int r = mf_expand_alpha(be, v, tune, changed, st);
    st[7] = (int64_t)be.xcd_searches * 1000000 + be.xcd_round_launches;   // (what of the above ran the one-launch way)
    if (stats) for (int k = 0; k < 8; ++k) stats[k] = st[k];
    return r;
}
