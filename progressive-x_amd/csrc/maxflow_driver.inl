// maxflow_driver.inl — host orchestration of one expansion move, shared by the HIP backend (maxflow.hip) and the
// sequential CPU emulation used by the host-logic tests (tests/emu).  `Backend` provides one method per kernel.
//
//   count/setup -> init sites -> repeat { global relabel (level-synchronous BFS from t) ; stop if nothing with excess
//   can reach t ; wave pass down the BFS levels ; a batch of push-relabel sweeps } -> apply cut.
#pragma once
#include <chrono>
#include <cstdio>

#include "maxflow_body.hip.h"

namespace pgx {

struct MfTuning {
    int bfs_batch = 8;        // BFS levels issued before the first flag read-back (without a depth hint)
    int bfs_next = 8;         // ... before the second; 16 from then on
    int sweeps_per_relabel = 24;  // over all sites (48 before the BFS got cheaper: C4 end-to-end 3.13 -> 2.99 s)
    int sweep_check = 8;      // read the work-left flag every this many sweeps
    int max_relabels = 4096;  // hard cap on global relabels per move
    int debug = 0;            // PGX_MF_DEBUG: one stderr line per global relabel
    int wave = 1;             // run the level-ordered wave pass after each global relabel
    int wave_max = 24;        // ... only after searches at most this deep (0 = always; PGX_MF_WAVE_MAX), or from the wave_from-th relabel of a move on
    int wave_from = 12;       // (PGX_MF_WAVE_FROM)
    int list_div = 8;         // sweeps visit a work list instead of all sites when <= n / list_div sites are active (0 = never)
    int sweeps_list = 96;     // sweeps per global relabel in list mode (they cost a fraction of a full sweep)
    int stall_sweeps = 8;     // leave a round after max(this, depth of the search + 2) consecutive sweeps without flow reaching t (0 = never)
    int source_reach = 0;     // take alpha only where the SOURCE reaches (minimal source side), see maxflow.hip mf_k_src_*
    int preinit = 0;          // init_sites has already run for this move (the region path declined it); count_and_setup has not
    int xcd = 0;              // 1: after a round whose sweeps ended with work left, the further hub-free rounds run in one persistent launch on one XCD
                              // (maxflow_xcd.hip.h) before the next ordinary search; PGX_MF_XCD
    int xcd_max_rounds = 64;
    int xcd_search = 0;       // 1: a search predicted deeper than xcd_search_min levels (depth of the previous search of its kind) runs its level loop inside
                              // one launch on one XCD (maxflow_xcd.hip.h mf_k_xcd_search) while the hubs are passive; PGX_MF_XCD_SEARCH
    int xcd_search_min = 24;
    int* bfs_hint = nullptr;  // in/out (may be null) [2]: depth of the previous FIRST search of a move / of the previous later search; sizes the first batch
};

// returns 0 on success, 1 if the cap on global relabels was hit
template <class Backend>
int mf_expand_alpha(Backend& be, const MfView& v, const MfTuning& tune, int64_t* changed, int64_t stats[8])
{
    *changed = 0;
    be.count_and_setup(v);   // (labels -> counts -> hub flags, small state: independent of the per-site initialisation)
    // The number of sites that already carry alpha travels back with the first flag read-back of the move (a read-back of
    // its own was a synchronisation per move: 5 % of a findVanishingPoints call).  If EVERY site carries alpha the move's
    // graph is empty: the first search finds nothing and the move ends there, as the early return used to.
    int cnt_alpha = -1;
    if (!tune.preinit) be.init_sites(v);
    stats[0] += 1;
    int sweep_id = 0;
    bool converged = false;
    bool xcd_declined = false;
    for (int it = 0; it < tune.max_relabels && !converged; ++it) {
        // ---- global relabel
        const auto t_search = std::chrono::steady_clock::now();   // (PGX_MF_DEBUG: wall time of the search, read-backs included)
        int level = 1, last = 1;
        const int slot = (sweep_id + 2) % 3;
        int fl[kMfFlags];
        int* hint = tune.bfs_hint ? tune.bfs_hint + (it > 0 ? 1 : 0) : nullptr;
        bool xs = false;   // this search ran inside one launch (no level table: no wave pass behind it)
        if (tune.xcd_search && !xcd_declined && v.gate && v.off != nullptr && hint && *hint > tune.xcd_search_min) {
            int ca = -1, xl = 0;
            if (be.xcd_search(v, slot, fl, &ca, &xl)) {
                xs = true;
                level = xl;
                last = fl[0];
                if (cnt_alpha < 0) cnt_alpha = ca;
                *hint = last > 1 ? last : 1;
            } else xcd_declined = true;   // a member holds hub flow (or the launch is unavailable): level launches for the rest of the move
        }
        if (!xs) {
            be.bfs_reset(v);
            be.bfs_init(v);
            // A read-back costs about as much as three empty level launches: most searches are ~9 levels deep (one batch of
            // eight, then four), deep ones double the batch up to 64.  `last` = the last level that labelled a site.
            // The first batch is sized by the depth of the previous search of the same kind - the first search of a move (from
            // the initial preflow: deep) or a later one (after the sweeps: a few levels) - which in a steady-state cycle are
            // about equally deep from move to move; a fixed batch of eight launched most of its levels on empty frontiers
            // at C5.  A search whose last labelled level is `last` needs the levels 2 .. last + 2.
            int first = tune.bfs_batch;
            if (hint && *hint > 0) first = *hint + 2 < 64 ? *hint + 2 : 64;
            // The search's epilogue and the count of active sites are enqueued right behind the first batch and ONE read-back
            // brings everything: with the batch sized by the hint the search is nearly always complete (two synchronisations
            // per search before: 10 % of a findVanishingPoints call).  If it is not, more levels follow and both are redone
            // (the epilogue restarts the count; everything else it writes only becomes more complete).
            for (int round = 0;; ++round) {
                // later batches of a fixed 16: doubling up to 64 overshot deep searches by dozens of launches on empty frontiers
                // (26 % of all level launches of a find6DPoses + findVanishingPoints run; an empty level costs ~5 us, a read-back ~25)
                const int batch = round == 0 ? first : (round == 1 ? tune.bfs_next : 16);
                for (int b = 0; b < batch; ++b) be.bfs_level(v, ++level);
                if (!v.gate) {   // a materialised alpha hub: its epilogue moves the BFS result slot and must run once, at the end
                    last = be.read_flag(v, 0);
                    if (last <= level - 2 || level >= v.hmax) break;
                    continue;
                }
                be.bfs_finish(v, slot, level);
                be.count_active(v);
                if (cnt_alpha < 0) be.read_flags_and_count(v, fl, &cnt_alpha);
                else be.read_flags(v, fl);
                last = fl[0];
                if (last <= level - 2 || level >= v.hmax) break;
            }
            if (hint) *hint = last > 1 ? last : 1;
            if (!v.gate) {
                be.bfs_finish(v, slot, level);
                be.count_active(v);
                be.read_flags(v, fl);
            }
        }
        stats[2] += 1;
        stats[3] += level;
        if (fl[1] == 0) { converged = true; break; }
        if (tune.debug)
            std::fprintf(stderr, "[mf] alpha=%d relabel=%d levels=%d active_sites=%d hub=%d search_us=%.0f\n", v.alpha, it, level, fl[3], fl[7],
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_search).count());
        if (tune.debug > 1) be.debug_dump(v, tune.debug == 4 ? level : fl[3]);
        // ---- wave pass over the BFS levels, farthest first
        // One launch per level: after a deep search (86 levels at C4) the pass costs more than the list sweeps it saves
        // (find6DPoses PEARL 2.96 -> 2.70 s without it), after a shallow one (7 levels at C5) it pays (2.7 vs 3.2 s).  A move
        // that still needs many relabels gets it back: it is what moved excess along 100-arc paths in round 1.
        if (tune.wave && !xs && v.off != nullptr && (tune.wave_max <= 0 || last <= tune.wave_max || it >= tune.wave_from)) {
            const int kstart = last + 1 < level ? last + 1 : level;  // levels beyond `last` are empty
            for (int k = kstart; k >= 1; --k) be.wave(v, k);
            stats[5] += 1;
        }
        // ---- push-relabel sweeps: over all sites, or over a work list while few sites are active and no beta hub can
        // deliver (a hub that receives excess back needs every member again: flags[6] ends list mode for the round)
        bool list_mode = tune.list_div > 0 && v.off != nullptr && fl[7] == 0 && (int64_t)fl[3] * tune.list_div <= v.n;
        const int budget = list_mode ? tune.sweeps_list : tune.sweeps_per_relabel;
        const int stamp = be.take_stamps(v, budget + 2);
        if (list_mode) be.build_list(v, stamp);
        int parity = 0, s = 0;
        bool round_done = false;
        for (; s < budget && !round_done; ++s) {
            const int cur = sweep_id % 3, prev = (sweep_id + 2) % 3, next = (sweep_id + 1) % 3;
            const bool read_follows = (s + 1) % tune.sweep_check == 0;   // the epilogue then hands the flags to the host itself
            if (list_mode) {
                be.sweep_list(v, prev, cur, parity, stamp + 1 + s);
                be.sweep_epilogue(v, cur, next, parity, read_follows);
                parity ^= 1;
                stats[6] += 1;
            } else {
                be.sweep(v, prev, cur);
                be.sweep_epilogue(v, cur, next, -1, read_follows);
            }
            ++sweep_id;
            stats[1] += 1;
            if ((s + 1) % tune.sweep_check != 0) continue;
            be.read_flags(v, fl);
            if (tune.debug == 5) std::fprintf(stderr, "[mf-sweeps] alpha=%d it=%d s=%d work=%d stall=%d hub=%d list=%d active0=%d\n", v.alpha, it, s + 1, fl[4], fl[11], fl[7], (int)list_mode, fl[3]);
            if (fl[4] == 0) { round_done = true; break; }
            // No flow has reached t for longer than the search was deep (a site `last` levels out needs that many sweeps to
            // deliver): what still moves is excess bouncing between sites that cannot reach t any more, climbing a level or two
            // per bounce towards "unreachable" - the next search settles that at once.  (The inlier / outlier cut of a 10^6-point
            // pose problem spent ~80 of the 96 list sweeps of a round this way.)
            if (tune.stall_sweeps > 0 && fl[11] >= (tune.stall_sweeps > last + 2 ? tune.stall_sweeps : last + 2)) break;
            if (list_mode && fl[6] != 0) list_mode = false;
        }
        // ---- the sweeps ended with work left (budget or stall rule): a hard move.  Its further rounds are short chains of dependent
        // steps on a few thousand sites: they run without hubs in ONE launch on one XCD (maxflow_xcd.hip.h) and leave a preflow from
        // which the next ordinary search - the only place where the move is declared finished - has little or nothing left to do.
        if (tune.xcd && !round_done && v.off != nullptr && v.gate && fl[7] == 0 && (int64_t)fl[3] * tune.list_div <= v.n) {
            int xo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const auto t_xcd = std::chrono::steady_clock::now();
            if (be.xcd_rounds(v, tune, xo)) {
                stats[2] += xo[0];
                stats[3] += xo[1];
                stats[1] += xo[2];
                stats[6] += xo[2];
                if (tune.debug)
                    std::fprintf(stderr, "[mf] alpha=%d xcd rounds=%d levels=%d sweeps=%d status=%d workgroups=%d first_list=%d us=%.0f\n", v.alpha, xo[0], xo[1], xo[2], xo[3], xo[4], xo[5],
                                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_xcd).count());
            }
        }
    }
    if (!converged) return 1;
    if (v.gate && cnt_alpha == 0 && v.h_q > 0) {
        // alpha is not in use: taking it costs h once.  Worth it iff the stranded excess (sites + hubs) reaches h.
        if (be.stuck_excess(v) < v.h_q) return 0;
    }
    if (tune.source_reach) be.keep_source_reachable_only(v);   // (label costs are not supported in this mode: h_q == 0)
    be.apply(v);
    *changed = be.read_flag(v, 2);
    stats[4] += *changed;
    return 0;
}

}  // namespace pgx
