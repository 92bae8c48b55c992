// maxflow_l0.hip.h — closed-form expansion move for lambda = 0 (no pairwise term): unary + per-label costs only.
//
// Without n-links the binary problem of a move on alpha (maxflow_body.hip.h header) decouples into the label groups
// P_beta, coupled only by "does anybody switch" when alpha is unused:
//   ex_p = max(0, keep_p - take_p)  gain of switching p,   rt_p = max(0, take_p - keep_p)  loss of switching p
//   group beta (label cost h):  switch ALL members iff sum_{P_beta} rt <= h   (saves h; ties -> alpha)
//                               otherwise only members with keep_p >= take_p switch (ties -> alpha)
//   gain_beta = sum ex + max(0, h - sum rt);  alpha unused: switching at all costs h once, so nothing switches unless
//   sum_beta gain_beta >= h (ties -> alpha: the union of all optimal moves, i.e. BK's default-SOURCE rule [U-5]).
// This is exactly the minimal-sink-side cut of the hub graph (beta hub: s->y (h), y->p (inf); alpha hub: p->y (inf),
// y->t (h)): hub excess h is absorbed by sum rt, what is left flows into the alpha hub together with sum ex, and the
// alpha hub saturates iff sum gain >= h.  The push-relabel path needs hundreds of sweeps to drain large label costs
// through one-contender-per-wave pulls; here a move is one reduction pass + one apply pass.  Host/device so that the CPU
// emulation (tests/emu) checks the same decision code against the oracle's Dinic solver.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PGX_L0_HD __host__ __device__ __forceinline__
#else
#define PGX_L0_HD inline
#endif

namespace pgx {

struct L0Decision {
    int switch_any;     // 0: nobody changes label in this move
    int all[64];        // per label: 1 = every member switches to alpha
};

// sums[l*2+0] = sum rt, sums[l*2+1] = sum ex over the active members of label l; cnt[l] = sites carrying label l
PGX_L0_HD void l0_decide(int L, int alpha, long long h_q, const long long* sums, const int* cnt, L0Decision* out)
{
    long long gain = 0;
    for (int l = 0; l < L; ++l) {
        out->all[l] = 0;
        if (l == alpha || cnt[l] == 0) continue;
        const long long srt = sums[l * 2], sex = sums[l * 2 + 1];
        if (h_q > 0 && srt <= h_q) { out->all[l] = 1; gain += sex + (h_q - srt); }
        else gain += sex;
    }
    const bool alpha_hub = h_q > 0 && cnt[alpha] == 0;
    out->switch_any = (!alpha_hub || gain >= h_q) ? 1 : 0;
}

// per-site pieces
PGX_L0_HD void l0_site_terms(const long long* dq, int64_t n, int64_t u, int lu, int alpha, long long* rt, long long* ex)
{
    const long long keep = dq[(int64_t)lu * n + u], take = dq[(int64_t)alpha * n + u];
    *rt = take > keep ? take - keep : 0;
    *ex = keep > take ? keep - take : 0;
}

PGX_L0_HD bool l0_site_switches(const long long* dq, int64_t n, int64_t u, int lu, int alpha, int switch_any, int all_l)
{
    if (!switch_any) return false;
    if (all_l) return true;
    return dq[(int64_t)lu * n + u] >= dq[(int64_t)alpha * n + u];  // keep >= take: ties -> alpha
}

}  // namespace pgx
