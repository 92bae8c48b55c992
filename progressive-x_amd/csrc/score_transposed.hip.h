// score_transposed.hip.h — the group-major score kernel with the lanes TRANSPOSED (included by score.hip).
//
// Replaces (for the steps it takes): the filter loop of score_group_kernel, i.e. the batched form of
//           MSACScoringFunctionWithCompoundModel::getScore, /root/reference/src/pyprogressivex/include/
//           scoring_function_with_compound_model.h:78-121.
//
// score_group_kernel gives every lane one POINT of a 64-point group and streams the group's surviving hypotheses past them:
// per (hypothesis, group) step 14 f32 constants come out of LDS as 64-lane broadcasts (28 LDS cycles) for 20 VALU
// instructions, and the loop is bound by VALU issue and by the LDS pipe at the same level (docs/lab-notebook.md, round 3).
// Here a lane owns a TASK = (surviving hypothesis, half of the group's points): the hypothesis' constants sit in the lane's
// registers for the whole pass, the points come out of LDS two at a time (pair-interleaved rows: 48 B = 3 x ds_read_b128 per
// pair for 64 tasks) and the filter runs in packed f32 (v_pk_fma_f32, the constant broadcast to both halves): per 64 tasks x
// 2 points 16 packed + 4 compare + 4 mask instructions and 24 LDS cycles, against 2 x (20 + 28) before.  Tasks are numbered
// over ALL survivors of the group, 64 per pass, so the lanes are full whatever the number of survivors; the passes of a group
// are dealt round-robin to its waves.
// Candidates: a task ends with a 32-bit mask of the points its hypothesis may have as inliers.  Dense tasks (most of the
// half) are evaluated in place with lanes = points, exactly as score_group_kernel's `direct`; the others are expanded into the
// LDS queue of (hypothesis, point) pairs and evaluated 64 at a time on the exact FP64 path in the oracle's operation order
// (the pair's point from the LDS copy of the f64 rows, its model gathered from the component-major copy).  The filter
// evaluates the same expression tree as Filter32<MT>::reject with the same IEEE operations (a packed FMA rounds like a scalar
// one), so it takes bit for bit the same decisions; every contribution is converted to 2^-q fixed point before any summation:
// the accumulators are the same integers whatever kernel, geometry or batching produced them.
#pragma once

namespace pgx {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float x) { return f32x2{x, x}; }

// two points against one hypothesis: rej[k] = Filter32<MT>::reject(point k).  Generic: the scalar filter twice.
template <int MT> struct Filter32Pair {
    static __device__ __forceinline__ void reject2(const f32x2 (&p)[8], const typename Filter32<MT>::Lane& ln, float T2d, bool& r0, bool& r1)
    {
        float a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = p[k].x; b[k] = p[k].y; }
        r0 = Filter32<MT>::reject(a, ln, T2d);
        r1 = Filter32<MT>::reject(b, ln, T2d);
    }
};

// PnP: Filter32<kPnP>::reject, operation for operation, on (point 0, point 1) pairs
template <> struct Filter32Pair<kPnP> {
    static __device__ __forceinline__ void reject2(const f32x2 (&p)[8], const Filter32<kPnP>::Lane& ln, float T2d, bool& r0, bool& r1)
    {
        const float* m = ln.m;
        const f32x2 px = pk_fma(pk_splat(m[0]), p[2], pk_fma(pk_splat(m[1]), p[3], pk_fma(pk_splat(m[2]), p[4], pk_splat(m[3]))));
        const f32x2 py = pk_fma(pk_splat(m[4]), p[2], pk_fma(pk_splat(m[5]), p[3], pk_fma(pk_splat(m[6]), p[4], pk_splat(m[7]))));
        const f32x2 pz = pk_fma(pk_splat(m[8]), p[2], pk_fma(pk_splat(m[9]), p[3], pk_fma(pk_splat(m[10]), p[4], pk_splat(m[11]))));
        const f32x2 a = pk_fma(p[0], pz, -px);
        const f32x2 b = pk_fma(p[1], pz, -py);
        const f32x2 lhs = pk_fma(b, b, a * a);
        const f32x2 rhs = (pz * pz) * pk_splat(T2d);
        const f32x2 tr = pk_fma(pk_splat(ln.c1), p[5], pk_splat(ln.c0));
        r0 = (tr.x <= fabsf(pz.x)) && (lhs.x > rhs.x);   // false on NaN, as the scalar filter
        r1 = (tr.y <= fabsf(pz.y)) && (lhs.y > rhs.y);
    }
};

constexpr int kTSub = 32;                      // points per task
constexpr int kTTasks = 64 / kTSub;            // tasks per surviving hypothesis
constexpr int kTChunkWords = 16;               // hypothesis words whose survivors are listed at a time (1024 hypotheses: a queue entry
                                               // = hypothesis within the chunk (10 bits) | point (6 bits) fits 16 bits)
constexpr int kTDenseMax = 32;                 // most pairs a queued task can hold (a hypothesis with more candidates than this in the group is evaluated in place)
constexpr int kTQueue = 64 + 64 * (kTDenseMax - 1);   // pairs a pass can add behind the remainder of the previous one

template <int MT, bool STATS>
__global__ __launch_bounds__(64, 8) void score_groupT_kernel(
    const double* __restrict__ comp, int64_t n, int groups, const double* __restrict__ models, int W, double T2, int has_comp,
    const unsigned long long* __restrict__ keep, const float* __restrict__ hyp32, double qscale,
    unsigned long long* __restrict__ acc /* [nrep][3][Mpad] */, int Mpad, int split, int xcd_local, const double* __restrict__ models_t,
    unsigned long long* __restrict__ stats, const double* __restrict__ pts_g /* [groups][D][64] */, const float* __restrict__ p32_g /* [groups][8][64] */,
    int nrep, int dense_min /* candidates of kTSub from which a task is evaluated in place */)
{
    using R = Residual<MT>;
    using F32 = Filter32<MT>;
    using LaneT = typename F32::Lane;
    constexpr int kComps = (F32::kRowVals + 1) & ~1;      // f32 values per point the filter reads, even (pairs of float4)
    const int lane = (int)threadIdx.x;
    int g, part;
    if (xcd_local) {  // all parts of a group on one XCD (workgroup ids go round-robin over the 8 XCDs)
        const int slot = (int)(blockIdx.x >> 3);
        g = (slot / split) * 8 + (int)(blockIdx.x & 7u);
        part = slot % split;
        if (g >= groups) return;
    } else {
        g = (int)blockIdx.x / split;
        part = (int)blockIdx.x % split;
    }
    g = __builtin_amdgcn_readfirstlane(g);
    part = __builtin_amdgcn_readfirstlane(part);
    {   // nothing survived the cull for this group
        unsigned long long any = 0;
        for (int w = lane; w < W; w += 64) any |= keep[(int64_t)g * W + w];
        if (__ballot(any != 0) == 0) return;
    }
    __shared__ __attribute__((aligned(16))) float s_p32[32][kComps][2];   // pair-interleaved f32 rows: [pair][component][point of the pair]
    __shared__ double s_pt[R::D + 1][64];                                   // f64 rows (+ the compound value), component-major
    __shared__ unsigned short s_list[kTChunkWords * 64];
    __shared__ unsigned short s_queue[kTQueue];
    const int64_t j = (int64_t)g * 64 + lane;
    const int nvalid = n - (int64_t)g * 64 >= 64 ? 64 : (int)(n - (int64_t)g * 64);
#pragma unroll
    for (int q = 0; q < R::D; ++q) s_pt[q][lane] = pts_g[((int64_t)g * R::D + q) * 64 + lane];
    s_pt[R::D][lane] = has_comp ? comp[j < n ? j : n - 1] : 0.0;
#pragma unroll
    for (int q = 0; q < kComps; ++q) s_p32[lane >> 1][q][lane & 1] = q < F32::kRowVals ? p32_g[((int64_t)g * 8 + q) * 64 + lane] : 0.0f;
    acc += (size_t)(blockIdx.x % (unsigned)nrep) * 3 * (size_t)Mpad;
    const float T2d32 = f32_up(T2 * (1.0 + kFilter32Delta));
    int qn = 0;   // queued candidate pairs (wave-uniform)
    unsigned long long st_steps = 0, st_exact = 0, st_inl = 0;
    unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = STATS ? wall_clock64() : 0;
    auto lap = [&](int k) __attribute__((always_inline)) { if (STATS) { const unsigned long long now = wall_clock64(); tp[k] += now - tlast; tlast = now; } };

    // exact evaluation of the queued pairs [base, base + c), one per lane - score_group_kernel's drain with the point from LDS
    auto drain = [&](int base, int c, int m0) __attribute__((always_inline)) {
        const bool act = lane < c;
        const unsigned e = act ? (unsigned)s_queue[base + lane] : 0u;
        const int m = act ? m0 + (int)(e >> 6) : -1 - lane;
        const int pi = (int)(e & 63u);
        long long cnt = 0, val = 0, shq = 0;
        if (act) {  // exact path: oracle operation order, no contraction
            double q_pt[R::D], mdl[R::P];
#pragma unroll
            for (int k = 0; k < R::D; ++k) q_pt[k] = s_pt[k][pi];
#pragma unroll
            for (int k = 0; k < R::P; ++k) mdl[k] = models_t[(int64_t)k * Mpad + m];
            const double sq = R::squared(q_pt, mdl);
            if (STATS) ++st_exact;
            if (sq < T2) {  // strict, scoring_function_with_compound_model.h:85
                const double sc = cv_max(0.0, 1.0 - sq / T2);                       // :94
                cnt = 1;
                if (STATS) ++st_inl;
                val = to_fixed(sc * qscale);
                if (has_comp) shq = to_fixed(cv_min(s_pt[R::D][pi], sc) * qscale);  // :115-117
            }
        }
        for (int off = 1; off < 64; off <<= 1) {  // segmented sums over runs of equal hypotheses (a task's pairs are contiguous)
            const int mo = __shfl_down(m, off, 64);
            const bool same = lane + off < 64 && mo == m;
            if (__ballot(same) == 0) break;
            const long long c2 = __shfl_down(cnt, off, 64), v2 = __shfl_down(val, off, 64), s2 = __shfl_down(shq, off, 64);
            if (same) { cnt += c2; val += v2; shq += s2; }
        }
        const int mp = __shfl_up(m, 1, 64);
        if (act && (lane == 0 || mp != m) && cnt > 0) {
            atomicAdd(&acc[m], (unsigned long long)cnt);
            atomicAdd(&acc[(int64_t)Mpad + m], (unsigned long long)val);
            if (has_comp) atomicAdd(&acc[2 * (int64_t)Mpad + m], (unsigned long long)shq);
        }
    };
    // exact evaluation in place: lanes = the group's points, one hypothesis (score_group_kernel's direct)
    auto direct = [&](int m, bool cand) __attribute__((always_inline)) {
        double sc = 0.0, shv = 0.0;
        bool inl = false;
        if (cand) {
            double mdl[R::P], pt[R::D];
#pragma unroll
            for (int k = 0; k < R::P; ++k) mdl[k] = models[(int64_t)m * R::P + k];
#pragma unroll
            for (int k = 0; k < R::D; ++k) pt[k] = s_pt[k][lane];   // (the rows live in LDS, not in registers: occupancy)
            const double sq = R::squared(pt, mdl);
            inl = sq < T2;
            if (STATS) { ++st_exact; if (inl) ++st_inl; }
            if (inl) {
                sc = cv_max(0.0, 1.0 - sq / T2);
                if (has_comp) shv = cv_min(s_pt[R::D][lane], sc);
            }
        }
        const unsigned long long bm = __ballot(inl);
        if (bm == 0) return;
        long long val = inl ? to_fixed(sc * qscale) : 0, shq = (inl && has_comp) ? to_fixed(shv * qscale) : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            val += __shfl_down(val, off, 64);
            shq += __shfl_down(shq, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&acc[m], (unsigned long long)__popcll(bm));
            atomicAdd(&acc[(int64_t)Mpad + m], (unsigned long long)val);
            if (has_comp) atomicAdd(&acc[2 * (int64_t)Mpad + m], (unsigned long long)shq);
        }
    };

    if (dense_min > kTDenseMax + 1) dense_min = kTDenseMax + 1;   // (of the hypothesis' 64 points; a queued task then holds at most kTDenseMax pairs)
    lap(0);   // entry: keep words, rows
    int qbase = 0;   // passes of this group dealt so far (round-robin over the group's waves)
    for (int wc = 0; wc < W; wc += kTChunkWords) {
        // survivors of this chunk of hypothesis words: per-word counts, exclusive prefix over the words (lanes 0..31)
        const unsigned long long word = (lane < kTChunkWords && wc + lane < W) ? keep[(int64_t)g * W + wc + lane] : 0ull;
        const int cntw = __popcll(word);
        int incl = cntw;
#pragma unroll
        for (int off = 1; off < kTChunkWords; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        const int S = __shfl(incl, kTChunkWords - 1, 64);
        if (S == 0) continue;
        if (STATS && part == 0) st_steps += (unsigned long long)S;
        const int tasks = S * kTTasks, passes = (tasks + 63) >> 6;
        int first = qbase + ((part - qbase % split) % split + split) % split;   // this wave's first pass of the chunk
        if (first >= qbase + passes) { qbase += passes; continue; }
        const int pre = incl - cntw;
        __builtin_amdgcn_wave_barrier();   // the previous chunk's list has been read
        unsigned long long nz = __ballot(word != 0);
        while (nz != 0) {
            const int w = __builtin_ctzll(nz);
            nz &= nz - 1;
            const unsigned long long ww = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(word >> 32), w) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)word, w);
            const int pw = __builtin_amdgcn_readlane(pre, w);
            if ((ww >> lane) & 1ull)
                s_list[pw + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ww >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ww, 0u))] = (unsigned short)(w * 64 + lane);
        }
        __builtin_amdgcn_wave_barrier();
        lap(1);   // survivor list
        for (int q = first; q < qbase + passes; q += split) {
            const int t = ((q - qbase) << 6) + lane;
            const bool active = t < tasks;
            const int sub = t % kTTasks;
            const int m = wc * 64 + (int)s_list[active ? t / kTTasks : 0];
            const LaneT ln = lane_load<LaneT>(hyp32 + (int64_t)m * kHypRow);   // the task's constants stay in registers for the pass
            if (STATS) { if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(&ln)) == 12345) tp[7] += 1; lap(2); }   // constants arrived
            unsigned mask = 0;
#pragma unroll 2
            for (int jp = 0; jp < kTSub / 2; ++jp) {
                const float4* row = reinterpret_cast<const float4*>(&s_p32[sub * (kTSub / 2) + jp][0][0]);
                f32x2 p[8];
#pragma unroll
                for (int k = 0; k < kComps / 2; ++k) {
                    const float4 v = row[k];
                    p[2 * k] = f32x2{v.x, v.y};
                    p[2 * k + 1] = f32x2{v.z, v.w};
                }
#pragma unroll
                for (int k = kComps; k < 8; ++k) p[k] = f32x2{0.0f, 0.0f};
                bool r0, r1;
                Filter32Pair<MT>::reject2(p, ln, T2d32, r0, r1);
                mask |= (r0 ? 0u : 1u) << (2 * jp);
                mask |= (r1 ? 0u : 2u) << (2 * jp);
            }
            {   // points past the end of the data (the tail group's padding) are nobody's candidates
                const int left = nvalid - sub * kTSub;
                const unsigned vm = left >= kTSub ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
                mask = active ? (mask & vm) : 0u;
            }
            lap(3);   // filter loop
            // dense hypotheses (both tasks of a hypothesis sit in neighbouring lanes of the same pass): in place, lanes = points
            const int pc = __popc(mask) + __popc((unsigned)__shfl_xor((int)mask, 1, 64));
            unsigned long long dn = __ballot(pc >= dense_min && sub == 0);
            while (dn != 0) {
                const int L = __builtin_ctzll(dn);
                dn &= dn - 1;
                const int mL = __builtin_amdgcn_readlane(m, L);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)mask, L), hi = (unsigned)__builtin_amdgcn_readlane((int)mask, L + 1);
                direct(mL, (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) != 0);
            }
            lap(4);   // in-place exact
            // the others: their pairs into the queue (a task's pairs contiguous), evaluated 64 at a time
            unsigned rest = pc >= dense_min ? 0u : mask;
            int inc = __popc(rest);
            const int mine = inc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int tt = __shfl_up(inc, off, 64); if (lane >= off) inc += tt; }
            const int total = __shfl(inc, 63, 64);
            if (total > 0) {
                int o = qn + inc - mine;
                while (rest != 0) {
                    const int b = __builtin_ctz(rest);
                    rest &= rest - 1;
                    s_queue[o++] = (unsigned short)(((unsigned)(m - wc * 64) << 6) | (unsigned)(sub * kTSub + b));
                }
                qn += total;
                __builtin_amdgcn_wave_barrier();
                lap(5);   // queue expansion
                int qh = 0;
                for (; qn - qh >= 64; qh += 64) drain(qh, 64, wc * 64);
                if (qh > 0) {   // the remainder to the front
                    const unsigned short mv = lane < qn - qh ? s_queue[qh + lane] : (unsigned short)0;
                    __builtin_amdgcn_wave_barrier();
                    s_queue[lane] = mv;
                    qn -= qh;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        lap(6);   // queued exact
        qbase += passes;
        if (qn > 0) {   // the queue's entries are relative to this chunk of hypotheses: empty it before the next one
            __builtin_amdgcn_wave_barrier();
            drain(0, qn, wc * 64);
            qn = 0;
        }
    }
    if (STATS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            st_exact += __shfl_down(st_exact, off, 64);
            st_inl += __shfl_down(st_inl, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&stats[0], st_steps);
            atomicAdd(&stats[1], st_exact);
            atomicAdd(&stats[2], st_inl);
            for (int k = 0; k < 8; ++k) atomicAdd(&stats[8 + k], tp[k]);
            atomicAdd(&stats[3], 1ull);
        }
    }
}

}  // namespace pgx
