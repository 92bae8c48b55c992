"""Proposal engine: samplers + batched hypothesis generation + GPU scoring + the sequential RANSAC semantics.

Replaces: gcransac::GCRANSAC<Estimator, Graph, MSACScoringFunctionWithCompoundModel>::run as called by
ProgressiveX::run (/root/reference/src/pyprogressivex/include/progressive_x.h:294-299, settings :541-545) and the
sampler classes selected at progressivex_python.cpp:112-115,215-245,353-366,466-482,579-609.  The GC-RANSAC sources are
absent from the snapshot (empty submodule), so the loop is restated [UPSTREAM-MEMORY, U-9]:

  * all `max_iters` minimal samples of one proposal are drawn up front and solved in a batch (host, numpy);
  * every resulting hypothesis is scored on the GPU in ONE launch (pgx_score) with the compound-model term;
  * the reference's sequential behaviour is then replayed on the host over the score table, in hypothesis order:
    the early exit `count + 1 < best.inlier_number` (scoring_function_with_compound_model.h:105-106), "first strictly
    better score wins", and the adaptive iteration bound log(1-conf)/log(1-(inl/N)^m) — so for a given hypothesis list
    the CPU restatement and the GPU path select the same model;
  * the local optimisation - GC-RANSAC's graph-cut + inner-RANSAC procedure (`_graph_cut_lo`; the cut runs on the GPU,
    pgx_gc_labeling, and is the scorer's inlier mask when lambda = 0) with max_local_optimization_number = 50 inner
    trials (progressive_x.h:68) - runs inside that replay at every new so-far-best model, as the sequential loop does.
"""
import numpy as np

from . import _lib


# ---------------------------------------------------------------------------------------------------------------------
# samplers  (ids as in progressivex_python.cpp:215-245)
# ---------------------------------------------------------------------------------------------------------------------
def _distinct_rows(rng, tops, m, retries=4):
    """Rows of m DISTINCT integers, row r uniform over range(tops[r]) (tops[r] >= m).  Fast path: draw with replacement and
    redraw the rows that hold duplicates; rows still bad after a few rounds (likely when tops[r] is close to m: the 7-point
    solver on 7..10 points, PROSAC's first rows) get an exact draw — a partial Fisher-Yates shuffle of their range — so a
    sample never carries repeated indices (the reference's samplers return distinct indices by construction)."""
    tops = np.asarray(tops, dtype=np.int64)
    count = tops.shape[0]
    s = (rng.random((count, m)) * tops[:, None]).astype(np.int64)
    if m < 2:
        return s
    dense = tops < 4 * m                      # rejection succeeds with probability < ~0.5 per round there: go exact at once
    # (which rows hold a repeated index: libpgx's host code - the sort + compare + any of numpy was a third of a draw's time)
    for _ in range(retries):
        bad = _lib.host_rows_with_duplicates(s) & ~dense
        if not bad.any():
            break
        s[bad] = (rng.random((int(bad.sum()), m)) * tops[bad][:, None]).astype(np.int64)
    bad = np.nonzero(_lib.host_rows_with_duplicates(s))[0]
    _fisher_yates_rows(rng, s, tops, bad, m)
    return s


def _fisher_yates_rows(rng, s, tops, bad, m):
    """Exact draw for the rows `bad` of s: m steps of Fisher-Yates on range(tops[r]) with a sparse swap table, all rows at once.
    The offsets come from ONE rng.integers call over a [rows, m] array of bounds - numpy fills it in row-major order, the order in
    which the per-row Python loop this replaces made its scalar calls - so the generator's stream, and with it every seeded result,
    is unchanged (checked bit for bit against that loop, tests/test_host_logic.py); the loop was 1.4 ms per NAPSAC batch of 1 000
    samples: a fifth of a findHomographies call on the reference's own scenes."""
    k = len(bad)
    if k == 0:
        return
    top = tops[bad]
    draws = rng.integers(0, top[:, None] - np.arange(m)[None, :])
    if s.dtype == np.int64 and s.flags.c_contiguous:   # the table walk in libpgx's host code (same rows: tests/test_host_logic.py)
        _lib.host_fisher_yates_rows(draws, bad, s)
        return
    vals = np.tile(np.arange(m, dtype=np.int64), (k, 1))     # the value at positions 0 .. m-1
    epos = np.full((k, m), -1, dtype=np.int64)               # swap table for positions >= m: at most one new entry per step
    evals = np.zeros((k, m), dtype=np.int64)
    rows = np.arange(k)
    for j in range(m):
        kk = j + draws[:, j]
        low = kk < m
        hit = epos == kk[:, None]
        has = hit.any(axis=1)
        col = hit.argmax(axis=1)
        vk = np.where(low, vals[rows, np.minimum(kk, m - 1)], np.where(has, evals[rows, col], kk))
        vj = vals[rows, j].copy()
        vals[rows, j] = vk
        lo = np.nonzero(low)[0]
        vals[lo, kk[lo]] = vj[lo]
        hi = np.nonzero(~low)[0]
        if hi.size:
            c = np.where(has[hi], col[hi], j)
            epos[hi, c] = kk[hi]
            evals[hi, c] = vj[hi]
        s[bad, j] = vk


def _fisher_yates_rows_scalar(rng, s, tops, bad, m):
    """the per-row loop _fisher_yates_rows replaces (kept as the statement of what it computes; tests compare the two)"""
    for r in bad:
        top, swaps = int(tops[r]), {}
        for j in range(m):
            k = j + int(rng.integers(0, top - j))
            vj, vk = swaps.get(j, j), swaps.get(k, k)
            swaps[j], swaps[k] = vk, vj
            s[r, j] = vk


class UniformSampler:
    """gcransac::sampler::UniformSampler: m distinct indices uniformly at random."""

    def __init__(self, n, rng):
        self.n, self.rng = n, rng

    def reset(self):
        pass

    def draw(self, count, m):
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        return _distinct_rows(self.rng, np.full(count, self.n, dtype=np.int64), m)


class PhiloxUniformSampler(UniformSampler):
    """gcransac::sampler::UniformSampler on the in-repo counter-based generator (_rng.py / csrc/rng.hip.h): draw number b of
    this sampler is batch b under the sampler's 64-bit key, so a context that can (`solve_minimal_sampled`) draws the batch on
    the device, inside the solver's launch, and every other consumer gets the same rows from `draw`.  Opt-in
    (`sampler_rng="philox"` on the drop-in calls): the default uniform sampler keeps numpy's stream."""

    def __init__(self, n, rng):
        super().__init__(n, rng)
        self.key = int(rng.integers(0, 1 << 63))     # one draw from the call's seeded generator: the key of every batch
        self.batch = 0                               # draws made so far (NOT restarted by reset(): a new proposal, new samples)
        self.last = None                             # (batch, count, m) of the latest draw: ProposalEngine hands it to the device

    def draw(self, count, m):
        from . import _rng
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        self.last = (self.batch, int(count), int(m))
        self.batch += 1
        return _rng.uniform_samples(self.key, self.last[0], int(count), self.n, int(m))


class PhiloxNapsacSampler(PhiloxUniformSampler):
    """NapsacSampler on the in-repo generator: a uniform centre and m - 1 distinct members of its neighbour list.  Unlike the
    numpy-stream NapsacSampler above, a centre with too few neighbours keeps its row (all -1: a NaN model, the iteration is
    spent) instead of being dropped, so sample s is a pure function of (key, batch, s) and the device can draw the batch."""
    kind = "napsac"

    def __init__(self, n, rng, graph):
        super().__init__(n, rng)
        self.off, self.idx = np.asarray(graph[0], dtype=np.int64), np.asarray(graph[1], dtype=np.int64)

    def draw(self, count, m):
        from . import _rng
        if self.n < m or m < 2:
            return np.zeros((0, m), dtype=np.int64)
        self.last = (self.batch, int(count), int(m))
        self.batch += 1
        return _rng.napsac_samples(self.key, self.last[0], int(count), self.n, int(m), self.off, self.idx)


def prosac_growth_function(n, m, t_n):
    """Chum & Matas' PROSAC growth function T'_n as USAC / GC-RANSAC tabulate it [UPSTREAM-MEMORY]: g[i] = number of
    samples after which the hypothesis-generation set grows beyond its i + 1 best points (g[i] = 1 for i < m)."""
    g = np.ones(n, dtype=np.int64)
    T_n = float(t_n)
    for i in range(m):
        T_n *= (m - i) / (n - i)
    tp = 1
    for i in range(n):
        if i + 1 <= m:
            g[i] = tp
            continue
        T_next = (i + 1) * T_n / (i + 1 - m)
        tp = tp + int(np.ceil(T_next - T_n))
        g[i] = tp
        T_n = T_next
    return g


class ProsacSampler(UniformSampler):
    """gcransac::sampler::ProsacSampler (progressivex_python.cpp:222) [UPSTREAM-MEMORY: the USAC formulation].  Points
    are assumed ordered by quality.  Sample number k (1-based, restarted by reset(), progressive_x.h:290) draws from the
    best n_k points, n_k = the smallest n >= m with T'_n >= k: m - 1 of the first n_k - 1 at random plus point n_k - 1
    itself; after `convergence_iterations` samples (GC-RANSAC's ransac_convergence_iterations, 100 000) it is uniform."""

    def __init__(self, n, rng, sample_size=None, convergence_iterations=100000):
        super().__init__(n, rng)
        self.prosac_m = sample_size          # findLines builds it with the HOMOGRAPHY sample size (quirk, :463)
        self.t_n = int(convergence_iterations)
        self._growth = {}

    def growth(self, m):
        if m not in self._growth:
            self._growth[m] = prosac_growth_function(self.n, m, self.t_n)
        return self._growth[m]

    def subset_sizes(self, first, count, m):
        """n_k for the sample numbers first .. first + count - 1"""
        k = np.arange(first, first + count, dtype=np.int64)
        return np.minimum(self.n, np.maximum(m, np.searchsorted(self.growth(m), k, side="left") + 1))

    def draw(self, count, m, first=1):
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        gm = max(m, min(self.n, self.prosac_m or m))
        k = np.arange(first, first + count, dtype=np.int64)
        nk = np.minimum(self.n, np.maximum(gm, np.searchsorted(self.growth(gm), k, side="left") + 1))
        out = np.empty((count, m), dtype=np.int64)
        if m > 1:
            out[:, :m - 1] = _distinct_rows(self.rng, nk - 1, m - 1)
        out[:, m - 1] = nk - 1
        late = k > self.t_n
        if late.any():
            out[late] = _distinct_rows(self.rng, np.full(int(late.sum()), self.n, dtype=np.int64), m)
        return out


class PhiloxProsacSampler(ProsacSampler):
    """ProsacSampler on the in-repo counter-based generator (csrc/rng.hip.h sample_prosac, _rng.prosac_samples): the growth
    function stays the host's table (a sequential floating-point recurrence; the sample numbers restart at 1 with every
    proposal, so the subset sizes of a draw are the same every time and go to the device once - pgx_sampler_prosac_set), the
    m - 1 random members of every sample come from (key, batch, sample) on the device."""
    kind = "prosac"

    def __init__(self, n, rng, sample_size=None, convergence_iterations=100000):
        super().__init__(n, rng, sample_size=sample_size, convergence_iterations=convergence_iterations)
        self.key = int(rng.integers(0, 2 ** 63))
        self.batch = 0
        self.last = None
        self.tops = None                             # subset size per sample of the latest draw (0 = uniform)

    def draw(self, count, m, first=1):
        from . import _rng
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        gm = max(m, min(self.n, self.prosac_m or m))
        k = np.arange(first, first + count, dtype=np.int64)
        nk = np.minimum(self.n, np.maximum(gm, np.searchsorted(self.growth(gm), k, side="left") + 1))
        self.tops = np.where(k > self.t_n, 0, nk).astype(np.int32)
        self.last = (self.batch, int(count), int(m))
        self.batch += 1
        return _rng.prosac_samples(self.key, self.last[0], int(count), self.n, int(m), self.tops)


class NapsacSampler(UniformSampler):
    """NAPSAC: first point uniform, the remaining m-1 from its neighbourhood ball; samples whose centre has fewer than
    m-1 neighbours are skipped (the reference's sampler fails and the RANSAC iteration is spent) [UPSTREAM-MEMORY]."""

    def __init__(self, n, rng, graph):
        super().__init__(n, rng)
        self.off, self.idx = np.asarray(graph[0], dtype=np.int64), np.asarray(graph[1], dtype=np.int64)

    def draw(self, count, m):
        centers = self.rng.integers(0, self.n, count)
        deg = self.off[centers + 1] - self.off[centers]
        ok = deg >= m - 1
        centers, deg = centers[ok], deg[ok]
        if len(centers) == 0:
            return np.zeros((0, m), dtype=np.int64)
        pick = _distinct_rows(self.rng, deg, m - 1)
        nbr = self.idx[self.off[centers][:, None] + pick]
        return np.column_stack([centers, nbr])


class ProgressiveNapsacSampler(UniformSampler):
    """gcransac::sampler::ProgressiveNapsacSampler<4>(&points, {16, 8, 4, 2}, m, {w1, h1, w2, h2}, 0.5)
    (progressivex_python.cpp:229-238) after Barath et al., "MAGSAC++ / Progressive NAPSAC" [UPSTREAM-MEMORY].  Points are
    assumed ordered by quality.

      * the centre of sample k comes from a one-point PROSAC sampler (its growth function with m = 1, T_N = n is
        T'_i = i + 1: sample k takes point k - 1, after n samples a uniformly random point);
      * every point p keeps its own hit counter and neighbourhood size s_p (start: m); s_p grows while
        hits_p > T''_{s_p}, T'' = the PROSAC growth function for m - 1 points and T_N = blend * n samples;
      * the neighbourhood is the cell of p in the finest of the grid layers (16, 8, 4, 2 cells per image dimension, 4-D
        cells over (x1, y1, x2, y2); members in index = quality order) that holds at least s_p points; the sample is p,
        the s_p-th member of the cell and m - 2 random ones among the members before it (PROSAC inside the cell);
        selected members get a hit;
      * a point whose coarsest cell is too small, and every sample after blend * n, is drawn by the global PROSAC
        sampler (sample number = the running count)."""

    def __init__(self, n, rng, pts, sizes, sample_size, layers=(16, 8, 4, 2), blend=0.5):
        super().__init__(n, rng)
        self.m = int(sample_size)
        self.max_local = int(blend * n)
        self.prosac = ProsacSampler(n, rng)
        pts = np.asarray(pts, dtype=np.float64)[:, :4]
        sizes = np.asarray(sizes, dtype=np.float64).reshape(-1)[:pts.shape[1]]
        self.cells = []          # per layer: (cell id per point, members per cell in index order)
        for div in layers:
            cell = np.clip(np.floor(pts / (sizes / div)), 0, div - 1).astype(np.int64)
            cid = np.zeros(n, dtype=np.int64)
            for d in range(cell.shape[1]):
                cid = cid * div + cell[:, d]
            order = np.argsort(cid, kind="stable")
            bounds = np.nonzero(np.diff(cid[order]))[0] + 1
            members = dict(zip(cid[order][np.concatenate([[0], bounds])].tolist(), np.split(order, bounds)))
            self.cells.append((cid, members))
        lm = max(self.m - 1, 1)
        self.growth_local = prosac_growth_function(n, lm, max(self.max_local, 1)) if n > lm else np.ones(n, dtype=np.int64)
        self.reset()

    def reset(self):
        self.hits = np.zeros(self.n, dtype=np.int64)
        self.subset = np.full(self.n, self.m, dtype=np.int64)
        self.layer = np.zeros(self.n, dtype=np.int64)

    def draw(self, count, m):
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        out = np.empty((count, m), dtype=np.int64)
        n_local = min(count, self.max_local)
        glob = np.zeros(count, dtype=bool)
        glob[n_local:] = True
        rng = self.rng
        late_centres = rng.integers(0, self.n, max(0, n_local - self.n))
        for k in range(n_local):
            p = k if k < self.n else int(late_centres[k - self.n])
            self.hits[p] += 1
            sp = int(self.subset[p])
            while sp < self.n and self.hits[p] > self.growth_local[sp - 1]:
                sp += 1
            self.subset[p] = sp
            lay = int(self.layer[p])
            nb = None
            while lay < len(self.cells):
                cid, members = self.cells[lay]
                nb = members[int(cid[p])]
                if len(nb) >= sp:
                    break
                lay += 1
                nb = None
            self.layer[p] = lay
            if nb is None:
                glob[k] = True
                continue
            others = nb[:sp]
            others = others[others != p]                 # the centre is part of its own cell
            if len(others) < m - 1:
                glob[k] = True
                continue
            last = others[-1]                            # "the farthest one" in PROSAC order: always part of the sample
            if m > 2:
                pick = others[:-1][rng.permutation(len(others) - 1)[:m - 2]]
                out[k, :m - 2] = pick
                self.hits[pick] += 1
            out[k, m - 2] = last
            self.hits[last] += 1
            out[k, m - 1] = p
        gi = np.nonzero(glob)[0]
        if len(gi):
            # kth sample number of the global PROSAC sampler = the running sample count (setSampleNumber)
            nk = self.prosac.subset_sizes(1, count, m)[gi]
            g = np.empty((len(gi), m), dtype=np.int64)
            if m > 1:
                g[:, :m - 1] = _distinct_rows(rng, nk - 1, m - 1)
            g[:, m - 1] = nk - 1
            out[gi] = g
        return out


class PhiloxProgressiveNapsacSampler(ProgressiveNapsacSampler):
    """ProgressiveNapsacSampler on the in-repo counter-based generator: the same statement with every random choice a function of
    (key, batch, sample number) - _rng.pnapsac_samples - and the sequential draw run by libpgx.so's host code (csrc/sampler_host.hip,
    pgx_pnapsac_*: ~0.1 us per sample against ~25 us of the interpreted loop above).  Opt-in (`sampler_rng="philox"`)."""

    def __init__(self, n, rng, pts, sizes, sample_size, layers=(16, 8, 4, 2), blend=0.5):
        UniformSampler.__init__(self, n, rng)
        from . import _lib
        self.m = int(sample_size)
        self.max_local = int(blend * n)
        self.prosac = ProsacSampler(n, rng)
        lm = max(self.m - 1, 1)
        self.growth_local = prosac_growth_function(n, lm, max(self.max_local, 1)) if n > lm else np.ones(n, dtype=np.int64)
        self.key = int(rng.integers(0, 2 ** 63))
        self.batch = 0
        self.native = _lib.PnapsacSampler(pts, sizes, self.m, layers) if n >= self.m else None

    def reset(self):
        pass                                         # (a draw starts from a fresh state by itself)

    def draw(self, count, m):
        if self.n < m or self.native is None:
            return np.zeros((0, m), dtype=np.int64)
        if m != self.m:
            raise ValueError("PhiloxProgressiveNapsacSampler: built for samples of %d points, asked for %d" % (self.m, m))
        tops = self.prosac.subset_sizes(1, count, m)
        batch, self.batch = self.batch, self.batch + 1
        return self.native.draw(self.key, batch, int(count), tops, self.growth_local, self.max_local)


# ---------------------------------------------------------------------------------------------------------------------
# sequential replay of the RANSAC loop over a scored batch
# ---------------------------------------------------------------------------------------------------------------------
def ransac_iteration_bound(inlier_number, n, sample_size, confidence):
    """Standard termination criterion used by GC-RANSAC [UPSTREAM-MEMORY]: log(1-conf) / log(1 - q^m)."""
    q = min(1.0, max(0.0, inlier_number / float(n)))
    qm = q ** sample_size
    if qm <= 0.0:
        return np.inf
    if qm >= 1.0:
        return 1.0
    den = np.log1p(-qm)
    return np.log(1.0 - confidence) / den if den < 0.0 else np.inf


def replay_sequential(counts, scores, iteration_of, n, sample_size, confidence, max_iters, min_iters=0):
    """Walks the scored hypotheses in generation order exactly as a sequential RANSAC would.

    Returns (best_index or -1, iteration_number, list of indices that became so-far-best in order)."""
    best, best_score, best_count = -1, -np.inf, 0
    bound = float(max_iters)
    history = []
    it_best = 0
    for h in range(len(counts)):
        it = int(iteration_of[h]) + 1
        if it > bound and it > min_iters:
            break
        c = int(counts[h])
        if c + 1 < best_count:          # scoring_function_with_compound_model.h:105-106 -> Score()
            continue
        s = float(scores[h])
        if c > 0 and s > best_score:    # strictly better => replaces the so-far-best
            best, best_score, best_count, it_best = h, s, c, it
            history.append(h)
            bound = min(float(max_iters), ransac_iteration_bound(c, n, sample_size, confidence))
    iterations = int(max(it_best, min(float(max_iters), np.ceil(bound)), 1))
    return best, iterations, history


# ---------------------------------------------------------------------------------------------------------------------
# the proposal engine
# ---------------------------------------------------------------------------------------------------------------------
# [U-15] experiment switches for the three hypotheses of docs/experiments-cubetoy.md §3 about what keeps upstream's proposal
# stage from accepting the "mixed" fundamental matrix first (all inside the absent graph-cut-ransac).
# Defaults = the shipped behaviour; scripts/exp_cubetoy.py flips them.  Not read from the environment, not part of the API.
U15 = {
    "rank": "value",          # "count": a so-far-best is ranked by inlier count first, MSAC value second
    "stop_at_first": False,   # True: the main loop ends at the first valid so-far-best found after min_iteration_number_before_lo
                              # iterations (the most aggressive early termination any confidence rule could produce)
}


# decision events of one proposal (ProposalEngine.run -> trace.walk): (code, a, b, c, x)
#   WK_BEST (h, iteration, inlier count, score)            a hypothesis became the so-far-best
#   WK_LO_ROUND (branch, candidates, updated, best score)  one round of the graph-cut local optimisation (branch 1: inner RANSAC of refits,
#                                                          2: one refit of all cut inliers, 0: too few inliers)
#   WK_LO_END (cuts so far, best count, LO runs, score)    a local optimisation returned
#   WK_WALK_END (iterations, best h or -1, LO runs, score) the main loop ended; `iterations` is what ProgressiveX::run is told
#   WK_LSQ (steps, 0, 0, score)                            the final iterated least squares
#   WK_FINAL (0, 0, 0, score)                              the score of the model handed back
WK_BEST, WK_LO_ROUND, WK_LO_END, WK_WALK_END, WK_LSQ, WK_FINAL = range(1, 7)


class ProposalEngine:
    """gcransac::GCRANSAC::run, restated [UPSTREAM-MEMORY, U-9] around ONE scoring launch per proposal:

        draw all samples -> solve them (GPU) -> score every hypothesis (GPU) -> walk the table in generation order:
          a hypothesis that beats the so-far-best score replaces it and tightens the iteration bound;
          if that happens after `min_iteration_number_before_lo` iterations the LOCAL OPTIMISATION runs right there
          (graph cut -> inner RANSAC of non-minimal refits -> one scoring launch, `_graph_cut_lo`) and its result, if
          better, is what later hypotheses have to beat - the cadence of the sequential loop (progressive_x.h:294-299
          with max_local_optimization_number = 50 inner trials, :68); the cuts of one proposal share the budget
          max_graph_cut_number (10), as upstream's statistics.graph_cut_number does;
        after the walk: one local optimisation if none ran, then the final iterated least squares (<= 10 refits)."""

    def __init__(self, ctx, estimator, pts, sampler, settings, exchange=None, trace=None):
        self.ctx, self.est, self.pts, self.sampler, self.s = ctx, estimator, pts, sampler, settings
        self.exchange = exchange     # parallel.RcclExchange for multi-GPU sharding, None = single GPU
        # diagnostics: a trace hook (pyprogressivex._engine) that ALSO has .walk(record) receives, per proposal, the score table the walk
        # saw, what every local-optimisation round / least-squares step got back from the cut and the refit solver, and the decisions
        # taken (WK_* below); tests/ replays the record through an independent restatement of the loop (oracle/progx_proposal.c)
        self.trace = trace if callable(getattr(trace, "walk", None)) else None
        self._rec = None
        self.n = pts.shape[0]
        self.lo_runs = 0             # statistics of the last proposal
        self.graph_cuts = 0
        self._prosac_table = None    # the PROSAC subset sizes resident on the device (PhiloxProsacSampler)

    def _score(self, models, T2, has_compound, exponent):
        if self.exchange is not None and self.exchange.world > 1:
            from . import parallel
            return parallel.score_sharded(self.exchange, models, T2, has_compound, exponent)
        return self.ctx.score(models, T2, has_compound=has_compound, exponent=exponent)

    def run(self, T2, has_compound, exponent, weights=None):
        """One GC-RANSAC-style proposal.  Returns dict(model, inliers (ascending indices), iterations) or None."""
        est, s = self.est, self.s
        self.sampler.reset()                                                   # progressive_x.h:290
        samples = self.sampler.draw(int(s.max_iteration_number), est.sample_size)
        if len(samples) == 0:
            return None
        sharded = self.exchange is not None and (self.exchange.world > 1 or getattr(self.ctx, "force_comm", False))
        model_of = None
        if est.device_minimal and not sharded:
            # hypotheses generated on the GPU from the resident points and scored where they are (no model upload);
            # a degenerate sample is a NaN model: never an inlier, never the winner
            philox = getattr(self.sampler, "last", None) if hasattr(self.ctx, "solve_minimal_sampled") else None
            if philox is not None and philox[1] == len(samples):
                # the batch is drawn on the device from (key, batch): the same rows as `samples` (tests), no index upload
                kind = getattr(self.sampler, "kind", "uniform")
                if kind == "prosac" and self._prosac_table is not self.sampler.tops:
                    if self._prosac_table is None or not np.array_equal(self._prosac_table, self.sampler.tops):
                        self.ctx.sampler_prosac_set(self.sampler.tops)         # (the same table every proposal: uploaded once)
                    self._prosac_table = self.sampler.tops
                models, _ = self.ctx.solve_minimal_sampled(self.sampler.key, philox[0], len(samples), sampler=kind)
            else:
                models = self.ctx.solve_minimal(samples)
            src = np.repeat(np.arange(len(samples), dtype=np.int64), est.device_slots)
            self.ctx.score_launch(T2, has_compound=has_compound)
            table = self.ctx.score_fetch(exponent)
        elif est.device_minimal:
            # multi-GPU: every rank solves and scores ITS slice of the (identical) sample list on its own GPU; one
            # all-gather of the (count, value, shared) triples gives every rank the whole table, and every rank walks it
            # the same way.  Models are not exchanged: the few the walk needs are re-solved from their samples (the device
            # solvers are deterministic, so every rank gets the bits the owning rank scored).
            from . import parallel
            ex = self.exchange
            shard, per, bounds = parallel.shard_samples(np.asarray(samples), ex.world, ex.rank)
            self.ctx.solve_minimal(shard, fetch=False)
            self.ctx.score_launch(T2, has_compound=has_compound)
            table = parallel.merge_sample_tables(ex.gather_scores(self.ctx, exponent), len(samples), ex.world, est.device_slots)
            src = np.repeat(np.arange(len(samples), dtype=np.int64), est.device_slots)
            models = None
            smp = np.asarray(samples)

            def model_of(h):
                s_idx, slot = divmod(int(h), est.device_slots)
                return np.asarray(self.ctx.solve_minimal(smp[s_idx:s_idx + 1])[slot], dtype=np.float64).copy()
        else:
            models, src = est.minimal(self.pts, samples)
            if len(models) == 0:
                return dict(model=None, inliers=np.zeros(0, np.int64), iterations=len(samples))
            table = self._score(models, T2, has_compound, exponent)
        counts = np.asarray(table["counts"], dtype=np.int64)
        scores = np.where(counts > 0, np.asarray(table["scores"], dtype=np.float64), -np.inf)
        scores = np.where(np.isnan(scores), -np.inf, scores)
        if U15["rank"] == "count":     # [U-15 i] count first, value second (a value never exceeds its count)
            scores = np.where(np.isfinite(scores), counts * float(self.n + 2) + scores, -np.inf)
        check = getattr(est, "validity", "off") != "off"                       # [U-14] model validity, both overloads
        smp_arr = np.asarray(samples)
        if check and models is not None:
            # isValidModel(model, data, sample, threshold): a hypothesis that fails is never scored upstream (`continue`)
            scores = np.where(est.valid_samples(self.pts, smp_arr, np.asarray(models), np.asarray(src)), scores, -np.inf)

        def rescore(cands):
            t = self.ctx.score(np.ascontiguousarray(cands), T2, has_compound=has_compound, exponent=exponent)
            return np.where(np.asarray(t["counts"]) > 0, np.asarray(t["scores"], dtype=np.float64), -np.inf)
        max_iters = float(s.max_iteration_number)
        min_iters = int(getattr(s, "min_iteration_number", 0))
        lo_after = int(getattr(s, "min_iteration_number_before_lo", 0))
        every_best = getattr(s, "lo_cadence", "every_best") == "every_best"
        self.lo_runs, self.graph_cuts = 0, 0
        rec = None
        if self.trace is not None and not check and self._use_graph_cut() and U15["rank"] == "value" and not U15["stop_at_first"]:
            rec = dict(n=int(self.n), samples=int(len(samples)), sample_size=int(est.sample_size),
                       nonminimal_sample_size=int(est.nonminimal_sample_size), confidence=float(s.confidence), max_iters=int(max_iters),
                       min_iters=min_iters, lo_after=lo_after, every_best=bool(every_best), max_cuts=int(getattr(s, "max_graph_cut_number", 10)),
                       lsq_budget=int(getattr(s, "max_least_squares_iterations", 10)), counts=counts.copy(), scores=scores.copy(),
                       src=np.asarray(src, dtype=np.int64).copy(), events=[], rounds=[], lsq=[])
        self._rec = rec
        model, best_score, best_count, it_best = None, -np.inf, 0, 0
        bound = max_iters
        h, H = 0, len(counts)
        ahead, ai = None, 0
        while h < H:
            # first strictly better score wins.  The hypotheses ahead that beat the so-far-best are listed once per so-far-best
            # (a scan per visited hypothesis was 40 % of a findTwoViewMotions call on the `book` scene: 30 000 hypotheses per
            # proposal, ~4 000 of them visited and passed over by the count test below); entries the walk masks on its way
            # (scores[h] = -inf) lie behind it.
            if ahead is None:
                ahead, ai = h + np.nonzero(scores[h:] > best_score)[0], 0
            if ai >= len(ahead):
                break
            h = int(ahead[ai])
            ai += 1
            it = int(src[h]) + 1
            if it > bound and it > min_iters:
                break
            c = int(counts[h])
            if c + 1 < best_count:          # scoring_function_with_compound_model.h:105-106 -> Score(): not considered
                h += 1
                continue
            cand = model_of(h) if model_of is not None else models[h].copy()
            cand_score, cand_count = float(scores[h]), c
            if check:
                smp = smp_arr[int(src[h])]
                if models is None and not est.valid_samples(self.pts, smp_arr, cand[None, :], np.asarray([int(src[h])]))[0]:
                    scores[h] = -np.inf
                    h += 1
                    continue
                # isValidModel(model, data, inliers, sample, threshold, updated): an invalid so-far-best is dropped
                ok, upd = est.valid_best(self.ctx, self.pts, cand, smp, float(s.inlier_outlier_threshold), rescore)
                if not ok:
                    scores[h] = -np.inf
                    h += 1
                    continue
                if upd is not cand:
                    one = self.ctx.score(upd[None, :], T2, has_compound=has_compound, exponent=exponent)
                    cand, cand_score, cand_count = upd, float(one["scores"][0]), int(one["counts"][0])
            model = cand
            best_score, best_count, it_best = cand_score, cand_count, it
            if rec is not None:
                rec["events"].append((WK_BEST, int(h), it, cand_count, cand_score))
            ahead = None                    # (also after the local optimisation below: it only raises best_score further)
            if every_best and it > lo_after and c > est.sample_size:
                model, best_score, best_count = self._local_optimization(model, best_score, best_count, T2, has_compound,
                                                                          exponent, weights)
            bound = min(max_iters, ransac_iteration_bound(best_count, self.n, est.sample_size, s.confidence))
            h += 1
            if U15["stop_at_first"] and it > lo_after:     # [U-15 ii]
                break
        iterations = int(max(it_best, min(max_iters, np.ceil(bound)), min(min_iters, len(samples)), 1))
        if rec is not None:
            last_best = [e[1] for e in rec["events"] if e[0] == WK_BEST]
            rec["events"].append((WK_WALK_END, iterations, last_best[-1] if last_best else -1, self.lo_runs, float(best_score)))
        if model is None:
            if rec is not None:
                self._rec = None
                self.trace.walk(rec)
            return dict(model=None, inliers=np.zeros(0, np.int64), iterations=iterations)
        if self.lo_runs == 0:               # "apply the local optimisation if it has not been applied yet"
            model, best_score, best_count = self._local_optimization(model, best_score, best_count, T2, has_compound,
                                                                      exponent, weights)
        if self._use_graph_cut():           # final iterated least squares on the inliers
            model, best_score = self._lsq_lo(model, best_score, T2, has_compound, exponent, weights,
                                             budget=int(getattr(s, "max_least_squares_iterations", 10)), final=True)
        final = self.ctx.score(model[None, :], T2, has_compound=has_compound, exponent=exponent, want_masks=True)
        if rec is not None:
            rec["events"].append((WK_FINAL, 0, 0, 0, float(final["scores"][0])))
            self._rec = None
            self.trace.walk(rec)
        return dict(model=model, inliers=self._mask_inliers(final), iterations=iterations,
                    score=float(final["scores"][0]))

    # -- local optimisation ------------------------------------------------------------------------------------------
    def _use_graph_cut(self):
        # "auto": GC-RANSAC's graph-cut local optimisation (with lambda = 0 the cut is plain thresholding: the inlier
        # mask); "lsq" forces the refit-only stand-in of round 1
        return getattr(self.s, "local_optimization", "auto") != "lsq"

    def _local_optimization(self, model, score, count, T2, has_compound, exponent, weights):
        self.lo_runs += 1
        if self._use_graph_cut():
            return self._graph_cut_lo(model, score, count, T2, has_compound, exponent, weights)
        m, sc = self._lsq_lo(model, score, T2, has_compound, exponent, weights, budget=int(self.s.max_local_optimization_number))
        if sc > score:
            one = self.ctx.score(m[None, :], T2, has_compound=has_compound, exponent=exponent)
            return m, sc, int(one["counts"][0])
        return model, score, count

    def _lsq_lo(self, model, score, T2, has_compound, exponent, weights, budget, final=False):
        """iterated least-squares refits on the inliers, scored with the same compound term: a refit is kept while the
        score improves (gcransac's iteratedLeastSquaresFitting [UPSTREAM-MEMORY])"""
        est = self.est
        rec = self._rec if final else None
        steps = 0
        while budget > 0:
            budget -= 1
            steps += 1
            one = self.ctx.score(model[None, :], T2, has_compound=has_compound, exponent=exponent, want_masks=True)
            inl = self._mask_inliers(one)
            if len(inl) < est.nonminimal_sample_size:
                if rec is not None:
                    rec["lsq"].append((len(inl), 0, 0, -np.inf))
                break
            fits = est.nonminimal(self.ctx, ("index", inl), weights, init=model)
            if len(fits) != 1:
                if rec is not None:
                    rec["lsq"].append((len(inl), len(fits), 0, -np.inf))
                break
            cand = self.ctx.score(fits[0][None, :], T2, has_compound=has_compound, exponent=exponent)
            if rec is not None:
                sc = float(cand["scores"][0])
                rec["lsq"].append((len(inl), 1, int(cand["counts"][0]), sc if sc == sc else -np.inf))
            if float(cand["scores"][0]) > score and int(cand["counts"][0]) > 0:
                model, score = np.asarray(fits[0], dtype=np.float64), float(cand["scores"][0])
            else:
                break
        if rec is not None:
            rec["events"].append((WK_LSQ, steps, 0, 0, float(score)))
        return model, score

    def _cut_inliers(self, model, T2, has_compound, exponent):
        """GCRANSAC::labeling: the inlier/outlier graph cut of `model`.  One exact min-cut on the GPU (pgx_gc_labeling)
        when 0 < lambda < 1; without a pairwise term the cut decouples into r^2 < T^2, i.e. the scorer's inlier mask."""
        lam = float(self.s.spatial_coherence_weight)
        if 0.0 < lam < 1.0:
            if hasattr(self.ctx, "gc_inliers"):      # the GPU context compacts the indices on the device
                return self.ctx.gc_inliers(model, T2, lam)
            return np.flatnonzero(self.ctx.gc_labeling(model, T2, lam) != 0).astype(np.int64)   # (bool scan: half the time of nonzero on int32)
        one = self.ctx.score(model[None, :], T2, has_compound=has_compound, exponent=exponent, want_masks=True)
        return self._mask_inliers(one)

    def _mask_inliers(self, table):
        """ascending inlier indices of the single hypothesis just scored with masks: compacted on the device when the context
        offers it (pgx_score_inliers), else unpacked from the mask row (the oracle-backed context of the tests) - the same set"""
        if hasattr(self.ctx, "score_inliers"):
            return self.ctx.score_inliers(0)
        return mask_to_indices(table["masks"][0], self.n)

    def _graph_cut_lo(self, model, score, count, T2, has_compound, exponent, weights):
        """gcransac::GCRANSAC::graphCutLocalOptimization, restated [UPSTREAM-MEMORY, U-12] and batched:

          repeat (while the proposal's graph-cut budget max_graph_cut_number = 10 lasts):
            inliers <- the inlier/outlier graph cut of the best model (`_cut_inliers`);
            inner RANSAC: max_local_optimization_number samples of min(7 * sample size, |inliers|) inliers, each refitted
            by the non-minimal solver (batched Gram pass on the GPU: pgx_gram_batch, all samples per launch); all candidates are scored in ONE launch and then walked in
            order - a strictly better score replaces the best model, exactly what the sequential loop would keep;
            stop when a round brought no improvement."""
        est, s = self.est, self.s
        rng = self.sampler.rng
        limit = 7 * est.sample_size
        trials = int(s.max_local_optimization_number)
        max_cuts = int(getattr(s, "max_graph_cut_number", 10))
        rec = self._rec
        while self.graph_cuts < max_cuts:
            self.graph_cuts += 1
            inl = self._cut_inliers(model, T2, has_compound, exponent)
            size = min(limit, len(inl))
            cands = []
            branch = 0
            if size < len(inl) and size >= est.nonminimal_sample_size:
                branch = 1
                picks = np.array([np.sort(rng.choice(inl, size, replace=False)) for _t in range(trials)])
                for fits in est.nonminimal_batch(self.ctx, picks, weights, init=model):   # one launch per refit step
                    cands.extend(fits)
            elif est.sample_size < len(inl) and len(inl) >= est.nonminimal_sample_size:
                branch = 2
                cands.extend(est.nonminimal(self.ctx, ("index", inl), weights, init=model))
            if not cands:
                if rec is not None:
                    rec["rounds"].append((len(inl), np.zeros(0, np.int64), np.zeros(0)))
                    rec["events"].append((WK_LO_ROUND, branch, 0, 0, float(score)))
                break
            cands = np.asarray(cands, dtype=np.float64)
            # (multi-GPU: every rank scores the few refits itself - replicas, no exchange)
            table = self.ctx.score(cands, T2, has_compound=has_compound, exponent=exponent)
            updated = False
            for h in range(len(cands)):
                if int(table["counts"][h]) > 0 and float(table["scores"][h]) > score:
                    model, score, count, updated = cands[h].copy(), float(table["scores"][h]), int(table["counts"][h]), True
            if rec is not None:
                tc = np.asarray(table["counts"], dtype=np.int64).copy()
                ts = np.asarray(table["scores"], dtype=np.float64).copy()
                rec["rounds"].append((len(inl), tc, np.where(np.isnan(ts), -np.inf, ts)))
                rec["events"].append((WK_LO_ROUND, branch, len(cands), int(updated), float(score)))
            if not updated:
                break
        if rec is not None:
            rec["events"].append((WK_LO_END, self.graph_cuts, int(count), self.lo_runs, float(score)))
        return model, score, count


def mask_to_indices(mask_row, n):
    bits = np.unpackbits(np.ascontiguousarray(mask_row).view(np.uint8), bitorder="little")[:n]
    return np.nonzero(bits)[0].astype(np.int64)
