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
  * local optimisation of the winner: GC-RANSAC's graph-cut + inner-RANSAC procedure (`_graph_cut_lo`, the cut runs on
    the GPU: pgx_gc_labeling) when the spatial coherence weight is in (0, 1); iterated least-squares refits on the
    inliers otherwise (with lambda = 0 the cut is plain thresholding).  Both are capped by
    max_local_optimization_number = 50 (progressive_x.h:68).
"""
import numpy as np


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
    for _ in range(retries):
        srt = np.sort(s, axis=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(axis=1) & ~dense
        if not bad.any():
            break
        s[bad] = (rng.random((int(bad.sum()), m)) * tops[bad][:, None]).astype(np.int64)
    srt = np.sort(s, axis=1)
    bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
    for r in bad:                             # exact: m steps of Fisher-Yates on range(tops[r]) with a sparse swap table
        top, swaps = int(tops[r]), {}
        for j in range(m):
            k = j + int(rng.integers(0, top - j))
            vj, vk = swaps.get(j, j), swaps.get(k, k)
            swaps[j], swaps[k] = vk, vj
            s[r, j] = vk
    return s


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


class ProsacSampler(UniformSampler):
    """PROSAC-style progressive sampling: points are assumed ordered by quality; sample t draws from the top-n_t
    prefix whose length grows linearly to n over `count` draws (simplified growth function) [UPSTREAM-MEMORY]."""

    def draw(self, count, m):
        if self.n < m:
            return np.zeros((0, m), dtype=np.int64)
        tops = np.minimum(self.n, np.maximum(m, (m + (self.n - m) * (np.arange(count) + 1) / count).astype(np.int64)))
        return _distinct_rows(self.rng, tops, m)


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


class ProgressiveNapsacSampler(NapsacSampler):
    """P-NAPSAC stand-in: NAPSAC blended linearly into global uniform sampling over the first 0.5 * n draws
    (the reference's blending length, progressivex_python.cpp:235) [UPSTREAM-MEMORY]."""

    def draw(self, count, m):
        local = super().draw(count, m)
        glob = UniformSampler.draw(self, count, m)
        blend = max(1.0, 0.5 * self.n)
        k = min(len(local), len(glob))
        use_global = self.rng.random(k) < np.minimum(1.0, np.arange(k) / blend)
        out = local[:k].copy()
        out[use_global] = glob[:k][use_global]
        return out


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
    return np.log(1.0 - confidence) / np.log(1.0 - qm)


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
class ProposalEngine:
    def __init__(self, ctx, estimator, pts, sampler, settings, exchange=None):
        self.ctx, self.est, self.pts, self.sampler, self.s = ctx, estimator, pts, sampler, settings
        self.exchange = exchange     # parallel.RcclExchange for multi-GPU sharding, None = single GPU
        self.n = pts.shape[0]

    def _score(self, models, T2, has_compound, exponent):
        if self.exchange is not None and self.exchange.world > 1:
            from . import parallel
            return parallel.score_sharded(self.exchange, models, T2, has_compound, exponent)
        return self.ctx.score(models, T2, has_compound=has_compound, exponent=exponent)

    def run(self, T2, has_compound, exponent, weights=None):
        """One GC-RANSAC-style proposal.  Returns dict(model, inliers (ascending indices), iterations) or None."""
        est, s = self.est, self.s
        self.sampler.reset()
        samples = self.sampler.draw(int(s.max_iteration_number), est.sample_size)
        if len(samples) == 0:
            return None
        on_device = est.device_minimal and (self.exchange is None or self.exchange.world == 1)
        if on_device:
            # hypotheses generated on the GPU from the resident points and scored where they are (no model upload);
            # a degenerate sample is a NaN model: never an inlier, never the winner
            models = self.ctx.solve_minimal(samples)
            src = np.repeat(np.arange(len(samples), dtype=np.int64), est.device_slots)
            self.ctx.score_launch(T2, has_compound=has_compound)
            table = self.ctx.score_fetch(exponent)
        else:
            models, src = est.minimal(self.pts, samples)
            if len(models) == 0:
                return dict(model=None, inliers=np.zeros(0, np.int64), iterations=len(samples))
            table = self._score(models, T2, has_compound, exponent)
        best, iters, history = replay_sequential(table["counts"], table["scores"], src, self.n, est.sample_size,
                                                 s.confidence, s.max_iteration_number)
        if best < 0:
            return dict(model=None, inliers=np.zeros(0, np.int64), iterations=iters)
        model, score = models[best].copy(), float(table["scores"][best])
        if self._use_graph_cut():
            model, score = self._graph_cut_lo(model, score, T2, has_compound, exponent, weights)
        else:
            model, score = self._lsq_lo(model, score, T2, has_compound, exponent, weights)
        final = self.ctx.score(model[None, :], T2, has_compound=has_compound, exponent=exponent, want_masks=True)
        return dict(model=model, inliers=mask_to_indices(final["masks"][0], self.n), iterations=iters,
                    score=float(final["scores"][0]))


    # -- local optimisation ------------------------------------------------------------------------------------------
    def _use_graph_cut(self):
        # "auto": the graph cut whenever it is not trivial (0 < lambda < 1); "lsq" forces the refit-only stand-in
        lam = float(self.s.spatial_coherence_weight)
        return getattr(self.s, "local_optimization", "auto") != "lsq" and 0.0 < lam < 1.0

    def _lsq_lo(self, model, score, T2, has_compound, exponent, weights):
        """iterated least-squares refits on the inliers, scored with the same compound term (the stand-in used when the
        spatial coherence weight is 0 and the graph cut degenerates to thresholding)"""
        est, s = self.est, self.s
        lo_budget = int(s.max_local_optimization_number)
        while lo_budget > 0:
            lo_budget -= 1
            one = self.ctx.score(model[None, :], T2, has_compound=has_compound, exponent=exponent, want_masks=True)
            inl = mask_to_indices(one["masks"][0], self.n)
            if len(inl) < est.nonminimal_sample_size:
                break
            fits = est.nonminimal(self.ctx, ("index", inl), weights, init=model)
            if len(fits) != 1:
                break
            cand = self.ctx.score(fits[0][None, :], T2, has_compound=has_compound, exponent=exponent)
            if float(cand["scores"][0]) > score and int(cand["counts"][0]) > 0:
                model, score = np.asarray(fits[0], dtype=np.float64), float(cand["scores"][0])
            else:
                break
        return model, score

    def _graph_cut_lo(self, model, score, T2, has_compound, exponent, weights):
        """gcransac::GCRANSAC::graphCutLocalOptimization, restated [UPSTREAM-MEMORY, U-12] and batched:

          repeat (at most max_graph_cut_number = 10 times):
            inliers <- the inlier/outlier graph cut of the best model (one exact min-cut on the GPU, pgx_gc_labeling);
            inner RANSAC: max_local_optimization_number samples of min(7 * sample size, |inliers|) inliers, each refitted
            by the non-minimal solver (batched Gram pass on the GPU: pgx_gram_batch, all samples per launch); all candidates are scored in ONE launch and then walked in
            order - a strictly better score replaces the best model, exactly what the sequential loop would keep;
            stop when a round brought no improvement."""
        est, s = self.est, self.s
        lam = float(s.spatial_coherence_weight)
        rng = self.sampler.rng
        limit = 7 * est.sample_size
        trials = int(s.max_local_optimization_number)
        for _ in range(int(getattr(s, "max_graph_cut_number", 10))):
            flags = self.ctx.gc_labeling(model, T2, lam)
            inl = np.nonzero(flags)[0].astype(np.int64)
            size = min(limit, len(inl))
            cands = []
            if size < len(inl) and size >= est.nonminimal_sample_size:
                picks = np.array([np.sort(rng.choice(inl, size, replace=False)) for _t in range(trials)])
                for fits in est.nonminimal_batch(self.ctx, picks, weights, init=model):   # one launch per refit step
                    cands.extend(fits)
            elif est.sample_size < len(inl) and len(inl) >= est.nonminimal_sample_size:
                cands.extend(est.nonminimal(self.ctx, ("index", inl), weights, init=model))
            if not cands:
                break
            cands = np.asarray(cands, dtype=np.float64)
            table = self.ctx.score(cands, T2, has_compound=has_compound, exponent=exponent)
            updated = False
            for h in range(len(cands)):
                if int(table["counts"][h]) > 0 and float(table["scores"][h]) > score:
                    model, score, updated = cands[h].copy(), float(table["scores"][h]), True
            if not updated:
                break
        return model, score


def mask_to_indices(mask_row, n):
    bits = np.unpackbits(np.ascontiguousarray(mask_row).view(np.uint8), bitorder="little")[:n]
    return np.nonzero(bits)[0].astype(np.int64)
