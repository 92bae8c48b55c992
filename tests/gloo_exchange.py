"""GlooExchange - CPU stand-in for pyprogressivex.parallel.RcclExchange in the world_size-2 tests of the sharded host logic:
the same two methods, the exchange over torch.distributed / gloo.  TEST INFRASTRUCTURE (torch is not a dependency of the
product package)."""
import numpy as np


class GlooExchange:
    def __init__(self, world, rank, scorer=None):
        self.scorer, self.world, self.rank = scorer, world, rank

    def _gather(self, local):
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return local
        out = {}
        for key in ("counts", "values", "shared", "scores"):
            t = torch.from_numpy(np.ascontiguousarray(local[key]))
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t)
            out[key] = torch.cat(parts).numpy()
        return out

    def gather_scores(self, ctx, exponent):
        """after ctx.score_launch on every rank's shard (ctx: any context with score_fetch, e.g. the oracle-backed one)"""
        return self._gather(ctx.score_fetch(exponent))

    def score_shard(self, shard, T2, has_compound, exponent):
        return self._gather(self.scorer(shard, T2, has_compound, exponent))

    # -- the pipelined interface of RcclExchange (two batches in flight): begin() scores and starts an ASYNCHRONOUS all_gather,
    #    end() waits for that slot only
    def begin(self, slot, shard, T2, has_compound):
        import torch
        import torch.distributed as dist
        local = self.scorer(shard, T2, has_compound, self._exponent)
        self._inflight = getattr(self, "_inflight", {})
        assert slot not in self._inflight, "slot still in flight"
        work = {}
        for key in ("counts", "values", "shared", "scores"):
            t = torch.from_numpy(np.ascontiguousarray(local[key]))
            parts = [torch.empty_like(t) for _ in range(self.world)]
            h = dist.all_gather(parts, t, async_op=True) if self.world > 1 else None
            work[key] = (h, parts if self.world > 1 else [t])
        self._inflight[slot] = work

    def end(self, slot, exponent):
        import torch
        work = self._inflight.pop(slot)
        out = {}
        for key, (h, parts) in work.items():
            if h is not None:
                h.wait()
            out[key] = torch.cat(parts).numpy()
        return out

    _exponent = 2

    # -- point-sharded jobs (parallel.score_point_sharded): acc_scorer(models, T2, has_compound) -> integer accumulators of THIS
    #    rank's slice (helpers.fixed_point_accumulators: the oracle's residuals in the score path's fixed point); all_reduce(sum)
    acc_scorer = None
    n_total = 0

    def reduce_scores(self, models, T2, has_compound, exponent):
        import torch
        import torch.distributed as dist
        from pyprogressivex import parallel
        acc = self.acc_scorer(models, T2, has_compound)
        red = {}
        for key in ("counts", "values_q", "shared_q"):
            t = torch.from_numpy(np.ascontiguousarray(acc[key]).astype(np.int64))
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            red[key] = t.numpy()
        return parallel.table_from_accumulators(red["counts"], red["values_q"], red["shared_q"], self.n_total, has_compound, exponent)
