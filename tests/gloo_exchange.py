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
