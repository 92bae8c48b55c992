"""StubContext — the slice of pyprogressivex._lib.Context that bench.py's main() drives, answered by the CPU oracle, with
torch.distributed/gloo standing in for RCCL.  TEST INFRASTRUCTURE (tests/test_bench_multirank.py): runs bench.py's multi-rank
control flow and JSON arithmetic on a box without a GPU.  Loaded only through the PGX_BENCH_STUB test hook."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "oracle")]
import pgx_oracle as O  # noqa: E402


def init_comm(ctx, rank, world):
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx.nranks, ctx.rank = world, rank


class StubContext:
    def __init__(self, device_id=0):
        self.device_id = device_id
        self.nranks, self.rank = 1, 0
        self.global_n = 0
        self.slots = {}
        self.pending = {}

    def device_info(self):
        return dict(name="stub (CPU oracle + gloo)", cu_count=0)

    def score_profile(self, mode):
        pass

    def set_points(self, mt, pts):
        self.mt, self.pts = mt, np.ascontiguousarray(pts, dtype=np.float64)
        self.n = self.pts.shape[0]
        self.comp = np.zeros(self.n)

    def score_set_global_n(self, n):
        self.global_n = int(n)

    def preference(self, model, T2, slot, want_pref=False):
        self.slots[slot] = O.preference(self.mt, self.pts, model, T2)

    def compound_update(self, slots, want_compound=False):
        self.comp = O.compound_max(np.stack([self.slots[s] for s in slots]))

    def get_compound(self):
        return self.comp.copy()

    def score_upload(self, hyps):
        self.hyps = np.ascontiguousarray(hyps, dtype=np.float64)
        self.M = self.hyps.shape[0]

    def score_buffers(self):
        return None

    def score_launch(self, T2, has_compound=False, want_masks=False):
        self.res = O.score(self.mt, self.pts, self.hyps, T2, compound=self.comp, has_compound=has_compound, exponent=2)

    def _table(self, counts, values, shared, exponent):
        return dict(counts=counts, values=values, shared=shared, scores=values - np.power(shared, float(exponent)))

    def score_fetch(self, exponent=2, out=None, want_masks=False):
        return self._table(self.res["counts"], self.res["values"], self.res["shared"], exponent)

    def score_kernel_times(self):
        return np.array([0.01, 0.1, 0.01, 0.0])     # (cull, group-major, finish, exact queue) ms: placeholders, the line says "stub"

    # point-sharded exchange: sums over the ranks (the product adds integer accumulators: exact; doubles here - the stub checks flow, not bits)
    def score_allreduce_begin(self, slot):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.concatenate([self.res["counts"].astype(np.float64), self.res["values"], self.res["shared"]]))
        self.pending[slot] = (t, dist.all_reduce(t, async_op=True))

    def score_allreduce_end(self, slot, exponent=2):
        t, work = self.pending.pop(slot)
        work.wait()
        a = t.numpy()
        M = self.M
        return self._table(np.rint(a[:M]).astype(np.int64), a[M:2 * M].copy(), a[2 * M:].copy(), exponent)

    # hypothesis-sharded exchange: rank-major rows of every rank's shard
    def score_allgather_begin(self, slot):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.concatenate([self.res["counts"].astype(np.float64), self.res["values"], self.res["shared"]]))
        outs = [torch.empty_like(t) for _ in range(self.nranks)]
        self.pending[slot] = (outs, dist.all_gather(outs, t, async_op=True))

    def score_allgather_end(self, slot, exponent=2):
        outs, work = self.pending.pop(slot)
        work.wait()
        M = self.M
        a = [o.numpy() for o in outs]
        return self._table(np.concatenate([np.rint(x[:M]).astype(np.int64) for x in a]), np.concatenate([x[M:2 * M] for x in a]),
                           np.concatenate([x[2 * M:] for x in a]), exponent)

    def comm_barrier(self):
        import torch.distributed as dist
        dist.barrier()

    def comm_allreduce_max(self, x):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def comm_destroy(self):
        import torch.distributed as dist
        dist.destroy_process_group()

    def sync(self):
        pass

    def close(self):
        pass

    def score_algorithmic_bytes(self):
        d, p = O.POINT_DIM[self.mt], O.PARAM_DIM[self.mt]
        return self.n * d * 8 + self.M * p * 8 + self.M * 16 + self.n * 8, self.n * self.M

    def score_stats(self, T2, has_compound=False):
        inl = int(self.res["counts"].sum())
        return dict(pairs=self.n * self.M, group_pairs=0, surviving_group_steps=0, exact_evaluations=self.n * self.M, inlier_pairs=inl)
