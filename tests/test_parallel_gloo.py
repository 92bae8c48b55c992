"""N > 1 host logic on CPU: world_size = 2 over torch.distributed/gloo.  Each rank scores its shard of the hypothesis
batch (scorer injected: the oracle), the all-gather + merge must give every rank the full table in global hypothesis
order, identical to the single-process result, and both ranks must select the same winner."""
import os
import socket
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(r"{root}", "progressive-x_amd"), os.path.join(r"{root}", "oracle"), r"{here}"]
import torch.distributed as dist
import pgx_oracle as O
from pyprogressivex import parallel
from helpers import make_case
from gloo_exchange import GlooExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mt, pts, models, thr = make_case("pnp", 3000, 37, seed=4)      # 37 hypotheses: ragged shards (19 + 18)
T2 = 2.25 * thr * thr
comp = np.linspace(0, 1, 3000)
def scorer(shard, T2, has_compound, exponent):
    return O.score(mt, pts, shard, T2, compound=comp, has_compound=has_compound, exponent=exponent)
ex = GlooExchange(world, rank, scorer)
table = parallel.score_sharded(ex, models, T2, has_compound=True, exponent=2)
full = O.score(mt, pts, models, T2, compound=comp, has_compound=True, exponent=2)
for k in ("counts", "values", "shared", "scores"):
    assert np.array_equal(table[k], full[k]), k
best = parallel.select_best(table["scores"], table["counts"])
assert best == parallel.select_best(full["scores"], full["counts"])
shard, lo, hi = parallel.shard_hypotheses(models, world, rank)
assert shard.shape[0] == 19 and np.isnan(shard[hi - lo:]).all()
# two batches in flight (asynchronous all_gather per piece, the next piece scored meanwhile): bitwise the serial exchange
serial = ex.score_shard(shard, T2, True, 2)
for pieces in (2, 3, 19):
    piped = parallel.score_shard_pipelined(ex, shard, T2, True, 2, pieces=pieces)
    for k in ("counts", "values", "shared", "scores"):
        assert np.array_equal(piped[k], serial[k], equal_nan=True), (pieces, k)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok best", best)
'''


def test_sharded_scoring_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, here=HERE))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok best" in o for o in outs)


WORKER_E2E = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(r"{root}", "progressive-x_amd"), os.path.join(r"{root}", "oracle"), r"{here}"]
import torch.distributed as dist
from pyprogressivex import _engine, _estimators, _proposal, datasets
from oracle_ctx import OracleContext
from gloo_exchange import GlooExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

def run(kind, exchange):
    if kind == "homography":
        pts, gt, _ = datasets.make_homographies(n_per_plane=200, n_planes=3, n_outliers=300, seed=2)
        est, thr, lam = _estimators.HomographyEstimator(), 3.0, 0.0
    else:
        x1, x2, K, gt, poses = datasets.make_poses(n_per_object=300, n_objects=2, n_outliers=200, seed=3)
        pts, f = datasets.normalize_pnp(x1, x2, K)
        est, thr, lam = _estimators.PnPEstimator(), 4.0 / f, 0.0
    s = _engine.MultiModelSettings()
    s.minimum_number_of_inliers = 30
    s.inlier_outlier_threshold = thr
    s.set_confidence(0.99)
    s.spatial_coherence_weight = lam
    s.max_iteration_number = 301                      # odd: ragged sample shards (151 + 150)
    ctx = OracleContext()
    rng = np.random.default_rng(5)                    # the same seed on every rank
    px = _engine.ProgressiveX(ctx, est, pts, None, _proposal.UniformSampler(len(pts), rng), s, exchange=exchange)
    models, st = px.run()
    return np.array([m.descriptor for m in models]), np.asarray(st.labeling)

for kind in ("homography", "pnp"):
    m1, l1 = run(kind, None)                                  # world 1: the whole batch on this "GPU"
    m2, l2 = run(kind, GlooExchange(world, rank))             # world 2: half of the samples per rank + all-gather
    assert len(m1) >= 2 and m1.shape == m2.shape and np.array_equal(m1, m2), kind
    assert np.array_equal(l1, l2), kind
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok e2e")
'''


def test_progressive_x_sharded_over_two_ranks_equals_one_rank(tmp_path):
    """ProposalEngine.run / ProgressiveX.run end to end with the proposal batches sharded over world_size = 2 (every rank
    solves and scores its slice of the samples, gloo all-gather of the score triples): models and labelling are BITWISE
    those of the unsharded run, on both ranks."""
    script = tmp_path / "worker_e2e.py"
    script.write_text(WORKER_E2E.format(root=ROOT, here=HERE))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok e2e" in o for o in outs)


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    from pyprogressivex import parallel
    monkeypatch.setenv("PGX_RDV_DIR", str(tmp_path))
    uid = bytes(range(128))
    assert parallel.exchange_unique_id(0, 2, lambda: uid) == uid
    assert parallel.exchange_unique_id(1, 2, lambda: b"", timeout=5) == uid
    parallel.cleanup_unique_id(0)
    assert not list(tmp_path.iterdir())


WORKER_POINTS = r'''
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(r"{root}", "progressive-x_amd"), os.path.join(r"{root}", "oracle"), r"{here}"]
import torch.distributed as dist
import pgx_oracle as O
from pyprogressivex import parallel
from helpers import make_case, fixed_point_accumulators
from gloo_exchange import GlooExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mt, pts, models, thr = make_case("pnp", 3001, 37, seed=4)      # 3001 points: ragged slices (1501 + 1500)
T2 = 2.25 * thr * thr
n = pts.shape[0]
comp = np.linspace(0, 1, n)
lo, hi = parallel.point_slice(n, world, rank)
assert (lo, hi) == ((0, 1501) if rank == 0 else (1501, 3001))
ex = GlooExchange(world, rank)
ex.n_total = n
ex.acc_scorer = lambda mdl, T2, hc: fixed_point_accumulators(O, mt, pts[lo:hi], mdl, T2, comp[lo:hi] if hc else None, n_total=n)
table = parallel.score_point_sharded(ex, models, T2, has_compound=True, exponent=2)
# the sum over the slices IS the unsharded accumulation (integers): bitwise
one = fixed_point_accumulators(O, mt, pts, models, T2, comp, n_total=n)
ref = parallel.table_from_accumulators(one["counts"], one["values_q"], one["shared_q"], n, True, 2)
for k in ("counts", "values", "shared", "scores"):
    assert np.array_equal(table[k], ref[k]), k
# ... and it is the oracle's table: counts exactly, sums to the quantisation of the fixed point
full = O.score(mt, pts, models, T2, compound=comp, has_compound=True, exponent=2)
assert np.array_equal(table["counts"], full["counts"])
for k in ("values", "shared"):
    assert np.max(np.abs(table[k] - full[k]) / np.maximum(np.abs(full[k]), 1e-300)) <= 1e-9, k
assert parallel.select_best(table["scores"], table["counts"]) == parallel.select_best(full["scores"], full["counts"])
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok points")
'''


def test_point_sharded_scoring_world2_gloo(tmp_path):
    """parallel.score_point_sharded: every rank scores all hypotheses against its slice of the points, integer accumulators
    all-reduced (sum): bitwise the unsharded table."""
    script = tmp_path / "worker_points.py"
    script.write_text(WORKER_POINTS.format(root=ROOT, here=HERE))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok points" in o for o in outs)
