"""Multi-GPU sharding of the proposal loop — one process per GPU.

Hypotheses are independent given (points, compound preference vector, T2): a batch of M hypotheses is split into
contiguous shards, every rank scores its shard against ALL points on its own GPU, and one all-gather of the
per-hypothesis (count, value, shared) triples (24 B each) makes the full score table available on every rank, which
then runs the same deterministic selection (ties -> lowest hypothesis index, i.e. the sequential "first best wins").
The reference has no counterpart (single-threaded CPU: progressive_x.h:251-489, no collectives anywhere).

The other split (`score_point_sharded`, north_star's "all-reduce of per-model inlier counts"): every rank holds a SLICE of the
points and scores the whole batch against it; the launch's integer accumulators (count, 2^-q fixed-point sums) are added over
the ranks by an RCCL all-reduce - exact in any order, so the table is bitwise the single-GPU one, and cull / dispatch / group
work all divide by the number of ranks (the BASELINE metric's strong-scaling mode, bench.py --scaling strong-points).

Data plane: RCCL all-gather inside libpgx.so (pgx_score_allgather), bootstrapped here with a file rendezvous for the
ncclUniqueId (single node, which is what the launch contract covers).  For CPU tests of this host logic the exchange
runs over torch.distributed/gloo (tests/gloo_exchange.py, a test double outside this package).
"""
import os
import time

import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# sharding / merging (pure host logic, backend independent)
# ---------------------------------------------------------------------------------------------------------------------
def shard_bounds(M, world):
    """Contiguous shards of equal padded length: (per_rank, [(lo, hi)] * world)."""
    per = (M + world - 1) // world
    return per, [(min(M, r * per), min(M, (r + 1) * per)) for r in range(world)]


def shard_hypotheses(models, world, rank):
    """This rank's shard, padded with NaN models (a NaN model never has an inlier) to the common shard length."""
    models = np.ascontiguousarray(models, dtype=np.float64)
    M, P = models.shape
    per, bounds = shard_bounds(M, world)
    lo, hi = bounds[rank]
    shard = np.full((per, P), np.nan, dtype=np.float64)
    shard[: hi - lo] = models[lo:hi]
    return shard, lo, hi


def merge_gathered(gathered, M, world):
    """rank-major gathered arrays of length world*per -> the first M entries in global hypothesis order."""
    per, bounds = shard_bounds(M, world)
    out = {}
    for key, arr in gathered.items():
        arr = np.asarray(arr).reshape(world, per)
        out[key] = np.concatenate([arr[r, : hi - lo] for r, (lo, hi) in enumerate(bounds)])
    return out


def select_best(scores, counts):
    """Deterministic winner: highest score, ties -> lowest index; hypotheses without inliers never win."""
    scores = np.where(np.asarray(counts) > 0, np.asarray(scores, dtype=np.float64), -np.inf)
    scores = np.where(np.isnan(scores), -np.inf, scores)
    best = int(np.argmax(scores))
    return best if np.isfinite(scores[best]) else -1


# ---------------------------------------------------------------------------------------------------------------------
# rank environment + file rendezvous for the RCCL unique id
# ---------------------------------------------------------------------------------------------------------------------
def rank_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def _launcher_key():
    """Same value in every worker of one torchrun launch on this node, different between launches."""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: starttime
    except OSError:
        pass
    return f"{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"


def _rendezvous_dir():
    """A directory only this user can write: $PGX_RDV_DIR, else $XDG_RUNTIME_DIR, else /tmp/pgx_rdv_<uid> created 0700.
    Refuses a directory owned by somebody else or writable by group/others (another local user could pre-create the id
    file and hang the bootstrap or join the ranks to a foreign communicator)."""
    import stat

    def usable(path):
        try:
            st = os.stat(path)
        except OSError:
            return False
        return stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & (stat.S_IWGRP | stat.S_IWOTH))

    explicit = os.environ.get("PGX_RDV_DIR")
    if explicit:
        if not usable(explicit):
            raise PermissionError(f"PGX_RDV_DIR={explicit} must exist, be owned by uid {os.getuid()} and not be group/world writable")
        return explicit
    xdg = os.environ.get("XDG_RUNTIME_DIR")
    if xdg and usable(xdg):          # often set but absent inside containers: then fall through
        return xdg
    base = os.path.join("/tmp", f"pgx_rdv_{os.getuid()}")
    os.makedirs(base, mode=0o700, exist_ok=True)
    if not usable(base):
        raise PermissionError(f"rendezvous directory {base} must be owned by uid {os.getuid()} and not group/world writable")
    return base


def _launcher_start_time():
    """Wall-clock start of the launcher (the common parent of all ranks); files older than this are stale."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(float(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, StopIteration, ValueError):
        return 0.0


def _rendezvous_path():
    return os.path.join(_rendezvous_dir(), f"pgx_rdv_{_launcher_key()}.id")


def exchange_unique_id(rank, world, make_id, timeout=300.0):
    """rank 0 creates the 128-byte ncclUniqueId and publishes it through an atomically renamed file in a private
    directory; readers accept it only if this user wrote it after the launcher started."""
    if world == 1:
        return make_id()
    path = _rendezvous_path()
    if rank == 0:
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    not_before = _launcher_start_time() - 2.0
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            with open(path, "rb") as f:
                st = os.fstat(f.fileno())
                uid = f.read()
            if len(uid) == 128 and st.st_uid == os.getuid() and st.st_mtime >= not_before:
                return uid
        except OSError:
            pass
        time.sleep(0.05)
    raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout}s")


def cleanup_unique_id(rank):
    if rank == 0:
        try:
            os.remove(_rendezvous_path())
        except OSError:
            pass


def init_rccl(ctx, rank, world):
    """Binds `ctx` (a _lib.Context on this rank's GPU) into the node-wide RCCL communicator."""
    import sys
    from . import _lib
    # RCCL prints a version banner on stdout when it initialises; callers (bench.py) reserve stdout for one JSON line,
    # so the C-level stdout is pointed at stderr for the duration of the bootstrap.
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        uid = exchange_unique_id(rank, world, _lib.comm_unique_id)
        ctx.comm_init(world, rank, uid)
        ctx.comm_barrier()
    finally:
        try:   # the banner sits in the C stdio buffer (a pipe is block-buffered): push it out while fd 1 is still stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)
    cleanup_unique_id(rank)


# ---------------------------------------------------------------------------------------------------------------------
# sharded scoring
# ---------------------------------------------------------------------------------------------------------------------
class RcclExchange:
    """Data plane on the GPUs: libpgx's RCCL all-gather over xGMI."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.world, self.rank = ctx.nranks, ctx.rank

    def gather_scores(self, ctx, exponent):
        """After ctx.score_launch on every rank's shard: the rank-major table of all shards, on every rank."""
        if self.world == 1 and not getattr(ctx, "force_comm", False):
            return ctx.score_fetch(exponent)
        ctx.score_allgather()
        return ctx.score_fetch_all(exponent)

    def score_shard(self, shard, T2, has_compound, exponent):
        self.ctx.score_upload(shard)
        self.ctx.score_launch(T2, has_compound=has_compound)
        return self.gather_scores(self.ctx, exponent)

    # -- two batches in flight: the exchange of one overlaps the scoring of the next ------------------------------------------
    def begin(self, slot, shard, T2, has_compound):
        """upload + launch `shard`, start its exchange in `slot` (0 / 1) and return at once"""
        self.ctx.score_upload(shard)
        self.ctx.score_launch(T2, has_compound=has_compound)
        self.ctx.score_allgather_begin(slot)

    def end(self, slot, exponent):
        """the rank-major table of the batch that was begun in `slot`"""
        return self.ctx.score_allgather_end(slot, exponent)

    # -- point-sharded: the context holds this rank's slice of the points (score_point_sharded) -------------------------------
    def reduce_scores(self, models, T2, has_compound, exponent):
        """all hypotheses against this rank's points, accumulators summed over the ranks: the job's table, on every rank"""
        self.ctx.score_upload(models)
        self.ctx.score_launch(T2, has_compound=has_compound)
        if self.world > 1 or getattr(self.ctx, "force_comm", False):
            self.ctx.score_allreduce()
        return self.ctx.score_fetch(exponent)

    def begin_reduce(self, slot, models, T2, has_compound):
        self.ctx.score_upload(models)
        self.ctx.score_launch(T2, has_compound=has_compound)
        self.ctx.score_allreduce_begin(slot)

    def end_reduce(self, slot, exponent):
        return self.ctx.score_allreduce_end(slot, exponent)


def score_shard_pipelined(exchange, shard, T2, has_compound, exponent, pieces=2):
    """`exchange.score_shard(shard, ...)` with the shard cut into `pieces` consecutive parts whose exchanges overlap the scoring
    of the following part (`begin` / `end`, two in flight).  Every hypothesis is scored independently of its batch, so the
    rank-major table is bitwise the one `score_shard` returns."""
    per = shard.shape[0]
    pieces = max(1, min(int(pieces), per))
    if pieces == 1 or not hasattr(exchange, "begin"):
        return exchange.score_shard(shard, T2, has_compound, exponent)
    cuts = [(per * k) // pieces for k in range(pieces + 1)]
    tables = []
    open_slots = []          # begun, not yet collected: drained on the way out of an exception, or every later call would find them busy
    try:
        for k in range(pieces):
            exchange.begin(k & 1, np.ascontiguousarray(shard[cuts[k]:cuts[k + 1]]), T2, has_compound)
            open_slots.append(k & 1)
            if k >= 1:
                open_slots.remove((k - 1) & 1)
                tables.append(exchange.end((k - 1) & 1, exponent))
        open_slots.remove((pieces - 1) & 1)
        tables.append(exchange.end((pieces - 1) & 1, exponent))
    finally:
        for slot in open_slots:
            try:
                exchange.end(slot, exponent)
            except Exception:
                pass
    out = {}
    for key in ("counts", "values", "shared", "scores"):      # [rank][piece rows] per piece -> [rank][all rows of the rank]
        parts = [np.asarray(t[key]).reshape(exchange.world, cuts[k + 1] - cuts[k]) for k, t in enumerate(tables)]
        out[key] = np.concatenate(parts, axis=1).reshape(-1)
    return out


_process_exchange = None


def default_exchange(ctx, distributed=None):
    """The exchange the drop-in API uses.  Sharding is OPT-IN: `distributed=True` on the call, or PGX_MULTI_GPU=1 in the
    environment; a torch.distributed.run job whose ranks call the API independently (different data, different call counts)
    gets None even though WORLD_SIZE > 1 - every proposal runs a collective, so ranks that do not make the same calls on the
    same data would deadlock or merge unrelated score tables.  With the opt-in and WORLD_SIZE > 1 (one process per GPU, any
    launcher that sets RANK / LOCAL_RANK / WORLD_SIZE) the node-wide RCCL communicator is created once per process; the ranks
    must sit on ONE node (the unique id travels through a local file).  PGX_FORCE_COMM=1 puts RCCL on the path with a single
    rank (tests, bench leg)."""
    global _process_exchange
    rank, world, _ = rank_env()
    force = os.environ.get("PGX_FORCE_COMM") == "1"
    want = distributed if distributed is not None else os.environ.get("PGX_MULTI_GPU") == "1"
    if not force and (world == 1 or not want):
        return None
    local_world = os.environ.get("LOCAL_WORLD_SIZE")
    nnodes = os.environ.get("GROUP_WORLD_SIZE") or os.environ.get("NNODES")
    if (local_world is not None and int(local_world) != world) or (nnodes is not None and nnodes.isdigit() and int(nnodes) > 1):
        raise RuntimeError(f"pyprogressivex: hypothesis sharding needs all {world} ranks on one node (LOCAL_WORLD_SIZE = {local_world}): "
                           "the RCCL unique id is exchanged through a node-local file")
    if _process_exchange is None or _process_exchange.ctx is not ctx:
        if getattr(ctx, "nranks", 1) != world or not getattr(ctx, "_comm_ready", False):
            init_rccl(ctx, rank, world)
            ctx._comm_ready = True
        ctx.force_comm = force
        _process_exchange = RcclExchange(ctx)
    return _process_exchange


def check_same_problem(exchange, pts, params=None):
    """Every rank of a sharded call must hold the same points AND make the same call: max and min over the ranks of a 52-bit
    digest of (shape, bytes, repr of the call's scalar parameters - thresholds, iteration limits, sampler, seed ...) must agree
    (ranks that differ in any of them run different numbers of collectives).  Raises on every rank alike (the reduction is
    collective), before any proposal is exchanged."""
    import hashlib
    if exchange is None or exchange.world == 1 or not hasattr(exchange.ctx, "comm_allreduce_max"):
        return
    a = np.ascontiguousarray(pts)
    h = hashlib.sha256(repr(a.shape).encode() + a.tobytes() + repr(params).encode()).digest()
    v = float(int.from_bytes(h[:8], "little") >> 12)            # exact in a double
    hi = exchange.ctx.comm_allreduce_max(v)
    lo = -exchange.ctx.comm_allreduce_max(-v)
    if hi != lo:
        raise RuntimeError("pyprogressivex: the ranks of this sharded call hold different point sets (or shapes) or passed different parameters; sharding needs every "
                           "rank to make the same call on the same data - unset PGX_MULTI_GPU / distributed= for independent calls")


def shared_seed():
    """A seed every rank of one launch agrees on without communicating (the API's seed=None in a multi-rank launch)."""
    import hashlib
    return int.from_bytes(hashlib.sha256(_launcher_key().encode()).digest()[:8], "little")


def shard_samples(samples, world, rank):
    """Contiguous shard of the sample list for this rank, padded to the common length by repeating the first sample (the
    padding's hypotheses are cut away when the gathered tables are merged)."""
    S = samples.shape[0]
    per, bounds = shard_bounds(S, world)
    lo, hi = bounds[rank]
    shard = np.repeat(samples[:1], per, axis=0)
    shard[: hi - lo] = samples[lo:hi]
    return np.ascontiguousarray(shard), per, bounds


def merge_sample_tables(gathered, S, world, slots):
    """rank-major tables of per * slots hypotheses per rank -> the S * slots hypotheses in global sample order."""
    per, bounds = shard_bounds(S, world)
    out = {}
    for key in ("counts", "values", "shared", "scores"):
        arr = np.asarray(gathered[key]).reshape(world, per * slots)
        out[key] = np.concatenate([arr[r, : (hi - lo) * slots] for r, (lo, hi) in enumerate(bounds)])
    return out


def score_sharded(exchange, models, T2, has_compound=False, exponent=2):
    """Scores `models` (the same full batch on every rank) cooperatively; every rank returns the full table."""
    models = np.ascontiguousarray(models, dtype=np.float64)
    M = models.shape[0]
    shard, lo, hi = shard_hypotheses(models, exchange.world, exchange.rank)
    # two pieces in flight once a rank's shard is large enough for the second piece's scoring to hide the first one's exchange
    comm_on = exchange.world > 1 or getattr(getattr(exchange, "ctx", None), "force_comm", False)
    pieces = 2 if comm_on and shard.shape[0] >= 256 else 1
    gathered = score_shard_pipelined(exchange, shard, T2, has_compound, exponent, pieces)
    keys = {k: gathered[k] for k in ("counts", "values", "shared", "scores")}
    return merge_gathered(keys, M, exchange.world)


# ---------------------------------------------------------------------------------------------------------------------
# point-sharded scoring
# ---------------------------------------------------------------------------------------------------------------------
def point_slice(n, world, rank):
    """Contiguous slice [lo, hi) of the job's n points for this rank (the last ranks may hold one point less)."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def fixed_point_scale(n_total):
    """The 2^q the score path accumulates its sums in for a job of n_total points (score.hip: q = min(62 - ceil(log2(n + 1)), 50))."""
    lg = 0
    while (1 << lg) < int(n_total) + 1:
        lg += 1
    return float(2 ** min(62 - lg, 50))


def table_from_accumulators(counts, values_q, shared_q, n_total, has_compound, exponent):
    """counts / values / shared / scores from summed integer accumulators - what score_finish_kernel and pgx_score_fetch do."""
    q = fixed_point_scale(n_total)
    values = np.asarray(values_q).astype(np.int64).astype(np.float64) / q
    shared = np.asarray(shared_q).astype(np.int64).astype(np.float64) / q
    scores = values - np.power(shared, float(exponent)) if has_compound else values.copy()
    return dict(counts=np.asarray(counts).astype(np.int64), values=values, shared=shared, scores=scores)


def score_point_sharded(exchange, models, T2, has_compound=False, exponent=2, pieces=1):
    """Scores `models` (the same full batch on every rank) against the points of ALL ranks: exchange.ctx holds this rank's
    slice (`ctx.set_points(slice)`, `ctx.score_set_global_n(n_total)`, compound slice likewise).  Every rank returns the job's
    table, bitwise the one a single GPU holding all the points returns.  pieces > 1: the batch in consecutive parts, the
    reduction of one overlapping the scoring of the next (begin_reduce / end_reduce)."""
    models = np.ascontiguousarray(models, dtype=np.float64)
    M = models.shape[0]
    pieces = max(1, min(int(pieces), M))
    if pieces == 1 or not hasattr(exchange, "begin_reduce"):
        return exchange.reduce_scores(models, T2, has_compound, exponent)
    cuts = [(M * k) // pieces for k in range(pieces + 1)]
    tables, open_slots = [], []
    try:
        for k in range(pieces):
            exchange.begin_reduce(k & 1, np.ascontiguousarray(models[cuts[k]:cuts[k + 1]]), T2, has_compound)
            open_slots.append(k & 1)
            if k >= 1:
                open_slots.remove((k - 1) & 1)
                tables.append(exchange.end_reduce((k - 1) & 1, exponent))
        open_slots.remove((pieces - 1) & 1)
        tables.append(exchange.end_reduce((pieces - 1) & 1, exponent))
    finally:
        for slot in open_slots:
            try:
                exchange.end_reduce(slot, exponent)
            except Exception:
                pass
    return {key: np.concatenate([t[key] for t in tables]) for key in ("counts", "values", "shared", "scores")}
