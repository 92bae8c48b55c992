"""bench.py --gpus N end to end on CPU: N = 2 and 4 gloo ranks run bench.py's own main() - sharding of the
points / hypotheses, the two-in-flight exchange pipeline, barriers, max-over-ranks timing, rank 0's JSON line - with the CPU
oracle behind the context surface (tests/stub_bench_ctx.py).  Checks the line's contract fields and arithmetic for N > 1 and that
every split names the same winner as one rank.  The real RCCL path needs GPUs: tests/test_gpu_multi.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_bench_ctx.py")
ARGS = ["--stub", "--steps", "3", "--warmup", "1", "--points", "4000", "--hyps", "64", "--no-legs", "--no-cpu-baseline"]


def _run(world, scaling):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PGX_BENCH_STUB=STUB)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--scaling", scaling] + ARGS,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert not any(ln.startswith("{") for o, _ in outs[1:] for ln in o.splitlines())     # only rank 0 prints the line (gloo's own banner aside)
    lines = [ln for ln in outs[0][0].strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                           # ONE JSON line
    last = outs[0][0].strip().splitlines()[-1]
    assert last == lines[0] and len(last) < 8000                     # ... the LAST one, and short enough for the driver's 8 KB tail
    assert any(ln.startswith("BENCH_DETAIL {") for ln in outs[0][0].splitlines())      # the full record travels on an earlier line
    return json.loads(last)


@pytest.fixture(scope="module")
def single():
    return _run(1, "strong-points")


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("scaling", ["strong-points", "strong", "weak"])
def test_bench_line_for_n_ranks(single, world, scaling):
    d = _run(world, scaling)
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["data"] == "stub"
    assert d["scaling"] == ("weak" if scaling == "weak" else "strong") and d["scaling_mode"] == scaling
    cfg = d["config"]
    n, total = cfg["points"], cfg["hypotheses_total"]
    assert n == 4000 and total == (64 * world if scaling == "weak" else 64)
    assert cfg["parallelism"] == (f"point-sharded x{world}" if scaling == "strong-points" else f"hypothesis-sharded x{world}")
    assert cfg["points_per_gpu"] == (n // world if scaling == "strong-points" else n)
    assert cfg["hypotheses_per_gpu"] == (64 // world if scaling == "strong" else 64)
    assert "all-reduce" in cfg["exchange"] if scaling == "strong-points" else "all-gather" in cfg["exchange"]
    # value = WHOLE-JOB pairs / max-over-ranks time per step; models/s likewise
    assert abs(d["value"] - n * total / (d["ms_per_step"] * 1e-3)) <= 1e-9 * d["value"]
    assert abs(d["models_per_sec"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-9 * d["models_per_sec"]
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64"
    if scaling != "weak":     # the same 64 hypotheses against the same points, however they are split: the same winner
        assert d["winner"] == single["winner"]


def test_single_rank_line(single):
    assert single["n_gpus"] == 1 and single["config"]["parallelism"] == "1 GPU" and single["config"]["exchange"] == "none"
