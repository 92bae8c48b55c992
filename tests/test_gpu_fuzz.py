"""Differential fuzz of the alpha-expansion: random realistic problems, GPU against the CPU oracle's Dinic solver - labels,
energy and cycle count identical.  (A longer one-off run of the same loop - 530 problems, 2 ... 20 000 sites - is on record in
DESIGN.md 5.4; this is the committed slice.)"""
import numpy as np
import pytest

from helpers import realistic_labeling_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_expansion_fuzz_against_the_oracle(gpu_ctx, oracle, seed):
    rng = np.random.default_rng(seed)
    for trial in range(30):
        n = int(rng.choice([2, 7, 50, 300, 1500, 4000, 9000]))
        L = int(rng.integers(2, 9))
        lam = float(rng.choice([0.02, 0.1, 0.3, 0.6, 0.9]))
        h = float(rng.choice([0.0, 0.5, 3.0, 20.0, 200.0]))
        Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=int(rng.integers(1 << 30)))
        lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
        start = rng.integers(0, L, n).astype(np.int32) if trial % 2 else np.zeros(n, np.int32)
        ref, re, rc = oracle.expansion(Dq, graph, lq, hq, start.copy())
        gpu_ctx.set_unary_q(Dq)
        gpu_ctx.set_graph(*graph)
        gpu_ctx.set_labels(start.copy())
        eq, e, cyc = gpu_ctx.expansion(lam, h)
        got = gpu_ctx.get_labels()
        assert np.array_equal(got, ref) and eq == re and cyc == rc, (seed, trial, n, L, lam, h, int((got != ref).sum()))


def test_expansion_fuzz_on_region_moves(oracle, monkeypatch):
    """The same loop with the one-workgroup whole-graph path switched off (PGX_MF_TILE=0): every move goes through the region
    path - compaction of the open sites, weak-sink promotion, one-workgroup solve, a cycle's moves enqueued back to back - and
    the ones it declines through maxflow.hip's level-synchronous launches."""
    from pyprogressivex import _lib
    monkeypatch.setenv("PGX_MF_TILE", "0")
    ctx = _lib.Context(0)
    try:
        rng = np.random.default_rng(77)
        for trial in range(24):
            n = int(rng.choice([50, 1500, 5000, 9000, 20000]))
            L = int(rng.integers(2, 9))
            lam = float(rng.choice([0.02, 0.1, 0.3, 0.6]))
            h = float(rng.choice([0.0, 3.0, 20.0]))
            Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=int(rng.integers(1 << 30)))
            lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
            start = rng.integers(0, L, n).astype(np.int32) if trial % 2 else np.zeros(n, np.int32)
            ref, re, rc = oracle.expansion(Dq, graph, lq, hq, start.copy())
            ctx.set_unary_q(Dq)
            ctx.set_graph(*graph)
            ctx.set_labels(start.copy())
            eq, e, cyc = ctx.expansion(lam, h)
            got = ctx.get_labels()
            assert np.array_equal(got, ref) and eq == re and cyc == rc, (trial, n, L, lam, h, int((got != ref).sum()))
        paths = ctx.expansion_paths()
        assert paths["region"] > 0 and paths["one_workgroup"] == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_scoring_soak_slice(oracle, seed):
    """A bounded slice of tests/soak_scoring.py: random sizes and thresholds (some exactly on a residual), hypotheses a hair from
    the truth, garbage, uniformly rescaled by up to 10^+-160 and with entries of wildly different magnitude, all six residuals -
    counts, masks and values against the oracle.  (This loop is what found the two scale holes of DESIGN.md 5.2h.)"""
    from soak_scoring import soak
    assert soak(seed, 60, verbose=False) == 0


@pytest.mark.parametrize("seed", [201, 202])
def test_pointwise_soak_slice(oracle, seed):
    """A bounded slice of tests/soak_pointwise.py: preference vectors, compound maximum, unary table, residual sums, minimal
    solvers, neighbourhood graph, the inlier/outlier cut and the greedy labelling on the same wild hypotheses and point sets
    (absurd scales, NaN / Inf / 1e200 coordinates, duplicates) - bit-exact against the oracle."""
    from soak_pointwise import soak
    assert soak(seed, 150, verbose=False) == 0


@pytest.mark.parametrize("seed", [301, 302])
def test_api_soak_slice(oracle, seed):
    """A bounded slice of tests/soak_api.py: the five drop-in calls with random arguments on random small problems, on the GPU
    and through the same host code over the CPU oracle.  A differing call is replayed with every context call recorded and
    classified: only a call that returns different integers (or floats beyond 1e-9) for bitwise identical inputs is a failure;
    a Gauss-Newton refit that amplified the rounding of a Gram sum is reported and tolerated (2 of 3 000 calls)."""
    from soak_api import soak
    assert soak(seed, 30, verbose=False) == 0


@pytest.mark.parametrize("seed", [401, 402])
def test_expansion_soak_slice(oracle, seed):
    """A bounded slice of tests/soak_expansion.py: long paths, stars, cliques, grids, disconnected and empty graphs, unary tables
    of ties / zeros / identical rows / costs up to 2^40, lambda 0.001 ... 1, label costs 0 ... 1e5, 2 ... 12 labels - labels,
    energy and cycle count identical to the oracle's Dinic solver (4 200 such problems on record: no mismatch)."""
    from soak_expansion import soak
    assert soak(seed, 120, verbose=False, max_n=2500) == 0   # (the oracle's Dinic solver needs a minute for some 9000-site cases)


@pytest.mark.parametrize("seed", [32, 403])
def test_expansion_soak_slice_through_region_moves(oracle, seed, monkeypatch):
    """The same soak with the one-workgroup whole-graph solver switched off, so that every move of these small odd problems goes
    through the region path (open sites compacted, weak sinks promoted, hubs checked) or is declined by it.  Seed 32 holds the
    problem - a long path, ten labels, a unary table full of ties, lambda = 1 - on which the non-strict forms of the region's
    validity conditions (need <= rt, pool - needsum >= h) returned a different minimum cut than the minimal sink side."""
    from soak_expansion import soak
    monkeypatch.setenv("PGX_MF_TILE", "0")
    assert soak(seed, 120, verbose=False, max_n=2500) == 0


def test_replay_soak_slice():
    """60 random drop-in calls through libpgx, every decision against the independent replay of progressive_x.h / PEARL.h
    (tests/soak_replay.py; the long campaigns are in profiles/)"""
    import soak_replay
    assert soak_replay.soak(2024, 60, verbose=False) == 0
