"""oracle/bk_maxflow.c (Boykov-Kolmogorov, the solver family behind GCoptimization / PEARL.h:550) against the oracle's Dinic,
scipy and brute force - it backs bench.py's CPU labelling baseline, so it has to be the same function."""
import itertools

import numpy as np

from helpers import random_sym_graph, realistic_labeling_problem
from soak_expansion import odd_graph, odd_unary


def test_bk_maxflow_against_scipy_and_dinic(oracle):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    rng = np.random.default_rng(10)
    for _ in range(400):
        nn = int(rng.integers(3, 16))
        na = int(rng.integers(3, 70))
        fr, to = rng.integers(0, nn, na), rng.integers(0, nn, na)
        keep = (fr != to) & (to != 0) & (fr != nn - 1)
        fr, to = fr[keep], to[keep]
        if len(fr) == 0:
            continue
        cap = rng.integers(0, 30, len(fr))
        f, side = oracle.maxflow_bk(nn, fr, to, cap, 0, nn - 1)
        fd, side_d = oracle.maxflow(nn, fr, to, cap, 0, nn - 1)
        A = np.zeros((nn, nn), dtype=np.int32)
        np.add.at(A, (fr, to), cap)
        assert f == fd == maximum_flow(csr_matrix(A), 0, nn - 1).flow_value
        assert np.array_equal(side, side_d)               # the minimal sink side is unique


def test_bk_move_is_optimal_and_maximal(oracle):
    """as tests/test_oracle.py::test_expansion_move_is_optimal_and_maximal, for the BK-backed move"""
    from test_oracle import _energy_py
    rng = np.random.default_rng(11)
    for trial in range(200):
        n, L = int(rng.integers(2, 10)), int(rng.integers(2, 5))
        Dq = rng.integers(0, 12, (n, L)).astype(np.int64)
        graph = random_sym_graph(rng, n, 0.4)
        lq, hq = int(rng.integers(0, 4)) * 2, int(rng.integers(0, 8))
        lab = rng.integers(0, L, n).astype(np.int32)
        alpha = int(rng.integers(0, L))
        new, changed, flow = oracle.expand_alpha_bk(Dq, graph, lq, hq, alpha, lab)
        ref, rchanged, rflow = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        assert np.array_equal(new, ref) and changed == rchanged and flow == rflow
        act = [i for i in range(n) if lab[i] != alpha]
        best = min(_energy_py(Dq, graph, lq, hq, np.where(np.isin(np.arange(n), [i for b, i in zip(bits, act) if b == 0]), alpha, lab))
                   for bits in itertools.product([0, 1], repeat=len(act)))
        assert _energy_py(Dq, graph, lq, hq, new) == best


def test_bk_expansion_equals_dinic_on_the_soak_problems(oracle):
    """the odd problems of tests/soak_expansion.py (paths, stars, grids, cliques, ties, 2^40 costs, label costs far beyond the data
    term, up to 12 labels, odd starts): labels, energy, cycle count identical to the Dinic-backed expansion"""
    rng = np.random.default_rng(12)
    for trial in range(120):
        n = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 1000]))
        L = int(rng.integers(2, 13))
        lam = float(rng.choice([0.001, 0.02, 0.1, 0.3, 0.6, 0.9, 0.99, 1.0]))
        h = float(rng.choice([0.0, 0.0, 1e-6, 0.5, 3.0, 20.0, 200.0, 1e5]))
        graph = odd_graph(rng, n)
        Dq = odd_unary(rng, n, L, lam)
        sk = int(rng.integers(0, 4))
        start = (rng.integers(0, L, n) if sk == 0 else np.zeros(n) if sk == 1 else np.full(n, L - 1) if sk == 2
                 else np.argmin(Dq, axis=1)).astype(np.int32)
        lq, hq = oracle.quantize_lambda(lam), oracle.quantize(h)
        ref, re, rc = oracle.expansion(Dq, graph, lq, hq, start.copy())
        got, ge, gc, cuts = oracle.expansion_bk(Dq, graph, lq, hq, start.copy())
        assert np.array_equal(got, ref) and ge == re and gc == rc and cuts == gc * L, (trial, n, L, lam, h)


def test_bk_expansion_on_a_realistic_problem(oracle):
    prob = realistic_labeling_problem(6000, 5, 0.1, seed=3)
    Dq, graph = prob[0], prob[1]
    lq, hq = oracle.quantize_lambda(0.1), oracle.quantize(20.0)
    z = np.zeros(Dq.shape[0], np.int32)
    ref, re, rc = oracle.expansion(Dq, graph, lq, hq, z)
    got, ge, gc, cuts = oracle.expansion_bk(Dq, graph, lq, hq, z)
    assert np.array_equal(got, ref) and ge == re and gc == rc
