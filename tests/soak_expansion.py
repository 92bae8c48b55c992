"""Soak of the alpha-expansion against the CPU oracle's Dinic solver on deliberately odd problems (tests/test_gpu_fuzz.py runs a
bounded slice): graph shapes that stress the push-relabel machinery (long paths, stars, cliques, grids, disconnected pieces,
isolated sites, multiplicities), unary tables full of ties, zeros, identical rows and costs up to 2^40, lambda from 0.001 to 1,
label costs from 0 to far beyond the data term, up to 12 labels, odd starting labellings.  Labels, energy and cycle count must be
identical.  usage: python tests/soak_expansion.py <seed> <trials>"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "progressive-x_amd"), os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]
import numpy as np
from helpers import csr_from_pairs, random_sym_graph, realistic_labeling_problem
from pyprogressivex import _lib
import pgx_oracle as O


def odd_graph(rng, n):
    kind = int(rng.integers(0, 8))
    if n < 3:
        kind = 0
    if kind == 0:      # sparse random, multiplicities 1..2
        return random_sym_graph(rng, n, min(1.0, 4.0 / max(n, 2)))
    if kind == 1:      # one long path (deep searches), randomly permuted site numbers
        p = rng.permutation(n)
        return csr_from_pairs(n, p[:-1], p[1:], rng.integers(1, 3, n - 1))
    if kind == 2:      # stars: a few hubs of high degree
        hubs = rng.choice(n, size=min(n, int(rng.integers(1, 4))), replace=False)
        a = np.setdiff1d(np.arange(n), hubs)
        b = hubs[rng.integers(0, len(hubs), len(a))]
        return csr_from_pairs(n, a, b, rng.integers(1, 3, len(a)))
    if kind == 3:      # grid
        w = max(2, int(np.sqrt(n)))
        idx = np.arange(n)
        right = idx[(idx % w != w - 1) & (idx + 1 < n)]
        down = idx[idx + w < n]
        a, b = np.concatenate([right, down]), np.concatenate([right + 1, down + w])
        return csr_from_pairs(n, a, b, np.full(len(a), 2))
    if kind == 4:      # disjoint small cliques + isolated sites
        a, b = [], []
        i = 0
        while i + 1 < n:
            c = int(rng.integers(1, 7))
            m = np.arange(i, min(n, i + c))
            if rng.random() < 0.7:
                iu, ju = np.triu_indices(len(m), 1)
                a.append(m[iu]); b.append(m[ju])
            i += c
        a = np.concatenate(a) if a else np.zeros(0, int)
        b = np.concatenate(b) if b else np.zeros(0, int)
        return csr_from_pairs(n, a, b, np.full(len(a), 2))
    if kind == 5:      # no arcs at all
        return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    if kind == 6:      # dense random
        return random_sym_graph(rng, n, min(1.0, 40.0 / max(n, 2)))
    return realistic_labeling_problem(n, 3, 0.1, seed=int(rng.integers(1 << 30)))[1]


def odd_unary(rng, n, L, lam):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        D = rng.integers(0, 1 << 33, (n, L))
    elif kind == 1:    # ties everywhere: small integers times a large unit
        D = rng.integers(0, 4, (n, L)) << int(rng.integers(0, 34))
    elif kind == 2:    # all zero
        D = np.zeros((n, L), np.int64)
    elif kind == 3:    # huge
        D = rng.integers(0, 1 << 40, (n, L))
    elif kind == 4:    # identical rows
        D = np.tile(rng.integers(0, 1 << 33, (1, L)), (n, 1))
    elif kind == 5:    # one label free for everyone
        D = rng.integers(0, 1 << 33, (n, L))
        D[:, int(rng.integers(0, L))] = 0
    else:              # PEARL-like: clusters + constant outlier column
        D = np.rint(rng.random((n, L)) * 2 * (1 - min(lam, 0.999)) * 2.0 ** 32).astype(np.int64)
        D[np.arange(n), rng.integers(0, L, n)] >>= 5
        D[:, L - 1] = int((1 - min(lam, 0.999)) * 2.0 ** 32)
    return np.ascontiguousarray(D.astype(np.int64))


def soak(seed, trials, verbose=True, max_n=9000):
    rng = np.random.default_rng(seed)
    ctx = _lib.Context(0)
    bad = 0
    t0 = time.time()
    for trial in range(trials):
        n = min(max_n, int(rng.choice([1, 2, 3, 17, 64, 65, 300, 1000, 2500, 9000])))
        L = int(rng.integers(2, 13))
        lam = float(rng.choice([0.001, 0.02, 0.1, 0.3, 0.6, 0.9, 0.99, 1.0]))
        h = float(rng.choice([0.0, 0.0, 1e-6, 0.5, 3.0, 20.0, 200.0, 1e5]))
        graph = odd_graph(rng, n)
        Dq = odd_unary(rng, n, L, lam)
        sk = int(rng.integers(0, 4))
        start = (rng.integers(0, L, n) if sk == 0 else np.zeros(n) if sk == 1 else np.full(n, L - 1) if sk == 2
                 else np.argmin(Dq, axis=1)).astype(np.int32)
        lq, hq = O.quantize_lambda(lam), O.quantize(h)
        ref, re, rc = O.expansion(Dq, graph, lq, hq, start.copy())
        ctx.set_unary_q(Dq)
        ctx.set_graph(*graph)
        ctx.set_labels(start.copy())
        eq, e, cyc = ctx.expansion(lam, h)
        got = ctx.get_labels()
        if not (np.array_equal(got, ref) and eq == re and cyc == rc):
            bad += 1
            print("MISMATCH", dict(seed=seed, trial=trial, n=n, L=L, lam=lam, h=h, arcs=len(graph[1])), "labels differing", int((got != ref).sum()),
                  "energy", eq, re, "cycles", cyc, rc, flush=True)
    ctx.close()
    if verbose:
        print(f"expansion soak done: seed {seed}, {trials} problems, {bad} mismatches, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    soak(int(sys.argv[1]), int(sys.argv[2]))
