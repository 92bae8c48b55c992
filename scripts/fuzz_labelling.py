"""One-off differential fuzz: full alpha-expansion on the GPU vs the CPU oracle on random realistic problems."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), ROOT]
import numpy as np
from pyprogressivex import _lib
from helpers import realistic_labeling_problem, random_sym_graph
import pgx_oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ctx = _lib.Context(0)
bad = 0
t0 = time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n = int(rng.choice([2, 7, 50, 300, 1500, 4000, 9000, 20000]))
    L = int(rng.integers(2, 9))
    lam = float(rng.choice([0.02, 0.1, 0.3, 0.6, 0.9]))
    h = float(rng.choice([0.0, 0.5, 3.0, 20.0, 200.0]))
    Dq, graph = realistic_labeling_problem(n, L=L, lam=lam, seed=int(rng.integers(1 << 30)))
    lq, hq = O.quantize_lambda(lam), O.quantize(h)
    start = rng.integers(0, L, n).astype(np.int32) if trial % 2 else np.zeros(n, np.int32)
    ref, re, rc = O.expansion(Dq, graph, lq, hq, start.copy())
    ctx.set_unary_q(Dq)
    ctx.set_graph(*graph)
    ctx.set_labels(start.copy())
    eq, e, cyc = ctx.expansion(lam, h)
    got = ctx.get_labels()
    ok = np.array_equal(got, ref) and eq == re and cyc == rc
    if not ok:
        bad += 1
        print("MISMATCH", trial, n, L, lam, h, int((got != ref).sum()), eq, re, cyc, rc, flush=True)
print(f"fuzz done: {trial + 1} problems, {bad} mismatches, {time.time() - t0:.0f} s")
