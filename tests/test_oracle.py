"""CPU tests of the oracle itself: hand-checkable known answers, exact-rational cross-checks of every residual formula,
max-flow against scipy, expansion moves against brute force, and the committed golden vectors.

The reference ships no tests or golden vectors for this path => parity is UNPINNED against upstream;
these tests pin the oracle against the mathematics it restates and against regressions.
"""
import itertools
import os
from fractions import Fraction as Fr

import numpy as np
import pytest

from helpers import MODEL_CASES, make_case, random_sym_graph

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_v1.npz")


# ---- hand-checkable residuals ----------------------------------------------------------------------------------------
def test_residual_known_answers(oracle):
    O = oracle
    # axis-aligned line x = 3:  (1, 0, -3)
    assert O.squared_residuals(O.LINE2D, [[5.0, 9.0], [3.0, -1.0]], [1.0, 0.0, -3.0]).tolist() == [4.0, 0.0]
    # identity homography: residual = squared displacement
    assert O.squared_residuals(O.HOMOGRAPHY, [[1, 2, 4, 6]], np.eye(3).reshape(-1)).tolist() == [25.0]
    # pure translation (2, 0): maps (1,2) -> (3,2)
    H = np.array([1, 0, 2, 0, 1, 0, 0, 0, 1.0])
    assert O.squared_residuals(O.HOMOGRAPHY, [[1, 2, 3, 2]], H).tolist() == [0.0]
    # symmetric transfer error under identity = 2 x one-way
    Hs = np.concatenate([np.eye(3).reshape(-1), np.eye(3).reshape(-1)])
    assert O.squared_residuals(O.HOMOGRAPHY_SYM, [[1, 2, 4, 6]], Hs).tolist() == [50.0]
    # pure x-translation F = [t]_x with t = (1,0,0): epipolar lines are horizontal, Sampson = dy^2 / 2
    F = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0.0])
    assert O.squared_residuals(O.FUNDAMENTAL, [[0, 0, 7, 0], [0, 0, 7, 2]], F).tolist() == [0.0, 2.0]
    # PnP, P = [I | 0]: projection of (2, 4, 2) is (1, 2)
    P = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(-1)
    assert O.squared_residuals(O.PNP, [[1, 2, 2, 4, 2], [0, 0, 2, 4, 2]], P).tolist() == [0.0, 5.0]
    # vanishing point at infinity in direction x (v2 = 0): horizontal segments have zero residual
    vp = np.array([1.0, 0.0, 0.0])
    assert O.squared_residuals(O.VANISHING_POINT, [[0, 5, 10, 5]], vp).tolist() == [0.0]
    # vertical segment of half-length 3: its start point is 3 px away from the horizontal line through its midpoint
    assert O.squared_residuals(O.VANISHING_POINT, [[4, 2, 4, 8]], vp).tolist() == [9.0]
    # unsquared residual used by PEARL's refit test
    assert O.residual(O.LINE2D, [5.0, 9.0], [1.0, 0.0, -3.0]) == 2.0
    assert O.residual(O.HOMOGRAPHY, [1, 2, 4, 6], np.eye(3).reshape(-1)) == 5.0


def _exact_sq(name, p, m):
    p = [Fr(float(x)) for x in p]
    m = [Fr(float(x)) for x in m]
    if name == "line":
        return (m[0] * p[0] + m[1] * p[1] + m[2]) ** 2
    if name in ("homography", "homography_sym"):
        t1 = m[0] * p[0] + m[1] * p[1] + m[2]
        t2 = m[3] * p[0] + m[4] * p[1] + m[5]
        t3 = m[6] * p[0] + m[7] * p[1] + m[8]
        r = (p[2] - t1 / t3) ** 2 + (p[3] - t2 / t3) ** 2
        if name == "homography_sym":
            g = m[9:]
            s1 = g[0] * p[2] + g[1] * p[3] + g[2]
            s2 = g[3] * p[2] + g[4] * p[3] + g[5]
            s3 = g[6] * p[2] + g[7] * p[3] + g[8]
            r += (p[0] - s1 / s3) ** 2 + (p[1] - s2 / s3) ** 2
        return r
    if name == "fundamental":
        x1, y1, x2, y2 = p
        Fx = [m[0] * x1 + m[1] * y1 + m[2], m[3] * x1 + m[4] * y1 + m[5], m[6] * x1 + m[7] * y1 + m[8]]
        Ftx = [m[0] * x2 + m[3] * y2 + m[6], m[1] * x2 + m[4] * y2 + m[7], m[2] * x2 + m[5] * y2 + m[8]]
        num = (x2 * Fx[0] + y2 * Fx[1] + Fx[2]) ** 2
        return num / (Fx[0] ** 2 + Fx[1] ** 2 + Ftx[0] ** 2 + Ftx[1] ** 2)
    if name == "pnp":
        u, v, X, Y, Z = p
        px = m[0] * X + m[1] * Y + m[2] * Z + m[3]
        py = m[4] * X + m[5] * Y + m[6] * Z + m[7]
        pz = m[8] * X + m[9] * Y + m[10] * Z + m[11]
        return (u - px / pz) ** 2 + (v - py / pz) ** 2
    if name == "vanishing_point":
        xs, ys, xe, ye = p
        mx, my = (xs + xe) / 2, (ys + ye) / 2
        lx, ly, lz = my * m[2] - m[1], -(mx * m[2] - m[0]), mx * m[1] - my * m[0]
        return (lx * xs + ly * ys + lz) ** 2 / (lx * lx + ly * ly)
    raise KeyError(name)


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_residuals_against_exact_rational_arithmetic(oracle, name):
    """Independent check of every formula: evaluate the published definition in exact rational arithmetic."""
    mt, pts, models, thr = make_case(name, 40, 4, seed=1)
    for model in models:
        got = oracle.squared_residuals(mt, pts, model)
        for i in range(pts.shape[0]):
            ex = float(_exact_sq(name, pts[i], model))
            assert abs(got[i] - ex) <= 1e-9 * max(abs(ex), 1e-30) + 1e-18 * max(1.0, abs(ex)), (name, i)


# ---- a1 semantics -----------------------------------------------------------------------------------------------------
def test_score_semantics(oracle):
    O = oracle
    pts = np.array([[0.0, 0], [1.0, 0], [2.0, 0], [3.0, 0]])
    model = np.array([[1.0, 0, 0]])  # r^2 = x^2 : 0, 1, 4, 9
    s = O.score(O.LINE2D, pts, model, 4.0, want_masks=True)
    assert s["counts"][0] == 2 and s["masks"][0, 0] == 0b0011          # strict <
    assert s["values"][0] == (1 - 0 / 4) + (1 - 1 / 4) and s["shared"][0] == 0.0
    assert s["scores"][0] == s["values"][0]                            # empty compound: no subtraction (:110)
    comp = np.array([0.5, 1.0, 1.0, 1.0])
    s2 = O.score(O.LINE2D, pts, model, 4.0, compound=comp, exponent=2)
    assert s2["shared"][0] == min(0.5, 1.0) + min(1.0, 0.75)           # :115-117
    assert s2["scores"][0] == s2["values"][0] - 1.25 ** 2              # :120
    s3 = O.score(O.LINE2D, pts, model, 4.0, compound=comp, exponent=0)
    assert s3["scores"][0] == s3["values"][0] - 1.0                    # pow(x, 0) == 1 is still subtracted
    # early exit (:105-106): Score() iff count + 1 < best
    assert O.score(O.LINE2D, pts, model, 4.0, best_inlier_number=[3])["counts"][0] == 2
    assert O.score(O.LINE2D, pts, model, 4.0, best_inlier_number=[4])["counts"][0] == 0


def test_preference_tanimoto_compound_unseen(oracle):
    O = oracle
    pts = np.array([[0.0, 0], [1.0, 0], [3.0, 0]])
    pref = O.preference(O.LINE2D, pts, [1.0, 0, 0], 4.0)
    assert pref.tolist() == [1.0, 0.75, 0.0]                           # max(0, 1 - r^2/T^2), progx_model.h:85
    d, a, b = O.tanimoto_terms(pref, np.array([1.0, 0.0, 1.0]))
    assert (d, a, b) == (1.0, 1.0 + 0.5625, 2.0)
    ok, t = O.is_valid_tanimoto(d, a, b, 0.5)
    assert t == 1.0 / (1.5625 + 2.0 - 1.0) and ok                      # 0.39 < 0.5 -> valid
    assert not O.is_valid_tanimoto(d, a, b, 0.3)[0]
    ok, t = O.is_valid_tanimoto(0.0, 0.0, 0.0, 0.4)                    # 0/0 = NaN -> valid (progressive_x.h:587)
    assert ok and np.isnan(t)
    assert O.compound_max(np.array([[0.1, 0.9, 0.0], [0.5, 0.2, 0.0]])).tolist() == [0.5, 0.9, 0.0]
    # progressive_x.h:495-513
    for args in [(0.5, 4, 1000, 1, 5000), (0.1, 3, 400, 600, 1886), (0.05, 2, 10, 0, 100)]:
        omc, m, it, cov, n = args
        want = int(round((n - cov) * (1 - omc ** (1.0 / it)) ** (1.0 / m)))
        assert O.predicted_unseen_inliers(*args) == want


def test_unary_semantics(oracle):
    O = oracle
    pts = np.array([[0.0, 0], [1.0, 0], [2.0, 0], [2.5, 0]])
    thr, lam = 4.0 / 3.0, 0.25                                         # T2 = 9/4 thr^2 = 4
    D = O.unary(O.LINE2D, pts, [[1.0, 0, 0]], thr, lam)
    T2 = 9.0 / 4.0 * thr * thr
    assert D[:, 1].tolist() == [0.75] * 4                              # outlier label: 1 - lambda (PEARL.h:100-101)
    assert D[0, 0] == 0.0 and D[1, 0] == 0.75 * 1.0 / T2
    assert D[2, 0] == 0.75 * 4.0 / T2                                  # r^2 == T2 is NOT beyond the threshold (:123)
    assert D[3, 0] == 1.5                                              # beyond: 2 (1 - lambda)
    Dq = O.unary_q(O.LINE2D, pts, [[1.0, 0, 0]], thr, lam)
    assert Dq[3, 0] == 3 << 31 and Dq[0, 1] == 3 << 30
    assert O.quantize_lambda(0.25) == 1 << 30 and O.quantize_lambda(0.1) % 2 == 0


# ---- max-flow / expansion ----------------------------------------------------------------------------------------------
def test_maxflow_against_scipy(oracle):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    rng = np.random.default_rng(0)
    for _ in range(150):
        nn = int(rng.integers(4, 14))
        na = int(rng.integers(5, 50))
        fr, to = rng.integers(0, nn, na), rng.integers(0, nn, na)
        keep = fr != to
        fr, to = fr[keep], to[keep]
        cap = rng.integers(0, 30, len(fr))
        f, side = oracle.maxflow(nn, fr, to, cap, 0, nn - 1)
        A = np.zeros((nn, nn), dtype=np.int32)
        np.add.at(A, (fr, to), cap)
        assert f == maximum_flow(csr_matrix(A), 0, nn - 1).flow_value
        assert side[nn - 1] == 1 and (f == 0 or side[0] == 0)


def _energy_py(Dq, graph, lq, hq, lab):
    off, idx, mult = graph
    e = sum(int(Dq[i, lab[i]]) for i in range(len(lab)))
    for i in range(len(lab)):
        for a in range(off[i], off[i + 1]):
            if idx[a] < i and lab[idx[a]] != lab[i]:
                e += lq * int(mult[a])
    return e + hq * len(set(int(x) for x in lab))


def test_expansion_move_is_optimal_and_maximal(oracle):
    """Every move must (1) reach the minimum over all 2^k binary moves and (2) be the UNION of all optimal moves
    (ties -> alpha; BK's what_segment default SOURCE) — brute force on <= 9 sites."""
    rng = np.random.default_rng(1)
    for trial in range(250):
        n, L = int(rng.integers(2, 10)), int(rng.integers(2, 5))
        Dq = rng.integers(0, 12, (n, L)).astype(np.int64)
        graph = random_sym_graph(rng, n, 0.4)
        lq, hq = int(rng.integers(0, 4)) * 2, int(rng.integers(0, 8))
        lab = rng.integers(0, L, n).astype(np.int32)
        assert oracle.energy(Dq, graph, lq, hq, lab) == _energy_py(Dq, graph, lq, hq, lab)
        alpha = int(rng.integers(0, L))
        new, changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        act = [i for i in range(n) if lab[i] != alpha]
        best, union = None, lab.copy()
        for bits in itertools.product([0, 1], repeat=len(act)):
            l2 = lab.copy()
            for b, i in zip(bits, act):
                if b == 0:
                    l2[i] = alpha
            e = _energy_py(Dq, graph, lq, hq, l2)
            if best is None or e < best:
                best, union = e, l2.copy()
            elif e == best:
                union[l2 == alpha] = alpha
        assert _energy_py(Dq, graph, lq, hq, new) == best
        assert np.array_equal(new, union)
        assert changed == int(np.sum(new != lab))


def test_full_expansion_small_reaches_global_optimum_mostly(oracle):
    """alpha-expansion is a local search; on tiny instances it must at least never be worse than any single-label
    labelling and never increase the energy from its start."""
    rng = np.random.default_rng(4)
    for _ in range(40):
        n, L = int(rng.integers(3, 9)), int(rng.integers(2, 4))
        Dq = rng.integers(0, 20, (n, L)).astype(np.int64)
        graph = random_sym_graph(rng, n, 0.5)
        lq, hq = 2 * int(rng.integers(0, 4)), int(rng.integers(0, 10))
        start = np.zeros(n, np.int32)
        lab, e, cyc = oracle.expansion(Dq, graph, lq, hq, start)
        assert e == oracle.energy(Dq, graph, lq, hq, lab) <= oracle.energy(Dq, graph, lq, hq, start)
        for l in range(L):
            assert e <= oracle.energy(Dq, graph, lq, hq, np.full(n, l, np.int32))
        assert 1 <= cyc <= 1000


def test_bucket_and_residual_sum(oracle):
    labels = np.array([2, 0, 1, 0, 5, 2, 1], dtype=np.int32)
    counts, order = oracle.bucket(labels, 3)                           # L = 3: labels >= 2 are the outlier bucket
    assert counts.tolist() == [2, 2, 3] and order.tolist() == [1, 3, 2, 6, 0, 4, 5]
    pts = np.array([[1.0, 0], [2.0, 0], [3.0, 0]])
    assert oracle.residual_sum(0, pts, [1.0, 0, 0], np.array([1, 0, 1], np.int32), 1) == 4.0


# ---- committed golden vectors ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_golden_vectors(oracle, name):
    g = np.load(GOLDEN)
    mt = MODEL_CASES[name]
    pts, models, thr, comp = g[f"{name}_pts"], g[f"{name}_models"], float(g[f"{name}_thr"][0]), g[f"{name}_comp"]
    T2 = 2.25 * thr * thr
    assert np.array_equal(oracle.squared_residuals(mt, pts, models[0]), g[f"{name}_sq0"])
    sc = oracle.score(mt, pts, models, T2, compound=comp, has_compound=True, exponent=2, want_masks=True)
    for k in ("counts", "values", "shared", "scores", "masks"):
        assert np.array_equal(sc[k], g[f"{name}_{k}"]), k
    assert np.array_equal(oracle.preference(mt, pts, models[0], T2), g[f"{name}_pref0"])
    assert np.array_equal(oracle.unary_q(mt, pts, models[:3], thr, 0.1), g[f"{name}_unary_q"])


def test_golden_expansion(oracle):
    g = np.load(GOLDEN)
    graph = (g["exp_off"], g["exp_idx"], g["exp_mult"])
    lab, e, cyc = oracle.expansion(g["exp_Dq"], graph, int(g["exp_lq"][0]), int(g["exp_hq"][0]),
                                   np.zeros(400, np.int32))
    assert np.array_equal(lab, g["exp_labels"]) and e == int(g["exp_energy"][0]) and cyc == int(g["exp_cycles"][0])


def test_minimal_solvers_agree_with_the_host_formulas():
    """oracle C restatement of the 2-point line / 2-segment vanishing point solvers vs the numpy estimators"""
    import pgx_oracle as O
    from pyprogressivex import _estimators, datasets
    rng = np.random.default_rng(0)
    pts, _, _ = datasets.make_lines(seed=2)
    smp = rng.integers(0, len(pts), (500, 2)).astype(np.int32)
    smp[:10, 1] = smp[:10, 0]
    ref, src = _estimators.LineEstimator().minimal(pts, smp)
    got = O.solve_minimal(O.LINE2D, pts, smp)
    assert np.isnan(got[:10]).all() and np.array_equal(np.nonzero(~np.isnan(got[:, 0]))[0], src)
    assert np.abs(got[src] - ref).max() < 1e-12
    segs, _, _ = datasets.make_vanishing_points(n_inliers=600, n_vps=3, n_outliers=200, seed=1)
    smp = rng.integers(0, len(segs), (500, 2)).astype(np.int32)
    ref, src = _estimators.VanishingPointEstimator().minimal(segs, smp)
    got = O.solve_minimal(O.VANISHING_POINT, segs, smp)
    assert np.array_equal(np.nonzero(~np.isnan(got[:, 0]))[0], src)
    assert np.abs(got[src] - ref).max() < 1e-12


# ---- committed vectors + hand-checkable cases ------------------------------------------------------------
GOLDEN_NEXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_next_v1.npz")


def test_golden_next_rows(oracle):
    g = np.load(GOLDEN_NEXT)
    for d in (2, 4, 5):
        for tag, kind, radius, k in (("ball", 0, 6.0, 5), ("knn", 2, 0.0, 8)):
            got = oracle.graph_build(g[f"g{d}_pts"], kind, radius=radius, k=k)
            for name, a in zip(("off", "idx", "mult"), got):
                assert np.array_equal(a, g[f"g{d}_{tag}_{name}"]), (d, tag, name)
    prm = np.array([0.01, 300.0, 200.0, 0.012, 310.0, 190.0])
    for name, kinds in (("line", [(oracle.GRAM_AFFINE, None)]), ("vanishing_point", [(oracle.GRAM_VP, None)]),
                        ("homography", [(oracle.GRAM_AFFINE, None), (oracle.GRAM_DLT_H, prm)]),
                        ("fundamental", [(oracle.GRAM_EPI_F, prm)]), ("pnp", [(oracle.GRAM_PNP_GN, "model")])):
        for kind, p in kinds:
            p = g[f"m_{name}_model"][:12] if isinstance(p, str) else p
            G, cnt, bad = oracle.gram(kind, g[f"m_{name}_pts"], g[f"m_{name}_idx"], params=p, weights=g[f"m_{name}_w"], wpow=2)
            ref = g[f"m_{name}_G{kind}"]
            assert cnt == 120 and bad == 0 and np.abs(G - ref).max() <= 1e-12 * np.abs(ref).max()
    for name in ("line", "vanishing_point", "homography", "fundamental", "pnp"):
        got = oracle.solve_minimal(MODEL_CASES[name], g[f"s_{name}_pts"], g[f"s_{name}_samples"])
        assert np.array_equal(got, g[f"s_{name}_models"], equal_nan=True)


def test_graph_lists_hand_checked(oracle):
    # five collinear points at 0, 1, 2, 4, 8: the lists can be read off
    pts = np.array([[0.0, 0], [1, 0], [2, 0], [4, 0], [8, 0]])
    lists = oracle.graph_lists(pts, 2)
    assert lists.tolist() == [[1, 2], [0, 2], [1, 0], [2, 1], [3, 2]]     # ties (|2-1| = |2-... ) broken by index
    ball = oracle.graph_lists(pts, 2, radius=2.0)
    assert ball.tolist() == [[1, 2], [0, 2], [1, 0], [2, -1], [-1, -1]]   # squared distance <= 4
    off, idx, mult = oracle.graph_from_lists(ball)
    assert off.tolist() == [0, 2, 4, 7, 8, 8] and idx.tolist() == [1, 2, 0, 2, 0, 1, 3, 2]
    assert mult.tolist() == [2, 2, 2, 2, 2, 2, 1, 1]                      # 3 lists 2, 2 does not list 3


def test_gram_and_solvers_hand_checked(oracle):
    pts = np.array([[0.0, 0.0], [2.0, 0.0], [2.0, 2.0]])
    G, cnt, bad = oracle.gram(oracle.GRAM_AFFINE, pts, [0, 1, 2])
    assert cnt == 3 and bad == 0 and G.tolist() == [[3.0, 4.0, 2.0], [4.0, 8.0, 4.0], [2.0, 4.0, 4.0]]
    m = oracle.solve_minimal(oracle.LINE2D, pts, np.array([[0, 1], [1, 2], [0, 0]], np.int32))
    assert m[0].tolist() == [-0.0, 1.0, -0.0] or m[0].tolist() == [0.0, 1.0, 0.0]   # the x axis: normal (0, 1), c = 0
    assert m[1].tolist() == [-1.0, 0.0, 2.0] and np.isnan(m[2]).all()              # x = 2
    segs = np.array([[0.0, 0, 1, 1], [0.0, 2, 1, 1], [0.0, 0, 2, 2]])               # two segments meeting at (1, 1)
    v = oracle.solve_minimal(oracle.VANISHING_POINT, segs, np.array([[0, 1], [0, 2]], np.int32))
    assert np.allclose(v[0, :2] / v[0, 2], [1.0, 1.0]) and np.isnan(v[1]).all()     # collinear segments: no model


def test_device_solvers_agree_with_independent_numpy_solvers(oracle):
    """The C restatements the GPU is held to (elimination / bisection, no library calls) against the host numpy solvers
    that use different methods (LAPACK solve, SVD null space + companion eigenvalues, Kabsch alignment): same solution
    sets up to the methods' own accuracy."""
    from pyprogressivex import _estimators, datasets
    rng = np.random.default_rng(0)
    pts, gt, _ = datasets.make_homographies(seed=0)
    smp = rng.integers(0, len(pts), (300, 4)).astype(np.int32)
    m = oracle.solve_minimal(oracle.HOMOGRAPHY, pts, smp)
    ref, src = _estimators.HomographyEstimator().minimal(pts, smp)
    ok = np.nonzero(~np.isnan(m[:, 0]))[0]
    assert abs(len(ok) - len(src)) <= 2
    for i in np.intersect1d(ok, src):
        r = ref[np.nonzero(src == i)[0][0]]
        assert np.abs(m[i] - r).max() <= 1e-9 * max(1.0, np.abs(r).max())
    pts, gt, _ = datasets.make_two_view_motions(seed=0)
    sel = np.nonzero(gt == 1)[0]
    smp = np.stack([rng.choice(sel, 7, replace=False) for _ in range(200)]).astype(np.int32)
    m = oracle.solve_minimal(oracle.FUNDAMENTAL, pts, smp)
    ref, src = _estimators.FundamentalEstimator().minimal(pts, smp)
    for s in range(len(smp)):
        A = [ref[i] for i in np.nonzero(src == s)[0]]
        B = [m[3 * s + q] for q in range(3) if not np.isnan(m[3 * s + q, 0])]
        assert len(A) == len(B)
        for a in A:
            assert min(min(np.abs(a - b).max(), np.abs(a + b).max()) for b in B) < 1e-8
    x1, x2, K, gtp, poses = datasets.make_poses(n_per_object=400, n_objects=3, n_outliers=300, seed=0)
    pp, f = datasets.normalize_pnp(x1, x2, K)
    smp = np.stack([rng.choice(np.nonzero(gtp == 1 + s % 3)[0], 3, replace=False) for s in range(300)]).astype(np.int32)
    m = oracle.solve_minimal(oracle.PNP, pp, smp)
    ref, src = _estimators.PnPEstimator().minimal(pp, smp)
    same = 0
    for s in range(len(smp)):
        A = [ref[i] for i in np.nonzero(src == s)[0]]
        B = [m[4 * s + q] for q in range(4) if not np.isnan(m[4 * s + q, 0])]
        if len(A) != len(B):
            continue                                   # borderline roots: the consistency thresholds decide differently
        same += 1
        for a in A:
            assert min(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for b in B) < 1e-5
        for b in B:                                    # a rotation, and it reprojects its three sample points
            P = b.reshape(3, 4)
            assert abs(np.linalg.det(P[:, :3]) - 1.0) < 1e-9
            Xc = pp[smp[s], 2:] @ P[:, :3].T + P[:, 3]
            assert np.abs(Xc[:, :2] / Xc[:, 2:] - pp[smp[s], :2]).max() < 1e-6
    assert same >= 290


def test_gc_labeling_minimises_the_restated_energy(oracle):
    """GC-RANSAC's inlier/outlier cut [U-12]: the oracle builds the graph the way upstream's Energy::add_term1/add_term2
    would; here the energy is written down directly (integer terms, every undirected pair once) and minimised by brute
    force on <= 10 points of a 2-D line problem.  The cut must reach the minimum and, among the minimisers, return the
    smallest inlier set (sink segment = sites that still reach t; ties -> outlier)."""
    from pyprogressivex import _lib
    rng = np.random.default_rng(5)
    q = lambda x: int(np.rint(x * 4294967296.0))
    hit_tie = 0
    for trial in range(200):
        n = int(rng.integers(2, 11))
        pts = np.column_stack([rng.random(n) * 10, rng.normal(0, 1.0, n)])
        if trial % 3 == 0:
            pts[:, 1] = np.round(pts[:, 1])          # many equal residuals -> ties
        model = np.array([0.0, 1.0, 0.0])            # the line y = 0: r = |y|
        T2, lam = float(rng.choice([0.25, 1.0, 2.25])), float(rng.choice([0.1, 0.3, 0.5, 0.9]))
        graph = random_sym_graph(rng, n, 0.5)
        flags = oracle.gc_labeling(_lib.LINE2D, pts, model, T2, lam, graph)
        sq = pts[:, 1] ** 2
        inl = sq <= T2
        e = np.where(inl, sq / T2, 1.0)
        lq = 2 * int(np.rint(lam * 2147483648.0))
        off, idx = graph[0], graph[1]
        pairs = [(i, int(j)) for i in range(n) for j in idx[off[i]:off[i + 1]] if i < j]

        def energy(x):   # x[i] = 1 inlier
            tot = 0
            for i in range(n):
                if inl[i]:
                    tot += q((1 - lam) * (1 - e[i])) if x[i] == 0 else 0
                else:
                    tot += q((1 - lam) * e[i]) if x[i] == 1 else 0
            for i, j in pairs:
                if x[i] == 0 and x[j] == 0:
                    tot += 2 * q(lam * 0.25 * (e[i] + e[j]))
                elif x[i] != x[j]:
                    tot += lq
            return tot

        best, inter, nbest = None, None, 0
        for x in itertools.product([0, 1], repeat=n):
            en = energy(x)
            if best is None or en < best:
                best, inter, nbest = en, np.array(x), 1
            elif en == best:
                inter, nbest = inter & np.array(x), nbest + 1
        hit_tie += nbest > 1
        assert energy(tuple(int(f) for f in flags)) == best
        assert np.array_equal(flags, inter)
    assert hit_tie > 5


def test_greedy_labeling_u8_against_a_python_transcription_and_brute_force(oracle):
    """U-8: GCO-v3's labelling of an energy without smooth costs.  The C restatement is checked against a direct Python
    transcription of the greedy facility-location rule on random instances, against brute force over all L^n labellings
    (<= 12 sites: greedy is a heuristic - never better than the optimum, optimal on instances with well separated
    clusters), and the h = 0 case (per-site argmin) IS the optimum."""
    import itertools
    rng = np.random.default_rng(5)

    def transcription(D, h):
        n, L = D.shape
        if h <= 0:
            return np.argmin(D, axis=1).astype(np.int32)
        e = np.full(n, 1 << 35, dtype=np.int64)
        lab = np.zeros(n, dtype=np.int32)
        closed = list(range(L))
        while closed:
            deltas = [h + int(np.minimum(D[:, l] - e, 0).sum()) for l in closed]
            k = int(np.argmin(deltas))                         # first minimum = lowest label index
            if deltas[k] >= 0:
                break
            l = closed.pop(k)
            take = D[:, l] < e
            e[take] = D[take, l]
            lab[take] = l
        return lab

    def energy(D, h, lab):
        return int(D[np.arange(len(lab)), lab].sum()) + h * len(set(lab.tolist()))

    optimal = 0
    for trial in range(300):
        n, L = int(rng.integers(1, 13)), int(rng.integers(2, 5))
        D = rng.integers(0, 2 << 32, (n, L)).astype(np.int64)
        if trial % 3 == 0:                                      # ties everywhere
            D = (D >> 31) << 31
        if trial % 5 == 0:                                      # separated clusters: one cheap label per site
            D[np.arange(n), rng.integers(0, L, n)] >>= 8
        h = int(rng.integers(0, 3)) * int(rng.integers(0, 4 << 32))
        lab, e, opened = oracle.greedy_labeling(D, h)
        ref = transcription(D, h)
        assert np.array_equal(lab, ref), (trial, lab, ref)
        assert e == energy(D, h, lab) and opened >= len(set(lab.tolist()))
        if L ** n <= 30000:
            best = min(energy(D, h, np.array(c, dtype=np.int32)) for c in itertools.product(range(L), repeat=n))
            assert e >= best
            optimal += e == best
            if h == 0:
                assert e == best
    assert optimal > 100          # the heuristic finds the optimum on most of these small instances
    # golden case readable by hand: two sites, label 1 is cheaper for both but costs a second opening
    D = np.array([[10, 4], [10, 4]], dtype=np.int64) << 32
    lab, e, opened = oracle.greedy_labeling(D, 3 << 32)
    assert lab.tolist() == [1, 1] and opened == 1 and e == (8 + 3) << 32
    D = np.array([[1, 9], [9, 1]], dtype=np.int64) << 32
    assert oracle.greedy_labeling(D, 20 << 32)[0].tolist() == [0, 0]      # a second label is not worth 20
    assert oracle.greedy_labeling(D, 2 << 32)[0].tolist() == [0, 1]


def test_round2_golden_vectors(oracle):
    """tests/golden/kat_r2.npz (make_golden_r2.py): greedy labelling, PROSAC growth function, sampler draws for fixed seeds.
    Regression fixtures of the build - the reference holds none."""
    import os
    from pyprogressivex import _proposal
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_r2.npz"))
    for k in range(3):
        lab, e, opened = oracle.greedy_labeling(g[f"greedy{k}_D"], oracle.quantize(float(g[f"greedy{k}_h"][0])))
        assert np.array_equal(lab, g[f"greedy{k}_labels"]) and e == int(g[f"greedy{k}_energy"][0]) and opened == int(g[f"greedy{k}_opened"][0])
    gf = _proposal.prosac_growth_function(249, 7, 100000)
    assert np.array_equal(gf, g["prosac_growth_249_7"])
    # properties of T'_n: 1 for n <= m, strictly increasing afterwards, T'_N close to T_N (the last subset is drawn T_N times)
    assert np.all(gf[:7] == 1) and np.all(np.diff(gf[6:]) >= 1) and 0.9e5 < gf[-1] < 1.1e5
    assert np.array_equal(_proposal.prosac_growth_function(2000, 4, 100000)[::50], g["prosac_growth_2000_4"])
    assert np.array_equal(_proposal.ProsacSampler(300, np.random.default_rng(1)).draw(400, 4), g["prosac_draw"])
    pn = _proposal.ProgressiveNapsacSampler(300, np.random.default_rng(2), g["pnapsac_pts"], (640, 480, 640, 480), 4)
    draw = pn.draw(400, 4)
    assert np.array_equal(draw, g["pnapsac_draw"])
    # the first blend * n samples are local: centre k - 1 in the last column, companions from the centre's grid cells
    assert np.array_equal(draw[:150, 3][:100], np.arange(100))
    cid, members = pn.cells[-1]                     # coarsest layer (2 cells per dimension)
    local = sum(set(r[:3]).issubset(set(members[int(cid[r[3]])].tolist())) for r in draw[:150])
    assert local >= 100                             # the rest fell through to the global PROSAC sampler


def test_jacobi_eigen_solver_against_lapack_and_by_hand(oracle):
    """Round 6: the refits' small dense solve restated as cyclic Jacobi (pgxo_eigh_smallest; the device kernel runs the same operation
    order and must return the same bits: tests/test_gpu_parity.py).  Against numpy's LAPACK, tolerance stated: the eigenVALUE to
    64 eps relative to the spectral radius; the sign-normalised eigenVECTOR to 1e-12 when the smallest eigenvalue is separated from the
    next by >= 1e-3 of the spectral radius (the refits' A^T A of noisy data), and always an eigenvector in the residual sense
    |A v - val v| <= 1e-13 |A|.  Known answers: diag(3, 1, 2) -> e_2; [[2, 1], [1, 2]] -> (1, -1) / sqrt 2 with value 1."""
    vec, val, sw = oracle.eigh_smallest(np.array([np.diag([3.0, 1.0, 2.0])]))
    assert np.array_equal(vec[0], [0.0, 1.0, 0.0]) and val[0] == 1.0 and sw[0] == 0
    vec, val, _ = oracle.eigh_smallest(np.array([[[2.0, 1.0], [1.0, 2.0]]]))
    assert abs(val[0] - 1.0) < 1e-15 and np.allclose(np.abs(vec[0]), np.sqrt(0.5), atol=1e-15) and vec[0][0] * vec[0][1] < 0
    rng = np.random.default_rng(12)
    for q in (2, 3, 7, 9):
        X = rng.standard_normal((300, 30, q))
        A = np.einsum("bni,bnj->bij", X, X)
        A[:40] *= 1e-9
        A[40:80] *= 1e9
        A[80:100, :, 0] *= 1e-4              # badly scaled columns (what Hartley normalisation is there to avoid)
        A[80:100, 0, :] *= 1e-4
        vec, val, sw = oracle.eigh_smallest(A)
        w, V = np.linalg.eigh(A)
        rad = np.abs(w).max(axis=1)
        assert (np.abs(val - w[:, 0]) <= 64 * 2.3e-16 * rad).all()
        res = np.abs(np.einsum("bij,bj->bi", A, vec) - val[:, None] * vec).max(axis=1)
        assert (res <= 1e-13 * rad).all() and np.allclose((vec * vec).sum(1), 1.0, atol=1e-14)
        gap = (w[:, 1] - w[:, 0]) >= 1e-3 * rad
        sgn = np.sign((V[:, :, 0] * vec).sum(axis=1))
        assert gap.sum() > 100 and np.abs(vec * sgn[:, None] - V[:, :, 0])[gap].max() <= 1e-12
        assert sw.max() <= 12
    # NaN / Inf / zero matrices come back without hanging (NaN rows stay NaN, the zero matrix gives e_0 with value 0)
    bad = np.zeros((3, 3, 3))
    bad[0, 1, 1] = np.nan
    bad[1, 0, 2] = bad[1, 2, 0] = np.inf
    vec, val, sw = oracle.eigh_smallest(bad)
    assert np.array_equal(vec[2], [1.0, 0.0, 0.0]) and val[2] == 0.0
