"""Drop-in boundary checks that need no GPU: signatures/defaults of the five entry points, the reference's literal
error messages (bindings.cpp) and the unknown-sampler behaviour (progressivex_python.cpp:240-245)."""
import inspect

import numpy as np
import pytest

import pyprogressivex as px


def _defaults(fn):
    sig = inspect.signature(fn)
    return [(k, v.default) for k, v in sig.parameters.items() if v.kind == v.POSITIONAL_OR_KEYWORD]


def test_signatures_match_bindings_cpp():
    common = [("threshold", 4.0), ("conf", 0.5), ("spatial_coherence_weight", 0.0),
              ("neighborhood_ball_radius", 200.0), ("maximum_tanimoto_similarity", 0.4), ("max_iters", 1000),
              ("minimum_point_number", 10), ("maximum_model_number", -1), ("sampler_id", 3)]
    E = inspect.Parameter.empty
    assert _defaults(px.findHomographies) == [("corrs", E), ("w1", E), ("h1", E), ("w2", E), ("h2", E)] + common + \
        [("scoring_exponent", 2), ("do_logging", False)]                      # bindings.cpp:410-426
    assert _defaults(px.findTwoViewMotions) == [("corrs", E), ("w1", E), ("h1", E), ("w2", E), ("h2", E)] + common + \
        [("scoring_exponent", 3), ("do_logging", False)]                      # :445-461
    assert _defaults(px.findVanishingPoints) == [("lines", E), ("weights", E), ("w", E), ("h", E)] + common + \
        [("scoring_exponent", 2), ("do_logging", False)]                      # :428-443
    lines_common = [("threshold", 2.0)] + common[1:]
    assert _defaults(px.findLines) == [("points", E), ("weights", E), ("w", E), ("h", E)] + lines_common + \
        [("scoring_exponent", 2), ("do_logging", False)]                      # :476-491
    assert _defaults(px.find6DPoses) == [("x1y1", E), ("x2y2z2", E), ("K", E), ("threshold", 4.0), ("conf", 0.90),
                                         ("spatial_coherence_weight", 0.1), ("neighborhood_ball_radius", 20.0),
                                         ("maximum_tanimoto_similarity", 0.9), ("max_iters", 400),
                                         ("minimum_point_number", 6), ("maximum_model_number", -1)]   # :463-474
    assert px.findFundamentalMatrices is px.findTwoViewMotions               # north-star alias


def test_error_messages_are_the_reference_literals():
    with pytest.raises(ValueError, match=r"corrs should be an array with dims \[n,4\], n>=4"):
        px.findHomographies(np.zeros((10, 3)), 100, 100, 100, 100)
    with pytest.raises(ValueError, match=r"corrs should be an array with dims \[n,4\], n>=4"):
        px.findHomographies(np.zeros((3, 4)), 100, 100, 100, 100)
    with pytest.raises(ValueError, match=r"corrs should be an array with dims \[n,4\], n>=7"):
        px.findTwoViewMotions(np.zeros((6, 4)), 100, 100, 100, 100)
    with pytest.raises(ValueError, match=r"lines should be an array with dims \[n,4\], n>=2"):
        px.findVanishingPoints(np.zeros((1, 4)), np.array(0), 100, 100)
    with pytest.raises(ValueError, match=r"Points should be an array with dims \[n,3\], n>=2"):
        px.findLines(np.zeros((5, 3)), np.array(0), 100, 100)
    with pytest.raises(ValueError, match=r"x1y1 should be an array with dims \[n,2\], n>=3"):
        px.find6DPoses(np.zeros((2, 2)), np.zeros((2, 3)), np.eye(3))
    with pytest.raises(ValueError, match=r"x2y2z2 should be an array with dims \[n,3\], n>=3"):
        px.find6DPoses(np.zeros((5, 2)), np.zeros((5, 2)), np.eye(3))
    with pytest.raises(ValueError, match="x1y1 and x2y2z2 should be the same size"):
        px.find6DPoses(np.zeros((5, 2)), np.zeros((6, 3)), np.eye(3))
    with pytest.raises(ValueError, match=r"K should be an array with dims \[3,3\]"):
        px.find6DPoses(np.zeros((5, 2)), np.zeros((5, 3)), np.eye(4))


def test_unknown_sampler_returns_zero_models_not_an_exception(capsys):
    # findLines / findVanishingPoints: the DEFAULT sampler_id = 3 is not valid for these drivers
    lines, labels = px.findLines(np.random.default_rng(0).random((50, 2)), np.array(0), 100, 100)
    assert lines.shape == (0, 3) and labels.dtype == np.int32 and labels.shape == (50,) and not labels.any()
    assert "Unknown sampler identifier: 3" in capsys.readouterr().err
    vps, labels = px.findVanishingPoints(np.random.default_rng(0).random((20, 4)), np.array(0), 100, 100)
    assert vps.shape == (0, 3) and labels.shape == (20,)
    H, labels = px.findHomographies(np.random.default_rng(0).random((20, 4)), 1, 1, 1, 1, sampler_id=7)
    assert H.shape == (0, 3) and H.dtype == np.float64


def test_host_helpers():
    import host_graph as _graph
    from pyprogressivex import _engine, _proposal, parallel
    # progressive_x.h:495-513 incl. std::round semantics
    assert _engine.predicted_unseen_inliers(0.5, 4, 1000, 1, 5000) == \
        int(round((5000 - 1) * (1 - 0.5 ** (1 / 1000)) ** 0.25))
    # sequential replay: first strictly better wins; early exit predicate; adaptive bound stops the walk
    counts = np.array([5, 9, 9, 3, 50])
    scores = np.array([4.0, 8.0, 8.0, 2.5, 40.0])
    best, iters, hist = _proposal.replay_sequential(counts, scores, np.arange(5), 100, 2, 0.5, 1000)
    assert best == 4 and hist == [0, 1, 4]
    best, iters, hist = _proposal.replay_sequential(np.array([90, 95]), np.array([80.0, 85.0]), np.array([0, 40]),
                                                    100, 2, 0.99, 1000)
    assert best == 0 and iters <= 6          # 90% inliers => bound ~2.8 iterations: hypothesis at iteration 41 unseen
    # graph symmetrisation: directed k-NN entries -> multiplicities 1 or 2
    off, idx, mult = _graph.symmetrize(4, [0, 1, 1, 2, 3, 3], [1, 0, 2, 3, 2, 3])
    assert off.tolist() == [0, 1, 3, 5, 6] and idx.tolist() == [1, 0, 2, 1, 3, 2] and mult.tolist() == [2, 2, 1, 1, 2, 2]
    per, bounds = parallel.shard_bounds(10, 4)
    assert per == 3 and bounds == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert parallel.select_best([1.0, 5.0, 5.0, np.nan], [1, 2, 2, 3]) == 1
    assert parallel.select_best([1.0, 5.0], [0, 0]) == -1


def test_weights_length_is_validated_before_anything_runs():
    # the reference reads weights[i] for every point without a length check (solver_vanishing_point_two_lines.h:204-207):
    # a short array is a heap over-read there, a ValueError here (raised before the GPU context is touched)
    lines = np.random.default_rng(0).random((20, 4))
    with pytest.raises(ValueError, match="weights should have one entry per row"):
        px.findVanishingPoints(lines, np.ones(5), 100, 100, sampler_id=0)
    # findLines parses the weights and never uses them (progressivex_python.cpp:466-482): any length passes, as upstream
    L, lab = px.findLines(np.random.default_rng(0).random((50, 2)), np.ones(5), 100, 100)
    assert L.shape == (0, 3)


def test_samplers_return_distinct_indices_even_when_n_is_close_to_m():
    from pyprogressivex import _proposal
    rng = np.random.default_rng(0)
    for n, m in ((7, 7), (8, 7), (10, 7), (4, 4), (100, 4), (2, 2)):
        for cls in (_proposal.UniformSampler, _proposal.ProsacSampler):
            s = cls(n, rng).draw(300, m)
            assert s.shape == (300, m) and s.min() >= 0 and s.max() < n
            srt = np.sort(s, axis=1)
            assert not (srt[:, 1:] == srt[:, :-1]).any(), (cls.__name__, n, m)
    # PROSAC's first rows draw from a prefix of exactly m points: they must be permutations of range(m)
    s = _proposal.ProsacSampler(1000, rng).draw(2000, 4)
    assert sorted(s[0].tolist()) == [0, 1, 2, 3]
    # NAPSAC on a ring graph with degree 2 and m = 3: both neighbours, distinct
    n = 50
    off = np.arange(0, 2 * n + 1, 2)
    idx = np.column_stack([(np.arange(n) - 1) % n, (np.arange(n) + 1) % n]).reshape(-1)
    s = _proposal.NapsacSampler(n, rng, (off, idx)).draw(200, 3)
    assert len(s) == 200 and all(len(set(r)) == 3 for r in s.tolist())
    # the exact fallback is uniform: every index equally likely in every column
    s = _proposal.UniformSampler(8, rng).draw(40000, 7)
    assert np.abs(np.bincount(s[:, 0], minlength=8) / 40000 - 0.125).max() < 0.01


def test_refit_solver_switch_on_the_cpu_restatement(monkeypatch):
    """refit_solver="jacobi" (round 6): the refits' eigen-solves through the context's batched Jacobi solver instead of numpy.  Same
    labels and the same models to 1e-9 on a homography and a vanishing-point scene; an unknown value is refused."""
    import pyprogressivex as px
    from oracle_ctx import OracleContext
    from pyprogressivex import _api, datasets
    monkeypatch.setattr(_api, "_ctx", OracleContext())
    pts, gt, _ = datasets.make_homographies(n_per_plane=200, n_planes=3, n_outliers=200, seed=3)
    kw = dict(threshold=3.0, conf=0.99, sampler_id=0, seed=4, minimum_point_number=40, max_iters=300)
    H0, l0 = px.findHomographies(pts, 1000, 1000, 1000, 1000, **kw)
    H1, l1 = px.findHomographies(pts, 1000, 1000, 1000, 1000, refit_solver="jacobi", **kw)
    assert H0.shape == H1.shape and H0.shape[0] >= 9 and np.array_equal(l0, l1) and np.allclose(H0, H1, rtol=1e-9, atol=1e-12)
    seg, gt, _ = datasets.make_vanishing_points(n_inliers=900, n_vps=3, n_outliers=300, seed=2)
    kw = dict(threshold=1.5, conf=0.99, sampler_id=0, seed=1, minimum_point_number=100)
    V0, l0 = px.findVanishingPoints(seg, np.array(0), 1000, 1000, **kw)
    V1, l1 = px.findVanishingPoints(seg, np.array(0), 1000, 1000, refit_solver="jacobi", **kw)
    assert V0.shape == V1.shape and V0.shape[0] >= 3 and np.array_equal(l0, l1)
    assert np.allclose(np.abs((V0 * V1).sum(1)), 1.0, atol=1e-9)          # unit vectors up to sign
    with pytest.raises(ValueError):
        px.findHomographies(pts, 1000, 1000, 1000, 1000, refit_solver="eigen", threshold=3.0)
