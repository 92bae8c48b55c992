"""The in-repo counter-based generator: Philox4x32-10 against the published known-answer vectors
(Random123's kat_vectors: Salmon, Moraes, Dror, Shaw, SC'11), three restatements against each other - numpy
(pyprogressivex/_rng.py), C (oracle/pgx_oracle.c) and, on the GPU, the device kernel (csrc/rng.hip.h) - and the uniform
minimal-sample sampler built on it: distinct indices, in range, uniform, a pure function of (key, batch, sample)."""
import numpy as np
import pytest

from pyprogressivex import _lib, _proposal, _rng

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers(oracle):
    for ctr, key, want in KAT:
        assert tuple(int(x) for x in _rng.philox4x32(*ctr, *key)) == want
        assert tuple(oracle.philox4x32(ctr, key)) == want
    rng = np.random.default_rng(0)
    for _ in range(200):
        ctr, key = rng.integers(0, 1 << 32, 4), rng.integers(0, 1 << 32, 2)
        assert [int(x) for x in _rng.philox4x32(*ctr, *key)] == oracle.philox4x32(ctr, key)


@pytest.mark.parametrize("n,m", [(2, 2), (7, 7), (9, 7), (10, 3), (1000, 4), (1000003, 3), (2 ** 31 - 1, 8)])
def test_uniform_samples_numpy_equals_c(oracle, n, m):
    key, batch = 0x1234567890ABCDEF, 5
    a = _rng.uniform_samples(key, batch, 3000, n, m, first=7)
    assert np.array_equal(a, oracle.sample_uniform(key, batch, 7, 3000, n, m))
    srt = np.sort(a, axis=1)
    assert a.min() >= 0 and a.max() < n and (srt[:, 1:] != srt[:, :-1]).all()
    assert np.array_equal(a[10:20], _rng.uniform_samples(key, batch, 10, n, m, first=17))          # a pure function of the sample number
    assert not np.array_equal(a, _rng.uniform_samples(key, batch + 1, 3000, n, m, first=7))


def test_uniform_samples_are_uniform():
    s = _rng.uniform_samples(99, 0, 400000, 9, 7)
    for j in (0, 3, 6):            # every position uniform over the 9 indices: 7 / 9 of the mass each way
        f = np.bincount(s[:, j], minlength=9) / s.shape[0]
        assert np.abs(f - 1.0 / 9.0).max() < 0.004
    pairs = np.bincount(s[:, 0] * 9 + s[:, 1], minlength=81).reshape(9, 9) / s.shape[0]
    assert np.abs(pairs[~np.eye(9, dtype=bool)] - 1.0 / 72.0).max() < 0.002 and np.all(np.diag(pairs) == 0)


def test_philox_sampler_batches():
    smp = _proposal.PhiloxUniformSampler(500, np.random.default_rng(3))
    a, b = smp.draw(100, 4), smp.draw(100, 4)
    assert a.shape == (100, 4) and not np.array_equal(a, b) and smp.last == (1, 100, 4)
    again = _proposal.PhiloxUniformSampler(500, np.random.default_rng(3))
    assert np.array_equal(again.draw(100, 4), a)
    assert _proposal.PhiloxUniformSampler(3, np.random.default_rng(0)).draw(10, 4).shape == (0, 4)


def test_napsac_samples_are_a_centre_and_distinct_members_of_its_list(oracle):
    pts = np.random.default_rng(0).random((600, 4)) * 100
    off, idx, _ = oracle.graph_build(pts, 0, radius=25.0, k=5)     # symmetric lists of the 5 nearest inside the ball: 0 .. ~10 neighbours
    for m in (2, 4, 7):
        s = _rng.napsac_samples(77, 2, 3000, 600, m, off, idx)
        deg = np.diff(off)
        ok = s[:, 0] >= 0
        assert ok.any() and (m < 7 or (~ok).any())                     # some centres are short of 6 neighbours
        assert (s[~ok] == -1).all()
        for row in s[ok][:400]:
            members = idx[off[row[0]]:off[row[0] + 1]]
            assert deg[row[0]] >= m - 1 and np.isin(row[1:], members).all() and len(set(row.tolist())) == m
        assert np.array_equal(s[50:60], _rng.napsac_samples(77, 2, 10, 600, m, off, idx, first=50))
        assert np.array_equal(s, oracle.sample_napsac(77, 2, 0, 3000, 600, off, idx, m))          # the C restatement draws the same rows
    smp = _proposal.PhiloxNapsacSampler(600, np.random.default_rng(1), (off, idx))
    a = smp.draw(500, 4)
    assert a.shape == (500, 4) and smp.last == (0, 500, 4) and smp.kind == "napsac"


def test_prosac_samples_follow_the_subset_sizes(oracle):
    """PROSAC on the in-repo generator: m - 1 distinct indices below n_k - 1 plus point n_k - 1; 0 = uniform over all points;
    numpy == C restatement; the sampler class feeds the growth function's sizes and restarts the sample numbers per draw."""
    rng = np.random.default_rng(0)
    n = 500
    for m in (1, 2, 3, 4, 7):
        tops = rng.integers(m, n + 1, 4000)
        tops[::7] = 0
        tops[5], tops[6], tops[8], tops[9] = m, n, (m - 1 if m > 1 else n + 1), n + 1
        a = _rng.prosac_samples(0x1234567890ABCDEF, 3, 4000, n, m, tops)
        assert np.array_equal(a, oracle.sample_prosac(0x1234567890ABCDEF, 3, 0, 4000, n, tops, m))
        good, late = (tops >= m) & (tops <= n), tops == 0
        assert (a[~(good | late)] == -1).all() and (a[good | late] >= 0).all()
        assert all(len(set(r)) == m for r in a[good | late].tolist())
        assert (a[good][:, -1] == tops[good] - 1).all() and (a[good][:, :-1] < (tops[good] - 1)[:, None]).all()
        assert np.array_equal(a[late], _rng.uniform_samples(0x1234567890ABCDEF, 3, 4000, n, m)[late])
        assert np.array_equal(a[100:150], _rng.prosac_samples(0x1234567890ABCDEF, 3, 50, n, m, tops[100:150], first=100))
    smp = _proposal.PhiloxProsacSampler(300, np.random.default_rng(2))
    a = smp.draw(2000, 4)
    ref = _proposal.ProsacSampler(300, np.random.default_rng(2))
    assert smp.kind == "prosac" and smp.last == (0, 2000, 4) and np.array_equal(smp.tops, ref.subset_sizes(1, 2000, 4))
    assert np.array_equal(a[:, 3], smp.tops - 1) and smp.tops[0] == 4 and smp.tops[-1] > 4 and (np.diff(smp.tops) >= 0).all()
    b = smp.draw(2000, 4)                                              # the next proposal: same subset sizes, new members
    assert np.array_equal(b[:, 3], a[:, 3]) and not np.array_equal(a[50:], b[50:])
    late = _proposal.PhiloxProsacSampler(300, np.random.default_rng(2), convergence_iterations=10)
    c = late.draw(50, 4)
    assert (late.tops[10:] == 0).all() and (late.tops[:10] > 0).all() and len(np.unique(c[10:, 3])) > 10


@pytest.mark.parametrize("n,m,count,variant", [(300, 7, 1000, "plain"), (187, 7, 10000, "plain"), (50, 4, 400, "duplicates"), (2000, 4, 3000, "outside"),
                                               (8, 7, 50, "plain"), (40, 2, 100, "plain"), (3000, 7, 4000, "clustered")])
def test_progressive_napsac_native_draw_equals_the_numpy_restatement(oracle, n, m, count, variant):
    """Progressive NAPSAC on the in-repo generator: libpgx.so's host code (csrc/sampler_host.hip, pgx_pnapsac_*; no GPU involved)
    against _rng.pnapsac_samples and the oracle's C restatement (pgxo_sample_pnapsac) row for row; the grid cells are those of the numpy-stream sampler of _proposal.py; rows are m
    distinct indices; a local row ends with (the last member of the centre's neighbourhood, the centre = the sample number)."""
    rng = np.random.default_rng(n + m)
    sizes = [1024.0, 768.0, 1024.0, 768.0]
    pts = rng.random((n, 4)) * sizes
    if variant == "duplicates":
        pts[:20] = pts[0]
    if variant == "outside":
        pts[:, 0] = 2000.0                     # beyond the image: clipped into the last cell
        pts[::7, 1] = -5.0
    if variant == "clustered":
        pts[:, :2] = pts[:, :2] * 0.05 + 300.0
    ref = _proposal.ProgressiveNapsacSampler(n, rng, pts, sizes, m)
    cells = _rng.pnapsac_cells(pts, sizes)
    for (c1, m1), (c2, m2) in zip(cells, ref.cells):
        assert np.array_equal(c1, c2) and set(m1) == set(m2) and all(np.array_equal(m1[k], m2[k]) for k in m2)
    tops = ref.prosac.subset_sizes(1, count, m)
    nat = _lib.PnapsacSampler(pts, sizes, m)
    key = int(rng.integers(0, 1 << 63))
    for batch in (0, 3):
        a = _rng.pnapsac_samples(key, batch, count, n, m, cells, ref.growth_local, ref.max_local, tops)
        b = nat.draw(key, batch, count, tops, ref.growth_local, ref.max_local)
        assert b.dtype == np.int64 and np.array_equal(a, b)
        assert np.array_equal(a, oracle.sample_pnapsac(key, batch, count, pts, sizes, m, tops, ref.growth_local, ref.max_local))
        assert (b >= 0).all() and (b < n).all() and all(len(set(r)) == m for r in b.tolist())
    assert not np.array_equal(a, nat.draw(key, 0, count, tops, ref.growth_local, ref.max_local))     # another batch, other rows
    n_local = min(count, ref.max_local)
    local = b[:n_local][b[:n_local, m - 1] == np.arange(n_local)]          # (rows that fell back to PROSAC end with n_k - 1 instead)
    assert len(local) > 0 or n < 2 * m
    nat.close()


def test_progressive_napsac_sampler_class_and_error_paths():
    rng = np.random.default_rng(5)
    pts = rng.random((120, 4)) * 500
    s = _proposal.PhiloxProgressiveNapsacSampler(120, np.random.default_rng(1), pts, [500, 500, 500, 500], 4)
    a = s.draw(300, 4)
    s.reset()
    b = s.draw(300, 4)
    assert a.shape == (300, 4) and not np.array_equal(a, b)                       # every proposal: a new batch
    s2 = _proposal.PhiloxProgressiveNapsacSampler(120, np.random.default_rng(1), pts, [500, 500, 500, 500], 4)
    assert np.array_equal(a, s2.draw(300, 4))                                     # a function of the call's seed
    with pytest.raises(ValueError):
        s.draw(10, 5)
    assert _proposal.PhiloxProgressiveNapsacSampler(3, np.random.default_rng(1), pts[:3], [500] * 4, 4).draw(10, 4).shape == (0, 4)
    with pytest.raises(_lib.PgxError):
        _lib.PnapsacSampler(pts, [500] * 4, 1)                                    # sample size 1: no neighbourhood to draw from
    with pytest.raises(_lib.PgxError):
        _lib.PnapsacSampler(pts, [500] * 4, 4, layers=(0,))
    nat = _lib.PnapsacSampler(pts, [500] * 4, 4)
    with pytest.raises(ValueError):
        nat.draw(1, 0, 50, np.zeros(10, np.int32), np.ones(120, np.int64), 60)   # fewer subset sizes than samples


@pytest.mark.gpu
@pytest.mark.parametrize("name,m", [("line", 2), ("pnp", 3), ("homography", 4), ("fundamental", 7)])
def test_device_prosac_draws_the_same_rows_and_models(gpu_ctx, name, m):
    from helpers import make_case
    n = 5000
    mt, pts, models, thr = make_case(name, n, 2, seed=9)
    gpu_ctx.set_points(mt, pts)
    key, batch, S = 0x0123456789ABCDEF, 5, 3000
    smp = _proposal.PhiloxProsacSampler(n, np.random.default_rng(4), convergence_iterations=2500)
    smp.key, smp.batch = key, batch
    want = smp.draw(S, m)
    assert (smp.tops[2500:] == 0).all() and smp.tops[0] == m
    with pytest.raises(_lib.PgxError, match="pgx_sampler_prosac_set"):
        gpu_ctx.solve_minimal_sampled(key, batch, S, sampler="prosac")                 # no table for these points yet
    gpu_ctx.sampler_prosac_set(smp.tops)
    got, rows = gpu_ctx.solve_minimal_sampled(key, batch, S, fetch=True, fetch_samples=True, sampler="prosac")
    assert np.array_equal(rows, want)
    assert np.array_equal(got, gpu_ctx.solve_minimal(want.astype(np.int32)), equal_nan=True)
    with pytest.raises(_lib.PgxError, match="entries"):
        gpu_ctx.solve_minimal_sampled(key, batch, S + 1, sampler="prosac")             # more samples than the table holds
    with pytest.raises(_lib.PgxError, match="outside"):
        gpu_ctx.sampler_prosac_set(np.array([m, n + 1], dtype=np.int32))
    odd = smp.tops.copy()
    odd[7] = m - 1                                                                     # a size the table never produces: no sample
    gpu_ctx.sampler_prosac_set(odd)
    got, rows = gpu_ctx.solve_minimal_sampled(key, batch, S, fetch=True, fetch_samples=True, sampler="prosac")
    assert np.array_equal(rows, _rng.prosac_samples(key, batch, S, n, m, odd)) and (rows[7] == -1).all()
    gpu_ctx.set_points(mt, pts)                                                        # new points: the table is gone
    with pytest.raises(_lib.PgxError, match="pgx_sampler_prosac_set"):
        gpu_ctx.solve_minimal_sampled(key, batch, S, sampler="prosac")


@pytest.mark.gpu
@pytest.mark.parametrize("name,m", [("line", 2), ("homography", 4), ("fundamental", 7)])
def test_device_napsac_draws_the_same_rows_and_models(gpu_ctx, name, m):
    from helpers import make_case
    mt, pts, models, thr = make_case(name, 3000, 2, seed=5)
    gpu_ctx.set_points(mt, pts)
    off, idx, _ = gpu_ctx.graph_build(pts, _lib.GRAPH_KNN_IN_BALL, radius=40.0, k=5)
    key, batch, S = 0xABCDEF0123456789, 3, 1500
    got, smp = gpu_ctx.solve_minimal_sampled(key, batch, S, fetch=True, fetch_samples=True, sampler="napsac")
    want = _rng.napsac_samples(key, batch, S, len(pts), m, off, idx)
    assert np.array_equal(smp, want) and (want[:, 0] >= 0).any()
    ref = gpu_ctx.solve_minimal(want.astype(np.int32))
    assert np.array_equal(got, ref, equal_nan=True)
    slots = got.shape[0] // S
    assert np.isnan(got.reshape(S, slots, -1)[want[:, 0] < 0]).all()    # no sample -> NaN models in every slot


@pytest.mark.gpu
@pytest.mark.parametrize("name,m", [("line", 2), ("vanishing_point", 2), ("pnp", 3), ("homography", 4), ("fundamental", 7)])
@pytest.mark.parametrize("n", [8, 1000, 100003])
def test_device_sampler_draws_the_same_rows_and_models(gpu_ctx, name, m, n):
    from helpers import make_case
    mt, pts, models, thr = make_case(name, n, 2, seed=n)
    gpu_ctx.set_points(mt, pts)
    key, batch, S = 0xFEEDFACE12345678, 11, 777
    got, smp = gpu_ctx.solve_minimal_sampled(key, batch, S, fetch=True, fetch_samples=True)
    want = _rng.uniform_samples(key, batch, S, n, m)
    assert np.array_equal(smp, want)
    ref = gpu_ctx.solve_minimal(want.astype(np.int32))
    assert np.array_equal(got, ref, equal_nan=True)          # the same solver on the same samples: bit for bit
