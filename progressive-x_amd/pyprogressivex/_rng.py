"""The in-repo counter-based random number generator (Philox4x32-10) and the uniform minimal-sample sampler built on it -
the numpy restatement of csrc/rng.hip.h, word for word ("a counter-based RNG identical in Python and
C++"; §8(f1): "uniform sampler with the in-repo RNG").  Replaces gcransac::sampler::UniformSampler
(progressivex_python.cpp:121, 215-245; source absent, seeded from std::random_device upstream).

Sample s of batch b under a 64-bit key is a pure function of (key, b, s): the device generates a batch inside
pgx_solve_minimal_sampled's launch, the host (and the oracle-backed context of the tests) the same rows here."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-R on arrays of 32-bit words (broadcast together): four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(x, dtype=np.uint64) & _MASK for x in (c0, c1, c2, c3)))
    k0, k1 = np.uint64(int(k0) & 0xFFFFFFFF), np.uint64(int(k1) & 0xFFFFFFFF)
    for _ in range(rounds):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = (p1 >> _S32) ^ c1 ^ k0, p1 & _MASK, (p0 >> _S32) ^ c3 ^ k1, p0 & _MASK
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def uniform_samples(key, batch, count, n, m, first=0):
    """[count, m] int64: samples first .. first + count - 1 of batch `batch` under `key` - m DISTINCT indices of range(n) each
    (csrc/rng.hip.h sample_distinct: position j takes the r-th index not taken yet, r = (word * (n - j)) >> 32)."""
    if not (1 <= m <= 8) or np.min(n) < m:
        raise ValueError("uniform_samples: need 1 <= m <= 8 <= ... and n >= m")
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    n = np.asarray(n, dtype=np.int64)                 # (a scalar, or one range per row: prosac_samples)
    s = np.arange(first, first + count, dtype=np.uint64)
    out = np.empty((count, m), dtype=np.int64)
    taken = np.empty((count, m), dtype=np.int64)      # ascending per row
    words = None
    for j in range(m):
        if j % 4 == 0:
            words = philox4x32(s & _MASK, s >> _S32, int(batch) & 0xFFFFFFFF, j // 4, key & 0xFFFFFFFF, key >> 32)
        r = ((words[j % 4].astype(np.uint64) * (n - j).astype(np.uint64)) >> _S32).astype(np.int64)
        pos = np.zeros(count, dtype=np.int64)
        for q in range(j):                            # step past every taken index <= r, in ascending order
            hit = (pos == q) & (taken[:, q] <= r)
            r = r + hit
            pos = pos + hit
        for q in range(j, 0, -1):                     # insert at pos, keeping the row ascending
            shift = pos < q
            taken[:, q] = np.where(shift, taken[:, q - 1], taken[:, q])
        taken[np.arange(count), pos] = r
        out[:, j] = r
    return out


def napsac_samples(key, batch, count, n, m, off, idx, first=0):
    """[count, m] int64: NAPSAC samples (csrc/rng.hip.h sample_napsac) - word 0 draws the centre uniformly, words 1 .. m-1 draw
    m - 1 DISTINCT entries of the centre's row of the CSR neighbourhood graph (off, idx); a centre with fewer than m - 1
    neighbours gives the row -1 .. -1 (no sample: the solvers return a NaN model, the iteration is spent)."""
    if not (2 <= m <= 8):
        raise ValueError("napsac_samples: need 2 <= m <= 8")
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    off, idx = np.asarray(off, dtype=np.int64), np.asarray(idx, dtype=np.int64)
    s = np.arange(first, first + count, dtype=np.uint64)
    blk = lambda b: philox4x32(s & _MASK, s >> _S32, int(batch) & 0xFFFFFFFF, b, key & 0xFFFFFFFF, key >> 32)   # noqa: E731
    words = blk(0)
    c = ((words[0].astype(np.uint64) * np.uint64(n)) >> _S32).astype(np.int64)
    a0, deg = off[c], off[c + 1] - off[c]
    valid = deg >= m - 1
    out = np.full((count, m), -1, dtype=np.int64)
    out[:, 0] = c
    taken = np.zeros((count, m), dtype=np.int64)
    for j in range(1, m):
        if j % 4 == 0:
            words = blk(j // 4)
        top = np.maximum(deg - (j - 1), 1).astype(np.uint64)      # (rows without a sample carry a dummy range: discarded below)
        r = ((words[j % 4].astype(np.uint64) * top) >> _S32).astype(np.int64)
        pos = np.zeros(count, dtype=np.int64)
        for q in range(j - 1):
            hit = (pos == q) & (taken[:, q] <= r)
            r = r + hit
            pos = pos + hit
        for q in range(j - 1, 0, -1):
            shift = pos < q
            taken[:, q] = np.where(shift, taken[:, q - 1], taken[:, q])
        taken[np.arange(count), pos] = r
        out[:, j] = idx[np.minimum(a0 + r, len(idx) - 1)] if len(idx) else -1
    out[~valid] = -1
    return out


def prosac_samples(key, batch, count, n, m, tops, first=0):
    """[count, m] int64: PROSAC samples (csrc/rng.hip.h sample_prosac) - row t draws m - 1 DISTINCT indices of the best
    tops[t] - 1 points plus point tops[t] - 1 (tops = the growth function's subset sizes n_k, _proposal.ProsacSampler);
    tops[t] == 0: uniform over all n points; tops[t] < m or > n: the row -1 .. -1."""
    if not (1 <= m <= 8) or n < m:
        raise ValueError("prosac_samples: need 1 <= m <= 8 and n >= m")
    tops = np.asarray(tops, dtype=np.int64)[:count]
    if len(tops) != count:
        raise ValueError("prosac_samples: one subset size per sample")
    out = np.full((count, m), -1, dtype=np.int64)
    late = tops == 0
    good = (tops >= m) & (tops <= n)
    if m > 1:
        rng_n = np.where(good, tops - 1, m - 1)                       # (rows without a PROSAC sample carry a dummy range: overwritten below)
        out[:, :m - 1] = uniform_samples(key, batch, count, rng_n, m - 1, first=first)
    out[:, m - 1] = tops - 1
    if late.any():
        out[late] = uniform_samples(key, batch, count, n, m, first=first)[late]
    out[~(late | good)] = -1
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Progressive NAPSAC on the same generator (csrc/sampler_host.hip pnapsac_draw): the statement of _proposal.ProgressiveNapsacSampler
# with every random choice taken from (key, batch, sample number)
# ---------------------------------------------------------------------------------------------------------------------
PNAPSAC_LAYERS = (16, 8, 4, 2)     # gcransac::sampler::ProgressiveNapsacSampler<4>(&points, {16, 8, 4, 2}, ...), progressivex_python.cpp:229-238


def pnapsac_cells(pts, sizes, layers=PNAPSAC_LAYERS):
    """Per grid layer: (cell id per point, {cell id: member indices in ascending = quality order}).  Cell coordinate of a point in
    dimension d = clip(floor(x_d / (size_d / div)), 0, div - 1), id = the coordinates as digits in base div."""
    pts = np.asarray(pts, dtype=np.float64)[:, :4]
    n, dims = pts.shape
    sizes = np.asarray(sizes, dtype=np.float64).reshape(-1)[:dims]
    out = []
    for div in layers:
        cell = np.clip(np.floor(pts / (sizes / div)), 0, div - 1).astype(np.int64)
        cid = np.zeros(n, dtype=np.int64)
        for d in range(dims):
            cid = cid * div + cell[:, d]
        order = np.argsort(cid, kind="stable")
        bounds = np.nonzero(np.diff(cid[order]))[0] + 1
        starts = np.concatenate([[0], bounds]) if n else np.zeros(0, dtype=np.int64)
        out.append((cid, dict(zip(cid[order][starts].tolist(), np.split(order, bounds)))))
    return out


def _words(s, batch, block, key):
    """the four words of (sample s, batch, block) under key, as Python ints"""
    s = np.array([s], dtype=np.uint64)
    return [int(x[0]) for x in philox4x32(s & _MASK, s >> _S32, int(batch) & 0xFFFFFFFF, block, key & 0xFFFFFFFF, key >> 32)]


def pnapsac_samples(key, batch, count, n, m, cells, growth_local, max_local, tops):
    """[count, m] int64: one draw of Progressive NAPSAC (a fresh sampler state: the hit counters restart with every proposal,
    progressive_x.h:290).  Sample k < min(count, max_local) is local: centre k (past n points: word 0 uniform), its hit counter
    and neighbourhood size s_p grow along growth_local, the neighbourhood is the first s_p members of the finest layer's cell that
    holds that many; the row is m - 2 DISTINCT members before the last one (word 1 + j picks the r-th not taken yet), the last
    one, the centre.  A point without a large enough cell, and every later sample, is a PROSAC row (prosac_samples' rule with
    tops[k]).  growth_local = PROSAC's growth function for m - 1 points and max_local samples; tops = n_k of the global sampler."""
    if not (2 <= m <= 8) or n < m:
        raise ValueError("pnapsac_samples: need 2 <= m <= 8 and n >= m")
    key = int(key) & 0xFFFFFFFFFFFFFFFF
    tops = np.asarray(tops, dtype=np.int64)
    growth_local = np.asarray(growth_local, dtype=np.int64)
    out = np.full((count, m), -1, dtype=np.int64)
    hits = np.zeros(n, dtype=np.int64)
    subset = np.full(n, m, dtype=np.int64)
    layer = np.zeros(n, dtype=np.int64)
    n_local = min(int(count), int(max_local))
    glob = np.zeros(count, dtype=bool)
    glob[n_local:] = True
    for k in range(n_local):
        w = _words(k, batch, 0, key)
        p = k if k < n else (w[0] * n) >> 32
        hits[p] += 1
        sp = int(subset[p])
        while sp < n and hits[p] > growth_local[sp - 1]:
            sp += 1
        subset[p] = sp
        lay, nb = int(layer[p]), None
        while lay < len(cells):
            cid, members = cells[lay]
            nb = members[int(cid[p])]
            if len(nb) >= sp:
                break
            lay += 1
            nb = None
        layer[p] = lay
        if nb is None:
            glob[k] = True
            continue
        others = nb[:sp]
        others = others[others != p]
        if len(others) < m - 1:
            glob[k] = True
            continue
        taken = []
        for j in range(m - 2):
            if (1 + j) % 4 == 0:
                w = _words(k, batch, (1 + j) // 4, key)
            r = (w[(1 + j) % 4] * (len(others) - 1 - j)) >> 32
            pos = 0
            while pos < j and taken[pos] <= r:
                r += 1
                pos += 1
            taken.insert(pos, r)
            out[k, j] = others[r]
            hits[others[r]] += 1
        out[k, m - 2] = others[-1]
        hits[others[-1]] += 1
        out[k, m - 1] = p
    gi = np.nonzero(glob)[0]
    if len(gi):
        allg = prosac_samples(key, batch, count, n, m, tops)
        out[gi] = allg[gi]
    return out
