"""CPU: the oracle's batched landmark refresh (oracle/landmark_oracle.c) against a literal numpy restatement of
data/landmark.cc:199-318 (sort-based median, Eigen-style normalisation)."""
import numpy as np

from oracle import oracle as O


def _scene(rng, n, kmax=40):
    k = np.concatenate([rng.integers(1, 8, n - 6), [1, 2, 3, kmax, 64, 65]])
    off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc = np.repeat(base, k, axis=0)
    noise = rng.integers(0, 256, (len(desc), 6))
    for j in range(6):                       # observations of one landmark: the same descriptor with a few flipped bits
        desc[np.arange(len(desc)), noise[:, j] // 8] ^= (1 << (noise[:, j] % 8)).astype(np.uint8)
    desc[off[3]:off[4]] = desc[off[3]]       # a landmark whose observations are all identical (all medians 0: first wins)
    return off, desc


def _literal_descriptor(off, desc):
    best = []
    for l in range(len(off) - 1):
        D = np.unpackbits(desc[off[l]:off[l + 1]], axis=1).astype(np.int32)
        H = (D[:, None, :] != D[None, :, :]).sum(2)
        k = len(D)
        med = np.sort(H, axis=1)[:, int(0.5 * (k - 1))]
        b, bm = 0, 256
        for i in range(k):
            if med[i] < bm:
                b, bm = i, med[i]
        best.append(b)
    return np.array(best, np.int32)


def test_compute_descriptor_against_literal():
    rng = np.random.default_rng(0)
    off, desc = _scene(rng, 300)
    best, out = O.landmarks_compute_descriptor(off, desc)
    lit = _literal_descriptor(off, desc)
    assert np.array_equal(best, lit)
    assert np.array_equal(out, desc[off[:-1] + lit])
    assert best[3] == 0 and (best > 0).sum() > 50


def test_update_geometry_against_literal():
    rng = np.random.default_rng(1)
    n = 500
    k = rng.integers(1, 12, n)
    off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    pos = rng.uniform(-5, 5, (n, 3))
    cams = np.repeat(pos, k, axis=0) + rng.normal(0, 4.0, (off[-1], 3))
    cams[off[7]] = pos[7]                      # an observation from the landmark's own position: zero vector, added un-normalised
    ref = cams[off[:-1] + rng.integers(0, k)]
    sf = np.float32(1.2) ** rng.integers(0, 8, n).astype(np.float32)
    inv_last = np.float32(1.0) / np.float32(1.2) ** np.float32(7)
    mnrm, mx, mn = O.landmarks_update_geometry(off, cams, pos, ref, sf, inv_last)
    for l in range(n):
        m = np.zeros(3)
        for o in range(off[l], off[l + 1]):
            v = pos[l] - cams[o]
            sq = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]
            m = m + (v / np.sqrt(sq) if sq > 0 else v)
        sq = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]
        m = m / np.sqrt(sq) if sq > 0 else m
        assert np.array_equal(mnrm[l], m), l
        w = pos[l] - ref[l]
        d = np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
        assert mx[l] == np.float32(d * np.float64(sf[l])) and mn[l] == np.float32(mx[l] * inv_last)
    assert np.abs(np.linalg.norm(mnrm, axis=1) - 1).max() < 1e-12
