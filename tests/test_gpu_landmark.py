"""GPU parity: batched landmark refresh (compute_descriptor, update_mean_normal_and_obs_scale_variance) through the C ABI,
bit-exact against the CPU oracle (integer medians; fp64 built from + - * / sqrt in the reference's order)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_landmark import _scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from stella_vslam_amd import data, feature
    return data, feature.Context(0)


def test_compute_descriptor_bit_exact(env):
    data, ctx = env
    rng = np.random.default_rng(3)
    off, desc = _scene(rng, 5000, kmax=200)
    best, out = data.landmarks_compute_descriptor(ctx, off, desc)
    obest, oout = O.landmarks_compute_descriptor(off, desc)
    assert np.array_equal(best, obest) and np.array_equal(out, oout)
    assert (best > 0).sum() > 500
    # single landmark, single observation; and an empty batch
    b1, o1 = data.landmarks_compute_descriptor(ctx, [0, 1], desc[:1])
    assert b1[0] == 0 and np.array_equal(o1[0], desc[0])
    b0, o0 = data.landmarks_compute_descriptor(ctx, [0], np.zeros((0, 32), np.uint8))
    assert len(b0) == 0
    from stella_vslam_amd._lib import SvgpuError
    with pytest.raises(SvgpuError):  # a landmark without observations (the reference asserts !observations_.empty())
        data.landmarks_compute_descriptor(ctx, [0, 2, 2], desc[:2])


def test_update_geometry_bit_exact(env):
    data, ctx = env
    rng = np.random.default_rng(4)
    n = 20000
    k = rng.integers(1, 12, n)
    off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    pos = rng.uniform(-5, 5, (n, 3))
    cams = np.repeat(pos, k, axis=0) + rng.normal(0, 4.0, (off[-1], 3))
    cams[off[7]] = pos[7]
    ref = cams[off[:-1] + rng.integers(0, k)]
    sf = np.float32(1.2) ** rng.integers(0, 8, n).astype(np.float32)
    inv_last = float(np.float32(1.0) / np.float32(1.2) ** np.float32(7))
    got = data.landmarks_update_mean_normal_and_obs_scale_variance(ctx, off, cams, pos, ref, sf, inv_last)
    exp = O.landmarks_update_geometry(off, cams, pos, ref, sf, inv_last)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
