"""GPU: the hand-written device scan and stable radix sort (csrc/sv_sort.hip) against numpy, at sizes either side of every tile boundary
(one-workgroup scan <= 16384 elements, many-workgroup scan in tiles of 4096, radix tiles of 2048)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, L, values=None, keys=None, bits=32):
    n = len(values if values is not None else keys)
    scan = np.zeros(n + 1, np.int32) if values is not None else None
    idx = np.zeros(max(n, 1), np.int32) if keys is not None else None
    p = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
    L.svgpu_selftest_scan_sort.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rc = L.svgpu_selftest_scan_sort(ctx.handle, n, p(values), p(scan), p(keys), bits, p(idx))
    ctx.check(rc, "svgpu_selftest_scan_sort")
    return scan, idx


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 4095, 4096, 4097, 16383, 16384, 16385, 20480, 65537, 200001, 1 << 20])
def test_scan_matches_numpy(n):
    from stella_vslam_amd import feature
    from stella_vslam_amd._lib import lib
    rng = np.random.default_rng(n)
    v = rng.integers(0, 30, n).astype(np.int32)
    scan, _ = _run(feature.Context(0), lib(), values=v)
    ref = np.concatenate([[0], np.cumsum(v, dtype=np.int64)]).astype(np.int32)
    assert np.array_equal(scan, ref)


@pytest.mark.parametrize("n,bits", [(1, 1), (2047, 6), (2048, 7), (2049, 12), (50000, 17), (300000, 18), (1 << 20, 32)])
def test_radix_sort_is_stable(n, bits):
    from stella_vslam_amd import feature
    from stella_vslam_amd._lib import lib
    rng = np.random.default_rng(n + bits)
    k = rng.integers(0, min(1 << bits, 1 << 31), n, dtype=np.int64).astype(np.uint32)
    _, idx = _run(feature.Context(0), lib(), keys=k, bits=bits)
    ref = np.argsort(k, kind="stable").astype(np.int32)
    assert np.array_equal(idx[:n], ref)
