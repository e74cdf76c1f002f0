"""Pins the oracle against the REFERENCE's own compiled code where that code builds from its own sources: util::cos / util::sin
(util/trigonometric.h), util::angle::diff (util/angle.cc) and the rBRIEF sampling pattern (feature/orb_point_pairs.h) are compiled
from /root/reference by oracle/ref_local/Makefile into oracle/_ref/libsvref.so (built by __graft_entry__.build() where the checkout
exists; the .so travels to the GPU box with the snapshot).  Everything else of the front end needs OpenCV and stays pinned by the
reference's own test vectors only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    L = C.CDLL(_SO)
    L.svref_util_cos.restype = L.svref_util_sin.restype = L.svref_angle_diff.restype = C.c_float
    L.svref_util_cos.argtypes = L.svref_util_sin.argtypes = [C.c_float]
    L.svref_angle_diff.argtypes = [C.c_float, C.c_float]
    L.svref_orb_point_pairs.restype = C.POINTER(C.c_float)
    L.svref_orb_point_pairs_size.restype = C.c_uint
    return L


def _sweep():
    rng = np.random.default_rng(11)
    dense = np.linspace(-4 * np.pi, 4 * np.pi, 400001).astype(np.float32)
    # every angle the extractor can produce: fastAtan2 degrees -> radians in float, as orb_impl.cc:96 computes it
    deg = np.arange(0, 360.0, 0.001, dtype=np.float32)
    rad = (deg * np.float32(np.pi / 180.0)).astype(np.float32)
    wild = rng.uniform(-1e4, 1e4, 200000).astype(np.float32)
    edges = np.array([0.0, -0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, -np.pi / 2, 3.14159265358979, 1.57079632679, 4.71238898038, 6.28318530718],
                     np.float32)
    edges = np.concatenate([edges, np.nextafter(edges, np.float32(10)), np.nextafter(edges, np.float32(-10))])
    return np.concatenate([dense, rad, wild, edges])


def test_util_cos_sin_bit_exact_against_the_reference(ref):
    v = _sweep()
    c, s = np.empty_like(v), np.empty_like(v)
    ref.svref_util_cos_sin_array(C.c_void_p(v.ctypes.data), len(v), C.c_void_p(c.ctypes.data), C.c_void_p(s.ctypes.data))
    L = O.lib()
    oc = np.array([L.orc_util_cos(float(x)) for x in v[::7]], np.float32)
    os_ = np.array([L.orc_util_sin(float(x)) for x in v[::7]], np.float32)
    assert np.array_equal(oc.view(np.uint32), c[::7].view(np.uint32))
    assert np.array_equal(os_.view(np.uint32), s[::7].view(np.uint32))
    # and the reference's own tolerance test (test/stella_vslam/util/trigonometric.cc:8-20) holds for its compiled code
    near = np.abs(v) <= 4 * np.pi  # the float range reduction loses digits far from 0; the reference's test stays within a few periods
    assert np.abs(c - np.cos(v.astype(np.float64)))[near].max() < 1e-3 and np.abs(s - np.sin(v.astype(np.float64)))[near].max() < 1e-3


def test_angle_diff_bit_exact_against_the_reference(ref):
    rng = np.random.default_rng(12)
    a = np.concatenate([rng.uniform(0, 360, 200000), [0, 0, 180, 180, 359.99, 0.0, 360.0, 270]]).astype(np.float32)
    b = np.concatenate([rng.uniform(0, 360, 200000), [180, 0, 0, 360, 0.0, 359.99, 0.0, 90]]).astype(np.float32)
    out = np.empty_like(a)
    ref.svref_angle_diff_array(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), len(a), C.c_void_p(out.ctypes.data))
    got = np.array([O.angle_diff(float(x), float(y)) for x, y in zip(a[::5], b[::5])], np.float32)
    assert np.array_equal(got.view(np.uint32), out[::5].view(np.uint32))


def test_rbrief_pattern_is_the_reference_table(ref):
    n = ref.svref_orb_point_pairs_size()
    assert n == 1024
    table = np.ctypeslib.as_array(ref.svref_orb_point_pairs(), shape=(n,)).copy()
    assert np.array_equal(table, np.round(table)) and np.abs(table).max() <= 15  # small integers stored as floats
    L = O.lib()
    L.orc_orb_pattern.restype = C.POINTER(C.c_int8)
    mine = np.ctypeslib.as_array(L.orc_orb_pattern(), shape=(1024,)).astype(np.float32)
    assert np.array_equal(mine, table)
    # the device kernel includes the very same table file
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = open(os.path.join(here, "oracle", "orb_pattern_i8.inc")).read()
    b = open(os.path.join(here, "stella_vslam_amd", "csrc", "orb_pattern_i8.inc")).read()
    assert a == b


# ------------------------------------------------------------------ the reference's own extractor code over the stand-in OpenCV types
# oracle/ref_local/shim: feature/orb_extractor.cc, orb_impl.cc and orb_params.cc compiled where they lie; cv::resize / GaussianBlur / FAST
# / fastAtan2 behind the stand-in headers are the oracle's restatements, so these tests pin everything the REFERENCE itself wrote.
from stella_vslam_amd import synthetic as S  # noqa: E402

KP_FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def _ref_extract(ref, img, mask=None, rects=None, scale_factor=1.2, num_levels=8, ini_thr=20, min_thr=7, min_area=800, cap=30000):
    h, w = img.shape
    kp = np.zeros((cap, 7), np.float32)
    desc = np.zeros((cap, 32), np.uint8)
    sizes = O.level_sizes(w, h, scale_factor, num_levels)
    pyr = np.zeros(sum(a * b for a, b in sizes[1:]) + 1, np.uint8)
    r = None if not rects else np.ascontiguousarray(rects, np.float32)
    ref.svref_orb_extract.restype = C.c_int
    n = ref.svref_orb_extract(C.c_void_p(img.ctypes.data), w, h, img.strides[0], None if mask is None else C.c_void_p(mask.ctypes.data),
                              0 if mask is None else mask.strides[0], C.c_float(scale_factor), num_levels, ini_thr, min_thr, min_area,
                              None if r is None else C.c_void_p(r.ctypes.data), 0 if r is None else len(r), C.c_void_p(kp.ctypes.data),
                              C.c_void_p(desc.ctypes.data), cap, C.c_void_p(pyr.ctypes.data))
    assert n >= 0
    levels, off = [], 0
    for (lw, lh) in sizes[1:]:
        levels.append(pyr[off:off + lw * lh].reshape(lh, lw))
        off += lw * lh
    return kp[:n], desc[:n], levels


def _same(ref_kp, ref_desc, k, d):
    assert len(ref_kp) == len(k)
    for j, f in enumerate(KP_FIELDS):
        assert np.array_equal(ref_kp[:, j].view(np.uint32), k[f].astype(np.float32).view(np.uint32)), f
    assert np.array_equal(ref_desc, d)


@pytest.mark.parametrize("w,h,seed,kw", [(640, 480, 1, {}), (320, 240, 2, {}), (641, 479, 3, {}), (1241, 376, 4, dict(ini_thr=12)),
                                         (200, 150, 5, dict(min_area=100)), (640, 480, 6, dict(scale_factor=1.5, num_levels=4)),
                                         (160, 140, 7, {}), (640, 480, 8, dict(ini_thr=40, min_thr=30))])  # (levels narrower than the two 19-px borders wrap the reference's unsigned arithmetic: not a case)
def test_extractor_equals_the_reference_code(ref, w, h, seed, kw):
    img = S.frame_sequence(1, w, h, seed=seed)[0]
    rk, rd, rp = _ref_extract(ref, img, **kw)
    k, d, _, pyr = O.orb_extract(img, want_pyramid=True, **kw)
    assert len(k) > 0
    _same(rk, rd, k, d)
    for a, b in zip(rp, pyr[1:]):
        assert np.array_equal(a, b)


def test_extractor_with_masks_equals_the_reference_code(ref):
    img = S.frame_sequence(1, 640, 480, seed=9)[0]
    yy, xx = np.mgrid[0:480, 0:640]
    mask = np.ones((480, 640), np.uint8)
    mask[(xx - 320) ** 2 + (yy - 200) ** 2 <= 120 ** 2] = 0
    mask[400:, :] = 0
    rk, rd, _ = _ref_extract(ref, img, mask=mask)
    k, d, _ = O.orb_extract(img, mask=mask)
    assert 0 < len(k)
    _same(rk, rd, k, d)
    # rectangle masks: create_rectangle_mask (orb_extractor.cc:138-151) runs inside the reference; the oracle gets the mask the
    # Python mirror builds (stella_vslam_amd/feature.py:rectangle_mask, the adaptor-side restatement)
    from stella_vslam_amd import feature
    rects = [[0.0, 1.0, 0.0, 0.2], [0.3, 0.55, 0.45, 0.75], [0.8, 1.0, 0.0, 1.0], [0.1, 0.1203125, 0.5, 0.503125]]  # incl. x.5 products
    rk, rd, _ = _ref_extract(ref, img, rects=rects)
    k, d, _ = O.orb_extract(img, mask=feature.rectangle_mask(rects, 640, 480))
    assert 0 < len(k)
    _same(rk, rd, k, d)


def test_extractor_on_the_reference_test_images(ref):
    Image = pytest.importorskip("PIL.Image")
    here = os.path.dirname(os.path.abspath(__file__))
    for i in (1, 2):
        img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(here, "golden", f"equirect_00{i}_gray.png"))), dtype=np.uint8)
        rk, rd, _ = _ref_extract(ref, img, min_area=1000 if i == 2 else 800, cap=60000)
        k, d, _ = O.orb_extract(img, min_area=1000 if i == 2 else 800, cap=60000)
        assert len(k) > 1000
        _same(rk, rd, k, d)


def test_orb_impl_and_params_equal_the_reference_code(ref):
    img = S.frame_sequence(1, 320, 240, seed=10)[0]
    rng = np.random.default_rng(3)
    xy = np.stack([rng.integers(19, 320 - 19, 500), rng.integers(19, 240 - 19, 500)], 1).astype(np.float32)
    xy[::7] += np.float32(0.5)  # cvRound ties
    ang = np.zeros(500, np.float32)
    desc = np.zeros((500, 32), np.uint8)
    ref.svref_orb_impl(C.c_void_p(img.ctypes.data), 320, 240, img.strides[0], C.c_void_p(xy.ctypes.data), 500, C.c_void_p(ang.ctypes.data),
                       C.c_void_p(desc.ctypes.data))
    L = O.lib()
    L.orc_ic_angle.restype = C.c_float
    for i in range(500):
        a = L.orc_ic_angle(C.c_void_p(img.ctypes.data), img.strides[0], C.c_float(xy[i, 0]), C.c_float(xy[i, 1]))
        assert np.float32(a).view(np.uint32) == ang[i].view(np.uint32)
        d = np.zeros(32, np.uint8)
        L.orc_compute_orb_descriptor(C.c_void_p(img.ctypes.data), img.strides[0], C.c_float(xy[i, 0]), C.c_float(xy[i, 1]), C.c_float(ang[i]),
                                     C.c_void_p(d.ctypes.data))
        assert np.array_equal(d, desc[i]), i
    for sf, nl in ((1.2, 8), (1.5, 4), (2.0, 3), (1.1, 12)):
        t = [np.zeros(nl, np.float32) for _ in range(4)]
        ref.svref_orb_params_tables(C.c_float(sf), nl, *[C.c_void_p(a.ctypes.data) for a in t])
        mine = O.scale_tables(sf, nl)
        for a, b in zip(t, mine):
            assert np.array_equal(a.view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed,kw", [(640, 480, 1, {}), (1241, 376, 4, dict(ini_thr=12)), (641, 479, 3, {})])
def test_gpu_extractor_equals_the_reference_code(ref, w, h, seed, kw):
    """The device path against the reference's own extractor code directly (libsvref.so travels to the GPU box with the snapshot)."""
    from stella_vslam_amd import feature
    img = S.frame_sequence(1, w, h, seed=seed)[0]
    rk, rd, _ = _ref_extract(ref, img, **kw)
    ext = feature.orb_extractor(feature.orb_params(ini_fast_thr=kw.get("ini_thr", 20)))
    k, d = ext.extract(img)
    _same(rk, rd, k, d)
