"""Device-resident frame observations (svgpu_frame_*): a frame built from what the extractor left on the device must be
the frame system.cc:384-395 builds on the host, and every projection-family matcher must give the same list whether its keypoint side comes
from host arrays or from a bound resident frame -- proven by handing the bound call GARBAGE keypoint-side arrays."""
import numpy as np
import pytest

from tests import match_problems as MP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from stella_vslam_amd import feature
    return feature.Context()


def _records(view, with_size=False):
    from stella_vslam_amd.feature import KEYPOINT_DTYPE as KP_DTYPE
    k = np.zeros(len(view["xy"]), KP_DTYPE)
    k["x"], k["y"] = view["xy"][:, 0], view["xy"][:, 1]
    k["octave"], k["angle"] = view["octave"], view["angle"]
    return k


def test_adopted_extraction_equals_the_host_built_observation():
    from stella_vslam_amd import camera, data, feature, synthetic
    img = synthetic.frame_sequence(1, 640, 480, seed=77)[0]
    ext = feature.orb_extractor(feature.orb_params())
    k, d = ext.extract(img)
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    cam = camera.perspective("t", "Monocular", "Gray", 640, 480, 30.0, 458.654, 457.296, 367.215, 248.375, *dist, ctx=ext.ctx)
    obs = data.frame_observation(cam, k, d)
    rf = data.resident_frame(ext.ctx)
    und, brg = rf.adopt_extraction(cam, 64, 48)
    assert rf.size == len(k) > 1500
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(und[f], obs.undist_keypts_[f]), f
    assert np.array_equal(brg, obs.bearings_)
    # the grid and the descriptors on the device are the host observation's: a cell matcher bound to the frame equals the unbound one
    from stella_vslam_amd import match
    rng = np.random.default_rng(5)
    nq = 700
    pick = rng.integers(0, len(k), nq)
    qd = d[pick] ^ (rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 2, (nq, 32), dtype=np.uint8) * 3)
    q_xy = np.stack([und["x"][pick], und["y"][pick]], 1) + rng.normal(0, 2.0, (nq, 2)).astype(np.float32)
    q_margin = rng.uniform(4, 12, nq).astype(np.float32)
    lo, hi = np.maximum(und["octave"][pick] - 1, 0).astype(np.int32), np.minimum(und["octave"][pick] + 1, 7).astype(np.int32)
    occ = (rng.uniform(0, 1, len(k)) < 0.1).astype(np.uint8)
    m = match.projection(0.8, False, ext.ctx)
    bounds = (cam.c_.min_x, cam.c_.max_x, cam.c_.min_y, cam.c_.max_y)
    t_xy = np.stack([und["x"], und["y"]], 1)
    ref, rnum = m.match_in_cells(qd, q_xy, q_margin, d, t_xy, und["octave"], bounds, match.MATCH_RATIO_SAME_OCTAVE, 100, q_min_level=lo, q_max_level=hi, occupied=occ)
    rf.bind()
    got, num = m.match_in_cells(qd, q_xy, q_margin, np.zeros_like(d), np.zeros_like(t_xy), np.zeros_like(und["octave"]), (0.0, 1.0, 0.0, 1.0),
                                match.MATCH_RATIO_SAME_OCTAVE, 100, q_min_level=lo, q_max_level=hi, occupied=occ, grid_cols=3, grid_rows=3)
    assert rnum > 200 and num == rnum and np.array_equal(got, ref)
    # one-shot: the next call reads its arguments again
    again, anum = m.match_in_cells(qd, q_xy, q_margin, d, t_xy, und["octave"], bounds, match.MATCH_RATIO_SAME_OCTAVE, 100, q_min_level=lo, q_max_level=hi, occupied=occ)
    assert np.array_equal(again, ref)


@pytest.mark.parametrize("stereo", [False, True])
def test_projection_matchers_on_a_bound_frame(ctx, stereo):
    from stella_vslam_amd import data, match
    sc = MP.scene(seed=9 if stereo else 7, stereo=stereo)
    gcam = MP.make_cams(sc, "svgpu")

    def resident(view):
        return data.resident_frame(ctx).upload(gcam, _records(view), view["desc"], view["x_right"] if stereo else None)

    def garbage(kw, keys):
        g = dict(kw)
        for k in keys:
            if g.get(k) is not None:
                g[k] = np.zeros_like(np.asarray(g[k]))
        return g

    # match_current_and_last_frames: keypoint side = the current frame
    kw = MP.current_and_last(sc)
    cur = sc["views"][1] if np.array_equal(kw["tdesc"], sc["views"][1]["desc"]) else sc["views"][0]
    P = match.projection_flat(0.9, True, ctx)
    ref, rnum = P.match_current_and_last_frames(gcam, **kw)
    rf = resident(cur)
    rf.bind()
    got, num = P.match_current_and_last_frames(gcam, **garbage(kw, ("tdesc", "t_xy", "t_octave", "t_angle", "t_xright")))
    assert rnum > 300 and num == rnum and np.array_equal(got, ref)
    # match_frame_and_keyframe (relocalisation) and match_by_Sim3_transform
    kw = MP.frame_and_keyframe(sc)
    frm = sc["views"][1] if np.array_equal(kw["tdesc"], sc["views"][1]["desc"]) else sc["views"][0]
    ref, rnum = P.match_frame_and_keyframe(gcam, **kw)
    rf2 = resident(frm)
    rf2.bind()
    got, num = P.match_frame_and_keyframe(gcam, **garbage(kw, ("tdesc", "t_xy", "t_octave", "t_angle")))
    assert rnum > 200 and num == rnum and np.array_equal(got, ref)
    kw = MP.by_sim3(sc)
    ref, rnum = P.match_by_Sim3_transform(gcam, **kw)
    rf3 = resident(sc["views"][1])  # (kept alive: a bound frame must outlive the call it is bound for)
    rf3.bind()
    got, num = P.match_by_Sim3_transform(gcam, **garbage(kw, ("tdesc", "t_xy", "t_octave")))
    assert rnum > 100 and num == rnum and np.array_equal(got, ref)
    # fuse::detect_duplication
    kw = MP.fuse(sc)
    F = match.fuse(0.6, ctx)
    ref, rnum = F.detect_duplication(gcam, **kw)
    rf3.bind()
    got, num = F.detect_duplication(gcam, **garbage(kw, ("tdesc", "t_xy", "t_octave", "t_xright")))
    assert num == rnum and np.array_equal(got, ref)
