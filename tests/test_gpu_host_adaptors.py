"""The C++ adaptor classes (reference signatures, stella_vslam_amd/host/) drive the C ABI end to end on the GPU."""
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_cpp_adaptors_end_to_end():
    exe = ROOT / "stella_vslam_amd" / "host" / "test_adaptors"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(exe.parent)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adaptors ok" in out.stdout


@pytest.mark.gpu
def test_cpp_drop_in_classes_on_a_toy_map():
    """host/drop_in: match::hip::{robust, bow_tree, projection, fuse, area} and optimize::local_bundle_adjuster_hip with the reference's own
    signatures (data::frame&, shared_ptr<keyframe>, map_database*), driven on a toy map built from the stand-in data:: classes."""
    exe = ROOT / "stella_vslam_amd" / "host" / "test_drop_in"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(exe.parent)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "drop-in classes ok" in out.stdout


def test_cpp_adaptors_compile_against_standin_headers():
    """CPU-only: the adaptor library and its test program build with plain g++ (no OpenCV / g2o / HIP headers)."""
    subprocess.check_call(["make", "-C", str(ROOT / "stella_vslam_amd" / "host"), "-B"], stdout=subprocess.DEVNULL)
    assert (ROOT / "stella_vslam_amd" / "host" / "libsvgpu_host.so").exists()
    assert (ROOT / "stella_vslam_amd" / "host" / "test_drop_in").exists()


@pytest.mark.gpu
def test_tracked_frame_chain_equals_the_per_call_drop_in_classes():
    """One tracked frame through the C++ drop-in classes three ways (stella_vslam_amd/host/tracked_frame.cpp): the CHAIN (tracked_frame_chain:
    landmark ids + the device-resident landmark table fed by data::landmark's notifications, two submissions), the per-call classes on
    resident frames, and the per-call classes uploading every frame.  Same keypoints, matches, inliers and local-map visibility everywhere,
    the same optimised pose to the micrometre -- and the chain stays within its launch / synchronisation budget."""
    import ctypes as C

    import numpy as np

    from stella_vslam_amd import synthetic
    host = C.CDLL(str(ROOT / "stella_vslam_amd" / "host" / "libsvgpu_host.so"))
    host.svgpu_host_tracked_frame_counters.restype = None
    seq = np.ascontiguousarray(synthetic.frame_sequence(4, 640, 480, seed=0x5EED))
    got = {}
    for mode in (2, 1, 0):
        ms, cnt = np.zeros(8), np.zeros(8, np.int32)
        rc = host.svgpu_host_tracked_frame(C.c_void_p(seq.ctypes.data), len(seq), 640, 480, 3, mode, C.c_void_p(ms.ctypes.data), C.c_void_p(cnt.ctypes.data))
        assert rc == 0, mode
        got[mode] = cnt.copy()
    assert got[2][0] > 1800 and got[2][2] > 800 and got[2][3] > 600 and got[2][5] > 100   # keypoints, matches 1, inliers 1, matches 2
    assert np.array_equal(got[2], got[1]) and np.array_equal(got[1], got[0]), got
    assert got[2][7] < 5000   # translation error of the optimised pose, micrometres (the guess is off by 4 mm)
    launches, syncs = C.c_double(0), C.c_double(0)
    host.svgpu_host_tracked_frame_counters(C.byref(launches), C.byref(syncs))
    assert syncs.value == 2.0 and launches.value <= 15.0


@pytest.mark.gpu
def test_chain_second_half_after_a_failed_or_skipped_motion_track():
    """tracking_module's fallback paths (tracking_module.cc:326-370): motion_based_track fails or is skipped, another tracker sets the frame's
    pose, then track_local_map runs.  The chain's second half must start from the FRAME's pose: the same matches, inliers and pose bits whether
    the device holds the failed attempt's pose, a successful attempt's pose the caller replaced, or nothing at all (first frame after
    initialisation).  The fused extraction's frame rebuild keeps ref_keyfrm_ (dereferenced by the BoW fallback, :346)."""
    import ctypes as C

    import numpy as np

    from stella_vslam_amd import synthetic
    host = C.CDLL(str(ROOT / "stella_vslam_amd" / "host" / "libsvgpu_host.so"))
    seq = np.ascontiguousarray(synthetic.frame_sequence(4, 640, 480, seed=0x5EED))
    out, poses = np.zeros(8, np.int32), np.zeros(36)
    rc = host.svgpu_host_chain_fallback_test(C.c_void_p(seq.ctypes.data), len(seq), 640, 480, C.c_void_p(out.ctypes.data), C.c_void_p(poses.ctypes.data))
    assert rc == 0
    assert out[0] == 7, out            # all three second halves tracked
    assert out[7] == 1                 # ref_keyfrm_ survived the rebuild
    assert out[1] > 500 and out[2] > 400, out
    assert out[1] == out[3] == out[5] and out[2] == out[4] == out[6], out
    P = poses.reshape(3, 12)
    assert np.array_equal(P[0], P[1]) and np.array_equal(P[0], P[2])
    # ... and it is a pose of this frame: the scene's true translation to a few millimetres (the fallback pose was 2 cm off)
    gt = np.array([-3 * 3.0 * 5.0 / 500.0, -3 * 1.0 * 5.0 / 500.0, 0.0])
    assert np.abs(P[0].reshape(3, 4)[:, 3] - gt).max() < 5e-3


@pytest.mark.gpu
def test_mapping_keyframe_batched_refresh_equals_the_per_landmark_methods():
    """local_bundle_adjuster_hip::optimize writes its window back with ONE svgpu_landmarks_update_geometry (+ one svgpu_landmarks_compute_descriptor for
    the landmarks that lost an outlier observation) instead of update_mean_normal_and_obs_scale_variance / compute_descriptor per landmark
    (optimize/local_bundle_adjuster_g2o.cc:352-411, data/landmark.cc:199-318).  On an object graph of stand-in keyframes / landmarks: what the batched
    write-back stored in every live landmark equals what the landmark's own methods compute afterwards on the same objects."""
    import ctypes as C

    import numpy as np
    from stella_vslam_amd import synthetic
    lib = ROOT / "stella_vslam_amd" / "host" / "libsvgpu_host.so"
    if not lib.exists():
        subprocess.check_call(["make", "-C", str(lib.parent)])
    host = C.CDLL(str(lib))
    sc = synthetic.ba_scene(num_kf=10, num_lm=1500, obs_per_lm=5, num_fixed=3, seed=77, outlier_frac=0.05)
    P, Lm, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    a = dict(pose=np.ascontiguousarray(sc["pose_cw"], np.float64), fixed=np.ascontiguousarray(sc["pose_fixed"], np.uint8), pts=np.ascontiguousarray(sc["points"], np.float64),
             op=np.ascontiguousarray(sc["obs_pose"], np.int32), ol=np.ascontiguousarray(sc["obs_point"], np.int32), uvr=np.ascontiguousarray(sc["obs_uvr"], np.float32),
             octave=np.clip(np.rint(np.log(1.0 / np.asarray(sc["obs_inv_sigma_sq"], np.float64)) / np.log(1.44)), 0, 7).astype(np.int32),
             intr=np.ascontiguousarray(sc["intr"][0], np.float64))
    v = lambda x: C.c_void_p(x.ctypes.data)
    ms, st, chk = np.zeros(7), np.zeros(8, np.int32), np.zeros(4, np.int32)
    assert host.svgpu_host_mapping_keyframe(P, Lm, E, v(a["pose"]), v(a["fixed"]), v(a["pts"]), v(a["op"]), v(a["ol"]), v(a["uvr"]), v(a["octave"]), v(a["intr"]), 0, 1,
                                            v(ms), v(st)) == 0
    host.svgpu_host_mapping_keyframe_refresh_check(v(chk))
    compared, bad_geometry, bad_descriptor, lost = (int(x) for x in chk)
    assert st[6] > 0 and lost > 0          # outlier observations were erased: the descriptor batch had work
    assert compared > 1000 and bad_geometry == 0 and bad_descriptor == 0, chk
