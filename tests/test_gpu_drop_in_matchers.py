"""The drop-in bar for the matcher family: the very cases of tests/test_ref_local_match.py -- there the REFERENCE's match::robust / bow_tree /
projection / fuse / area classes (compiled from /root/reference into oracle/_ref/libsvref.so) against the per-method oracles -- are run
again through oracle/_ref/libsvref_mdropin.so: the same fixtures (oracle/ref_local/ref_match_exports.cc), the same stand-in data:: objects,
but the PRODUCT's match::hip::* classes (stella_vslam_amd/host/drop_in/hip_backend.cc in its reference-tree mode, linked to libsvgpu.so) in
place of the reference's.  Both libraries are held to the same expected lists, so the two class families are interchangeable on these
object graphs: same matched pairs, same landmarks written into the frames, same return values."""
import ctypes as C
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "oracle", "_ref", "libsvref_mdropin.so")

_spec = importlib.util.spec_from_file_location("_ref_local_match_cases", os.path.join(_HERE, "test_ref_local_match.py"))
_T = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_T)

sc = _T.sc
sc_stereo = _T.sc_stereo


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref_mdropin.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


test_brute_force_match = _T.test_brute_force_match
test_match_for_triangulation = _T.test_match_for_triangulation
test_robust_wrappers = _T.test_robust_wrappers
test_bow_match = _T.test_bow_match
test_match_current_and_last_frames = _T.test_match_current_and_last_frames
test_match_frame_and_keyframe_projection = _T.test_match_frame_and_keyframe_projection
test_match_by_sim3_transform = _T.test_match_by_sim3_transform
test_match_keyframes_mutually = _T.test_match_keyframes_mutually
test_fuse_detect_duplication = _T.test_fuse_detect_duplication
test_match_frame_and_landmarks = _T.test_match_frame_and_landmarks
test_match_in_consistent_area = _T.test_match_in_consistent_area
