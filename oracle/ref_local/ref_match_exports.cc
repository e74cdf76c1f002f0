// The reference's matcher methods (match/robust.cc, bow_tree.cc, projection.cc, fuse.cc, area.cc, compiled where they lie) behind C
// exports on the flat arrays the tests use.  Every export builds the object graph the method takes (stand-in data classes of shim/),
// calls the method, and flattens what it wrote back into the per-query index arrays the oracle's functions return.
// Test infrastructure only (tests/test_ref_local_match.py).
#include <set>

#include "ref_support.h"
#include "stella_vslam/solve/essential_solver.h"  // the scripted stand-in of shim_data/ (both class families call it)
#ifdef SVREF_DROP_IN
// the same fixtures around the PRODUCT's drop-in classes (stella_vslam_amd/host/drop_in/hip_backend.h, reference-tree mode, compiled against
// the same stand-in data:: headers): libsvref_mdropin.so, linked to libsvgpu.so; tests/test_gpu_drop_in_matchers.py runs the cases of
// tests/test_ref_local_match.py through it
#include "drop_in/hip_backend.h"
namespace M = stella_vslam::match::hip;
#else
#include "stella_vslam/match/area.h"
#include "stella_vslam/match/bow_tree.h"
#include "stella_vslam/match/fuse.h"
#include "stella_vslam/match/projection.h"
#include "stella_vslam/match/robust.h"
#include "stella_vslam/match/stereo.h"
namespace M = stella_vslam::match;
#endif

using namespace stella_vslam;
using svref::camera_fixture;
using svref::feat_vec;
using svref::fill_observation;
using svref::make_landmark;

namespace {
struct Params {
    feature::orb_params p;
    Params(float scale_factor, unsigned num_levels) : p("ref", scale_factor, num_levels, 20, 7) {}
};
// landmarks on the keypoints flagged in `has` (NULL = every keypoint), id = keypoint index
void attach_landmarks(std::vector<std::shared_ptr<data::landmark>>& lms, const uint8_t* has, int n) {
    lms.assign(n, nullptr);
    for (int i = 0; i < n; ++i)
        if (!has || has[i]) lms[i] = make_landmark((unsigned)i, nullptr, nullptr, 0.f, 0.f, nullptr);
}
}  // namespace

extern "C" {

// robust::brute_force_match (match/robust.cc:232-328): side 1 = the frame observation, side 2 = the keyframe (valid2 = holds a live landmark)
int svref_brute_force_match(const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                            float lowe_ratio, int check_orientation, int32_t* matched_2_in_1) {
    Params P(1.2f, 8);
    data::frame_observation fo;
    fill_observation(fo, desc1, nullptr, nullptr, angle1, nullptr, nullptr, n1, 64, 48);
    auto kf = std::make_shared<data::keyframe>(1, nullptr, &P.p);
    fill_observation(kf->frm_obs_, desc2, nullptr, nullptr, angle2, nullptr, nullptr, n2, 64, 48);
    attach_landmarks(kf->landmarks_, valid2, n2);
    std::vector<std::pair<int, int>> matches;
    const unsigned num = M::robust(lowe_ratio, check_orientation != 0).brute_force_match(fo, kf, matches);
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    for (const auto& m : matches) matched_2_in_1[m.first] = m.second;
    return (int)num;
}

// robust::match_frame_and_keyframe (match/robust.cc:194-230; `keyframes` == 0) and robust::match_keyframes (:148-192; `keyframes` == 1):
// brute force -> solve::essential_solver -> landmark assignment.  The solver is the scripted stand-in of shim_data/ (Eigen's SVD is not
// available here): `script_valid` / `script_mod` set its outcome, `rec` returns what the class asked it for
// {calls, max_num_iter, recompute, use_fixed_seed, matches handed over, bearings_1, bearings_2}.  out_lm_in_frm[i] = id of the landmark
// written at frame keypoint i (= keyframe keypoint index), -1 = none.
int svref_robust_match_wrapped(int keyframes, const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2,
                               int n2, float lowe_ratio, int check_orientation, int validate, int use_fixed_seed, int script_valid, int script_mod,
                               int32_t* out_lm_in_frm, int32_t* rec) {
    Params P(1.2f, 8);
    auto& S = solve::essential_script();
    S = solve::essential_solver_script();
    S.valid = script_valid != 0;
    S.inlier_mod = script_mod;
    auto kf2 = std::make_shared<data::keyframe>(2, nullptr, &P.p);
    fill_observation(kf2->frm_obs_, desc2, nullptr, nullptr, angle2, nullptr, nullptr, n2, 64, 48);
    kf2->frm_obs_.bearings_.resize(n2);
    attach_landmarks(kf2->landmarks_, valid2, n2);
    std::vector<std::shared_ptr<data::landmark>> matched(3, nullptr);  // a stale vector: the method must re-size it
    unsigned num = 0;
    if (keyframes) {
        auto kf1 = std::make_shared<data::keyframe>(1, nullptr, &P.p);
        fill_observation(kf1->frm_obs_, desc1, nullptr, nullptr, angle1, nullptr, nullptr, n1, 64, 48);
        kf1->frm_obs_.bearings_.resize(n1);
        num = M::robust(lowe_ratio, check_orientation != 0).match_keyframes(kf1, kf2, matched, validate != 0, use_fixed_seed != 0);
    }
    else {
        data::frame frm(1, nullptr, &P.p);
        fill_observation(frm.frm_obs_, desc1, nullptr, nullptr, angle1, nullptr, nullptr, n1, 64, 48);
        frm.frm_obs_.bearings_.resize(n1);
        num = M::robust(lowe_ratio, check_orientation != 0).match_frame_and_keyframe(frm, kf2, matched, use_fixed_seed != 0);
    }
    if ((int)matched.size() != n1) return -1;
    for (int i = 0; i < n1; ++i) out_lm_in_frm[i] = matched[i] ? (int32_t)matched[i]->id_ : -1;
    rec[0] = S.calls;
    rec[1] = (int32_t)S.last_max_num_iter;
    rec[2] = S.last_recompute;
    rec[3] = S.last_use_fixed_seed;
    rec[4] = (int32_t)S.last_num_matches;
    rec[5] = (int32_t)S.last_bearings_1;
    rec[6] = (int32_t)S.last_bearings_2;
    return (int)num;
}

// robust::match_for_triangulation (match/robust.cc:14-146) when node1 == NULL, bow_tree::match_for_triangulation (match/bow_tree.cc:11-167) otherwise.
// The epipole is computed by the method itself from keyframe 1's centre and keyframe 2's pose / camera.
int svref_match_for_triangulation(const orc_camera* cam2, const double* rot_1w, const double* trans_1w, const double* rot_2w, const double* trans_2w,
                                  const uint8_t* desc1, const float* angle1, const int32_t* octave1, const double* bearings1, const uint8_t* has_lm1,
                                  const float* xright1, int n1, const uint8_t* desc2, const float* angle2, const double* bearings2, const uint8_t* has_lm2,
                                  const float* xright2, int n2, const int32_t* node1, const int32_t* node2, const double* E_12, float scale_factor,
                                  unsigned num_levels, float residual_rad_thr, float lowe_ratio, int check_orientation, int32_t* matched_2_in_1) {
    Params P(scale_factor, num_levels);
    camera_fixture cam(cam2, true, 0.0);
    auto k1 = std::make_shared<data::keyframe>(1, &cam, &P.p), k2 = std::make_shared<data::keyframe>(2, &cam, &P.p);
    k1->set_pose_cw(svref::pose44(rot_1w, trans_1w));
    k2->set_pose_cw(svref::pose44(rot_2w, trans_2w));
    fill_observation(k1->frm_obs_, desc1, nullptr, octave1, angle1, xright1, bearings1, n1, 64, 48);
    fill_observation(k2->frm_obs_, desc2, nullptr, nullptr, angle2, xright2, bearings2, n2, 64, 48);
    attach_landmarks(k1->landmarks_, has_lm1, n1);
    attach_landmarks(k2->landmarks_, has_lm2, n2);
    if (!has_lm1) k1->landmarks_.assign(n1, nullptr);
    if (!has_lm2) k2->landmarks_.assign(n2, nullptr);
    k1->bow_feat_vec_ = feat_vec(node1, n1);
    k2->bow_feat_vec_ = feat_vec(node2, n2);
    std::vector<std::pair<unsigned int, unsigned int>> pairs;
    const Mat33_t E = svref::mat33(E_12);
    const unsigned num = node1 ? M::bow_tree(lowe_ratio, check_orientation != 0).match_for_triangulation(k1, k2, E, pairs, residual_rad_thr)
                               : M::robust(lowe_ratio, check_orientation != 0).match_for_triangulation(k1, k2, E, pairs, residual_rad_thr);
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    for (const auto& m : pairs) matched_2_in_1[m.first] = (int32_t)m.second;
    return (int)num;
}

// bow_tree::match_frame_and_keyframe (match/bow_tree.cc:169-256; side 1 = keyframe, side 2 = frame; valid2 must be NULL) and
// bow_tree::match_keyframes (:258-366; valid2 = keypoint of keyframe 2 holds a live landmark)
int svref_bow_match(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const int32_t* node1, int n1, const uint8_t* desc2, const float* angle2,
                    const uint8_t* valid2, const int32_t* node2, int n2, int keyframes, float lowe_ratio, int check_orientation, int32_t* match_1to2) {
    Params P(1.2f, 8);
    auto k1 = std::make_shared<data::keyframe>(1, nullptr, &P.p);
    fill_observation(k1->frm_obs_, desc1, nullptr, nullptr, angle1, nullptr, nullptr, n1, 64, 48);
    attach_landmarks(k1->landmarks_, valid1, n1);
    k1->bow_feat_vec_ = feat_vec(node1, n1);
    for (int i = 0; i < n1; ++i) match_1to2[i] = -1;
    std::vector<std::shared_ptr<data::landmark>> matched;
    unsigned num;
    const M::bow_tree matcher(lowe_ratio, check_orientation != 0);
    if (!keyframes) {
        data::frame frm(2, nullptr, &P.p);
        fill_observation(frm.frm_obs_, desc2, nullptr, nullptr, angle2, nullptr, nullptr, n2, 64, 48);
        frm.landmarks_.assign(n2, nullptr);
        frm.bow_feat_vec_ = feat_vec(node2, n2);
        num = matcher.match_frame_and_keyframe(k1, frm, matched);
        for (int j = 0; j < n2; ++j)  // matched[frame keypoint] = the keyframe's landmark (id = its keyframe keypoint)
            if (matched[j]) match_1to2[matched[j]->id_] = j;
    }
    else {
        auto k2 = std::make_shared<data::keyframe>(2, nullptr, &P.p);
        fill_observation(k2->frm_obs_, desc2, nullptr, nullptr, angle2, nullptr, nullptr, n2, 64, 48);
        attach_landmarks(k2->landmarks_, valid2, n2);
        k2->bow_feat_vec_ = feat_vec(node2, n2);
        num = matcher.match_keyframes(k1, k2, matched);
        for (int i = 0; i < n1; ++i)  // matched[keypoint of keyframe 1] = the landmark of keyframe 2 (id = its keypoint there)
            if (matched[i]) match_1to2[i] = (int32_t)matched[i]->id_;
    }
    return (int)num;
}

// projection::match_current_and_last_frames (match/projection.cc:95-207).  Queries = the last frame's keypoints that hold a landmark (valid).
// `occupied` current keypoints start with a landmark that has an observation (never replaced).  Out: for every CURRENT keypoint the index of the
// last-frame keypoint whose landmark it holds at the end (-1 none, -2 still the initial occupant), and the method's return value.
int svref_match_current_and_last_frames(const orc_camera* camd, const double* rot_cw, const double* trans_cw, const double* rot_lw, const double* trans_lw,
                                        int is_monocular, float true_baseline, int n_last, const double* pos_w, const uint8_t* valid, const uint8_t* lm_desc,
                                        const int32_t* octave_last, const float* angle_last, const uint8_t* lm_has_observation, float scale_factor,
                                        unsigned num_levels, float margin, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_angle,
                                        int nt, const uint8_t* occupied, const float* t_xright, int grid_cols, int grid_rows, int check_orientation,
                                        int32_t* holder_of_current) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, is_monocular != 0, true_baseline);
    data::frame curr(2, &cam, &P.p), last(1, &cam, &P.p);
    curr.set_pose_cw(svref::pose44(rot_cw, trans_cw));
    last.set_pose_cw(svref::pose44(rot_lw, trans_lw));
    fill_observation(last.frm_obs_, lm_desc /* unused as keypoint descriptors */, nullptr, octave_last, angle_last, nullptr, nullptr, n_last, grid_cols, grid_rows);
    last.landmarks_.assign(n_last, nullptr);
    for (int i = 0; i < n_last; ++i)
        if (valid[i]) last.landmarks_[i] = make_landmark((unsigned)i, pos_w + 3 * i, lm_desc + 32 * (size_t)i, 0.f, 0.f, nullptr, !lm_has_observation || lm_has_observation[i]);
    fill_observation(curr.frm_obs_, tdesc, t_xy, t_octave, t_angle, t_xright, nullptr, nt, grid_cols, grid_rows);
    curr.landmarks_.assign(nt, nullptr);
    const auto occupant = make_landmark(0xFFFFFFFEu, nullptr, nullptr, 0.f, 0.f, nullptr, true);
    for (int j = 0; j < nt; ++j)
        if (occupied && occupied[j]) curr.landmarks_[j] = occupant;
    const unsigned num = M::projection(0.0f, check_orientation != 0).match_current_and_last_frames(curr, last, margin);
    for (int j = 0; j < nt; ++j) {
        const auto lm = curr.landmarks_[j];
        holder_of_current[j] = !lm ? -1 : (lm == occupant ? -2 : (int32_t)lm->id_);
    }
    svref::forget_grids();
    return (int)num;
}

// projection::match_frame_and_keyframe (match/projection.cc:209-319).  Queries = the keyframe's keypoints that hold a landmark (valid).
// Out: for every frame keypoint the keyframe keypoint whose landmark it received (-1 none, -2 initial occupant).
int svref_match_frame_and_keyframe_projection(const orc_camera* camd, const double* rot_cw, const double* trans_cw, int n_kf, const double* pos_w,
                                              const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist, const uint8_t* lm_desc,
                                              const float* angle_kf, float scale_factor, unsigned num_levels, float margin, unsigned hamm_dist_thr,
                                              const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_angle, int nt,
                                              const uint8_t* occupied, int grid_cols, int grid_rows, int check_orientation, int32_t* holder_of_current) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, true, 0.0);
    data::frame curr(2, &cam, &P.p);
    curr.set_pose_cw(svref::pose44(rot_cw, trans_cw));
    fill_observation(curr.frm_obs_, tdesc, t_xy, t_octave, t_angle, nullptr, nullptr, nt, grid_cols, grid_rows);
    curr.landmarks_.assign(nt, nullptr);
    const auto occupant = make_landmark(0xFFFFFFFEu, nullptr, nullptr, 0.f, 0.f, nullptr, true);
    for (int j = 0; j < nt; ++j)
        if (occupied && occupied[j]) curr.landmarks_[j] = occupant;
    auto kf = std::make_shared<data::keyframe>(1, &cam, &P.p);
    fill_observation(kf->frm_obs_, lm_desc, nullptr, nullptr, angle_kf, nullptr, nullptr, n_kf, grid_cols, grid_rows);
    kf->landmarks_.assign(n_kf, nullptr);
    for (int i = 0; i < n_kf; ++i)
        if (valid[i]) kf->landmarks_[i] = make_landmark((unsigned)i, pos_w + 3 * i, lm_desc + 32 * (size_t)i, min_valid_dist[i], max_valid_dist[i], nullptr);
    const std::set<std::shared_ptr<data::landmark>> already;
    const unsigned num = M::projection(0.0f, check_orientation != 0).match_frame_and_keyframe(curr, kf, already, margin, hamm_dist_thr);
    for (int j = 0; j < nt; ++j) {
        const auto lm = curr.landmarks_[j];
        holder_of_current[j] = !lm ? -1 : (lm == occupant ? -2 : (int32_t)lm->id_);
    }
    svref::forget_grids();
    return (int)num;
}

// projection::match_by_Sim3_transform (match/projection.cc:321-416).  Queries = `landmarks` (valid = not about to be erased); `occupied` keypoints of the
// keyframe start with a match.  Out: for every keyframe keypoint the query it received (-1 none, -2 initial occupant).
int svref_match_by_sim3_transform(const orc_camera* camd, const double* sim3_cw /* 4 x 4 row-major */, int n, const double* pos_w, const uint8_t* valid,
                                  const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc,
                                  float scale_factor, unsigned num_levels, float margin, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave,
                                  int nt, const uint8_t* occupied, int grid_cols, int grid_rows, int32_t* holder_of_target) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, true, 0.0);
    auto kf = std::make_shared<data::keyframe>(1, &cam, &P.p);
    fill_observation(kf->frm_obs_, tdesc, t_xy, t_octave, nullptr, nullptr, nullptr, nt, grid_cols, grid_rows);
    Mat44_t S;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) S(i, j) = sim3_cw[4 * i + j];
    std::vector<std::shared_ptr<data::landmark>> lms;
    for (int i = 0; i < n; ++i) {
        auto lm = make_landmark((unsigned)i, pos_w + 3 * i, lm_desc + 32 * (size_t)i, min_valid_dist[i], max_valid_dist[i], mean_normal + 3 * i);
        lm->will_be_erased_ = valid && !valid[i];
        lms.push_back(lm);
    }
    const auto occupant = make_landmark(0xFFFFFFFEu, nullptr, nullptr, 0.f, 0.f, nullptr);
    std::vector<std::shared_ptr<data::landmark>> matched(nt, nullptr);
    for (int j = 0; j < nt; ++j)
        if (occupied && occupied[j]) matched[j] = occupant;
    const unsigned num = M::projection(0.0f, false).match_by_Sim3_transform(kf, S, lms, matched, margin);
    for (int j = 0; j < nt; ++j) holder_of_target[j] = !matched[j] ? -1 : (matched[j] == occupant ? -2 : (int32_t)matched[j]->id_);
    svref::forget_grids();
    return (int)num;
}

// projection::match_keyframes_mutually (match/projection.cc:418-629).  valid* = the keypoint holds a live landmark; no pair is matched beforehand.
// Out: mutual_2_in_1[i] = keypoint of keyframe 2 whose landmark keyframe 1's keypoint i received, or -1.
int svref_match_keyframes_mutually(const orc_camera* camd, const double* rot_1w, const double* trans_1w, const double* rot_2w, const double* trans_2w, float s_12,
                                   const double* rot_12, const double* trans_12, int n1, const double* pos_w1, const uint8_t* valid1, const float* min_valid1,
                                   const float* max_valid1, const uint8_t* lm_desc1, const uint8_t* desc1, const float* xy1, const int32_t* octave1, int n2,
                                   const double* pos_w2, const uint8_t* valid2, const float* min_valid2, const float* max_valid2, const uint8_t* lm_desc2,
                                   const uint8_t* desc2, const float* xy2, const int32_t* octave2, float scale_factor, unsigned num_levels, float margin,
                                   int grid_cols, int grid_rows, int32_t* mutual_2_in_1) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, true, 0.0);
    auto k1 = std::make_shared<data::keyframe>(1, &cam, &P.p), k2 = std::make_shared<data::keyframe>(2, &cam, &P.p);
    k1->set_pose_cw(svref::pose44(rot_1w, trans_1w));
    k2->set_pose_cw(svref::pose44(rot_2w, trans_2w));
    fill_observation(k1->frm_obs_, desc1, xy1, octave1, nullptr, nullptr, nullptr, n1, grid_cols, grid_rows);
    fill_observation(k2->frm_obs_, desc2, xy2, octave2, nullptr, nullptr, nullptr, n2, grid_cols, grid_rows);
    k1->landmarks_.assign(n1, nullptr);
    k2->landmarks_.assign(n2, nullptr);
    for (int i = 0; i < n1; ++i)
        if (valid1[i]) k1->landmarks_[i] = make_landmark((unsigned)i, pos_w1 + 3 * i, lm_desc1 + 32 * (size_t)i, min_valid1[i], max_valid1[i], nullptr);
    for (int i = 0; i < n2; ++i)
        if (valid2[i]) k2->landmarks_[i] = make_landmark(1000000u + (unsigned)i, pos_w2 + 3 * i, lm_desc2 + 32 * (size_t)i, min_valid2[i], max_valid2[i], nullptr);
    std::vector<std::shared_ptr<data::landmark>> matched(n1, nullptr);
    const float s = s_12;
    const unsigned num = M::projection(0.0f, false).match_keyframes_mutually(k1, k2, matched, s, svref::mat33(rot_12), svref::vec3(trans_12), margin);
    for (int i = 0; i < n1; ++i) mutual_2_in_1[i] = matched[i] ? (int32_t)(matched[i]->id_ - 1000000u) : -1;
    svref::forget_grids();
    return (int)num;
}

// fuse::detect_duplication<std::vector<...>> (match/fuse.cc:11-154).  Every second keypoint of the keyframe holds a landmark, so both output
// maps are exercised: best_idx[i] = the keyframe keypoint landmark i was fused with (from duplicated_lms_in_keyfrm or new_connections), or -1.
int svref_fuse_detect_duplication(const orc_camera* camd, const double* rot_cw, const double* trans_cw, int n, const double* pos_w, const uint8_t* valid,
                                  const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc, float scale_factor,
                                  unsigned num_levels, float margin, int do_reprojection_matching, const uint8_t* tdesc, const float* t_xy,
                                  const int32_t* t_octave, const float* t_xright, int nt, int grid_cols, int grid_rows, int32_t* best_idx) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, t_xright == nullptr, 0.0);
    auto kf = std::make_shared<data::keyframe>(1, &cam, &P.p);
    fill_observation(kf->frm_obs_, tdesc, t_xy, t_octave, nullptr, t_xright, nullptr, nt, grid_cols, grid_rows);
    kf->landmarks_.assign(nt, nullptr);
    for (int j = 0; j < nt; j += 2) kf->landmarks_[j] = make_landmark(1000000u + (unsigned)j, nullptr, nullptr, 0.f, 0.f, nullptr);
    std::vector<std::shared_ptr<data::landmark>> lms(n, nullptr);
    for (int i = 0; i < n; ++i)
        if (!valid || valid[i]) lms[i] = make_landmark((unsigned)i, pos_w + 3 * i, lm_desc + 32 * (size_t)i, min_valid_dist[i], max_valid_dist[i], mean_normal + 3 * i);
    std::unordered_map<std::shared_ptr<data::landmark>, std::shared_ptr<data::landmark>> duplicated;
    std::unordered_map<unsigned int, std::shared_ptr<data::landmark>> fresh;
    const unsigned num = M::fuse(0.0f).detect_duplication(kf, svref::mat33(rot_cw), svref::vec3(trans_cw), lms, margin, duplicated, fresh,
                                                                        do_reprojection_matching != 0);
    for (int i = 0; i < n; ++i) best_idx[i] = -1;
    for (const auto& d : duplicated) best_idx[d.first->id_] = (int32_t)(d.second->id_ - 1000000u);
    for (const auto& f : fresh) best_idx[f.second->id_] = (int32_t)f.first;
    svref::forget_grids();
    return (int)num;
}

// projection::match_frame_and_landmarks (match/projection.cc:13-93) on the maps frame::can_observe filled (given here per landmark: observable,
// reprojection, x_right, predicted level).  `occupied` frame keypoints hold a landmark with an observation.  Out: holder of every frame keypoint.
int svref_match_frame_and_landmarks(const orc_camera* camd, int is_monocular, int n, const uint8_t* observable, const double* reproj, const float* x_right,
                                    const int32_t* pred_level, const uint8_t* lm_desc, float scale_factor, unsigned num_levels, float margin, float lowe_ratio,
                                    const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_xright, int nt, const uint8_t* occupied,
                                    int grid_cols, int grid_rows, int32_t* holder_of_target) {
    svref::forget_grids();
    Params P(scale_factor, num_levels);
    camera_fixture cam(camd, is_monocular != 0, 0.0);
    data::frame frm(1, &cam, &P.p);
    fill_observation(frm.frm_obs_, tdesc, t_xy, t_octave, nullptr, t_xright, nullptr, nt, grid_cols, grid_rows);
    frm.landmarks_.assign(nt, nullptr);
    const auto occupant = make_landmark(0xFFFFFFFEu, nullptr, nullptr, 0.f, 0.f, nullptr, true);
    for (int j = 0; j < nt; ++j)
        if (occupied && occupied[j]) frm.landmarks_[j] = occupant;
    std::vector<std::shared_ptr<data::landmark>> lms;
    eigen_alloc_unord_map<unsigned int, Vec2_t> lm_to_reproj;
    std::unordered_map<unsigned int, float> lm_to_x_right;
    std::unordered_map<unsigned int, unsigned int> lm_to_scale;
    for (int i = 0; i < n; ++i) {
        lms.push_back(make_landmark((unsigned)i, nullptr, lm_desc + 32 * (size_t)i, 0.f, 0.f, nullptr, false));
        if (observable[i]) {
            lm_to_reproj[(unsigned)i] = Vec2_t(reproj[2 * i], reproj[2 * i + 1]);
            lm_to_x_right[(unsigned)i] = x_right[i];
            lm_to_scale[(unsigned)i] = (unsigned)pred_level[i];
        }
    }
    const unsigned num = M::projection(lowe_ratio, false).match_frame_and_landmarks(frm, lms, lm_to_reproj, lm_to_x_right, lm_to_scale, margin);
    for (int j = 0; j < nt; ++j) {
        const auto lm = frm.landmarks_[j];
        holder_of_target[j] = !lm ? -1 : (lm == occupant ? -2 : (int32_t)lm->id_);
    }
    svref::forget_grids();
    return (int)num;
}

// area::match_in_consistent_area (match/area.cc:8-98): the initializer's matcher.  prev_xy is updated in place as the method does.
int svref_match_in_consistent_area(const orc_camera* camd, const uint8_t* desc1, const int32_t* octave1, const float* angle1, int n1, float* prev_xy,
                                   const uint8_t* desc2, const float* xy2, const int32_t* octave2, const float* angle2, int n2, int margin, float lowe_ratio,
                                   int check_orientation, int grid_cols, int grid_rows, int32_t* matched_2_in_1) {
    svref::forget_grids();
    Params P(1.2f, 8);
    camera_fixture cam(camd, true, 0.0);
    data::frame f1(1, &cam, &P.p), f2(2, &cam, &P.p);
    fill_observation(f1.frm_obs_, desc1, nullptr, octave1, angle1, nullptr, nullptr, n1, grid_cols, grid_rows);
    fill_observation(f2.frm_obs_, desc2, xy2, octave2, angle2, nullptr, nullptr, n2, grid_cols, grid_rows);
    std::vector<cv::Point2f> prev(n1);
    for (int i = 0; i < n1; ++i) prev[i] = cv::Point2f(prev_xy[2 * i], prev_xy[2 * i + 1]);
    std::vector<int> out;
    const unsigned num = M::area(lowe_ratio, check_orientation != 0).match_in_consistent_area(f1, f2, prev, out, margin);
    for (int i = 0; i < n1; ++i) {
        matched_2_in_1[i] = out[i];
        prev_xy[2 * i] = prev[i].x, prev_xy[2 * i + 1] = prev[i].y;
    }
    svref::forget_grids();
    return (int)num;
}

#ifndef SVREF_DROP_IN
// match::stereo::compute (match/stereo.cc:20-251) on two extractor outputs: 28-byte keypoints, descriptors, the two image pyramids.
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} svref_kp28;
void svref_stereo_compute(const svref_kp28* kl, const uint8_t* dl, int nl, const svref_kp28* kr, const uint8_t* dr, int nr, const uint8_t* const* pyr_left,
                          const uint8_t* const* pyr_right, const int* lw, const int* lh, const int* ls_l, const int* ls_r, float scale_factor, int num_levels,
                          float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths) {
    Params P(scale_factor, (unsigned)num_levels);
    std::vector<cv::Mat> pl, pr;
    for (int l = 0; l < num_levels; ++l) {
        pl.push_back(cv::Mat(lh[l], lw[l], CV_8UC1, const_cast<uint8_t*>(pyr_left[l]), (size_t)ls_l[l]));
        pr.push_back(cv::Mat(lh[l], lw[l], CV_8UC1, const_cast<uint8_t*>(pyr_right[l]), (size_t)ls_r[l]));
    }
    auto kps = [](const svref_kp28* k, int n) {
        std::vector<cv::KeyPoint> v(n);
        for (int i = 0; i < n; ++i) v[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
        return v;
    };
    const std::vector<cv::KeyPoint> vl = kps(kl, nl), vr = kps(kr, nr);
    const cv::Mat descl(nl, 32, CV_8UC1, const_cast<uint8_t*>(dl), 32), descr(nr, 32, CV_8UC1, const_cast<uint8_t*>(dr), 32);
    std::vector<float> xr, dp;
    match::stereo(pl, pr, vl, vr, descl, descr, P.p.scale_factors_, P.p.inv_scale_factors_, focal_x_baseline, true_baseline).compute(xr, dp);
    for (int i = 0; i < nl; ++i) {
        stereo_x_right[i] = xr[i];
        depths[i] = dp[i];
    }
}

#endif
}  // extern "C"
