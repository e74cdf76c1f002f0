// The per-frame tracker, reference beside product, on IDENTICAL data::frame / data::landmark objects.  Compiled twice
// over the same stand-in headers (shim_data/ + shim_mdrop/ + shim/):
//   oracle/_ref/libsvref_trk.so      the REFERENCE's own module/frame_tracker.cc (motion_based_track: motion-model pose, projection matcher,
//                                    2 x margin retry, num_matches_thr, pose optimisation, discard_outliers) with its own
//                                    match/projection.cc and optimize/pose_optimizer_g2o.cc (+ terminate_action.cc) where they lie; g2o's
//                                    optimize() is the oracle's pose-only LM (ref_pose_hook.h), the grid lookup / reprojection behind the
//                                    stand-in data:: / camera:: headers the oracle's (data_shim.cc).  The bodies of
//                                    tracking_module::search_local_landmarks (tracking_module.cc:533-608) and
//                                    optimize_current_frame_with_local_map (:441-455) are members of a class that drags the whole system
//                                    in: they are restated below over the same objects, line-cited.  frame_tracker's two other trackers
//                                    (BoW / robust) are linked as stubs that throw: this fixture never calls them.
//   oracle/_ref/libsvref_tdropin.so  (-DSVREF_DROP_IN) the PRODUCT's hip::tracked_frame_chain (host/drop_in/tracking_hip.cc in its
//                                    reference-tree mode) compiled against the same headers, linked to libsvgpu.so; the landmark table is
//                                    fed through hip::map_mirror exactly as the six notifications of INTEGRATION.md 3c would.
// Both export svref_track_frame with one signature; tests/test_gpu_drop_in_vs_reference.py calls both with the same arrays.
// Test infrastructure only.
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <unordered_set>
#include <vector>

#include "ref_support.h"

#ifdef SVREF_DROP_IN
#include "drop_in/map_mirror.h"
#include "drop_in/tracking_hip.h"
#else
#include "ref_pose_hook.h"
#include "stella_vslam/match/bow_tree.h"
#include "stella_vslam/match/projection.h"
#include "stella_vslam/match/robust.h"
#include "stella_vslam/module/frame_tracker.h"
#include "stella_vslam/optimize/pose_optimizer_g2o.h"
#endif

using namespace stella_vslam;
using svref::camera_fixture;
using svref::fill_observation;
using svref::make_landmark;

#ifndef SVREF_DROP_IN
// frame_tracker.cc also holds the BoW / robust trackers: their matchers are not part of this fixture
namespace stella_vslam {
namespace match {
unsigned int bow_tree::match_frame_and_keyframe(const std::shared_ptr<data::keyframe>&, data::frame&, std::vector<std::shared_ptr<data::landmark>>&) const {
    throw std::runtime_error("ref_trk_exports: bow_tree::match_frame_and_keyframe is not linked into this fixture");
}
unsigned int robust::match_frame_and_keyframe(data::frame&, const std::shared_ptr<data::keyframe>&, std::vector<std::shared_ptr<data::landmark>>&, bool) const {
    throw std::runtime_error("ref_trk_exports: robust::match_frame_and_keyframe is not linked into this fixture");
}
}  // namespace match
}  // namespace stella_vslam

namespace {
// tracking_module::search_local_landmarks (tracking_module.cc:533-608) over (curr_frm, local_landmarks)
bool search_local_landmarks(data::frame& curr_frm, const std::vector<std::shared_ptr<data::landmark>>& local_landmarks, unsigned int fixed_keyframe_id_threshold,
                            float margin, float lowe_ratio) {
    // select the landmarks which can be reprojected from the ones observed in the current frame (:535-551)
    std::unordered_set<unsigned int> curr_landmark_ids;
    for (const auto& lm : curr_frm.get_landmarks()) {
        if (!lm) continue;
        if (lm->will_be_erased()) continue;
        curr_landmark_ids.insert(lm->id_);
        lm->increase_num_observable();
    }
    bool found_proj_candidate = false;
    Vec2_t reproj;
    float x_right;
    unsigned int pred_scale_level;
    eigen_alloc_unord_map<unsigned int, Vec2_t> lm_to_reproj;
    std::unordered_map<unsigned int, float> lm_to_x_right;
    std::unordered_map<unsigned int, unsigned int> lm_to_scale;
    for (const auto& lm : local_landmarks) {  // :560-594
        if (curr_landmark_ids.count(lm->id_)) continue;
        if (lm->will_be_erased()) continue;
        if (fixed_keyframe_id_threshold > 0) {
            const auto observations = lm->get_observations();
            unsigned int temporal_observations = 0;
            for (auto obs : observations) {
                auto keyfrm = obs.first.lock();
                if (keyfrm->id_ >= fixed_keyframe_id_threshold) ++temporal_observations;
            }
            const double temporal_ratio_thr = 0.5;
            double temporal_ratio = static_cast<double>(temporal_observations) / observations.size();
            if (temporal_ratio > temporal_ratio_thr) continue;
        }
        if (curr_frm.can_observe(lm, 0.5, reproj, x_right, pred_scale_level)) {
            lm_to_reproj[lm->id_] = reproj;
            lm_to_x_right[lm->id_] = x_right;
            lm_to_scale[lm->id_] = pred_scale_level;
            lm->increase_num_observable();
            found_proj_candidate = true;
        }
    }
    if (!found_proj_candidate) return false;  // :596-599
    match::projection projection_matcher(lowe_ratio);  // :602 (0.8)
    projection_matcher.match_frame_and_landmarks(curr_frm, local_landmarks, lm_to_reproj, lm_to_x_right, lm_to_scale, margin);
    return true;
}
// tracking_module::optimize_current_frame_with_local_map (:441-455), the part that touches the frame
void optimize_current_frame_with_local_map(const optimize::pose_optimizer& pose_optimizer, data::frame& curr_frm) {
    Mat44_t optimized_pose;
    std::vector<bool> outlier_flags;
    pose_optimizer.optimize(curr_frm, optimized_pose, outlier_flags);
    curr_frm.set_pose_cw(optimized_pose);
    for (unsigned int idx = 0; idx < curr_frm.frm_obs_.undist_keypts_.size(); ++idx) {
        if (!outlier_flags.at(idx)) continue;
        curr_frm.erase_landmark_with_index(idx);
    }
}
}  // namespace
#endif

namespace {
struct Params {
    feature::orb_params p;
    Params(float scale_factor, unsigned num_levels) : p("ref", scale_factor, num_levels, 20, 7) {}
};
void put_pose(const Mat44_t& T, double* out12) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out12[4 * i + j] = T(i, j);
}
Mat44_t get_pose(const double* p12) {
    Mat44_t T = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) T(i, j) = p12[4 * i + j];
    return T;
}
}  // namespace

extern "C" {
// One tracked frame: frame_tracker::motion_based_track(curr, last, velocity), then -- whether or not it succeeded, when run_local_map is set: with
// the frame's pose replaced by pose_override12 if given, as a fallback tracker would leave it -- search_local_landmarks +
// optimize_current_frame_with_local_map.
//   landmarks      n_lm records: id, pos_w, mean normal, valid distance range, representative descriptor, has_observation, will_be_erased
//   last frame     keypoints (octave, angle; the matcher reads nothing else of them) and per keypoint the landmark INDEX it holds (-1 none)
//   current frame  the observation (keypoints, descriptors, x_right for stereo set-ups)
//   local map      indices into the landmark records
// Outputs: ret[0] = motion_based_track's return value, ret[1] = search_local_landmarks' (-1: not run); cur_lm_motion / cur_lm_final = landmark id
// per current keypoint after the first / second half (-1 none); the frame's pose after each; num_observable per landmark record at the end.
int svref_track_frame(const orc_camera* camd, int is_monocular, float true_baseline, float scale_factor, unsigned num_levels, int grid_cols, int grid_rows,
                      int n_lm, const uint32_t* lm_id, const double* lm_pos, const double* lm_normal, const float* lm_min_dist, const float* lm_max_dist,
                      const uint8_t* lm_desc, const uint8_t* lm_has_observation, const uint8_t* lm_will_be_erased, int n_last, const int32_t* last_octave,
                      const float* last_angle, const int32_t* last_lm, const double* pose_last12, int n_cur, const uint8_t* cur_desc, const float* cur_xy,
                      const int32_t* cur_octave, const float* cur_angle, const float* cur_xright, const double* cur_bearings, const double* velocity16,
                      unsigned num_matches_thr, float margin, int run_local_map, const double* pose_override12, int n_local, const int32_t* local_lm,
                      unsigned fixed_keyframe_id_threshold, float margin_local, float lowe_local, int32_t* ret, int32_t* cur_lm_motion, double* pose_motion12,
                      int32_t* cur_lm_final, double* pose_final12, int32_t* num_observable) {
    try {
        svref::forget_grids();
        Params P(scale_factor, num_levels);
        camera_fixture cam(camd, is_monocular != 0, true_baseline);
        std::vector<std::shared_ptr<data::landmark>> lms(n_lm);
        for (int i = 0; i < n_lm; ++i) {
            lms[i] = make_landmark(lm_id[i], lm_pos + 3 * i, lm_desc + 32 * (size_t)i, lm_min_dist[i], lm_max_dist[i], lm_normal + 3 * i, lm_has_observation[i] != 0);
            lms[i]->will_be_erased_ = lm_will_be_erased[i] != 0;
        }
        data::frame last(1, &cam, &P.p), curr(2, &cam, &P.p);
        last.set_pose_cw(get_pose(pose_last12));
        fill_observation(last.frm_obs_, lm_desc /* unused as keypoint descriptors */, nullptr, last_octave, last_angle, nullptr, nullptr, n_last, grid_cols, grid_rows);
        last.landmarks_.assign(n_last, nullptr);
        for (int i = 0; i < n_last; ++i)
            if (last_lm[i] >= 0) last.landmarks_[i] = lms[last_lm[i]];
        fill_observation(curr.frm_obs_, cur_desc, cur_xy, cur_octave, cur_angle, cur_xright, cur_bearings, n_cur, grid_cols, grid_rows);
        curr.landmarks_.assign(n_cur, nullptr);
        std::vector<std::shared_ptr<data::landmark>> local(n_local);
        for (int i = 0; i < n_local; ++i) local[i] = lms[local_lm[i]];
        Mat44_t velocity = Mat44_t::Identity();
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) velocity(i, j) = velocity16[4 * i + j];
        auto dump = [&](int32_t* ids, double* pose12) {
            const auto held = curr.get_landmarks();
            for (int j = 0; j < n_cur; ++j) ids[j] = held[j] ? (int32_t)held[j]->id_ : -1;
            put_pose(curr.get_pose_cw(), pose12);
        };
        ret[0] = ret[1] = -1;
#ifdef SVREF_DROP_IN
        // the table the chain reads: what data::landmark's mutators would have notified
        for (int i = 0; i < n_lm; ++i) {
            const auto& lm = lms[i];
            const Vec3_t p = lm->get_pos_in_world(), nv = lm->get_obs_mean_normal();
            const double pw[3] = {p(0), p(1), p(2)}, nrm[3] = {nv(0), nv(1), nv(2)};
            hip::map_mirror::landmark_created(lm->id_, pw);
            hip::map_mirror::set_geometry(lm->id_, nrm, lm->get_min_valid_distance(), lm->get_max_valid_distance());
            hip::map_mirror::set_descriptor(lm->id_, lm_desc + 32 * (size_t)i);
            hip::map_mirror::set_has_observation(lm->id_, lm->has_observation());
            if (lm_will_be_erased[i]) hip::map_mirror::landmark_erased(lm->id_);
        }
        hip::tracked_frame_chain chain(hip::context(), &cam, &P.p, (unsigned)grid_cols, (unsigned)grid_rows, 2, 2, 10);
        ret[0] = chain.motion_based_track(curr, last, velocity, num_matches_thr, margin, nullptr, nullptr) ? 1 : 0;
        dump(cur_lm_motion, pose_motion12);
        if (run_local_map) {
            if (pose_override12) curr.set_pose_cw(get_pose(pose_override12));
            ret[1] = chain.track_local_map(curr, local, fixed_keyframe_id_threshold, margin_local, lowe_local) ? 1 : 0;
        }
        dump(cur_lm_final, pose_final12);
        hip::forget_frame(curr.id_);
        hip::forget_frame(last.id_);
        for (int i = 0; i < n_lm; ++i) hip::map_mirror::landmark_erased(lms[i]->id_);  // (the table is process-wide: leave it as it was found)
#else
        const double intr5[5] = {camd->fx, camd->fy, camd->cx, camd->cy, camd->focal_x_baseline};
        g2o::SparseOptimizer::default_hook() = svref::make_pose_lm_hook(intr5, 0, nullptr);
        auto pose_opt = std::make_shared<optimize::pose_optimizer_g2o>(2, 2, 10);  // pose_optimizer_factory.h:23-25
        const module::frame_tracker tracker(&cam, pose_opt, num_matches_thr, false, margin);
        ret[0] = tracker.motion_based_track(curr, last, velocity) ? 1 : 0;
        dump(cur_lm_motion, pose_motion12);
        if (run_local_map) {
            if (pose_override12) curr.set_pose_cw(get_pose(pose_override12));
            const bool ok = search_local_landmarks(curr, local, fixed_keyframe_id_threshold, margin_local, lowe_local);
            ret[1] = ok ? 1 : 0;
            if (ok) optimize_current_frame_with_local_map(*pose_opt, curr);  // (tracking_module.cc:295-300: only behind a successful search)
        }
        dump(cur_lm_final, pose_final12);
        g2o::SparseOptimizer::default_hook() = nullptr;
#endif
        for (int i = 0; i < n_lm; ++i) num_observable[i] = (int32_t)lms[i]->num_observable_;
        svref::forget_grids();
        return 0;
    }
    catch (const std::exception& e) {
        std::fprintf(stderr, "svref_track_frame: %s\n", e.what());
        return -1;
    }
}
}
