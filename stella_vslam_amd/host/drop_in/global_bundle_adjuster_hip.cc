// optimize::global_bundle_adjuster_hip (global_bundle_adjuster_hip.h): gather -> svgpu_global_ba -> the reference's post-conditions.
#include "drop_in/global_bundle_adjuster_hip.h"

#include <cmath>
#include <cstring>
#include <unordered_map>

namespace stella_vslam {
namespace optimize {

namespace {
using kf_ptr = std::shared_ptr<data::keyframe>;
using lm_ptr = std::shared_ptr<data::landmark>;
using mk_ptr = std::shared_ptr<data::marker>;

struct flat_graph {  // what optimize_impl (global_bundle_adjuster.cc:26-192) puts into the optimizer, as the arrays of svgpu_ba_problem
    std::unordered_map<unsigned int, int> pose_of;   // keyframe id -> pose slot (shot_vertex_container: keyed by id)
    std::vector<double> pose_cw, intr, points;
    std::vector<uint8_t> pose_fixed, point_fixed;
    std::vector<int> point_of_lm;                    // lms[i] -> point slot, -1 = no vertex (or removed again: no edge)
    std::unordered_map<unsigned int, int> point_of_marker;  // marker id -> first of its four point slots
    std::vector<int32_t> obs_pose, obs_point;
    std::vector<float> obs_uvr, obs_w, obs_huber;
    std::vector<double> pose_out, points_out;
};

void gather(flat_graph& g, const std::vector<kf_ptr>& keyfrms, const std::vector<lm_ptr>& lms, const std::vector<mk_ptr>& markers, std::vector<bool>& is_optimized_lm,
            const bool use_huber_kernel, const bool fix_markers) {
    for (const auto& keyfrm : keyfrms) {  // :55-66
        if (!keyfrm || keyfrm->will_be_erased()) continue;
        if (g.pose_of.count(keyfrm->id_)) continue;
        g.pose_of[keyfrm->id_] = (int)g.pose_fixed.size();
        g.pose_fixed.push_back(keyfrm->graph_node_->is_spanning_root() ? 1 : 0);
        const Mat44_t T = keyfrm->get_pose_cw();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) g.pose_cw.push_back(T(i, j));
        const svgpu_camera c = hip::to_svgpu_camera(keyfrm->camera_);
        if (c.model == SVGPU_CAM_EQUIRECTANGULAR) g.intr.insert(g.intr.end(), {0.0, 0.0, (double)c.cols, (double)c.rows, 0.0});
        else g.intr.insert(g.intr.end(), {c.fx, c.fy, c.cx, c.cy, c.focal_x_baseline});
    }
    constexpr float chi_sq_2D = 5.99146;
    const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D);
    constexpr float chi_sq_3D = 7.81473;
    const float sqrt_chi_sq_3D = std::sqrt(chi_sq_3D);
    g.point_of_lm.assign(lms.size(), -1);
    for (unsigned int i = 0; i < lms.size(); ++i) {  // :82-128
        const auto& lm = lms.at(i);
        if (!lm || lm->will_be_erased()) continue;
        const int slot = (int)g.point_fixed.size();
        const size_t first_edge = g.obs_pose.size();
        for (const auto& obs : lm->get_observations()) {
            const auto keyfrm = obs.first.lock();
            const auto idx = obs.second;
            if (!keyfrm || keyfrm->will_be_erased()) continue;
            const auto it = g.pose_of.find(keyfrm->id_);
            if (it == g.pose_of.end()) continue;
            const auto& undist_keypt = keyfrm->frm_obs_.undist_keypts_.at(idx);
            const float x_right = keyfrm->frm_obs_.stereo_x_right_.empty() ? -1.0f : keyfrm->frm_obs_.stereo_x_right_.at(idx);
            g.obs_pose.push_back(it->second);
            g.obs_point.push_back(slot);
            g.obs_uvr.insert(g.obs_uvr.end(), {undist_keypt.pt.x, undist_keypt.pt.y, x_right});
            g.obs_w.push_back(keyfrm->orb_params_->inv_level_sigma_sq_.at(undist_keypt.octave));
            const float sqrt_chi_sq = keyfrm->camera_->setup_type_ == camera::setup_type_t::Monocular ? sqrt_chi_sq_2D : sqrt_chi_sq_3D;
            g.obs_huber.push_back(use_huber_kernel ? sqrt_chi_sq : 0.0f);
        }
        if (g.obs_pose.size() == first_edge) {  // :124-127: the vertex is removed again
            is_optimized_lm.at(i) = false;
            continue;
        }
        g.point_of_lm[i] = slot;
        g.point_fixed.push_back(0);
        const Vec3_t p = lm->get_pos_in_world();
        for (int k = 0; k < 3; ++k) g.points.push_back(p(k));
    }
    for (const auto& mkr : markers) {  // :131-181
        if (!mkr) continue;
        if (!fix_markers && !mkr->keep_fixed_ && !mkr->initialized_before_) continue;
        const int first = (int)g.point_fixed.size();
        g.point_of_marker[mkr->id_] = first;
        for (unsigned int corner_idx = 0; corner_idx < 4; ++corner_idx) {
            g.point_fixed.push_back(fix_markers || mkr->keep_fixed_ ? 1 : 0);
            const Vec3_t p = mkr->corners_pos_w_.at(corner_idx);
            for (int k = 0; k < 3; ++k) g.points.push_back(p(k));
            for (const auto& id_keyfrm : mkr->observations_) {
                const auto& keyfrm = id_keyfrm.second;
                if (!keyfrm || keyfrm->will_be_erased()) continue;
                const auto it = g.pose_of.find(keyfrm->id_);
                if (it == g.pose_of.end()) continue;
                const auto& undist_pt = keyfrm->markers_2d_.at(mkr->id_).undist_corners_.at(corner_idx);
                g.obs_pose.push_back(it->second);
                g.obs_point.push_back(first + (int)corner_idx);
                g.obs_uvr.insert(g.obs_uvr.end(), {undist_pt.x, undist_pt.y, -1.0f});
                g.obs_w.push_back(1.0f);
                g.obs_huber.push_back(0.0f);  // use_huber_kernel = false for marker edges (:172-174)
            }
        }
    }
}

int run(flat_graph& g, const unsigned int num_iter, const double gain_threshold, bool* const force_stop_flag, svgpu_ba_stats& stats) {
    svgpu_ba_problem pr;
    std::memset(&pr, 0, sizeof(pr));
    pr.num_poses = (int)g.pose_fixed.size(), pr.num_points = (int)g.point_fixed.size(), pr.num_obs = (int)g.obs_pose.size();
    pr.pose_cw = g.pose_cw.data(), pr.pose_fixed = g.pose_fixed.data(), pr.points = g.points.data(), pr.point_fixed = g.point_fixed.data();
    pr.obs_pose = g.obs_pose.data(), pr.obs_point = g.obs_point.data(), pr.obs_uvr = g.obs_uvr.data(), pr.obs_inv_sigma_sq = g.obs_w.data();
    pr.obs_huber_delta = g.obs_huber.data(), pr.intrinsics = g.intr.data();
    pr.num_first_iter = (int)num_iter, pr.num_second_iter = 0;
    pr.gain_threshold = gain_threshold;
    g.pose_out.assign(g.pose_cw.size(), 0.0);
    g.points_out.assign(g.points.size() + 3, 0.0);
    static_assert(sizeof(bool) == 1, "force_stop_flag is polled as one byte");
    return svgpu_global_ba(hip::context(), &pr, reinterpret_cast<volatile uint8_t*>(force_stop_flag), g.pose_out.data(), g.points_out.data(), &stats);
}

Mat44_t pose44(const flat_graph& g, const int slot) {
    Mat44_t T = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) T(i, j) = g.pose_out[(size_t)slot * 12 + 4 * i + j];
    return T;
}
Vec3_t point3(const flat_graph& g, const int slot) {
    Vec3_t p;
    for (int k = 0; k < 3; ++k) p(k) = g.points_out[(size_t)slot * 3 + k];
    return p;
}
}  // namespace

global_bundle_adjuster_hip::global_bundle_adjuster_hip(const unsigned int num_iter, const bool use_huber_kernel, const bool verbose)
    : num_iter_(num_iter), use_huber_kernel_(use_huber_kernel), verbose_(verbose) {}

void global_bundle_adjuster_hip::optimize_for_initialization(const std::vector<kf_ptr>& keyfrms, const std::vector<lm_ptr>& lms, const std::vector<mk_ptr>& markers,
                                                             float gain_threshold, bool fix_markers, bool* const force_stop_flag) const {
    std::vector<bool> is_optimized_lm(lms.size(), true);
    flat_graph g;
    gather(g, keyfrms, lms, markers, is_optimized_lm, use_huber_kernel_, fix_markers);
    last_status_ = run(g, num_iter_, gain_threshold, force_stop_flag, last_stats_);
    if (last_status_ != SVGPU_STOPPED) hip::check(last_status_, "svgpu_global_ba");
    if (force_stop_flag && *force_stop_flag) return;  // :226-228
    for (const auto& keyfrm : keyfrms) {  // :232-240
        if (keyfrm->will_be_erased()) continue;
        keyfrm->set_pose_cw(pose44(g, g.pose_of.at(keyfrm->id_)));
    }
    for (unsigned int i = 0; i < lms.size(); ++i) {  // :242-260
        if (!is_optimized_lm.at(i)) continue;
        const auto& lm = lms.at(i);
        if (!lm || lm->will_be_erased()) continue;
        lm->set_pos_in_world(point3(g, g.point_of_lm.at(i)));
        lm->update_mean_normal_and_obs_scale_variance();
    }
    for (const auto& mkr : markers) {  // :262-276
        if (fix_markers || mkr->keep_fixed_) continue;
        if (!mkr->initialized_before_) continue;
        const auto it = g.point_of_marker.find(mkr->id_);
        if (it == g.point_of_marker.end()) continue;
        for (int corner_idx = 0; corner_idx < 4; ++corner_idx) mkr->corners_pos_w_[corner_idx] = point3(g, it->second + corner_idx);
    }
}

bool global_bundle_adjuster_hip::optimize(const std::vector<kf_ptr>& keyfrms, std::unordered_set<unsigned int>& optimized_keyfrm_ids,
                                          std::unordered_set<unsigned int>& optimized_landmark_ids, std::unordered_set<unsigned int>& optimized_marker_ids,
                                          eigen_alloc_unord_map<unsigned int, Vec3_t>& lm_to_pos_w_after_global_BA,
                                          eigen_alloc_unord_map<unsigned int, Mat44_t>& keyfrm_to_pose_cw_after_global_BA,
                                          eigen_alloc_unord_map<unsigned int, std::array<Vec3_t, 4>>& marker_to_pos_w_after_global_BA, bool* const force_stop_flag) const {
    std::unordered_set<unsigned int> already_found_landmark_ids;  // :287-304: landmarks of the keyframes, first seen first
    std::vector<lm_ptr> lms;
    for (const auto& keyfrm : keyfrms)
        for (const auto& lm : keyfrm->get_landmarks()) {
            if (!lm || lm->will_be_erased()) continue;
            if (!already_found_landmark_ids.insert(lm->id_).second) continue;
            lms.push_back(lm);
        }
    std::unordered_set<unsigned int> already_found_marker_ids;  // :306-320
    std::vector<mk_ptr> markers;
    for (const auto& keyfrm : keyfrms)
        for (const auto& mkr : keyfrm->get_markers()) {
            if (!mkr) continue;
            if (!already_found_marker_ids.insert(mkr->id_).second) continue;
            markers.push_back(mkr);
        }
    std::vector<bool> is_optimized_lm(lms.size(), true);
    flat_graph g;
    gather(g, keyfrms, lms, markers, is_optimized_lm, use_huber_kernel_, false);
    last_status_ = run(g, num_iter_, 1e-3, force_stop_flag, last_stats_);
    if (last_status_ != SVGPU_STOPPED) hip::check(last_status_, "svgpu_global_ba");
    // :341-343: a stop raised by the caller discards the result, one raised by the gain rule (written through the same flag) keeps it
    if (force_stop_flag && *force_stop_flag && !last_stats_.stopped_by_terminate_action) return false;

    for (const auto& keyfrm : keyfrms) {  // :349-358
        if (keyfrm->will_be_erased()) continue;
        keyfrm_to_pose_cw_after_global_BA[keyfrm->id_] = pose44(g, g.pose_of.at(keyfrm->id_));
        optimized_keyfrm_ids.insert(keyfrm->id_);
    }
    for (unsigned int i = 0; i < lms.size(); ++i) {  // :360-378
        if (!is_optimized_lm.at(i)) continue;
        const auto& lm = lms.at(i);
        if (!lm || lm->will_be_erased()) continue;
        lm_to_pos_w_after_global_BA[lm->id_] = point3(g, g.point_of_lm.at(i));
        optimized_landmark_ids.insert(lm->id_);
    }
    for (const auto& mkr : markers) {  // :380-408
        if (mkr->keep_fixed_) continue;
        if (!mkr->initialized_before_) continue;
        const auto it = g.point_of_marker.find(mkr->id_);
        if (it == g.point_of_marker.end()) continue;
        bool changed = false;
        std::array<Vec3_t, 4> new_pos_corners;
        for (int corner_idx = 0; corner_idx < 4; ++corner_idx) {
            const Vec3_t orig_pos = mkr->corners_pos_w_[corner_idx];
            const Vec3_t new_pos = point3(g, it->second + corner_idx);
            for (int k = 0; k < 3; ++k)
                if (orig_pos(k) != new_pos(k)) changed = true;
            new_pos_corners[corner_idx] = new_pos;
        }
        if (!changed) continue;
        optimized_marker_ids.insert(mkr->id_);
        marker_to_pos_w_after_global_BA[mkr->id_] = new_pos_corners;
    }
    return true;
}

}  // namespace optimize
}  // namespace stella_vslam
