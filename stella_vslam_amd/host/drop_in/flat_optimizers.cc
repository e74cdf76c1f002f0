#include "flat_optimizers.h"

#include <stdexcept>
#include <string>

namespace stella_vslam_hip {
namespace optimize {

void local_bundle_adjuster_hip::optimize_flat(const flat_ba_problem& p, bool* const force_stop_flag, flat_ba_result& r) const {
    static_assert(sizeof(bool) == 1, "force_stop_flag is read and written as one byte");
    svgpu_ba_problem q{};
    q.num_poses = (int32_t)p.pose_fixed.size();
    q.num_points = (int32_t)(p.points.size() / 3);
    q.num_obs = (int32_t)p.obs_pose.size();
    q.pose_cw = p.pose_cw.data();
    q.pose_fixed = p.pose_fixed.data();
    q.points = p.points.data();
    q.point_fixed = p.point_fixed.empty() ? nullptr : p.point_fixed.data();
    q.obs_pose = p.obs_pose.data();
    q.obs_point = p.obs_point.data();
    q.obs_uvr = p.obs_uvr.data();
    q.obs_inv_sigma_sq = p.obs_inv_sigma_sq.data();
    q.obs_huber_delta = p.obs_huber_delta.empty() ? nullptr : p.obs_huber_delta.data();
    q.intrinsics = p.intrinsics.data();
    q.num_first_iter = (int32_t)num_first_iter_;
    q.num_second_iter = (int32_t)num_second_iter_;
    q.gain_threshold = 1e-3;  // terminateAction->setGainThreshold(1e-3), local_bundle_adjuster_g2o.cc:158
    r.pose_cw.assign(p.pose_cw.size(), 0.0);
    r.points.assign(p.points.size(), 0.0);
    r.outlier.assign(p.obs_pose.size() ? p.obs_pose.size() : 1, 0);
    r.status = svgpu_local_ba(ctx_, &q, reinterpret_cast<volatile uint8_t*>(force_stop_flag), r.pose_cw.data(), r.points.data(),
                              r.outlier.data(), &r.stats);
    r.outlier.resize(p.obs_pose.size());
    // the reference's optimize() returns void and fails silently (:308-310); a HIP error is NOT silent here
    if (r.status != SVGPU_OK && r.status != SVGPU_STOPPED)
        throw std::runtime_error(std::string("svgpu_local_ba: ") + svgpu_status_string(r.status) + " (" + svgpu_last_error(ctx_) + ")");
}

bool global_bundle_adjuster_hip::optimize_flat(const flat_ba_problem& p, bool* const force_stop_flag, flat_ba_result& r) const {
    svgpu_ba_problem q{};
    q.num_poses = (int32_t)p.pose_fixed.size();
    q.num_points = (int32_t)(p.points.size() / 3);
    q.num_obs = (int32_t)p.obs_pose.size();
    q.pose_cw = p.pose_cw.data();
    q.pose_fixed = p.pose_fixed.data();
    q.points = p.points.data();
    q.point_fixed = p.point_fixed.empty() ? nullptr : p.point_fixed.data();
    q.obs_pose = p.obs_pose.data();
    q.obs_point = p.obs_point.data();
    q.obs_uvr = p.obs_uvr.data();
    q.obs_inv_sigma_sq = p.obs_inv_sigma_sq.data();
    q.obs_huber_delta = (use_huber_kernel_ && !p.obs_huber_delta.empty()) ? p.obs_huber_delta.data() : nullptr;  // :93-101
    q.intrinsics = p.intrinsics.data();
    q.num_first_iter = (int32_t)num_iter_;
    q.num_second_iter = 0;
    q.gain_threshold = 1e-3;  // global_bundle_adjuster.cc:335
    r.pose_cw.assign(p.pose_cw.size(), 0.0);
    r.points.assign(p.points.size(), 0.0);
    r.outlier.assign(p.obs_pose.size(), 0);  // global BA has no outlier stage
    r.status = svgpu_global_ba(ctx_, &q, reinterpret_cast<volatile uint8_t*>(force_stop_flag), r.pose_cw.data(), r.points.data(), &r.stats);
    if (r.status != SVGPU_OK && r.status != SVGPU_STOPPED)
        throw std::runtime_error(std::string("svgpu_global_ba: ") + svgpu_status_string(r.status) + " (" + svgpu_last_error(ctx_) + ")");
    if (force_stop_flag && *force_stop_flag && !r.stats.stopped_by_terminate_action) return false;  // :341-343
    return true;
}

unsigned int pose_optimizer_hip::optimize_flat(const double* cam_pose_cw, const std::vector<double>& pos_w, const std::vector<float>& obs_uvr,
                                               const std::vector<float>& inv_sigma_sq, const std::vector<float>& huber_delta,
                                               const double* intrinsics, double* optimized_pose_cw, std::vector<uint8_t>& outlier_flags) const {
    const int n = (int)inv_sigma_sq.size();
    outlier_flags.assign((size_t)(n > 0 ? n : 1), 0);
    int num_valid = 0;
    const int rc = svgpu_pose_optimize(ctx_, cam_pose_cw, n, pos_w.data(), obs_uvr.data(), inv_sigma_sq.data(), huber_delta.data(), intrinsics,
                                       (int)num_trials_robust_, (int)num_trials_, (int)num_each_iter_, 0, optimized_pose_cw,
                                       outlier_flags.data(), &num_valid, nullptr);
    outlier_flags.resize((size_t)n);
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_pose_optimize: ") + svgpu_status_string(rc) + " (" + svgpu_last_error(ctx_) + ")");
    return (unsigned int)num_valid;
}

}  // namespace optimize
}  // namespace stella_vslam_hip
