// Flat-problem adaptors for stella_vslam::optimize::{local_bundle_adjuster, global_bundle_adjuster, pose_optimizer} over the C ABI:
// optimize_flat() is steps 2-6 of local_bundle_adjuster_g2o::optimize (optimize/local_bundle_adjuster_g2o.cc:149-348) on a problem the
// caller has already gathered into arrays.  The class that DERIVES from optimize::local_bundle_adjuster and does the gather
// (:38-147) and the write-back (:352-430) itself -- the `backend: "hip"` sibling of local_bundle_adjuster_g2o / _gtsam -- is
// stella_vslam::optimize::local_bundle_adjuster_hip in host/drop_in/hip_backend.h.
#pragma once
#include <cstdint>
#include <vector>

#include "svgpu.h"

namespace stella_vslam_hip {
namespace optimize {

struct flat_ba_problem {
    std::vector<double> pose_cw;          // P x 12 rows of [R|t]
    std::vector<uint8_t> pose_fixed;      // P
    std::vector<double> points;           // L x 3
    std::vector<uint8_t> point_fixed;     // empty or L
    std::vector<int32_t> obs_pose, obs_point;  // E
    std::vector<float> obs_uvr;           // E x 3 (u, v, x_right or -1)
    std::vector<float> obs_inv_sigma_sq;  // E
    std::vector<float> obs_huber_delta;   // E (sqrt(5.99146) mono keyframes, sqrt(7.81473) otherwise; 0 = no kernel)
    std::vector<double> intrinsics;       // P x 5 (fx fy cx cy focal_x_baseline)
};

struct flat_ba_result {
    std::vector<double> pose_cw, points;
    std::vector<uint8_t> outlier;  // per observation
    svgpu_ba_stats stats;
    int status;
};

class local_bundle_adjuster_hip {
public:
    explicit local_bundle_adjuster_hip(svgpu_ctx* ctx, unsigned int num_first_iter = 5, unsigned int num_second_iter = 10)
        : ctx_(ctx), num_first_iter_(num_first_iter), num_second_iter_(num_second_iter) {}
    virtual ~local_bundle_adjuster_hip() = default;

    //! steps 2-6 of local_bundle_adjuster_g2o::optimize on the flattened problem; force_stop_flag as in the reference
    void optimize_flat(const flat_ba_problem& problem, bool* const force_stop_flag, flat_ba_result& result) const;

private:
    svgpu_ctx* ctx_;
    const unsigned int num_first_iter_;
    const unsigned int num_second_iter_;
};

//! optimize::global_bundle_adjuster sibling (optimize/global_bundle_adjuster.h:23-55; constructor defaults num_iter = 10,
//! use_huber_kernel = true).  optimize_flat() is global_bundle_adjuster::optimize (global_bundle_adjuster.cc:279-412) between its
//! gather (all landmarks of `keyfrms`, root keyframe fixed: the caller fills pose_fixed / huber deltas, or zeroes the deltas when
//! use_huber_kernel is false) and its write-back maps: ONE LM run of num_iter iterations with terminate_action(1e-3), no outlier
//! stage.  Returns false exactly where the reference does: the caller's force_stop_flag is up and it was not the gain rule that
//! raised it (:341-343) -- the result must then be discarded.
class global_bundle_adjuster_hip {
public:
    explicit global_bundle_adjuster_hip(svgpu_ctx* ctx, unsigned int num_iter = 10, bool use_huber_kernel = true)
        : ctx_(ctx), num_iter_(num_iter), use_huber_kernel_(use_huber_kernel) {}
    virtual ~global_bundle_adjuster_hip() = default;
    bool optimize_flat(const flat_ba_problem& problem, bool* const force_stop_flag, flat_ba_result& result) const;

private:
    svgpu_ctx* ctx_;
    const unsigned int num_iter_;
    const bool use_huber_kernel_;
};

//! optimize::pose_optimizer sibling (optimize/pose_optimizer.h, g2o defaults of pose_optimizer_factory.h:18-26).  The binding gathers,
//! per keypoint with a live landmark, the landmark position, the undistorted keypoint (+ stereo x_right), inv_level_sigma_sq
//! and the Huber delta (pose_optimizer_g2o.cc:86-106) and scatters outlier flags back by keypoint index.
class pose_optimizer_hip {
public:
    explicit pose_optimizer_hip(svgpu_ctx* ctx, unsigned int num_trials_robust = 2, unsigned int num_trials = 2, unsigned int num_each_iter = 10)
        : ctx_(ctx), num_trials_robust_(num_trials_robust), num_trials_(num_trials), num_each_iter_(num_each_iter) {}
    virtual ~pose_optimizer_hip() = default;
    //! returns the number of valid observations; optimized_pose_cw 3x4 row-major
    unsigned int optimize_flat(const double* cam_pose_cw, const std::vector<double>& pos_w, const std::vector<float>& obs_uvr,
                               const std::vector<float>& inv_sigma_sq, const std::vector<float>& huber_delta, const double* intrinsics,
                               double* optimized_pose_cw, std::vector<uint8_t>& outlier_flags) const;

private:
    svgpu_ctx* ctx_;
    const unsigned int num_trials_robust_, num_trials_, num_each_iter_;
};

}  // namespace optimize
}  // namespace stella_vslam_hip
