// optimize::pose_optimizer_hip -- `backend: "hip"` of the PoseOptimizer YAML node: the sibling of pose_optimizer_g2o
// (optimize/pose_optimizer_g2o.h:25-60) behind the reference's abstract optimize::pose_optimizer (optimize/pose_optimizer.h:24-40), all
// three overloads.  The gather is steps 2-3 of pose_optimizer_g2o.cc:62-109 (keypoints that hold a live landmark; inverse sigma of the
// keypoint's octave; the Huber width by the camera's set-up; a monocular edge where x_right < 0), steps 4-5 are ONE svgpu_pose_optimize
// call (one kernel launch).  Compiles against the reference tree with -DSVGPU_WITH_STELLA_VSLAM, or against host/standin/stella_standin.h.
#pragma once
#ifdef SVGPU_WITH_STELLA_VSLAM
#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/frame.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/optimize/pose_optimizer.h"
#else
#include "standin/stella_standin.h"
#endif
#include <memory>
#include <string>
#include <vector>

#include "svgpu.h"

namespace stella_vslam {
namespace hip {  // (hip_backend.h / hip_backend.cc)
svgpu_ctx* context();
svgpu_camera to_svgpu_camera(const camera::base* camera);
void check(int status, const char* where);
}  // namespace hip

namespace optimize {

#ifndef SVGPU_WITH_STELLA_VSLAM
class pose_optimizer {  // optimize/pose_optimizer.h:24-40
public:
    virtual ~pose_optimizer() = default;
    virtual unsigned int optimize(const data::frame& frm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const = 0;
    virtual unsigned int optimize(const data::keyframe* keyfrm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const = 0;
    virtual unsigned int optimize(const Mat44_t& cam_pose_cw, const data::frame_observation& frm_obs, const feature::orb_params* orb_params,
                                  const camera::base* camera, const std::vector<std::shared_ptr<data::landmark>>& landmarks, Mat44_t& optimized_pose,
                                  std::vector<bool>& outlier_flags) const = 0;
};
#endif

class pose_optimizer_hip : public pose_optimizer {
public:
    explicit pose_optimizer_hip(unsigned int num_trials_robust = 2, unsigned int num_trials = 2, unsigned int num_each_iter = 10);
    virtual ~pose_optimizer_hip() = default;
    unsigned int optimize(const data::frame& frm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const override;
    unsigned int optimize(const data::keyframe* keyfrm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const override;
    unsigned int optimize(const Mat44_t& cam_pose_cw, const data::frame_observation& frm_obs, const feature::orb_params* orb_params, const camera::base* camera,
                          const std::vector<std::shared_ptr<data::landmark>>& landmarks, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const override;
    //! Levenberg-Marquardt iterations of the last call, all rounds (diagnostics; the reference logs nothing here)
    mutable int last_lm_iterations_ = 0;
    //! 0 = g2o's literal behaviour: the stop flag the gain rule raised keeps suppressing the later rounds (svgpu.h, svgpu_pose_optimize)
    int reset_stop_flag_each_round_ = 0;

private:
    const unsigned int num_trials_robust_, num_trials_, num_each_iter_;
};

namespace hip_backend {
//! What pose_optimizer_factory::create (optimize/pose_optimizer_factory.h:18-47) returns for a backend string "hip" with the keys
//! num_trials_robust / num_trials / num_each_iter read by the caller from the `hip` (or `g2o`) sub-node; nullptr for another backend.
std::unique_ptr<pose_optimizer> create_pose_optimizer(const std::string& backend, unsigned int num_trials_robust = 2, unsigned int num_trials = 2,
                                                      unsigned int num_each_iter = 10);
}  // namespace hip_backend

}  // namespace optimize
}  // namespace stella_vslam
