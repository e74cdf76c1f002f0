// optimize::pose_optimizer_hip (pose_optimizer_hip.h): gather -> svgpu_pose_optimize -> pose and flags back.
#include "drop_in/pose_optimizer_hip.h"

#include <cmath>

namespace stella_vslam {
namespace optimize {

pose_optimizer_hip::pose_optimizer_hip(const unsigned int num_trials_robust, const unsigned int num_trials, const unsigned int num_each_iter)
    : num_trials_robust_(num_trials_robust), num_trials_(num_trials), num_each_iter_(num_each_iter) {}

unsigned int pose_optimizer_hip::optimize(const data::frame& frm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const {
    return optimize(frm.get_pose_cw(), frm.frm_obs_, frm.orb_params_, frm.camera_, frm.get_landmarks(), optimized_pose, outlier_flags);  // pose_optimizer_g2o.cc:26-30
}

unsigned int pose_optimizer_hip::optimize(const data::keyframe* keyfrm, Mat44_t& optimized_pose, std::vector<bool>& outlier_flags) const {
    return optimize(keyfrm->get_pose_cw(), keyfrm->frm_obs_, keyfrm->orb_params_, keyfrm->camera_, keyfrm->get_landmarks(), optimized_pose, outlier_flags);  // :32-36
}

unsigned int pose_optimizer_hip::optimize(const Mat44_t& cam_pose_cw, const data::frame_observation& frm_obs, const feature::orb_params* orb_params,
                                          const camera::base* camera, const std::vector<std::shared_ptr<data::landmark>>& landmarks, Mat44_t& optimized_pose,
                                          std::vector<bool>& outlier_flags) const {
    const unsigned int num_keypts = frm_obs.undist_keypts_.size();
    outlier_flags.resize(num_keypts);  // :68-70
    std::fill(outlier_flags.begin(), outlier_flags.end(), false);
    constexpr float chi_sq_2D = 5.99146;
    const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D);
    constexpr float chi_sq_3D = 7.81473;
    const float sqrt_chi_sq_3D = std::sqrt(chi_sq_3D);
    const float sqrt_chi_sq = camera->setup_type_ == camera::setup_type_t::Monocular ? sqrt_chi_sq_2D : sqrt_chi_sq_3D;  // :103-105

    std::vector<unsigned int> kp_of_obs;
    std::vector<double> pos_w;
    std::vector<float> uvr, inv_sigma_sq, huber;
    for (unsigned int idx = 0; idx < num_keypts; ++idx) {  // :86-109
        const auto& lm = landmarks.at(idx);
        if (!lm || lm->will_be_erased()) continue;
        const auto& undist_keypt = frm_obs.undist_keypts_.at(idx);
        const float x_right = frm_obs.stereo_x_right_.empty() ? -1.0f : frm_obs.stereo_x_right_.at(idx);
        const Vec3_t p = lm->get_pos_in_world();
        kp_of_obs.push_back(idx);
        for (int k = 0; k < 3; ++k) pos_w.push_back(p(k));
        uvr.push_back(undist_keypt.pt.x);
        uvr.push_back(undist_keypt.pt.y);
        uvr.push_back(x_right);
        inv_sigma_sq.push_back(orb_params->inv_level_sigma_sq_.at(undist_keypt.octave));
        huber.push_back(sqrt_chi_sq);
    }
    const int n = (int)kp_of_obs.size();
    last_lm_iterations_ = 0;
    if (n < 5) return 0;  // :111-113 (optimized_pose is left as the caller passed it)

    const svgpu_camera c = hip::to_svgpu_camera(camera);
    double K[5];  // the edge the wrapper builds per camera model (pose_opt_edge_wrapper.h:57-200): perspective, fisheye and radial division
                  // all take the perspective edge on the undistorted keypoints; equirectangular its own
    if (c.model == SVGPU_CAM_EQUIRECTANGULAR) K[0] = 0, K[1] = 0, K[2] = c.cols, K[3] = c.rows, K[4] = 0;
    else K[0] = c.fx, K[1] = c.fy, K[2] = c.cx, K[3] = c.cy, K[4] = c.focal_x_baseline;
    double pose_in[12], pose_out[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) pose_in[4 * i + j] = cam_pose_cw(i, j);
    std::vector<uint8_t> flags((size_t)n, 0);
    int num_valid = 0;
    hip::check(svgpu_pose_optimize(hip::context(), pose_in, n, pos_w.data(), uvr.data(), inv_sigma_sq.data(), huber.data(), K, (int)num_trials_robust_,
                                   (int)num_trials_, (int)num_each_iter_, reset_stop_flag_each_round_, pose_out, flags.data(), &num_valid, &last_lm_iterations_),
               "svgpu_pose_optimize");
    for (int k = 0; k < n; ++k) outlier_flags.at(kp_of_obs[k]) = flags[k] != 0;  // :130-160: flags by keypoint index
    optimized_pose = Mat44_t::Identity();  // :171 (to_eigen_mat of the vertex estimate)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) optimized_pose(i, j) = pose_out[4 * i + j];
    return (unsigned int)num_valid;
}

namespace hip_backend {
std::unique_ptr<pose_optimizer> create_pose_optimizer(const std::string& backend, const unsigned int num_trials_robust, const unsigned int num_trials,
                                                      const unsigned int num_each_iter) {
    if (backend == "hip") return std::unique_ptr<pose_optimizer>(new pose_optimizer_hip(num_trials_robust, num_trials, num_each_iter));
    return nullptr;
}
}  // namespace hip_backend

}  // namespace optimize
}  // namespace stella_vslam
