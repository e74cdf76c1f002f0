// data/frame.cc compiled from the reference with its REAL data/frame.h and data/landmark.h (over the keyframe / map-database stand-ins of
// shim_lm and the BoW stand-in of shim_frm): frame::set_pose_cw and frame::can_observe -- the frustum test behind
// projection::match_frame_and_landmarks (tracking_module / local-map search).  Separate library (oracle/_ref/libsvref_frm.so).
// Test infrastructure only (tests/test_ref_local_camera.py).
#include <cstring>
#include <memory>
#include <unordered_map>

#include "stella_vslam/camera/equirectangular.h"
#include "stella_vslam/camera/fisheye.h"
#include "stella_vslam/camera/perspective.h"
#include "stella_vslam/camera/radial_division.h"
#include "stella_vslam/data/frame.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"

using namespace stella_vslam;

namespace {
std::unique_ptr<camera::base> make(int model, int stereo, unsigned cols, unsigned rows, const double* k) {
    const auto setup = stereo ? camera::setup_type_t::Stereo : camera::setup_type_t::Monocular;
    const auto col = camera::color_order_t::Gray;
    switch (model) {
        case 0: return std::unique_ptr<camera::base>(new camera::perspective("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, 0, k[4]));
        case 1: return std::unique_ptr<camera::base>(new camera::fisheye("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, k[4]));
        case 2: return std::unique_ptr<camera::base>(new camera::equirectangular("ref", col, cols, rows, 30.0));
        default: return std::unique_ptr<camera::base>(new camera::radial_division("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, k[4]));
    }
}
}  // namespace

extern "C" {
// n landmarks, each observed once from a reference keyframe (position, octave of the observing keypoint) so that the reference's own
// landmark::update_mean_normal_and_obs_scale_variance gives it a mean normal and a valid-distance range; then frame::can_observe at
// the frame pose.  The landmark state is returned as well: it is what the oracle's can_observe takes as input.
void svref_frame_can_observe(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, const double* pose_cw12, int n,
                             const double* pos_w, const double* ref_trans_wc, const int32_t* ref_octave, float ray_cos_thr, float scale_factor,
                             unsigned num_levels, uint8_t* visible, double* reproj, float* x_right, int32_t* level, double* mean_normal,
                             float* min_valid_dist, float* max_valid_dist, double* trans_wc_out) {
    auto cam = make(model, stereo_cam, cols, rows, intr5);
    feature::orb_params params("ref", scale_factor, num_levels, 20, 7);
    data::frame_observation obs;
    data::frame frm(0, 0.0, cam.get(), &params, obs, std::unordered_map<unsigned int, data::marker2d>());
    Mat44_t T = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) T(i, j) = pose_cw12[4 * i + j];
    frm.set_pose_cw(T);
    const Vec3_t twc = frm.get_trans_wc();
    for (int c = 0; c < 3; ++c) trans_wc_out[c] = twc(c);
    uint8_t zeros[32] = {0};
    for (int l = 0; l < n; ++l) {
        auto kf = std::make_shared<data::keyframe>((unsigned)l, &params);
        kf->frm_obs_.descriptors_ = cv::Mat(1, 32, CV_8UC1, zeros, 32);
        kf->frm_obs_.undist_keypts_.resize(1);
        kf->frm_obs_.undist_keypts_[0].octave = ref_octave[l];
        kf->trans_wc_ = Vec3_t(ref_trans_wc[3 * l], ref_trans_wc[3 * l + 1], ref_trans_wc[3 * l + 2]);
        auto lm = std::make_shared<data::landmark>((unsigned)l, Vec3_t(pos_w[3 * l], pos_w[3 * l + 1], pos_w[3 * l + 2]), kf);
        lm->add_observation(kf, 0);
        lm->update_mean_normal_and_obs_scale_variance();
        const Vec3_t nrm = lm->get_obs_mean_normal();
        for (int c = 0; c < 3; ++c) mean_normal[3 * l + c] = nrm(c);
        min_valid_dist[l] = lm->get_min_valid_distance();
        max_valid_dist[l] = lm->get_max_valid_distance();
        Vec2_t rp;
        float xr = 0.f;
        unsigned int lv = 0;
        visible[l] = frm.can_observe(lm, ray_cos_thr, rp, xr, lv) ? 1 : 0;
        reproj[2 * l] = rp(0), reproj[2 * l + 1] = rp(1);
        x_right[l] = xr;
        level[l] = (int32_t)lv;
    }
}
}
