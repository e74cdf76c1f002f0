// The reference's data/landmark.cc (compiled where it lies over stand-in keyframe / map_database headers, with the reference's REAL
// data/landmark.h) behind a C export on CSR observation lists: landmark::compute_descriptor (data/landmark.cc:199-254) and
// landmark::update_mean_normal_and_obs_scale_variance (:256-318).  Third library, oracle/_ref/libsvref_lm.so.  Test infrastructure only.
#include <cstring>
#include <memory>
#include <vector>

#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"

using namespace stella_vslam;

extern "C" {
// Every observation gets its own one-keypoint keyframe (id = observation index, so the id-ordered observation map iterates in CSR
// order); ref_in_row[l] = which observation of landmark l is its reference keyframe.
void svref_landmarks_refresh(int n, const int32_t* obs_off, const uint8_t* obs_desc, const double* obs_trans_wc, const int32_t* obs_octave,
                             const int32_t* ref_in_row, const double* pos_w, float scale_factor, unsigned num_levels, uint8_t* descriptor,
                             double* mean_normal, float* max_valid_dist, float* min_valid_dist) {
    const feature::orb_params params("ref", scale_factor, num_levels, 20, 7);
    for (int l = 0; l < n; ++l) {
        std::vector<std::shared_ptr<data::keyframe>> kfs;
        for (int o = obs_off[l]; o < obs_off[l + 1]; ++o) {
            auto kf = std::make_shared<data::keyframe>((unsigned)o, &params);
            kf->frm_obs_.descriptors_ = cv::Mat(1, 32, CV_8UC1, const_cast<uint8_t*>(obs_desc + 32 * (size_t)o), 32);
            kf->frm_obs_.undist_keypts_.resize(1);
            kf->frm_obs_.undist_keypts_[0].octave = obs_octave[o];
            kf->trans_wc_ = Vec3_t(obs_trans_wc[3 * o], obs_trans_wc[3 * o + 1], obs_trans_wc[3 * o + 2]);
            kfs.push_back(kf);
        }
        auto lm = std::make_shared<data::landmark>((unsigned)l, Vec3_t(pos_w[3 * l], pos_w[3 * l + 1], pos_w[3 * l + 2]), kfs.at(ref_in_row[l]));
        for (const auto& kf : kfs) lm->add_observation(kf, 0);
        lm->compute_descriptor();
        lm->update_mean_normal_and_obs_scale_variance();
        const cv::Mat d = lm->get_descriptor();
        memcpy(descriptor + 32 * (size_t)l, d.ptr(0), 32);
        const Vec3_t m = lm->get_obs_mean_normal();
        mean_normal[3 * l] = m(0), mean_normal[3 * l + 1] = m(1), mean_normal[3 * l + 2] = m(2);
        max_valid_dist[l] = lm->get_max_valid_distance();
        min_valid_dist[l] = lm->get_min_valid_distance();
    }
}

// landmark::is_inside_in_orb_scale (data/landmark.h:88-92) and landmark::predict_scale_level (data/landmark.cc:336-353), the two landmark members
// frame::can_observe (data/frame.cc:59-85) calls, for landmark l observed once from `ref_trans_wc` at level `ref_octave` and asked about
// `cam_to_lm_dist`.  Also returns the valid-distance range the landmark derived.
void svref_landmark_scale_queries(int n, const double* pos_w, const double* ref_trans_wc, const int32_t* ref_octave, const double* cam_to_lm_dist,
                                  float scale_factor, unsigned num_levels, uint8_t* inside, int32_t* level, float* max_valid_dist, float* min_valid_dist) {
    const feature::orb_params params("ref", scale_factor, num_levels, 20, 7);
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
        const auto margin_far = 1.3;  // frame.cc:71-72
        const auto margin_near = 1.0 / margin_far;
        inside[l] = lm->is_inside_in_orb_scale(cam_to_lm_dist[l], margin_far, margin_near);
        level[l] = (int32_t)lm->predict_scale_level(cam_to_lm_dist[l], params.num_levels_, params.log_scale_factor_);
        max_valid_dist[l] = lm->get_max_valid_distance();
        min_valid_dist[l] = lm->get_min_valid_distance();
    }
}
}
