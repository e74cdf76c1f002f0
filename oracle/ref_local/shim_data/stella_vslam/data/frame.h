// Stand-in (see ../../README.md) for data/frame.h: plain data + the accessors the matcher sources call.
#ifndef SVGPU_SHIM_STELLA_DATA_FRAME_H
#define SVGPU_SHIM_STELLA_DATA_FRAME_H
#include <memory>
#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/bow_vocabulary.h"
#include "stella_vslam/data/common.h"
#include "stella_vslam/data/frame_observation.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"
namespace stella_vslam {
namespace data {
class frame {
public:
    frame(unsigned int id, camera::base* camera, const feature::orb_params* orb_params) : id_(id), camera_(camera), orb_params_(orb_params) {}
    void set_pose_cw(const Mat44_t& pose_cw) {
        pose_cw_ = pose_cw;
        pose_is_valid_ = true;
    }
    Mat44_t get_pose_cw() const { return pose_cw_; }
    Mat33_t get_rot_cw() const { return pose_cw_.block<3, 3>(0, 0); }
    Vec3_t get_trans_cw() const { return pose_cw_.block<3, 1>(0, 3); }
    Vec3_t get_trans_wc() const { return -(get_rot_cw().transpose()) * get_trans_cw(); }
    bool pose_is_valid() const { return pose_is_valid_; }
    void add_landmark(const std::shared_ptr<landmark>& lm, const unsigned int idx) { landmarks_.at(idx) = lm; }
    std::shared_ptr<landmark> get_landmark(const unsigned int idx) const { return landmarks_.at(idx); }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    bool has_landmark(const std::shared_ptr<landmark>& lm) const {
        for (const auto& l : landmarks_)
            if (l == lm) return true;
        return false;
    }
    void set_landmarks(const std::vector<std::shared_ptr<landmark>>& lms) { landmarks_ = lms; }
    void erase_landmarks() { std::fill(landmarks_.begin(), landmarks_.end(), nullptr); }
    std::vector<unsigned int> get_keypoints_in_cell(const float ref_x, const float ref_y, const float margin, const int min_level = -1,
                                                    const int max_level = -1) const {
        return data::get_keypoints_in_cell(camera_, frm_obs_, ref_x, ref_y, margin, min_level, max_level);
    }
    unsigned int id_;
    camera::base* camera_;
    const feature::orb_params* orb_params_;
    frame_observation frm_obs_;
    bow_vector bow_vec_;
    bow_feature_vector bow_feat_vec_;
    std::vector<std::shared_ptr<landmark>> landmarks_;

private:
    Mat44_t pose_cw_ = Mat44_t::Identity();
    bool pose_is_valid_ = false;
};
}  // namespace data
}  // namespace stella_vslam
#endif
