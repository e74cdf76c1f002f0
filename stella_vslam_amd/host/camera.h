// C++ adaptors for stella_vslam::camera::* (reference: src/stella_vslam/camera/base.h:56-201 and the four models) over the
// C ABI: the batched members the tracking front end calls.  Constructor arguments are the reference's model parameters;
// img_bounds_ is filled in the constructor (compute_image_bounds) as the reference does.
#pragma once
#include <array>
#include <map>
#include <string>
#include <vector>

#include "orb_extractor.h"

namespace stella_vslam_hip {

using Vec3_t = std::array<double, 3>;  // stands for Eigen::Vector3d (type.h:50), same memory layout
using Vec2_t = std::array<double, 2>;
using Mat33_t = std::array<double, 9>;  // ROW-major here (Eigen's Matrix3d is column-major: pass rot.transpose().data())

namespace camera {

enum class model_type_t { Perspective = 0, Fisheye = 1, Equirectangular = 2, RadialDivision = 3 };  // camera/base.h:24-29

struct image_bounds {  // camera/base.h:46-54
    float min_x_, max_x_, min_y_, max_y_;
};

//! flat landmark view read by data::frame::can_observe (data/frame.cc:59-85): get_pos_in_world(), get_obs_mean_normal(),
//! get_min_valid_distance(), get_max_valid_distance(); `skip` = not offered (already tracked / will_be_erased)
struct landmark_set {
    std::vector<Vec3_t> pos_w, mean_normal;
    std::vector<float> min_valid_dist, max_valid_dist;
    std::vector<unsigned char> skip;  // empty or n
    cv::Mat descriptors;              // n x 32 (landmark::get_descriptor), only read by the matcher
};

struct observability {  // the outputs of can_observe, per landmark
    std::vector<unsigned char> visible;
    std::vector<Vec2_t> reproj;
    std::vector<float> x_right;
    std::vector<int> pred_scale_level;
};

class base {
public:
    base(svgpu_ctx* ctx, model_type_t model, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy,
         const std::vector<double>& dist, double focal_x_baseline);
    virtual ~base() = default;

    //! camera::*::compute_image_bounds
    image_bounds compute_image_bounds() const;
    //! camera::*::undistort_keypoints (base.cc:139-148 and the overrides)
    void undistort_keypoints(const std::vector<cv::KeyPoint>& dist_keypts, std::vector<cv::KeyPoint>& undist_keypts) const;
    //! camera::base::convert_keypoints_to_bearings (base.cc:160-164)
    void convert_keypoints_to_bearings(const std::vector<cv::KeyPoint>& undist_keypts, std::vector<Vec3_t>& bearings) const;
    //! system.cc:384-395 in one device call: undistort + bearings + data::assign_keypoints_to_grid (CSR, cell = col * rows + row)
    void observe(const std::vector<cv::KeyPoint>& dist_keypts, unsigned int num_grid_cols, unsigned int num_grid_rows,
                 std::vector<cv::KeyPoint>& undist_keypts, std::vector<Vec3_t>& bearings, std::vector<int>& cell_off,
                 std::vector<int>& cell_items) const;
    //! data::frame::can_observe for every landmark of `lms` (tracking_module.cc:554-594)
    void can_observe(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& trans_wc, const landmark_set& lms, float ray_cos_thr,
                     unsigned int num_levels, float log_scale_factor, observability& out) const;

    const svgpu_camera& c_abi() const { return c_; }
    svgpu_ctx* context() const { return ctx_; }

    const model_type_t model_type_;
    const unsigned int cols_, rows_;
    const double focal_x_baseline_;
    image_bounds img_bounds_;

protected:
    svgpu_ctx* ctx_;
    svgpu_camera c_;
};

class perspective final : public base {  // camera/perspective.h
public:
    perspective(svgpu_ctx* ctx, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double k1, double k2,
                double p1, double p2, double k3, double focal_x_baseline = 0.0)
        : base(ctx, model_type_t::Perspective, cols, rows, fx, fy, cx, cy, {k1, k2, p1, p2, k3}, focal_x_baseline) {}
};
class fisheye final : public base {  // camera/fisheye.h
public:
    fisheye(svgpu_ctx* ctx, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double k1, double k2, double k3,
            double k4, double focal_x_baseline = 0.0)
        : base(ctx, model_type_t::Fisheye, cols, rows, fx, fy, cx, cy, {k1, k2, k3, k4}, focal_x_baseline) {}
};
class equirectangular final : public base {  // camera/equirectangular.h
public:
    equirectangular(svgpu_ctx* ctx, unsigned int cols, unsigned int rows) : base(ctx, model_type_t::Equirectangular, cols, rows, 0, 0, 0, 0, {}, 0.0) {}
};
class radial_division final : public base {  // camera/radial_division.h
public:
    radial_division(svgpu_ctx* ctx, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double distortion,
                    double focal_x_baseline = 0.0)
        : base(ctx, model_type_t::RadialDivision, cols, rows, fx, fy, cx, cy, {distortion}, focal_x_baseline) {}
};

}  // namespace camera

namespace data {
//! data::landmark::compute_descriptor (data/landmark.cc:199-254) for many landmarks at once.  obs_off: CSR offsets over the
//! landmarks; obs_desc: one descriptor row per observation, in the iteration order of observations_ with will_be_erased
//! keyframes dropped.  best_obs[l] indexes landmark l's own list; descriptors row l = its new descriptor_.
void compute_descriptors(svgpu_ctx* ctx, const std::vector<int>& obs_off, const cv::Mat& obs_desc, std::vector<int>& best_obs, cv::Mat& descriptors);
//! data::landmark::update_mean_normal_and_obs_scale_variance (data/landmark.cc:285-318) for many landmarks at once.
//! obs_trans_wc: keyfrm->get_trans_wc() per observation; ref_trans_wc / ref_scale_factor: the reference keyframe's centre and
//! scale_factors_[octave of the landmark's keypoint there]; inv_scale_factor_last = inv_scale_factors_[num_levels_ - 1].
void update_mean_normal_and_obs_scale_variance(svgpu_ctx* ctx, const std::vector<int>& obs_off, const std::vector<Vec3_t>& obs_trans_wc,
                                               const std::vector<Vec3_t>& pos_w, const std::vector<Vec3_t>& ref_trans_wc,
                                               const std::vector<float>& ref_scale_factor, float inv_scale_factor_last,
                                               std::vector<Vec3_t>& mean_normal, std::vector<float>& max_valid_dist,
                                               std::vector<float>& min_valid_dist);

//! data::bow_vocabulary (data/bow_vocabulary.h) on a flat tree kept resident on the device: node 0 = root, children of node i =
//! children[child_off[i] .. child_off[i+1]) (none = leaf), node_desc n x 32, weight / word id per node, `depth` = L.
//! compute_bow() is bow_vocabulary_util::compute_bow (data/bow_vocabulary.cc:18-24): the descent runs on the device, the two sparse
//! maps are assembled here.  `fbow_k` > 0 selects the reference's DEFAULT build -- fbow::Vocabulary::transform(descriptors, 4, ...): store level
//! counted from the root, bow_feat_vec keyed by FBoW's path code, every word summed, L2-normalised as upstream FBoW does -- with k = the
//! vocabulary's branching factor; 0 = the USE_DBOW2 build (levels up from the leaves, node-index keys, stop words skipped, L1).
class bow_vocabulary_hip {
public:
    bow_vocabulary_hip(svgpu_ctx* ctx, const std::vector<int>& child_off, const std::vector<int>& children, const cv::Mat& node_desc,
                       const std::vector<float>& node_weight, const std::vector<int>& word_id, int depth, int fbow_k = 0);
    ~bow_vocabulary_hip();
    bow_vocabulary_hip(const bow_vocabulary_hip&) = delete;
    bow_vocabulary_hip& operator=(const bow_vocabulary_hip&) = delete;
    void compute_bow(const cv::Mat& descriptors, std::map<unsigned int, double>& bow_vec,
                     std::map<unsigned int, std::vector<unsigned int>>& bow_feat_vec, int levels_up = 4) const;

private:
    svgpu_ctx* ctx_;
    svgpu_vocabulary* vocab_ = nullptr;
    int depth_, fbow_k_ = 0;
};
}  // namespace data
}  // namespace stella_vslam_hip
