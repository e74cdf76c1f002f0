#include "camera.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace stella_vslam_hip {
namespace camera {

namespace {
void check(svgpu_ctx* ctx, int rc, const char* where) {
    if (rc != SVGPU_OK) throw std::runtime_error(std::string(where) + ": " + svgpu_last_error(ctx));
}
static_assert(sizeof(cv::KeyPoint) == sizeof(svgpu_keypoint), "KeyPoint layout");
static_assert(sizeof(Vec3_t) == 24 && sizeof(Vec2_t) == 16, "Eigen-compatible packing");
}  // namespace

base::base(svgpu_ctx* ctx, model_type_t model, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy,
           const std::vector<double>& dist, double focal_x_baseline)
    : model_type_(model), cols_(cols), rows_(rows), focal_x_baseline_(focal_x_baseline), ctx_(ctx), c_{} {
    c_.model = (int32_t)model;
    c_.cols = cols;
    c_.rows = rows;
    c_.fx = fx, c_.fy = fy, c_.cx = cx, c_.cy = cy;
    for (size_t i = 0; i < dist.size() && i < 5; ++i) c_.dist[i] = dist[i];
    c_.focal_x_baseline = focal_x_baseline;
    img_bounds_ = compute_image_bounds();
    c_.min_x = img_bounds_.min_x_, c_.max_x = img_bounds_.max_x_, c_.min_y = img_bounds_.min_y_, c_.max_y = img_bounds_.max_y_;
}

image_bounds base::compute_image_bounds() const {
    svgpu_camera c = c_;
    check(ctx_, svgpu_camera_image_bounds(ctx_, &c), "svgpu_camera_image_bounds");
    return image_bounds{c.min_x, c.max_x, c.min_y, c.max_y};
}

void base::undistort_keypoints(const std::vector<cv::KeyPoint>& dist_keypts, std::vector<cv::KeyPoint>& undist_keypts) const {
    undist_keypts.resize(dist_keypts.size());
    check(ctx_, svgpu_frame_observation(ctx_, &c_, reinterpret_cast<const svgpu_keypoint*>(dist_keypts.data()), (int)dist_keypts.size(), 1, 1,
                                        reinterpret_cast<svgpu_keypoint*>(undist_keypts.data()), nullptr, nullptr, nullptr),
          "svgpu_frame_observation");
}

void base::convert_keypoints_to_bearings(const std::vector<cv::KeyPoint>& undist_keypts, std::vector<Vec3_t>& bearings) const {
    bearings.resize(undist_keypts.size());
    check(ctx_, svgpu_keypoints_to_bearings(ctx_, &c_, reinterpret_cast<const svgpu_keypoint*>(undist_keypts.data()), (int)undist_keypts.size(),
                                            reinterpret_cast<double*>(bearings.data())),
          "svgpu_keypoints_to_bearings");
}

void base::observe(const std::vector<cv::KeyPoint>& dist_keypts, unsigned int num_grid_cols, unsigned int num_grid_rows,
                   std::vector<cv::KeyPoint>& undist_keypts, std::vector<Vec3_t>& bearings, std::vector<int>& cell_off,
                   std::vector<int>& cell_items) const {
    const size_t n = dist_keypts.size();
    undist_keypts.resize(n);
    bearings.resize(n);
    cell_off.assign((size_t)num_grid_cols * num_grid_rows + 1, 0);
    cell_items.assign(n ? n : 1, 0);
    check(ctx_, svgpu_frame_observation(ctx_, &c_, reinterpret_cast<const svgpu_keypoint*>(dist_keypts.data()), (int)n, (int)num_grid_cols,
                                        (int)num_grid_rows, reinterpret_cast<svgpu_keypoint*>(undist_keypts.data()),
                                        reinterpret_cast<double*>(bearings.data()), cell_off.data(), cell_items.data()),
          "svgpu_frame_observation");
    cell_items.resize((size_t)cell_off.back());
}

void base::can_observe(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& trans_wc, const landmark_set& lms, float ray_cos_thr,
                       unsigned int num_levels, float log_scale_factor, observability& out) const {
    const int n = (int)lms.pos_w.size();
    out.visible.assign((size_t)n, 0);
    out.reproj.assign((size_t)n, Vec2_t{0, 0});
    out.x_right.assign((size_t)n, 0.f);
    out.pred_scale_level.assign((size_t)n, -1);
    check(ctx_, svgpu_reproject_landmarks(ctx_, &c_, rot_cw.data(), trans_cw.data(), trans_wc.data(), n, reinterpret_cast<const double*>(lms.pos_w.data()),
                                          reinterpret_cast<const double*>(lms.mean_normal.data()), lms.min_valid_dist.data(), lms.max_valid_dist.data(),
                                          lms.skip.empty() ? nullptr : lms.skip.data(), ray_cos_thr, (int)num_levels, log_scale_factor,
                                          out.visible.data(), reinterpret_cast<double*>(out.reproj.data()), out.x_right.data(), out.pred_scale_level.data()),
          "svgpu_reproject_landmarks");
}

}  // namespace camera

namespace data {
void compute_descriptors(svgpu_ctx* ctx, const std::vector<int>& obs_off, const cv::Mat& obs_desc, std::vector<int>& best_obs, cv::Mat& descriptors) {
    const int n = (int)obs_off.size() - 1;
    best_obs.assign((size_t)std::max(n, 1), 0);
    if (n <= 0) return;
    std::vector<uint8_t> packed((size_t)obs_desc.rows * 32);
    for (int i = 0; i < obs_desc.rows; ++i) std::memcpy(&packed[(size_t)i * 32], obs_desc.ptr(i), 32);
    descriptors.create(n, 32, CV_8U);
    const int rc = svgpu_landmarks_compute_descriptor(ctx, n, obs_off.data(), packed.data(), best_obs.data(), descriptors.ptr(0));
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_landmarks_compute_descriptor: ") + svgpu_last_error(ctx));
    best_obs.resize((size_t)n);
}

void update_mean_normal_and_obs_scale_variance(svgpu_ctx* ctx, const std::vector<int>& obs_off, const std::vector<Vec3_t>& obs_trans_wc,
                                               const std::vector<Vec3_t>& pos_w, const std::vector<Vec3_t>& ref_trans_wc,
                                               const std::vector<float>& ref_scale_factor, float inv_scale_factor_last,
                                               std::vector<Vec3_t>& mean_normal, std::vector<float>& max_valid_dist,
                                               std::vector<float>& min_valid_dist) {
    const int n = (int)pos_w.size();
    mean_normal.assign((size_t)n, Vec3_t{0, 0, 0});
    max_valid_dist.assign((size_t)n, 0.f);
    min_valid_dist.assign((size_t)n, 0.f);
    const int rc = svgpu_landmarks_update_geometry(ctx, n, obs_off.data(), reinterpret_cast<const double*>(obs_trans_wc.data()),
                                                   reinterpret_cast<const double*>(pos_w.data()), reinterpret_cast<const double*>(ref_trans_wc.data()),
                                                   ref_scale_factor.data(), inv_scale_factor_last, reinterpret_cast<double*>(mean_normal.data()),
                                                   max_valid_dist.data(), min_valid_dist.data());
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_landmarks_update_geometry: ") + svgpu_last_error(ctx));
}

bow_vocabulary_hip::bow_vocabulary_hip(svgpu_ctx* ctx, const std::vector<int>& child_off, const std::vector<int>& children, const cv::Mat& node_desc,
                                       const std::vector<float>& node_weight, const std::vector<int>& word_id, int depth, int fbow_k)
    : ctx_(ctx), depth_(depth), fbow_k_(fbow_k) {
    const int n = (int)child_off.size() - 1;
    std::vector<uint8_t> packed((size_t)node_desc.rows * 32);
    for (int i = 0; i < node_desc.rows; ++i) std::memcpy(&packed[(size_t)i * 32], node_desc.ptr(i), 32);
    const int rc = svgpu_bow_vocabulary_upload(ctx, n, child_off.data(), children.data(), packed.data(), node_weight.data(), word_id.data(), &vocab_);
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_bow_vocabulary_upload: ") + svgpu_last_error(ctx));
}

bow_vocabulary_hip::~bow_vocabulary_hip() { svgpu_bow_vocabulary_free(vocab_); }

void bow_vocabulary_hip::compute_bow(const cv::Mat& descriptors, std::map<unsigned int, double>& bow_vec,
                                     std::map<unsigned int, std::vector<unsigned int>>& bow_feat_vec, int levels_up) const {
    bow_vec.clear();
    bow_feat_vec.clear();
    const int n = descriptors.rows;
    if (n == 0) return;
    std::vector<uint8_t> packed((size_t)n * 32);
    for (int i = 0; i < n; ++i) std::memcpy(&packed[(size_t)i * 32], descriptors.ptr(i), 32);
    std::vector<int32_t> word((size_t)n), node((size_t)n);
    std::vector<float> weight((size_t)n);
    if (fbow_k_ > 0) {  // fbow::Vocabulary::transform(descriptors, levels_up (= 4, counted from the ROOT), bow_vec, bow_feat_vec)
        std::vector<uint32_t> code((size_t)n);
        const int rf = svgpu_fbow_transform(ctx_, vocab_, packed.data(), n, levels_up, fbow_k_, word.data(), weight.data(), code.data());
        if (rf != SVGPU_OK) throw std::runtime_error(std::string("svgpu_fbow_transform: ") + svgpu_last_error(ctx_));
        for (int i = 0; i < n; ++i) {
            bow_vec[(unsigned int)word[(size_t)i]] += (double)weight[(size_t)i];
            bow_feat_vec[code[(size_t)i]].push_back((unsigned int)i);
        }
        double norm = 0.0;
        for (const auto& kv : bow_vec) norm += kv.second * kv.second;
        if (norm > 0.0)
            for (auto& kv : bow_vec) kv.second *= 1.0 / std::sqrt(norm);
        return;
    }
    const int rc = svgpu_bow_transform(ctx_, vocab_, packed.data(), n, std::max(depth_ - levels_up, 0), word.data(), weight.data(), node.data());
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_bow_transform: ") + svgpu_last_error(ctx_));
    for (int i = 0; i < n; ++i)
        if (weight[(size_t)i] > 0.f) {  // zero-weight words (stop words) are skipped
            bow_vec[(unsigned int)word[(size_t)i]] += (double)weight[(size_t)i];
            bow_feat_vec[(unsigned int)node[(size_t)i]].push_back((unsigned int)i);
        }
    double norm = 0.0;
    for (const auto& kv : bow_vec) norm += std::fabs(kv.second);
    if (norm > 0.0)
        for (auto& kv : bow_vec) kv.second /= norm;
}
}  // namespace data
}  // namespace stella_vslam_hip
