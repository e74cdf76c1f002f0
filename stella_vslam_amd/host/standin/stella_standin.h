// Stand-ins for the stella_vslam types the drop-in classes of host/drop_in/ are written against, used ONLY when the reference's
// headers (and Eigen / OpenCV / yaml-cpp) are not available -- this container.  With the reference tree on the include path compile
// with -DSVGPU_WITH_STELLA_VSLAM and host/drop_in/*.cc include "stella_vslam/..." instead; every member used there exists under
// the same name and meaning in the reference (file:line cited per class).  The stand-ins implement just enough behaviour
// (landmark <-> keyframe bookkeeping, pose accessors) for host/test_drop_in.cpp to run the classes on a toy map.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "cv_standin.h"
#include "../drop_in/map_mirror.h"

namespace stella_vslam {

// ---- type.h:44-62 (Eigen fixed-size matrices: column-major storage, (i, j) element access)
template <int R, int C>
struct small_mat {
    double v[R * C] = {};
    double& operator()(int i, int j) { return v[j * R + i]; }
    double operator()(int i, int j) const { return v[j * R + i]; }
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double* data() { return v; }
    const double* data() const { return v; }
    static small_mat Identity() {
        small_mat m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
        return m;
    }
    static small_mat Zero() { return small_mat(); }
};
// matrix product as Eigen evaluates a fixed-size one: every element the dot product of a row and a column, accumulated left to right
template <int R, int K, int C>
inline small_mat<R, C> operator*(const small_mat<R, K>& a, const small_mat<K, C>& b) {
    small_mat<R, C> m;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) {
            double acc = a(i, 0) * b(0, j);
            for (int k = 1; k < K; ++k) acc += a(i, k) * b(k, j);
            m(i, j) = acc;
        }
    return m;
}
using Mat33_t = small_mat<3, 3>;
using Mat44_t = small_mat<4, 4>;
using Vec2_t = small_mat<2, 1>;
using Vec3_t = small_mat<3, 1>;
template <typename T>
using eigen_alloc_vector = std::vector<T>;
template <typename T, typename U>
using eigen_alloc_unord_map = std::unordered_map<T, U>;

// ---- YAML::Node as far as the factories read it: yaml_node["key"].as<T>(default)
namespace yaml_standin {
struct Value {
    const std::string* s = nullptr;
    template <typename T>
    T as(const T& def) const;
};
template <>
inline std::string Value::as<std::string>(const std::string& def) const { return s ? *s : def; }
template <>
inline bool Value::as<bool>(const bool& def) const { return s ? (*s == "true" || *s == "1") : def; }
struct Node {
    std::map<std::string, std::string> kv;
    Value operator[](const std::string& k) const {
        auto it = kv.find(k);
        return Value{it == kv.end() ? nullptr : &it->second};
    }
};
}  // namespace yaml_standin

namespace solve {
// solve/essential_solver.h:12-80 -- INTERFACE stand-in.  The reference's class (5-point / 8-point RANSAC on Eigen's SVD and eigen-solver)
// is host code the drop-in classes call as it is; without the reference tree this stand-in keeps match::hip::robust compilable and
// testable: it accepts every match of a large enough set (so the wrappers' own logic -- list order, landmark assignment, return value --
// runs), and records what it was asked for.
class essential_solver {
public:
    essential_solver(const eigen_alloc_vector<Vec3_t>&, const eigen_alloc_vector<Vec3_t>&, const std::vector<std::pair<int, int>>& matches_12, bool use_fixed_seed = false)
        : matches_12_(matches_12), use_fixed_seed_(use_fixed_seed) {}
    void find_via_ransac(const unsigned int max_num_iter, const bool recompute = true, const unsigned int min_set_size = 5) {
        max_num_iter_ = max_num_iter;
        recompute_ = recompute;
        solution_is_valid_ = matches_12_.size() >= min_set_size;  // essential_solver.cc:19-23
        is_inlier_match_.assign(matches_12_.size(), solution_is_valid_);
    }
    bool solution_is_valid() const { return solution_is_valid_; }
    std::vector<bool> get_inlier_matches() const { return is_inlier_match_; }
    unsigned int max_num_iter_ = 0;
    bool recompute_ = false;

private:
    const std::vector<std::pair<int, int>>& matches_12_;
    bool use_fixed_seed_, solution_is_valid_ = false;
    std::vector<bool> is_inlier_match_;
};
}  // namespace solve

namespace feature {
// feature/orb_params.h:13-71
struct orb_params {
    orb_params(float scale_factor = 1.2f, unsigned int num_levels = 8) : scale_factor_(scale_factor), log_scale_factor_(std::log(scale_factor)), num_levels_(num_levels) {
        scale_factors_.assign(num_levels, 1.0f);
        for (unsigned int l = 1; l < num_levels; ++l) scale_factors_[l] = scale_factor * scale_factors_[l - 1];
        for (unsigned int l = 0; l < num_levels; ++l) {
            inv_scale_factors_.push_back(1.0f / scale_factors_[l]);
            level_sigma_sq_.push_back(scale_factors_[l] * scale_factors_[l]);
            inv_level_sigma_sq_.push_back(1.0f / level_sigma_sq_[l]);
        }
    }
    float scale_factor_, log_scale_factor_;
    unsigned int num_levels_;
    std::vector<float> scale_factors_, inv_scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
};
}  // namespace feature

namespace camera {
enum class setup_type_t { Monocular = 0, Stereo = 1, RGBD = 2 };                                  // camera/base.h:18-22
enum class model_type_t { Perspective = 0, Fisheye = 1, Equirectangular = 2, RadialDivision = 3 };  // camera/base.h:24-29
struct image_bounds {
    float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;
};
class base {  // camera/base.h:56-201
public:
    base(setup_type_t setup, model_type_t model, unsigned int cols, unsigned int rows, double focal_x_baseline, double true_baseline)
        : setup_type_(setup), model_type_(model), cols_(cols), rows_(rows), focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline) {}
    virtual ~base() = default;
    const setup_type_t setup_type_;
    const model_type_t model_type_;
    const unsigned int cols_, rows_;
    const double focal_x_baseline_, true_baseline_;
    image_bounds img_bounds_;
};
class perspective final : public base {  // camera/perspective.h:54-73
public:
    perspective(setup_type_t setup, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double k1, double k2, double p1,
                double p2, double k3, double focal_x_baseline = 0.0)
        : base(setup, model_type_t::Perspective, cols, rows, focal_x_baseline, fx != 0 ? focal_x_baseline / fx : 0.0), fx_(fx), fy_(fy), cx_(cx), cy_(cy), k1_(k1),
          k2_(k2), p1_(p1), p2_(p2), k3_(k3) {}
    const double fx_, fy_, cx_, cy_, k1_, k2_, p1_, p2_, k3_;
};
class fisheye final : public base {  // camera/fisheye.h:48-67
public:
    fisheye(setup_type_t setup, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double k1, double k2, double k3, double k4,
            double focal_x_baseline = 0.0)
        : base(setup, model_type_t::Fisheye, cols, rows, focal_x_baseline, fx != 0 ? focal_x_baseline / fx : 0.0), fx_(fx), fy_(fy), cx_(cx), cy_(cy), k1_(k1), k2_(k2),
          k3_(k3), k4_(k4) {}
    const double fx_, fy_, cx_, cy_, k1_, k2_, k3_, k4_;
};
class radial_division final : public base {  // camera/radial_division.h:46-61
public:
    radial_division(setup_type_t setup, unsigned int cols, unsigned int rows, double fx, double fy, double cx, double cy, double distortion,
                    double focal_x_baseline = 0.0)
        : base(setup, model_type_t::RadialDivision, cols, rows, focal_x_baseline, fx != 0 ? focal_x_baseline / fx : 0.0), fx_(fx), fy_(fy), cx_(cx), cy_(cy),
          distortion_(distortion) {}
    const double fx_, fy_, cx_, cy_, distortion_;
};
class equirectangular final : public base {  // camera/equirectangular.h
public:
    equirectangular(unsigned int cols, unsigned int rows) : base(setup_type_t::Monocular, model_type_t::Equirectangular, cols, rows, 0.0, 0.0) {}
};
}  // namespace camera

namespace data {

class keyframe;
class landmark;
class map_database;

// data/frame_observation.h:12-38
struct frame_observation {
    cv::Mat descriptors_;
    std::vector<cv::KeyPoint> undist_keypts_;
    eigen_alloc_vector<Vec3_t> bearings_;
    std::vector<float> stereo_x_right_;
    std::vector<float> depths_;
    unsigned int num_grid_cols_ = 64, num_grid_rows_ = 48;
};
using bow_feature_vector = std::map<unsigned int, std::vector<unsigned int>>;  // data/bow_vocabulary_fwd.h (fbow::BoWFeatVector)

template <class T>
struct id_less {  // type.h: ordering of the weak_ptr keys of landmark::observations_t by id (T = std::weak_ptr<keyframe>)
    bool operator()(const T& a, const T& b) const;
};

class graph_node {  // data/graph_node.h:65, 149
public:
    std::vector<std::shared_ptr<keyframe>> get_covisibilities() const { return covisibilities_; }
    bool is_spanning_root() const { return spanning_root_; }
    std::vector<std::shared_ptr<keyframe>> covisibilities_;
    bool spanning_root_ = false;
};

class map_database {  // data/map_database.h:52, 270
public:
    unsigned int get_fixed_keyframe_id_threshold() { return fixed_keyframe_id_threshold_; }
    static std::mutex mtx_database_;
    unsigned int fixed_keyframe_id_threshold_ = 0;
    unsigned int num_erased_landmarks_ = 0;
};
inline std::mutex map_database::mtx_database_;

class landmark : public std::enable_shared_from_this<landmark> {  // data/landmark.h:29-171
public:
    using observations_t = std::map<std::weak_ptr<keyframe>, unsigned int, id_less<std::weak_ptr<keyframe>>>;
    // (the hip::map_mirror calls are the lines the reference's data/landmark.cc gains with this backend: drop_in/map_mirror.h)
    landmark(unsigned int id, const Vec3_t& pos_w) : id_(id), pos_w_(pos_w) {
        const double p[3] = {pos_w(0), pos_w(1), pos_w(2)};
        hip::map_mirror::landmark_created(id_, p);
    }
    void set_pos_in_world(const Vec3_t& pos_w) {
        pos_w_ = pos_w;
        const double p[3] = {pos_w(0), pos_w(1), pos_w(2)};
        hip::map_mirror::set_position(id_, p);
    }
    Vec3_t get_pos_in_world() const { return pos_w_; }
    Vec3_t get_obs_mean_normal() const { return mean_normal_; }
    float get_min_valid_distance() const { return min_valid_dist_; }
    float get_max_valid_distance() const { return max_valid_dist_; }
    cv::Mat get_descriptor() const { return descriptor_; }
    observations_t get_observations() const { return observations_; }
    unsigned int num_observations() const { return (unsigned)observations_.size(); }
    bool has_observation() const { return !observations_.empty(); }
    bool will_be_erased() { return will_be_erased_; }
    int get_index_in_keyframe(const std::shared_ptr<keyframe>& keyfrm) const {
        auto it = observations_.find(keyfrm);
        return it == observations_.end() ? -1 : (int)it->second;
    }
    bool is_observed_in_keyframe(const std::shared_ptr<keyframe>& keyfrm) const { return observations_.count(keyfrm) != 0; }
    void add_observation(const std::shared_ptr<keyframe>& keyfrm, unsigned int idx) {
        observations_[keyfrm] = idx;
        hip::map_mirror::set_has_observation(id_, true);
    }
    void erase_observation(map_database* map_db, const std::shared_ptr<keyframe>& keyfrm) {  // data/landmark.cc:100-140
        observations_.erase(keyfrm);
        hip::map_mirror::set_has_observation(id_, !observations_.empty());
        if (observations_.size() <= 2) {
            will_be_erased_ = true;
            hip::map_mirror::landmark_erased(id_);
            if (map_db) ++map_db->num_erased_landmarks_;
        }
    }
    unsigned int predict_scale_level(const float cam_to_lm_dist, float num_scale_levels, float log_scale_factor) const {  // landmark.cc:336-353
        const float ratio = max_valid_dist_ / cam_to_lm_dist;
        const auto pred_scale_level = static_cast<int>(std::ceil(std::log(ratio) / log_scale_factor));
        if (pred_scale_level < 0) return 0;
        else if (num_scale_levels <= static_cast<unsigned int>(pred_scale_level)) return num_scale_levels - 1;
        else return static_cast<unsigned int>(pred_scale_level);
    }
    void increase_num_observable(unsigned int num_observable = 1) { num_observable_ += num_observable; }   // data/landmark.cc:355-361
    void increase_num_observed(unsigned int num_observed = 1) { num_observed_ += num_observed; }
    std::atomic<unsigned int> num_observable_{1}, num_observed_{1};
    void compute_descriptor();                         // defined behind keyframe (data/landmark.cc:199-254)
    void update_mean_normal_and_obs_scale_variance();  // data/landmark.cc:256-318
    std::shared_ptr<keyframe> get_ref_keyframe() const { return ref_keyfrm_.lock(); }  // data/landmark.cc:66-69
    // The two setters the batched write-back of optimize::local_bundle_adjuster_hip needs (what data/landmark.h gains):
    // they store what update_mean_normal_and_obs_scale_variance() / compute_descriptor() would have computed -- the values come from
    // svgpu_landmarks_update_geometry / svgpu_landmarks_compute_descriptor, one device call for all landmarks of the window.
#define SVGPU_LANDMARK_HAS_BATCH_SETTERS 1
    void set_prediction_parameters(const Vec3_t& mean_normal, float min_valid_dist, float max_valid_dist) {
        ++num_geometry_refreshes_;
        mean_normal_ = mean_normal;
        min_valid_dist_ = min_valid_dist;
        max_valid_dist_ = max_valid_dist;
        const double nv[3] = {mean_normal(0), mean_normal(1), mean_normal(2)};
        hip::map_mirror::set_geometry(id_, nv, min_valid_dist_, max_valid_dist_);
    }
    void set_representative_descriptor(const unsigned char* descriptor32) {
        ++num_descriptor_refreshes_;
        descriptor_.create(1, 32, CV_8U);
        std::memcpy(descriptor_.ptr(0), descriptor32, 32);
        hip::map_mirror::set_descriptor(id_, descriptor_.ptr(0));
    }
    unsigned int id_;
    // test bookkeeping
    unsigned int num_descriptor_refreshes_ = 0, num_geometry_refreshes_ = 0;
    std::weak_ptr<keyframe> ref_keyfrm_;
    Vec3_t pos_w_, mean_normal_;
    float min_valid_dist_ = 0, max_valid_dist_ = 0;
    cv::Mat descriptor_;

private:
    observations_t observations_;
    std::atomic<bool> will_be_erased_{false};
};

class marker2d {  // data/marker2d.h: the undistorted image corners global BA reads
public:
    std::vector<cv::Point2f> undist_corners_;
};
class marker {  // data/marker.h:18-48
public:
    unsigned int id_ = 0;
    bool keep_fixed_ = false, initialized_before_ = false;
    std::map<unsigned int, std::shared_ptr<keyframe>> observations_;
    eigen_alloc_vector<Vec3_t> corners_pos_w_;
};

class keyframe : public std::enable_shared_from_this<keyframe> {  // data/keyframe.h:73-325
public:
    std::unordered_map<unsigned int, marker2d> markers_2d_;
    std::vector<std::shared_ptr<marker>> markers_;
    std::vector<std::shared_ptr<marker>> get_markers() const { return markers_; }
    keyframe(unsigned int id, camera::base* camera, const feature::orb_params* orb_params) : id_(id), camera_(camera), orb_params_(orb_params), graph_node_(new graph_node()) {}
    void set_pose_cw(const Mat44_t& pose_cw) { pose_cw_ = pose_cw; }
    Mat44_t get_pose_cw() const { return pose_cw_; }
    Mat33_t get_rot_cw() const {
        Mat33_t R;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R(i, j) = pose_cw_(i, j);
        return R;
    }
    Vec3_t get_trans_cw() const {
        Vec3_t t;
        for (int i = 0; i < 3; ++i) t(i) = pose_cw_(i, 3);
        return t;
    }
    Vec3_t get_trans_wc() const {  // -R^T t
        Vec3_t c;
        for (int i = 0; i < 3; ++i) c(i) = ((-pose_cw_(0, i)) * pose_cw_(0, 3) + (-pose_cw_(1, i)) * pose_cw_(1, 3)) + (-pose_cw_(2, i)) * pose_cw_(2, 3);
        return c;
    }
    void add_landmark(std::shared_ptr<landmark> lm, const unsigned int idx) { landmarks_.at(idx) = lm; }
    void erase_landmark(const std::shared_ptr<landmark>& lm) {
        const int idx = lm->get_index_in_keyframe(shared_from_this());
        if (0 <= idx) landmarks_.at(idx) = nullptr;
    }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    std::shared_ptr<landmark>& get_landmark(const unsigned int idx) { return landmarks_.at(idx); }
    bool will_be_erased() { return will_be_erased_; }
    unsigned int id_;
    camera::base* camera_;
    const feature::orb_params* orb_params_;
    frame_observation frm_obs_;
    bow_feature_vector bow_feat_vec_;
    std::unique_ptr<graph_node> graph_node_;
    std::vector<std::shared_ptr<landmark>> landmarks_;
    std::atomic<bool> will_be_erased_{false};

private:
    Mat44_t pose_cw_ = Mat44_t::Identity();
};

template <class T>
bool id_less<T>::operator()(const T& a, const T& b) const {
    const auto pa = a.lock(), pb = b.lock();
    return (pa ? (long long)pa->id_ : -1) < (pb ? (long long)pb->id_ : -1);
}

inline void landmark::compute_descriptor() {  // median-of-distances representative (data/landmark.cc:199-254)
    std::vector<const uint8_t*> descs;
    for (const auto& obs : observations_)
        if (auto kf = obs.first.lock())
            if (!kf->will_be_erased()) descs.push_back(kf->frm_obs_.descriptors_.ptr((int)obs.second));
    ++num_descriptor_refreshes_;
    if (descs.empty()) return;
    const size_t n = descs.size();
    unsigned best_median = 257;
    size_t best = 0;
    for (size_t i = 0; i < n; ++i) {
        std::vector<unsigned> d(n);
        for (size_t j = 0; j < n; ++j) {
            unsigned h = 0;
            for (int b = 0; b < 32; ++b) h += (unsigned)__builtin_popcount(descs[i][b] ^ descs[j][b]);
            d[j] = h;
        }
        std::sort(d.begin(), d.end());
        const unsigned med = d[(unsigned)(0.5 * (n - 1))];
        if (med < best_median) {
            best_median = med;
            best = i;
        }
    }
    descriptor_.create(1, 32, CV_8U);
    std::memcpy(descriptor_.ptr(0), descs[best], 32);
    hip::map_mirror::set_descriptor(id_, descriptor_.ptr(0));
}

inline void landmark::update_mean_normal_and_obs_scale_variance() {  // data/landmark.cc:256-318
    ++num_geometry_refreshes_;
    if (observations_.empty()) return;
    double m[3] = {0, 0, 0};
    for (const auto& obs : observations_) {
        auto kf = obs.first.lock();
        if (!kf) continue;
        const Vec3_t c = kf->get_trans_wc();
        const double v[3] = {pos_w_(0) - c(0), pos_w_(1) - c(1), pos_w_(2) - c(2)};
        const double nrm = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        for (int k = 0; k < 3; ++k) m[k] += v[k] / nrm;
    }
    const double mn = std::sqrt((m[0] * m[0] + m[1] * m[1]) + m[2] * m[2]);
    for (int k = 0; k < 3; ++k) mean_normal_(k) = mn > 0 ? m[k] / mn : 0.0;
    if (auto ref = ref_keyfrm_.lock()) {
        const int idx = get_index_in_keyframe(ref);
        if (0 <= idx) {
            const Vec3_t c = ref->get_trans_wc();
            const double v[3] = {pos_w_(0) - c(0), pos_w_(1) - c(1), pos_w_(2) - c(2)};
            const double dist = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
            const auto& sf = ref->orb_params_->scale_factors_;
            max_valid_dist_ = (float)(dist * sf.at(ref->frm_obs_.undist_keypts_.at(idx).octave));
            min_valid_dist_ = max_valid_dist_ * ref->orb_params_->inv_scale_factors_.at(ref->orb_params_->num_levels_ - 1);
        }
    }
    const double nv[3] = {mean_normal_(0), mean_normal_(1), mean_normal_(2)};
    hip::map_mirror::set_geometry(id_, nv, min_valid_dist_, max_valid_dist_);
}

class frame {  // data/frame.h:42-183
public:
    frame(unsigned int id, camera::base* camera, const feature::orb_params* orb_params) : id_(id), camera_(camera), orb_params_(orb_params) {}
    // data/frame.cc:18-22: the frame of an observation -- one (empty) landmark slot per keypoint
    frame(unsigned int id, camera::base* camera, const feature::orb_params* orb_params, const frame_observation& frm_obs)
        : id_(id), camera_(camera), orb_params_(orb_params), frm_obs_(frm_obs), landmarks_(frm_obs.undist_keypts_.size(), nullptr) {}
    void set_pose_cw(const Mat44_t& pose_cw) { pose_cw_ = pose_cw; }
    Mat44_t get_pose_cw() const { return pose_cw_; }
    Mat33_t get_rot_cw() const {
        Mat33_t R;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R(i, j) = pose_cw_(i, j);
        return R;
    }
    Vec3_t get_trans_cw() const {
        Vec3_t t;
        for (int i = 0; i < 3; ++i) t(i) = pose_cw_(i, 3);
        return t;
    }
    void add_landmark(const std::shared_ptr<landmark>& lm, const unsigned int idx) { landmarks_.at(idx) = lm; }
    void erase_landmark_with_index(const unsigned int idx) { landmarks_.at(idx) = nullptr; }                  // data/frame.cc:101-105
    void erase_landmarks() { std::fill(landmarks_.begin(), landmarks_.end(), nullptr); }                      // :118-121
    std::shared_ptr<landmark> get_landmark(const unsigned int idx) const { return landmarks_.at(idx); }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    void set_landmarks(const std::vector<std::shared_ptr<landmark>>& lms) { landmarks_ = lms; }
    unsigned int id_;
    camera::base* camera_;
    const feature::orb_params* orb_params_;
    frame_observation frm_obs_;
    bow_feature_vector bow_feat_vec_;
    std::shared_ptr<keyframe> ref_keyfrm_ = nullptr;  // data/frame.h:183 (set by tracking_module.cc:202 before track_current_frame)
    std::vector<std::shared_ptr<landmark>> landmarks_;

private:
    Mat44_t pose_cw_ = Mat44_t::Identity();
};

}  // namespace data
}  // namespace stella_vslam

namespace YAML {
using Node = stella_vslam::yaml_standin::Node;
}
