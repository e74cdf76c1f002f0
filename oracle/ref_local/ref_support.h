// Helpers of the matcher fixtures (oracle/ref_local/ref_match_exports.cc): the concrete stand-in camera (reprojections = the oracle's)
// and builders that turn the flat arrays of the tests into the object graph the reference's matcher methods take.
#ifndef SVREF_SUPPORT_H
#define SVREF_SUPPORT_H
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "stella_vslam/camera/base.h"
#if defined(SVREF_DROP_IN) || defined(SVREF_CAMERA_PARAMS)
#define SVREF_ANY_MODEL 1  // the camera carries its parameter members (shim_mdrop/): the product's camera conversion and the reference's BA edges read them
#include "stella_vslam/camera/any_model.h"
#endif
#include "stella_vslam/data/frame.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"

extern "C" {
typedef struct {
    int32_t model;
    int32_t pad_;
    double cols, rows;
    double fx, fy, cx, cy;
    double dist[5];
    double focal_x_baseline;
    float min_x, max_x, min_y, max_y;
} orc_camera;
int orc_reproject_to_image(const orc_camera* c, const double* R, const double* t, const double* pw, double* reproj, float* x_right);
int orc_reproject_to_bearing(const orc_camera* c, const double* rot_cw, const double* trans_cw, const double* pos_w, double* bearing);
}

namespace svref {
void forget_grids();

using namespace stella_vslam;

inline void to_row_major(const Mat33_t& R, double* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = R(i, j);
}
inline Mat33_t mat33(const double* row_major) {
    Mat33_t R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R(i, j) = row_major[3 * i + j];
    return R;
}
inline Vec3_t vec3(const double* p) { return Vec3_t(p[0], p[1], p[2]); }
inline Mat44_t pose44(const double* rot_row_major, const double* trans) {
    Mat44_t T = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T(i, j) = rot_row_major[3 * i + j];
        T(i, 3) = trans[i];
    }
    return T;
}

#ifdef SVREF_ANY_MODEL
using camera_parent = camera::any_model;  // shim_mdrop/: carries the parameter members the product's camera conversion reads
#else
using camera_parent = camera::base;
#endif
class camera_fixture final : public camera_parent {
public:
    camera_fixture(const orc_camera* c, bool monocular, double true_baseline)
        : camera_parent(monocular ? camera::setup_type_t::Monocular : camera::setup_type_t::Stereo, (camera::model_type_t)c->model, (unsigned)c->cols,
                       (unsigned)c->rows, c->focal_x_baseline, true_baseline),
          oc_(*c) {
        img_bounds_.min_x_ = c->min_x, img_bounds_.max_x_ = c->max_x, img_bounds_.min_y_ = c->min_y, img_bounds_.max_y_ = c->max_y;
#ifdef SVREF_ANY_MODEL
        fx_ = c->fx, fy_ = c->fy, cx_ = c->cx, cy_ = c->cy;
        if (c->model == 0) k1_ = c->dist[0], k2_ = c->dist[1], p1_ = c->dist[2], p2_ = c->dist[3], k3_ = c->dist[4];
        if (c->model == 1) k1_ = c->dist[0], k2_ = c->dist[1], k3_ = c->dist[2], k4_ = c->dist[3];
        if (c->model == 3) distortion_ = c->dist[0];
#endif
    }
    bool reproject_to_image(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec2_t& reproj, float& x_right) const override {
        double R[9], r[2];
        to_row_major(rot_cw, R);
        const int ok = orc_reproject_to_image(&oc_, R, trans_cw.data(), pos_w.data(), r, &x_right);
        reproj(0) = r[0], reproj(1) = r[1];
        return ok != 0;
    }
    bool reproject_to_bearing(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec3_t& reproj) const override {
        double R[9];
        to_row_major(rot_cw, R);
        return orc_reproject_to_bearing(&oc_, R, trans_cw.data(), pos_w.data(), reproj.data()) != 0;
    }

private:
    orc_camera oc_;
};

// keypoints + descriptors of one image (descriptors are viewed, not copied)
inline void fill_observation(data::frame_observation& fo, const uint8_t* desc, const float* xy, const int32_t* octave, const float* angle, const float* xright,
                             const double* bearings, int n, int grid_cols, int grid_rows) {
    fo.descriptors_ = cv::Mat(n, 32, CV_8UC1, const_cast<uint8_t*>(desc), 32);
    fo.undist_keypts_.resize(n);
    for (int i = 0; i < n; ++i) {
        cv::KeyPoint& k = fo.undist_keypts_[i];
        k.pt.x = xy ? xy[2 * i] : 0.f, k.pt.y = xy ? xy[2 * i + 1] : 0.f;
        k.octave = octave ? octave[i] : 0;
        k.angle = angle ? angle[i] : 0.f;
    }
    if (xright) fo.stereo_x_right_.assign(xright, xright + n);
    if (bearings) {
        fo.bearings_.resize(n);
        for (int i = 0; i < n; ++i) fo.bearings_[i] = vec3(bearings + 3 * i);
    }
    fo.num_grid_cols_ = grid_cols, fo.num_grid_rows_ = grid_rows;
}
inline data::bow_feature_vector feat_vec(const int32_t* node, int n) {
    data::bow_feature_vector f;
    if (node)
        for (int i = 0; i < n; ++i)
            if (node[i] >= 0) f[(unsigned)node[i]].push_back((unsigned)i);
    return f;
}
inline std::shared_ptr<data::landmark> make_landmark(unsigned id, const double* pos_w, const uint8_t* desc32, float min_valid, float max_valid, const double* normal,
                                                     bool has_observation = true) {
    auto lm = std::make_shared<data::landmark>(id, pos_w ? vec3(pos_w) : Vec3_t());
    if (desc32) lm->descriptor_ = cv::Mat(1, 32, CV_8UC1, const_cast<uint8_t*>(desc32), 32);
    lm->min_valid_dist_ = min_valid, lm->max_valid_dist_ = max_valid;
    if (normal) lm->mean_normal_ = vec3(normal);
    lm->has_observation_ = has_observation;
    return lm;
}
}  // namespace svref
#endif
