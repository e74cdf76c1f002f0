// C exports over the parts of the reference that compile from their own sources (built by oracle/ref_local/Makefile from
// /root/reference into oracle/_ref/libsvref.so; tests/test_ref_local.py pins the oracle against them).  Test infrastructure only.
//   without any stand-in but cvFloor: util::cos / util::sin (util/trigonometric.h:19-49), util::angle::diff (util/angle.cc:7-16),
//                                     the rBRIEF sampling pattern (feature/orb_point_pairs.h:47)
//   over the stand-in OpenCV types of shim/ (primitives = the oracle's): feature::orb_impl (feature/orb_impl.cc: the orientation
//                                     moments, the rotated BRIEF tests and their bit order), feature::orb_params
//                                     (feature/orb_params.cc: the four tables), feature::orb_extractor::extract
//                                     (feature/orb_extractor.cc: everything around the primitives)
#include <cstring>
#include <vector>

#include "stella_vslam/feature/orb_extractor.h"
#include "stella_vslam/feature/orb_impl.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/feature/orb_point_pairs.h"
#include "stella_vslam/util/angle.h"
#include "stella_vslam/util/trigonometric.h"

extern "C" {
float svref_util_cos(float v) { return stella_vslam::util::cos(v); }
float svref_util_sin(float v) { return stella_vslam::util::sin(v); }
float svref_angle_diff(float a, float b) { return stella_vslam::util::angle::diff(a, b); }
unsigned int svref_orb_point_pairs_size(void) { return stella_vslam::feature::orb_point_pairs_size; }
const float* svref_orb_point_pairs(void) { return stella_vslam::feature::orb_point_pairs; }
void svref_util_cos_sin_array(const float* v, int n, float* c, float* s) {
    for (int i = 0; i < n; ++i) {
        c[i] = stella_vslam::util::cos(v[i]);
        s[i] = stella_vslam::util::sin(v[i]);
    }
}
void svref_angle_diff_array(const float* a, const float* b, int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = stella_vslam::util::angle::diff(a[i], b[i]);
}

// orb_impl on a caller image: angles of n points, then descriptors at those points with those angles
void svref_orb_impl(const unsigned char* img, int w, int h, int stride, const float* xy, int n, float* angle_out, unsigned char* desc_out) {
    const stella_vslam::feature::orb_impl impl;
    const cv::Mat image(h, w, CV_8UC1, const_cast<unsigned char*>(img), (size_t)stride);
    for (int i = 0; i < n; ++i) {
        angle_out[i] = impl.ic_angle(image, cv::Point2f(xy[2 * i], xy[2 * i + 1]));
        cv::KeyPoint kp(xy[2 * i], xy[2 * i + 1], 31.f, angle_out[i]);
        impl.compute_orb_descriptor(kp, image, desc_out + 32 * (size_t)i);
    }
}

// the four tables of orb_params (each num_levels floats)
void svref_orb_params_tables(float scale_factor, unsigned int num_levels, float* scale_factors, float* inv_scale_factors, float* level_sigma_sq,
                             float* inv_level_sigma_sq) {
    const stella_vslam::feature::orb_params p("ref", scale_factor, num_levels, 20, 7);
    memcpy(scale_factors, p.scale_factors_.data(), 4 * num_levels);
    memcpy(inv_scale_factors, p.inv_scale_factors_.data(), 4 * num_levels);
    memcpy(level_sigma_sq, p.level_sigma_sq_.data(), 4 * num_levels);
    memcpy(inv_level_sigma_sq, p.inv_level_sigma_sq_.data(), 4 * num_levels);
}

// orb_extractor::extract.  mask may be NULL; mask_rects = n_rects x {x_min/cols, x_max/cols, y_min/rows, y_max/rows}.
// kp_out: cap x 7 floats (x, y, size, angle, response, octave, class_id); returns the number of keypoints (or -1 if cap is too small);
// pyramid_out (may be NULL): levels 1.. packed back to back, tight rows.
int svref_orb_extract(const unsigned char* img, int w, int h, int stride, const unsigned char* mask, int mask_stride, float scale_factor,
                      unsigned int num_levels, unsigned int ini_thr, unsigned int min_thr, unsigned int min_area, const float* mask_rects,
                      int n_rects, float* kp_out, unsigned char* desc_out, int cap, unsigned char* pyramid_out) {
    const stella_vslam::feature::orb_params params("ref", scale_factor, num_levels, ini_thr, min_thr);
    std::vector<std::vector<float>> rects;
    for (int r = 0; r < n_rects; ++r) rects.push_back(std::vector<float>(mask_rects + 4 * r, mask_rects + 4 * r + 4));
    stella_vslam::feature::orb_extractor ext(&params, min_area, stella_vslam::feature::descriptor_type::ORB, rects);
    const cv::Mat image(h, w, CV_8UC1, const_cast<unsigned char*>(img), (size_t)stride);
    cv::Mat mask_mat;
    if (mask) mask_mat = cv::Mat(h, w, CV_8UC1, const_cast<unsigned char*>(mask), (size_t)mask_stride);
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    ext.extract(image, mask_mat, kps, desc);
    if ((int)kps.size() > cap) return -1;
    for (size_t i = 0; i < kps.size(); ++i) {
        float* o = kp_out + 7 * i;
        o[0] = kps[i].pt.x, o[1] = kps[i].pt.y, o[2] = kps[i].size, o[3] = kps[i].angle, o[4] = kps[i].response;
        o[5] = (float)kps[i].octave, o[6] = (float)kps[i].class_id;
        memcpy(desc_out + 32 * i, desc.ptr((int)i), 32);
    }
    if (pyramid_out)
        for (unsigned int l = 1; l < num_levels; ++l) {
            const cv::Mat& m = ext.image_pyramid_.at(l);
            for (int y = 0; y < m.rows; ++y) {
                memcpy(pyramid_out, m.ptr(y), m.cols);
                pyramid_out += m.cols;
            }
        }
    return (int)kps.size();
}
}
