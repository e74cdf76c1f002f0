// The reference's camera models (camera/base.cc, perspective.cc, fisheye.cc, equirectangular.cc, radial_division.cc, compiled where they
// lie over the stand-in headers of shim/) behind C exports on flat arrays.  Separate library (oracle/_ref/libsvref_cam.so): it uses the
// reference's REAL camera/base.h, the matcher library a stand-in of it.  Test infrastructure only (tests/test_ref_local_camera.py).
#include <cstring>
#include <memory>
#include <vector>

#include "stella_vslam/camera/equirectangular.h"
#include "stella_vslam/camera/fisheye.h"
#include "stella_vslam/camera/perspective.h"
#include "stella_vslam/camera/radial_division.h"
#include "stella_vslam/data/common.h"
#include "stella_vslam/data/frame_observation.h"

using namespace stella_vslam;

namespace {
// model: 0 perspective (dist = k1 k2 p1 p2 k3), 1 fisheye (k1..k4), 2 equirectangular, 3 radial_division (dist[0])
std::unique_ptr<camera::base> make(int model, int stereo, unsigned cols, unsigned rows, double fx, double fy, double cx, double cy, const double* d, double fxb) {
    const auto setup = stereo ? camera::setup_type_t::Stereo : camera::setup_type_t::Monocular;
    const auto col = camera::color_order_t::Gray;
    switch (model) {
        case 0: return std::unique_ptr<camera::base>(new camera::perspective("ref", setup, col, cols, rows, 30.0, fx, fy, cx, cy, d[0], d[1], d[2], d[3], d[4], fxb));
        case 1: return std::unique_ptr<camera::base>(new camera::fisheye("ref", setup, col, cols, rows, 30.0, fx, fy, cx, cy, d[0], d[1], d[2], d[3], fxb));
        case 2: return std::unique_ptr<camera::base>(new camera::equirectangular("ref", col, cols, rows, 30.0));
        default: return std::unique_ptr<camera::base>(new camera::radial_division("ref", setup, col, cols, rows, 30.0, fx, fy, cx, cy, d[0], fxb));
    }
}
Mat33_t mat33(const double* r) {
    Mat33_t R;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R(i, j) = r[3 * i + j];
    return R;
}
}  // namespace

extern "C" {
// compute_image_bounds (constructor), undistort_keypoints, convert_keypoints_to_bearings, true_baseline_
void svref_camera_observation(int model, int stereo, unsigned cols, unsigned rows, double fx, double fy, double cx, double cy, const double* dist, double fxb,
                              int n, const float* xy, float* bounds4, float* undist_xy, double* bearings, double* true_baseline) {
    const auto cam = make(model, stereo, cols, rows, fx, fy, cx, cy, dist, fxb);
    bounds4[0] = cam->img_bounds_.min_x_, bounds4[1] = cam->img_bounds_.max_x_, bounds4[2] = cam->img_bounds_.min_y_, bounds4[3] = cam->img_bounds_.max_y_;
    *true_baseline = cam->true_baseline_;
    std::vector<cv::KeyPoint> in(n), out;
    for (int i = 0; i < n; ++i) in[i] = cv::KeyPoint(xy[2 * i], xy[2 * i + 1], 31.f, 12.5f, 1.f, 2, -1);
    cam->undistort_keypoints(in, out);
    eigen_alloc_vector<Vec3_t> b;
    cam->convert_keypoints_to_bearings(out, b);
    for (int i = 0; i < n; ++i) {
        undist_xy[2 * i] = out[i].pt.x, undist_xy[2 * i + 1] = out[i].pt.y;
        bearings[3 * i] = b[i](0), bearings[3 * i + 1] = b[i](1), bearings[3 * i + 2] = b[i](2);
    }
}

// reproject_to_image and reproject_to_bearing of n world points
void svref_camera_reproject(int model, int stereo, unsigned cols, unsigned rows, double fx, double fy, double cx, double cy, const double* dist, double fxb,
                            const double* rot_cw, const double* trans_cw, int n, const double* pos_w, uint8_t* in_image, double* reproj, float* x_right,
                            uint8_t* bearing_ok, double* bearing) {
    const auto cam = make(model, stereo, cols, rows, fx, fy, cx, cy, dist, fxb);
    const Mat33_t R = mat33(rot_cw);
    const Vec3_t t(trans_cw[0], trans_cw[1], trans_cw[2]);
    for (int i = 0; i < n; ++i) {
        const Vec3_t pw(pos_w[3 * i], pos_w[3 * i + 1], pos_w[3 * i + 2]);
        Vec2_t rp;
        float xr = 0.f;
        in_image[i] = cam->reproject_to_image(R, t, pw, rp, xr);
        reproj[2 * i] = rp(0), reproj[2 * i + 1] = rp(1);
        x_right[i] = xr;
        Vec3_t b;
        bearing_ok[i] = cam->reproject_to_bearing(R, t, pw, b);
        bearing[3 * i] = b(0), bearing[3 * i + 1] = b(1), bearing[3 * i + 2] = b(2);
    }
}

// data::assign_keypoints_to_grid + data::get_keypoints_in_cell (data/common.cc:83-190, compiled where it lies) for nq queries
// {ref_x, ref_y, margin} with level bounds; out: CSR of the returned indices in the returned order.  Returns the total count (or -1: cap).
int svref_grid_lookup(int model, unsigned cols, unsigned rows, double fx, double fy, double cx, double cy, const double* dist, int n, const float* xy,
                      const int32_t* octave, unsigned grid_cols, unsigned grid_rows, int nq, const float* q_xym, const int32_t* q_levels, int32_t* out_off,
                      int32_t* out_idx, int cap) {
    const auto cam = make(model, 0, cols, rows, fx, fy, cx, cy, dist, 0.0);
    data::frame_observation fo;
    fo.undist_keypts_.resize(n);
    for (int i = 0; i < n; ++i) {
        fo.undist_keypts_[i].pt.x = xy[2 * i], fo.undist_keypts_[i].pt.y = xy[2 * i + 1];
        fo.undist_keypts_[i].octave = octave[i];
    }
    fo.num_grid_cols_ = grid_cols, fo.num_grid_rows_ = grid_rows;
    data::assign_keypoints_to_grid(cam.get(), fo.undist_keypts_, fo.keypt_indices_in_cells_, grid_cols, grid_rows);
    int total = 0;
    out_off[0] = 0;
    for (int q = 0; q < nq; ++q) {
        const auto idx = data::get_keypoints_in_cell(cam.get(), fo, q_xym[3 * q], q_xym[3 * q + 1], q_xym[3 * q + 2], q_levels[2 * q], q_levels[2 * q + 1]);
        for (const auto i : idx) {
            if (total >= cap) return -1;
            out_idx[total++] = (int32_t)i;
        }
        out_off[q + 1] = total;
    }
    return total;
}
}
