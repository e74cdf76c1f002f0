// Device functions of the frame-observation / reprojection kernels, shared by frame_kernels.hip (one launch per step) and
// track_kernels.hip (the tracked-frame chain, where the same steps run fused inside other kernels).  Every expression keeps the
// reference's operation order (fp64, compiled without contraction): both users produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "frame_kernels.h"

namespace svfd {

constexpr double kPi = 3.14159265358979323846;

// cv::undistortPoints (R = I, P = K as CV_32F; criteria EPS | MAX_ITER, 20, 1e-6) -- camera/perspective.cc:21-22, 259-263
__device__ inline void cv_undistort_point(const svgpu_camera& c, float px, float py, float& ox, float& oy) {
    const double fx = (double)(float)c.fx, fy = (double)(float)c.fy, cx = (double)(float)c.cx, cy = (double)(float)c.cy;
    const double k0 = (double)(float)c.dist[0], k1 = (double)(float)c.dist[1], k2 = (double)(float)c.dist[2],
                 k3 = (double)(float)c.dist[3], k4 = (double)(float)c.dist[4];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double u = px, v = py;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    double error = 1.7976931348623157e308;
    for (int j = 0; j < 20; ++j) {
        if (error < 1e-6) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);  // k[5..7] = 0
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;  // k[8..11] = 0
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
        r2 = x * x + y * y;
        const double r4 = r2 * r2, r6 = r4 * r2, a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k0 * r2 + k1 * r4 + k4 * r6;
        const double icdist2 = 1. / (1 + 0. * r2 + 0. * r4 + 0. * r6);
        const double xd = x * cdist * icdist2 + k2 * a1 + k3 * a2 + 0. * r2 + 0. * r4;
        const double yd = y * cdist * icdist2 + k2 * a3 + k3 * a1 + 0. * r2 + 0. * r4;
        const double xp = xd * fx + cx, yp = yd * fy + cy;
        error = sqrt((xp - u) * (xp - u) + (yp - v) * (yp - v));
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);  // RR = P * I
    ox = (float)(xx * ww);
    oy = (float)(yy * ww);
}

// cv::fisheye::undistortPoints (P = K as CV_32F, default criteria MAX_ITER + EPS, 10, 1e-8) -- camera/fisheye.cc:21-22, 297
__device__ inline void cv_fisheye_undistort_point(const svgpu_camera& c, float px, float py, float& ox, float& oy) {
    const double fx = (double)(float)c.fx, fy = (double)(float)c.fy, cx = (double)(float)c.cx, cy = (double)(float)c.cy;
    const double k0 = (double)(float)c.dist[0], k1 = (double)(float)c.dist[1], k2 = (double)(float)c.dist[2], k3 = (double)(float)c.dist[3];
    const double pwx = ((double)px - cx) / fx, pwy = ((double)py - cy) / fy;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    theta_d = fmin(fmax(-kPi / 2., theta_d), kPi / 2.);
    bool converged = false;
    double theta = theta_d, scale = 0.0;
    if (fabs(theta_d) > 1e-8) {
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double a = k0 * t2, b = k1 * t4, cc = k2 * t6, d = k3 * t8;
            const double fix = (theta * (1 + a + b + cc + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * cc + 9 * d);
            theta = theta - fix;
            if (fabs(fix) < 1e-8) {
                converged = true;
                break;
            }
        }
        scale = tan(theta) / theta_d;
    }
    else converged = true;
    const bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
    if (converged && !flipped) {
        const double pux = pwx * scale, puy = pwy * scale;
        const double pr0 = fx * pux + 0. * puy + cx * 1.0, pr1 = 0. * pux + fy * puy + cy * 1.0, pr2 = 0. * pux + 0. * puy + 1. * 1.0;
        ox = (float)(pr0 / pr2);
        oy = (float)(pr1 / pr2);
    }
    else {
        ox = -1000000.0f;
        oy = -1000000.0f;
    }
}

__device__ inline void undistort_point(const svgpu_camera& c, float px, float py, float& ox, float& oy) {
    switch (c.model) {
        case SVGPU_CAM_PERSPECTIVE: cv_undistort_point(c, px, py, ox, oy); break;
        case SVGPU_CAM_FISHEYE: cv_fisheye_undistort_point(c, px, py, ox, oy); break;
        case SVGPU_CAM_RADIAL_DIVISION: {  // camera/radial_division.cc:83-98
            const double x = (px - c.cx) / c.fx, y = (py - c.cy) / c.fy;
            const double r2 = x * x + y * y;
            const double und = 1.0 + c.dist[0] * r2;
            ox = (float)((x / und) * c.fx + c.cx);
            oy = (float)((y / und) * c.fy + c.cy);
            break;
        }
        default: ox = px; oy = py; break;  // equirectangular.cc:129-131
    }
}

// camera/base.cc:130-148 + convert_keypoints_to_bearings (perspective.cc:117-122, equirectangular.cc:41-48) for ONE keypoint
struct ObsOut {
    float ux, uy;
    svgpu_keypoint undist;
    double b0, b1, b2;
};
__device__ inline void frame_obs_one(const svgpu_camera& cam, const svgpu_keypoint& kp, bool already_undistorted, ObsOut& o) {
    float ux = kp.x, uy = kp.y;
    if (!already_undistorted) undistort_point(cam, kp.x, kp.y, ux, uy);
    o.ux = ux, o.uy = uy;
    o.undist = kp;  // equirectangular: whole keypoint copied
    if (cam.model != SVGPU_CAM_EQUIRECTANGULAR) {  // camera/base.cc:130-148, perspective.cc:266-273: a fresh cv::KeyPoint
        o.undist.response = 0.f;
        o.undist.class_id = -1;
    }
    o.undist.x = ux;
    o.undist.y = uy;
    if (cam.model == SVGPU_CAM_EQUIRECTANGULAR) {  // equirectangular.cc:41-48
        // equirectangular.cc:45-46: `undist_pt.x / cols_` is a FLOAT division (float by unsigned int); "- 0.5" then promotes to double
        const double lon = ((double)(ux / (float)(unsigned)cam.cols) - 0.5) * (2.0 * kPi);
        const double lat = -((double)(uy / (float)(unsigned)cam.rows) - 0.5) * kPi;
        o.b0 = cos(lat) * sin(lon);
        o.b1 = -sin(lat);
        o.b2 = cos(lat) * cos(lon);
    }
    else {  // perspective.cc:117-122 (same text in fisheye.cc / radial_division.cc)
        const double x = (ux - cam.cx) / cam.fx, y = (uy - cam.cy) / cam.fy;
        const double l2 = sqrt(x * x + y * y + 1.0);
        o.b0 = x / l2;
        o.b1 = y / l2;
        o.b2 = 1.0 / l2;
    }
}

// data::frame::can_observe (data/frame.cc:59-85) and the per-matcher variants (frame_kernels.h) for ONE landmark.  `P` carries the
// camera, pose and modes; the landmark's own fields come as arguments so that the caller may read them from flat arrays or from a
// resident landmark table; the pose likewise (kernel argument, or the device copy a previous optimisation left behind).
struct ReprojOut {
    bool vis;
    double rx, ry;
    float xr;
    int level;
};
__device__ inline void reproject_one(const ReprojProblem& P, const double* __restrict__ rot_cw, const double* __restrict__ trans_cw,
                                     const double* __restrict__ trans_wc, bool offered, double pw0, double pw1, double pw2, double nx, double ny, double nz, float minv,
                                     float maxv, int q_level, bool has_q_level, ReprojOut& o) {
    const svgpu_camera& c = P.cam;
    bool vis = offered;
    double rx = 0.0, ry = 0.0;
    float xr = 0.f;
    int level = -1;
    if (vis) {  // camera::*::reproject_to_image
        const double* R = rot_cw;
        const double X = (R[0] * pw0 + R[1] * pw1 + R[2] * pw2) + trans_cw[0];
        const double Y = (R[3] * pw0 + R[4] * pw1 + R[5] * pw2) + trans_cw[1];
        const double Z = (R[6] * pw0 + R[7] * pw1 + R[8] * pw2) + trans_cw[2];
        if (c.model == SVGPU_CAM_EQUIRECTANGULAR) {  // equirectangular.cc:60-75
            const double nrm = sqrt((X * X + Y * Y) + Z * Z);
            const double bx = X / nrm, by = Y / nrm, bz = Z / nrm;
            const double latitude = -asin(by), longitude = atan2(bx, bz);
            rx = c.cols * (0.5 + longitude / (2.0 * kPi));
            ry = c.rows * (0.5 - latitude / kPi);
        }
        else if (Z <= 0.0) vis = false;
        else {  // perspective.cc:130-148
            const double z_inv = 1.0 / Z;
            rx = c.fx * X * z_inv + c.cx;
            ry = c.fy * Y * z_inv + c.cy;
            xr = (float)(rx - c.focal_x_baseline * z_inv);
            if (c.model == SVGPU_CAM_RADIAL_DIVISION)  // inclusive bounds, radial_division.cc:124-129
                vis = !(rx < c.min_x || rx > c.max_x) && !(ry < c.min_y || ry > c.max_y);
            else vis = c.min_x < rx && rx < c.max_x && c.min_y < ry && ry < c.max_y;
        }
    }
    if (vis && has_q_level) level = q_level;  // match_current_and_last_frames: in-image is the only visibility test
    else if (vis) {  // data/frame.cc:68-84 and the per-matcher variants (frame_kernels.h)
        double vx = pw0 - trans_wc[0], vy = pw1 - trans_wc[1], vz = pw2 - trans_wc[2];
        if (P.center_mode == 1) {  // the point in the (similarity-transformed) camera frame
            const double* R = rot_cw;
            vx = (R[0] * pw0 + R[1] * pw1 + R[2] * pw2) + trans_cw[0];
            vy = (R[3] * pw0 + R[4] * pw1 + R[5] * pw2) + trans_cw[1];
            vz = (R[6] * pw0 + R[7] * pw1 + R[8] * pw2) + trans_cw[2];
        }
        const double dist = sqrt((vx * vx + vy * vy) + vz * vz);
        const float fdist = (float)dist, far_ = (float)1.3, near_ = (float)(1.0 / 1.3);
        if (P.dist_mode == 0) {
            const float max_dist = far_ * maxv, min_dist = near_ * minv;  // landmark.h:88-92
            vis = (min_dist <= fdist && fdist <= max_dist);
        }
        else if (P.dist_mode == 1) {
            const double margin_far = 1.3, margin_near = 1.0 / margin_far;
            const double max_d = margin_far * (double)maxv, min_d = margin_near * (double)minv;
            vis = !(dist < min_d || max_d < dist);
        }
        if (vis && P.normal_mode == 0) {
            const double ray_cos = ((vx * nx + vy * ny) + vz * nz) / dist;
            vis = !(ray_cos < P.ray_cos_thr);
        }
        else if (vis && P.normal_mode == 1) {
            const double dot = (vx * nx + vy * ny) + vz * nz;
            vis = !(dot < 0.5 * dist);
        }
        if (vis) {  // landmark.cc:336-353; std::log(float): fp64 log rounded to fp32
            const float ratio = maxv / fdist;
            const float lg = (float)log((double)ratio);
            const int lvl = (int)ceilf(lg / P.log_scale_factor);
            const float nlv = (float)P.num_levels;
            if (lvl < 0) level = 0;
            else if (nlv <= (float)(unsigned)lvl) level = (int)(unsigned)(nlv - 1);
            else level = lvl;
        }
    }
    o.vis = vis;
    o.rx = rx, o.ry = ry, o.xr = xr, o.level = level;
}
// the query window of the cell matcher for a reprojected landmark (match/projection.cc:30-37, 138-157)
__device__ inline void reproject_window(const ReprojProblem& P, const ReprojOut& o, float& qx, float& qy, float& margin, int& min_level, int& max_level) {
    const int lv = o.vis ? o.level : 0;
    qx = (float)o.rx;
    qy = (float)o.ry;
    margin = P.margin * P.scale_factors[lv];
    min_level = P.window_mode == 1 ? lv : max(0, lv - 1);
    max_level = P.window_mode == 2 ? lv : (int)min(P.num_levels - 1u, (unsigned)lv + 1u);
}

}  // namespace svfd
