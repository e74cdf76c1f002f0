/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Never linked or called by the product path (stella_vslam_amd/libsvgpu.so).
 *
 * The host steps either side of extract / match:
 *   camera::*::undistort_keypoints          camera/perspective.cc:245-274, fisheye.cc:282-311, radial_division.cc:83-98,
 *                                           equirectangular.cc:129-131 (identity)
 *   camera::*::convert_keypoints_to_bearings camera/base.cc:160-164 -> perspective.cc:117-122, fisheye.cc:157-162,
 *                                           radial_division.cc:100-105, equirectangular.cc:41-48
 *   camera::*::compute_image_bounds         perspective.cc:70-96, fisheye.cc:68-134, radial_division.cc:57-81, equirectangular.cc:32-36
 *   camera::*::reproject_to_image           perspective.cc:130-148, fisheye.cc:170-188, radial_division.cc:113-132, equirectangular.cc:60-75
 *   data::frame::can_observe                data/frame.cc:59-85 with landmark::is_inside_in_orb_scale (data/landmark.h:88-92)
 *                                           and landmark::predict_scale_level (data/landmark.cc:336-353)
 *
 * Third-party arithmetic restated from the published algorithms (OpenCV 4.7.0 is not in /root/reference):
 *   cv::undistortPoints (calib3d/undistort.dispatch.cpp, cvUndistortPointsInternal) with the reference's criteria
 *   (EPS | MAX_ITER, 20, 1e-6) and cv::fisheye::undistortPoints (calib3d/fisheye.cpp) with its default criteria
 *   (MAX_ITER + EPS, 10, 1e-8).  // VERIFY-AGAINST-OPENCV-4.7
 * PARITY UNPINNED for these two: the reference's tests hold no vectors for them; the tests here pin them through the
 * forward distortion model (distort -> undistort round trip) and the zero-distortion identity.
 * Everything ELSE in this file that restates the reference's camera/*.cc (bounds, marshalling, bearings, the two reprojections, the
 * radial-division and equirectangular closed forms) is pinned against the reference's own compiled camera sources
 * (oracle/ref_local -> oracle/_ref/libsvref_cam.so, tests/test_ref_local_camera.py); orc_can_observe against the reference's data/frame.cc
 * compiled with its real frame.h / landmark.h (libsvref_frm.so, same test file).
 *
 * Pitfall restated on purpose: the reference hands OpenCV a CV_32F camera matrix and CV_32F distortion vector
 * (perspective.cc:21-22, fisheye.cc:21-22), so inside the undistortion fx, fy, cx, cy, k* are the FLOAT-rounded values,
 * while bearings / reprojection use the double members.
 * Eigen reductions of a Vector3d are taken as (x*x + y*y) + z*z (packet of two, then the tail).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int32_t model; /* camera/base.h:24-29: 0 perspective, 1 fisheye, 2 equirectangular, 3 radial_division */
    int32_t pad_;
    double cols, rows;
    double fx, fy, cx, cy;
    double dist[5]; /* perspective k1 k2 p1 p2 k3 | fisheye k1 k2 k3 k4 | radial_division distortion */
    double focal_x_baseline;
    float min_x, max_x, min_y, max_y; /* img_bounds_ */
} orc_camera;

/* Round to float and back.  Through a volatile on purpose: gcc 11 -O3 (SLP vectoriser) drops a plain (double)(float)v round
 * trip of neighbouring struct members, which silently turns the CV_32F camera matrix back into the double one. */
static double f32r(double v) {
    volatile float f = (float)v;
    return (double)f;
}

enum { CAM_PERSPECTIVE = 0, CAM_FISHEYE = 1, CAM_EQUIRECT = 2, CAM_RADIAL_DIVISION = 3 };

/* cv::undistortPoints, R = identity, P = K (float), no tilt terms (k[5..13] = 0). */
static void cv_undistort_point(const orc_camera* c, float px, float py, float* ox, float* oy) {
    const double fx = f32r(c->fx), fy = f32r(c->fy), cx = f32r(c->cx), cy = f32r(c->cy);
    const double k0 = f32r(c->dist[0]), k1 = f32r(c->dist[1]), k2 = f32r(c->dist[2]),
                 k3 = f32r(c->dist[3]), k4 = f32r(c->dist[4]);
    const double ifx = 1. / fx, ify = 1. / fy;
    const double u = px, v = py;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    double error = 1.7976931348623157e308;
    for (int j = 0;; ++j) {
        if (j >= 20) break;
        if (error < 1e-6) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
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
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}

/* cv::fisheye::undistortPoints, R empty, P = K (float). */
static void cv_fisheye_undistort_point(const orc_camera* c, float px, float py, float* ox, float* oy) {
    const double fx = f32r(c->fx), fy = f32r(c->fy), cx = f32r(c->cx), cy = f32r(c->cy);
    const double k0 = f32r(c->dist[0]), k1 = f32r(c->dist[1]), k2 = f32r(c->dist[2]), k3 = f32r(c->dist[3]);
    const double pwx = ((double)px - cx) / fx, pwy = ((double)py - cy) / fy;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    const double half_pi = 3.1415926535897932384626433832795 / 2.;
    theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
    int converged = 0;
    double theta = theta_d, scale = 0.0;
    if (fabs(theta_d) > 1e-8) {
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double a = k0 * t2, b = k1 * t4, cc = k2 * t6, d = k3 * t8;
            const double fix = (theta * (1 + a + b + cc + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * cc + 9 * d);
            theta = theta - fix;
            if (fabs(fix) < 1e-8) {
                converged = 1;
                break;
            }
        }
        scale = tan(theta) / theta_d;
    }
    else converged = 1;
    const int flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
    if (converged && !flipped) {
        const double pux = pwx * scale, puy = pwy * scale;
        const double pr0 = fx * pux + 0. * puy + cx * 1.0, pr1 = 0. * pux + fy * puy + cy * 1.0, pr2 = 0. * pux + 0. * puy + 1. * 1.0;
        *ox = (float)(pr0 / pr2);
        *oy = (float)(pr1 / pr2);
    }
    else {
        *ox = (float)-1000000.0;
        *oy = (float)-1000000.0;
    }
}

static void undistort_point(const orc_camera* c, float px, float py, float* ox, float* oy) {
    switch (c->model) {
        case CAM_PERSPECTIVE: cv_undistort_point(c, px, py, ox, oy); break;
        case CAM_FISHEYE: cv_fisheye_undistort_point(c, px, py, ox, oy); break;
        case CAM_RADIAL_DIVISION: { /* radial_division.cc:83-98 */
            const double x = (px - c->cx) / c->fx, y = (py - c->cy) / c->fy;
            const double r2 = x * x + y * y;
            const double und = 1.0 + c->dist[0] * r2;
            const double ux = x / und, uy = y / und;
            *ox = (float)(ux * c->fx + c->cx);
            *oy = (float)(uy * c->fy + c->cy);
            break;
        }
        default: *ox = px; *oy = py; break;
    }
}

/* xy: n x 2 floats in, n x 2 floats out. */
void orc_undistort_keypoints(const orc_camera* c, int n, const float* xy, float* out) {
    for (int i = 0; i < n; ++i) undistort_point(c, xy[2 * i], xy[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

void orc_keypoints_to_bearings(const orc_camera* c, int n, const float* xy, double* bearings) {
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        const float ux = xy[2 * i], uy = xy[2 * i + 1];
        double* b = bearings + 3 * i;
        if (c->model == CAM_EQUIRECT) {
            /* equirectangular.cc:45-46: `undist_pt.x / cols_` divides a float by an unsigned int, i.e. in FLOAT; only then does "- 0.5"
             * promote to double (found by pinning against the reference's compiled camera code, tests/test_ref_local_camera.py) */
            const double lon = ((double)(ux / (float)(unsigned)c->cols) - 0.5) * (2.0 * pi);
            const double lat = -((double)(uy / (float)(unsigned)c->rows) - 0.5) * pi;
            b[0] = cos(lat) * sin(lon);
            b[1] = -sin(lat);
            b[2] = cos(lat) * cos(lon);
        }
        else {
            const double x = (ux - c->cx) / c->fx, y = (uy - c->cy) / c->fy;
            const double l2 = sqrt(x * x + y * y + 1.0);
            b[0] = x / l2;
            b[1] = y / l2;
            b[2] = 1.0 / l2;
        }
    }
}

/* compute_image_bounds: min_x max_x min_y max_y (floats, as image_bounds holds them). */
void orc_image_bounds(const orc_camera* c, float* bounds) {
    const float cols = (float)c->cols, rows = (float)c->rows;
    int none = 1;
    const int nd = c->model == CAM_PERSPECTIVE ? 5 : c->model == CAM_FISHEYE ? 4 : c->model == CAM_RADIAL_DIVISION ? 1 : 0;
    for (int i = 0; i < nd; ++i) none &= (c->dist[i] == 0);
    if (none) {
        bounds[0] = 0.0f; bounds[1] = cols; bounds[2] = 0.0f; bounds[3] = rows;
        return;
    }
    float q[8], u[8];
    if (c->model == CAM_FISHEYE) {
        const double pwx = (0.0 - c->cx) / c->fx, pwy = (0.0 - c->cy) / c->fy;
        const double theta_d = sqrt(pwx * pwx + pwy * pwy);
        if (theta_d > 3.14159265358979323846 / 2) { /* fisheye.cc:83-116 */
            q[0] = (float)c->cx; q[1] = 0; q[2] = cols; q[3] = (float)c->cy; q[4] = 0; q[5] = (float)c->cy; q[6] = (float)c->cx; q[7] = rows;
            orc_undistort_keypoints(c, 4, q, u);
            const float deg_thr = 5.0f;
            const float tx = (float)(c->fx / tan(deg_thr * 3.14159265358979323846 / 180.0)), ty = (float)(c->fy / tan(deg_thr * 3.14159265358979323846 / 180.0));
            const float mnx = (float)(-tx + c->cx), mxx = (float)(tx + c->cx), mny = (float)(-ty + c->cy), mxy = (float)(ty + c->cy);
            const float a = u[4], b = u[2], cc = u[1], d = u[7];
            bounds[0] = (a < mnx || a > c->cx) ? mnx : a;
            bounds[1] = (b > mxx || b < c->cx) ? mxx : b;
            bounds[2] = (cc < mny || cc > c->cy) ? mny : cc;
            bounds[3] = (d > mxy || d < c->cy) ? mxy : d;
            return;
        }
    }
    q[0] = 0; q[1] = 0; q[2] = cols; q[3] = 0; q[4] = 0; q[5] = rows; q[6] = cols; q[7] = rows;
    orc_undistort_keypoints(c, 4, q, u);
    bounds[0] = fminf(u[0], u[4]);
    bounds[1] = fmaxf(u[2], u[6]);
    bounds[2] = fminf(u[1], u[3]);
    bounds[3] = fmaxf(u[5], u[7]);
}

/* reproject_to_image.  rot_cw row-major 3x3. */
static int reproject_to_image(const orc_camera* c, const double* R, const double* t, const double* pw, double* reproj, float* x_right) {
    const double X = (R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2]) + t[0];
    const double Y = (R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2]) + t[1];
    const double Z = (R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2]) + t[2];
    if (c->model == CAM_EQUIRECT) {
        const double pi = 3.14159265358979323846;
        const double nrm = sqrt((X * X + Y * Y) + Z * Z);
        const double bx = X / nrm, by = Y / nrm, bz = Z / nrm;
        const double latitude = -asin(by), longitude = atan2(bx, bz);
        reproj[0] = c->cols * (0.5 + longitude / (2.0 * pi));
        reproj[1] = c->rows * (0.5 - latitude / pi);
        *x_right = 0.0f;
        return 1;
    }
    if (Z <= 0.0) return 0;
    const double z_inv = 1.0 / Z;
    reproj[0] = c->fx * X * z_inv + c->cx;
    reproj[1] = c->fy * Y * z_inv + c->cy;
    *x_right = (float)(reproj[0] - c->focal_x_baseline * z_inv);
    if (c->model == CAM_RADIAL_DIVISION) /* inclusive bounds: radial_division.cc:124-129 */
        return !(reproj[0] < c->min_x || reproj[0] > c->max_x) && !(reproj[1] < c->min_y || reproj[1] > c->max_y);
    return c->min_x < reproj[0] && reproj[0] < c->max_x && c->min_y < reproj[1] && reproj[1] < c->max_y;
}

/* exported for oracle/match2_oracle.c (the projection-family matchers call camera_->reproject_to_image themselves) */
int orc_reproject_to_image(const orc_camera* c, const double* R, const double* t, const double* pw, double* reproj, float* x_right) {
    return reproject_to_image(c, R, t, pw, reproj, x_right);
}

/* frame::can_observe for n landmarks.
 *   rot_cw 9 (row-major), trans_cw 3, trans_wc 3 (camera centre in world)
 *   pos_w n x 3, mean_normal n x 3, min_valid_dist / max_valid_dist n floats
 *   visible n bytes; reproj n x 2 doubles, x_right n floats, pred_level n ints (written where visible, else 0 / -1) */
void orc_can_observe(const orc_camera* c, const double* rot_cw, const double* trans_cw, const double* trans_wc, int n,
                     const double* pos_w, const double* mean_normal, const float* min_valid_dist, const float* max_valid_dist,
                     float ray_cos_thr, unsigned num_levels, float log_scale_factor, uint8_t* visible, double* reproj, float* x_right,
                     int32_t* pred_level) {
    for (int i = 0; i < n; ++i) {
        visible[i] = 0;
        reproj[2 * i] = reproj[2 * i + 1] = 0.0;
        x_right[i] = 0.0f;
        pred_level[i] = -1;
        const double* pw = pos_w + 3 * i;
        double rp[2];
        float xr;
        if (!reproject_to_image(c, rot_cw, trans_cw, pw, rp, &xr)) continue;
        const double vx = pw[0] - trans_wc[0], vy = pw[1] - trans_wc[1], vz = pw[2] - trans_wc[2];
        const double dist = sqrt((vx * vx + vy * vy) + vz * vz);
        /* is_inside_in_orb_scale(float dist, float 1.3, float 1/1.3) */
        const float fdist = (float)dist, far_ = (float)1.3, near_ = (float)(1.0 / 1.3);
        const float max_dist = far_ * max_valid_dist[i], min_dist = near_ * min_valid_dist[i];
        if (!(min_dist <= fdist && fdist <= max_dist)) continue;
        const double* nv = mean_normal + 3 * i;
        const double ray_cos = ((vx * nv[0] + vy * nv[1]) + vz * nv[2]) / dist;
        if (ray_cos < ray_cos_thr) continue;
        /* predict_scale_level(float dist, float num_levels, float log_scale_factor) */
        const float ratio = max_valid_dist[i] / fdist;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);
        const float nlv = (float)num_levels;
        int out;
        if (lvl < 0) out = 0;
        else if (nlv <= (float)(unsigned)lvl) out = (int)(unsigned)(nlv - 1);
        else out = lvl;
        visible[i] = 1;
        reproj[2 * i] = rp[0];
        reproj[2 * i + 1] = rp[1];
        x_right[i] = xr;
        pred_level[i] = out;
    }
}

/* Forward distortion models, used by the tests only to pin the two restated OpenCV routines (distort -> undistort). */
void orc_test_distort_points(const orc_camera* c, int n, const double* xy_norm, float* pix) {
    for (int i = 0; i < n; ++i) {
        const double x = xy_norm[2 * i], y = xy_norm[2 * i + 1];
        const double fx = f32r(c->fx), fy = f32r(c->fy), cx = f32r(c->cx), cy = f32r(c->cy);
        double xd, yd;
        if (c->model == CAM_FISHEYE) {
            const double k0 = f32r(c->dist[0]), k1 = f32r(c->dist[1]), k2 = f32r(c->dist[2]), k3 = f32r(c->dist[3]);
            const double r = sqrt(x * x + y * y), th = atan(r), t2 = th * th;
            const double thd = th * (1 + t2 * (k0 + t2 * (k1 + t2 * (k2 + t2 * k3))));
            const double s = r > 1e-12 ? thd / r : 1.0;
            xd = x * s;
            yd = y * s;
        }
        else {
            const double k0 = f32r(c->dist[0]), k1 = f32r(c->dist[1]), p1 = f32r(c->dist[2]), p2 = f32r(c->dist[3]), k4 = f32r(c->dist[4]);
            const double r2 = x * x + y * y, cd = 1 + r2 * (k0 + r2 * (k1 + r2 * k4));
            xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        }
        pix[2 * i] = (float)(xd * fx + cx);
        pix[2 * i + 1] = (float)(yd * fy + cy);
    }
}
