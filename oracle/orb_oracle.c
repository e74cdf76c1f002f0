/*
 * oracle/orb_oracle.c -- CPU restatement of stella_vslam's ORB front end.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Nothing under stella_vslam_amd/ may include, link or call
 * this file.  It is the checker for the HIP kernels (tests/, __graft_entry__.smoke()) and the
 * timed "port" CPU baseline of bench.py -- never the thing shipped.
 *
 * What it restates (reference = /root/reference/src/stella_vslam, v0.6.0):
 *   feature/orb_params.cc:41-71        scale tables by repeated fp32 multiply
 *   feature/orb_extractor.cc:28-136    extract(): orchestration and output order
 *   feature/orb_extractor.cc:153-162   compute_image_pyramid(): chained cv::resize INTER_LINEAR
 *   feature/orb_extractor.cc:164-287   compute_fast_keypoints(): 64-px cells, +6 overlap, thr retry, mask
 *   feature/orb_extractor.cc:289-329   distribute_keypoints(): grid arg-max, first wins ties
 *   feature/orb_extractor.cc:337-345   correct_keypoint_scale()
 *   feature/orb_impl.cc:51-66          u_max_ table
 *   feature/orb_impl.cc:68-91          ic_angle()
 *   feature/orb_impl.cc:93-154         compute_orb_descriptor() (scalar GET_VALUE macro :128-133)
 *   util/trigonometric.h:17-46         util::cos / util::sin polynomial
 *
 * Third-party arithmetic the reference calls but which is NOT in /root/reference (absent,
 * un-vendored): OpenCV, pinned 4.7.0 by the reference image (Dockerfile.desktop:110).  The
 * published algorithms of these primitives are restated here from the OpenCV 4.x sources:
 *   cv::resize(INTER_LINEAR, 8UC1)      imgproc/src/resize.cpp  (fixed point, 11-bit coefficients)
 *   cv::FAST(thr, nms=true) TYPE_9_16   features2d/src/fast.cpp, fast_score.cpp
 *   cv::GaussianBlur(7x7, sigma 2) 8UC1 imgproc/src/smooth.dispatch.cpp (Q8.8 fixed-point kernel,
 *                                       error-diffusion rounding), fixedpoint smoothing
 *   cv::fastAtan2                       core/src/mathfuncs_core.simd.hpp (atan_f32)
 *   cvRound / cvFloor                   core/fast_math.hpp (round-half-even / floor)
 *
 * PARITY STATUS.  Pinned against the reference's own compiled code (oracle/ref_local: feature/orb_extractor.cc, orb_impl.cc,
 * orb_params.cc, util/angle.cc, util/trigonometric.h and the pattern table compiled where they lie; tests/test_ref_local.py): everything
 * in this file that restates the REFERENCE -- the extractor's control flow, cell layout, retry, masks, grid selection, orientation
 * moments, rotated BRIEF and bit order, scale correction, util::cos / sin, the scale tables -- equals it bit for bit.
 * Pinned by the reference's own tests: the scale tables (test/stella_vslam/feature/orb_params.cc:27-70), util::cos/sin within 1e-3
 * (test/stella_vslam/util/trigonometric.cc:8-20), the structural extractor invariants (test/stella_vslam/feature/orb_extractor.cc).
 * "Parity unpinned": the five OpenCV primitives listed above (resize, GaussianBlur, FAST, fastAtan2, cvRound) -- OpenCV cannot be
 * built here; they are the same code on both sides of the comparison above.  oracle/ref_recipe/ pins them where OpenCV exists.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared  (x86-64 baseline, no FMA: mirrors the
 * reference default build, CMakeLists.txt:75-81).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_MAX_LEVELS 16

typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint; /* cv::KeyPoint layout, 28 bytes */

static const int8_t k_pattern[1024] = {
#include "orb_pattern_i8.inc"
};
const int8_t* orc_orb_pattern(void) { return k_pattern; } /* 256 x (x0, y0, x1, y1): feature/orb_point_pairs.h:47 */

/* ---------------------------------------------------------------- small OpenCV primitives */

/* cvRound(float/double): SSE cvtss2si under the default rounding mode = round half to even. */
static inline int orc_cvround_f(float v) { return (int)lrintf(v); }
static inline int orc_cvround_d(double v) { return (int)lrint(v); }
static inline int orc_cvfloor_f(float v) {
    int i = (int)v;
    return i - (i > v);
}

/* cv::fastAtan2(y, x) in degrees; core/src/mathfuncs_core.simd.hpp atan_f32 (baseline build). */
float orc_fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* util/trigonometric.h:17-46 */
static inline float orc__cos(float v) {
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}
float orc_util_cos(float v) {
    const float PI_ = 3.14159265358979f;
    const float PI_2 = PI_ / 2.0f;
    const float TWO_PI = 2.0f * PI_;
    const float INV_TWO_PI = 1.0f / TWO_PI;
    const float THREE_PI_2 = 3.0f * PI_2;
    v = v - orc_cvfloor_f(v * INV_TWO_PI) * TWO_PI;
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return orc__cos(v);
    else if (v < PI_) return -orc__cos(PI_ - v);
    else if (v < THREE_PI_2) return -orc__cos(v - PI_);
    else return orc__cos(TWO_PI - v);
}
float orc_util_sin(float v) {
    const float PI_2 = 3.14159265358979f / 2.0f;
    return orc_util_cos(PI_2 - v);
}

/* ---------------------------------------------------------------- orb_params.cc:41-71 */
void orc_orb_scale_tables(float scale_factor, int num_levels, float* scale_factors, float* inv_scale_factors,
                          float* level_sigma_sq, float* inv_level_sigma_sq) {
    float s = 1.0f, inv = 1.0f;
    for (int l = 0; l < num_levels; ++l) {
        if (l > 0) {
            s = scale_factor * s;
            inv = (1.0f / scale_factor) * inv;
        }
        scale_factors[l] = s;
        inv_scale_factors[l] = inv;
        level_sigma_sq[l] = (l == 0) ? 1.0f : s * s;
        inv_level_sigma_sq[l] = (l == 0) ? 1.0f : 1.0f / (s * s);
    }
}

/* level sizes: orb_extractor.cc:157-159, std::round(cols * 1.0 / (double)scale_factors[l]) */
void orc_level_size(int w0, int h0, float scale, int* w, int* h) {
    const double s = (double)scale;
    *w = (int)round(w0 * 1.0 / s);
    *h = (int)round(h0 * 1.0 / s);
}

/* ---------------------------------------------------------------- cv::resize INTER_LINEAR 8UC1
 * resize.cpp: scale = 1/((double)dst/src); fx=(float)((dx+0.5)*scale-0.5); sx=floor; coefficients
 * saturate_cast<short>(w*2048) (round-half-even); horizontal int32 pass; vertical
 * ((b0*(r0>>4))>>16)+((b1*(r1>>4))>>16)+2)>>2.  Border: x resets the weight, y clamps row index only. */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* alpha = (short*)malloc(sizeof(short) * 2 * dw);
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = orc_cvfloor_f(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx >= sw - 1) {
            fx = 0;
            sx = sw - 1;
        }
        xofs[dx] = sx;
        alpha[2 * dx] = (short)orc_cvround_f((1.f - fx) * 2048);
        alpha[2 * dx + 1] = (short)orc_cvround_f(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = orc_cvfloor_f(fy);
        fy -= sy;
        const short b0 = (short)orc_cvround_f((1.f - fy) * 2048);
        const short b1 = (short)orc_cvround_f(fy * 2048);
        int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* weight is 0 there */
            row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx1] * alpha[2 * dx + 1];
            row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx1] * alpha[2 * dx + 1];
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs);
    free(alpha);
    free(row0);
    free(row1);
}

/* ---------------------------------------------------------------- cv::GaussianBlur 7x7 sigma=2
 * Fixed-point branch (8UC1, non-submatrix): taps = Q8.8 conversion of the normalised Gaussian by
 * error diffusion from the edges inwards, centre tap = 256 - sum(others).  Computed (not
 * hard-coded) so that the table is checked by a test: {18,34,48,56,48,34,18}. */
void orc_gauss_taps_q8(int n, double sigma, int* taps) {
    double k[33], sum = 0;
    const int r = n / 2;
    for (int i = 0; i < n; ++i) {
        const double x = i - r;
        k[i] = exp(-0.5 * x * x / (sigma * sigma));
        sum += k[i];
    }
    double err = 0;
    int64_t isum = 0;
    for (int i = 0; i < r; ++i) {
        const double adj = k[i] / sum * 256.0 + err;
        const int v = orc_cvround_d(adj);
        err = adj - v;
        taps[i] = taps[n - 1 - i] = v;
        isum += v;
    }
    taps[r] = (int)(256 - 2 * isum);
}

static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int taps[7];
    orc_gauss_taps_q8(7, 2.0, taps);
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* S = src + (size_t)y * sstride;
        uint16_t* T = tmp + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (x >= 3 && x < w - 3) {
                T[x] = (uint16_t)(taps[0] * (S[x - 3] + S[x + 3]) + taps[1] * (S[x - 2] + S[x + 2])
                                  + taps[2] * (S[x - 1] + S[x + 1]) + taps[3] * S[x]);
                continue;
            }
            unsigned acc = 0;
            for (int k = 0; k < 7; ++k) acc += (unsigned)taps[k] * S[reflect101(x + k - 3, w)];
            T[x] = (uint16_t)acc; /* <= 256*255 = 65280, no saturation */
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t* D = dst + (size_t)y * dstride;
        const uint16_t* R[7];
        for (int k = 0; k < 7; ++k) R[k] = tmp + (size_t)reflect101(y + k - 3, h) * w;
        for (int x = 0; x < w; ++x) {
            const uint32_t acc = (uint32_t)taps[0] * ((uint32_t)R[0][x] + R[6][x]) + (uint32_t)taps[1] * ((uint32_t)R[1][x] + R[5][x])
                                 + (uint32_t)taps[2] * ((uint32_t)R[2][x] + R[4][x]) + (uint32_t)taps[3] * R[3][x];
            D[x] = (uint8_t)((acc + 32768u) >> 16);
        }
    }
    free(tmp);
}

/* ---------------------------------------------------------------- cv::FAST TYPE_9_16, nms
 * Literal restatement of features2d/src/fast.cpp FAST_t<16> (scalar path) with its three rolling
 * score rows, and fast_score.cpp cornerScore<16>.  Emission order: increasing y then x.
 * out: (x, y, score) triples; returns the number of corners (may exceed cap; only cap are stored). */
static const int k_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int N = 25;
    int d[25];
    const int v = ptr[0];
    for (int k = 0; k < N; ++k) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 8; ++m)
            if (d[k + m] < a) a = d[k + m];
        int t0 = a < d[k] ? a : d[k];
        int t1 = a < d[k + 9] ? a : d[k + 9];
        if (t0 > a0) a0 = t0;
        if (t1 > a0) a0 = t1;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 8; ++m)
            if (d[k + m] > b) b = d[k + m];
        int t0 = b > d[k] ? b : d[k];
        int t1 = b > d[k + 9] ? b : d[k + 9];
        if (t0 < b0) b0 = t0;
        if (t1 < b0) b0 = t1;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold, int nms, int* out_xys, int cap) {
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = k_circle[k][0] + k_circle[k][1] * step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    const int K = 8, N = 25;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    int n_out = 0;
    if (cols < 7 || rows < 7) return 0;
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; ++i) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    uint8_t* buf[3];
    int* cpbuf[3];
    for (int i = 0; i < 3; ++i) {
        buf[i] = (uint8_t*)calloc(cols, 1);
        cpbuf[i] = (int*)calloc(cols + 1, sizeof(int));
    }
    for (int i = 3; i < rows - 2; ++i) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3] + 1;
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; ++j, ++ptr) {
                const int v = ptr[0];
                /* high-speed test on opposite pairs (fast.cpp): a necessary condition for a 9-arc */
                const uint8_t* tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
                int is_corner = 0;
                if (d & 1) { /* darker arc: x < v - t */
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] < vt) {
                            if (++count > K) {
                                is_corner = 1;
                                break;
                            }
                        }
                        else count = 0;
                    }
                }
                if ((d & 2) && !is_corner) { /* brighter arc: x > v + t */
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] > vt) {
                            if (++count > K) {
                                is_corner = 1;
                                break;
                            }
                        }
                        else count = 0;
                    }
                }
                if (is_corner) {
                    cornerpos[ncorners++] = j;
                    if (nms) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; ++k) {
            const int j = cornerpos[k];
            const int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j]
                         && score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                if (n_out < cap) {
                    out_xys[3 * n_out] = j;
                    out_xys[3 * n_out + 1] = i - 1;
                    out_xys[3 * n_out + 2] = score;
                }
                ++n_out;
            }
        }
    }
    for (int i = 0; i < 3; ++i) {
        free(buf[i]);
        free(cpbuf[i]);
    }
    return n_out;
}

/* ---------------------------------------------------------------- orb_impl.cc */
static int g_umax[16];
static int g_umax_ready = 0;
static void init_umax(void) { /* orb_impl.cc:51-66, fast_half_patch_size_ = 15 */
    const int hp = 15;
    const int vmax = (int)floor(hp * sqrt(2.0) / 2 + 1);
    const int vmin = (int)ceil(hp * sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) g_umax[v] = (int)round(sqrt((double)hp * hp - (double)v * v));
    for (int v = hp, v0 = 0; vmin <= v; --v) {
        while (g_umax[v0] == g_umax[v0 + 1]) ++v0;
        g_umax[v] = v0;
        ++v0;
    }
    g_umax_ready = 1;
}
void orc_umax(int* out16) {
    if (!g_umax_ready) init_umax();
    memcpy(out16, g_umax, sizeof(int) * 16);
}

/* orb_impl.cc:68-91 */
float orc_ic_angle(const uint8_t* img, int step, float px, float py) {
    if (!g_umax_ready) init_umax();
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)orc_cvround_f(py) * step + orc_cvround_f(px);
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0;
        const int d = g_umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step];
            const int val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* orb_impl.cc:93-154 (scalar macro :128-133): bit k of byte i <- pair 8i+k, I(p0) < I(p1). */
void orc_compute_orb_descriptor(const uint8_t* img, int step, float px, float py, float angle_deg, uint8_t* desc) {
    const float angle = (float)(angle_deg * M_PI / 180.0);
    const float cos_angle = orc_util_cos(angle);
    const float sin_angle = orc_util_sin(angle);
    const uint8_t* center = img + (size_t)orc_cvround_f(py) * step + orc_cvround_f(px);
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            const int8_t* p = k_pattern + (size_t)(8 * i + k) * 4;
            const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            const int r0 = orc_cvround_f(x0 * sin_angle + y0 * cos_angle);
            const int c0 = orc_cvround_f(x0 * cos_angle - y0 * sin_angle);
            const int r1 = orc_cvround_f(x1 * sin_angle + y1 * cos_angle);
            const int c1 = orc_cvround_f(x1 * cos_angle - y1 * sin_angle);
            const int a = center[r0 * step + c0];
            const int b = center[r1 * step + c1];
            val |= (a < b) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- distribute_keypoints :289-329
 * in: corners (x,y region coords as float, response) in emission order; out: selected indices. */
int orc_distribute_keypoints(const float* xs, const float* ys, const float* resp, int n, int min_x, int max_x, int min_y,
                             int max_y, float scale_factor, unsigned min_area_sqrt, int* sel_idx, int* grid_dims) {
    const double scaled_min_area_sqrt = min_area_sqrt / scale_factor; /* fp32 division widened */
    const unsigned num_x_grid = (unsigned)ceil((max_x - min_x) / scaled_min_area_sqrt);
    const unsigned num_y_grid = (unsigned)ceil((max_y - min_y) / scaled_min_area_sqrt);
    const double delta_x = (double)(max_x - min_x) / num_x_grid;
    const double delta_y = (double)(max_y - min_y) / num_y_grid;
    if (grid_dims) {
        grid_dims[0] = (int)num_x_grid;
        grid_dims[1] = (int)num_y_grid;
    }
    const unsigned ncell = num_x_grid * num_y_grid;
    int* best = (int*)malloc(sizeof(int) * ncell);
    for (unsigned c = 0; c < ncell; ++c) best[c] = -1;
    for (int i = 0; i < n; ++i) {
        const unsigned ix = (unsigned)(xs[i] / delta_x);
        const unsigned iy = (unsigned)(ys[i] / delta_y);
        const unsigned idx = ix + iy * num_x_grid;
        if (idx >= ncell) continue; /* the reference would write out of bounds; cannot happen for in-range pts */
        if (best[idx] < 0) best[idx] = i;
        else if ((double)resp[i] > (double)resp[best[idx]]) best[idx] = i;
    }
    int m = 0;
    for (unsigned c = 0; c < ncell; ++c)
        if (best[c] >= 0) sel_idx[m++] = best[c];
    free(best);
    return m;
}

/* ---------------------------------------------------------------- extract() :28-136 */
static inline int mask_hit(const uint8_t* mask, int mstride, int mw, int mh, unsigned y, unsigned x, float s) {
    int my = (int)(y * s), mx = (int)(x * s);
    if (my >= mh) my = mh - 1; /* the reference indexes unchecked; clamp instead of UB */
    if (mx >= mw) mx = mw - 1;
    return mask[(size_t)my * mstride + mx] == 0;
}

/* Full extractor.  pyr_out (optional) receives levels 1..L-1 tightly packed (row stride = level
 * width), level 0 is the caller's image.  level_counts (optional) gets keypoints per level.
 * Returns 0, or -1 if more than cap keypoints were found (n_out still holds the true count). */
int orc_orb_extract(const uint8_t* img, int w, int h, int stride, const uint8_t* mask, int mask_stride,
                    float scale_factor, int num_levels, int ini_thr, int min_thr, unsigned min_area,
                    orc_keypoint* kps, uint8_t* desc, int cap, int* n_out, uint8_t* pyr_out, int* level_counts) {
    float sf[ORC_MAX_LEVELS], isf[ORC_MAX_LEVELS], ss[ORC_MAX_LEVELS], iss[ORC_MAX_LEVELS];
    if (num_levels > ORC_MAX_LEVELS) return -2;
    orc_orb_scale_tables(scale_factor, num_levels, sf, isf, ss, iss);
    const unsigned min_area_sqrt = (unsigned)sqrt((double)min_area);
    const unsigned patch_radius = 19, overlap = 6, cell_size = 64;

    /* pyramid (chained) */
    uint8_t* lev[ORC_MAX_LEVELS];
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS], ls[ORC_MAX_LEVELS];
    lev[0] = (uint8_t*)img;
    lw[0] = w;
    lh[0] = h;
    ls[0] = stride;
    size_t pyr_off = 0;
    for (int l = 1; l < num_levels; ++l) {
        orc_level_size(w, h, sf[l], &lw[l], &lh[l]);
        ls[l] = lw[l];
        lev[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        orc_resize_linear_u8(lev[l - 1], lw[l - 1], lh[l - 1], ls[l - 1], lev[l], lw[l], lh[l], ls[l]);
        if (pyr_out) {
            memcpy(pyr_out + pyr_off, lev[l], (size_t)lw[l] * lh[l]);
            pyr_off += (size_t)lw[l] * lh[l];
        }
    }

    int total = 0, overflow = 0;
    const int raw_cap = 70 * 70;
    int* cell_xys = (int*)malloc(sizeof(int) * 3 * raw_cap);
    for (int level = 0; level < num_levels; ++level) {
        const float s = sf[level];
        const unsigned min_bx = patch_radius, min_by = patch_radius;
        int n_level = 0;
        if (lw[level] > 2 * (int)patch_radius && lh[level] > 2 * (int)patch_radius) {
            const unsigned max_bx = lw[level] - patch_radius, max_by = lh[level] - patch_radius;
            const unsigned width = max_bx - min_bx, height = max_by - min_by;
            const unsigned num_cols = width / cell_size + 1, num_rows = height / cell_size + 1;
            size_t rcap = 1024, rn = 0;
            float* rx = (float*)malloc(sizeof(float) * rcap);
            float* ry = (float*)malloc(sizeof(float) * rcap);
            float* rr = (float*)malloc(sizeof(float) * rcap);
            for (unsigned i = 0; i < num_rows; ++i) {
                const unsigned min_y = min_by + i * cell_size;
                if (max_by - overlap <= min_y) continue;
                unsigned max_y = min_y + cell_size + overlap;
                if (max_by < max_y) max_y = max_by;
                for (unsigned j = 0; j < num_cols; ++j) {
                    const unsigned min_x = min_bx + j * cell_size;
                    if (max_bx - overlap <= min_x) continue;
                    unsigned max_x = min_x + cell_size + overlap;
                    if (max_bx < max_x) max_x = max_bx;
                    if (mask) {
                        if (mask_hit(mask, mask_stride, w, h, min_y, min_x, s) || mask_hit(mask, mask_stride, w, h, max_y, min_x, s)
                            || mask_hit(mask, mask_stride, w, h, min_y, max_x, s)
                            || mask_hit(mask, mask_stride, w, h, max_y, max_x, s))
                            continue;
                    }
                    const uint8_t* roi = lev[level] + (size_t)min_y * ls[level] + min_x;
                    int nc = orc_fast9_16(roi, max_x - min_x, max_y - min_y, ls[level], ini_thr, 1, cell_xys, raw_cap);
                    if (nc == 0) nc = orc_fast9_16(roi, max_x - min_x, max_y - min_y, ls[level], min_thr, 1, cell_xys, raw_cap);
                    for (int k = 0; k < nc; ++k) {
                        const float kx = (float)cell_xys[3 * k] + (float)(j * cell_size);
                        const float ky = (float)cell_xys[3 * k + 1] + (float)(i * cell_size);
                        if (mask && mask_hit(mask, mask_stride, w, h, (unsigned)(min_by + ky), (unsigned)(min_bx + kx), s)) continue;
                        if (rn == rcap) {
                            rcap *= 2;
                            rx = (float*)realloc(rx, sizeof(float) * rcap);
                            ry = (float*)realloc(ry, sizeof(float) * rcap);
                            rr = (float*)realloc(rr, sizeof(float) * rcap);
                        }
                        rx[rn] = kx;
                        ry[rn] = ky;
                        rr[rn] = (float)cell_xys[3 * k + 2];
                        ++rn;
                    }
                }
            }
            int* sel = (int*)malloc(sizeof(int) * (rn + 1));
            const int m = orc_distribute_keypoints(rx, ry, rr, (int)rn, min_bx, max_bx, min_by, max_by, s, min_area_sqrt, sel, NULL);
            const unsigned scaled_patch_size = (unsigned)(31 * s);
            for (int q = 0; q < m; ++q) {
                if (total + n_level >= cap) {
                    overflow = 1;
                    ++n_level;
                    continue;
                }
                orc_keypoint* kp = &kps[total + n_level];
                kp->x = rx[sel[q]] + (float)min_bx;
                kp->y = ry[sel[q]] + (float)min_by;
                kp->response = rr[sel[q]];
                kp->size = (float)scaled_patch_size;
                kp->octave = level;
                kp->class_id = -1;
                kp->angle = orc_ic_angle(lev[level], ls[level], kp->x, kp->y);
                ++n_level;
            }
            free(sel);
            free(rx);
            free(ry);
            free(rr);
        }
        if (level_counts) level_counts[level] = n_level;
        /* descriptors on the blurred level, then scale correction (:96-129) */
        if (n_level > 0 && !overflow) {
            uint8_t* blurred = (uint8_t*)malloc((size_t)lw[level] * lh[level]);
            orc_gaussian_blur7_u8(lev[level], lw[level], lh[level], ls[level], blurred, lw[level]);
            for (int q = 0; q < n_level; ++q) {
                orc_keypoint* kp = &kps[total + q];
                orc_compute_orb_descriptor(blurred, lw[level], kp->x, kp->y, kp->angle, desc + (size_t)(total + q) * 32);
                if (level != 0) {
                    kp->x *= s;
                    kp->y *= s;
                }
            }
            free(blurred);
        }
        total += n_level;
    }
    free(cell_xys);
    for (int l = 1; l < num_levels; ++l) free(lev[l]);
    *n_out = total;
    return overflow ? -1 : 0;
}
