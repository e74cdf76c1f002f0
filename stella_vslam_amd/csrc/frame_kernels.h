// Device-side frame observation and landmark reprojection: the host steps either side of
// extract / match.  camera::*::undistort_keypoints + convert_keypoints_to_bearings (system.cc:384-389) and
// data::frame::can_observe (data/frame.cc:59-85) as one-thread-per-item kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svgpu_internal.h"

struct FrameObsProblem {
    svgpu_camera cam;
    const svgpu_keypoint* kps;  // n distorted keypoints (extractor output)
    int n;
    int already_undistorted;    // 1: kps are undistorted keypoints (convert_keypoints_to_bearings only)
    svgpu_keypoint* undist;     // n: pt undistorted, angle / size / octave copied (camera/base.cc:139-148); nullable
    float* undist_xy;           // n x 2 (feeds the grid build)
    double* bearings;           // n x 3; nullable
};
void sv_launch_frame_observation(hipStream_t s, const FrameObsProblem& P);

struct ReprojProblem {
    svgpu_camera cam;
    double rot_cw[9], trans_cw[3], trans_wc[3];
    int n;
    const double* pos_w;         // n x 3
    const double* mean_normal;   // n x 3
    const float* min_valid_dist; // n
    const float* max_valid_dist; // n
    const uint8_t* skip;         // nullable: landmark not offered to can_observe (already tracked / will_be_erased)
    float ray_cos_thr;
    unsigned num_levels;
    float log_scale_factor;
    uint8_t* visible;
    double* reproj;              // n x 2
    float* x_right;
    int32_t* pred_level;
    // optional: the query arrays of projection::match_frame_and_landmarks (match/projection.cc:30-37, 57-62)
    float margin;
    float scale_factors[SV_MAX_LEVELS];
    float* q_xy;                 // nullable = no query output
    float* q_margin;
    int32_t* q_min_level;
    int32_t* q_max_level;
    // variants of the projection-family matchers (all zero = data::frame::can_observe, data/frame.cc:59-85)
    int dist_mode;               // 0: float margins 1.3f / (1/1.3)f (landmark::is_inside_in_orb_scale); 1: the double-precision test of
                                 //    match/projection.cc:233-241, 357-365, fuse.cc:52-61; 2: no distance test (projection.cc:128-135)
    int normal_mode;             // 0: ray_cos < ray_cos_thr (frame.cc:76-80); 1: dot(v, normal) < 0.5 |v| (projection.cc:370-372, fuse.cc:67-69); 2: none
    int center_mode;             // 0: |pos_w - trans_wc|; 1: |rot pos_w + trans| (match_keyframes_mutually, projection.cc:487, 555: the
                                 //    Sim3-transformed point; rot_cw then carries the scale)
    const int32_t* q_level;      // nullable: scale level given per query (octave of the last frame's keypoint, projection.cc:138) instead
                                 //    of landmark::predict_scale_level
    int window_mode;             // level window of the grid lookup: 0 [l-1, l+1]; 1 [l, l+1] (assume_forward, :141-143); 2 [l-1, l] (:145-147)
};
void sv_launch_reproject(hipStream_t s, const ReprojProblem& P);
