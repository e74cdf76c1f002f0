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
};
void sv_launch_reproject(hipStream_t s, const ReprojProblem& P);
