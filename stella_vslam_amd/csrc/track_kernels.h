// Kernel-argument structs of the tracked-frame chain (internal; svgpu_track.hip / track_kernels.hip).
#pragma once
#include <cstdint>

#include "frame_kernels.h"
#include "match_kernels.h"

// system.cc:384-395 for a frame whose keypoints the extractor has just left on the device, in ONE single-workgroup launch:
// undistort_keypoints + convert_keypoints_to_bearings + the split arrays of the matchers + assign_keypoints_to_grid.  The keypoint
// count is read from device memory: the host does not know it yet (the whole chain is enqueued before the first read-back).
struct TrackFrameProblem {
    svgpu_camera cam;
    const svgpu_keypoint* kps;  // extractor output (distorted keypoints)
    const int32_t* n_dev;       // their count
    int cap;                    // capacity of every per-keypoint array
    svgpu_keypoint* undist;
    float* xy;
    int32_t* octave;
    float* angle;
    double* bearings;
    GridProblem G;              // bounds, grid size, cell_of / cell_off / cell_items (nt ignored)
    int32_t* n_host;            // page-locked: the count, for the host
    int32_t* counter_reset;     // nullable: device word zeroed here (the matcher's list allocation counter)
    // RGB-D frames (system.cc:492-510): depth image in metres (convert_to_true_depth already applied), sampled at the DISTORTED keypoint
    const float* depth_img;     // nullable
    int depth_pitch;            // floats per row
    double focal_x_baseline;
    float* xright;              // stereo_x_right_ = undist.x - focal_x_baseline / depth, -1 where depth <= 0
    float* depth_out;           // depths_
};
void sv_launch_track_frame(svgpu_ctx* ctx, hipStream_t s, const TrackFrameProblem& P);

// Reprojection of the queries' landmarks (read from the resident table by id) + grid walk + gated Hamming distances + per-list sort in
// ONE launch, one wave per query: what k_can_observe, k_grid_walk<false>, k_scan_i32, k_grid_walk<true> and k_cand_dist do for the
// host-flattened entry points.  Lists are allocated from one atomic counter (cand_off[nq]); k_cand_replay* reads them through
// cand_off / cand_cnt.
//   mode 0  projection::match_current_and_last_frames (projection.cc:95-207): query q = keypoint q of the last frame, landmark q_ids[q],
//           level = that keypoint's octave, orientation gate against its angle
//   mode 1  frame::can_observe (data/frame.cc:59-85) + projection::match_frame_and_landmarks (projection.cc:13-93): query q = local landmark q
struct TrackCandProblem {
    ReprojProblem R;            // camera, variant switches, margin, scale factors; the pose fields are used unless pose_dev is set
    const double* pose_dev;     // nullable: [R|t] row-major 3x4 in device memory
    int mode;
    int nq;
    const int32_t* q_ids;       // nq landmark ids (-1: nothing to offer)
    const void* map;            // svgpu_landmark_record table
    int map_cap;
    const int32_t* q_octave;    // mode 0: last frame's keypoint octaves / angles
    const float* q_angle;
    int check_orientation;
    unsigned thr;               // the matcher's distance threshold and ratio (mode 1): candidates that can take part in no verdict are not listed
    float lowe_ratio;
    // the frame the queries are matched into (resident observation)
    const uint32_t* tdesc;
    const float* t_xy;
    const int32_t* t_octave;
    const float* t_angle;
    const float* t_xright;      // nullable
    const int32_t* cell_off;
    const int32_t* cell_items;
    float min_x, min_y;
    double inv_w, inv_h;
    int cols, rows;
    // mode 1: which keypoints already hold a landmark with observations (projection.cc:52-55)
    const int32_t* cur_lm;      // nullable
    const int32_t* nt_dev;      // nullable
    int nt;
    uint8_t* occupied;          // nt
    // outputs
    int32_t* cand_off;          // nq
    int32_t* counter;           // allocation counter of the lists beyond their slot (zero on entry; a word of its own: the two halves of a frame have different nq)
    int32_t* cand_cnt;          // nq
    uint32_t* dist;             // nq * TRACK_SLOT + cap entries: (distance << 22) | keypoint, 0xFFFFFFFF = gated out; lists up to 1 024 entries sorted
    int cap;                    // entries behind the slots
    uint8_t* q_valid;           // nq
    uint8_t* q_blocks;          // nq: the landmark has observations (an accepted match then closes its keypoint for later queries)
    uint8_t* visible;           // mode 1, nq
    double* reproj;             // mode 1, nq x 2
    float* x_right;             // mode 1
    int32_t* pred_level;        // mode 1
    uint8_t* visible_host;      // nullable, page-locked
};
#define TRACK_SLOT 64  // entries of a query's own list slot (longer lists are appended behind the slots)
void sv_launch_track_cand(svgpu_ctx* ctx, hipStream_t s, const TrackCandProblem& P);
