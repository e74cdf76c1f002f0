/*
 * svgpu.h -- C ABI of the MI355X-native stella_vslam hot path (ORB front end, Hamming matchers,
 * local bundle adjustment).  Plain pointers and sizes only; no C++/torch/OpenCV types.
 *
 * The reference (stella-cv/stella_vslam v0.6.0) has no FFI layer: its boundary for this path is three
 * C++ class surfaces.  Every entry point below names the reference interface it stands behind; the
 * C++ adaptor classes in stella_vslam_amd/host/ (same names and signatures as the reference classes)
 * flatten the object graph, call this ABI and write the results back.  INTEGRATION.md shows the
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns an svgpu_status (0 = OK); nothing throws, nothing aborts
 *   - "host" pointers are ordinary CPU memory; "dev" pointers are HIP device memory on the context's GPU
 *   - `stream` arguments are a hipStream_t passed as void* (NULL = the context's own stream)
 *   - one svgpu_ctx per (device, caller thread): two extractor instances (stereo: system.cc:427-434)
 *     use two contexts and run concurrently
 *   - all device entry points are asynchronous on their stream unless stated otherwise
 *   - a context owns ONE set of scratch buffers (pyramid, score maps, candidate lists, BA arenas), reused by every call without a
 *     fence of its own: all calls on one context must be ordered on ONE stream at a time (the context's, or one caller stream).
 *     To move a context to another stream, synchronise the old one first; for concurrent streams use one context per stream
 *     (bench.py: an extraction context and a matching context, ordered by events on the buffers they hand over)
 */
#ifndef SVGPU_H
#define SVGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGPU_ABI_VERSION 1

typedef enum svgpu_status {
    SVGPU_OK = 0,
    SVGPU_ERR_INVALID = 1,     /* bad argument */
    SVGPU_ERR_HIP = 2,         /* a HIP runtime call failed; svgpu_last_error() has the text */
    SVGPU_ERR_CAPACITY = 3,    /* an output did not fit; counts are still the true totals */
    SVGPU_ERR_NOT_CONFIGURED = 4,
    SVGPU_ERR_NO_DEVICE = 5,
    SVGPU_ERR_NUMERIC = 6,     /* e.g. reduced camera system not positive definite at every damping */
    SVGPU_STOPPED = 7          /* the caller's stop flag was set before any work was done */
} svgpu_status;

typedef struct svgpu_ctx svgpu_ctx;

/* cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct svgpu_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} svgpu_keypoint;

/* ------------------------------------------------------------------------------------------------ core */
int svgpu_abi_version(void);
int svgpu_device_count(void);
int svgpu_create(int device, svgpu_ctx** out);
/* Same with a stream priority hint: > 0 highest, < 0 lowest, 0 default.  A pipeline that runs extraction and matching of
 * consecutive batches on two streams gives the longer leg (extraction) the higher priority (bench.py). */
int svgpu_create_with_priority(int device, int priority, svgpu_ctx** out);
void svgpu_destroy(svgpu_ctx* ctx);
const char* svgpu_last_error(const svgpu_ctx* ctx);
const char* svgpu_status_string(int status);
int svgpu_synchronize(svgpu_ctx* ctx);
void* svgpu_stream(svgpu_ctx* ctx); /* the context's hipStream_t */

/* Per-kernel timing with HIP events recorded on the launch stream (measurement aid for bench.py's roofline
 * object).  svgpu_profile_select brackets every later launch of the named kernel class (NULL = off, "*" = every class: one timed
 * region then yields each kernel's mean launch time as it runs inside the caller's pipeline);
 * svgpu_profile_read synchronises and returns the accumulated device time and the number of launches of the selected class,
 * svgpu_profile_read_class the same for a named class (the way to read a "*" selection). */
const char* svgpu_profile_kernels(void); /* comma-separated class names */
int svgpu_profile_select(svgpu_ctx* ctx, const char* kernel_name);
int svgpu_profile_read(svgpu_ctx* ctx, double* total_ms, long long* launches);
int svgpu_profile_read_class(svgpu_ctx* ctx, const char* kernel_name, double* total_ms, long long* launches);
/* int8 multiply-add operations the matrix cores executed in the profiled launches of the brute-force distance kernel since the
 * last svgpu_profile_select (counted by the kernel itself: 2 x 64 x 32 x 256 per multiplied patch); 0 when it was not profiled */
int svgpu_profile_mfma_ops(svgpu_ctx* ctx, unsigned long long* int8_ops);

/* ------------------------------------------------------------------------------------------------ ORB front end
 * Stands behind  stella_vslam::feature::orb_extractor  (feature/orb_extractor.h:46-71):
 *   orb_extractor(const orb_params*, unsigned min_area, descriptor_type, mask_rects)   -> svgpu_orb_configure
 *   void extract(in_image, in_image_mask, std::vector<cv::KeyPoint>&, out_descriptors) -> svgpu_orb_extract
 *   public member image_pyramid_ (read by match::stereo, match/stereo.cc:20-114)       -> svgpu_orb_pyramid_download
 * and behind orb_params' scale tables (feature/orb_params.cc:41-71)                    -> svgpu_orb_scale_tables
 */

/* orb_params::calc_* : four tables of num_levels floats, computed by the reference's fp32 recurrences. */
int svgpu_orb_scale_tables(float scale_factor, int num_levels, float* scale_factors, float* inv_scale_factors,
                           float* level_sigma_sq, float* inv_level_sigma_sq);

/* Fix the image geometry and ORB parameters; (re)allocates device workspaces for up to max_batch frames
 * per launch.  min_area is the constructor argument (system.cc:95 default 800); its integer square
 * root is taken as in orb_extractor.cc:20. */
int svgpu_orb_configure(svgpu_ctx* ctx, int width, int height, int max_batch, float scale_factor, int num_levels,
                        int ini_fast_thr, int min_fast_thr, unsigned min_area);

/* Pipeline scheduling aid for callers that run other work beside a batch extraction on a second stream: `stream` waits until the LAST
 * svgpu_orb_extract_batch_device enqueued on `ctx` has reached a stage -- 0: pyramid and blur done, FAST about to start; 1: FAST and the
 * selection done, the descriptor kernel about to start.  bench.py holds the matcher of batch t back until the extraction of batch t+1 is
 * at its descriptor kernel: that kernel waits on patch fetches and leaves issue slots to the matcher, the pyramid does not.
 * No extraction enqueued yet: returns at once. */
int svgpu_orb_stream_wait_stage(svgpu_ctx* ctx, int stage, void* stream);

/* Upper bound on keypoints per frame for the configured geometry (= number of selection-grid cells). */
int svgpu_orb_max_keypoints(const svgpu_ctx* ctx);
/* Level geometry of the configured pyramid. */
int svgpu_orb_level_size(const svgpu_ctx* ctx, int level, int* width, int* height);

/* extract(): one frame, host in / host out, synchronous.
 *   img        8UC1, `stride` bytes per row, configured width x height
 *   mask       nullable 8UC1 of the same size (0 = masked), mask_stride bytes per row
 *   kps/desc   caller-owned, room for `cap` keypoints / cap*32 bytes
 *   n_out      number of keypoints (true total even when > cap -> SVGPU_ERR_CAPACITY)
 *   level_counts  nullable, num_levels ints
 * Output order = the reference's: level-major, selection-grid cell order within a level. */
int svgpu_orb_extract(svgpu_ctx* ctx, const uint8_t* img, int stride, const uint8_t* mask, int mask_stride,
                      svgpu_keypoint* kps, uint8_t* desc, int cap, int* n_out, int* level_counts);

/* Throughput path: `batch` frames already resident in HBM, results left in HBM, asynchronous.
 *   imgs_dev       batch frames, frame b at imgs_dev + b*frame_stride, rows `row_stride` bytes apart
 *   mask_dev       nullable; mask_frame_stride 0 = one mask shared by all frames
 *   kps_dev        batch * cap records;  desc_dev  batch * cap * 32 bytes
 *   counts_dev     batch * (1 + num_levels) int32: [total, per-level...]; total may exceed cap (then
 *                  only the first cap keypoints were written) */
int svgpu_orb_extract_batch_device(svgpu_ctx* ctx, const uint8_t* imgs_dev, int batch, size_t frame_stride,
                                   int row_stride, const uint8_t* mask_dev, size_t mask_frame_stride,
                                   int mask_row_stride, svgpu_keypoint* kps_dev, uint8_t* desc_dev, int cap,
                                   int32_t* counts_dev, void* stream);

/* The same, and the keypoint angles once more as a packed float array (batch * cap; nullable): what the angle-bin sort of the batched
 * matcher wants to read -- 4 bytes per keypoint instead of walking the 28-byte records (svgpu_match_consecutive_batch_device_angles). */
int svgpu_orb_extract_batch_device_angles(svgpu_ctx* ctx, const uint8_t* imgs_dev, int batch, size_t frame_stride,
                                          int row_stride, const uint8_t* mask_dev, size_t mask_frame_stride,
                                          int mask_row_stride, svgpu_keypoint* kps_dev, uint8_t* desc_dev, int cap,
                                          int32_t* counts_dev, float* angles_dev, void* stream);

/* image_pyramid_[level] of frame `frame` of the last extract call, copied to host (level 0 is the
 * caller's own image and is not stored).  Synchronous. */
int svgpu_orb_pyramid_download(svgpu_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride);
/* Gaussian-blurred level (the image descriptors are sampled from); debugging / tests. */
int svgpu_orb_blurred_download(svgpu_ctx* ctx, int frame, int level, uint8_t* dst, int dst_stride);

/* ------------------------------------------------------------------------------------------------ matchers
 * Stand behind  stella_vslam::match::*  (match/base.h:15-91 and the six matcher classes).
 * The reference walks frame/keyframe/landmark objects; the adaptors flatten them to descriptor rows
 * (N x 32 bytes), angles, octaves and CSR candidate lists in the reference's scan order.
 * Every matcher resolves the reference's SEQUENTIAL greedy bookkeeping exactly (same result as the
 * serial loops), see DESIGN.md "greedy replay".
 */

/* compute_descriptor_distance_32 (match/base.h:20-41) for n pairs of 32-byte rows; host in/out, synchronous. */
int svgpu_hamming_distance(svgpu_ctx* ctx, const uint8_t* a, const uint8_t* b, int n, uint32_t* dist);

/* Full n2 x n1 distance matrix (uint16) -- building block / diagnostics; host in/out, synchronous. */
int svgpu_hamming_matrix(svgpu_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, uint16_t* out);

/* robust::brute_force_match (match/robust.cc:232-328).
 *   index 1 = frame side (scanned, claimed once), index 2 = keyframe side (outer loop in index order)
 *   valid2      nullable; 0 = keyframe keypoint without a live landmark (skipped, :258-263)
 *   matched_2_in_1[n1]  keyframe index matched to each frame keypoint or -1; returns count in *num_matches
 * Host in/out, synchronous. */
int svgpu_match_bruteforce(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2,
                           const float* angle2, const uint8_t* valid2, int n2, float lowe_ratio,
                           int check_orientation, int32_t* matched_2_in_1, int* num_matches);

/* Batched, device-resident brute force: `pairs` independent (frame, keyframe) problems.
 *   desc1_dev   pairs * cap1 * 32 bytes, angle via the keypoint records (kps1_dev, pairs*cap1) written by
 *               svgpu_orb_extract_batch_device; n1_dev[p*n_stride] = count of pair p; likewise side 2
 *   valid2_dev  nullable (pairs * cap2)
 *   matched_dev pairs * cap1 int32;  num_dev pairs int32
 * Asynchronous on `stream`. */
int svgpu_match_bruteforce_batch_device(svgpu_ctx* ctx, int pairs, const uint8_t* desc1_dev,
                                        const svgpu_keypoint* kps1_dev, const int32_t* n1_dev, int cap1,
                                        const uint8_t* desc2_dev, const svgpu_keypoint* kps2_dev,
                                        const int32_t* n2_dev, int cap2, int n_stride, const uint8_t* valid2_dev,
                                        float lowe_ratio, int check_orientation, int32_t* matched_dev,
                                        int32_t* num_dev, void* stream);

/* The same matcher over a RING of `frames` extractor outputs that already sit in one device batch (the layout
 * svgpu_orb_extract_batch_device writes): pair t = robust::brute_force_match(frame (t + 1) % frames, keyframe = frame t),
 * t = 0 .. frames-1, without copying any slot.  matched_dev: frames x cap, row t indexed by the keypoints of frame (t+1) % frames. */
int svgpu_match_consecutive_batch_device(svgpu_ctx* ctx, int frames, const uint8_t* desc_dev, const svgpu_keypoint* kps_dev,
                                         const int32_t* n_dev, int cap, int n_stride, const uint8_t* valid_dev, float lowe_ratio,
                                         int check_orientation, int32_t* matched_dev, int32_t* num_dev, void* stream);

/* The same with the keypoint angles as the packed array svgpu_orb_extract_batch_device_angles wrote (kps_dev is still what the orientation
 * check of robust.cc:283-287 is defined on: the two hold the same values; angles_dev == NULL reads the records). */
int svgpu_match_consecutive_batch_device_angles(svgpu_ctx* ctx, int frames, const uint8_t* desc_dev, const svgpu_keypoint* kps_dev,
                                                const float* angles_dev, const int32_t* n_dev, int cap, int n_stride, const uint8_t* valid_dev,
                                                float lowe_ratio, int check_orientation, int32_t* matched_dev, int32_t* num_dev, void* stream);

typedef enum svgpu_match_mode {
    SVGPU_MATCH_BEST_ONLY = 0,        /* projection::match_current_and_last_frames (match/projection.cc:95-207) */
    SVGPU_MATCH_RATIO_SAME_OCTAVE = 1,/* projection::match_frame_and_landmarks     (match/projection.cc:13-93)  */
    SVGPU_MATCH_RATIO = 2,            /* bow_tree::match_frame_and_keyframe / match_keyframes (match/bow_tree.cc:169-366) */
    SVGPU_MATCH_TRIANGULATION = 3,    /* bow_tree / robust ::match_for_triangulation (match/bow_tree.cc:11-167, robust.cc:14-146) */
    SVGPU_MATCH_AREA = 4              /* area::match_in_consistent_area (match/area.cc:8-98): ratio test; a closer later query takes
                                         the target from its holder (occupied / stereo inputs are not used) */
} svgpu_match_mode;

/* Candidate-list matcher: query q scans targets cand_idx[cand_off[q] .. cand_off[q+1]) in order.
 *   cand_skip    nullable, one byte per CSR entry: non-zero drops that (query, target) pair -- gates the caller evaluated
 *                on its object graph (epipolar constraint base.h:67-79, chi-square reprojection gate fuse.cc:92-119,
 *                "keypoint already has a landmark" bow_tree.cc:72-76)
 *   q_valid      nullable, 0 = query skipped
 *   occupied     nullable nt, 1 = target already holds an observed landmark (projection.cc:52-55)
 *   q_angle/t_angle + check_orientation : |angle::diff| > 30 gate (projection.cc:179-181)
 *   q_xright/t_xright/q_xr_tol : nullable stereo gate (projection.cc:57-62)
 *   thr, lowe_ratio, mode : acceptance rule
 *   match_q[nq]  target index or -1.  Host in/out, synchronous. */
int svgpu_match_candidates(svgpu_ctx* ctx, const uint8_t* qdesc, int nq, const uint8_t* tdesc, const int32_t* t_octave,
                           int nt, const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* cand_skip,
                           const uint8_t* q_valid, const uint8_t* occupied, const float* q_angle, const float* t_angle, int check_orientation,
                           const float* q_xright, const float* t_xright, const float* q_xr_tol, unsigned thr,
                           float lowe_ratio, int mode, int32_t* match_q, int* num_matches);

/* The same matcher with the candidate lists built ON THE DEVICE: data::assign_keypoints_to_grid (data/common.cc:83-108)
 * bins the target keypoints, and query q gets the result of
 *     get_keypoints_in_cell(q_xy[q], q_margin[q], q_min_level[q], q_max_level[q])            (data/common.cc:127-190)
 * in the reference's order (cells column-major inside the window, keypoints of a cell in index order, level bounds < 0 =
 * unbounded, strictly inside the margin square).  This is the loop that dominates projection::* / fuse on the CPU.
 *   t_xy          nt x 2 undistorted keypoint positions; t_octave nt
 *   grid          image bounds min_x max_x min_y max_y (camera::base img_bounds) and num_grid_cols / rows (default 64 x 48)
 *   everything else as svgpu_match_candidates (no cand_skip: pair gates that need the object graph use the CSR entry point). */
int svgpu_match_in_cells(svgpu_ctx* ctx, const uint8_t* qdesc, int nq, const float* q_xy, const float* q_margin,
                         const int32_t* q_min_level, const int32_t* q_max_level, const uint8_t* q_valid, const float* q_angle,
                         const float* q_xright, const float* q_xr_tol, const uint8_t* tdesc, const float* t_xy,
                         const int32_t* t_octave, int nt, const uint8_t* occupied, const float* t_angle, const float* t_xright,
                         float min_x, float max_x, float min_y, float max_y, int grid_cols, int grid_rows,
                         int check_orientation, unsigned thr, float lowe_ratio, int mode, int32_t* match_q, int* num_matches);

/* ------------------------------------------------------------------------------ frame observation / reprojection
 * The host steps either side of extract and match, on the device so that a tracked frame
 * goes extractor -> undistort -> bearings -> grid -> landmark reprojection -> cell matcher without per-query host work. */
typedef enum svgpu_camera_model { /* camera/base.h:24-29 model_type_t, same values */
    SVGPU_CAM_PERSPECTIVE = 0,
    SVGPU_CAM_FISHEYE = 1,
    SVGPU_CAM_EQUIRECTANGULAR = 2,
    SVGPU_CAM_RADIAL_DIVISION = 3
} svgpu_camera_model;

/* camera::base + the model's parameters (camera/perspective.h:57-73, fisheye.h:51-67, radial_division.h:49-61,
 * equirectangular.h).  dist: perspective k1 k2 p1 p2 k3 | fisheye k1 k2 k3 k4 | radial_division distortion | unused.
 * min_x..max_y = img_bounds_ (camera/base.h image_bounds, floats); svgpu_camera_image_bounds fills them. */
typedef struct svgpu_camera {
    int32_t model;
    int32_t pad_;
    double cols, rows;
    double fx, fy, cx, cy;
    double dist[5];
    double focal_x_baseline;
    float min_x, max_x, min_y, max_y;
} svgpu_camera;

/* camera::*::compute_image_bounds (perspective.cc:70-96, fisheye.cc:68-134, radial_division.cc:57-81,
 * equirectangular.cc:32-36): undistorts the image corners on the device and writes cam->min_x .. max_y. */
int svgpu_camera_image_bounds(svgpu_ctx* ctx, svgpu_camera* cam);

/* data::frame_observation from the extractor's keypoints (system.cc:384-395):
 *   undist_kps   camera::*::undistort_keypoints   (cv::undistortPoints / cv::fisheye::undistortPoints restated with the
 *                reference's CV_32F camera matrix and criteria; radial_division closed form; equirectangular copy)
 *   bearings     camera::*::convert_keypoints_to_bearings, n x 3 doubles
 *   cell_off / cell_items   data::assign_keypoints_to_grid (data/common.cc:83-108) as CSR over cell = col * grid_rows + row:
 *                cell_off has grid_cols * grid_rows + 1 entries, cell_items cell_off[last] <= n keypoint indices (index order
 *                inside a cell, as the reference's push_back order)
 * Any output may be NULL.  Host in/out, synchronous. */
int svgpu_frame_observation(svgpu_ctx* ctx, const svgpu_camera* cam, const svgpu_keypoint* kps, int n, int grid_cols,
                            int grid_rows, svgpu_keypoint* undist_kps, double* bearings, int32_t* cell_off, int32_t* cell_items);

/* camera::*::convert_keypoints_to_bearings (camera/base.cc:160-164) alone, for callers that hold UNDISTORTED keypoints. */
int svgpu_keypoints_to_bearings(svgpu_ctx* ctx, const svgpu_camera* cam, const svgpu_keypoint* undist_kps, int n, double* bearings);

/* data::frame::can_observe (data/frame.cc:59-85) for n landmarks at once -- the loop of
 * tracking_module::search_local_landmarks (tracking_module.cc:554-594) and relocalizer.cc:345:
 * camera::*::reproject_to_image, landmark::is_inside_in_orb_scale (margins 1.3 and 1/1.3), the viewing-angle test
 * against ray_cos_thr and landmark::predict_scale_level (data/landmark.cc:336-353).
 *   rot_cw 9 doubles row-major, trans_cw 3, trans_wc 3 (the frame's camera centre, frame.cc:29)
 *   pos_w / mean_normal n x 3 doubles; min_valid_dist / max_valid_dist n floats
 *   skip     nullable n bytes: non-zero = landmark not offered (already tracked, will_be_erased, temporal-ratio rule)
 *   visible n bytes; reproj n x 2 doubles; x_right n floats; pred_scale_level n ints (-1 where not visible). */
int svgpu_reproject_landmarks(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw,
                              const double* trans_wc, int n, const double* pos_w, const double* mean_normal,
                              const float* min_valid_dist, const float* max_valid_dist, const uint8_t* skip, float ray_cos_thr,
                              int num_levels, float log_scale_factor, uint8_t* visible, double* reproj, float* x_right,
                              int32_t* pred_scale_level);

/* can_observe + projection::match_frame_and_landmarks (match/projection.cc:13-93) in one call, nothing returning to the
 * host in between: the visible landmarks become the queries of the cell matcher (window margin * scale_factors[pred level],
 * levels pred-1 .. pred+1, stereo gate against x_right when t_xright is given, SVGPU_MATCH_RATIO_SAME_OCTAVE rule with
 * thr = HAMMING_DIST_THR_HIGH) over the frame's keypoints binned on the device.
 *   lm_desc n x 32; scale_factors num_levels floats (orb_params::scale_factors_)
 *   frame side: tdesc nt x 32, t_xy nt x 2 undistorted positions, t_octave nt, occupied nullable (keypoint already holds an
 *   observed landmark, projection.cc:52-55), t_xright nullable (frm_obs.stereo_x_right_)
 *   match_lm n: keypoint index or -1 (the frm.add_landmark calls, in landmark order); visible / reproj / x_right /
 *   pred_scale_level as svgpu_reproject_landmarks (nullable). */
int svgpu_match_frame_and_landmarks(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw,
                                    const double* trans_wc, int n, const double* pos_w, const double* mean_normal,
                                    const float* min_valid_dist, const float* max_valid_dist, const uint8_t* skip,
                                    const uint8_t* lm_desc, float ray_cos_thr, int num_levels, const float* scale_factors,
                                    float log_scale_factor, float margin, const uint8_t* tdesc, const float* t_xy,
                                    const int32_t* t_octave, int nt, const uint8_t* occupied, const float* t_xright,
                                    int grid_cols, int grid_rows, unsigned thr, float lowe_ratio, int32_t* match_lm,
                                    int* num_matches, uint8_t* visible, double* reproj, float* x_right,
                                    int32_t* pred_scale_level);

/* ------------------------------------------------------------------------------ device-resident frame observations
 * data::frame_observation (data/frame_observation.h:12-38) of a frame or keyframe kept ON THE DEVICE: descriptors, undistorted keypoints,
 * stereo x_right, bearings and the keypoint grid of data::assign_keypoints_to_grid.  A tracked frame (tracking_module.cc:533-608) goes
 * extract -> match_current_and_last_frames -> pose optimizer -> match_frame_and_landmarks -> pose optimizer; with a resident frame its
 * descriptors cross PCIe once (down, for the host-side data::frame) and are never uploaded again.
 *   svgpu_frame_adopt_extraction  what system.cc:384-395 does after extract(): undistort_keypoints, convert_keypoints_to_bearings,
 *                                 assign_keypoints_to_grid -- on the keypoints / descriptors the context's LAST svgpu_orb_extract left on
 *                                 the device; undist_kps / bearings (nullable) return the host copies data::frame_observation holds
 *   svgpu_frame_upload            the same for an observation that exists on the host (keyframes of the map): undistorted keypoint
 *                                 records, descriptors, stereo x_right (nullable)
 *   svgpu_frame_set_stereo        stereo_x_right_ once match::stereo has produced it (NULL: monocular again)
 *   svgpu_frame_bind              ONE-SHOT: the next call of svgpu_match_in_cells / svgpu_match_frame_and_landmarks /
 *                                 svgpu_match_current_and_last_frames / svgpu_match_frame_and_keyframe_projection /
 *                                 svgpu_match_by_sim3_transform / svgpu_fuse_detect_duplication on `ctx` takes its keypoint side (tdesc, t_xy,
 *                                 t_octave, t_angle, t_xright, nt, image bounds, grid) from the frame: those arguments are then ignored (pass
 *                                 NULL / 0); `occupied` stays a host array of svgpu_frame_size bytes.  A frame may be bound on any context of
 *                                 its device.
 *   svgpu_match_set_query_blocks  ONE-SHOT for the next svgpu_match_in_cells: q_blocks[q] = 0 -> an accepted query q does NOT occupy its
 *                                 keypoint for later queries (the landmark it carries has no observation: `lm && lm->has_observation()`,
 *                                 projection.cc:52-55); NULL = every accepted query does */
typedef struct svgpu_frame svgpu_frame;
int svgpu_frame_create(svgpu_ctx* ctx, svgpu_frame** out);
void svgpu_frame_destroy(svgpu_frame* frame);
int svgpu_frame_size(const svgpu_frame* frame);
int svgpu_frame_adopt_extraction(svgpu_ctx* ctx, svgpu_frame* frame, const svgpu_camera* cam, int grid_cols, int grid_rows,
                                 svgpu_keypoint* undist_kps, double* bearings);
int svgpu_frame_upload(svgpu_ctx* ctx, svgpu_frame* frame, const svgpu_camera* cam, const svgpu_keypoint* undist_kps, const uint8_t* desc,
                       const float* x_right, int n, int grid_cols, int grid_rows);
int svgpu_frame_set_stereo(svgpu_ctx* ctx, svgpu_frame* frame, const float* x_right);
int svgpu_frame_bind(svgpu_ctx* ctx, const svgpu_frame* frame);
int svgpu_match_set_query_blocks(svgpu_ctx* ctx, const uint8_t* q_blocks);

/* ------------------------------------------------------------------------------ device-resident local map + tracked-frame chain
 * What tracking_module does per image between extract() and the keyframe decision (tracking_module.cc:253-275, 333-355, 533-608;
 * module/frame_tracker.cc:22-60) reads, of every landmark, five things: pos_w_, mean_normal_, the valid-distance range, the
 * representative descriptor and whether it has observations.  svgpu_map keeps exactly those ON THE DEVICE, indexed by
 * data::landmark::id_ (ids are dense: one atomic counter hands them out, data/landmark.cc:13), and is kept current by the few
 * mutators of data::landmark; a tracked frame then hands over landmark IDS -- of the last frame's
 * keypoints, of the local map -- instead of flattening thousands of records out of the shared_ptr graph, and the two halves of the
 * per-frame chain are ONE submission each with ONE synchronisation at the end:
 *   svgpu_track_motion     [extract -> undistort / bearings / grid ->] projection::match_current_and_last_frames -> pose_optimizer
 *   svgpu_track_local_map  frame::can_observe over the local landmarks -> projection::match_frame_and_landmarks -> pose_optimizer
 * (between the two the host runs update_local_map on the landmarks the first half matched: object-graph work, stays the reference's).
 * Results are identical to the separate entry points (svgpu_match_current_and_last_frames, svgpu_pose_optimize,
 * svgpu_reproject_landmarks, svgpu_match_in_cells) fed from host-flattened arrays: tests/test_gpu_track.py. */
typedef struct svgpu_map svgpu_map;
enum {
    SVGPU_LM_PRESENT = 1,          /* the landmark exists and !will_be_erased() */
    SVGPU_LM_HAS_OBSERVATION = 2,  /* landmark::has_observation() */
    SVGPU_LM_HAS_DESCRIPTOR = 4    /* !get_descriptor().empty() */
};
typedef struct svgpu_landmark_record { /* 96 bytes */
    double pos_w[3];                   /* landmark::pos_w_ */
    double mean_normal[3];             /* landmark::mean_normal_ */
    float min_valid_dist, max_valid_dist;
    uint8_t descriptor[32];
    uint32_t flags;                    /* SVGPU_LM_* */
    uint32_t reserved;
} svgpu_landmark_record;
int svgpu_map_create(svgpu_ctx* ctx, svgpu_map** out);
void svgpu_map_destroy(svgpu_map* map);
int svgpu_map_capacity(const svgpu_map* map); /* ids below this are addressable (grows with the largest id upserted) */
/* whole-record upsert of n landmarks (ids[i] -> records[i]; a later entry of the same id wins); synchronous */
int svgpu_map_upsert(svgpu_ctx* ctx, svgpu_map* map, int n, const uint32_t* ids, const svgpu_landmark_record* records);
/* landmark::prepare_for_erasing: flags -> 0 (ids beyond the capacity are ignored) */
int svgpu_map_erase(svgpu_ctx* ctx, svgpu_map* map, int n, const uint32_t* ids);
/* test hook: the records as the device holds them (flags 0 for ids it has never seen) */
int svgpu_map_download(svgpu_ctx* ctx, const svgpu_map* map, int n, const uint32_t* ids, svgpu_landmark_record* records);

typedef struct svgpu_tracker svgpu_tracker;
typedef struct svgpu_track_config {
    int num_levels;
    float scale_factors[16];         /* orb_params::scale_factors_ */
    float inv_level_sigma_sq[16];    /* orb_params::inv_level_sigma_sq_ */
    float log_scale_factor;          /* orb_params::log_scale_factor_ */
    int grid_cols, grid_rows;        /* frame_observation::num_grid_cols_ / rows_ */
    int is_monocular;                /* camera::setup_type_t::Monocular */
    float true_baseline;             /* camera::base::true_baseline_ */
    /* pose_optimizer_factory.h:18-26 (2 / 2 / 10) and the stop-flag reading of svgpu_pose_optimize */
    int po_num_trials_robust, po_num_trials, po_num_each_iter, po_reset_stop_flag_each_round;
} svgpu_track_config;
typedef struct svgpu_track_result {
    int n_keypoints;     /* of the current frame */
    int num_matches;     /* the matcher's return value */
    int num_valid;       /* pose optimizer: observations that are inliers at the end (0: fewer than 5 observations, pose unchanged) */
    int lm_iterations;
    int num_observations;/* edges the pose optimizer was given */
    int num_candidates;  /* entries of the candidate lists beyond their queries' own slots (diagnostics) */
    int replay_sweeps;   /* sweeps of the matcher's greedy replay, summed over its chunks (diagnostics) */
    int reserved;
    double pose_cw[12];  /* optimised [R|t], row-major 3x4 */
} svgpu_track_result;
/* One tracker per tracking thread: owns the chain's device and page-locked buffers.  `ctx` is the context its launches go to; for the
 * fused extraction of svgpu_track_motion it must be configured (svgpu_orb_configure). */
int svgpu_tracker_create(svgpu_ctx* ctx, svgpu_map* map, const svgpu_camera* cam, const svgpu_track_config* cfg, svgpu_tracker** out);
void svgpu_tracker_destroy(svgpu_tracker* tracker);
/* frame_tracker::motion_based_track (module/frame_tracker.cc:22-60) up to discard_outliers, as one submission.
 *   img != NULL   system::create_monocular_frame's device work comes first: ORB extraction of `img` (row stride `stride`), undistortion,
 *                 bearings and grid -> `cur` becomes the resident observation; kps / desc / undist_kps / bearings (each cap entries)
 *                 receive the host copies data::frame_observation holds (all four NULL: left in the tracker's own page-locked buffer,
 *                 svgpu_tracker_observation).  img == NULL: `cur` already holds the observation
 *                 (svgpu_frame_adopt_extraction / svgpu_frame_upload) and those four outputs are ignored.
 *   last, last_lm_ids   the last frame's resident observation and, per keypoint of it, the landmark id it holds (-1: none)
 *   pose_guess_cw / pose_last_cw   3x4 [R|t] row-major: velocity * last pose, and the last frame's pose (assume_forward / backward)
 *   match_last    svgpu_frame_size(last) entries: keypoint of `cur` the landmark of last keypoint i was matched to, or -1; the caller
 *                 replays curr_frm.add_landmark(lm, match_last[i]) in increasing i (projection.cc:202)
 *   outlier       per keypoint of `cur` (with an image: cap entries, cap >= svgpu_orb_max_keypoints covers every frame): the pose optimizer's flag
 * The caller compares result->num_matches with its threshold and, below it, calls again with img = NULL and twice the margin
 * (frame_tracker.cc:36-40): the optimisation that was enqueued behind the first matcher is then simply discarded. */
int svgpu_track_motion(svgpu_tracker* tracker, svgpu_frame* cur, const uint8_t* img, int stride, const svgpu_frame* last,
                       const int32_t* last_lm_ids, const double* pose_guess_cw, const double* pose_last_cw, float margin,
                       int check_orientation, svgpu_keypoint* kps, uint8_t* desc, svgpu_keypoint* undist_kps, double* bearings, int cap,
                       int32_t* match_last, uint8_t* outlier, svgpu_track_result* result);
/* The observation the last svgpu_track_motion with an image brought back, where the copy engine left it (page-locked memory of the tracker;
 * valid until the tracker's next call): pass NULL for kps / desc / undist_kps / bearings there and read it here without a second copy.
 * Each pointer is nullable; returns the keypoint count (0 when there is none). */
int svgpu_tracker_observation(const svgpu_tracker* tracker, const svgpu_keypoint** kps, const uint8_t** desc, const svgpu_keypoint** undist_kps,
                              const double** bearings);
/* The same for a STEREO frame (system::create_stereo_frame, system.cc:406-447, then the tracker): both images go down, the right one is
 * extracted on `ctx_right` (a second context of the tracker's device, configured like the tracker's) and its own stream beside the left
 * one's, match::stereo::compute (match/stereo.cc:20-114) runs behind both on the device-resident keypoints, descriptors and pyramids, the
 * left observation (undistortion, bearings, grid) follows, then the matcher and the optimiser with the stereo gates and edges -- still ONE
 * submission and ONE synchronisation.  The observation stays in the tracker's page-locked buffer: svgpu_tracker_observation for the
 * keypoints / descriptors / undistorted keypoints / bearings, svgpu_tracker_observation_stereo for stereo_x_right_ / depths_
 * (-1 where the left keypoint found no partner).  The tracker must have been created with is_monocular = 0. */
int svgpu_track_motion_stereo(svgpu_tracker* tracker, svgpu_ctx* ctx_right, svgpu_frame* cur, const uint8_t* img_left, int stride_left,
                              const uint8_t* img_right, int stride_right, const svgpu_frame* last, const int32_t* last_lm_ids,
                              const double* pose_guess_cw, const double* pose_last_cw, float margin, int check_orientation, int cap,
                              int32_t* match_last, uint8_t* outlier, svgpu_track_result* result);
int svgpu_tracker_observation_stereo(const svgpu_tracker* tracker, const float** stereo_x_right, const float** depths);
/* ... and for an RGB-D frame (system::create_RGBD_frame, system.cc:466-526): `depth` = the depth image in metres as CV_32F
 * (util::convert_to_true_depth already applied by the caller) with the SAME height and width as `img` (the library reads height rows of
 * width floats; it cannot check the size of the buffer behind the pointer), `depth_stride` in FLOATS per row (>= width).  The depth is sampled at the distorted
 * keypoint (img_depth.at<float>(y, x), coordinates truncated), stereo_x_right_ = undist.x - focal_x_baseline / depth (-1 where depth <= 0),
 * inside the frame-observation kernel of the same submission; read back with svgpu_tracker_observation_stereo. */
int svgpu_track_motion_rgbd(svgpu_tracker* tracker, svgpu_frame* cur, const uint8_t* img, int stride, const float* depth, int depth_stride,
                            const svgpu_frame* last, const int32_t* last_lm_ids, const double* pose_guess_cw, const double* pose_last_cw,
                            float margin, int check_orientation, int cap, int32_t* match_last, uint8_t* outlier, svgpu_track_result* result);
/* tracking_module::search_local_landmarks + optimize_current_frame_with_local_map's optimisation (tracking_module.cc:533-608, 441-446)
 * as one submission.
 *   cur_lm_ids    per keypoint of `cur`: the landmark id the frame holds now (-1: none) -- after discard_outliers and update_local_map's
 *                 clean-up, i.e. curr_frm.get_landmarks() as ids
 *   local_ids     n_local entries: landmark ids of local_landmarks_ in order; -1 = not offered to can_observe (already in the frame,
 *                 will_be_erased, temporal-ratio rule :565-580)
 *   pose_cw       nullable: NULL = the pose the tracker's last optimisation left on the device
 *   match_local   n_local: keypoint the landmark was matched to or -1 (replay frm.add_landmark in increasing order, projection.cc:88)
 *   visible       nullable, n_local: can_observe's verdict (increase_num_observable, tracking_module.cc:588)
 *   outlier       per keypoint of `cur` */
int svgpu_track_local_map(svgpu_tracker* tracker, const svgpu_frame* cur, const int32_t* cur_lm_ids, int n_local, const int32_t* local_ids,
                          const double* pose_cw, float margin, float lowe_ratio, float ray_cos_thr, int32_t* match_local, uint8_t* visible,
                          uint8_t* outlier, svgpu_track_result* result);
/* lm_to_reproj / lm_to_x_right / lm_to_scale of the last svgpu_track_local_map, fetched only when somebody wants them (each nullable) */
int svgpu_track_local_map_observability(svgpu_tracker* tracker, int n_local, double* reproj, float* x_right, int32_t* pred_scale_level);
/* diagnostics: kernel launches + runtime copies the tracker enqueued, and stream synchronisations it waited on, since creation */
int svgpu_tracker_counters(const svgpu_tracker* tracker, long long* launches, long long* host_syncs);
/* debug (SVGPU_TRACK_STAMPS set): 100 MHz wall-clock stamps of the last optimisation kernel's phases, [0] = how many follow; else NULL / zeros */
const unsigned long long* svgpu_tracker_debug_stamps(const svgpu_tracker* tracker);

/* ------------------------------------------------------------------------------ function-specific matchers
 * One entry point per reference method: candidate generation (reprojection + grid cells, or BoW buckets), the method's own pair
 * gates and the exact sequential bookkeeping all run on the device; the adaptor classes of stella_vslam_amd/host/ flatten the
 * frame / keyframe / landmark objects into these arrays and write the results back.  Common conventions:
 *   landmarks (queries)  pos_w n x 3 doubles, `valid` 0 = not offered (null, will_be_erased, already matched ...: the method's own
 *                        `continue` tests on the object graph), min/max_valid_dist (landmark::get_min/max_valid_distance),
 *                        mean_normal n x 3 (get_obs_mean_normal), lm_desc n x 32 (get_descriptor)
 *   keypoint side        tdesc nt x 32, t_xy nt x 2 undistorted positions, t_octave, t_angle, t_xright (stereo_x_right_, nullable),
 *                        binned on the device by data::assign_keypoints_to_grid over cam->min_x .. max_y and grid_cols x grid_rows
 *   scale_factors / inv_level_sigma_sq   orb_params tables of num_levels floats, log_scale_factor = orb_params::log_scale_factor_
 *   outputs              per query the matched keypoint index or -1, in query order; the adaptor replays them onto the objects
 * All host in/out, synchronous. */

/* camera::*::reproject_to_bearing (perspective.cc:150-170, fisheye.cc:189-209, equirectangular.cc:75-80, radial_division.cc:135-156):
 * used by the triangulation matchers for the epipole of keyframe 1 in keyframe 2.  Pure host function. */
int svgpu_reproject_to_bearing(const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, const double* pos_w, double* bearing,
                               int* valid);

/* projection::match_current_and_last_frames (match/projection.cc:95-207): the landmarks of the last frame's keypoints reprojected
 * with the current pose guess; level window from the keypoint's octave in the last frame and the forward / backward motion test
 * (:108-157; is_monocular = camera setup Monocular, true_baseline = camera::base::true_baseline_); gates: keypoint already holds an
 * observed landmark (`occupied`), stereo x_right, orientation; best <= HAMMING_DIST_THR_HIGH.
 *   lm_has_observation  nullable (all 1): 0 = the landmark carries no observation, so the keypoint it is attached to stays open for
 *                       later landmarks (`curr_lm && curr_lm->has_observation()`, :167-170) and a later match overwrites it
 *   match_last[i]       keypoint of the current frame matched to landmark i (curr_frm.add_landmark(lm, best_idx) in order) */
int svgpu_match_current_and_last_frames(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, const double* rot_lw,
                                        const double* trans_lw, int is_monocular, float true_baseline, int n_last, const double* pos_w,
                                        const uint8_t* valid, const uint8_t* lm_desc, const int32_t* octave_last, const float* angle_last,
                                        const uint8_t* lm_has_observation, int num_levels, const float* scale_factors, float margin,
                                        const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_angle, int nt,
                                        const uint8_t* occupied, const float* t_xright, int grid_cols, int grid_rows, int check_orientation,
                                        int32_t* match_last, int* num_matches);

/* projection::match_frame_and_keyframe (match/projection.cc:209-319; relocalisation): the keyframe's landmarks into the frame;
 * double-precision distance-range test (:233-241), landmark::predict_scale_level, window [l-1, l+1]; gates: frame keypoint already
 * holds a landmark (`occupied` = frm_landmarks non-null), orientation against the KEYFRAME keypoint's angle; best <= hamm_dist_thr.
 *   valid  = landmark non-null, not will_be_erased, not in already_matched_lms */
int svgpu_match_frame_and_keyframe_projection(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, int n_kf,
                                              const double* pos_w, const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist,
                                              const uint8_t* lm_desc, const float* angle_kf, int num_levels, const float* scale_factors,
                                              float log_scale_factor, float margin, unsigned hamm_dist_thr, const uint8_t* tdesc, const float* t_xy,
                                              const int32_t* t_octave, const float* t_angle, int nt, const uint8_t* occupied, int grid_cols,
                                              int grid_rows, int check_orientation, int32_t* match_kf, int* num_matches);

/* projection::match_by_Sim3_transform (match/projection.cc:321-416; loop detection): sim3_cw 4x4 row-major; converted to SE3 as the
 * reference does (:326-330); distance range, viewing-angle test dot(v, normal) >= 0.5 |v|, predicted level; candidates skip keypoints
 * that already hold a landmark (`occupied` = matched_lms_in_keyfrm non-null); best <= HAMMING_DIST_THR_LOW, no orientation test.
 *   valid  = landmark not will_be_erased and not already in matched_lms_in_keyfrm */
int svgpu_match_by_sim3_transform(svgpu_ctx* ctx, const svgpu_camera* cam, const double* sim3_cw, int n, const double* pos_w, const uint8_t* valid,
                                  const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc,
                                  int num_levels, const float* scale_factors, float log_scale_factor, float margin, const uint8_t* tdesc,
                                  const float* t_xy, const int32_t* t_octave, int nt, const uint8_t* occupied, int grid_cols, int grid_rows,
                                  int32_t* match_lm, int* num_matches);

/* projection::match_keyframes_mutually (match/projection.cc:418-629): landmarks of keyframe 1 into keyframe 2 through the similarity
 * (s_12, rot_12, trans_12) and vice versa, two independent passes (no bookkeeping inside a pass), then the cross-check.
 *   valid1 / valid2   landmark non-null, not will_be_erased, not already matched between the two keyframes (:441-450, 466-468)
 *   desc*, xy*, octave*   the keyframes' own keypoints (the side that is searched in the other pass)
 *   matched_2_in_1 / matched_1_in_2   the two passes; mutual_2_in_1[i] = idx_2 where both agree (:598-610), num_matches counts those.
 * As in the reference, both passes test "inside the image" with keyframe 2's camera (:477, 550). */
int svgpu_match_keyframes_mutually(svgpu_ctx* ctx, const svgpu_camera* cam1, const svgpu_camera* cam2, const double* rot_1w, const double* trans_1w,
                                   const double* rot_2w, const double* trans_2w, float s_12, const double* rot_12, const double* trans_12,
                                   int n1, const double* pos_w1, const uint8_t* valid1, const float* min_valid1, const float* max_valid1,
                                   const uint8_t* lm_desc1, const uint8_t* desc1, const float* xy1, const int32_t* octave1,
                                   int n2, const double* pos_w2, const uint8_t* valid2, const float* min_valid2, const float* max_valid2,
                                   const uint8_t* lm_desc2, const uint8_t* desc2, const float* xy2, const int32_t* octave2,
                                   int num_levels, const float* scale_factors, float log_scale_factor, float margin, int grid_cols, int grid_rows,
                                   int32_t* matched_2_in_1, int32_t* matched_1_in_2, int32_t* mutual_2_in_1, int* num_matches);

/* fuse::detect_duplication<T> (match/fuse.cc:11-154): landmarks_to_check reprojected into a keyframe; distance range, viewing angle,
 * predicted level; per candidate the optional chi-square reprojection gate (do_reprojection_matching: 5.99146 / 7.81473 against the
 * squared error times inv_level_sigma_sq of the keypoint's octave, 3 dof when the keypoint has a stereo x_right, :92-119) and
 * `already_matched_idx_in_keyfrm`; best <= HAMMING_DIST_THR_LOW.
 *   valid     = landmark non-null, not will_be_erased, not observed in the keyframe (:27-35)
 *   best_idx  per landmark the keyframe keypoint it fuses with, or -1; the adaptor sorts them into duplicated_lms_in_keyfrm /
 *             new_connections by keyfrm->get_landmark(best_idx) (:130-147) */
int svgpu_fuse_detect_duplication(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, int n, const double* pos_w,
                                  const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal,
                                  const uint8_t* lm_desc, int num_levels, const float* scale_factors, const float* inv_level_sigma_sq,
                                  float log_scale_factor, float margin, int do_reprojection_matching, const uint8_t* tdesc, const float* t_xy,
                                  const int32_t* t_octave, const float* t_xright, int nt, int grid_cols, int grid_rows, int32_t* best_idx,
                                  int* num_fused);

/* robust::match_for_triangulation (match/robust.cc:14-146) and bow_tree::match_for_triangulation (match/bow_tree.cc:11-167).
 * Keypoints WITHOUT a landmark on both sides (has_lm* = 1 excludes); per pair: orientation, Hamming <= 50 and <= the running best,
 * the epipole test (cos > 0.99862953475 rejected unless one of the two is a stereo keypoint, xright* >= 0) and
 * check_epipolar_constraint(bearing_1, bearing_2, E_12, ...) in fp64 (match/base.h:67-79) with the threshold
 * residual_rad_thr * scale_factors[octave1]; strict '<' updates, Lowe ratio, a keypoint of keyframe 2 is taken once.
 *   node1 / node2   both NULL: robust (all keypoints of keyframe 2 are candidates, queries in index order);
 *                   both given: bow_tree -- the bow_feat_vec_ node of every keypoint (svgpu_bow_transform's node_id, < 0 = none); the
 *                   merge-join of the two maps (:37-40, 142-153) runs on the device: queries in (node, index) order, candidates = the
 *                   keypoints of keyframe 2 in the same node, index order
 *   epipole_in_2 / valid_epipole   svgpu_reproject_to_bearing(cam2, rot_2w, trans_2w, keyfrm_1 camera centre)   (:22-27)
 *   matched_2_in_1[i]   keypoint of keyframe 2 or -1 (the reference returns the pairs sorted by i). */
int svgpu_match_for_triangulation(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, const int32_t* octave1, const double* bearings1,
                                  const uint8_t* has_lm1, const float* xright1, int n1, const uint8_t* desc2, const float* angle2,
                                  const double* bearings2, const uint8_t* has_lm2, const float* xright2, int n2, const int32_t* node1,
                                  const int32_t* node2, const double* E_12, const double* epipole_in_2, int valid_epipole, const float* scale_factors,
                                  int num_levels, float residual_rad_thr, float lowe_ratio, int check_orientation, int32_t* matched_2_in_1,
                                  int* num_matches);

/* bow_tree::match_frame_and_keyframe (match/bow_tree.cc:169-256) and bow_tree::match_keyframes (:258-366).
 * Side 1 = the keyframe whose landmarks are handed over (queries: valid1 = keypoint holds a live landmark), side 2 = the frame /
 * the other keyframe (valid2 nullable = every keypoint, or "holds a live landmark" for match_keyframes; occupied2 nullable = keypoints
 * that must not be matched from the start).  Device-side merge-join of the node ids as above; per pair orientation; best / second
 * best with strict '<', best <= HAMMING_DIST_THR_LOW, Lowe ratio; a side-2 keypoint is taken once.
 *   match_1to2[i]   side-2 keypoint matched to side-1 keypoint i, or -1. */
int svgpu_bow_match(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const int32_t* node1, int n1,
                    const uint8_t* desc2, const float* angle2, const uint8_t* valid2, const int32_t* node2, int n2, const uint8_t* occupied2,
                    float lowe_ratio, int check_orientation, int32_t* match_1to2, int* num_matches);

/* ------------------------------------------------------------------------------ landmark refresh (batched)
 * data::landmark::compute_descriptor (data/landmark.cc:199-254) for n landmarks: the representative descriptor is the
 * observation whose row of the k x k Hamming matrix has the smallest lower median (sorted index (unsigned)(0.5 (k-1)); first
 * row wins ties).  obs_off: n + 1 CSR offsets (obs_off[0] = 0, every landmark >= 1 observation, in the iteration order of the
 * reference's observations_ map with will_be_erased keyframes already dropped), obs_desc: obs_off[n] x 32 descriptor rows.
 *   best_obs[l]  index inside landmark l's list;  descriptor  n x 32. */
int svgpu_landmarks_compute_descriptor(svgpu_ctx* ctx, int n, const int32_t* obs_off, const uint8_t* obs_desc, int32_t* best_obs,
                                       uint8_t* descriptor);

/* data::landmark::update_mean_normal_and_obs_scale_variance (data/landmark.cc:256-318) for n landmarks:
 *   mean_normal = normalized( sum over observations of normalized(pos_w - keyfrm->get_trans_wc()) )   (summed in list order)
 *   max_valid_dist = |pos_w - ref_trans_wc| * ref_scale_factor ;  min_valid_dist = max_valid_dist * inv_scale_factor_last
 * obs_trans_wc: obs_off[n] x 3 doubles (camera centre of each observing keyframe); ref_trans_wc n x 3 (the landmark's reference
 * keyframe); ref_scale_factor[l] = scale_factors_[octave of its keypoint there]; inv_scale_factor_last =
 * inv_scale_factors_[num_levels - 1]. */
int svgpu_landmarks_update_geometry(svgpu_ctx* ctx, int n, const int32_t* obs_off, const double* obs_trans_wc, const double* pos_w,
                                    const double* ref_trans_wc, const float* ref_scale_factor, float inv_scale_factor_last,
                                    double* mean_normal, float* max_valid_dist, float* min_valid_dist);

/* ------------------------------------------------------------------------------ BoW transform (tree descent)
 * data::bow_vocabulary_util::compute_bow (data/bow_vocabulary.cc:18-24): per descriptor, descend the vocabulary tree choosing the
 * child with the smallest Hamming distance (first child wins ties) down to a leaf; word_id / weight = the leaf's, node_id = the
 * node passed at depth `node_level` (DBoW2's transform(.., levelsup = 4) records depth L - levelsup, clamped at 0 = root; the
 * bucket key of match::bow_tree).  The binding accumulates bow_vec (word -> weight, then the vocabulary's normalisation) and
 * bow_feat_vec (node -> feature indices in order) from these arrays.
 * Flat tree: node 0 = root, children of node i = children[child_off[i] .. child_off[i+1]) (none => leaf), ids strictly > 0;
 * node_desc n_nodes x 32, node_weight / word_id per node.  The vocabulary stays resident on ctx's device until freed. */
typedef struct svgpu_vocabulary svgpu_vocabulary;
int svgpu_bow_vocabulary_upload(svgpu_ctx* ctx, int n_nodes, const int32_t* child_off, const int32_t* children, const uint8_t* node_desc,
                                const float* node_weight, const int32_t* word_id, svgpu_vocabulary** out);
void svgpu_bow_vocabulary_free(svgpu_vocabulary* vocab);
int svgpu_bow_transform(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int node_level, int32_t* word_id,
                        float* weight, int32_t* node_id);
/* The reference's DEFAULT BoW build: fbow::Vocabulary::transform(features, level = 4, bow_vec, bow_feat_vec) (data/bow_vocabulary.cc:20-22; FBoW =
 * stella-cv/FBoW, restated from its published sources -- the submodule is empty in the reference checkout, so parity is unpinned).  Same
 * descent; differences to the DBoW2 form above: `store_level` counts DOWN from the root (DBoW2's levelsup counts up from the leaves), the
 * bow_feat_vec key is FBoW's PATH CODE of the node ((code << ceil(log2 k)) | child index per level), and a leaf met above the store level
 * files its feature under the code of the block it was found in.  k = the vocabulary's branching factor. */
int svgpu_fbow_transform(svgpu_ctx* ctx, const svgpu_vocabulary* vocab, const uint8_t* desc, int n, int store_level, int k,
                         int32_t* word_id, float* weight, uint32_t* node_code);

/* match::stereo::compute (match/stereo.cc:20-114): for every left keypoint the closest right keypoint in its row band
 * (rows +-2*scale, octave +-1, disparity in [0, focal_x_baseline / true_baseline], Hamming < 75), then the 11x11 L1 patch
 * slide (+-5 px) on the keypoint's pyramid level with parabolic sub-pixel refinement, finally the 2x-median correlation
 * filter.  The two pyramids are those of the LAST extract call on ctx_left / ctx_right (both contexts on one device;
 * the reference reads extractor_left_->image_pyramid_ / extractor_right_->image_pyramid_, system.cc:443-447).
 * stereo_x_right / depths: n_left floats, -1 where no match.  Host in/out, synchronous. */
int svgpu_stereo_match(svgpu_ctx* ctx_left, svgpu_ctx* ctx_right, const svgpu_keypoint* kps_left, const uint8_t* desc_left,
                       int n_left, const svgpu_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                       float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths);

/* The same for `pairs` stereo pairs at once, everything resident in HBM (BASELINE config 4: KITTI stereo batches): the keypoints /
 * descriptors / counts are the outputs of svgpu_orb_extract_batch_device on ctx_left and ctx_right (pair p at p * cap records,
 * count at n_*_dev[p * n_stride]), the pyramids those calls left in the two contexts.  The 2 x median correlation filter runs on the
 * device as well (rank-size/2 value by bisection).  stereo_x_right_dev / depths_dev: pairs * cap floats (-1 = no match).
 * Asynchronous on `stream` (NULL = ctx_left's); the caller orders it behind both extractions. */
int svgpu_stereo_match_batch_device(svgpu_ctx* ctx_left, svgpu_ctx* ctx_right, int pairs, const svgpu_keypoint* kps_left_dev, const uint8_t* desc_left_dev,
                                    const int32_t* n_left_dev, const svgpu_keypoint* kps_right_dev, const uint8_t* desc_right_dev,
                                    const int32_t* n_right_dev, int cap, int n_stride, float focal_x_baseline, float true_baseline,
                                    float* stereo_x_right_dev, float* depths_dev, void* stream);

/* ------------------------------------------------------------------------------------------------ local BA
 * Stands behind  stella_vslam::optimize::local_bundle_adjuster::optimize(map_db, curr_keyfrm, force_stop_flag)
 * (optimize/local_bundle_adjuster.h:23; g2o implementation optimize/local_bundle_adjuster_g2o.cc:36-431).
 * The adaptor performs steps 1 (gather) and 7-8 (outlier list, write-back under mtx_database_) on the
 * host exactly as the g2o/gtsam backends do and hands steps 2-6 to this call.
 */
typedef struct svgpu_ba_problem {
    int32_t num_poses;               /* local + fixed keyframes */
    int32_t num_points;              /* local landmarks (+ marker corners) */
    int32_t num_obs;                 /* reprojection edges */
    const double* pose_cw;           /* num_poses x 12: rows of [R|t] (3x4, row-major), world -> camera */
    const uint8_t* pose_fixed;       /* num_poses: 1 = fixed keyframe vertex */
    const double* points;            /* num_points x 3 */
    const uint8_t* point_fixed;      /* nullable num_points: 1 = fixed vertex (kept-fixed marker corners) */
    const int32_t* obs_pose;         /* num_obs */
    const int32_t* obs_point;        /* num_obs */
    const float* obs_uvr;            /* num_obs x 3: undistorted u, v, u_right (< 0 => monocular edge) */
    const float* obs_inv_sigma_sq;   /* num_obs: orb_params::inv_level_sigma_sq_[octave] */
    const float* obs_huber_delta;    /* num_obs: sqrt(chi_sq) of the Huber kernel, <= 0 => no kernel; < 0 => a marker-corner edge (no kernel, and
                                      * outside the chi-square / depth gate and the outlier list: local_bundle_adjuster_g2o.cc:246-304) */
    const double* intrinsics;        /* num_poses x 5: fx fy cx cy focal_x_baseline (perspective / fisheye / radial-division cameras:
                                        all use the perspective edge on undistorted keypoints, reproj_edge_wrapper.h:64-201);
                                        {0, 0, cols, rows, 0} selects the equirectangular edge (equirectangular_reproj_edge.h:64-134,
                                        monocular only, no depth gate) */
    int32_t num_first_iter;          /* 5  (local_bundle_adjuster_factory.h) */
    int32_t num_second_iter;         /* 10 */
    double gain_threshold;           /* terminate_action gain, 1e-3 */
} svgpu_ba_problem;

typedef struct svgpu_ba_stats {
    double chi2_initial, chi2_final;
    int32_t iters_stage1, iters_stage2, stage2_entered, num_gated;
    int32_t lm_trials, cholesky_failures;
    double lambda_final;
    int32_t stopped_by_terminate_action; /* the gain rule (not the caller) raised the stop flag (global_bundle_adjuster.cc:341) */
    int32_t pcg_iterations;              /* total PCG iterations over all damping trials (0 with the Cholesky solvers) */
} svgpu_ba_stats;

/* Linear solver of the reduced camera system (BlockSolver_6_3 + LinearSolverEigen / LinearSolverCSparse in the reference,
 * optimize/local_bundle_adjuster_g2o.cc:151-164, optimize/global_bundle_adjuster.cc:66-85):
 *   AUTO      dense LL^T in one workgroup's LDS while it fits (6 * free poses <= ~135), the LDS-resident PCG while THAT fits, the
 *             block envelope Cholesky beyond (the one-launch-per-iteration PCG only when the envelope is too large)
 *   PCG       block-Jacobi PCG on the block-sparse Schur complement: inside ONE workgroup (blocks, block rows, vectors all in
 *             its LDS) while the kept 6x6 blocks fit ~150 KB and 6 * free poses <= 512, else one kernel launch per iteration
 *   CHOLESKY  the LDS LL^T (larger systems fall back to PCG): the register-tile LDL^T with 3x3 pivots
 *   CHOLESKY_MFMA  the same system by a blocked right-looking LL^T, 16-column panels, the trailing update on v_mfma_f64_16x16x4_f64
 *             (DESIGN section 5 has the measured comparison of the two)
 *   DENSE     dense image in global memory + the library's own one-workgroup LL^T (no vendor solver is loaded anywhere)
 *   ENVELOPE  direct block envelope (skyline) LL^T after a reverse Cuthill-McKee ordering of the keyframe graph -- what AUTO takes beyond
 *             the on-chip solvers while the envelope stays under 256 MB (the reference factors this system with a sparse Cholesky)
 * pcg_tolerance: relative residual |r| / |g| (<= 0: 1e-10); pcg_max_iterations <= 0: max(2000, 4 n).  A solve that hits the
 * cap is taken as an inexact step when the residual fell below 1e-6, else the damping trial counts as a solver failure. */
typedef enum svgpu_ba_solver {
    SVGPU_BA_SOLVER_AUTO = 0,
    SVGPU_BA_SOLVER_CHOLESKY = 1,
    SVGPU_BA_SOLVER_PCG = 2,
    SVGPU_BA_SOLVER_DENSE = 3,
    SVGPU_BA_SOLVER_PCG_MULTI = 4, /* PCG with one kernel launch per iteration even when the system would fit one workgroup's LDS */
    SVGPU_BA_SOLVER_CHOLESKY_MFMA = 7, /* CHOLESKY with the blocked LL^T whose trailing updates run on the matrix cores (up to 126 unknowns; the register-tile form beyond) */
    SVGPU_BA_SOLVER_ENVELOPE = 6   /* block envelope Cholesky at any size (falls back to the PCG when the envelope of the ordered block graph exceeds 256 MB) */
} svgpu_ba_solver;
int svgpu_ba_set_solver(svgpu_ctx* ctx, int solver, double pcg_tolerance, int pcg_max_iterations);

/* How the context's last envelope factorisation was planned (diagnostics for tests / bench.py): info[8] = {0 one-sided | 1 two-sided |
 * 2 segmented, block rows, widest column; segmented: cuts, jobs, jobs on this rank, separator rows, longest job (columns)}. */
int svgpu_ba_last_envelope_plan(svgpu_ctx* ctx, int* info);

/* Planner self-test of the segmented envelope factorisation (host arithmetic only, runs without a device; NOT a solver path: the
 * bundle adjusters never call it).  Plans the elimination of the 6x6-block SPD system {blk_ab[NB] upper blocks a <= b, Sblk NB x 36
 * row-major, g 6 nP} exactly as the global / sharded bundle adjusters do (RCM order, `cuts` separators or the planner's choice when
 * <= 0, jobs spread over `world` ranks), walks the same plan arrays and maps the kernels walk and returns the solution x.
 * info[8] = {segmented, cuts, jobs, separator rows, longest job, separator banded, widest job column, widest column of the unsegmented
 * envelope}.  Returns 0, 1 = the system is not segmented (too short / not banded), 2 = not positive definite, < 0 = bad arguments. */
int svgpu_selftest_segmented_solve(int nP, int NB, const int* blk_ab, const double* Sblk, const double* g, int cuts, int world,
                                   double* x, int* info);

/* The same walk for ONE RANK of the distributed factorisation (tests/test_distributed_cpu.py drives it over gloo): the rank eliminates only
 * the jobs it owns; `allreduce(user, host_buf, count, NULL)` sums a HOST buffer of doubles in place across the ranks (the exchange of the
 * separator contributions and of the solution, as in the sharded bundle adjusters).  Every rank returns the single-rank solution, bit for
 * bit; info[5] = jobs this rank eliminated.  allreduce == NULL: svgpu_selftest_segmented_solve. */
int svgpu_selftest_segmented_solve_rank(int nP, int NB, const int* blk_ab, const double* Sblk, const double* g, int cuts, int rank, int world,
                                        int (*allreduce)(void* user, double* buf, size_t count, void* stream), void* allreduce_user, double* x,
                                        int* info);

/* Device self-test of the hand-written scan and radix sort the BA structure builder and the BoW merge-join run on (csrc/sv_sort.hip; NOT a
 * product entry point).  values[n] >= 0 are scanned in place semantics: scan_out[n + 1] receives the exclusive prefix sums and the total
 * (n above 16384 takes the many-workgroup path).  keys[n] (low `bits` bits significant) are sorted stably: sorted_idx[n] receives the
 * permutation.  Either half is skipped when its output pointer is NULL.  Host in / out, synchronous. */
int svgpu_selftest_scan_sort(svgpu_ctx* ctx, int n, const int32_t* values, int32_t* scan_out, const uint32_t* keys, int bits, int32_t* sorted_idx);

/* Host in/out, synchronous.  The Levenberg-Marquardt loop (damping trials, rho test, terminate_action) runs on the device; the
 * host enqueues the trials of a stage and reads the control block back once per stage (plus once per rejected trial).
 *   stop        nullable; the caller's force_stop_flag (mapping_module.h:232).  Polled at every damping-trial boundary
 *               (mirrored into page-locked memory while the host waits) and -- reference quirk, terminate_action.cc:36-76 --
 *               SET when the gain rule stops stage 1, so that stage 2 is skipped exactly as in the reference.
 *   pose_out    num_poses x 12, points_out num_points x 3, outlier_out num_obs (1 = outlier observation)
 * Limits: num_obs < 2^32 / 144 (29.8 M) and num_points < 2^32 / 48 per call -- per RANK of the sharded variants --: the record gathers of the
 * reduced-system kernel address their arrays with 32-bit byte offsets; beyond, SVGPU_ERR_INVALID (shard the landmarks over more ranks). */
int svgpu_local_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out,
                   double* points_out, uint8_t* outlier_out, svgpu_ba_stats* stats);

/* ------------------------------------------------------------------------------------------------ pose optimizer
 * Stands behind  stella_vslam::optimize::pose_optimizer::optimize  (optimize/pose_optimizer.h; g2o implementation
 * optimize/pose_optimizer_g2o.cc:38-175; factory defaults 2 / 2 / 10, optimize/pose_optimizer_factory.h:18-26): motion-only
 * BA of one frame against its n observed landmarks -- (num_trials_robust + num_trials) rounds of <= num_each_iter LM
 * iterations, each followed by the chi-square (5.99146 / 7.81473) re-classification of every observation; Huber kernels
 * are dropped after round num_trials_robust.  Fewer than 5 observations => *num_valid = 0 and the pose is returned unchanged.
 *   pose_cw / pose_out   3x4 [R|t] row-major; pos_w n x 3; uvr n x 3 f32 (u_right < 0 => monocular edge)
 *   huber_delta          per observation (sqrt(5.99146) for monocular cameras else sqrt(7.81473)); <= 0 => no kernel
 *   reset_stop_flag_each_round  0 = literal g2o behaviour (the terminate action's flag, once raised by the gain rule, also
 *                        suppresses the LM iterations of the later rounds); 1 = reset it before every round
 * The whole optimisation is ONE single-workgroup kernel launch.  Host in/out, synchronous. */
int svgpu_pose_optimize(svgpu_ctx* ctx, const double* pose_cw, int n, const double* pos_w, const float* uvr,
                        const float* inv_sigma_sq, const float* huber_delta, const double* intrinsics /* fx fy cx cy fxb, or 0 0 cols rows 0 = equirectangular */,
                        int num_trials_robust, int num_trials, int num_each_iter, int reset_stop_flag_each_round,
                        double* pose_out, uint8_t* outlier_flags, int* num_valid, int* lm_iterations);

/* Same with the observations already on the context's device (pos_w_dev n x 3 f64, uvr_dev n x 3 f32, inv_sigma_sq_dev / huber_delta_dev n
 * f32), e.g. the gathered landmarks a projection matcher just worked on: no input copy; pose, intrinsics and the results stay host-side. */
int svgpu_pose_optimize_device(svgpu_ctx* ctx, const double* pose_cw, int n, const double* pos_w_dev, const float* uvr_dev,
                               const float* inv_sigma_sq_dev, const float* huber_delta_dev, const double* intrinsics, int num_trials_robust,
                               int num_trials, int num_each_iter, int reset_stop_flag_each_round, double* pose_out,
                               uint8_t* outlier_flags, int* num_valid, int* lm_iterations);

/* Global BA core (optimize/global_bundle_adjuster.cc:26-192, 279-412): the same graph over ALL keyframes (spanning root
 * fixed), ONE Levenberg-Marquardt run of problem->num_first_iter iterations with the terminate rule, optional Huber
 * (obs_huber_delta), no outlier gate (num_second_iter is ignored).  The caller applies the reference's post-conditions
 * (`force_stop_flag && *force_stop_flag && !stats->stopped_by_terminate_action` => discard, :341-343).
 * Reduced systems beyond the on-chip solvers are factored by the block envelope Cholesky of the block-sparse Schur complement
 * (svgpu_ba_set_solver; the reference uses a sparse CSparse Cholesky there; the block-Jacobi PCG stays selectable).  Host in/out, synchronous. */
int svgpu_global_ba(svgpu_ctx* ctx, const svgpu_ba_problem* problem, volatile uint8_t* stop, double* pose_out,
                    double* points_out, svgpu_ba_stats* stats);

/* Multi-GPU variant.  Every rank passes the FULL pose / point arrays and ITS SHARD of the observations; the shard
 * must be BY LANDMARK (all observations of one landmark on one rank) so that the Schur complement of a landmark is formed
 * locally: either by keyframe segment (svgpu_ba_partition_keyframe_segments below: per damping trial only the separator blocks of the
 * reduced system cross ranks) or any other way, e.g. obs_point % world == rank -- then, per damping trial, the kept 6x6 blocks of the
 * partial reduced camera systems and their right-hand sides (36 * blocks + 6 * free poses doubles) are summed with ONE all-reduce.  The
 * factorisation of a large reduced system is distributed over the ranks (segmented envelope elimination), small ones are solved on every rank; per linearisation the pose blocks Hpp/bp (42 doubles per free pose) and per trial four scalars
 * (chi2, step scale, solver failure, stop votes) are summed as well.  All collectives are enqueued on the context's stream:
 * the damping loop never waits for the host.
 *   allreduce == NULL   the context's own RCCL communicator is used (svgpu_comm_init below: no Python, no callback)
 *   allreduce != NULL   `allreduce(user, dev_buf, count, stream)` must sum `count` doubles in place across ranks, ordered on
 *                       `stream` (stella_vslam_amd/distributed.py binds it to torch.distributed: RCCL, or gloo in CPU-side tests)
 * All ranks return identical poses and points; outlier_out covers the local shard.  A rank's stop flag is a VOTE: the votes are
 * summed inside the per-trial all-reduce and every rank acts on the sum only, so ranks never diverge between two collectives
 * however the callers' flags are raised.  Whether a stop POINTER was passed is agreed the same way: if any rank passes one, every rank
 * behaves as if it had (a NULL `stop` on some ranks only is allowed and cannot desynchronise the early return / skipped second stage). */
typedef int (*svgpu_allreduce_fn)(void* user, double* dev_buf, size_t count, void* stream);
int svgpu_local_ba_sharded(svgpu_ctx* ctx, const svgpu_ba_problem* shard, int rank, int world,
                           svgpu_allreduce_fn allreduce, void* allreduce_user, volatile uint8_t* stop, double* pose_out,
                           double* points_out, uint8_t* outlier_out, svgpu_ba_stats* stats);
/* svgpu_global_ba over the same sharding (BASELINE config 5: landmark-sharded global BA at 1/2/4/8 GPUs). */
int svgpu_global_ba_sharded(svgpu_ctx* ctx, const svgpu_ba_problem* shard, int rank, int world,
                            svgpu_allreduce_fn allreduce, void* allreduce_user, volatile uint8_t* stop, double* pose_out,
                            double* points_out, svgpu_ba_stats* stats);

/* Keyframe-segment partition of a global-BA problem over `world` ranks (optimize/global_bundle_adjuster.cc:26-192 is the workload):
 * the ordered keyframe graph is cut by the vertex separators of the segmented envelope solve (the cuts the solve itself will make for
 * `world` ranks), every piece between two cuts is a job owned by one rank, and a landmark goes to the rank that owns the piece its
 * keyframes lie in -- co-observation makes a landmark's keyframes a clique of the keyframe graph, so they all lie in ONE piece plus
 * separator keyframes.  Host only (no device is touched, ctx-free).
 *   landmark_rank   num_points entries: the rank whose shard gets every observation of that landmark (a valid BY-LANDMARK sharding)
 *   info            12 ints: [0] 1 = keyframe segments, 0 = no segmented plan for this problem: landmark_rank = l % world;
 *                   [1] jobs  [2] cuts  [3] separator keyframes  [4] landmarks seen from a separator keyframe (their blocks are what
 *                   crosses ranks)  [5] of those, seen from separator keyframes only  [6] free keyframes  [7] kept 6x6 blocks
 *                   [8] kept blocks between two separator keyframes  [9] doubles the jobs leave on the separators per trial  [10..11] 0
 * svgpu_global_ba_sharded / svgpu_local_ba_sharded RECOGNISE shards cut this way (every observation of a piece's keyframe on the
 * piece's rank; agreed between the ranks by one flag at set-up) and then exchange, per damping trial, only the separator-by-separator
 * blocks and separator rows of the reduced system + what the jobs leave on the separators + the solution, instead of the whole
 * reduced system; any other by-landmark sharding (l % world) keeps the full exchange.  SVGPU_BA_EXCHANGE=full forces the latter. */
int svgpu_ba_partition_keyframe_segments(const svgpu_ba_problem* problem, int world, int32_t* landmark_rank, int32_t* info);
/* What the last sharded solve on this context all-reduced, in bytes of payload per rank: info[0] 0 = not sharded, 1 = whole reduced
 * system per trial, 2 = keyframe-segment exchange; [1] set-up  [2] pose blocks (per linearisation)  [3] reduced system (per trial)
 * [4] separator contributions of the jobs (per trial)  [5] solution (per trial)  [6] trial sums / damping slots; [7] all-reduce calls
 * [8] damping trials  [9] linearisations. */
int svgpu_ba_last_exchange(svgpu_ctx* ctx, int64_t* info);

/* ------------------------------------------------------------------------------------------------ RCCL communicator
 * One communicator per context (= per GPU / process), RCCL over xGMI, loaded with dlopen on first use.  Rank 0 calls
 * svgpu_comm_unique_id and hands the 128 bytes (ncclUniqueId) to the other ranks through whatever channel the host
 * application has (the reference has none: a multi-process launcher passes it via its own rendezvous; bench.py and the tests
 * broadcast it with torch.distributed); every rank then calls svgpu_comm_init. */
int svgpu_comm_unique_id(uint8_t* id128);
int svgpu_comm_init(svgpu_ctx* ctx, int rank, int world, const uint8_t* id128);
void svgpu_comm_destroy(svgpu_ctx* ctx);
/* Sum of `count` doubles in place over the communicator, ordered on `stream` (NULL = the context's stream). */
int svgpu_comm_allreduce_f64(svgpu_ctx* ctx, double* dev_buf, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVGPU_H */
