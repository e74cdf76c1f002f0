// The tracked-frame chain behind the reference's call sites: what tracking_module does per image
// (tracking_module.cc:253-275, 333-355, 533-608; module/frame_tracker.cc:22-60) through svgpu_track_motion / svgpu_track_local_map --
// landmark IDS instead of flattened landmark records (the records live on the device: drop_in/map_mirror.h), one submission and one
// synchronisation per half, the first optimisation's pose handed to the second half on the device.
// Compiles against the reference tree with -DSVGPU_WITH_STELLA_VSLAM, or stand-alone against host/standin/stella_standin.h.
#pragma once
#include "hip_backend.h"

namespace stella_vslam {
namespace hip {

//! Uploads what data::landmark's mutators have reported since the last flush (map_mirror.h) and returns the device table.  Called by the
//! chain before it reads the table and by the HIP bundle adjusters after their write-back (so the tracking thread rarely finds anything
//! left to upload).  Thread-safe.
svgpu_map* flush_map(svgpu_ctx* ctx);
//! number of landmarks whose record is waiting for the next flush (diagnostics / tests)
size_t pending_map_updates();
// explicit shutdown: destroys the device table (the mirror itself is never destroyed: no HIP call during static destruction); a later
// flush_map creates a fresh table and uploads every live record again
void release_map();

//! One per tracking thread (tracking_module owns it next to frame_tracker_).
class tracked_frame_chain {
public:
    //! `ctx`: the context the chain's launches go to -- for the fused extraction the extractor's own (orb_extractor::context())
    tracked_frame_chain(svgpu_ctx* ctx, const camera::base* camera, const feature::orb_params* orb_params, unsigned int num_grid_cols = 64,
                        unsigned int num_grid_rows = 48, unsigned int num_trials_robust = 2, unsigned int num_trials = 2, unsigned int num_each_iter = 10);
    ~tracked_frame_chain();
    tracked_frame_chain(const tracked_frame_chain&) = delete;
    tracked_frame_chain& operator=(const tracked_frame_chain&) = delete;

    //! frame_tracker::motion_based_track (module/frame_tracker.cc:22-60): motion-model pose, match_current_and_last_frames (margin, then
    //! twice the margin), pose optimisation, discard_outliers.  With `img` the frame's observation is created in the same submission
    //! (system.cc:380-395: extract, undistort_keypoints, convert_keypoints_to_bearings, assign_keypoints_to_grid): curr_frm.frm_obs_ is
    //! filled here, `keypts` receives the extractor's (distorted) keypoints, and the resident copy is registered under curr_frm.id_.
    //! With `img_right` as well (and a right context, set_right_context) the frame is a STEREO frame (system.cc:406-447): the right image is
    //! extracted on the right context beside the left one, match::stereo::compute fills frm_obs_.stereo_x_right_ / depths_ -- same submission.
    bool motion_based_track(data::frame& curr_frm, const data::frame& last_frm, const Mat44_t& velocity, unsigned int num_matches_thr, float margin,
                            const cv::Mat* img = nullptr, std::vector<cv::KeyPoint>* keypts = nullptr, const cv::Mat* img_right = nullptr,
                            const cv::Mat* img_depth = nullptr);  //!< RGB-D (system.cc:466-526): CV_32F depth in metres instead of a right image
    //! the context of the RIGHT camera's extractor (feature::orb_extractor::context() of extractor_right_; configured like the left one)
    void set_right_context(svgpu_ctx* ctx_right) { ctx_right_ = ctx_right; }

    //! tracking_module::search_local_landmarks (tracking_module.cc:533-608) followed by the optimisation and outlier rejection of
    //! optimize_current_frame_with_local_map (:441-455).  Returns false when no local landmark can be projected ("projection candidate
    //! not found"); then nothing has been changed.  The caller goes on with the counting loop of :457-480.
    bool track_local_map(data::frame& curr_frm, const std::vector<std::shared_ptr<data::landmark>>& local_landmarks, unsigned int fixed_keyframe_id_threshold,
                         float margin, float lowe_ratio = 0.8f);

    //! lm_to_reproj / lm_to_x_right / lm_to_scale of the last track_local_map, for callers that still want the reference's three maps
    void last_observability(eigen_alloc_unord_map<unsigned int, Vec2_t>& lm_to_reproj, std::unordered_map<unsigned int, float>& lm_to_x_right,
                            std::unordered_map<unsigned int, unsigned int>& lm_to_scale);

    svgpu_track_result last_motion_{}, last_local_{};  // statistics of the last calls
    //! launches + copies enqueued and stream synchronisations waited on, since construction
    void counters(long long& launches, long long& host_syncs) const;
    const unsigned long long* debug_stamps() const { return svgpu_tracker_debug_stamps(tracker_); }

private:
    svgpu_ctx* const ctx_;
    const camera::base* const camera_;
    const feature::orb_params* const orb_params_;
    const unsigned int num_grid_cols_, num_grid_rows_;
    svgpu_tracker* tracker_ = nullptr;
    svgpu_map* map_ = nullptr;
    // scratch kept across frames (no per-frame allocation)
    std::vector<int32_t> last_ids_, match_, cur_ids_, local_ids_;
    std::vector<uint8_t> outlier_, visible_;
    std::vector<svgpu_keypoint> kps_, und_;
    std::vector<uint8_t> desc_;
    std::vector<double> brg_;
    std::vector<unsigned int> last_local_lm_ids_;
    // the resident observations of the frames this chain has just handled, by frame id: the second half of a frame and the first half of the
    // next one take them from here instead of re-verifying a 150 KB observation against the cache (data::frame::frm_obs_ is constant)
    struct memo {
        unsigned int id = 0;
        frame_handle h;
    } memo_[2];
    frame_handle handle_of(const data::frame& frm);
    void remember(unsigned int frame_id, const frame_handle& h);
    uint32_t frame_serial_ = 0;
    svgpu_ctx* ctx_right_ = nullptr;
    bool device_pose_valid_ = false;       // the tracker's device pose is the pose motion_based_track gave frame device_pose_frame_
    unsigned int device_pose_frame_ = 0;
    std::vector<uint32_t> held_stamp_;  // per landmark id: serial of the frame that holds it (curr_landmark_ids of :536-551 without a hash set)
};

}  // namespace hip
}  // namespace stella_vslam
