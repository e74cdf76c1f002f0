// C++ adaptors for stella_vslam::match::* (reference: src/stella_vslam/match/base.h:81-91, robust.h, projection.h).
// Constructor arguments, thresholds and results are the reference's; the object graph the reference walks
// (frame / keyframe / landmark shared_ptrs) is flattened by the caller to the parts the arithmetic touches:
// descriptor rows, keypoints, "has a live landmark" flags and candidate index lists.
#pragma once
#include <utility>
#include <vector>

#include "camera.h"
#include "orb_extractor.h"

namespace stella_vslam_hip {
namespace data {
// the flat part of data::frame_observation (data/frame_observation.h:12-38) the matchers read
struct frame_observation {
    cv::Mat descriptors_;                      // N x 32, CV_8U
    std::vector<cv::KeyPoint> undist_keypts_;  // angle, octave, pt
    std::vector<float> stereo_x_right_;        // empty or N
};
}  // namespace data

namespace match {

static constexpr unsigned int HAMMING_DIST_THR_LOW = 50;
static constexpr unsigned int HAMMING_DIST_THR_HIGH = 100;
static constexpr unsigned int MAX_HAMMING_DIST = 256;

class base {
public:
    base(svgpu_ctx* ctx, float lowe_ratio, bool check_orientation) : ctx_(ctx), lowe_ratio_(lowe_ratio), check_orientation_(check_orientation) {}
    virtual ~base() = default;

protected:
    svgpu_ctx* ctx_;
    const float lowe_ratio_;
    const bool check_orientation_;
};

class robust final : public base {
public:
    explicit robust(svgpu_ctx* ctx, float lowe_ratio = 0.6f, bool check_orientation = true) : base(ctx, lowe_ratio, check_orientation) {}
    //! robust::brute_force_match (match/robust.cc:232-328).  keyfrm_lm_valid[idx_2] != 0 <=> the keyframe keypoint
    //! has a landmark that is not will_be_erased() (:258-263).  matches = (idx_1, idx_2) sorted by idx_1.
    unsigned int brute_force_match(const data::frame_observation& frm_obs, const data::frame_observation& keyfrm_obs,
                                   const std::vector<unsigned char>& keyfrm_lm_valid, std::vector<std::pair<int, int>>& matches) const;
};

class projection final : public base {
public:
    explicit projection(svgpu_ctx* ctx, float lowe_ratio = 0.6f, bool check_orientation = true) : base(ctx, lowe_ratio, check_orientation) {}
    struct query_set {                      // one entry per landmark / last-frame keypoint, in the reference's loop order
        cv::Mat descriptors;                // nq x 32
        std::vector<float> angle;           // nq (only read when check_orientation)
        std::vector<float> x_right, x_right_tol;  // empty or nq: stereo gate (projection.cc:57-62)
        std::vector<unsigned char> valid;   // empty or nq
        std::vector<int> cand_off, cand_idx;  // CSR of get_keypoints_in_cell results (data/common.cc:127-190), scan order kept
        std::vector<unsigned char> cand_skip; // empty or one per CSR entry: pair gates evaluated by the caller (epipolar, chi-square)
    };
    //! mode = svgpu_match_mode: projection::match_frame_and_landmarks (RATIO_SAME_OCTAVE, thr HIGH), match_current_and_last_frames
    //! and fuse::detect_duplication (BEST_ONLY), bow_tree::match_frame_and_keyframe (RATIO), *::match_for_triangulation (TRIANGULATION)
    unsigned int match(const query_set& q, const data::frame_observation& frm_obs, const std::vector<unsigned char>& occupied,
                       int mode, unsigned int hamm_dist_thr, std::vector<int>& matched_idx_for_query) const;

    //! Same, with the candidate lists built on the device: query i scans
    //! frm.get_keypoints_in_cell(ref_pts[i].x, ref_pts[i].y, margins[i], min_levels[i], max_levels[i]) (data/common.cc:127-190);
    //! q.cand_off / q.cand_idx / q.cand_skip are ignored.  img_bounds = {min_x, max_x, min_y, max_y} of camera::base.
    unsigned int match_in_cells(const query_set& q, const std::vector<cv::Point2f>& ref_pts, const std::vector<float>& margins,
                                const std::vector<int>& min_levels, const std::vector<int>& max_levels,
                                const data::frame_observation& frm_obs, const std::vector<unsigned char>& occupied, const float img_bounds[4],
                                int num_grid_cols, int num_grid_rows, int mode, unsigned int hamm_dist_thr,
                                std::vector<int>& matched_idx_for_query) const;

    //! projection::match_frame_and_landmarks (match/projection.cc:13-93) fused with the observability loop that feeds it
    //! (tracking_module.cc:554-594 -> data/frame.cc:59-85): nothing returns to the host between reprojection, window lookup
    //! (get_keypoints_in_cell) and matching.  matched_idx_for_landmark[i] = idx of frm.add_landmark(lm_i, idx) or -1; `obs` gets
    //! the lm_to_reproj / lm_to_x_right / lm_to_scale values (callers bump increase_num_observable from obs.visible).
    unsigned int match_frame_and_landmarks(const camera::base& cam, const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& trans_wc,
                                           const camera::landmark_set& lms, const data::frame_observation& frm_obs,
                                           const std::vector<unsigned char>& occupied, const feature::orb_params& orb_params,
                                           unsigned int num_grid_cols, unsigned int num_grid_rows, float margin,
                                           std::vector<int>& matched_idx_for_landmark, camera::observability& obs, float ray_cos_thr = 0.5f) const;
};

//! match::area (match/area.h): the monocular initialiser's matcher on the same flattened inputs.  query_set = the keypoints of
//! frame 1 in index order; cand list of idx_1 = frm_2.get_keypoints_in_cell(prev_matched_pts[idx_1], margin, 0, 0), empty for
//! keypoints above level 0 (match/area.cc:17-32).  The caller refreshes prev_matched_pts from the result (:91-95).
class area final : public base {
public:
    using base::base;
    unsigned int match_in_consistent_area(const projection::query_set& frm_1, const data::frame_observation& frm_2_obs,
                                          std::vector<int>& matched_indices_2_in_frm_1) const {
        return projection(ctx_, lowe_ratio_, check_orientation_).match(frm_1, frm_2_obs, {}, SVGPU_MATCH_AREA, 50, matched_indices_2_in_frm_1);
    }
};

//! match::stereo (match/stereo.h): same constructor roles; the image pyramids are the two extractors' (their last extract call)
class stereo {
public:
    stereo(const feature::orb_extractor* extractor_left, const feature::orb_extractor* extractor_right,
           const std::vector<cv::KeyPoint>& keypts_left, const std::vector<cv::KeyPoint>& keypts_right, const cv::Mat& descs_left,
           const cv::Mat& descs_right, float focal_x_baseline, float true_baseline)
        : el_(extractor_left), er_(extractor_right), keypts_left_(keypts_left), keypts_right_(keypts_right), descs_left_(descs_left),
          descs_right_(descs_right), focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline) {}
    virtual ~stereo() = default;
    //! Compute stereo matching in subpixel order (match/stereo.cc:20-114)
    void compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const;

private:
    const feature::orb_extractor* el_;
    const feature::orb_extractor* er_;
    const std::vector<cv::KeyPoint>& keypts_left_;
    const std::vector<cv::KeyPoint>& keypts_right_;
    const cv::Mat& descs_left_;
    const cv::Mat& descs_right_;
    const float focal_x_baseline_;
    const float true_baseline_;
};

}  // namespace match
}  // namespace stella_vslam_hip
