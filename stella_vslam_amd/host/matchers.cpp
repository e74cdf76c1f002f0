#include "matchers.h"

#include <stdexcept>
#include <string>

namespace stella_vslam_hip {
namespace match {

namespace {
void check(svgpu_ctx* ctx, int rc, const char* where) {
    if (rc != SVGPU_OK) throw std::runtime_error(std::string(where) + ": " + svgpu_status_string(rc) + " (" + svgpu_last_error(ctx) + ")");
}
std::vector<uint8_t> pack_rows(const cv::Mat& m) {
    std::vector<uint8_t> out((size_t)m.rows * 32);
    for (int r = 0; r < m.rows; ++r) std::memcpy(out.data() + (size_t)r * 32, m.ptr(r), 32);
    return out;
}
}  // namespace

unsigned int robust::brute_force_match(const data::frame_observation& frm_obs, const data::frame_observation& keyfrm_obs,
                                       const std::vector<unsigned char>& keyfrm_lm_valid, std::vector<std::pair<int, int>>& matches) const {
    const int n1 = (int)frm_obs.undist_keypts_.size(), n2 = (int)keyfrm_obs.undist_keypts_.size();
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; ++i) a1[i] = frm_obs.undist_keypts_[i].angle;
    for (int i = 0; i < n2; ++i) a2[i] = keyfrm_obs.undist_keypts_[i].angle;
    const auto d1 = pack_rows(frm_obs.descriptors_), d2 = pack_rows(keyfrm_obs.descriptors_);
    std::vector<int32_t> matched((size_t)std::max(n1, 1), -1);
    int num = 0;
    check(ctx_, svgpu_match_bruteforce(ctx_, d1.data(), a1.data(), n1, d2.data(), a2.data(), keyfrm_lm_valid.empty() ? nullptr : keyfrm_lm_valid.data(),
                                       n2, lowe_ratio_, check_orientation_ ? 1 : 0, matched.data(), &num),
          "svgpu_match_bruteforce");
    matches.clear();
    matches.reserve((size_t)num);
    for (int i = 0; i < n1; ++i)
        if (matched[i] >= 0) matches.emplace_back(i, matched[i]);
    return (unsigned int)num;
}

unsigned int projection::match(const query_set& q, const data::frame_observation& frm_obs, const std::vector<unsigned char>& occupied,
                               int mode, unsigned int hamm_dist_thr, std::vector<int>& matched_idx_for_query) const {
    const int nq = q.descriptors.rows, nt = (int)frm_obs.undist_keypts_.size();
    std::vector<float> ta(nt);
    std::vector<int32_t> toct(nt);
    for (int i = 0; i < nt; ++i) {
        ta[i] = frm_obs.undist_keypts_[i].angle;
        toct[i] = frm_obs.undist_keypts_[i].octave;
    }
    const auto qd = pack_rows(q.descriptors), td = pack_rows(frm_obs.descriptors_);
    const bool stereo = !frm_obs.stereo_x_right_.empty() && !q.x_right.empty();
    matched_idx_for_query.assign((size_t)std::max(nq, 1), -1);
    int num = 0;
    check(ctx_, svgpu_match_candidates(ctx_, qd.data(), nq, td.data(), toct.data(), nt, q.cand_off.data(), q.cand_idx.data(),
                                       q.cand_skip.empty() ? nullptr : q.cand_skip.data(), q.valid.empty() ? nullptr : q.valid.data(), occupied.empty() ? nullptr : occupied.data(),
                                       q.angle.empty() ? nullptr : q.angle.data(), ta.data(), (check_orientation_ && !q.angle.empty()) ? 1 : 0,
                                       stereo ? q.x_right.data() : nullptr, stereo ? frm_obs.stereo_x_right_.data() : nullptr,
                                       stereo ? q.x_right_tol.data() : nullptr, hamm_dist_thr, lowe_ratio_,
                                       mode,
                                       matched_idx_for_query.data(), &num),
          "svgpu_match_candidates");
    matched_idx_for_query.resize((size_t)nq);
    return (unsigned int)num;
}

unsigned int projection::match_in_cells(const query_set& q, const std::vector<cv::Point2f>& ref_pts, const std::vector<float>& margins,
                                        const std::vector<int>& min_levels, const std::vector<int>& max_levels,
                                        const data::frame_observation& frm_obs, const std::vector<unsigned char>& occupied,
                                        const float img_bounds[4], int num_grid_cols, int num_grid_rows, int mode, unsigned int hamm_dist_thr,
                                        std::vector<int>& matched_idx_for_query) const {
    const int nq = q.descriptors.rows, nt = (int)frm_obs.undist_keypts_.size();
    std::vector<float> ta(nt), txy(2 * (size_t)nt), qxy(2 * (size_t)nq);
    std::vector<int32_t> toct(nt), qlo(min_levels.begin(), min_levels.end()), qhi(max_levels.begin(), max_levels.end());
    for (int i = 0; i < nt; ++i) {
        ta[i] = frm_obs.undist_keypts_[i].angle;
        toct[i] = frm_obs.undist_keypts_[i].octave;
        txy[2 * i] = frm_obs.undist_keypts_[i].pt.x;
        txy[2 * i + 1] = frm_obs.undist_keypts_[i].pt.y;
    }
    for (int i = 0; i < nq; ++i) {
        qxy[2 * i] = ref_pts[i].x;
        qxy[2 * i + 1] = ref_pts[i].y;
    }
    const auto qd = pack_rows(q.descriptors), td = pack_rows(frm_obs.descriptors_);
    const bool stereo = !frm_obs.stereo_x_right_.empty() && !q.x_right.empty();
    matched_idx_for_query.assign((size_t)std::max(nq, 1), -1);
    int num = 0;
    check(ctx_, svgpu_match_in_cells(ctx_, qd.data(), nq, qxy.data(), margins.data(), qlo.empty() ? nullptr : qlo.data(),
                                     qhi.empty() ? nullptr : qhi.data(), q.valid.empty() ? nullptr : q.valid.data(),
                                     q.angle.empty() ? nullptr : q.angle.data(), stereo ? q.x_right.data() : nullptr,
                                     stereo ? q.x_right_tol.data() : nullptr, td.data(), txy.data(), toct.data(), nt,
                                     occupied.empty() ? nullptr : occupied.data(), ta.data(), stereo ? frm_obs.stereo_x_right_.data() : nullptr,
                                     img_bounds[0], img_bounds[1], img_bounds[2], img_bounds[3], num_grid_cols, num_grid_rows,
                                     (check_orientation_ && !q.angle.empty()) ? 1 : 0, hamm_dist_thr, lowe_ratio_, mode,
                                     matched_idx_for_query.data(), &num),
          "svgpu_match_in_cells");
    matched_idx_for_query.resize((size_t)nq);
    return (unsigned int)num;
}

unsigned int projection::match_frame_and_landmarks(const camera::base& cam, const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& trans_wc,
                                                   const camera::landmark_set& lms, const data::frame_observation& frm_obs,
                                                   const std::vector<unsigned char>& occupied, const feature::orb_params& orb_params,
                                                   unsigned int num_grid_cols, unsigned int num_grid_rows, float margin,
                                                   std::vector<int>& matched_idx_for_landmark, camera::observability& obs, float ray_cos_thr) const {
    const int n = (int)lms.pos_w.size(), nt = (int)frm_obs.undist_keypts_.size();
    std::vector<float> txy(2 * (size_t)nt);
    std::vector<int32_t> toct(nt);
    for (int i = 0; i < nt; ++i) {
        toct[i] = frm_obs.undist_keypts_[i].octave;
        txy[2 * i] = frm_obs.undist_keypts_[i].pt.x;
        txy[2 * i + 1] = frm_obs.undist_keypts_[i].pt.y;
    }
    const auto qd = pack_rows(lms.descriptors), td = pack_rows(frm_obs.descriptors_);
    matched_idx_for_landmark.assign((size_t)std::max(n, 1), -1);
    obs.visible.assign((size_t)n, 0);
    obs.reproj.assign((size_t)n, Vec2_t{0, 0});
    obs.x_right.assign((size_t)n, 0.f);
    obs.pred_scale_level.assign((size_t)n, -1);
    int num = 0;
    check(ctx_, svgpu_match_frame_and_landmarks(ctx_, &cam.c_abi(), rot_cw.data(), trans_cw.data(), trans_wc.data(), n,
                                                reinterpret_cast<const double*>(lms.pos_w.data()), reinterpret_cast<const double*>(lms.mean_normal.data()),
                                                lms.min_valid_dist.data(), lms.max_valid_dist.data(), lms.skip.empty() ? nullptr : lms.skip.data(),
                                                qd.data(), ray_cos_thr, (int)orb_params.num_levels_, orb_params.scale_factors_.data(),
                                                orb_params.log_scale_factor_, margin, td.data(), txy.data(), toct.data(), nt,
                                                occupied.empty() ? nullptr : occupied.data(),
                                                frm_obs.stereo_x_right_.empty() ? nullptr : frm_obs.stereo_x_right_.data(), (int)num_grid_cols,
                                                (int)num_grid_rows, HAMMING_DIST_THR_HIGH, lowe_ratio_, matched_idx_for_landmark.data(), &num,
                                                obs.visible.data(), reinterpret_cast<double*>(obs.reproj.data()), obs.x_right.data(),
                                                obs.pred_scale_level.data()),
          "svgpu_match_frame_and_landmarks");
    matched_idx_for_landmark.resize((size_t)n);
    return (unsigned int)num;
}

void stereo::compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const {
    const int nl = (int)keypts_left_.size(), nr = (int)keypts_right_.size();
    stereo_x_right.assign((size_t)nl, -1.0f);
    depths.assign((size_t)nl, -1.0f);
    if (nl == 0) return;
    const auto dl = pack_rows(descs_left_), dr = pack_rows(descs_right_);
    static_assert(sizeof(cv::KeyPoint) == sizeof(svgpu_keypoint), "KeyPoint layout");
    check(el_->context(), svgpu_stereo_match(el_->context(), er_->context(), reinterpret_cast<const svgpu_keypoint*>(keypts_left_.data()),
                                              dl.data(), nl, reinterpret_cast<const svgpu_keypoint*>(keypts_right_.data()), dr.data(), nr,
                                              focal_x_baseline_, true_baseline_, stereo_x_right.data(), depths.data()),
          "svgpu_stereo_match");
}

}  // namespace match
}  // namespace stella_vslam_hip
