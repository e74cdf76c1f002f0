// The PRODUCT's drop-in class optimize::local_bundle_adjuster_hip (stella_vslam_amd/host/drop_in/hip_backend.cc, compiled here with
// -DSVGPU_WITH_STELLA_VSLAM -DSVGPU_DROP_IN_OPTIMIZE_ONLY against the same stand-in data:: headers the reference's
// local_bundle_adjuster_g2o.cc is compiled against in ref_ba_exports.cc) on the toy map of svref_local_ba: the two classes see identical
// objects, tests/test_gpu_drop_in_vs_reference.py compares the maps they leave behind.  Links libsvgpu.so: needs a GPU to run.
#include <cstring>
#include <array>
#include <memory>
#include <unordered_set>
#include <vector>

#include "drop_in/global_bundle_adjuster_hip.h"
#include "drop_in/hip_backend.h"

using namespace stella_vslam;
std::mutex stella_vslam::data::map_database::mtx_database_;

namespace {
std::unique_ptr<camera::base> make_cam(int model, int stereo, unsigned cols, unsigned rows, const double* k) {
    const auto setup = stereo ? camera::setup_type_t::Stereo : camera::setup_type_t::Monocular;
    const auto col = camera::color_order_t::Gray;
    switch (model) {
        case 0: return std::unique_ptr<camera::base>(new camera::perspective("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, 0, k[4]));
        case 1: return std::unique_ptr<camera::base>(new camera::fisheye("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, k[4]));
        case 2: return std::unique_ptr<camera::base>(new camera::equirectangular("ref", col, cols, rows, 30.0));
        default: return std::unique_ptr<camera::base>(new camera::radial_division("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, k[4]));
    }
}
struct toy_map {  // the object graph of svref_local_ba / svref_global_ba (ref_ba_exports.cc), from the same flat arrays
    std::unique_ptr<camera::base> cam;
    feature::orb_params orb;
    std::vector<std::shared_ptr<data::keyframe>> kfs;
    std::vector<std::shared_ptr<data::landmark>> lms;
    toy_map(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, float scale_factor, int num_levels, int K, const unsigned* kf_id,
            const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos, const uint8_t* lm_erased, int O, const int* obs_kf,
            const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr, const int* obs_oct)
        : cam(make_cam(model, stereo_cam, cols, rows, intr5)), orb("ref", scale_factor, num_levels, 20, 7), kfs(K), lms(L) {
        for (int k = 0; k < K; ++k) {
            kfs[k] = std::make_shared<data::keyframe>();
            kfs[k]->id_ = kf_id[k];
            kfs[k]->camera_ = cam.get();
            kfs[k]->orb_params_ = &orb;
            kfs[k]->pose_cw_ = Mat44_t::Identity();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) kfs[k]->pose_cw_(i, j) = kf_pose[12 * k + 4 * i + j];
            kfs[k]->will_be_erased_ = kf_flags[k] & 1;
            kfs[k]->graph_node_->is_spanning_root_ = (kf_flags[k] & 2) != 0;
        }
        for (int l = 0; l < L; ++l) lms[l] = std::make_shared<data::landmark>(lm_id[l], Vec3_t(lm_pos[3 * l], lm_pos[3 * l + 1], lm_pos[3 * l + 2]), lm_erased[l] != 0);
        for (int o = 0; o < O; ++o) {
            auto& kf = kfs[obs_kf[o]];
            const size_t idx = (size_t)obs_idx[o];
            if (kf->frm_obs_.undist_keypts_.size() <= idx) {
                kf->frm_obs_.undist_keypts_.resize(idx + 1);
                kf->landmarks_.resize(idx + 1);
                if (stereo_cam) kf->frm_obs_.stereo_x_right_.resize(idx + 1, -1.0f);
            }
            cv::KeyPoint kp;
            kp.pt.x = obs_uv[2 * o];
            kp.pt.y = obs_uv[2 * o + 1];
            kp.octave = obs_oct[o];
            kf->frm_obs_.undist_keypts_[idx] = kp;
            if (stereo_cam) kf->frm_obs_.stereo_x_right_[idx] = obs_xr[o];
            kf->landmarks_[idx] = lms[obs_lm[o]];
            lms[obs_lm[o]]->observations_[kf] = (unsigned)idx;
        }
    }
};
}  // namespace

extern "C" {
// same arguments and outputs as svref_local_ba (ref_ba_exports.cc) minus the graph-order arrays; stats6 = {status, iterations of the two
// stages, stage 2 entered, gated observations, LM trials}
int svref_dropin_local_ba(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, float scale_factor, int num_levels, int K,
                          const unsigned* kf_id, const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos,
                          const uint8_t* lm_erased, int O, const int* obs_kf, const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr,
                          const int* obs_oct, int curr, int n_covis, const int* covis, unsigned fixed_threshold, int use_additional, int iters1, int iters2,
                          int stop_in, double* kf_pose_out, double* lm_pos_out, int* n_erased, int* erased_pairs, int* lm_counters, int* kf_set_pose,
                          int* stats6, uint8_t* stop_out) {
    toy_map M(model, stereo_cam, cols, rows, intr5, scale_factor, num_levels, K, kf_id, kf_pose, kf_flags, L, lm_id, lm_pos, lm_erased, O, obs_kf, obs_lm, obs_idx, obs_uv,
              obs_xr, obs_oct);
    auto& kfs = M.kfs;
    auto& lms = M.lms;
    for (int c = 0; c < n_covis; ++c) kfs[curr]->graph_node_->covisibilities_.push_back(covis[c] < 0 ? nullptr : kfs[covis[c]]);
    data::map_database map_db;
    map_db.fixed_keyframe_id_threshold_ = fixed_threshold;
    YAML::Node::forced_bool() = use_additional ? 1 : 0;
    optimize::local_bundle_adjuster_hip ba(YAML::Node(), (unsigned)iters1, (unsigned)iters2);
    YAML::Node::forced_bool() = -1;
    bool stop = stop_in > 0;
    ba.optimize(&map_db, kfs[curr], stop_in >= 0 ? &stop : nullptr);
    *stop_out = stop ? 1 : 0;
    stats6[0] = ba.last_status_;
    stats6[1] = ba.last_stats_.iters_stage1, stats6[2] = ba.last_stats_.iters_stage2, stats6[3] = ba.last_stats_.stage2_entered;
    stats6[4] = ba.last_stats_.num_gated, stats6[5] = ba.last_stats_.lm_trials;
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) kf_pose_out[12 * k + 4 * i + j] = kfs[k]->pose_cw_(i, j);
        kf_set_pose[k] = kfs[k]->num_set_pose_;
    }
    int ne = 0;
    for (int l = 0; l < L; ++l) {
        for (int c = 0; c < 3; ++c) lm_pos_out[3 * l + c] = lms[l]->pos_w_(c);
        lm_counters[4 * l] = lms[l]->num_set_pos_;
        lm_counters[4 * l + 1] = lms[l]->num_update_geometry_;
        lm_counters[4 * l + 2] = lms[l]->num_compute_descriptor_;
        lm_counters[4 * l + 3] = lms[l]->num_erase_observation_;
    }
    for (int k = 0; k < K; ++k)
        for (unsigned id : kfs[k]->erased_landmarks_) {
            int l = -1;
            for (int x = 0; x < L; ++x)
                if (lm_id[x] == id) l = x;
            erased_pairs[2 * ne] = k;
            erased_pairs[2 * ne + 1] = l;
            ++ne;
        }
    *n_erased = ne;
    return 0;
}

// optimize::global_bundle_adjuster_hip::optimize on the map of svref_global_ba (ref_ba_exports.cc), same outputs minus the graph-order arrays
int svref_dropin_global_ba(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, float scale_factor, int num_levels, int K,
                           const unsigned* kf_id, const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos,
                           const uint8_t* lm_erased, int O, const int* obs_kf, const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr,
                           const int* obs_oct, int n_order, const int* kf_order, int num_iter, int use_huber, int stop_in, double* kf_pose_out,
                           double* lm_pos_out, uint8_t* kf_opt, uint8_t* lm_opt, int* lm_iters, uint8_t* stop_out) {
    toy_map M(model, stereo_cam, cols, rows, intr5, scale_factor, num_levels, K, kf_id, kf_pose, kf_flags, L, lm_id, lm_pos, lm_erased, O, obs_kf, obs_lm, obs_idx, obs_uv,
              obs_xr, obs_oct);
    std::vector<std::shared_ptr<data::keyframe>> keyfrms;
    for (int i = 0; i < n_order; ++i) keyfrms.push_back(M.kfs[kf_order[i]]);
    optimize::global_bundle_adjuster_hip gba((unsigned)num_iter, use_huber != 0, false);
    std::unordered_set<unsigned int> okf, olm, omk;
    eigen_alloc_unord_map<unsigned int, Vec3_t> lm_to_pos;
    eigen_alloc_unord_map<unsigned int, Mat44_t> kf_to_pose;
    eigen_alloc_unord_map<unsigned int, std::array<Vec3_t, 4>> mk_to_pos;
    bool stop = stop_in > 0;
    const bool ok = gba.optimize(keyfrms, okf, olm, omk, lm_to_pos, kf_to_pose, mk_to_pos, stop_in >= 0 ? &stop : nullptr);
    *stop_out = stop ? 1 : 0;
    lm_iters[0] = gba.last_stats_.iters_stage1, lm_iters[1] = gba.last_stats_.stopped_by_terminate_action;
    for (int k = 0; k < K; ++k) {
        kf_opt[k] = okf.count(kf_id[k]) ? 1 : 0;
        const Mat44_t T = kf_opt[k] ? kf_to_pose.at(kf_id[k]) : M.kfs[k]->pose_cw_;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) kf_pose_out[12 * k + 4 * i + j] = T(i, j);
    }
    for (int l = 0; l < L; ++l) {
        lm_opt[l] = olm.count(lm_id[l]) ? 1 : 0;
        const Vec3_t p = lm_opt[l] ? lm_to_pos.at(lm_id[l]) : M.lms[l]->pos_w_;
        for (int c = 0; c < 3; ++c) lm_pos_out[3 * l + c] = p(c);
    }
    return ok ? 1 : 0;
}
}
