// optimize/local_bundle_adjuster_g2o.cc, compiled from the reference over the stand-in data:: headers of shim_ba/ and the g2o stand-in of
// shim/g2o (the optimizer as a container with a hook): the gather of local / fixed keyframes and local landmarks, the vertex and edge
// construction, the two-stage schedule with its chi-square / depth gate and kernel removal, the outlier list and the write-back are the
// reference's code; SparseOptimizer::optimize() -- g2o -- is played by the oracle's Levenberg-Marquardt (orc_dbg_ba_lm) on the graph the
// reference built, in the order it built it.  Separate library (oracle/_ref/libsvref_ba.so).  Test infrastructure only
// (tests/test_ref_local_ba.py).
#include <cstring>
#include <memory>
#include <array>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "stella_vslam/camera/equirectangular.h"
#include "stella_vslam/camera/fisheye.h"
#include "stella_vslam/camera/perspective.h"
#include "stella_vslam/camera/radial_division.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/data/map_database.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/optimize/internal/landmark_vertex.h"
#include "stella_vslam/optimize/internal/se3/reproj_edge_wrapper.h"
#include "stella_vslam/optimize/internal/se3/shot_vertex.h"
#include "stella_vslam/optimize/global_bundle_adjuster.h"
#include "stella_vslam/optimize/local_bundle_adjuster_g2o.h"
#include "stella_vslam/optimize/terminate_action.h"
#include "stella_vslam/util/converter.h"

#include <g2o/core/sparse_optimizer.h>
#include <yaml-cpp/yaml.h>

using namespace stella_vslam;
using namespace stella_vslam::optimize::internal;

std::mutex stella_vslam::data::map_database::mtx_database_;

extern "C" int orc_dbg_ba_lm(int P, int L, int E, double* q4, double* t3, const uint8_t* pose_fixed, double* pts, const uint8_t* point_fixed,
                             const int32_t* obs_pose, const int32_t* obs_point, const float* obs_uvr, const float* obs_inv_sigma_sq,
                             const float* obs_huber, const uint8_t* level, const uint8_t* robust, const double* intr, int iterations, double gain_thr,
                             uint8_t* stop, double* err);

namespace {
std::unique_ptr<camera::base> make(int model, int stereo, unsigned cols, unsigned rows, const double* k) {
    const auto setup = stereo ? camera::setup_type_t::Stereo : camera::setup_type_t::Monocular;
    const auto col = camera::color_order_t::Gray;
    switch (model) {
        case 0: return std::unique_ptr<camera::base>(new camera::perspective("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, 0, k[4]));
        case 1: return std::unique_ptr<camera::base>(new camera::fisheye("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, 0, 0, 0, k[4]));
        case 2: return std::unique_ptr<camera::base>(new camera::equirectangular("ref", col, cols, rows, 30.0));
        default: return std::unique_ptr<camera::base>(new camera::radial_division("ref", setup, col, cols, rows, 30.0, k[0], k[1], k[2], k[3], 0, k[4]));
    }
}
}  // namespace

namespace {
struct toy_map {
    std::unique_ptr<camera::base> cam;
    std::unique_ptr<feature::orb_params> orb;
    std::vector<std::shared_ptr<data::keyframe>> kfs;
    std::vector<std::shared_ptr<data::landmark>> lms;
    data::map_database map_db;
    int model = 0;
    unsigned cols = 0, rows = 0;
    double intr5[5] = {0, 0, 0, 0, 0};
    toy_map(int model_, int stereo_cam, unsigned cols_, unsigned rows_, const double* intr, float scale_factor, int num_levels, int K, const unsigned* kf_id,
            const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos, const uint8_t* lm_erased, int O,
            const int* obs_kf, const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr, const int* obs_oct)
        : cam(make(model_, stereo_cam, cols_, rows_, intr)), orb(new feature::orb_params("ref", scale_factor, num_levels, 20, 7)), kfs(K), lms(L), model(model_),
          cols(cols_), rows(rows_) {
        memcpy(intr5, intr, sizeof(intr5));
        for (int k = 0; k < K; ++k) {
            kfs[k] = std::make_shared<data::keyframe>();
            kfs[k]->id_ = kf_id[k];
            kfs[k]->camera_ = cam.get();
            kfs[k]->orb_params_ = orb.get();
            kfs[k]->pose_cw_ = Mat44_t::Identity();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) kfs[k]->pose_cw_(i, j) = kf_pose[12 * k + 4 * i + j];
            kfs[k]->will_be_erased_ = kf_flags[k] & 1;
            kfs[k]->graph_node_->is_spanning_root_ = (kf_flags[k] & 2) != 0;
        }
        for (int l = 0; l < L; ++l)
            lms[l] = std::make_shared<data::landmark>(lm_id[l], Vec3_t(lm_pos[3 * l], lm_pos[3 * l + 1], lm_pos[3 * l + 2]), lm_erased[l] != 0);
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

// g2o's optimize() for the graphs of the two bundle adjusters: the oracle's Levenberg-Marquardt on the vertices / edges the reference
// added, in the order it added them.  The first call records that order (vertices identified by their estimates: the fixtures' poses and
// positions are all different).
struct ba_hook {
    const toy_map& M;
    double gain_thr;
    int* counts;
    int *pose_order, *point_order, *edge_order, *lm_iters;
    int stage = 0;
    bool recorded = false;
    int operator()(g2o::SparseOptimizer& o, int iters) {
        const int K = (int)M.kfs.size(), L = (int)M.lms.size();
        std::vector<se3::shot_vertex*> pv;
        std::vector<landmark_vertex*> lv;
        std::unordered_map<const g2o::OptimizableGraph::Vertex*, int> vidx;
        for (auto* v : o.vertices()) {
            if (auto* s = dynamic_cast<se3::shot_vertex*>(v)) {
                vidx[v] = (int)pv.size();
                pv.push_back(s);
            }
            else if (auto* l = dynamic_cast<landmark_vertex*>(v)) {
                vidx[v] = (int)lv.size();
                lv.push_back(l);
            }
        }
        const int P = (int)pv.size(), Lc = (int)lv.size(), E = (int)o.edges().size();
        std::vector<double> q(4 * (size_t)P + 4), t(3 * (size_t)P + 3), pts(3 * (size_t)Lc + 3), intr(5 * (size_t)P + 5), err(3 * (size_t)E + 3, 0.0);
        std::vector<uint8_t> pfix(P + 1), lfix(Lc + 1), level(E + 1), robust(E + 1);
        std::vector<int32_t> op(E + 1), ol(E + 1);
        std::vector<float> uvr(3 * (size_t)E + 3), w(E + 1), hub(E + 1);
        for (int p = 0; p < P; ++p) {
            memcpy(&q[4 * p], pv[p]->estimate().q, 4 * sizeof(double));
            memcpy(&t[3 * p], pv[p]->estimate().t, 3 * sizeof(double));
            pfix[p] = pv[p]->fixed();
            for (int c = 0; c < 5; ++c) intr[5 * p + c] = M.model == 2 ? (c == 2 ? (double)M.cols : c == 3 ? (double)M.rows : 0.0) : M.intr5[c];
        }
        for (int l = 0; l < Lc; ++l) {
            for (int c = 0; c < 3; ++c) pts[3 * l + c] = lv[l]->estimate()(c);
            lfix[l] = lv[l]->fixed();
        }
        for (int e = 0; e < E; ++e) {
            auto* ed = o.edges()[e];
            ol[e] = vidx.at(ed->vertex(0));
            op[e] = vidx.at(ed->vertex(1));
            level[e] = ed->level() != 0;
            robust[e] = ed->robustKernel() != nullptr;
            hub[e] = robust[e] ? (float)ed->robustKernel()->delta() : 0.f;
            if (auto* m = dynamic_cast<se3::mono_perspective_reproj_edge*>(ed)) {
                uvr[3 * e] = (float)m->measurement()(0), uvr[3 * e + 1] = (float)m->measurement()(1), uvr[3 * e + 2] = -1.f;
                w[e] = (float)m->information()(0, 0);
                for (int c = 0; c < 2; ++c) err[3 * e + c] = m->error()(c);
            }
            else if (auto* s3 = dynamic_cast<se3::stereo_perspective_reproj_edge*>(ed)) {
                for (int c = 0; c < 3; ++c) uvr[3 * e + c] = (float)s3->measurement()(c), err[3 * e + c] = s3->error()(c);
                w[e] = (float)s3->information()(0, 0);
            }
            else {
                auto* qe = dynamic_cast<se3::equirectangular_reproj_edge*>(ed);
                uvr[3 * e] = (float)qe->measurement()(0), uvr[3 * e + 1] = (float)qe->measurement()(1), uvr[3 * e + 2] = -1.f;
                w[e] = (float)qe->information()(0, 0);
                for (int c = 0; c < 2; ++c) err[3 * e + c] = qe->error()(c);
            }
        }
        if (!recorded) {
            recorded = true;
            counts[0] = P, counts[1] = Lc, counts[2] = E;
            for (int p = 0; p < P; ++p) {
                int found = -1;
                for (int k = 0; k < K && found < 0; ++k) {
                    const g2o::SE3Quat Tk = util::converter::to_g2o_SE3(M.kfs[k]->pose_cw_);
                    if (!memcmp(Tk.q, &q[4 * p], 4 * sizeof(double)) && !memcmp(Tk.t, &t[3 * p], 3 * sizeof(double))) found = k;
                }
                pose_order[p] = found | (pfix[p] ? 1 << 30 : 0);
            }
            for (int l = 0; l < Lc; ++l) {
                int found = -1;
                for (int k = 0; k < L && found < 0; ++k)
                    if (M.lms[k]->pos_w_(0) == pts[3 * l] && M.lms[k]->pos_w_(1) == pts[3 * l + 1] && M.lms[k]->pos_w_(2) == pts[3 * l + 2]) found = k;
                point_order[l] = found;
            }
            for (int e = 0; e < E; ++e) {
                edge_order[2 * e] = pose_order[op[e]] & ~(1 << 30);
                edge_order[2 * e + 1] = point_order[ol[e]];
            }
        }
        const bool before = o.forceStopFlag() && *o.forceStopFlag();
        uint8_t stop = before ? 1 : 0;
        const int it = orc_dbg_ba_lm(P, Lc, E, q.data(), t.data(), pfix.data(), pts.data(), lfix.data(), op.data(), ol.data(), uvr.data(), w.data(), hub.data(),
                                     level.data(), robust.data(), intr.data(), iters, gain_thr, &stop, err.data());
        if (stop && !before) {  // raised by the gain rule: the terminate action writes through the optimizer's pointer and remembers it did
            if (o.forceStopFlag()) *o.forceStopFlag() = true;
            for (auto* a : o.post_iteration_actions)
                if (auto* ta = dynamic_cast<optimize::terminate_action*>(a)) ta->stopped_by_terminate_action_ = true;
        }
        for (int p = 0; p < P; ++p) {
            g2o::SE3Quat T;
            memcpy(T.q, &q[4 * p], 4 * sizeof(double));
            memcpy(T.t, &t[3 * p], 3 * sizeof(double));
            pv[p]->setEstimate(T);
        }
        for (int l = 0; l < Lc; ++l) lv[l]->setEstimate(Vec3_t(pts[3 * l], pts[3 * l + 1], pts[3 * l + 2]));
        for (auto* ed : o.edges())
            if (ed->level() == o.active_level) ed->computeError();  // what the last computeActiveErrors of the run leaves on the active edges
        if (stage < 2) lm_iters[stage] = it;
        ++stage;
        return it;
    }
};
}  // namespace

extern "C" {
// The toy map: K keyframes (id, pose 3x4 row-major, flags: 1 = will be erased, 2 = spanning root), L landmarks (id, position, erased), O
// observations (keyframe index, landmark index, keypoint index in the keyframe, u, v, x_right or -1, octave), the covisibility list of the
// current keyframe (indices, -1 = a null pointer), the map's fixed-keyframe id threshold.  One camera for all keyframes.
// Outputs: the optimizer's vertex / edge order as the reference built it (pose_order: keyframe index | fixed << 30; point_order: landmark
// index; edge_order: keyframe index, landmark index), the map after the call (poses, positions), the erased observations
// (keyframe index, landmark index), per-landmark and per-keyframe call counters, LM iterations of the two stages.
int svref_local_ba(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, float scale_factor, int num_levels, int K,
                   const unsigned* kf_id, const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos,
                   const uint8_t* lm_erased, int O, const int* obs_kf, const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr,
                   const int* obs_oct, int curr, int n_covis, const int* covis, unsigned fixed_threshold, int use_additional, int iters1, int iters2,
                   int stop_in, int* counts /* 3: poses, points, edges */, int* pose_order, int* point_order, int* edge_order, double* kf_pose_out,
                   double* lm_pos_out, int* n_erased, int* erased_pairs, int* lm_counters /* L x 4 */, int* kf_set_pose /* K */, int* lm_iters /* 2 */,
                   uint8_t* stop_out) {
    toy_map M(model, stereo_cam, cols, rows, intr5, scale_factor, num_levels, K, kf_id, kf_pose, kf_flags, L, lm_id, lm_pos, lm_erased, O, obs_kf, obs_lm, obs_idx,
              obs_uv, obs_xr, obs_oct);
    for (int c = 0; c < n_covis; ++c) M.kfs[curr]->graph_node_->covisibilities_.push_back(covis[c] < 0 ? nullptr : M.kfs[covis[c]]);
    M.map_db.fixed_keyframe_id_threshold_ = fixed_threshold;
    lm_iters[0] = lm_iters[1] = 0;
    counts[0] = counts[1] = counts[2] = 0;
    ba_hook hook{M, 1e-3, counts, pose_order, point_order, edge_order, lm_iters};
    g2o::SparseOptimizer::default_hook() = [&](g2o::SparseOptimizer& o, int iters) { return hook(o, iters); };
    YAML::Node::forced_bool() = use_additional ? 1 : 0;
    optimize::local_bundle_adjuster_g2o ba(YAML::Node(), (unsigned)iters1, (unsigned)iters2);
    YAML::Node::forced_bool() = -1;
    bool stop = stop_in > 0;
    ba.optimize(&M.map_db, M.kfs[curr], stop_in >= 0 ? &stop : nullptr);
    g2o::SparseOptimizer::default_hook() = nullptr;
    *stop_out = stop ? 1 : 0;
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) kf_pose_out[12 * k + 4 * i + j] = M.kfs[k]->pose_cw_(i, j);
        kf_set_pose[k] = M.kfs[k]->num_set_pose_;
    }
    int ne = 0;
    for (int l = 0; l < L; ++l) {
        for (int c = 0; c < 3; ++c) lm_pos_out[3 * l + c] = M.lms[l]->pos_w_(c);
        lm_counters[4 * l] = M.lms[l]->num_set_pos_;
        lm_counters[4 * l + 1] = M.lms[l]->num_update_geometry_;
        lm_counters[4 * l + 2] = M.lms[l]->num_compute_descriptor_;
        lm_counters[4 * l + 3] = M.lms[l]->num_erase_observation_;
    }
    for (int k = 0; k < K; ++k)
        for (unsigned id : M.kfs[k]->erased_landmarks_) {
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

// optimize/global_bundle_adjuster.cc: optimize() over the keyframes in the given order (kf_order: indices; the landmarks are those the
// keyframes hold, first seen first).  Returns what the reference returns (0 = discarded: stopped by the caller, not by the gain rule);
// kf_opt / lm_opt mark the ids in the two "optimized" sets, kf_pose_out / lm_pos_out hold the values of the two result maps.
int svref_global_ba(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, float scale_factor, int num_levels, int K,
                    const unsigned* kf_id, const double* kf_pose, const uint8_t* kf_flags, int L, const unsigned* lm_id, const double* lm_pos,
                    const uint8_t* lm_erased, int O, const int* obs_kf, const int* obs_lm, const int* obs_idx, const float* obs_uv, const float* obs_xr,
                    const int* obs_oct, int n_order, const int* kf_order, int num_iter, int use_huber, int stop_in, int* counts, int* pose_order,
                    int* point_order, int* edge_order, double* kf_pose_out, double* lm_pos_out, uint8_t* kf_opt, uint8_t* lm_opt, int* lm_iters,
                    uint8_t* stop_out) {
    toy_map M(model, stereo_cam, cols, rows, intr5, scale_factor, num_levels, K, kf_id, kf_pose, kf_flags, L, lm_id, lm_pos, lm_erased, O, obs_kf, obs_lm, obs_idx,
              obs_uv, obs_xr, obs_oct);
    lm_iters[0] = lm_iters[1] = 0;
    counts[0] = counts[1] = counts[2] = 0;
    ba_hook hook{M, 1e-3, counts, pose_order, point_order, edge_order, lm_iters};
    g2o::SparseOptimizer::default_hook() = [&](g2o::SparseOptimizer& o, int iters) { return hook(o, iters); };
    std::vector<std::shared_ptr<data::keyframe>> keyfrms;
    for (int i = 0; i < n_order; ++i) keyfrms.push_back(M.kfs[kf_order[i]]);
    optimize::global_bundle_adjuster gba((unsigned)num_iter, use_huber != 0, false);
    std::unordered_set<unsigned int> okf, olm, omk;
    eigen_alloc_unord_map<unsigned int, Vec3_t> lm_to_pos;
    eigen_alloc_unord_map<unsigned int, Mat44_t> kf_to_pose;
    eigen_alloc_unord_map<unsigned int, std::array<Vec3_t, 4>> mk_to_pos;
    bool stop = stop_in > 0;
    const bool ok = gba.optimize(keyfrms, okf, olm, omk, lm_to_pos, kf_to_pose, mk_to_pos, stop_in >= 0 ? &stop : nullptr);
    g2o::SparseOptimizer::default_hook() = nullptr;
    *stop_out = stop ? 1 : 0;
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
