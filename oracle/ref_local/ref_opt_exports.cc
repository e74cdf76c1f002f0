// The reference's bundle-adjustment graph elements (optimize/internal/landmark_vertex.h, se3/shot_vertex.h, se3/*_reproj_edge.h,
// se3/*_pose_opt_edge.h and the two wrappers that choose the edge per camera model, set its information matrix and Huber width), compiled
// where they lie over the g2o / Eigen stand-ins of shim/ and the reference's REAL camera classes, behind C exports on flat arrays.
// Separate library (oracle/_ref/libsvref_opt.so).  Test infrastructure only (tests/test_ref_local_optimize.py).
#include <cstring>
#include <memory>

#include "stella_vslam/data/frame.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/optimize/pose_optimizer_g2o.h"
#include "stella_vslam/optimize/terminate_action.h"
#include "stella_vslam/optimize/internal/se3/pose_opt_edge_wrapper.h"
#include "stella_vslam/optimize/internal/se3/reproj_edge_wrapper.h"

using namespace stella_vslam;
using namespace stella_vslam::optimize::internal;

#include "ref_pose_hook.h"

namespace {
// what reproj_edge_wrapper<T> reads of its keyframe (reproj_edge_wrapper.h:61)
struct shot_stub {
    camera::base* camera_;
};
// model: 0 perspective, 1 fisheye, 2 equirectangular, 3 radial_division; intr = fx fy cx cy fxb (the edges never read the distortion)
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
g2o::SE3Quat pose_of(const double* q4, const double* t3) {
    g2o::SE3Quat T;
    std::memcpy(T.q, q4, sizeof(T.q));
    std::memcpy(T.t, t3, sizeof(T.t));
    return T;
}
template <int D, int N>
void put(const svref_eigen::Matrix<D, N>& m, double* out) {  // row-major, D rows
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < N; ++j) out[i * N + j] = m(i, j);
}
// information = s * Identity? returns s, or NaN when it is anything else
template <int D>
double info_scale(const svref_eigen::Matrix<D, D>& m) {
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
            if (m(i, j) != (i == j ? m(0, 0) : 0.0)) return std::nan("");
    return m(0, 0);
}
}  // namespace

extern "C" {
// One landmark-shot edge as local / global BA build it (local_bundle_adjuster_g2o.cc:176-183 via reproj_edge_wrapper).
// Outputs: err[D], Ji[D x 3] (d e / d landmark), Jj[D x 6] (d e / d pose), meta = {information scale, Huber delta or -1, chi2, depth_is_positive,
// level after set_as_outlier, level after set_as_inlier}.  Returns D (2 or 3).
int svref_reproj_edge(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, const double* q4, const double* t3, const double* pos_w,
                      const float* uvr, float inv_sigma_sq, float sqrt_chi_sq, int use_huber, double* err, double* Ji, double* Jj, double* meta) {
    auto cam = make(model, stereo_cam, cols, rows, intr5);
    auto shot = std::make_shared<shot_stub>(shot_stub{cam.get()});
    auto lm = std::make_shared<data::landmark>();
    se3::shot_vertex sv;
    sv.setEstimate(pose_of(q4, t3));
    landmark_vertex lv;
    lv.setEstimate(Vec3_t(pos_w[0], pos_w[1], pos_w[2]));
    se3::reproj_edge_wrapper<shot_stub> w(shot, &sv, lm, &lv, 0, uvr[0], uvr[1], uvr[2], inv_sigma_sq, sqrt_chi_sq, use_huber != 0);
    w.edge_->computeError();
    w.edge_->linearizeOplus();
    int D;
    if (auto e = dynamic_cast<se3::mono_perspective_reproj_edge*>(w.edge_)) {
        D = 2, put(e->error(), err), put(e->jacobianOplusXi(), Ji), put(e->jacobianOplusXj(), Jj), meta[0] = info_scale(e->information());
    }
    else if (auto e3 = dynamic_cast<se3::stereo_perspective_reproj_edge*>(w.edge_)) {
        D = 3, put(e3->error(), err), put(e3->jacobianOplusXi(), Ji), put(e3->jacobianOplusXj(), Jj), meta[0] = info_scale(e3->information());
    }
    else {
        auto eq = dynamic_cast<se3::equirectangular_reproj_edge*>(w.edge_);
        D = 2, put(eq->error(), err), put(eq->jacobianOplusXi(), Ji), put(eq->jacobianOplusXj(), Jj), meta[0] = info_scale(eq->information());
    }
    meta[1] = w.edge_->robustKernel() ? w.edge_->robustKernel()->delta() : -1.0;
    meta[2] = w.edge_->chi2();
    meta[3] = w.depth_is_positive() ? 1.0 : 0.0;
    w.set_as_outlier();
    meta[4] = w.edge_->level() + (w.is_outlier() ? 10 : 0);
    w.set_as_inlier();
    meta[5] = w.edge_->level() + (w.is_inlier() ? 10 : 0);
    delete w.edge_;
    return D;
}

// One pose-only edge as the pose optimizer builds it (pose_optimizer_g2o.cc:62-66 via pose_opt_edge_wrapper).  Jj[D x 6]; meta as above.
int svref_pose_opt_edge(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, const double* q4, const double* t3, const double* pos_w,
                        const float* uvr, float inv_sigma_sq, float sqrt_chi_sq, double* err, double* Jj, double* meta) {
    auto cam = make(model, stereo_cam, cols, rows, intr5);
    se3::shot_vertex sv;
    sv.setEstimate(pose_of(q4, t3));
    se3::pose_opt_edge_wrapper w(cam.get(), &sv, Vec3_t(pos_w[0], pos_w[1], pos_w[2]), 0, uvr[0], uvr[1], uvr[2], inv_sigma_sq, sqrt_chi_sq);
    w.edge_->computeError();
    w.edge_->linearizeOplus();
    int D;
    if (auto e = dynamic_cast<se3::mono_perspective_pose_opt_edge*>(w.edge_)) {
        D = 2, put(e->error(), err), put(e->jacobianOplusXi(), Jj), meta[0] = info_scale(e->information());
    }
    else if (auto e3 = dynamic_cast<se3::stereo_perspective_pose_opt_edge*>(w.edge_)) {
        D = 3, put(e3->error(), err), put(e3->jacobianOplusXi(), Jj), meta[0] = info_scale(e3->information());
    }
    else {
        auto eq = dynamic_cast<se3::equirectangular_pose_opt_edge*>(w.edge_);
        D = 2, put(eq->error(), err), put(eq->jacobianOplusXi(), Jj), meta[0] = info_scale(eq->information());
    }
    meta[1] = w.edge_->robustKernel() ? w.edge_->robustKernel()->delta() : -1.0;
    meta[2] = w.edge_->chi2();
    meta[3] = w.depth_is_positive() ? 1.0 : 0.0;
    w.set_as_outlier();
    meta[4] = w.edge_->level() + (w.is_outlier() ? 10 : 0);
    w.set_as_inlier();
    meta[5] = w.edge_->level() + (w.is_inlier() ? 10 : 0);
    delete w.edge_;
    return D;
}

// shot_vertex::oplusImpl (shot_vertex.h:52-55) and landmark_vertex::oplusImpl (landmark_vertex.h:49-52), then setToOriginImpl
void svref_vertex_oplus(const double* q4, const double* t3, const double* upd6, double* q4_out, double* t3_out, const double* pos, const double* upd3, double* pos_out,
                        double* origin7_3) {
    se3::shot_vertex sv;
    sv.setEstimate(pose_of(q4, t3));
    sv.oplus(upd6);
    std::memcpy(q4_out, sv.estimate().q, 4 * sizeof(double));
    std::memcpy(t3_out, sv.estimate().t, 3 * sizeof(double));
    landmark_vertex lv;
    lv.setEstimate(Vec3_t(pos[0], pos[1], pos[2]));
    lv.oplus(upd3);
    for (int i = 0; i < 3; ++i) pos_out[i] = lv.estimate()(i);
    sv.setToOriginImpl();
    lv.setToOriginImpl();
    std::memcpy(origin7_3, sv.estimate().q, 4 * sizeof(double));
    std::memcpy(origin7_3 + 4, sv.estimate().t, 3 * sizeof(double));
    for (int i = 0; i < 3; ++i) origin7_3[7 + i] = lv.estimate()(i);
}

// optimize/terminate_action.cc driven over a scripted optimizer: step k reports chi2[k] as the active robust chi2 of iteration[k] (< 0: the
// "reset" call).  The caller's force-stop flag (install_flag) or the action's own one is cleared before every step; raised[k] = the flag
// after the step, last_chi[k] = _lastChi, by_action[k] = stopped_by_terminate_action_.
void svref_terminate_sequence(int n, const int* iteration, const double* chi2, double gain_thr, int install_flag, uint8_t* raised, double* last_chi,
                              uint8_t* by_action) {
    optimize::terminate_action act;
    act.setGainThreshold(gain_thr);
    g2o::SparseOptimizer opt;
    bool flag = false;
    if (install_flag) opt.setForceStopFlag(&flag);
    g2o::HyperGraphAction* a = &act;  // operator() is private in the derived class, public in the base
    for (int k = 0; k < n; ++k) {
        if (opt.forceStopFlag()) *opt.forceStopFlag() = false;
        opt.scripted_chi2 = chi2[k];
        g2o::HyperGraphAction::ParametersIteration p(iteration[k]);
        (*a)(&opt, &p);
        raised[k] = opt.forceStopFlag() && *opt.forceStopFlag() ? 1 : 0;
        last_chi[k] = act.lastChi();
        by_action[k] = act.stopped_by_terminate_action_ ? 1 : 0;
    }
}

// optimize/pose_optimizer_g2o.cc, compiled from the reference: edge creation (null / erased landmarks skipped, fewer than five -> 0),
// the rounds of optimize(num_each_iter) + chi-square classification, the removal of the kernels, the early exit and the return value are
// the reference's code; optimizer.optimize() itself -- g2o -- is the oracle's pose-only Levenberg-Marquardt with its terminate rule
// (orc_dbg_pose_lm), whose _lastChi and stop flag live in the stand-in optimizer; reset_each_round = g2o sending the "iteration -1" call.
// lm_state per keypoint: 0 = no landmark, 1 = landmark, 2 = landmark that will be erased.  overload: 0 = frame, 1 = keyframe, 2 = the flat one.
int svref_pose_optimize(int model, int stereo_cam, unsigned cols, unsigned rows, const double* intr5, const double* pose_cw12, int n_kp,
                        const float* kp_xy, const int* octave, const float* x_right, const double* pos_w, const uint8_t* lm_state,
                        float scale_factor, int num_levels, int trials_robust, int trials, int each_iter, int reset_each_round, int overload,
                        double* pose_out12, uint8_t* outlier_flags, int* lm_iterations) {
    auto cam = make(model, stereo_cam, cols, rows, intr5);
    feature::orb_params orb("ref", scale_factor, num_levels, 20, 7);
    data::keyframe frm;
    frm.pose_cw_ = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) frm.pose_cw_(i, j) = pose_cw12[4 * i + j];
    frm.orb_params_ = &orb;
    frm.camera_ = cam.get();
    for (int i = 0; i < n_kp; ++i) {
        cv::KeyPoint kp;
        kp.pt.x = kp_xy[2 * i];
        kp.pt.y = kp_xy[2 * i + 1];
        kp.octave = octave[i];
        frm.frm_obs_.undist_keypts_.push_back(kp);
        if (x_right) frm.frm_obs_.stereo_x_right_.push_back(x_right[i]);
        frm.landmarks_.push_back(lm_state[i] ? std::make_shared<data::landmark>(Vec3_t(pos_w[3 * i], pos_w[3 * i + 1], pos_w[3 * i + 2]), lm_state[i] == 2)
                                             : std::shared_ptr<data::landmark>());
    }
    int total_iters = 0;
    g2o::SparseOptimizer::default_hook() = svref::make_pose_lm_hook(intr5, reset_each_round, &total_iters);
    optimize::pose_optimizer_g2o opt((unsigned)trials_robust, (unsigned)trials, (unsigned)each_iter);
    Mat44_t out = frm.pose_cw_;
    std::vector<bool> flags;
    unsigned int valid;
    if (overload == 0) valid = opt.optimize(static_cast<const data::frame&>(frm), out, flags);
    else if (overload == 1) valid = opt.optimize(&frm, out, flags);
    else valid = opt.optimize(frm.pose_cw_, frm.frm_obs_, frm.orb_params_, frm.camera_, frm.landmarks_, out, flags);
    g2o::SparseOptimizer::default_hook() = nullptr;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) pose_out12[4 * i + j] = out(i, j);
    for (int i = 0; i < n_kp; ++i) outlier_flags[i] = i < (int)flags.size() && flags[i] ? 1 : 0;
    if (lm_iterations) *lm_iterations = total_iters;
    return (int)valid;
}
}
