// g2o's optimize() for the reference's pose_optimizer_g2o.cc, played by the oracle's pose-only Levenberg-Marquardt (orc_dbg_pose_lm) on the graph
// the reference built: the edges' measurements, information, levels and kernels are read back from the stand-in optimizer (shim/g2o), the pose
// goes back into its vertex.  Shared by the pose-optimizer fixture (ref_opt_exports.cc) and the tracker fixture (ref_trk_exports.cc).
// Test infrastructure only.
#ifndef SVREF_POSE_HOOK_H
#define SVREF_POSE_HOOK_H
#include <cstdint>
#include <functional>
#include <vector>

#include <g2o/core/sparse_optimizer.h>
#include <g2o/core/sparse_optimizer_terminate_action.h>

#include "stella_vslam/optimize/internal/se3/pose_opt_edge_wrapper.h"

extern "C" int orc_dbg_pose_lm(double* q4, double* t3, int n, const double* pos_w, const float* uvr, const float* inv_sigma_sq, const float* huber_delta,
                               const double* intr, const uint8_t* level, const uint8_t* robust, int num_each_iter, double gain_thr, uint8_t* flag,
                               double* last_chi);

namespace svref {
// intr5 = fx fy cx cy fxb of the camera (overwritten per edge by what the edge itself carries); reset_each_round = g2o sending the "iteration -1"
// call at the start of every optimize(); *total_iters accumulates the LM iterations
inline std::function<int(g2o::SparseOptimizer&, int)> make_pose_lm_hook(const double* intr5, int reset_each_round, int* total_iters) {
    using namespace stella_vslam::optimize::internal;
    return [intr5, reset_each_round, total_iters](g2o::SparseOptimizer& o, int iters) -> int {
        auto* v = dynamic_cast<se3::shot_vertex*>(o.vertices().at(0));
        const size_t n = o.edges().size();
        std::vector<double> pw(3 * n);
        std::vector<float> uvr(3 * n), w(n), hub(n);
        std::vector<uint8_t> level(n), robust(n);
        double K[5] = {intr5[0], intr5[1], intr5[2], intr5[3], intr5[4]};
        for (size_t k = 0; k < n; ++k) {
            auto* e = o.edges()[k];
            level[k] = e->level() != 0;
            robust[k] = e->robustKernel() != nullptr;
            hub[k] = robust[k] ? (float)e->robustKernel()->delta() : 0.f;
            if (auto* m = dynamic_cast<se3::mono_perspective_pose_opt_edge*>(e)) {
                for (int c = 0; c < 3; ++c) pw[3 * k + c] = m->pos_w_(c);
                uvr[3 * k] = (float)m->measurement()(0), uvr[3 * k + 1] = (float)m->measurement()(1), uvr[3 * k + 2] = -1.f;
                w[k] = (float)m->information()(0, 0);
                K[0] = m->fx_, K[1] = m->fy_, K[2] = m->cx_, K[3] = m->cy_;
            }
            else if (auto* s3 = dynamic_cast<se3::stereo_perspective_pose_opt_edge*>(e)) {
                for (int c = 0; c < 3; ++c) pw[3 * k + c] = s3->pos_w_(c);
                for (int c = 0; c < 3; ++c) uvr[3 * k + c] = (float)s3->measurement()(c);
                w[k] = (float)s3->information()(0, 0);
                K[0] = s3->fx_, K[1] = s3->fy_, K[2] = s3->cx_, K[3] = s3->cy_, K[4] = s3->focal_x_baseline_;
            }
            else {
                auto* q = dynamic_cast<se3::equirectangular_pose_opt_edge*>(e);
                for (int c = 0; c < 3; ++c) pw[3 * k + c] = q->pos_w_(c);
                uvr[3 * k] = (float)q->measurement()(0), uvr[3 * k + 1] = (float)q->measurement()(1), uvr[3 * k + 2] = -1.f;
                w[k] = (float)q->information()(0, 0);
                K[0] = 0, K[1] = 0, K[2] = q->cols_, K[3] = q->rows_, K[4] = 0;
            }
        }
        double gain_thr = 1e-6;
        for (auto* a : o.post_iteration_actions)
            if (auto* t = dynamic_cast<g2o::SparseOptimizerTerminateAction*>(a)) gain_thr = t->gainThreshold();
        if (reset_each_round) o.lm_stop = 0;
        g2o::SE3Quat T = v->estimate();
        const int it = orc_dbg_pose_lm(T.q, T.t, (int)n, pw.data(), uvr.data(), w.data(), hub.data(), K, level.data(), robust.data(), iters, gain_thr,
                                       &o.lm_stop, &o.lm_last_chi);
        v->setEstimate(T);
        for (auto* e : o.edges())
            if (e->level() == o.active_level) e->computeError();  // what the last computeActiveErrors of the run leaves behind
        if (total_iters) *total_iters += it;
        return it;
    };
}
}  // namespace svref
#endif
