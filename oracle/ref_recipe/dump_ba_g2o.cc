// Pins the Levenberg-Marquardt TRAJECTORY of local bundle adjustment against REAL g2o (what oracle/ref_local cannot reach: there g2o's
// optimize() is played by the oracle's own LM).  Builds the graph of optimize/local_bundle_adjuster_g2o.cc:149-348 for a flat monocular
// scene -- the reference's own vertex / edge classes (optimize/internal/se3/shot_vertex.h, landmark_vertex.h,
// se3/perspective_reproj_edge.h, compiled where they lie), its terminate_action, the solver stack BlockSolver_6_3 + LinearSolverEigen +
// OptimizationAlgorithmLevenberg, Huber kernels of width sqrt(5.991), five iterations, the chi-square / depth gate, ten iterations -- and
// writes the optimised poses, points, outlier flags and iteration counts as .npy.  Compiles only where g2o (20230223_git) and Eigen exist.
//   tools/export_ba_scene.py oracle/_ref/fixtures/ba_config3.bin            (the seeded config-3 scene of stella_vslam_amd/synthetic.py)
//   oracle/_ref/build/dump_ba_g2o oracle/_ref/fixtures/ba_config3.bin oracle/_ref/fixtures
//   python oracle/ref_recipe/pack_npz.py oracle/_ref/fixtures tests/golden   -> tests/golden/ref_ba_config3.npz
// File format of the scene (little endian): int32 P, L, E; then pose_cw P x 12 f64 (row-major [R|t]), pose_fixed P u8, points L x 3 f64,
// obs_pose E i32, obs_point E i32, obs_uvr E x 3 f32, obs_inv_sigma_sq E f32, intr P x 5 f64 (fx fy cx cy fx*b).
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include <g2o/core/block_solver.h>
#include <g2o/core/optimization_algorithm_levenberg.h>
#include <g2o/core/robust_kernel_impl.h>
#include <g2o/core/sparse_optimizer.h>
#include <g2o/solvers/eigen/linear_solver_eigen.h>

#include "stella_vslam/optimize/internal/landmark_vertex.h"
#include "stella_vslam/optimize/internal/se3/perspective_reproj_edge.h"
#include "stella_vslam/optimize/internal/se3/shot_vertex.h"
#include "stella_vslam/optimize/terminate_action.h"
#include "stella_vslam/util/converter.h"

namespace {
void write_npy(const std::string& path, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) hdr += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    hdr += "), }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[] = "\x93NUMPY\x01\x00";
    f.write(magic, 8);
    const uint16_t n = (uint16_t)hdr.size();
    f.write((const char*)&n, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char*)data, bytes);
}
template <class T>
std::vector<T> rd(std::ifstream& f, size_t n) {
    std::vector<T> v(n);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
    return v;
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: dump_ba_g2o <scene.bin> <out dir>\n");
        return 2;
    }
    using namespace stella_vslam;
    using namespace stella_vslam::optimize;
    std::ifstream f(argv[1], std::ios::binary);
    const auto hdr = rd<int32_t>(f, 3);
    const int P = hdr[0], L = hdr[1], E = hdr[2];
    const auto pose = rd<double>(f, (size_t)P * 12);
    const auto fixed = rd<uint8_t>(f, P);
    const auto points = rd<double>(f, (size_t)L * 3);
    const auto obs_pose = rd<int32_t>(f, E);
    const auto obs_point = rd<int32_t>(f, E);
    const auto uvr = rd<float>(f, (size_t)E * 3);
    const auto inv_sigma_sq = rd<float>(f, E);
    const auto intr = rd<double>(f, (size_t)P * 5);
    if (!f) {
        std::fprintf(stderr, "short scene file\n");
        return 2;
    }
    // local_bundle_adjuster_g2o.cc:151-160
    auto linear_solver = std::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>>();
    auto block_solver = std::make_unique<g2o::BlockSolver_6_3>(std::move(linear_solver));
    auto algorithm = new g2o::OptimizationAlgorithmLevenberg(std::move(block_solver));
    g2o::SparseOptimizer optimizer;
    auto terminateAction = new terminate_action;
    terminateAction->setGainThreshold(1e-3);
    optimizer.addPostIterationAction(terminateAction);
    optimizer.setAlgorithm(algorithm);
    bool force_stop_flag = false;  // the caller's abort_local_BA_: terminate_action writes through it (terminate_action.cc:36-76)
    optimizer.setForceStopFlag(&force_stop_flag);
    // vertices: keyframes first, then landmarks (:166-201); vertex id = running counter, as the containers number them
    std::vector<internal::se3::shot_vertex*> kf_vtx(P);
    std::vector<internal::landmark_vertex*> lm_vtx(L);
    int next_id = 0;
    for (int p = 0; p < P; ++p) {
        Mat44_t T = Mat44_t::Identity();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) T(i, j) = pose[(size_t)p * 12 + 4 * i + j];
        auto v = new internal::se3::shot_vertex();
        v->setId(next_id++);
        v->setEstimate(util::converter::to_g2o_SE3(T));
        v->setFixed(fixed[p] != 0);
        optimizer.addVertex(v);
        kf_vtx[p] = v;
    }
    for (int l = 0; l < L; ++l) {
        auto v = new internal::landmark_vertex();
        v->setId(next_id++);
        v->setEstimate(Vec3_t(points[3 * (size_t)l], points[3 * (size_t)l + 1], points[3 * (size_t)l + 2]));
        v->setFixed(false);
        v->setMarginalized(true);
        optimizer.addVertex(v);
        lm_vtx[l] = v;
    }
    // edges in observation order (:204-244): Huber width sqrt(chi_sq_2D), information = inv_sigma_sq * I
    constexpr float chi_sq_2D = 5.99146;
    const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D);
    std::vector<internal::se3::mono_perspective_reproj_edge*> edges(E);
    for (int e = 0; e < E; ++e) {
        auto edge = new internal::se3::mono_perspective_reproj_edge();
        edge->setMeasurement(Vec2_t{uvr[3 * (size_t)e], uvr[3 * (size_t)e + 1]});
        edge->setInformation(Mat22_t::Identity() * inv_sigma_sq[e]);
        const double* K = &intr[(size_t)obs_pose[e] * 5];
        edge->fx_ = K[0], edge->fy_ = K[1], edge->cx_ = K[2], edge->cy_ = K[3];
        edge->setVertex(0, lm_vtx[obs_point[e]]);
        edge->setVertex(1, kf_vtx[obs_pose[e]]);
        auto huber_kernel = new g2o::RobustKernelHuber();
        huber_kernel->setDelta(sqrt_chi_sq_2D);
        edge->setRobustKernel(huber_kernel);
        optimizer.addEdge(edge);
        edges[e] = edge;
    }
    // :306-348
    optimizer.initializeOptimization();
    const int it1 = optimizer.optimize(5);
    const bool flag_after_stage1 = force_stop_flag;
    std::vector<uint8_t> outlier(E, 0);
    int it2 = 0, gated = 0;
    if (!force_stop_flag) {
        for (int e = 0; e < E; ++e) {
            if (chi_sq_2D < edges[e]->chi2() || !edges[e]->mono_perspective_reproj_edge::depth_is_positive()) {
                edges[e]->setLevel(1);
                ++gated;
            }
            edges[e]->setRobustKernel(nullptr);
        }
        optimizer.initializeOptimization();
        it2 = optimizer.optimize(10);
    }
    for (int e = 0; e < E; ++e)  // :352-389
        outlier[e] = (chi_sq_2D < edges[e]->chi2() || !edges[e]->mono_perspective_reproj_edge::depth_is_positive()) ? 1 : 0;
    std::vector<double> pose_out((size_t)P * 12), pts_out((size_t)L * 3);
    for (int p = 0; p < P; ++p) {
        const Mat44_t T = util::converter::to_eigen_mat(kf_vtx[p]->estimate());
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) pose_out[(size_t)p * 12 + 4 * i + j] = T(i, j);
    }
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < 3; ++k) pts_out[3 * (size_t)l + k] = lm_vtx[l]->estimate()(k);
    const int32_t stats[5] = {it1, it2, gated, flag_after_stage1 ? 1 : 0, force_stop_flag ? 1 : 0};
    const std::string dir = argv[2];
    write_npy(dir + "/ref_ba_config3_pose_cw.npy", "<f8", {(size_t)P, 12}, pose_out.data(), pose_out.size() * 8);
    write_npy(dir + "/ref_ba_config3_points.npy", "<f8", {(size_t)L, 3}, pts_out.data(), pts_out.size() * 8);
    write_npy(dir + "/ref_ba_config3_outlier.npy", "|u1", {(size_t)E}, outlier.data(), outlier.size());
    write_npy(dir + "/ref_ba_config3_stats.npy", "<i4", {5}, stats, sizeof stats);
    std::printf("g2o: %d + %d iterations, %d gated, flag after stage 1: %d\n", it1, it2, gated, (int)flag_after_stage1);
    delete terminateAction;
    return 0;
}
