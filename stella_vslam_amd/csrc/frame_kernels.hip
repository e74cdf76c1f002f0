// Frame observation + landmark reprojection kernels (one thread per keypoint / landmark, fp64 as the reference).
// Every expression keeps the reference's operation order (fp64, no contraction), so the CPU restatement used by the tests
// reproduces it bit for bit for the models built from + - * / sqrt only (perspective, radial_division, fisheye's projective part);
// fisheye's tan and the equirectangular asin / atan2 / sin / cos go through the device math library and agree to the last
// few ulps of fp64 (the float outputs are then identical except at rounding ties).
#include "frame_kernels.h"
#include "frame_device.h"

namespace {
using namespace svfd;

__global__ void __launch_bounds__(256) k_frame_observation(FrameObsProblem P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    ObsOut o;
    frame_obs_one(P.cam, P.kps[i], P.already_undistorted != 0, o);
    if (P.undist) P.undist[i] = o.undist;
    if (P.undist_xy) {
        P.undist_xy[2 * i] = o.ux;
        P.undist_xy[2 * i + 1] = o.uy;
    }
    if (P.bearings) {
        P.bearings[3 * i] = o.b0;
        P.bearings[3 * i + 1] = o.b1;
        P.bearings[3 * i + 2] = o.b2;
    }
}

__global__ void __launch_bounds__(256) k_can_observe(ReprojProblem P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    ReprojOut o;
    const bool offered = !(P.skip && P.skip[i]);
    // (fields a variant does not read may be absent: mean_normal with normal_mode 2, the distances with q_level)
    const double nx = P.mean_normal ? P.mean_normal[3 * i] : 0.0, ny = P.mean_normal ? P.mean_normal[3 * i + 1] : 0.0, nz = P.mean_normal ? P.mean_normal[3 * i + 2] : 0.0;
    reproject_one(P, P.rot_cw, P.trans_cw, P.trans_wc, offered, P.pos_w[3 * i], P.pos_w[3 * i + 1], P.pos_w[3 * i + 2], nx, ny, nz, P.min_valid_dist ? P.min_valid_dist[i] : 0.f,
                  P.max_valid_dist ? P.max_valid_dist[i] : 0.f, P.q_level ? P.q_level[i] : 0, P.q_level != nullptr, o);
    P.visible[i] = o.vis ? 1 : 0;
    P.reproj[2 * i] = o.vis ? o.rx : 0.0;
    P.reproj[2 * i + 1] = o.vis ? o.ry : 0.0;
    P.x_right[i] = o.vis ? o.xr : 0.f;
    P.pred_level[i] = o.vis ? o.level : -1;
    if (P.q_xy)  // match/projection.cc:30-37
        reproject_window(P, o, P.q_xy[2 * i], P.q_xy[2 * i + 1], P.q_margin[i], P.q_min_level[i], P.q_max_level[i]);
}

}  // namespace

void sv_launch_frame_observation(hipStream_t s, const FrameObsProblem& P) {
    if (P.n > 0) hipLaunchKernelGGL(k_frame_observation, dim3((P.n + 255) / 256), dim3(256), 0, s, P);
}
void sv_launch_reproject(hipStream_t s, const ReprojProblem& P) {
    if (P.n > 0) hipLaunchKernelGGL(k_can_observe, dim3((P.n + 255) / 256), dim3(256), 0, s, P);
}
