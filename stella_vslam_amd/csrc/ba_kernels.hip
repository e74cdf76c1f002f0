// HIP kernels of local bundle adjustment for gfx950 (fp64 throughout, as g2o).
//   k_ba_lin_lm / k_ba_lin_pose   linearizeOplus + constructQuadraticForm of every active edge
//                                 (optimize/internal/se3/perspective_reproj_edge.h:67-120,173-238; g2o base_binary_edge)
//   k_ba_dinv / k_ba_schur / k_ba_rhs / k_ba_chol / k_ba_update   BlockSolver_6_3::solve (Schur complement over the
//                                 marginalised landmarks) + LinearSolver (dense LL^T) + back-substitution + oplus
//   k_ba_chi2                     computeActiveErrors + activeRobustChi2
//   k_ba_gate                     chi2 / depth gate (optimize/local_bundle_adjuster_g2o.cc:323-344, 354-375)
// Every reduction runs in a fixed order (no floating-point atomics): results are run-to-run reproducible.
#include <cstdlib>
#include <cstring>

#include "svgpu_internal.h"
#include "ba_kernels.h"

namespace {

struct EdgeLin {
    double r[3];   // error (obs - projection); r[2] = 0 for monocular edges
    double A[9];   // d e / d landmark  (rows 0..D-1)
    double B[18];  // d e / d pose      (rows 0..D-1), update = [omega, upsilon] applied on the left
    double w;      // rho' * inv_sigma_sq
    double chi;    // e^T Omega e (not robustified)
    double z;
    int D;
};

__device__ __forceinline__ void cam_point(const double* __restrict__ T, const double* __restrict__ X, double* pc) {
    pc[0] = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3];
    pc[1] = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7];
    pc[2] = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
}

// Camera model of a pose.  The flat problem carries fx fy cx cy fxb per pose; fx == fy == 0 selects the EQUIRECTANGULAR model
// with cols = K[2], rows = K[3] (optimize/internal/se3/equirectangular_reproj_edge.h:64-134: always monocular, and
// depth_is_positive() is constant true for it, reproj_edge_wrapper.h:247-249).
__device__ __forceinline__ bool cam_is_equirect(const double* __restrict__ K) { return K[0] == 0.0 && K[1] == 0.0; }
__device__ __forceinline__ void equirect_project(const double* __restrict__ K, const double* pc, double* u, double* v) {  // cam_project, :128-132
    constexpr double PI = 3.14159265358979323846;
    const double theta = atan2(pc[0], pc[2]);
    const double phi = -asin(pc[1] / sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]));
    *u = K[2] * (0.5 + theta / (2 * PI));
    *v = K[3] * (0.5 - phi / PI);
}
// linearizeOplus (:70-126): d error / d [rx ry rz tx ty tz] (Bj: rows 0, 1 of a D x 6 block) and, when Rcw != nullptr,
// d error / d landmark (A: rows 0, 1 of a D x 3 block; Rcw = [R|t] rows, stride 4)
__device__ __forceinline__ void equirect_jacobians(const double* __restrict__ K, const double* pc, const double* __restrict__ Rcw, double* A,
                                                   double* Bj) {
    constexpr double PI = 3.14159265358979323846;
    const double x = pc[0], y = pc[1], z = pc[2], L = sqrt(x * x + y * y + z * z);
    const double dX[9] = {0, z, -y, 1, 0, 0, Rcw ? Rcw[0] : 0, Rcw ? Rcw[1] : 0, Rcw ? Rcw[2] : 0};
    const double dY[9] = {-z, 0, x, 0, 1, 0, Rcw ? Rcw[4] : 0, Rcw ? Rcw[5] : 0, Rcw ? Rcw[6] : 0};
    const double dZ[9] = {y, -x, 0, 0, 0, 1, Rcw ? Rcw[8] : 0, Rcw ? Rcw[9] : 0, Rcw ? Rcw[10] : 0};
    const double c0 = -(K[2] / (2 * PI)) * (1.0 / (x * x + z * z));
    const double c1 = -(K[3] / PI) * (1.0 / (L * sqrt(x * x + z * z)));
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double dL = (1.0 / L) * (x * dX[k] + y * dY[k] + z * dZ[k]);
        const double j0 = c0 * (z * dX[k] - x * dZ[k]);
        const double j1 = c1 * (L * dY[k] - y * dL);
        if (k < 6) {
            Bj[k] = j0;
            Bj[6 + k] = j1;
        }
        else if (A) {
            A[k - 6] = j0;
            A[3 + k - 6] = j1;
        }
    }
}

// error only (computeError); returns chi2 = e^T Omega e.  EQ as in edge_linearize_core below.
template <bool EQ = true>
__device__ __forceinline__ double edge_error(const double* __restrict__ T, const double* __restrict__ X, const double* __restrict__ K,
                                             const float* __restrict__ uvr, double w0, double* r, double* z_out) {
    double pc[3];
    cam_point(T, X, pc);
    const bool eq = EQ && cam_is_equirect(K);
    double u = K[0] * pc[0] / pc[2] + K[2];
    double v = K[1] * pc[1] / pc[2] + K[3];
    if (eq) equirect_project(K, pc, &u, &v);
    r[0] = (double)uvr[0] - u;
    r[1] = (double)uvr[1] - v;
    r[2] = (uvr[2] < 0.f || eq) ? 0.0 : (double)uvr[2] - (u - K[4] / pc[2]);
    if (z_out) *z_out = eq ? 1.0 : pc[2];
    return (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * w0;
}

__device__ __forceinline__ void huber(double e, double delta, double* rho0, double* rho1) {  // g2o RobustKernelHuber
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        *rho0 = e;
        *rho1 = 1.0;
    }
    else {
        const double sqrte = sqrt(e);
        *rho0 = 2 * sqrte * delta - dsqr;
        *rho1 = delta / sqrte;
    }
}

// the two state buffers: ctl->cur holds the estimate (linearisation point), the other one receives the trial state
__device__ __forceinline__ const double* st_pose(const BaDev& D, int trial) { return (D.ctl->cur ^ trial) ? D.pose_buf[1] : D.pose_buf[0]; }
__device__ __forceinline__ const double* st_pt(const BaDev& D, int trial) { return (D.ctl->cur ^ trial) ? D.pt_buf[1] : D.pt_buf[0]; }

// (p, l, uvr, w0, hub: the edge's pose, landmark, observation, information and Huber width -- read by the caller from the landmark-major
//  arrays by edge index, or from the pose-major copies by position in the pose's list; e: the edge index, for the robust flag)
//  EQ = false: the problem holds no equirectangular camera (BaDev::any_equirect, decided on the host from the intrinsics): the
//  atan2 / asin path and its registers are compiled out
template <bool EQ>
__device__ __forceinline__ void edge_linearize_core(const BaDev& D, const double* __restrict__ pose_cur, const double* __restrict__ pt_cur, int e, int p, int l,
                                                    const float* __restrict__ uvr, double w0, float hub, EdgeLin& o);
template <bool EQ>
__device__ __forceinline__ void edge_linearize(const BaDev& D, const double* __restrict__ pose_cur, const double* __restrict__ pt_cur, int e, EdgeLin& o) {
    edge_linearize_core<EQ>(D, pose_cur, pt_cur, e, D.e_pose[e], D.e_point[e], D.e_uvr + (size_t)e * 3, (double)D.e_w[e], D.e_huber[e], o);
}
template <bool EQ>
__device__ __forceinline__ void edge_linearize_core(const BaDev& D, const double* __restrict__ pose_cur, const double* __restrict__ pt_cur, int e, int p, int l,
                                                    const float* __restrict__ uvr, double w0, float hub, EdgeLin& o) {
    const double* T = pose_cur + (size_t)p * 12;
    const double* X = pt_cur + (size_t)l * 3;
    const double* K = D.intr + (size_t)p * 5;
    double pc[3];
    cam_point(T, X, pc);
    const double fx = K[0], fy = K[1], fxb = K[4];
    const double x = pc[0], y = pc[1], z = pc[2];
    // one reciprocal instead of ~25 fp64 divisions (each is a 20-instruction sequence)
    const double iz = 1.0 / z, iz2 = iz * iz, xz = x * iz, yz = y * iz;
    const bool eq = EQ && cam_is_equirect(K);
    double u = fx * xz + K[2], v = fy * yz + K[3];
    if (eq) equirect_project(K, pc, &u, &v);
    const bool stereo = !(uvr[2] < 0.f) && !eq;
    o.D = stereo ? 3 : 2;
    o.z = eq ? 1.0 : z;
    o.r[0] = (double)uvr[0] - u;
    o.r[1] = (double)uvr[1] - v;
    o.r[2] = stereo ? (double)uvr[2] - (u - fxb * iz) : 0.0;
    o.chi = (o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2]) * w0;
    const double fxz = fx * iz, fyz = fy * iz, fxxz2 = fx * xz * iz, fyyz2 = fy * yz * iz, fbz2 = fxb * iz2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.A[c] = fxxz2 * T[8 + c] - fxz * T[c];
        o.A[3 + c] = fyyz2 * T[8 + c] - fyz * T[4 + c];
        o.A[6 + c] = stereo ? o.A[c] - fbz2 * T[8 + c] : 0.0;
    }
    o.B[0] = xz * yz * fx;
    o.B[1] = -(1.0 + xz * xz) * fx;
    o.B[2] = yz * fx;
    o.B[3] = -fxz;
    o.B[4] = 0.0;
    o.B[5] = xz * fxz;
    o.B[6] = (1.0 + yz * yz) * fy;
    o.B[7] = -xz * yz * fy;
    o.B[8] = -xz * fy;
    o.B[9] = 0.0;
    o.B[10] = -fyz;
    o.B[11] = yz * fyz;
    if (stereo) {
        o.B[12] = o.B[0] - fbz2 * y;
        o.B[13] = o.B[1] + fbz2 * x;
        o.B[14] = o.B[2];
        o.B[15] = o.B[3];
        o.B[16] = 0.0;
        o.B[17] = o.B[5] - fbz2;
    }
    else {
#pragma unroll
        for (int k = 12; k < 18; ++k) o.B[k] = 0.0;
    }
    if (eq) equirect_jacobians(K, pc, T, o.A, o.B);  // rows 0, 1; row 2 is already zero (monocular)
    double rho1 = 1.0;
    if (D.e_robust[e]) {
        double rho0;
        huber(o.chi, (double)hub, &rho0, &rho1);
    }
    o.w = w0 * rho1;
}

__device__ __forceinline__ double wave_sum_dpp(double v);  // (below: DPP row scans, VALU only)
// 64-lane sum, the total in every lane, fixed order.  (Until round 3 a butterfly of __shfl_xor: two LDS-crossbar round trips per step on
// the critical path of every kernel that ends in a sum.)
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_dpp(v); }

// block-wide sum of one double per thread (blockDim.x <= 1024), fixed order; result valid in every thread
__device__ __forceinline__ double block_sum_d(double v, double* sw /* 16 doubles */) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 0; k < nw; ++k) r += sw[k];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------------------------------------ linearise
// thread per landmark: Hll, bl and the Hpl block W of every coupled edge
// 8 lanes per landmark: lane `sub` linearises the observations sub, sub + 8, ... of the landmark (a landmark has ~6), the
// partial Hll / bl are combined with a fixed xor-shuffle tree (deterministic), lane 0 of the group stores them.  With one
// thread per landmark the 10 k-landmark problem filled 40 workgroups; this fills 8x as many.
#define LM_LANES 8
__device__ __forceinline__ double group_sum8(double v) {
#pragma unroll
    for (int off = 1; off < LM_LANES; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// max |diagonal| over all active vertices (computeLambdaInit); non-negative doubles order like their bit patterns
__global__ __launch_bounds__(256) void k_ba_maxdiag(BaDev D) {
    if (D.ctl->phase != 0 || D.ctl->it != 0) return;  // computeLambdaInit: first iteration of an optimize() call only
    const int i = blockIdx.x * 256 + threadIdx.x;  // (the landmark part of the maximum is taken by k_ba_lin)
    double m = 0.0;
    if (i < D.nP)
        for (int j = 0; j < 6; ++j) m = fmax(m, fabs(D.Hpp_full[(size_t)i * 36 + 7 * j]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0)
        atomicMax(&D.ctl->max_diag_bits, (unsigned long long)__double_as_longlong(m));
}
// sharded solve: every rank publishes its maximum in its own slot; the slots are summed over the ranks (one-hot => the sum IS
// the slot vector) and k_ba_prepare takes the largest
__global__ void k_ba_maxslot(BaDev D) {
    const int r = threadIdx.x;
    if (r < D.world) D.maxslots[r] = (r == D.rank) ? __longlong_as_double((long long)D.ctl->max_diag_bits) : 0.0;
}

// Dense block LDL^T of the reduced system (3x3 pivots) with the right-hand side carried as row n, then the back substitution.
// One workgroup; everything here is a chain of short dependent steps, so the design minimises the LENGTH of that chain:
//  * the matrix lives in REGISTERS: it is cut into 3x3 tiles (n = 6 nP is a multiple of 3; tile row R-1 is the right-hand side)
//    and thread t owns tile t of the lower triangle for the whole factorisation -- the trailing update never reads or writes its
//    own operand in LDS; tiles are numbered column by column, so the waves retire one after the other;
//  * step j needs ONE barrier: the owner of the diagonal tile (j, j) inverts it (adjugate / determinant: one reciprocal, no square
//    root -- S = L' D L'^T with D_j = the pivot tiles, so W_j = D_j^-1 is all the update needs) and publishes W_j; after the barrier
//    every owner of a tile (I, K), K > j, subtracts A_Ij W_j A_Kj^T, where the UNSCALED panel tiles A_*j were published by their
//    owners at the end of the previous step (the tiles of column j+1 are final after update j);
//  * the owner of (j+1, j+1) goes from its update straight into the inversion (its wave runs at raised priority);
//  * the right-hand side row rides along (z = L'^-1 g), and x_I = W_I (z_I - sum_{M > I} A_MI^T x_M) is back-substituted by wave 0
//    from the published tiles, three rows per dependent step.
// A non-positive leading minor of a pivot tile (the system is not positive definite) fails the trial, as a failed LL^T would.
// 35 us at n = 96 (74 us for the 16-column-panel LL^T with its operand in LDS that this replaces): ~1500 cycles per step,
// the latencies behind it are measured by tools/ubench/latency.hip (profiles/r02_ubench_latency.json).
#define CHOL_MAX_THREADS 1024
__device__ __forceinline__ double rcp_nr(double x) {  // 1 / x: hardware seed, two Newton steps
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
// tiles of the lower triangle in COLUMN-major order (column K: I = K .. R-1): the tiles that are still being updated at step j
// (K > j) are a suffix of the tile list, so whole waves retire as the factorisation advances
__device__ __forceinline__ void tile_of(int t, int R, int& I, int& K) {
    const float b = 2.f * (float)R + 1.f;
    int k = (int)((b - sqrtf(fmaxf(b * b - 8.f * (float)t, 0.f))) * 0.5f);
    k = min(max(k, 0), R - 1);
    while (k > 0 && k * R - k * (k - 1) / 2 > t) --k;
    while (k < R - 1 && (k + 1) * R - (k + 1) * k / 2 <= t) ++k;
    K = k;
    I = k + t - (k * R - k * (k - 1) / 2);
}
__device__ __forceinline__ double readlane_d(double v, int src_lane) {  // src_lane wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

template <int TPT>  // tiles per thread
__global__ __launch_bounds__(CHOL_MAX_THREADS) void k_ba_chol_tile(BaDev D) {
    extern __shared__ double s_A[];  // (n + 3) rows x ld: the published (unscaled) tiles below the diagonal, row n = z (rows n+1, n+2: padding)
    __shared__ double s_W[64][6];    // inverse pivot tiles: w00 w10 w11 w20 w21 w22
    __shared__ int s_fail;
    if (D.ctl->phase != 1) return;
    const int n = D.n, tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
    const int ld = n | 1, R = n / 3 + 1, NT = R * (R + 1) / 2;
    double* A = s_A;
    // the lower triangle of the reduced system from its kept upper blocks: S[6a+i][6b+j] = blk[i][j] (a <= b) sits at row 6b+j,
    // column 6a+i; blocks that were not kept are zero; row n = the right-hand side
    for (int k = tid; k < (n + 3) * ld; k += nt) A[k] = 0.0;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int k = tid; k < D.NB * 36; k += nt) {
        const int blk = k / 36, t = k - 36 * blk, i = t / 6, j = t - 6 * i;
        const int2 ab = D.blk_ab[blk];
        A[(6 * ab.y + j) * ld + 6 * ab.x + i] = D.Sblk[k];
    }
    for (int c = tid; c < n; c += nt) A[n * ld + c] = D.g[c];
    __syncthreads();
    int tI[TPT], tK[TPT];
    double C[TPT][3][3];
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
        const int t = tid + q * nt;
        if (t < NT) tile_of(t, R, tI[q], tK[q]);
        else tI[q] = tK[q] = -1;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) C[q][r][c] = t < NT ? A[(3 * tI[q] + r) * ld + min(3 * tK[q] + c, n - 1)] : 0.0;
    }
    for (int j = 0; j < R - 1; ++j) {
        // ---- the pivot tile's inverse (one thread)
#pragma unroll
        for (int q = 0; q < TPT; ++q)
            if (tI[q] == j && tK[q] == j) {
                const double c00 = C[q][0][0], c10 = C[q][1][0], c11 = C[q][1][1], c20 = C[q][2][0], c21 = C[q][2][1], c22 = C[q][2][2];
                const double m00 = fma(c11, c22, -(c21 * c21)), m10 = fma(c21, c20, -(c10 * c22)), m20 = fma(c10, c21, -(c11 * c20));
                const double m11 = fma(c00, c22, -(c20 * c20)), m21 = fma(c10, c20, -(c00 * c21)), m22 = fma(c00, c11, -(c10 * c10));
                const double det = fma(c20, m20, fma(c10, m10, c00 * m00));
                if (!(c00 > 0.0) || !(m22 > 0.0) || !(det > 0.0)) s_fail = 1;  // leading minors
                const double rd = rcp_nr(det);
                double* w = s_W[j];
                w[0] = m00 * rd;
                w[1] = m10 * rd;
                w[2] = m11 * rd;
                w[3] = m20 * rd;
                w[4] = m21 * rd;
                w[5] = m22 * rd;
            }
        __syncthreads();
        // ---- trailing update; the tiles of column j + 1 are final afterwards and are published.  All LDS reads of the step are
        // issued back to back and awaited once (left alone the scheduler strings them out between the multiply-adds: five
        // round trips instead of one on the critical path); the failure flag is read with them and tested after the update.
        bool next_pivot = false;
#pragma unroll
        for (int q = 0; q < TPT; ++q) next_pivot = next_pivot || (tI[q] == j + 1 && tK[q] == j + 1);
        if (__any(next_pivot)) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        const int failed = __hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool act[TPT], any_act = false;
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            act[q] = tK[q] > j;
            any_act = any_act || act[q];
        }
        if (any_act) {
            double w[6], a[TPT][3][3], b[TPT][3][3];
#pragma unroll
            for (int e = 0; e < 6; ++e) w[e] = s_W[j][e];
#pragma unroll
            for (int q = 0; q < TPT; ++q) {
                const double* la = A + (3 * max(tI[q], 0)) * ld + 3 * j;
                const double* lb = A + (3 * max(tK[q], 0)) * ld + 3 * j;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        a[q][r][k] = la[r * ld + k];
                        b[q][r][k] = lb[r * ld + k];
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
                const double w00 = w[0], w10 = w[1], w11 = w[2], w20 = w[3], w21 = w[4], w22 = w[5];
#pragma unroll
            for (int q = 0; q < TPT; ++q)
                if (act[q]) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double t0 = fma(a[q][r][2], w20, fma(a[q][r][1], w10, a[q][r][0] * w00));
                        const double t1 = fma(a[q][r][2], w21, fma(a[q][r][1], w11, a[q][r][0] * w10));
                        const double t2 = fma(a[q][r][2], w22, fma(a[q][r][1], w21, a[q][r][0] * w20));
#pragma unroll
                        for (int c = 0; c < 3; ++c) C[q][r][c] = fma(-t2, b[q][c][2], fma(-t1, b[q][c][1], fma(-t0, b[q][c][0], C[q][r][c])));
                    }
                    if (tK[q] == j + 1 && tI[q] > j + 1) {
                        double* o = A + (3 * tI[q]) * ld + 3 * (j + 1);
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c) o[r * ld + c] = C[q][r][c];
                    }
                }
        }
        if (failed) break;
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (s_fail) {
        if (tid == 0) D.ctl->solve_failed = 1;
        for (int i = tid; i < n; i += nt) D.dp[i] = 0.0;
        return;
    }
    if (tid < 64) {
        // x_I = W_I (z_I - sum_{M > I} A_MI^T x_M), I = R-2 .. 0.  Lane k < 63 holds entries k, k + 63, k + 126 of z (63 = 21 blocks:
        // a block never straddles two registers, so every register index below is a compile-time constant); the three finished
        // entries of block I are broadcast with v_readlane, x_I is computed by every lane, and the lanes holding earlier entries
        // subtract their share sum_r A[3 I + r][k] x_I[r].  The rows and W of block I - 1 are loaded while block I is applied.
        double z0, z1, z2;
        {
            const int k = min(lane, 62);
            z0 = k < n ? A[n * ld + k] : 0.0;
            z1 = k + 63 < n ? A[n * ld + k + 63] : 0.0;
            z2 = k + 126 < n ? A[n * ld + k + 126] : 0.0;
        }
        struct Blk {
            double r0[3], r1[3], r2[3], w[6];  // rows 3 I + {0, 1, 2} at this lane's entries (register q), the inverse pivot tile
        };
        auto fetch = [&](int I, int Q, Blk& b) {  // Q = I / 21: only registers q <= Q are still open
            const int Ic = max(I, 0), k = min(lane, 62);
            const double* row = A + (3 * Ic) * ld;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q <= Q) {
                    const int e = min(k + 63 * q, n - 1);
                    b.r0[q] = row[e];
                    b.r1[q] = row[ld + e];
                    b.r2[q] = row[2 * ld + e];
                }
#pragma unroll
            for (int e = 0; e < 6; ++e) b.w[e] = s_W[Ic][e];
        };
        auto apply = [&](int I, auto QC, const Blk& b) {
            constexpr int Q = decltype(QC)::value;
            const int base = 3 * (I - 21 * Q);  // lane of the block's first entry in register Q
            double& zq = Q == 0 ? z0 : (Q == 1 ? z1 : z2);
            const double s0 = readlane_d(zq, base), s1 = readlane_d(zq, base + 1), s2 = readlane_d(zq, base + 2);
            const double x0 = fma(b.w[3], s2, fma(b.w[1], s1, b.w[0] * s0));
            const double x1 = fma(b.w[4], s2, fma(b.w[2], s1, b.w[1] * s0));
            const double x2 = fma(b.w[5], s2, fma(b.w[4], s1, b.w[3] * s0));
            const int d = lane - base;
            if (d < 0) zq = fma(-b.r2[Q], x2, fma(-b.r1[Q], x1, fma(-b.r0[Q], x0, zq)));
            else if (d < 3) zq = d == 0 ? x0 : (d == 1 ? x1 : x2);
            if constexpr (Q >= 1) z0 = fma(-b.r2[0], x2, fma(-b.r1[0], x1, fma(-b.r0[0], x0, z0)));
            if constexpr (Q >= 2) z1 = fma(-b.r2[1], x2, fma(-b.r1[1], x1, fma(-b.r0[1], x0, z1)));
        };
        auto phase = [&](auto QC, int I_hi, int I_lo) {  // blocks I_hi .. I_lo (descending), all in register Q
            constexpr int Q = decltype(QC)::value;
            if (I_hi < I_lo) return;
            Blk ba, bb;
            fetch(I_hi, Q, ba);
            for (int I = I_hi; I >= I_lo; I -= 2) {
                fetch(max(I - 1, I_lo), Q, bb);
                apply(I, QC, ba);
                if (I - 1 >= I_lo) {
                    fetch(max(I - 2, I_lo), Q, ba);
                    apply(I - 1, QC, bb);
                }
            }
        };
        const int last = R - 2;
        phase(std::integral_constant<int, 2>{}, last, 42);
        phase(std::integral_constant<int, 1>{}, min(last, 41), 21);
        phase(std::integral_constant<int, 0>{}, min(last, 20), 0);
        if (lane < 63) {
            if (lane < n) D.dp[lane] = z0;
            if (lane + 63 < n) D.dp[lane + 63] = z1;
            if (lane + 126 < n) D.dp[lane + 126] = z2;
        }
    }
}

// ---- Blocked right-looking LL^T of the reduced camera system on the MATRIX CORES (north_star: "MFMA only for the dense reduced-camera-block
// solve"; local_bundle_adjuster_g2o.cc:151-164 hands this system to Eigen's dense LL^T).  Panels of 16 columns; per panel
//   1. ONE wave factors the panel with a ROW PER LANE (two rows per lane beyond 64 remaining rows, the right-hand side riding along as row
//      n): the 16 entries of a row sit in registers, column j's pivot and the multipliers l_cj of the pivot tile travel by v_readlane, every
//      lane scales and updates its own row.  No barrier, no LDS and no triangular solve inside the panel: the rows below the pivot tile are
//      finished by the same instructions that factor it;
//   2. the trailing update A_IJ -= L_Ik L_Jk^T is v_mfma_f64_16x16x4_f64 on 16 x 16 tiles (four k-steps per tile), the tiles of the lower
//      triangle dealt round-robin to the eight waves, operands read from LDS in the instruction's own layouts (A[l & 15][l >> 4],
//      B[l >> 4][l & 15], C row = (l >> 4) + 4 reg, col = l & 15).
// Two barriers per PANEL (6 panels at n = 96) where the register-tile LDL^T below needs one per 3 columns (32), and the update runs at the
// matrix pipe's rate.  Fixed order everywhere: run-to-run bit-identical.  z = L^-1 g arrives with the factorisation; L^T x = z is one wave,
// a column per step, the rows prefetched.  A non-positive pivot fails the trial (as a failed LL^T does).
#define CM_THREADS 512
#define CM_MAX_N 126  // n + 1 rows fit two rows per lane
typedef double cm_v4f64 __attribute__((ext_vector_type(4)));
__host__ __device__ inline int cm_ld(int n) { return 16 * ((n + 15) / 16) + 1; }                                  // odd pitch (doubles)
__host__ __device__ inline size_t cm_lds_bytes(int n) { return sizeof(double) * (size_t)(16 * ((n + 16) / 16)) * cm_ld(n); }  // rows 0 .. n, whole tiles
__device__ __forceinline__ double cm_rsqrt(double p) {  // 1 / sqrt(p): hardware seed, two Newton steps
    double y = __builtin_amdgcn_rsq(p);
    y = y * fma(-0.5 * p * y, y, 1.5);
    y = y * fma(-0.5 * p * y, y, 1.5);
    return y;
}
__global__ __launch_bounds__(CM_THREADS) void k_ba_chol_mfma(BaDev D) {
    extern __shared__ double s_M[];  // rows 0 .. n-1 the matrix (lower triangle), row n the right-hand side, then zero rows up to a whole tile
    __shared__ double s_dinv[128];   // 1 / L_jj
    __shared__ int s_fail;
    if (D.ctl->phase != 1) return;
#define CM_T(i) do { if (D.dbg && D.dbg_schur_on == 2 && threadIdx.x == 0) D.dbg[(i)] = wall_clock64(); } while (0)
    CM_T(0);
    const int n = D.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTc = (n + 15) / 16, NTr = (n + 16) / 16, ld = cm_ld(n), rows = 16 * NTr;
    double* const M = s_M;
    for (int k = tid; k < rows * ld; k += CM_THREADS) M[k] = 0.0;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    // the lower triangle from the kept upper blocks: S[6a+i][6b+j] = blk[i][j] (a <= b) sits at row 6b+j, column 6a+i
    for (int k = tid; k < D.NB * 36; k += CM_THREADS) {
        const int blk = k / 36, t = k - 36 * blk, i = t / 6, j = t - 6 * i;
        const int2 ab = D.blk_ab[blk];
        M[(6 * ab.y + j) * ld + 6 * ab.x + i] = D.Sblk[k];
    }
    for (int c = tid; c < n; c += CM_THREADS) M[n * ld + c] = D.g[c];
    __syncthreads();
    CM_T(1);
    for (int k = 0; k < NTc; ++k) {
        const int c0 = 16 * k;
        if (wave == 0) {
            // ---- panel factorisation: lane l owns rows c0 + l and c0 + 64 + l (rows beyond n do not exist: zeros, harmless)
            const int r0 = c0 + lane, r1 = c0 + 64 + lane;
            const bool has1 = c0 + 64 <= n;  // (wave-uniform: any second row at all)
            double a0[16], a1[16];
            {
                const double* p0 = M + (size_t)min(r0, rows - 1) * ld + c0;
                const double* p1 = M + (size_t)min(r1, rows - 1) * ld + c0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    a0[j] = p0[j];
                    a1[j] = p1[j];  // (no second row: a clamped address, a value nobody stores)
                }
            }
            bool bad = false;
            double dinv[16];
            // LEFT-looking inside the panel: column j is brought up to date with the finished columns m < j (multiplier l_jm = row j's entry in
            // column m, by v_readlane), then its pivot is taken and the column scaled.  Every broadcast value is consumed at once -- written
            // right-looking, the compiler deferred the updates of the later columns itself and kept all 120 multipliers alive in scalar
            // registers, spilling them through v_writelane: 15 cycles per instruction.  Two accumulators per row halve the dependent chain.
            // The second set of rows (more than 64 rows left) follows in a pass of its own with the finished multipliers.
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // the j multipliers first (independent v_readlane pairs, issued back to back), then the multiply-adds: a multiply-add behind
                // every pair waits out the scalar write each time
                double l[16];
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    if (m < j) l[m] = readlane_d(a0[m], j);
                __builtin_amdgcn_sched_barrier(0);
                double e0 = a0[j], o0 = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    if (m < j) {
                        if (m & 1) o0 = fma(-a0[m], l[m], o0);
                        else e0 = fma(-a0[m], l[m], e0);
                    }
                e0 += o0;
                const double piv = readlane_d(e0, j);
                const bool live = c0 + j < n;  // (columns beyond n in the last panel: nothing to eliminate)
                bad = bad || (live && !(piv > 0.0));
                const double r0 = cm_rsqrt(live ? piv : 1.0);  // (selects, no branch: the column loop stays one basic block)
                const double r = live ? r0 : 0.0;
                a0[j] = e0 * r;  // lane j: sqrt(piv) = L_jj; lanes below: l_rj
                dinv[j] = r;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has1) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double l[16];
#pragma unroll
                    for (int m = 0; m < 16; ++m)
                        if (m < j) l[m] = readlane_d(a0[m], j);
                    __builtin_amdgcn_sched_barrier(0);
                    double e1 = a1[j], o1 = 0.0;
#pragma unroll
                    for (int m = 0; m < 16; ++m)
                        if (m < j) {
                            if (m & 1) o1 = fma(-a1[m], l[m], o1);
                            else e1 = fma(-a1[m], l[m], e1);
                        }
                    a1[j] = (e1 + o1) * dinv[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (lane < 16) {  // 1 / L_jj for the back substitution: lane j stores the j-th (uniform) value
                double d = dinv[0];
#pragma unroll
                for (int j = 1; j < 16; ++j) d = lane == j ? dinv[j] : d;
                s_dinv[c0 + lane] = d;
            }
            if (bad && lane == 0) s_fail = 1;
            if (r0 < rows) {
                double* p0 = M + (size_t)r0 * ld + c0;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j <= lane) p0[j] = a0[j];  // the lower triangle of the pivot tile, whole rows below it
            }
            if (has1 && r1 < rows) {
                double* p1 = M + (size_t)r1 * ld + c0;
#pragma unroll
                for (int j = 0; j < 16; ++j) p1[j] = a1[j];
            }
            CM_T(2 + 3 * k);
        }
        __syncthreads();
        CM_T(3 + 3 * k);
        if (s_fail) break;
        // ---- trailing update on the matrix cores: tile (I, J), k < J <= I, J a column tile, I up to the right-hand side's tile
        {
            const int nJ = NTc - 1 - k;  // column tiles behind the panel
            int t = wave;
            for (int J = k + 1; J < NTc; ++J)
                for (int I = J; I < NTr; ++I, --t)
                    if (t == 0) {
                        t = CM_THREADS / 64;
                        const double* pa = M + (size_t)(16 * I + (lane & 15)) * ld + c0 + (lane >> 4);
                        const double* pb = M + (size_t)(16 * J + (lane & 15)) * ld + c0 + (lane >> 4);
                        double* pc = M + (size_t)(16 * I + (lane >> 4)) * ld + 16 * J + (lane & 15);
                        const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];
                        const double b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];
                        cm_v4f64 c;
                        c[0] = pc[0], c[1] = pc[(size_t)4 * ld], c[2] = pc[(size_t)8 * ld], c[3] = pc[(size_t)12 * ld];
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1, b1, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a2, b2, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a3, b3, c, 0, 0, 0);
                        pc[0] = c[0], pc[(size_t)4 * ld] = c[1], pc[(size_t)8 * ld] = c[2], pc[(size_t)12 * ld] = c[3];
                    }
            (void)nJ;
        }
        __syncthreads();
        CM_T(4 + 3 * k);
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) D.ctl->solve_failed = 1;
        for (int i = tid; i < n; i += CM_THREADS) D.dp[i] = 0.0;
        return;
    }
    if (wave == 0) {
        // ---- L^T x = z, a column per step from the last: lane c holds z_c (register A) and z_{64 + c} (register B); x_I = z_I / L_II is
        //      broadcast, every lane c < I subtracts L[I][c] x_I (row I of L: consecutive addresses over the lanes)
        const double* zr = M + (size_t)n * ld;
        double zA = lane < n ? zr[lane] : 0.0, zB = 64 + lane < n ? zr[64 + lane] : 0.0;
        int I = n - 1;
        double rowA = M[(size_t)I * ld + min(lane, I)], rowB = I >= 64 ? M[(size_t)I * ld + min(64 + lane, I)] : 0.0, di = s_dinv[I];
        for (; I >= 64; --I) {
            const int In = max(I - 1, 0);
            const double nA = M[(size_t)In * ld + min(lane, In)], nB = In >= 64 ? M[(size_t)In * ld + min(64 + lane, In)] : 0.0, nd = s_dinv[In];
            const double x = readlane_d(zB, I - 64) * di;
            if (64 + lane < I) zB = fma(-rowB, x, zB);
            else if (64 + lane == I) zB = x;
            zA = fma(-rowA, x, zA);
            rowA = nA, rowB = nB, di = nd;
        }
        for (; I >= 0; --I) {
            const int In = max(I - 1, 0);
            const double nA = M[(size_t)In * ld + min(lane, In)], nd = s_dinv[In];
            const double x = readlane_d(zA, I) * di;
            if (lane < I) zA = fma(-rowA, x, zA);
            else if (lane == I) zA = x;
            rowA = nA, di = nd;
        }
        if (lane < n) D.dp[lane] = zA;
        if (64 + lane < n) D.dp[64 + lane] = zB;
        CM_T(30);
    }
#undef CM_T
}

// Fallback for systems that do not fit LDS (n > ~140): same arithmetic on the global copy, one column per step.
__global__ __launch_bounds__(1024) void k_ba_chol_global(BaDev D) {
    if (D.ctl->phase != 1) return;
    __shared__ int s_fail;
    const int n = D.n, tid = threadIdx.x, nt = blockDim.x, ld = n;
    double* A = D.S;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const int t16i = tid >> 4, t16j = tid & 15, tstep = nt >> 4;
    for (int j = 0; j < n; ++j) {
        const double djj = A[(size_t)j * ld + j];
        if (!(djj > 0.0)) {
            if (tid == 0) s_fail = 1;
            break;
        }
        const double d = sqrt(djj);
        __syncthreads();
        for (int i = j + tid; i <= n; i += nt) A[(size_t)i * ld + j] = (i == j) ? d : A[(size_t)i * ld + j] / d;
        __syncthreads();
        for (int i = j + 1 + t16i; i <= n; i += tstep) {
            const double lij = A[(size_t)i * ld + j];
            const int kmax = i < n ? i : n - 1;
            for (int k = j + 1 + t16j; k <= kmax; k += 16) A[(size_t)i * ld + k] -= lij * A[(size_t)k * ld + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) D.ctl->solve_failed = 1;
        for (int i = tid; i < n; i += nt) D.dp[i] = 0.0;
        return;
    }
    for (int i = n - 1; i >= 0; --i) {
        const double xi = A[(size_t)n * ld + i] / A[(size_t)i * ld + i];
        __syncthreads();
        if (tid == 0) A[(size_t)n * ld + i] = xi;
        for (int k = tid; k < i; k += nt) A[(size_t)n * ld + k] -= A[(size_t)i * ld + k] * xi;
        __syncthreads();
    }
    for (int i = tid; i < n; i += nt) D.dp[i] = A[(size_t)n * ld + i];
}

// ------------------------------------------------------------------------------------------------ errors
// robust chi2 of the observations sub, sub + 8, ... of landmark l (lane `sub` of the landmark's group of 8), poses from `poses` (12 doubles
// each, global or LDS), the landmark at X.  Shared by k_ba_chi2 and the fused trial tail so that both sum in the same order.
template <bool EQ>
__device__ __forceinline__ double lm_chi2_lane(const BaDev& D, int l, int sub, const double* poses, const double* X, int store_cache) {
    double v = 0.0;
    for (int e = D.lm_off[l] + sub; e < D.lm_off[l + 1]; e += 8) {
        if (D.e_level[e]) continue;
        const int p = D.e_pose[e];
        double r[3];
        const double chi = edge_error<EQ>(poses + (size_t)p * 12, X, D.intr + (size_t)p * 5, D.e_uvr + (size_t)e * 3, (double)D.e_w[e], r, nullptr);
        if (store_cache) D.e_chi[e] = chi;
        if (D.e_robust[e]) {
            double rho0, rho1;
            huber(chi, (double)D.e_huber[e], &rho0, &rho1);
            v += rho0;
        }
        else v += chi;
    }
    return v;
}

// 8 lanes per landmark (the observations are sorted by landmark), one partial sum per workgroup of 32 landmarks
template <bool EQ>
__global__ __launch_bounds__(256) void k_ba_chi2(BaDev D, int use_trial, int store_cache, int guarded) {
    if (guarded && D.ctl->phase != 1) return;
    __shared__ double s4[16];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int l = min(t / 8, D.L - 1), sub = t % 8;
    double v = 0.0;
    if (t / 8 < D.L) {
        const double* Xg = st_pt(D, use_trial) + (size_t)l * 3;
        const double X[3] = {Xg[0], Xg[1], Xg[2]};
        v = lm_chi2_lane<EQ>(D, l, sub, st_pose(D, use_trial), X, store_cache);
    }
    const double tsum = block_sum_d(v, s4);
    if (threadIdx.x == 0) D.red[D.red_chi_off + blockIdx.x] = tsum;
}

// chi2 / depth gate on the cached chi2 (stale for excluded edges, exactly as g2o's cached _error)
__global__ __launch_bounds__(256) void k_ba_gate(BaDev D, int set_levels, uint8_t* __restrict__ outlier_out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= D.E) return;
    const int p = D.e_pose[e], l = D.e_point[e];
    double pc[3];
    cam_point(st_pose(D, 0) + (size_t)p * 12, st_pt(D, 0) + (size_t)l * 3, pc);
    const bool mono = D.e_uvr[(size_t)e * 3 + 2] < 0.f;
    const double thr = mono ? (double)5.99146f : (double)7.81473f;
    // a negative Huber width marks a marker-corner edge: those live in a container of their own that neither the gate loop nor the
    // outlier loop visits (local_bundle_adjuster_g2o.cc:246-304 against :324-343, :354-375)
    const bool out = !(D.e_huber[e] < 0.f) && (thr < D.e_chi[e] || !(cam_is_equirect(D.intr + (size_t)p * 5) || 0.0 < pc[2]));
    if (set_levels) {
        if (out) D.e_level[e] = 1;
        D.e_robust[e] = 0;
    }
    if (outlier_out) outlier_out[e] = out ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ pose optimizer
// pose_optimizer_g2o::optimize (optimize/pose_optimizer_g2o.cc:38-175) as ONE persistent workgroup: the problem is a single
// 6-dof vertex with <= a few thousand unary edges, i.e. launch-latency-bound if it were split into kernels.  All LM
// rounds, the 6x6 solves (thread 0) and the chi-square re-classification run on the device; the reductions are
// fixed-order (shuffles + wave partials in LDS).
#define PO_THREADS 512
#define WRED_PITCH 66
#define WRED_DOUBLES (18 * WRED_PITCH + 18 * 4)
template <int NV, class Store>
__device__ __forceinline__ void wave_reduce_lds(const double (&acc)[NV], double* __restrict__ sw, int lane, Store store);  // (below)
// Workgroup sums of NV per-thread values, fixed order: per wave through a transpose in wave-private LDS (wave_reduce_lds: ~60 instructions
// per 18 values), then the eight wave partials.  (A butterfly of __shfl_xor per value -- two LDS-crossbar round trips per step, 348 of
// them per wave for the 29 sums of a linearisation -- was most of an LM iteration of this kernel.)
template <int NV>
__device__ __forceinline__ void po_reduce(const double (&vals)[NV], double* __restrict__ s_wred, double (*s_part)[32], double* s_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (NV == 1) {
        const double t = wave_sum_dpp(vals[0]);
        if (lane == 0) s_part[wave][0] = t;
    }
    else wave_reduce_lds<NV>(vals, s_wred + (size_t)wave * WRED_DOUBLES, lane, [&](int k, double t) { s_part[wave][k] = t; });
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        double t = 0.0;
        for (int w = 0; w < PO_THREADS / 64; ++w) t += s_part[w][threadIdx.x];
        s_out[threadIdx.x] = t;
    }
    __syncthreads();
}
__device__ __forceinline__ void po_exp_mul(const double* u, const double* T, double* O) {  // O = exp(u) * T, as k_ba_update_pose
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double Om[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    double a, b, c, d;
    if (theta < 0.00001) {
        a = 1.0;
        b = 0.5;
        c = 0.5;
        d = 1.0 / 6.0;
    }
    else {
        const double st = sin(theta), ct = cos(theta);
        a = st / theta;
        b = (1 - ct) / (theta * theta);
        c = b;
        d = (theta - st) / (theta * theta * theta);
    }
    double R[9], V[9];
    for (int k = 0; k < 9; ++k) {
        const double I = (k % 4 == 0) ? 1.0 : 0.0;
        R[k] = I + a * Om[k] + b * O2[k];
        V[k] = I + c * Om[k] + d * O2[k];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) O[4 * i + j] = R[3 * i] * T[j] + R[3 * i + 1] * T[4 + j] + R[3 * i + 2] * T[8 + j];
        O[4 * i + 3] = R[3 * i] * T[3] + R[3 * i + 1] * T[7] + R[3 * i + 2] * T[11] + V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    }
}
// (fully unrolled: every index is a compile-time constant, the factor lives in registers -- with run-time loop bounds the 36-entry array sat
//  in scratch memory.  The divisions stay divisions: with one reciprocal per column the iterates moved by an ulp, and at a converged pose
//  the gain test of a further LM iteration is decided by exactly that -- the iteration counts left the reference's by two.)
template <class HP>
__device__ __forceinline__ bool po_chol6(HP H, double lambda, HP b, double* x) {  // (H + lambda I) x = b
    double A[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[7 * i] += lambda;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[7 * j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) d -= A[6 * j + k] * A[6 * j + k];
        ok = ok && d > 0.0;
        d = sqrt(d);
        A[7 * j] = d;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i > j) {
                double sum = A[6 * i + j];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) sum -= A[6 * i + k] * A[6 * j + k];
                A[6 * i + j] = sum / d;
            }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sum = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) sum -= A[6 * i + k] * x[k];
        x[i] = sum / A[7 * i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sum = x[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) sum -= A[6 * k + i] * x[k];
        x[i] = sum / A[7 * i];
    }
    return true;
}

// TRK: the tracked-frame chain's form (PoseOptDev::trk_*): the workgroup first applies the matcher's result to the frame's landmark ids and
// compacts the observations (keypoint order = the order pose_optimizer_g2o.cc:88-107 adds its edges in) out of the resident landmark table.
template <bool EQ, bool TRK = false>  // EQ false: the camera is not equirectangular (decided on the host from the intrinsics): no atan2 / asin path, fewer registers
__global__ __launch_bounds__(PO_THREADS) void k_pose_opt(PoseOptDev P) {
    extern __shared__ __attribute__((aligned(16))) double s_wred[];  // PO_THREADS / 64 transposition buffers of WRED_DOUBLES (dynamic: 81 KB)
    __shared__ double s_part[PO_THREADS / 64][32];
    __shared__ double s_red[32];
    __shared__ double s_T[12], s_Tt[12], s_x[6], s_H[36], s_b[6];
    __shared__ int s_ctl[4];  // [0] accept, [1] continue trials, [2] ok2
    const int tid = threadIdx.x;
    int n = P.n;
    int trk_nt = 0;
    int n_stamps = 0;
    auto stamp = [&]() {  // (debug: 100 MHz wall clock at a phase boundary)
        if (P.stamps && tid == 0 && n_stamps < 62) P.stamps[++n_stamps] = wall_clock64(), P.stamps[0] = n_stamps;
    };
    stamp();
    if constexpr (TRK) {
        __shared__ int s_nobs;
        if (*P.trk_overflow > P.trk_overflow_cap) return;  // the lists did not fit: the host re-runs the chain with a larger capacity
        const int nt = P.trk_nt_dev ? min(*P.trk_nt_dev, P.trk_nt) : P.trk_nt;
        trk_nt = nt;
        const svgpu_landmark_record* map = (const svgpu_landmark_record*)P.trk_map;
        // 1. matches onto the frame: add_landmark in increasing query order, a later query overwrites (projection.cc:88, :202).  The table of
        //    "last query that took keypoint k" lives in LDS (the reduction buffers are idle until the first linearisation).
        constexpr int TRK_ROUNDS = (8192 + PO_THREADS - 1) / PO_THREADS;  // <= 8 192 keypoints per frame (checked by the host)
        int* const s_who = reinterpret_cast<int*>(s_wred);
        int lm[TRK_ROUNDS];
#pragma unroll
        for (int r = 0; r < TRK_ROUNDS; ++r) {
            const int k = tid + r * PO_THREADS;
            if (k < nt) s_who[k] = -1;
            lm[r] = (k < nt && !P.trk_reset_cur) ? P.trk_cur_lm[k] : -1;  // (in flight while the matches are applied)
        }
        __syncthreads();
        for (int q = tid; q < P.trk_nq; q += PO_THREADS) {
            const int m = P.trk_match_q[q];
            if (m >= 0 && m < nt) atomicMax(&s_who[m], q);
        }
        __syncthreads();
        // 2. the landmark every keypoint holds now, and whether it yields an edge.  Keypoints are taken STRIDED (coalesced) and every level
        //    of the dependent chain (query id -> table record) is loaded for ALL of a thread's keypoints before the next level is touched:
        //    the round trips overlap instead of queueing (a thread walking its own contiguous run paid three per keypoint).
        unsigned okmask = 0;
#pragma unroll
        for (int r = 0; r < TRK_ROUNDS; ++r) {
            const int k = tid + r * PO_THREADS;
            const int w = k < nt ? s_who[k] : -1;
            if (w >= 0) lm[r] = P.trk_qid[w];
        }
        __syncthreads();  // (s_who is dead from here on: the buffer goes back to the reductions)
#pragma unroll
        for (int r = 0; r < TRK_ROUNDS; ++r) {
            const int k = tid + r * PO_THREADS;
            const bool ok = k < nt && lm[r] >= 0 && lm[r] < P.trk_map_cap && (map[lm[r]].flags & SVGPU_LM_PRESENT);  // !lm || lm->will_be_erased(): no edge (:81-87)
            okmask |= ok ? 1u << r : 0u;
            if (k < nt) P.trk_cur_lm[k] = lm[r];
        }
        // 3. keypoint order = edge order: exclusive scan of the edge flags over k = tid + r * PO_THREADS, i.e. round-major -- round r's
        //    PO_THREADS keypoints precede round r + 1's.  Per round a wave ballot + popcount, the waves' counts through LDS.
        __shared__ int s_wround[TRK_ROUNDS][PO_THREADS / 64];
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int r = 0; r < TRK_ROUNDS; ++r) {
            const unsigned long long b = __ballot((okmask >> r) & 1u);
            if (lane == 0) s_wround[r][wave] = __popcll(b);
        }
        __syncthreads();
        int nobs = 0;
#pragma unroll
        for (int r0 = 0; r0 < TRK_ROUNDS; r0 += 4) {  // four rounds at a time: their table reads are issued together, then the stores
            if (r0 * PO_THREADS >= nt) break;
            int j4[4];
            double px[4], py[4], pz[4];
            float ux[4], uy[4], ur[4], ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u;
                const unsigned long long b = __ballot((okmask >> r) & 1u);
                int before = 0, round_total = 0;
#pragma unroll
                for (int w = 0; w < PO_THREADS / 64; ++w) {
                    const int c = s_wround[r][w];
                    before += w < wave ? c : 0;
                    round_total += c;
                }
                j4[u] = -1;
                if ((okmask >> r) & 1u) {
                    j4[u] = nobs + before + __popcll(b & ((1ull << lane) - 1ull));
                    const int k = tid + r * PO_THREADS, id = lm[r];
                    px[u] = map[id].pos_w[0], py[u] = map[id].pos_w[1], pz[u] = map[id].pos_w[2];
                    ux[u] = P.trk_xy[2 * k], uy[u] = P.trk_xy[2 * k + 1];
                    ur[u] = P.trk_xright ? P.trk_xright[k] : -1.0f;
                    ww[u] = P.trk_inv_sigma_sq[P.trk_octave[k] & 15];
                }
                nobs += round_total;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j4[u];
                if (j < 0) continue;
                P.trk_pos[3 * (size_t)j] = px[u];
                P.trk_pos[3 * (size_t)j + 1] = py[u];
                P.trk_pos[3 * (size_t)j + 2] = pz[u];
                P.trk_uvr[3 * (size_t)j] = ux[u];
                P.trk_uvr[3 * (size_t)j + 1] = uy[u];
                P.trk_uvr[3 * (size_t)j + 2] = ur[u];
                P.trk_w[j] = ww[u];
                P.trk_h[j] = P.trk_huber;
                P.trk_kp_of[j] = tid + (r0 + u) * PO_THREADS;
            }
        }
        if (tid == 0) s_nobs = nobs;
        __syncthreads();
        stamp();
        n = s_nobs;
        for (int k = tid; k < nt; k += PO_THREADS) {
            P.trk_outlier_kp[k] = 0;
            P.host_outlier_kp[k] = 0;
        }
        if (n < 5) {  // pose_optimizer_g2o.cc:109-111: nothing to optimise, the pose is returned as it came
            if (tid < 12) {
                const double v = P.pose_in_dev ? P.pose_in_dev[tid] : P.pose_in[tid];
                P.pose_out[tid] = v;
                P.host_pose[tid] = v;
            }
            if (tid == 0) {
                P.host_result[0] = 0, P.host_result[1] = 0, P.host_result[2] = 0, P.host_result[3] = n;
                P.result[0] = 0, P.result[1] = 0, P.result[2] = 0;
                if (P.trk_counter_reset) *P.trk_counter_reset = 0;
            }
            return;
        }
    }
    if (tid < 12) s_T[tid] = (TRK && P.pose_in_dev) ? P.pose_in_dev[tid] : P.pose_in[tid];
    for (int i = tid; i < n; i += PO_THREADS) {
        P.level[i] = 0;
        P.robust[i] = (P.num_trials_robust != 0) && P.huber[i] > 0.f;
        P.outlier[i] = 0;
    }
    __syncthreads();
    auto chi_at = [&](int i, const double* T, double* r, double* pc) {
        cam_point(T, P.pos_w + (size_t)i * 3, pc);
        double u = P.intr[0] * pc[0] / pc[2] + P.intr[2], v = P.intr[1] * pc[1] / pc[2] + P.intr[3];
        const bool eq = EQ && cam_is_equirect(P.intr);
        if (eq) equirect_project(P.intr, pc, &u, &v);
        const float* o = P.uvr + (size_t)i * 3;
        r[0] = (double)o[0] - u;
        r[1] = (double)o[1] - v;
        r[2] = (o[2] < 0.f || eq) ? 0.0 : (double)o[2] - (u - P.intr[4] / pc[2]);
        return (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * (double)P.inv_sigma_sq[i];
    };
    bool flag = false;      // the terminate action's own stop flag (no caller flag exists in this path)
    double last_chi = 0.0;  // terminate_action::_lastChi persists across optimize() calls
    int num_bad = 0, total_iters = 0, iters_at_classification = -1;
    const int rounds = P.num_trials_robust + P.num_trials;
    for (int trial = 0; trial < rounds; ++trial) {
        if (P.reset_flag_each_round) flag = false;
        bool ok = true;
        double lambda = 0.0, ni = 2.0;
        for (int it = 0; it < P.num_each_iter && !flag && ok; ++it) {
            // ---- linearise at s_T: H (21 unique), b (6), robust chi2, active count
            double acc[29];
            for (int k = 0; k < 29; ++k) acc[k] = 0.0;
            for (int i = tid; i < n; i += PO_THREADS) {
                if (P.level[i]) continue;
                double r[3], pc[3];
                const double chi = chi_at(i, s_T, r, pc);
                const double x = pc[0], y = pc[1], z = pc[2], z_sq = z * z, fx = P.intr[0], fy = P.intr[1], fxb = P.intr[4];
                const bool eq = EQ && cam_is_equirect(P.intr);
                const bool stereo = !(P.uvr[(size_t)i * 3 + 2] < 0.f) && !eq;
                double J[18];
                J[0] = x * y / z_sq * fx;
                J[1] = -(1.0 + (x * x / z_sq)) * fx;
                J[2] = y / z * fx;
                J[3] = -1.0 / z * fx;
                J[4] = 0.0;
                J[5] = x / z_sq * fx;
                J[6] = (1.0 + y * y / z_sq) * fy;
                J[7] = -x * y / z_sq * fy;
                J[8] = -x / z * fy;
                J[9] = 0.0;
                J[10] = -1.0 / z * fy;
                J[11] = y / z_sq * fy;
                J[12] = stereo ? J[0] - fxb * y / z_sq : 0.0;
                J[13] = stereo ? J[1] + fxb * x / z_sq : 0.0;
                J[14] = stereo ? J[2] : 0.0;
                J[15] = stereo ? J[3] : 0.0;
                J[16] = 0.0;
                J[17] = stereo ? J[5] - fxb / z_sq : 0.0;
                if (eq) equirect_jacobians(P.intr, pc, nullptr, nullptr, J);  // equirectangular_pose_opt_edge.h:70-118 (rows 0, 1)
                double rho0 = chi, rho1 = 1.0;
                if (P.robust[i]) huber(chi, (double)P.huber[i], &rho0, &rho1);
                const double w = (double)P.inv_sigma_sq[i] * rho1;
                int k = 0;
                for (int a = 0; a < 6; ++a)
                    for (int c = a; c < 6; ++c) {
                        acc[k] += J[a] * w * J[c] + J[6 + a] * w * J[6 + c] + J[12 + a] * w * J[12 + c];
                        ++k;
                    }
                for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * (-w * r[0]) + J[6 + a] * (-w * r[1]) + J[12 + a] * (-w * r[2]);
                acc[27] += rho0;
                acc[28] += 1.0;
            }
            stamp();
            po_reduce<29>(acc, s_wred, s_part, s_red);
            stamp();
            if (s_red[28] == 0.0) break;  // no active edge: nothing to optimise in this round (uniform)
            // H (symmetric, from its 21 unique sums) and b stay in LDS: only the solving thread needs H, every thread reads b for the step scale
            if (tid < 36) {
                const int a = tid / 6, c = tid - 6 * a, lo = min(a, c), hi = max(a, c);
                s_H[tid] = s_red[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
            }
            else if (tid < 42) s_b[tid - 36] = s_red[21 + (tid - 36)];
            double cur = s_red[27];
            __syncthreads();
            if (it == 0) {
                double md = 0.0;
                for (int a = 0; a < 6; ++a) md = fmax(md, fabs(s_H[7 * a]));
                lambda = 1e-5 * md;
                ni = 2.0;
            }
            double rho = 0.0;
            int qmax = 0;
            do {
                if (tid == 0) {
                    double x[6] = {0, 0, 0, 0, 0, 0};
                    const bool ok2 = po_chol6((const double*)s_H, lambda, (const double*)s_b, x);
                    if (!ok2)
                        for (int a = 0; a < 6; ++a) x[a] = 0.0;
                    for (int a = 0; a < 6; ++a) s_x[a] = x[a];
                    po_exp_mul(x, s_T, s_Tt);
                    s_ctl[2] = ok2 ? 1 : 0;
                }
                __syncthreads();
                stamp();
                double tmp[1] = {0.0};
                for (int i = tid; i < n; i += PO_THREADS) {
                    if (P.level[i]) continue;
                    double r[3], pc[3];
                    const double chi = chi_at(i, s_Tt, r, pc);
                    double rho0 = chi, rho1 = 1.0;
                    if (P.robust[i]) huber(chi, (double)P.huber[i], &rho0, &rho1);
                    tmp[0] += rho0;
                }
                po_reduce<1>(tmp, s_wred, s_part, s_red);
                stamp();
                double temp_chi = s_red[0];
                if (!s_ctl[2]) temp_chi = 1.7976931348623157e308;
                double scale = 1e-3;
                for (int a = 0; a < 6; ++a) scale += s_x[a] * (lambda * s_x[a] + s_b[a]);
                rho = (cur - temp_chi) / scale;
                __syncthreads();
                if (rho > 0 && isfinite(temp_chi)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2.0;
                    cur = temp_chi;
                    if (tid < 12) s_T[tid] = s_Tt[tid];
                    __syncthreads();
                }
                else {
                    lambda *= ni;
                    ni *= 2.0;
                    if (!isfinite(lambda)) break;
                }
                ++qmax;
            } while (rho < 0 && qmax < 10 && !flag);
            if (qmax == 10 || rho == 0 || !isfinite(lambda)) ok = false;
            ++total_iters;
            if (it == 0) last_chi = cur;
            else {
                const double gain = (last_chi - cur) / cur;
                last_chi = cur;
                if (gain >= 0 && gain < P.gain_thr) flag = true;
            }
        }
        // ---- chi-square re-classification of every edge at the current pose (:127-160).  A round whose LM loop did not run (the terminate
        //      action's flag stays up once the gain rule has raised it) leaves the pose where the previous round classified it: the same
        //      errors, the same flags, the same count -- only the Huber switch-off of :150-157 is left to do
        if (trial == 0 || total_iters != iters_at_classification) {
            double bad[1] = {0.0};
            for (int i = tid; i < n; i += PO_THREADS) {
                double r[3], pc[3];
                const double chi = chi_at(i, s_T, r, pc);
                const double thr = P.uvr[(size_t)i * 3 + 2] < 0.f ? (double)5.99146f : (double)7.81473f;
                const bool out = thr < chi;
                P.outlier[i] = out;
                P.level[i] = out;
                bad[0] += out;
                if (P.num_trials != 0 && trial + 1 == P.num_trials_robust) P.robust[i] = 0;
            }
            po_reduce<1>(bad, s_wred, s_part, s_red);
            stamp();
            num_bad = (int)s_red[0];
            __syncthreads();
            iters_at_classification = total_iters;
        }
        else if (P.num_trials != 0 && trial + 1 == P.num_trials_robust) {
            for (int i = tid; i < n; i += PO_THREADS) P.robust[i] = 0;
            __syncthreads();
        }
        if (n - num_bad < 5) break;
    }
    if (tid < 12) P.pose_out[tid] = s_T[tid];
    if (tid == 0) {
        P.result[0] = n - num_bad;
        P.result[1] = total_iters;
        P.result[2] = num_bad;
    }
    if constexpr (TRK) {  // results where the host reads them after the chain's one synchronisation: outlier flags per KEYPOINT, pose, counts
        (void)trk_nt;
        for (int i = tid; i < n; i += PO_THREADS) {
            const int k = P.trk_kp_of[i];
            const uint8_t o = P.outlier[i];
            P.trk_outlier_kp[k] = o;
            if (o) P.host_outlier_kp[k] = 1;
        }
        stamp();
        if (tid < 12) P.host_pose[tid] = s_T[tid];
        if (tid == 0) {
            P.host_result[0] = n - num_bad, P.host_result[1] = total_iters, P.host_result[2] = num_bad, P.host_result[3] = n;
            if (P.trk_counter_reset) *P.trk_counter_reset = 0;
        }
    }
}

// ================================================================================================ fused trial pipeline
// One damping trial = k_ba_lin (+ k_ba_lin_fin) when a linearisation is due, then k_ba_schur_rhs, k_ba_sys_fin, the solver,
// k_ba_update, k_ba_chi2, k_ba_decide: 8 launches instead of 14 (a launch of a trivial kernel costs ~4.7 us on this part when
// its first instruction depends on a word the previous kernel wrote).  Y = W Hll^-1 and Y bl are no longer materialised
// (E x 192 bytes per trial): the 3x3 inverse is recomputed where it is used.

// 64-lane sum of a double with DPP row scans (VALU only; __shfl_xor on a double is two ds_bpermute round trips per step).
// Fixed order => bit-reproducible.  The total is returned in every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_d<0x111, 0xF>(v);  // row_shr:1
    v += dpp_d<0x112, 0xF>(v);  // row_shr:2
    v += dpp_d<0x114, 0xF>(v);  // row_shr:4
    v += dpp_d<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_d<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_d<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ void wave_lds_sync() {  // orders the wave's own LDS writes before its later LDS reads (no workgroup barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Sums of NV per-lane values over the 64 lanes of a wave through a TRANSPOSE in wave-private LDS: 18 values at a time, every lane
// stores its 18 (conflict-free rows of 64), then lane (k, q) = (lane % 18, lane / 18), q < 3, adds a third of row k with 16-byte
// reads and lanes k < 18 add the three partials -- ~60 instructions per 18 values in a fixed order (bit-reproducible), against
// ~20 per VALUE for a DPP butterfly (wave_sum_dpp), which cost more than the arithmetic it followed in k_ba_schur_rhs / k_ba_lin.
// `sw`: WRED_DOUBLES doubles owned by the wave.  store(k, total) is called by ONE lane per value.
template <int NV, class Store>
__device__ __forceinline__ void wave_reduce_lds(const double (&acc)[NV], double* __restrict__ sw, int lane, Store store) {
    const int k = lane % 18, q = lane / 18;
    double* part = sw + 18 * WRED_PITCH;
#pragma unroll
    for (int h0 = 0; h0 < NV; h0 += 18) {
        const int nv = NV - h0 < 18 ? NV - h0 : 18;
#pragma unroll
        for (int i = 0; i < 18; ++i)
            if (i < nv) sw[i * WRED_PITCH + lane] = acc[h0 + i];
        wave_lds_sync();
        if (q < 3 && k < nv) {
            const double2* row = reinterpret_cast<const double2*>(sw + k * WRED_PITCH + 22 * q);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int i = 0; i < 11; ++i)
                if (q < 2 || i < 10) {  // lanes 0..21 | 22..43 | 44..63
                    const double2 v = row[i];
                    a += v.x;
                    b += v.y;
                }
            part[4 * k + q] = a + b;
        }
        wave_lds_sync();
        if (lane < nv) store(h0 + lane, (part[4 * lane] + part[4 * lane + 1]) + part[4 * lane + 2]);
        wave_lds_sync();
    }
}

// (Hll + lambda I)^-1 of a landmark, 6 unique entries; all zero (and *ok = false) when the block is singular
__device__ __forceinline__ bool lm_dinv(const double* __restrict__ H, double lambda, double* I) {
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[3] + lambda, e_ = H[4], f = H[5] + lambda;
    const double c00 = d * f - e_ * e_, c01 = e_ * c - b * f, c02 = b * e_ - d * c;
    const double det = a * c00 + b * c01 + c * c02;
    if (det == 0.0 || !isfinite(det)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) I[k] = 0.0;
        return false;
    }
    const double id = 1.0 / det;
    I[0] = c00 * id;
    I[1] = c01 * id;
    I[2] = c02 * id;
    I[3] = (a * f - c * c) * id;
    I[4] = (b * c - a * e_) * id;
    I[5] = (a * d - b * b) * id;
    return true;
}

// linearisation: blocks [0, nb_lm) = landmark side (Hll, bl, W; 8 lanes per landmark), the rest = pose side (D.lin_split
// workgroups per free pose, partial Hpp / bp; the host sizes the split for ~3 edges per lane -- with a fixed split of 16 a
// config-5 workgroup held 150 edges, one per lane at most, and the 8 k workgroups spent their time in the 27-value reduction and
// in launch latency at 2 workgroups per CU).  EQ: see edge_linearize_core.
#define LIN_SPLIT_MAX 16
#define LIN_EDGES_PER_BLOCK 768
template <bool EQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EQ ? 2 : 4, 4))) void k_ba_lin(BaDev D, int nb_lm, int blk0) {
    if (D.ctl->phase != 0) return;
    const double* pose_cur = st_pose(D, 0);
    const double* pt_cur = st_pt(D, 0);
    const int bid = blockIdx.x + blk0;
    // one LDS block for both sides: the landmark side stages the W records of a wave's 64 lanes (144 bytes each), the pose side its reductions
    __shared__ __attribute__((aligned(16))) double s_buf[4 * 64 * 18];
    static_assert(4 * 64 * 18 >= 2 * WRED_DOUBLES, "the pose side's two transposition buffers live in the staging block");
    if (bid < nb_lm) {
        __shared__ int s_we[4][64];
        const int t = bid * 256 + threadIdx.x;
        const int l = min(t / LM_LANES, D.L - 1), sub = t % LM_LANES;
        const bool in_range = t / LM_LANES < D.L;
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        double* const my_w = s_buf + (size_t)wv * (64 * 18) + ln * 18;
        double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        const bool lfree = D.pt_free[l];
        const int hi = in_range ? D.lm_off[l + 1] : 0;
        // wave-uniform trip count: every lane takes part in the store of the wave's records
        for (int e = in_range ? D.lm_off[l] + sub : 0; __builtin_amdgcn_ballot_w64(e < hi) != 0; e += LM_LANES) {
            int we = -1;  // the edge whose W record this lane produced in this trip
            if (e < hi && !D.e_level[e]) {
                const int slot = D.pose_slot[D.e_pose[e]];
                if (lfree || slot >= 0) {
                    EdgeLin o;
                    edge_linearize<EQ>(D, pose_cur, pt_cur, e, o);
                    if (lfree) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const double a0 = o.A[3 * d], a1 = o.A[3 * d + 1], a2 = o.A[3 * d + 2];
                            const double wr = -o.w * o.r[d];
                            b[0] += a0 * wr;
                            b[1] += a1 * wr;
                            b[2] += a2 * wr;
                            H[0] += a0 * o.w * a0;
                            H[1] += a0 * o.w * a1;
                            H[2] += a0 * o.w * a2;
                            H[3] += a1 * o.w * a1;
                            H[4] += a1 * o.w * a2;
                            H[5] += a2 * o.w * a2;
                        }
                        if (slot >= 0) {
                            we = e;
#pragma unroll
                            for (int i = 0; i < 6; ++i)
#pragma unroll
                                for (int j = 0; j < 3; ++j)
                                    my_w[3 * i + j] = o.B[i] * o.w * o.A[j] + o.B[6 + i] * o.w * o.A[3 + j] + o.B[12 + i] * o.w * o.A[6 + j];
                        }
                    }
                }
            }
            // The records leave through LDS: piece g = k * 64 + lane of the wave's 64 x 9 sixteen-byte pieces belongs to the record of lane
            // g / 9, so consecutive lanes store consecutive 16 bytes (a wave's edges are contiguous in memory: the observations are sorted by
            // landmark).  Stored per lane -- nine 16-byte stores 144 bytes apart -- every store instruction put 64 partial lines to the L2:
            // 10.8 M write requests per config-5 launch, 79 us for the landmark side.
            s_we[wv][ln] = we;
            wave_lds_sync();
            {
                const double2* const src = reinterpret_cast<const double2*>(s_buf + (size_t)wv * (64 * 18));
                char* const Wg = reinterpret_cast<char*>(D.W);
#pragma unroll 3
                for (int k = 0; k < 9; ++k) {
                    const int g = k * 64 + ln, r = g / 9, piece = g - 9 * r, er = s_we[wv][r];
                    // (32-bit byte offset from the uniform base: svgpu_ba.hip refuses E >= 2^32 / 144)
                    if (er >= 0) *reinterpret_cast<double2*>(Wg + ((unsigned)er * 144u + (unsigned)piece * 16u)) = src[g];
                }
            }
            wave_lds_sync();
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) H[k] = group_sum8(H[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) b[k] = group_sum8(b[k]);
        double m = 0.0;
        if (in_range && lfree && sub == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) D.Hll[(size_t)l * 6 + k] = H[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) D.bl[(size_t)l * 3 + k] = b[k];
            m = fmax(fabs(H[0]), fmax(fabs(H[3]), fabs(H[5])));
        }
        if (D.ctl->it == 0) {  // computeLambdaInit: the landmark part of max |diagonal| (uniform branch), one value per workgroup;
            // k_ba_lin_fin folds them.  (An atomicMax per wave on the control block -- 25 k of them on one address at config 5 --
            // made the first linearisation of every optimize() call 1.37 ms instead of 70 us.)
            __shared__ double s_mx[4];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
            if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = m;
            __syncthreads();
            if (threadIdx.x == 0) D.lm_max[bid] = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
        }
        return;
    }
    const int split = D.lin_split;
    const int u = bid - nb_lm, s = u / split, share = u - s * split;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    const int pose = D.slot_pose[s], lo = D.pe_off[pose], n = D.pe_off[pose + 1] - lo;
    const int q0 = lo + (int)((long long)n * share / split), q1 = lo + (int)((long long)n * (share + 1) / split);
    for (int q = q0 + threadIdx.x; q < q1; q += 256) {
        const int e = D.pe_idx[q];
        if (D.e_level[e]) continue;  // lists are built once per call; excluded edges stay listed
        EdgeLin o;
        // the observation comes from the POSE-MAJOR copy (contiguous along the pose's list) when the arena holds one
        if (D.pm_point) edge_linearize_core<EQ>(D, pose_cur, pt_cur, e, pose, D.pm_point[q], D.pm_uvr + (size_t)q * 3, (double)D.pm_w[q], D.pm_hub[q], o);
        else edge_linearize<EQ>(D, pose_cur, pt_cur, e, o);
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                acc[k] += o.B[i] * o.w * o.B[j] + o.B[6 + i] * o.w * o.B[6 + j] + o.B[12 + i] * o.w * o.B[12 + j];
                ++k;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[21 + i] += o.B[i] * (-o.w * o.r[0]) + o.B[6 + i] * (-o.w * o.r[1]) + o.B[12 + i] * (-o.w * o.r[2]);
    }
    // the four waves reduce two at a time through two wave-private transposition buffers (half the LDS => one more workgroup per CU)
    __shared__ double s_w[4][27];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 2) wave_reduce_lds<27>(acc, s_buf + wave * WRED_DOUBLES, lane, [&](int k, double t) { s_w[wave][k] = t; });
    __syncthreads();
    if (wave >= 2) wave_reduce_lds<27>(acc, s_buf + (wave - 2) * WRED_DOUBLES, lane, [&](int k, double t) { s_w[wave][k] = t; });
    __syncthreads();
    if (threadIdx.x < 27) D.lp_part[((size_t)s * split + share) * 27 + threadIdx.x] = ((s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + s_w[2][threadIdx.x]) + s_w[3][threadIdx.x];
}

// one workgroup: pose blocks from their D.lin_split partials (share order), then -- unless the pose blocks still have to be summed
// over the ranks of a sharded solve -- the pose part of computeLambdaInit and the start-of-trial bookkeeping (k_ba_prepare)
__device__ __forceinline__ void ctl_prepare(BaDev& D, double max_diag) {
    BaCtl& c = *D.ctl;
    if (c.phase == 0) {
        if (c.it == 0) {
            c.lambda = 1e-5 * max_diag;
            c.ni = 2.0;
        }
        c.qmax = 0;
        c.rho = 0.0;
        c.phase = 1;
    }
    c.solve_failed = 0;
    c.pcg_done = 0;
    c.pcg_fail = 0;
    c.pcg_it = 0;
}
#define LIN_FIN_THREADS 1024
__global__ __launch_bounds__(LIN_FIN_THREADS) void k_ba_lin_fin(BaDev D, int do_prepare, int nb_lm) {
    const int phase = D.ctl->phase;
    if (phase == 2) return;
    __shared__ double s_m[LIN_FIN_THREADS / 64];
    double m = 0.0;
    if (phase == 0) {
        for (int item = threadIdx.x; item < D.nP * 27; item += LIN_FIN_THREADS) {
            const int s = item / 27, k = item - 27 * s;
            double t = 0.0;
#pragma unroll 4
            for (int h = 0; h < D.lin_split; ++h) t += D.lp_part[((size_t)s * D.lin_split + h) * 27 + k];
            if (k < 21) {
                int i = 0, base = 0;  // k -> (i, j), i <= j, rows of 6, 5, 4, ... entries
                while (k >= base + (6 - i)) {
                    base += 6 - i;
                    ++i;
                }
                const int j = i + (k - base);
                D.Hpp[(size_t)s * 36 + 6 * i + j] = t;
                D.Hpp[(size_t)s * 36 + 6 * j + i] = t;
                if (i == j) m = fmax(m, fabs(t));
            }
            else D.bp[(size_t)s * 6 + (k - 21)] = t;
        }
    }
    double ml = 0.0;  // landmark part of computeLambdaInit: the per-workgroup maxima of k_ba_lin
    if (phase == 0 && D.ctl->it == 0)
        for (int k = threadIdx.x; k < nb_lm; k += LIN_FIN_THREADS) ml = fmax(ml, D.lm_max[k]);
    if (!do_prepare) m = 0.0;  // sharded: the pose part comes from the summed blocks (k_ba_maxdiag)
    m = fmax(m, ml);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = 0.0;
        for (int k = 0; k < LIN_FIN_THREADS / 64; ++k) m = fmax(m, s_m[k]);
        m = fmax(m, __longlong_as_double((long long)D.ctl->max_diag_bits));
        if (do_prepare) ctl_prepare(D, m);
        else if (phase == 0 && D.ctl->it == 0) D.ctl->max_diag_bits = (unsigned long long)__double_as_longlong(m);
    }
}

// Reduced camera system, one WAVE per work unit (one unit per workgroup, no workgroup barrier):
//   unit blk * nshare + sh = share `sh` of block (a, b): sum over its (edge, edge) pairs of W_i Hll^-1 W_j^T -> sc_part
//   a share of a DIAGONAL block (a, a) walks pairs (i, i) over the edges of pose a: W_i and Hll^-1 are in hand, so the edge's term of
//   the right-hand side, W_i Hll^-1 bl, is summed along (-> rhs_part, nshare shares per pose) and W_j is not fetched a second time.
//   (Round 2 ran the right-hand side as nP * 16 units of its own behind the blocks: a second pass over every W record, 28 of the
//   kernel's 181 us at config 5.)
// A lane walks several pairs and the 36 (6) sums are reduced once per wave: with one pair per thread the 36 shuffle reductions
// cost more than the 162 multiply-adds of the pair.  The multiply-adds are explicit fma() (the build runs with -ffp-contract=off for the
// bit-exact fp32 front end; here contraction is wanted: 40 % fewer VALU instructions, one rounding less per term, still a fixed order).
//
// COOPERATIVE GATHER.  A lane needs the 144-byte W record of each edge of its pair and the 48-byte Hll of the landmark.  Fetched per
// lane (nine 16-byte loads from 64 different records) every wave-level load touches 64 cache lines; the vector L1 looks up one line
// per cycle, and at config 5 its ~1e8 lookups per launch (PMC TCP_TOTAL_CACHE_ACCESSES) were 170 of the kernel's 273 us.  Here the
// wave fetches the 64 records of a trip TOGETHER: piece g = k * 64 + lane of the 64 * NP pieces belongs to record g / NP, so
// consecutive lanes read consecutive 16 bytes (a line serves 8 lanes instead of 1), the pieces land in wave-private LDS in linear
// order (conflict-free) and every lane reads its own record back with ds_read_b128 (144-byte pitch: conflict-free).
#define RHS_SPLIT 16
#define SCH_W_DOUBLES (64 * 18)                      // one side's W records of a trip
#define SCH_H_DOUBLES (64 * 6)                       // the trip's Hll blocks
#define SCH_WAVE_DOUBLES (2 * SCH_W_DOUBLES + SCH_H_DOUBLES + 96 + 16)  // both sides + three index rows (64 ints each), padded; >= WRED_DOUBLES
#define SCH_WAVES 1  // waves (= units) per workgroup: nothing is shared between them, and 22 KB of LDS per wave leaves 7 waves per CU
template <int NP, typename T>
__device__ __forceinline__ void coop_issue(const T* __restrict__ base, const int* __restrict__ idx, int rec_pitch /* in T */, int lane, T (&v)[NP]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int g = k * 64 + lane, r = g / NP, piece = g - r * NP;
        // 32-bit byte offset from the uniform base (scalar base + vector offset addressing: one address register per load instead of two;
        // the arrays gathered here stay below 4 GB: svgpu_ba.hip refuses E >= 2^32 / 144)
        const unsigned off = ((unsigned)idx[r] * (unsigned)rec_pitch + (unsigned)piece) * (unsigned)sizeof(T);
        v[k] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
    }
}
template <int NP, typename T>
__device__ __forceinline__ void coop_commit(const T (&v)[NP], T* __restrict__ dst, int lane) {
#pragma unroll
    for (int k = 0; k < NP; ++k) dst[k * 64 + lane] = v[k];
}
#define SCH_T(i) do { if (D.dbg_schur_on && lane == 0) D.dbg[8 * (size_t)unit + (i)] = wall_clock64(); } while (0)
// pairs [q0, q1) of one block (DIAG: a diagonal block; the two code paths are separate loops so that neither carries the other's registers)
template <bool DIAG>
__device__ __forceinline__ void schur_unit(const BaDev& D, int unit, int lane, int q0, int q1, double lambda, double* __restrict__ sw,
                                           double* __restrict__ out, double* __restrict__ rhs_out) {
    double2* const s_W = reinterpret_cast<double2*>(sw);                                  // 64 records x 9 pieces (W_i)
    double2* const s_W2 = reinterpret_cast<double2*>(sw + SCH_W_DOUBLES);                 // 64 records x 9 pieces (W_j; DIAG: 64 x bl)
    double2* const s_H = reinterpret_cast<double2*>(sw + 2 * SCH_W_DOUBLES);              // 64 records x 3 pieces
    int* const s_ix = reinterpret_cast<int*>(sw + 2 * SCH_W_DOUBLES + SCH_H_DOUBLES);     // [0,64) e_i  [64,128) e_j  [128,192) landmark
    const double2* const Wg = reinterpret_cast<const double2*>(D.W);
    const double2* const Hg = reinterpret_cast<const double2*>(D.Hll);
    double acc[36], accr[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    constexpr bool diag = DIAG;
    // the pair and its landmark (blk_pair_l: Hll does not wait for e_point[pair.x]) are fetched one trip ahead; lanes past the end of the
    // share repeat its last pair (they take part in the gathers, not in the sums)
    int2 pr = make_int2(0, 0);
    int l = 0;
    SCH_T(1);
    if (q0 < q1) {
        const int qq = min(q0 + lane, q1 - 1);
        pr = D.blk_pairs[qq];
        l = D.blk_pair_l[qq];
    }
    for (int qb = q0; qb < q1; qb += 64) {  // wave-uniform
        const bool mine = qb + lane < q1;
        int2 prn = pr;
        int ln = l;
        if (qb + 64 < q1) {
            const int qq = min(qb + 64 + lane, q1 - 1);
            prn = D.blk_pairs[qq];
            ln = D.blk_pair_l[qq];
        }
        s_ix[lane] = pr.x;
        s_ix[64 + lane] = pr.y;
        s_ix[128 + lane] = l;
        wave_lds_sync();
        if (!diag) {
            double2 hv[3], wa[9], wb[9];
            coop_issue<3>(Hg, s_ix + 128, 3, lane, hv);
            coop_issue<9>(Wg, s_ix, 9, lane, wa);
            coop_issue<9>(Wg, s_ix + 64, 9, lane, wb);
            coop_commit<3>(hv, s_H, lane);
            coop_commit<9>(wa, s_W, lane);
            coop_commit<9>(wb, s_W2, lane);
        }
        else {  // W_j is W_i (a pose sees a landmark once; the rare second observation is fetched per lane below); bl rides in its place
            double2 hv[3], wa[9];
            double bv[3];
            coop_issue<3>(Hg, s_ix + 128, 3, lane, hv);
            coop_issue<9>(Wg, s_ix, 9, lane, wa);
            coop_issue<3>(D.bl, s_ix + 128, 3, lane, bv);
            coop_commit<3>(hv, s_H, lane);
            coop_commit<9>(wa, s_W, lane);
            coop_commit<3>(bv, reinterpret_cast<double*>(s_W2), lane);
        }
        wave_lds_sync();
        double I[6], Hl[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double2 t = s_H[3 * lane + k];
            Hl[2 * k] = t.x;
            Hl[2 * k + 1] = t.y;
        }
        lm_dinv(Hl, lambda, I);
        double y[18];
        {
            double wi[18];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double2 t = s_W[9 * lane + k];
                wi[2 * k] = t.x;
                wi[2 * k + 1] = t.y;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double w0 = wi[3 * i], w1 = wi[3 * i + 1], w2 = wi[3 * i + 2];
                y[3 * i] = fma(w2, I[2], fma(w1, I[1], w0 * I[0]));
                y[3 * i + 1] = fma(w2, I[4], fma(w1, I[3], w0 * I[1]));
                y[3 * i + 2] = fma(w2, I[5], fma(w1, I[4], w0 * I[2]));
            }
        }
        double w[18];
        {
            const double2* const src = diag ? s_W : s_W2;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double2 t = src[9 * lane + k];
                w[2 * k] = t.x;
                w[2 * k + 1] = t.y;
            }
        }
        if (diag) {
            if (pr.x != pr.y) {  // two observations of one landmark by one keyframe
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const double2 t = Wg[(size_t)pr.y * 9 + k];
                    w[2 * k] = t.x;
                    w[2 * k + 1] = t.y;
                }
            }
            else if (mine) {  // the edge's term of the right-hand side: W_i Hll^-1 bl
                const double* const bl = reinterpret_cast<const double*>(s_W2) + 3 * lane;
                const double b0 = bl[0], b1 = bl[1], b2 = bl[2];
#pragma unroll
                for (int i = 0; i < 6; ++i) accr[i] = fma(y[3 * i + 2], b2, fma(y[3 * i + 1], b1, fma(y[3 * i], b0, accr[i])));
            }
        }
        if (mine) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[6 * i + j] = fma(y[3 * i + 2], w[3 * j + 2], fma(y[3 * i + 1], w[3 * j + 1], fma(y[3 * i], w[3 * j], acc[6 * i + j])));
        }
        wave_lds_sync();  // the records are consumed: the next trip may overwrite the buffer
        if (qb == q0) SCH_T(2);
        pr = prn;
        l = ln;
    }
    SCH_T(3);
    wave_reduce_lds<36>(acc, sw, lane, [&](int k, double t) { out[k] = t; });
    if (diag) {  // this share of the pose's right-hand side (sys_fin adds the shares in order)
        double mine_v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double t = wave_sum_dpp(accr[k]);
            mine_v = (lane == k) ? t : mine_v;
        }
        if (lane < 6) rhs_out[lane] = mine_v;
    }
    SCH_T(4);
}

// Right-hand side as units of its own (small systems: the chip is not full, these units run beside the block shares):
// share `ru % RHS_SPLIT` of the edges of pose slot `ru / RHS_SPLIT`: sum of W_e Hll^-1 bl
__device__ __forceinline__ void rhs_unit(const BaDev& D, int ru, int lane, double lambda, double* __restrict__ sw, double* __restrict__ rhs_part) {
    double2* const s_W = reinterpret_cast<double2*>(sw);
    double2* const s_H = reinterpret_cast<double2*>(sw + 2 * SCH_W_DOUBLES);
    int* const s_ix = reinterpret_cast<int*>(sw + 2 * SCH_W_DOUBLES + SCH_H_DOUBLES);
    const double2* const Wg = reinterpret_cast<const double2*>(D.W);
    const double2* const Hg = reinterpret_cast<const double2*>(D.Hll);
    if (ru >= D.nP * RHS_SPLIT) return;
    const int s = ru / RHS_SPLIT, share = ru - s * RHS_SPLIT;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int pose = D.slot_pose[s], lo = D.pe_off[pose], n = D.pe_off[pose + 1] - lo;
    const int q0 = lo + (int)((long long)n * share / RHS_SPLIT), q1 = lo + (int)((long long)n * (share + 1) / RHS_SPLIT);
    for (int qb = q0; qb < q1; qb += 64) {  // wave-uniform; same cooperative gathers (W, Hll; bl is 24 bytes: per lane)
        const int q = min(qb + lane, q1 - 1);
        const int e = D.pe_idx[q];
        const int l = D.pm_point ? D.pm_point[q] : D.e_point[e];  // pose-major copy: no second hop behind pe_idx
        const bool mine = qb + lane < q1 && D.pt_free[l] && !D.e_level[e];
        s_ix[lane] = e;
        s_ix[128 + lane] = l;
        wave_lds_sync();
        double2 hv[3], wa[9];
        coop_issue<3>(Hg, s_ix + 128, 3, lane, hv);
        coop_issue<9>(Wg, s_ix, 9, lane, wa);
        const double* bl = D.bl + (size_t)l * 3;
        const double b0 = bl[0], b1 = bl[1], b2 = bl[2];
        coop_commit<3>(hv, s_H, lane);
        coop_commit<9>(wa, s_W, lane);
        wave_lds_sync();
        double I[6], Hl[6], Wd[18];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double2 t = s_H[3 * lane + k];
            Hl[2 * k] = t.x;
            Hl[2 * k + 1] = t.y;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double2 t = s_W[9 * lane + k];
            Wd[2 * k] = t.x;
            Wd[2 * k + 1] = t.y;
        }
        wave_lds_sync();  // consumed
        if (mine) {
            lm_dinv(Hl, lambda, I);
            const double d0 = I[0] * b0 + I[1] * b1 + I[2] * b2;
            const double d1 = I[1] * b0 + I[3] * b1 + I[4] * b2;
            const double d2 = I[2] * b0 + I[4] * b1 + I[5] * b2;
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = fma(Wd[3 * i + 2], d2, fma(Wd[3 * i + 1], d1, fma(Wd[3 * i], d0, acc[i])));
        }
    }
    double mine_v = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double t = wave_sum_dpp(acc[k]);
        mine_v = (lane == k) ? t : mine_v;
    }
    if (lane < 6) rhs_part[(size_t)ru * 6 + lane] = mine_v;
}

// (two waves per SIMD pinned: left alone, the register allocator takes 266 registers for the two unit forms and halves the occupancy)
__global__ __launch_bounds__(64 * SCH_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ba_schur_rhs(BaDev D, int nshare, double* __restrict__ rhs_part, int xcd_order) {
    if (D.ctl->phase != 1) return;
    __shared__ __attribute__((aligned(16))) double s_wave[SCH_WAVES][SCH_WAVE_DOUBLES];
    static_assert(SCH_WAVE_DOUBLES >= WRED_DOUBLES, "the reduction reuses the gather buffer");
    const double lambda = D.ctl->lambda;
    // XCD-aware order: workgroup i runs on XCD i % 8, each with its own 4 MB L2.  Consecutive units walk the blocks (a, a), (a, a + 1), ...
    // of one block row, which all read the W records of pose a and its neighbours: XCD x takes the x-th CONTIGUOUS eighth of the units, so
    // that the rows in flight on one L2 are ~3 instead of ~21 (half of the L2 requests missed at config 5 with the round-robin order:
    // 980 MB fetched per launch, 250 MB with this order; W alone is 173 MB).  The unit count is padded to a multiple of 8 by the launcher.
    if (D.unit_rec) {  // chunk-major units: XCD x walks the x-th contiguous eighth of the execution order, i.e. its own sequence of landmark chunks
        const int nu8 = (D.num_units + 7) & ~7;
        const int pos = ((int)blockIdx.x & 7) * (nu8 >> 3) + ((int)blockIdx.x >> 3);
        if (pos >= D.num_units) return;
        const int lane = threadIdx.x & 63;
        const int4 rec = D.unit_rec[pos];
        const int2 ab = D.blk_ab[rec.z];
        const int unit = rec.w;
        double* const sw = s_wave[0];
        SCH_T(0);
        if (D.dbg_schur_on && lane == 0) D.dbg[8 * (size_t)unit + 5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        if (ab.x == ab.y) schur_unit<true>(D, unit, lane, rec.x, rec.y, lambda, sw, D.sc_part + (size_t)unit * 36, D.rhs_unit + (size_t)unit * 6);
        else schur_unit<false>(D, unit, lane, rec.x, rec.y, lambda, sw, D.sc_part + (size_t)unit * 36, nullptr);
        return;
    }
    const int n_schur = D.NB * nshare;
    int wg = blockIdx.x;
    if (xcd_order) {
        const int ns8 = (n_schur + 7) & ~7;
        wg = (wg & 7) * (ns8 >> 3) + (wg >> 3);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, unit = wg * SCH_WAVES + wave;
    double* const sw = s_wave[wave];
    SCH_T(0);
    if (D.dbg_schur_on && lane == 0) D.dbg[8 * (size_t)unit + 5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) /* HW_ID */ | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID */ << 32);
    if (unit >= n_schur) {
        if (!xcd_order) rhs_unit(D, unit - n_schur, lane, lambda, sw, rhs_part);  // (folded into the diagonal blocks otherwise)
        return;
    }
    const int blk = unit / nshare, share = unit - blk * nshare;
    const int lo = D.blk_off[blk], np = D.blk_off[blk + 1] - lo;
    const int2 ab = D.blk_ab[blk];
    // (an even split; shares of whole trips -- multiples of 64 pairs, 67 k trips instead of 80 k at config 5 -- measured no faster there and
    //  slower in local BA, 13.4 -> 14.7 us, and so did the linearisation behind it, 11.6 -> 13.1 us)
    const int q0 = lo + (int)((long long)np * share / nshare), q1 = lo + (int)((long long)np * (share + 1) / nshare);
    double* const out = D.sc_part + (size_t)unit * 36;
    if (q0 >= q1) {  // wave-uniform
        if (lane < 36) out[lane] = 0.0;
        if (xcd_order && ab.x == ab.y && lane < 6) rhs_part[((size_t)ab.x * RHS_SPLIT + share) * 6 + lane] = 0.0;
        return;
    }
    if (xcd_order && ab.x == ab.y) schur_unit<true>(D, unit, lane, q0, q1, lambda, sw, out, rhs_part + ((size_t)ab.x * RHS_SPLIT + share) * 6);  // wave-uniform
    else schur_unit<false>(D, unit, lane, q0, q1, lambda, sw, out, nullptr);
}

// kept blocks S_ab = [a == b](Hpp_a + lambda I) - sum of the shares, right-hand side g_a = bp_a - sum of the shares
__global__ __launch_bounds__(256) void k_ba_sys_fin(BaDev D, int nshare, const double* __restrict__ rhs_part, int rhs_shares) {
    if (D.ctl->phase != 1) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < D.NB * 36) {
        const int blk = k / 36, t = k - 36 * blk;
        double sum = 0.0;
        if (D.unit_rec) {  // the units of the block, in landmark-chunk order
            const int u0 = D.blk_unit_off[blk], u1 = D.blk_unit_off[blk + 1];
#pragma unroll 4
            for (int u = u0; u < u1; ++u) sum += D.sc_part[(size_t)u * 36 + t];
        }
        else
#pragma unroll 8
        for (int h = 0; h < nshare; ++h) sum += D.sc_part[((size_t)blk * nshare + h) * 36 + t];  // (unrolled: the loads of a share run are independent)
        const int2 ab = D.blk_ab[blk];
        double v = -sum;
        if (ab.x == ab.y) {
            v += D.Hpp[(size_t)ab.x * 36 + t];
            if (t % 7 == 0 && (D.lam_slot ? D.lam_slot[ab.x] != 0 : D.add_lambda != 0)) v += D.ctl->lambda;
        }
        D.Sblk[k] = v;
    }
    else if (k - D.NB * 36 < D.n) {
        const int r = k - D.NB * 36;
        const int s = r / 6, i = r - 6 * s;
        double sum = 0.0;
        if (D.unit_rec) {
            const int blk = D.diag_blk[s], u0 = D.blk_unit_off[blk], u1 = D.blk_unit_off[blk + 1];
            for (int u = u0; u < u1; ++u) sum += D.rhs_unit[(size_t)u * 6 + i];
        }
        else
#pragma unroll 8
        for (int h = 0; h < rhs_shares; ++h) sum += rhs_part[((size_t)s * RHS_SPLIT + h) * 6 + i];  // nshare shares of block (s, s), or the RHS_SPLIT units of the pose
        D.g[r] = D.bp[r] - sum;
    }
}

// ---- landmark-sized exchanges of a sharded solve, on the device (svgpu_ba.hip)
// contract check: xch[l] = this rank holds observations of landmark l; xch[L] = its stop-pointer vote
__global__ __launch_bounds__(256) void k_ba_owned_mark(const int* __restrict__ lm_off, int L, double* __restrict__ xch, double stop_vote) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l < L) xch[l] = lm_off[l + 1] > lm_off[l] ? 1.0 : 0.0;
    else if (l == L) xch[L] = stop_vote;
}
// after the sum: verdict[0] = a landmark has two owners, verdict[1] = the summed stop-pointer vote; any_owner[l] for the final exchange
__global__ __launch_bounds__(256) void k_ba_owned_check(const double* __restrict__ xch, int L, uint8_t* __restrict__ any_owner, double* __restrict__ verdict) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l < L) {
        const double v = xch[l];
        any_owner[l] = v > 0.5;
        if (v > 1.5) verdict[0] = 1.0;  // (benign race: every writer stores 1)
    }
    else if (l == L) verdict[1] = xch[L];
}
// final estimate of the landmarks: dir 0: xch[3 l ..] = the point when this rank owns it, else 0; dir 1 (after the sum): owned landmarks take the sum
__global__ __launch_bounds__(256) void k_ba_points_share(BaDev D, double* __restrict__ points_out, double* __restrict__ xch, int dir) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * (size_t)D.L) return;
    const int l = (int)(i / 3);
    const int* const off = D.lm_off_caller ? D.lm_off_caller : D.lm_off;  // (points_out and xch are in the caller's numbering)
    if (dir == 0) xch[i] = off[l + 1] > off[l] ? points_out[i] : 0.0;
    else if (D.any_owner[l]) points_out[i] = xch[i];
}

// Keyframe-segment exchange of a sharded solve (svgpu_ba.hip): only the kept blocks between two separator rows and the separator rows of
// the right-hand side cross ranks.  dir 0: Sblk / g -> the compact buffer [36 nb | 6 ns]; dir 1: back.
__global__ __launch_bounds__(256) void k_ba_xs_move(BaDev D, const int* __restrict__ blk_idx, int nb, const int* __restrict__ slot_idx, int ns, double* __restrict__ buf, int dir) {
    if (D.ctl->phase != 1) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    double* src;
    if (k < nb * 36) src = D.Sblk + (size_t)blk_idx[k / 36] * 36 + k % 36;
    else if (k < nb * 36 + ns * 6) {
        const int r = k - nb * 36;
        src = D.g + (size_t)slot_idx[r / 6] * 6 + r % 6;
    }
    else return;
    if (dir == 0) buf[k] = *src;
    else *src = buf[k];
}
// Block-Jacobi preconditioned conjugate gradients on the reduced camera system with EVERYTHING in the LDS of one workgroup: the
// kept 6x6 blocks (36 doubles each), the block-row lists, the inverse diagonal blocks and the five vectors.  A local-BA system
// (6 * free keyframes <= a few hundred unknowns, <= ~450 blocks) converges to 1e-10 in a few dozen iterations of ~0.4 us each;
// (the solver policy in svgpu_ba.hip says when this is preferred to the dense LL^T, k_ba_chol_tile).
// Dynamic LDS layout (doubles): blk[36 NB] | Minv[36 nP] | x r z p q [5 n] | part[6 nent] | wsum[48]; then ints: rowoff[nP + 1], ent[2 nent]
#define PL_THREADS 512
__global__ __launch_bounds__(PL_THREADS) void k_ba_pcg_lds(BaDev D, int nent) {
    if (D.ctl->phase != 1) return;
    extern __shared__ double s_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = D.n, nP = D.nP, NB = D.NB;
    double* blk = s_d;
    double* Minv = blk + 36 * NB;
    double* x = Minv + 36 * nP;
    double* r = x + n;
    double* z = r + n;
    double* p = z + n;
    double* q = p + n;
    double* part = q + n;
    double* wsum = part + 6 * nent;
    int* rowoff = reinterpret_cast<int*>(wsum + 48);
    int2* ent = reinterpret_cast<int2*>(rowoff + ((nP + 2) & ~1));
    __shared__ int s_bad;
    for (int k = tid; k < 36 * NB; k += PL_THREADS) blk[k] = D.Sblk[k];
    for (int k = tid; k <= nP; k += PL_THREADS) rowoff[k] = D.prow_off[k];
    for (int k = tid; k < nent; k += PL_THREADS) ent[k] = D.prow_ent[k];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // inverse diagonal blocks (thread per block row: LL^T, L^-1, L^-T L^-1)
    if (tid < nP) {
        const double* B = blk + 36 * D.diag_blk[tid];
        double L[36], X[36];
        bool bad = false;
        for (int j = 0; j < 6; ++j) {
            double d = B[7 * j];
            for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
            if (!(d > 0.0)) {
                bad = true;
                d = 1.0;
            }
            d = sqrt(d);
            L[7 * j] = d;
            for (int i = j + 1; i < 6; ++i) {
                double sum = B[6 * j + i];
                for (int k = 0; k < j; ++k) sum -= L[6 * i + k] * L[6 * j + k];
                L[6 * i + j] = sum / d;
            }
        }
        for (int c = 0; c < 6; ++c)
            for (int i = 0; i < 6; ++i) {
                double sum = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) sum -= L[6 * i + k] * X[6 * k + c];
                X[6 * i + c] = (i >= c) ? sum / L[7 * i] : 0.0;
            }
        double* M = Minv + 36 * tid;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double sum = 0.0;
                for (int k = (i > j ? i : j); k < 6; ++k) sum += X[6 * k + i] * X[6 * k + j];
                M[6 * i + j] = sum;
            }
        if (bad) s_bad = 1;
    }
    if (tid < n) {
        x[tid] = 0.0;
        r[tid] = D.g[tid];
    }
    __syncthreads();
    // fixed-order block-wide sum of one value per thread; result in every thread
    auto bsum = [&](double v, int slot) -> double {
        const double t = wave_sum_dpp(v);
        if (lane == 0) wsum[slot * 8 + wave] = t;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < PL_THREADS / 64; ++w) tot += wsum[slot * 8 + w];
        return tot;
    };
    auto precond = [&](int row) -> double {  // z_row = sum_j Minv[a][i][j] r[6 a + j]
        const int a = row / 6, i = row - 6 * a;
        const double* M = Minv + 36 * a + 6 * i;
        const double* ra = r + 6 * a;
        return M[0] * ra[0] + M[1] * ra[1] + M[2] * ra[2] + M[3] * ra[3] + M[4] * ra[4] + M[5] * ra[5];
    };
    double zi = 0.0, ri = 0.0;
    if (tid < n) {
        ri = r[tid];
        zi = precond(tid);
        z[tid] = zi;
        p[tid] = zi;
    }
    double rz = bsum(ri * zi, 0);
    const double rr0 = bsum(ri * ri, 1);
    const double tol2 = D.ctl->pcg_tol2;
    const int max_it = D.ctl->pcg_max_it;
    int it = 0, fail = s_bad ? 1 : 0;
    double rr = rr0;
    __syncthreads();
    while (!fail && rr > tol2 * rr0 && it < max_it) {
        // q = S p: one task per (row entry, component), then the entries of a row in list order
        for (int t = tid; t < 6 * nent; t += PL_THREADS) {
            const int e = t / 6, i = t - 6 * e;
            const int2 en = ent[e];
            const double* B = blk + 36 * (en.x & 0x3fffffff);
            const double* pb = p + 6 * en.y;
            double v;
            if (en.x >> 30 & 1) v = B[i] * pb[0] + B[6 + i] * pb[1] + B[12 + i] * pb[2] + B[18 + i] * pb[3] + B[24 + i] * pb[4] + B[30 + i] * pb[5];
            else v = B[6 * i] * pb[0] + B[6 * i + 1] * pb[1] + B[6 * i + 2] * pb[2] + B[6 * i + 3] * pb[3] + B[6 * i + 4] * pb[4] + B[6 * i + 5] * pb[5];
            part[t] = v;
        }
        __syncthreads();
        double qi = 0.0, pi = 0.0;
        if (tid < n) {
            const int a = tid / 6, i = tid - 6 * a;
            for (int e = rowoff[a]; e < rowoff[a + 1]; ++e) qi += part[6 * e + i];
            pi = p[tid];
        }
        const double pq = bsum(pi * qi, 2);
        if (!(pq > 0.0) || !isfinite(pq)) {  // not positive definite / NaN input (uniform)
            fail = 1;
            break;
        }
        const double alpha = rz / pq;
        if (tid < n) {
            x[tid] += alpha * pi;
            ri = r[tid] - alpha * qi;
            r[tid] = ri;
        }
        __syncthreads();
        if (tid < n) {
            zi = precond(tid);
            z[tid] = zi;
        }
        const double rz_new = bsum(tid < n ? ri * zi : 0.0, 3);
        rr = bsum(tid < n ? ri * ri : 0.0, 4);
        const double beta = rz_new / rz;
        rz = rz_new;
        if (tid < n) p[tid] = zi + beta * pi;
        ++it;
        __syncthreads();
    }
    if (tid < n) D.dp[tid] = fail ? 0.0 : x[tid];
    if (tid == 0) {
        BaCtl& c = *D.ctl;
        c.pcg_it = it;
        c.pcg_total_it += it;
        c.pcg_solves += 1;
        c.pcg_done = it + 1;
        if (fail) {
            c.pcg_fail = 1;
            c.solve_failed = 1;
        }
        else if (rr > tol2 * rr0) {  // iteration cap: an inexact step is acceptable when the residual fell by 1e-6
            c.pcg_fail = 2;
            if (!(rr <= 1e-12 * rr0)) c.solve_failed = 1;
        }
    }
}

// back-substitution and trial state: blocks [0, nb_lm) = landmarks (8 lanes each), the rest = poses
// landmark part of the update, lane `sub` of landmark l's group of 8: X = estimate (+) delta (all lanes of the group return it), the trial
// state stored by lane 0; returns the landmark's share of delta^T (lambda delta + b) on lane 0, 0 elsewhere
__device__ __forceinline__ double lm_update_lane(const BaDev& D, int l, int sub, bool in_range, double lambda, double* X) {
    const double* pt_cur = st_pt(D, 0);
    double* pt_trial = const_cast<double*>(st_pt(D, 1));
    const bool lfree = in_range && D.pt_free[l];
    double sc = 0.0;
    double c[3] = {0.0, 0.0, 0.0};
    if (lfree)
        for (int e = D.lm_off[l] + sub; e < D.lm_off[l + 1]; e += LM_LANES) {
            if (D.e_level[e]) continue;
            const int slot = D.pose_slot[D.e_pose[e]];
            if (slot < 0) continue;
            const double* Wd = D.W + (size_t)e * 18;
            const double* xp = D.dp + (size_t)slot * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[0] -= Wd[3 * i] * xp[i];
                c[1] -= Wd[3 * i + 1] * xp[i];
                c[2] -= Wd[3 * i + 2] * xp[i];
            }
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = group_sum8(c[k]);
    X[0] = pt_cur[(size_t)l * 3], X[1] = pt_cur[(size_t)l * 3 + 1], X[2] = pt_cur[(size_t)l * 3 + 2];
    if (lfree) {  // every lane of the group computes the same step (the sums above are identical on all 8 lanes)
        const double* b = D.bl + (size_t)l * 3;
        c[0] += b[0];
        c[1] += b[1];
        c[2] += b[2];
        double I[6];
        if (!lm_dinv(D.Hll + (size_t)l * 6, lambda, I)) D.ctl->solve_failed = 1;  // benign race: every writer stores 1
        const double d0 = I[0] * c[0] + I[1] * c[1] + I[2] * c[2];
        const double d1 = I[1] * c[0] + I[3] * c[1] + I[4] * c[2];
        const double d2 = I[2] * c[0] + I[4] * c[1] + I[5] * c[2];
        X[0] += d0;
        X[1] += d1;
        X[2] += d2;
        if (sub == 0) sc = d0 * (lambda * d0 + b[0]) + d1 * (lambda * d1 + b[1]) + d2 * (lambda * d2 + b[2]);
    }
    if (in_range && sub == 0) {
        pt_trial[(size_t)l * 3] = X[0];
        pt_trial[(size_t)l * 3 + 1] = X[1];
        pt_trial[(size_t)l * 3 + 2] = X[2];
    }
    return sc;
}

// trial state of pose p into O (12 doubles); returns its share of delta^T (lambda delta + b)
__device__ __forceinline__ double pose_update(const BaDev& D, int p, double lambda, double* O) {
    const double* T = st_pose(D, 0) + (size_t)p * 12;
    const int slot = D.pose_slot[p];
    double sc = 0.0;
    if (slot < 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) O[k] = T[k];
    }
    else {
        const double* u = D.dp + (size_t)slot * 6;
        const double* bpv = D.bp_full + (size_t)slot * 6;
        if (D.scale_pose)
#pragma unroll
            for (int k = 0; k < 6; ++k) sc += u[k] * (lambda * u[k] + bpv[k]);
        po_exp_mul(u, T, O);
    }
    return sc;
}

__global__ __launch_bounds__(256) void k_ba_update(BaDev D, int nb_lm) {
    if (D.ctl->phase != 1) return;
    __shared__ double s4[16];
    const double lambda = D.ctl->lambda;
    double sc = 0.0;
    if ((int)blockIdx.x < nb_lm) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        double X[3];
        sc = lm_update_lane(D, min(t / LM_LANES, D.L - 1), t % LM_LANES, t / LM_LANES < D.L, lambda, X);
    }
    else {
        const int p = (blockIdx.x - nb_lm) * 256 + threadIdx.x;
        if (p < D.P) {
            double O[12];
            sc = pose_update(D, p, lambda, O);
            double* Og = const_cast<double*>(st_pose(D, 1)) + (size_t)p * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) Og[k] = O[k];
        }
    }
    const double tsum = block_sum_d(sc, s4);
    if (threadIdx.x == 0) D.red[D.red_scale_off + blockIdx.x] = tsum;
}

// W / Y of edges that left the active set (excluded by the gate, or whose landmark became inactive) are zeroed once
// per stage, so that the (edge, edge) pair lists built for the first stage stay valid: such pairs contribute 0.
__global__ __launch_bounds__(256) void k_ba_zero_inactive(BaDev D) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= D.E) return;
    if (!D.e_level[e] && D.pt_free[D.e_point[e]]) return;
    double2* Wd = reinterpret_cast<double2*>(D.W + (size_t)e * 18);
    const double2 z = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 9; ++k) Wd[k] = z;
}

// The stage boundary of the two-stage schedule without the host (local_bundle_adjuster_g2o.cc:323-344 leaves edge levels behind; what the
// second stage needs of them): workgroups [0, nb) take 256 landmarks each -- a landmark stays a free vertex while it keeps an active edge,
// the excluded edges are counted --, workgroups [nb, nb + P) one pose each -- a free pose that lost its last active edge would renumber the
// reduced system: flagged, and the stage enqueued behind refuses to start (k_ba_begin) so that the host can rebuild the structure.
__global__ __launch_bounds__(256) void k_ba_activity(BaDev D, uint8_t* __restrict__ pt_free, int nb) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if ((int)blockIdx.x < nb) {
        const int l = blockIdx.x * 256 + threadIdx.x;
        int gated = 0;
        if (l < D.L) {
            bool any = false;
            for (int e = D.lm_off[l]; e < D.lm_off[l + 1]; ++e) {
                const int lv = D.e_level[e];
                any = any || !lv;
                gated += lv != 0;
            }
            if (!any) pt_free[l] = 0;
        }
        if (gated) atomicAdd(&s_cnt, gated);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) atomicAdd(&D.ctl->gated, s_cnt);
    }
    else {
        const int p = blockIdx.x - nb;
        if (D.pose_slot[p] < 0) return;
        int any = 0;
        for (int k = D.pe_off[p] + threadIdx.x; k < D.pe_off[p + 1] && !any; k += 256) any = !D.e_level[D.pe_idx[k]];
        if (any) s_cnt = 1;  // (benign race: every writer stores 1)
        __syncthreads();
        if (threadIdx.x == 0 && !s_cnt) D.ctl->structure_changed = 1;
    }
}

// ------------------------------------------------------------------------------------------------ LM control on the device
// fixed-order sum of n doubles by one 256-thread workgroup (strided partial sums, shuffle tree, wave partials in wave order)
// (PEEK: the values were written by OTHER workgroups of the running kernel -- read them past this CU's L1)
template <bool PEEK>
__device__ __forceinline__ double ctl_peek(const double* p) { return PEEK ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <bool PEEK = false>
__device__ __forceinline__ double ctl_sum(const double* __restrict__ v, int n, double* sw) {
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) t += PEEK ? __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : v[i];
    return block_sum_d(t, sw);
}

// this rank's partial sums folded to out4 = {chi2, step scale, solver failure, stop vote}: the payload of the per-trial
// all-reduce of a sharded solve
__global__ __launch_bounds__(256) void k_ba_fold(BaDev D, double* __restrict__ out4, int with_scale) {
    __shared__ double sw[16];
    const double chi = D.E > 0 ? ctl_sum(D.red + D.red_chi_off, D.red_chi_n, sw) : 0.0;
    const double sc = with_scale ? ctl_sum(D.red + D.red_scale_off, D.red_scale_n, sw) : 0.0;
    if (threadIdx.x == 0) {
        out4[0] = chi;
        out4[1] = sc;
        out4[2] = with_scale ? (double)D.ctl->solve_failed : 0.0;
        out4[3] = (D.ctl->stop || *D.stop_mirror) ? 1.0 : 0.0;
    }
}

// SparseOptimizer::optimize(it_max) starts: activeRobustChi2 of the estimate (its partial sums were just written by k_ba_chi2),
// fresh OptimizationAlgorithmLevenberg state; the loop condition `it < iterations && !terminate()` is evaluated for it = 0
__global__ __launch_bounds__(256) void k_ba_begin(BaDev D, int it_max, int stop_in) {
    __shared__ double sw[16];
    double chi = 0.0;
    int stop = stop_in & 1;
    if (D.xsum) {
        chi = D.xsum[0];
        stop |= D.xsum[3] > 0.5;
    }
    else {
        if (D.E > 0) chi = ctl_sum(D.red + D.red_chi_off, D.red_chi_n, sw);
        stop |= *D.stop_mirror != 0;
    }
    if (threadIdx.x == 0) {
        BaCtl& c = *D.ctl;
        if ((stop_in & 2) && c.structure_changed) {  // a stage enqueued behind k_ba_activity: the host has to rebuild the structure first
            c.phase = 2;
            return;
        }
        c.current_chi = c.chi_begin = chi;
        c.it = 0;
        c.it_max = it_max;
        c.qmax = 0;
        c.ni = 2.0;
        c.ok = 1;
        c.stop = stop;
        c.max_diag_bits = 0ull;
        c.solve_failed = 0;
        c.phase = (it_max > 0 && !stop) ? 0 : 2;
    }
}

// after the linearisation of a step: computeLambdaInit on the first iteration, then the trial kernels may run
__global__ void k_ba_prepare(BaDev D) {
    BaCtl& c = *D.ctl;
    if (c.phase == 2) return;
    if (c.phase == 0) {
        if (c.it == 0) {
            // (sharded: the control block holds max(pose diagonals of the summed blocks -- the same on every rank --, this rank's landmarks);
            //  the slots hold every rank's landmark maximum)
            double md = __longlong_as_double((long long)c.max_diag_bits);
            if (D.world > 1)
                for (int r = 0; r < D.world; ++r) md = fmax(md, D.maxslots[r]);
            c.lambda = 1e-5 * md;
            c.ni = 2.0;
        }
        c.qmax = 0;
        c.rho = 0.0;
        c.phase = 1;
    }
    c.solve_failed = 0;
    c.pcg_done = 0;
    c.pcg_fail = 0;
    c.pcg_it = 0;
}

// end of a damping trial (OptimizationAlgorithmLevenberg::solve, the do { } while (rho < 0 && qmax < 10 && !terminate()) body)
// and, when the trial closes the LM iteration, the terminate_action hook (optimize/terminate_action.cc:36-76)
template <bool PEEK = false>
__device__ __forceinline__ void lm_decide(const BaDev& D, double* sw /* 16 doubles */) {
    double temp_chi, scale;
    int failed, stop_now;
    if (D.xsum) {
        temp_chi = D.xsum[0];
        scale = D.xsum[1];
        failed = D.xsum[2] > 0.5;
        stop_now = D.xsum[3] > 0.5;
    }
    else {
        // every load of the decision is issued before the first one is waited for (the host's stop word is a PCIe read: ~2 us on its own)
        const int mirror = *D.stop_mirror;
        failed = PEEK ? __hip_atomic_load(&D.ctl->solve_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : D.ctl->solve_failed;
        double t_chi = 0.0, t_sc = 0.0;
        if (D.E > 0)
            for (int i = threadIdx.x; i < D.red_chi_n; i += blockDim.x) t_chi += ctl_peek<PEEK>(D.red + D.red_chi_off + i);
        for (int i = threadIdx.x; i < D.red_scale_n; i += blockDim.x) t_sc += ctl_peek<PEEK>(D.red + D.red_scale_off + i);
        temp_chi = block_sum_d(t_chi, sw);
        scale = block_sum_d(t_sc, sw);
        stop_now = D.ctl->stop || mirror != 0;
    }
    if (threadIdx.x != 0) return;
    BaCtl& c = *D.ctl;
    ++c.lm_trials;
    if (failed) {
        temp_chi = 1.7976931348623157e308;
        ++c.solve_failures;
    }
    double rho = c.current_chi - temp_chi;
    scale += 1e-3;
    rho /= scale;
    c.temp_chi = temp_chi;
    c.scale = scale;
    bool lambda_bad = false;
    if (rho > 0 && isfinite(temp_chi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        c.lambda *= fmax(1. / 3., alpha);
        c.ni = 2.0;
        c.current_chi = temp_chi;
        c.cur ^= 1;  // accept: the trial state becomes the estimate
    }
    else {
        c.lambda *= c.ni;
        c.ni *= 2.0;
        lambda_bad = !isfinite(c.lambda);
    }
    ++c.qmax;
    c.rho = rho;
    c.stop = stop_now;
    if (!lambda_bad && rho < 0 && c.qmax < 10 && !stop_now) return;  // another trial on the same linearisation (phase stays 1)
    if (c.qmax == 10 || rho == 0 || !isfinite(c.lambda)) c.ok = 0;
    // postIteration: terminate_action on the chi2 of the estimate
    if (c.it == 0) c.last_chi = c.current_chi;
    else {
        const double gain = (c.last_chi - c.current_chi) / c.current_chi;
        c.last_chi = c.current_chi;
        if (gain >= 0 && gain < c.gain_thr) {
            c.stop = 1;
            c.stopped_by_terminate = 1;
        }
    }
    ++c.it;
    c.phase = (c.it < c.it_max && !c.stop && c.ok) ? 0 : 2;
}

__global__ __launch_bounds__(1024) void k_ba_decide(BaDev D) {  // one workgroup of 1024: the ~6 k chi2 partials of config 5 are 6 loads deep, not 25
    if (D.ctl->phase != 1) return;
    __shared__ double sw[16];
    lm_decide(D, sw);
}

// FUSED TRIAL TAIL of a local-BA sized problem (not sharded, P <= TAIL_MAX_POSES): back-substitution, trial state and its robust chi2 in ONE
// launch instead of two (k_ba_update, k_ba_chi2); the decision stays a launch of its own.  At this size a kernel's time is the number of
// DEPENDENT memory round trips on its longest path (~1 us each: the data was written by the previous kernel on other XCDs), so the kernel is
// laid out by hops:
//   hop 1   control block | lm_off, pt_free, Hll, bl of the thread's landmark | pose_slot, intrinsics of pose `tid` | dp
//   hop 2   the landmark's first edge per lane (level, pose, observation, W record) | current landmark | current pose `tid`
//   LDS     trial poses (every workgroup computes all P exponentials itself), intrinsics, slots, dp
//   hop 3   (only landmarks with more than 8 observations: the remaining edges of a lane)
// Stamps (SVGPU_BA_DBG, config 3, 313 workgroups): all workgroups entered by 0.4 us, hop 1 back by 1.1 .. 2.8, table staged by 3.7, trial poses
// by 5.4, sums written by 7.6 us.
// The decision is NOT folded in (tried three ways): handing over to a last / deciding workgroup inside the launch needs a device-scope
// release per workgroup, and on this part that is an L2 write-back towards the other XCDs -- a returning ticket on the control block cost
// 5 us for 313 workgroups (one or two levels alike), add-only arrivals watched by workgroup 0 cost 3 .. 12 us each -- more than the
// launch boundary it replaces, which does that write-back once.  Same per-landmark and per-workgroup sums, in the same order, as the two kernels.
#define TAIL_MAX_POSES 256
template <bool EQ>
__global__ __launch_bounds__(256) void k_ba_tail(BaDev D) {
    __shared__ double s4[16];
    __shared__ double s_pose[TAIL_MAX_POSES * 12];
    __shared__ double s_intr[TAIL_MAX_POSES * 5];
    __shared__ double s_dp[TAIL_MAX_POSES * 6];
    __shared__ int s_slot[TAIL_MAX_POSES];
    const int tid = threadIdx.x;
#define TAIL_T(i) do { if (D.dbg && !D.dbg_schur_on && tid == 0) D.dbg[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
    TAIL_T(0);
    // ---- hop 1 (nothing here depends on a loaded value)
    const int phase = D.ctl->phase, cur = D.ctl->cur;
    const double lambda = D.ctl->lambda;
    const int t = blockIdx.x * 256 + tid;
    const int l = min(t / LM_LANES, D.L - 1), sub = t % LM_LANES;
    const bool in_range = t / LM_LANES < D.L;
    const int lo = D.lm_off[l], hi = D.lm_off[l + 1];
    const bool lfree = in_range && D.pt_free[l];
    double Hl[6], b[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Hl[k] = D.Hll[(size_t)l * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = D.bl[(size_t)l * 3 + k];
    const bool has_pose = tid < D.P;
    const int pq = has_pose ? tid : 0;
    const int my_slot = D.pose_slot[pq];
    double my_intr[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) my_intr[k] = D.intr[(size_t)pq * 5 + k];
    const double my_dp = tid < D.n ? D.dp[tid] : 0.0;  // n = 6 nP <= 6 P; a workgroup of 256 stages up to 256 entries per pass
    if (phase != 1) return;
    TAIL_T(1);
    // ---- hop 2
    const double* pose_cur = D.pose_buf[cur & 1];
    const double* pt_cur = D.pt_buf[cur & 1];
    double* pt_trial = D.pt_buf[(cur & 1) ^ 1];
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = pose_cur[(size_t)pq * 12 + k];
    double X[3] = {pt_cur[(size_t)l * 3], pt_cur[(size_t)l * 3 + 1], pt_cur[(size_t)l * 3 + 2]};
    const int e0 = lo + sub;
    const bool has0 = in_range && e0 < hi;
    const int ee = has0 ? e0 : 0;
    const int lvl0 = D.e_level[ee], p0 = D.e_pose[ee], rob0 = D.e_robust[ee];
    const float u0 = D.e_uvr[(size_t)ee * 3], v0 = D.e_uvr[(size_t)ee * 3 + 1], r0 = D.e_uvr[(size_t)ee * 3 + 2], w0 = D.e_w[ee], hub0 = D.e_huber[ee];
    double W0[18];
    {
        const double2* Wd = reinterpret_cast<const double2*>(D.W + (size_t)ee * 18);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double2 q = Wd[k];
            W0[2 * k] = q.x;
            W0[2 * k + 1] = q.y;
        }
    }
    // ---- stage the pose table
    for (int i = tid; i < D.n; i += 256) s_dp[i] = i == tid ? my_dp : D.dp[i];
    if (has_pose) {
        s_slot[tid] = my_slot;
#pragma unroll
        for (int k = 0; k < 5; ++k) s_intr[tid * 5 + k] = my_intr[k];
    }
    __syncthreads();
    TAIL_T(2);
    double scp = 0.0;
    if (has_pose) {  // pose_update(), from the staged values
        double O[12];
        if (my_slot < 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) O[k] = T[k];
        }
        else {
            double u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) u[k] = s_dp[my_slot * 6 + k];
            if (D.scale_pose) {
                const double* bpv = D.bp_full + (size_t)my_slot * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) scp += u[k] * (lambda * u[k] + bpv[k]);
            }
            po_exp_mul(u, T, O);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) s_pose[tid * 12 + k] = O[k];
        if (blockIdx.x == 0) {
            double* Og = D.pose_buf[(cur & 1) ^ 1] + (size_t)tid * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) Og[k] = O[k];
        }
    }
    if (blockIdx.x == 0) {  // the pose share of delta^T (lambda delta + b): the slot k_ba_update's pose workgroup writes
        const double ts = block_sum_d(scp, s4);
        if (tid == 0) D.red[D.red_scale_off + gridDim.x] = ts;
    }
    __syncthreads();
    TAIL_T(3);
    // ---- back-substitution of the landmark (lm_update_lane)
    double sc = 0.0;
    double c[3] = {0.0, 0.0, 0.0};
    if (lfree) {
        if (has0 && !lvl0 && s_slot[p0] >= 0) {
            const double* xp = s_dp + s_slot[p0] * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[0] -= W0[3 * i] * xp[i];
                c[1] -= W0[3 * i + 1] * xp[i];
                c[2] -= W0[3 * i + 2] * xp[i];
            }
        }
        for (int e = e0 + LM_LANES; e < hi; e += LM_LANES) {
            if (D.e_level[e]) continue;
            const int slot = s_slot[D.e_pose[e]];
            if (slot < 0) continue;
            const double* Wd = D.W + (size_t)e * 18;
            const double* xp = s_dp + slot * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[0] -= Wd[3 * i] * xp[i];
                c[1] -= Wd[3 * i + 1] * xp[i];
                c[2] -= Wd[3 * i + 2] * xp[i];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = group_sum8(c[k]);
    if (lfree) {
        c[0] += b[0];
        c[1] += b[1];
        c[2] += b[2];
        double I[6];
        if (!lm_dinv(Hl, lambda, I)) D.ctl->solve_failed = 1;  // benign race: every writer stores 1
        const double d0 = I[0] * c[0] + I[1] * c[1] + I[2] * c[2];
        const double d1 = I[1] * c[0] + I[3] * c[1] + I[4] * c[2];
        const double d2 = I[2] * c[0] + I[4] * c[1] + I[5] * c[2];
        X[0] += d0;
        X[1] += d1;
        X[2] += d2;
        if (sub == 0) sc = d0 * (lambda * d0 + b[0]) + d1 * (lambda * d1 + b[1]) + d2 * (lambda * d2 + b[2]);
    }
    if (in_range && sub == 0) {
        pt_trial[(size_t)l * 3] = X[0];
        pt_trial[(size_t)l * 3 + 1] = X[1];
        pt_trial[(size_t)l * 3 + 2] = X[2];
    }
    // ---- robust chi2 of the landmark's observations at the trial state (lm_chi2_lane)
    double v = 0.0;
    if (has0 && !lvl0) {
        const float uvr[3] = {u0, v0, r0};
        double r[3];
        const double chi = edge_error<EQ>(s_pose + p0 * 12, X, s_intr + p0 * 5, uvr, (double)w0, r, nullptr);
        if (rob0) {
            double rho0, rho1;
            huber(chi, (double)hub0, &rho0, &rho1);
            v += rho0;
        }
        else v += chi;
    }
    if (in_range)
        for (int e = e0 + LM_LANES; e < hi; e += LM_LANES) {
            if (D.e_level[e]) continue;
            const int p = D.e_pose[e];
            double r[3];
            const double chi = edge_error<EQ>(s_pose + p * 12, X, s_intr + p * 5, D.e_uvr + (size_t)e * 3, (double)D.e_w[e], r, nullptr);
            if (D.e_robust[e]) {
                double rho0, rho1;
                huber(chi, (double)D.e_huber[e], &rho0, &rho1);
                v += rho0;
            }
            else v += chi;
        }
    const double tsc = block_sum_d(sc, s4);
    const double tchi = block_sum_d(v, s4);
    TAIL_T(4);
    if (tid == 0) {
        D.red[D.red_scale_off + blockIdx.x] = tsc;
        D.red[D.red_chi_off + blockIdx.x] = tchi;
    }
}

__global__ __launch_bounds__(256) void k_ba_expand_dense(BaDev D) {
    if (D.ctl->phase != 1) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const size_t n = D.n;
    if (k < D.NB * 36) {
        const int blk = k / 36, t = k - 36 * blk, i = t / 6, j = t - 6 * i;
        const int2 ab = D.blk_ab[blk];
        // (a diagonal block is symmetric up to the last bits of its two triangles' sums: written by BOTH of its entries (i, j) and (j, i),
        //  the dense matrix would hold whichever landed last -- run-to-run differences of 1e-13 in the estimate.  Its lower triangle decides.)
        if (ab.x != ab.y || j <= i) {
            D.S[(size_t)(6 * ab.x + i) * n + 6 * ab.y + j] = D.Sblk[k];
            D.S[(size_t)(6 * ab.y + j) * n + 6 * ab.x + i] = D.Sblk[k];
        }
    }
    if (k < D.n) D.S[n * n + k] = D.g[k];
}

}  // namespace

void sv_pose_opt(svgpu_ctx* ctx, hipStream_t s, const PoseOptDev& P) {
    SvProfScope ps(ctx, s, "k_pose_opt");
    const size_t lds = sizeof(double) * (PO_THREADS / 64) * WRED_DOUBLES;
    const bool eq = P.intr[0] == 0.0 && P.intr[1] == 0.0;  // cam_is_equirect
    const bool trk = P.trk_map != nullptr;
    const void* k = trk ? (eq ? (const void*)k_pose_opt<true, true> : (const void*)k_pose_opt<false, true>)
                        : (eq ? (const void*)k_pose_opt<true, false> : (const void*)k_pose_opt<false, false>);
    (void)sv_allow_dynamic_lds(k, lds);  // dynamic LDS above 64 KB must be allowed explicitly
    if (trk) {
        if (eq) hipLaunchKernelGGL((k_pose_opt<true, true>), dim3(1), dim3(PO_THREADS), lds, s, P);
        else hipLaunchKernelGGL((k_pose_opt<false, true>), dim3(1), dim3(PO_THREADS), lds, s, P);
    }
    else if (eq) hipLaunchKernelGGL((k_pose_opt<true, false>), dim3(1), dim3(PO_THREADS), lds, s, P);
    else hipLaunchKernelGGL((k_pose_opt<false, false>), dim3(1), dim3(PO_THREADS), lds, s, P);
}

void sv_ba_zero_inactive(hipStream_t s, const BaDev& D) {
    if (D.E > 0) hipLaunchKernelGGL(k_ba_zero_inactive, dim3((D.E + 255) / 256), dim3(256), 0, s, D);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int nb_lm_blocks(const BaDev& D) { return (D.L * LM_LANES + 255) / 256; }
int sv_ba_lm_blocks(int L) { return (L * LM_LANES + 255) / 256; }

// linearisation (+ pose blocks from their partials; unless the blocks must first be summed over ranks: lambda init and trial start)
void sv_ba_linearize(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int do_prepare) {
    SvProfScope ps(ctx, s, "ba_linearize");
    const int nb = nb_lm_blocks(D);
    static const bool two_launches = std::getenv("SVGPU_BA_LIN_TWO_LAUNCHES") != nullptr;  // profiling aid: times the two sides apart
    const int np = D.nP * D.lin_split;
    auto launch = [&](int blocks, int blk0) {
        if (blocks <= 0) return;
        if (D.any_equirect) hipLaunchKernelGGL(k_ba_lin<true>, dim3(blocks), dim3(256), 0, s, D, nb, blk0);
        else hipLaunchKernelGGL(k_ba_lin<false>, dim3(blocks), dim3(256), 0, s, D, nb, blk0);
    };
    if (two_launches) {
        launch(nb, 0);
        launch(np, nb);
    }
    else launch(nb + np, 0);
    hipLaunchKernelGGL(k_ba_lin_fin, dim3(1), dim3(LIN_FIN_THREADS), 0, s, D, do_prepare, nb);
}

// sharded solve only.  Before the pose blocks are summed over the ranks: the rank's slot = the largest diagonal of ITS landmarks (k_ba_lin left it in
// the control block); the slots ride behind the blocks in the same all-reduce.  After it: the pose part of the maximum from the summed blocks.
void sv_ba_maxslot(hipStream_t s, const BaDev& D) {
    if (D.world > 1) hipLaunchKernelGGL(k_ba_maxslot, dim3(1), dim3(64 > D.world ? 64 : D.world), 0, s, D);
}
void sv_ba_maxdiag(hipStream_t s, const BaDev& D) {
    if (D.nP > 0) hipLaunchKernelGGL(k_ba_maxdiag, dim3((D.nP + 255) / 256), dim3(256), 0, s, D);
}

void sv_ba_fold(hipStream_t s, const BaDev& D, double* out4, int with_scale) { hipLaunchKernelGGL(k_ba_fold, dim3(1), dim3(256), 0, s, D, out4, with_scale); }
void sv_ba_owned_mark(hipStream_t s, const int* lm_off, int L, double* xch, double stop_vote) {
    hipLaunchKernelGGL(k_ba_owned_mark, dim3((L + 1 + 255) / 256), dim3(256), 0, s, lm_off, L, xch, stop_vote);
}
void sv_ba_owned_check(hipStream_t s, const double* xch, int L, uint8_t* any_owner, double* verdict) {
    hipLaunchKernelGGL(k_ba_owned_check, dim3((L + 1 + 255) / 256), dim3(256), 0, s, xch, L, any_owner, verdict);
}
void sv_ba_points_share(hipStream_t s, const BaDev& D, double* points_out, double* xch, int dir) {
    if (D.L > 0) hipLaunchKernelGGL(k_ba_points_share, dim3((unsigned)((3 * (size_t)D.L + 255) / 256)), dim3(256), 0, s, D, points_out, xch, dir);
}
void sv_ba_xs_move(hipStream_t s, const BaDev& D, const int* blk_idx, int nb, const int* slot_idx, int ns, double* buf, int dir) {
    const int items = nb * 36 + ns * 6;
    if (items > 0) hipLaunchKernelGGL(k_ba_xs_move, dim3((items + 255) / 256), dim3(256), 0, s, D, blk_idx, nb, slot_idx, ns, buf, dir);
}
void sv_ba_activity(hipStream_t s, const BaDev& D, uint8_t* pt_free) {
    const int nb = (D.L + 255) / 256;
    hipLaunchKernelGGL(k_ba_activity, dim3(nb + D.P), dim3(256), 0, s, D, pt_free, nb);
}
void sv_ba_begin(hipStream_t s, const BaDev& D, int it_max, int stop_in) { hipLaunchKernelGGL(k_ba_begin, dim3(1), dim3(256), 0, s, D, it_max, stop_in); }
void sv_ba_prepare(hipStream_t s, const BaDev& D) { hipLaunchKernelGGL(k_ba_prepare, dim3(1), dim3(1), 0, s, D); }
void sv_ba_decide(hipStream_t s, const BaDev& D) { hipLaunchKernelGGL(k_ba_decide, dim3(1), dim3(1024), 0, s, D); }
bool sv_ba_tail_ok(const BaDev& D) { return D.world <= 1 && D.xsum == nullptr && D.P > 0 && D.P <= TAIL_MAX_POSES && D.L > 0; }
void sv_ba_tail(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {  // update + chi2 of the trial state + decide (sv_ba_tail_ok)
    SvProfScope ps(ctx, s, "ba_tail");
    if (D.any_equirect) hipLaunchKernelGGL(k_ba_tail<true>, dim3(nb_lm_blocks(D)), dim3(256), 0, s, D);
    else hipLaunchKernelGGL(k_ba_tail<false>, dim3(nb_lm_blocks(D)), dim3(256), 0, s, D);
}

int sv_ba_lin_split_max() { return LIN_SPLIT_MAX; }
int sv_ba_lin_split(int E, int nP) {  // workgroups per free pose on the pose side of k_ba_lin
    if (nP <= 0) return 1;
    const long long per_pose = ((long long)E + nP - 1) / nP;
    int k = (int)((per_pose + LIN_EDGES_PER_BLOCK - 1) / LIN_EDGES_PER_BLOCK);
    k = k < 1 ? 1 : (k > LIN_SPLIT_MAX ? LIN_SPLIT_MAX : k);
    // a small system (local BA: a few dozen free poses) leaves most of the chip idle at ~3 edges per lane: one edge per lane there
    while (k < LIN_SPLIT_MAX && nP * (k + 1) <= 512 && per_pose / (k + 1) >= 192) ++k;
    return k;
}
int sv_ba_rhs_split() { return RHS_SPLIT; }

// reduced camera system: shares of the blocks and of the right-hand side, then the kept blocks Sblk and g
void sv_ba_reduce(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    SvProfScope ps(ctx, s, "ba_schur");
    if (D.nP <= 0) return;
    static_assert(SCH_WAVES == 1, "the XCD-contiguous order permutes workgroups = units");
    if (D.unit_rec) {
        hipLaunchKernelGGL(k_ba_schur_rhs, dim3((D.num_units + 7) & ~7), dim3(64 * SCH_WAVES), 0, s, D, D.nshare, D.rhs_part, 1);
        hipLaunchKernelGGL(k_ba_sys_fin, dim3((D.NB * 36 + D.n + 255) / 256), dim3(256), 0, s, D, D.nshare, D.rhs_part, D.nshare);
        return;
    }
    if (D.nshare > RHS_SPLIT) return;  // (svgpu_ba.hip caps nshare at 16)
    const int n_schur = D.NB * D.nshare;
    // A large system (more than one resident round of units): XCD-contiguous unit order, right-hand side summed by the shares of the diagonal
    // blocks.  A small one (local BA) leaves the chip part-filled: the right-hand side runs beside the blocks as units of its own, which also
    // leaves the observation arrays warm for the next linearisation (folded there: k_ba_schur_rhs 13.3 -> 13.6 us, k_ba_lin 11.8 -> 13.0 us).
    int large = n_schur >= 8192;
    const char* const order_env = std::getenv("SVGPU_BA_SCHUR_ORDER");  // tests / experiments: 0 = small-system form, 1 = large-system form
    if (order_env) large = std::atoi(order_env) != 0;
    const int grid = large ? ((n_schur + 7) & ~7) : n_schur + D.nP * RHS_SPLIT;
    hipLaunchKernelGGL(k_ba_schur_rhs, dim3(grid), dim3(64 * SCH_WAVES), 0, s, D, D.nshare, D.rhs_part, large);
    hipLaunchKernelGGL(k_ba_sys_fin, dim3((D.NB * 36 + D.n + 255) / 256), dim3(256), 0, s, D, D.nshare, D.rhs_part, large ? D.nshare : RHS_SPLIT);
}

// LDS-resident PCG: bytes of dynamic LDS for this system, 0 = does not fit (n > 512 or more than ~150 KB)
size_t sv_ba_pcg_lds_bytes(const BaDev& D) {
    const size_t nent = 2 * (size_t)D.NB - D.nP;
    const size_t bytes = 8 * (36 * (size_t)D.NB + 36 * (size_t)D.nP + 5 * (size_t)D.n + 6 * nent + 48) + 4 * (size_t)((D.nP + 2) & ~1) + 8 * nent;
    return (D.n <= PL_THREADS && bytes <= 150 * 1024) ? bytes : 0;
}
void sv_ba_solve_pcg_lds(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    if (D.nP <= 0) return;
    SvProfScope ps(ctx, s, "ba_solve");
    const size_t lds = sv_ba_pcg_lds_bytes(D);
    (void)sv_allow_dynamic_lds((const void*)k_ba_pcg_lds, lds);
    hipLaunchKernelGGL(k_ba_pcg_lds, dim3(1), dim3(PL_THREADS), lds, s, D, 2 * D.NB - D.nP);
}

size_t sv_ba_chol_bytes(int n) { return sizeof(double) * (size_t)(n + 3) * (n | 1); }
// on-chip dense LL^T: the register-tile LDL^T while the matrix fits LDS (n <= 186, two tiles per thread); on request (solver
// CHOLESKY_MFMA, or SVGPU_BA_CHOL=mfma for A/B runs) the blocked MFMA factorisation up to CM_MAX_N unknowns.
void sv_ba_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    if (D.nP <= 0) return;
    SvProfScope ps(ctx, s, "ba_solve");
    static const bool force_mfma = std::getenv("SVGPU_BA_CHOL") && !strcmp(std::getenv("SVGPU_BA_CHOL"), "mfma");
    if (D.n <= CM_MAX_N && (D.chol_mfma || force_mfma)) {
        const size_t lds = cm_lds_bytes(D.n);
        (void)sv_allow_dynamic_lds((const void*)k_ba_chol_mfma, lds);
        hipLaunchKernelGGL(k_ba_chol_mfma, dim3(1), dim3(CM_THREADS), lds, s, D);
        return;
    }
    const size_t lds = sv_ba_chol_bytes(D.n);
    const int R = D.n / 3 + 1, NT = R * (R + 1) / 2;
    const int threads = std::min(CHOL_MAX_THREADS, (NT + 63) & ~63);
    if (NT <= threads) {
        (void)sv_allow_dynamic_lds((const void*)k_ba_chol_tile<1>, lds);  // dynamic LDS above 64 KB must be allowed explicitly
        hipLaunchKernelGGL(k_ba_chol_tile<1>, dim3(1), dim3(threads), lds, s, D);
    }
    else {
        (void)sv_allow_dynamic_lds((const void*)k_ba_chol_tile<2>, lds);
        hipLaunchKernelGGL(k_ba_chol_tile<2>, dim3(1), dim3(threads), lds, s, D);
    }
}

// dense image in global memory + the one-workgroup LL^T on it (solver = dense: systems beyond the on-chip solver's 186 unknowns that are
// to be factored densely all the same -- the default for those sizes is the block envelope Cholesky).  The library's own code throughout:
// no vendor solver is loaded anywhere.
void sv_ba_dense_tiled(hipStream_t s, const BaDev& D);
void sv_ba_solve_dense(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    if (D.nP <= 0) return;
    SvProfScope ps(ctx, s, "ba_solve");
    (void)hipMemsetAsync(D.S, 0, sizeof(double) * (size_t)(D.n + 1) * D.n, s);
    hipLaunchKernelGGL(k_ba_expand_dense, dim3((D.NB * 36 + 255) / 256), dim3(256), 0, s, D);
    static const bool one_wg = std::getenv("SVGPU_BA_DENSE_ONE_WG") != nullptr;  // A/B aid: the one-workgroup column-by-column form
    if (one_wg || D.n > 1536) hipLaunchKernelGGL(k_ba_chol_global, dim3(1), dim3(1024), 0, s, D);  // (the tiled form keeps the right-hand side in LDS: n <= 1536)
    else sv_ba_dense_tiled(s, D);  // ba_dense_tiled.hip: tiles of 48, panel + MFMA update per tile column
}

// back-substitution, trial state
void sv_ba_update(svgpu_ctx* ctx, hipStream_t s, const BaDev& D) {
    SvProfScope ps(ctx, s, "ba_update");
    const int nb = nb_lm_blocks(D);
    hipLaunchKernelGGL(k_ba_update, dim3(nb + (D.P + 255) / 256), dim3(256), 0, s, D, nb);
}

void sv_ba_chi2(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int use_trial, int store_cache, int guarded) {
    SvProfScope ps(ctx, s, "ba_chi2");
    if (D.E > 0) {
        if (D.any_equirect) hipLaunchKernelGGL(k_ba_chi2<true>, dim3(nb_lm_blocks(D)), dim3(256), 0, s, D, use_trial, store_cache, guarded);
        else hipLaunchKernelGGL(k_ba_chi2<false>, dim3(nb_lm_blocks(D)), dim3(256), 0, s, D, use_trial, store_cache, guarded);
    }
}

// the current estimate (the control block says which of the two state buffers holds it) -> one contiguous output block
__global__ __launch_bounds__(256) void k_ba_pack_out(BaDev D, double* __restrict__ out) {
    const int cur = D.ctl->cur & 1;
    const size_t np = 12 * (size_t)D.P, n = np + 3 * (size_t)D.L;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (i < np) out[i] = D.pose_buf[cur][i];
        else if (!D.lm_order) out[i] = D.pt_buf[cur][i - np];
        else {
            const size_t j = i - np, r = j / 3;
            out[np + 3 * (size_t)D.lm_order[r] + (j - 3 * r)] = D.pt_buf[cur][j];
        }
    }
}
void sv_ba_pack_out(hipStream_t s, const BaDev& D, double* out) {
    const size_t n = 12 * (size_t)D.P + 3 * (size_t)D.L;
    if (n) hipLaunchKernelGGL(k_ba_pack_out, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, D, out);
}
void sv_ba_gate(hipStream_t s, const BaDev& D, int set_levels, uint8_t* outlier_out) {
    if (D.E > 0) hipLaunchKernelGGL(k_ba_gate, dim3((D.E + 255) / 256), dim3(256), 0, s, D, set_levels, outlier_out);
}
