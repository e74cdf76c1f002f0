// Kernel-argument struct and launchers of the local-BA kernels (internal).
#pragma once
#include <cstdint>

struct BaDev {
    int P, L, E;   // poses (free + fixed), landmarks, observation edges (sorted by landmark)
    int nP;        // free, active poses = rows/6 of the reduced system
    int n;         // 6 * nP
    // state: [R|t] rows (12 doubles per pose), 3 doubles per landmark; cur = linearisation point, trial = cur (+) delta
    double* pose_cur;
    double* pose_trial;
    double* pt_cur;
    double* pt_trial;
    // observations
    const int* e_pose;
    const int* e_point;
    const float* e_uvr;    // u, v, u_right (<0: monocular)
    const float* e_w;      // inv_sigma_sq
    const float* e_huber;  // Huber delta
    uint8_t* e_level;      // 0 active, 1 excluded (set_as_outlier)
    uint8_t* e_robust;     // kernel present
    double* e_chi;         // cached chi2 = e^T Omega e of the last error computation over the active set
    const double* intr;    // P x 5
    // structure (rebuilt per stage on the host)
    const int* pose_slot;  // P: index among free active poses or -1
    const uint8_t* pt_free;  // L: 1 = active, non-fixed landmark vertex
    const int* lm_off;     // L + 1 (edges are sorted by landmark)
    const int* pe_off;     // nP + 1: pose -> its active edges
    const int* pe_idx;
    int NB;                // number of non-empty upper blocks (a <= b) of the reduced system
    const int* blk_off;    // NB + 1
    const int2* blk_pairs; // (edge whose pose is a, edge whose pose is b) sharing a landmark
    const int2* blk_ab;    // NB
    // linear system
    double* W;     // E x 18: Hpl block (6x3, row-major) of every edge with free pose and free landmark, else 0
    double* Y;     // E x 18: W * Dinv
    double* lp_part;     // P x LP_SPLIT x 27: partial pose blocks of k_ba_lin_pose
    double* sc_part;     // NB x SCHUR_SPLIT x 36: partial blocks of k_ba_schur
    double* GE;    // E x 6: Y * bl of the edge, written by k_ba_dinv
    double* Hll;   // L x 6 (xx xy xz yy yz zz)
    double* bl;    // L x 3
    double* Dinv;  // L x 6
    double* Hpp;   // nP x 36
    double* bp;    // nP x 6
    double* S;     // (n + 1) x n: reduced system, row n = right-hand side (then L^-1 g)
    double* dp;    // n
    double* dl;    // L x 3
    double* red;   // reduction scratch / read-back: see offsets below
    int red_chi_off, red_chi_n;      // per-block partial sums of the robust chi2
    int red_scale_off, red_scale_n;  // per-block partial sums of delta^T (lambda delta + b)
    int red_flag_off;                // [0] cholesky failure flag, [1] max diagonal
    double lambda;
    double lambda_diag;      // damping added to the diagonal of S by THIS rank (lambda, or 0 on ranks > 0 of a sharded solve)
    const double* Hpp_full;  // pose blocks summed over all ranks (== Hpp when not sharded): lambda init
    const double* bp_full;   // same for bp: step-scale term
    int scale_pose;          // 1 = this rank contributes the pose part of delta^T(lambda delta + b)
    int chol_in_lds;
};

// motion-only BA (pose_optimizer): everything lives in ONE persistent single-workgroup kernel
struct PoseOptDev {
    int n;
    const double* pos_w;      // n x 3 (fixed landmarks)
    const float* uvr;         // n x 3
    const float* inv_sigma_sq;
    const float* huber;
    double intr[5];
    double pose_in[12];
    int num_trials_robust, num_trials, num_each_iter, reset_flag_each_round;
    double gain_thr;
    double* pose_out;         // 12
    uint8_t* outlier;         // n
    int* result;              // [0] num_valid, [1] LM iterations run, [2] num_bad
    uint8_t* level;           // n (scratch)
    uint8_t* robust;          // n (scratch)
};
struct svgpu_ctx;
void sv_pose_opt(svgpu_ctx* ctx, hipStream_t s, const PoseOptDev& P);
void sv_ba_linearize(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);
void sv_ba_reduce(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);  // Dinv/Y, Schur complement, right-hand side
void sv_ba_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);   // reduced solve, back-substitution, trial state
void sv_ba_chi2(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int use_trial, int store_cache);
void sv_ba_gate(hipStream_t s, const BaDev& D, int set_levels, uint8_t* outlier_out);
