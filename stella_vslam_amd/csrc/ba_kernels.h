// Kernel-argument struct and launchers of the local-BA kernels (internal).
#pragma once
#include <cstdint>

// Levenberg-Marquardt control block, resident in device memory.  The whole damping loop of
// OptimizationAlgorithmLevenberg::solve and the terminate_action hook run on the device (k_ba_begin / k_ba_prepare /
// k_ba_decide): the host enqueues "steps" (one damping trial each, preceded by a linearisation when the previous trial
// closed an LM iteration) and reads this block back once per batch of steps instead of twice per trial.
//   phase 0  the next step starts an LM iteration: linearise, (iteration 0: lambda init), then run a trial
//   phase 1  a linearisation is in place; run (another) damping trial
//   phase 2  optimize() has finished: every kernel of the remaining steps returns at once
struct BaCtl {
    double lambda, ni, current_chi, last_chi, temp_chi, scale, rho, chi_begin;
    double gain_thr, pcg_rr0, pcg_tol2, pad0;
    unsigned long long max_diag_bits;  // bit pattern of the largest |diagonal| (non-negative doubles order like integers)
    int cur;                           // which of the two state buffers holds the estimate (the other receives the trial)
    int it, it_max, qmax, phase, ok;
    int stop;                          // the optimizer's force-stop flag as the device sees it
    int stopped_by_terminate;
    int lm_trials, solve_failures, solve_failed;
    int pcg_done, pcg_fail, pcg_it, pcg_max_it, pcg_total_it, pcg_solves;
    int gated;              // edges the gate excluded (k_ba_activity)
    int structure_changed;  // k_ba_activity: a free pose lost its last active edge -- the pose numbering of the reduced system has to be rebuilt on the host
    int pad1;
};

struct BaDev {
    int P, L, E;   // poses (free + fixed), landmarks, observation edges (sorted by landmark)
    int nP;        // free, active poses = rows/6 of the reduced system
    int n;         // 6 * nP
    int dbg_schur_on;  // SVGPU_BA_DBG=schur: dbg holds 8 stamps per unit of the last k_ba_schur_rhs launch instead (fills the padding: the struct is a kernel argument)
    BaCtl* ctl;
    unsigned long long* dbg;  // SVGPU_BA_DBG: 8 wall_clock64 stamps per workgroup of the last k_ba_tail launch (null otherwise)
    const volatile int* stop_mirror;  // page-locked host word the caller's force_stop_flag is mirrored into while the host waits
    const double* xsum;               // sharded solve: {chi2, step scale, solver failures, stop votes} summed over the ranks; else null
    // state: [R|t] rows (12 doubles per pose), 3 doubles per landmark; buffer ctl->cur = linearisation point, the other = cur (+) delta
    double* pose_buf[2];
    double* pt_buf[2];
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
    const int* pe_off;     // P + 1: pose (index, not slot) -> its edges in increasing edge order, whatever their level
    const int* pe_idx;
    const int* slot_pose;  // nP: free-pose slot -> pose index
    // pose-major copies of the observations (position q of the pose -> edge lists): what the pose side of the linearisation reads
    const int* pm_point;   // E (nullable: gather by pe_idx instead)
    const float* pm_uvr;   // E x 3
    const float* pm_w;     // E
    const float* pm_hub;   // E
    int NB;                // number of non-empty upper blocks (a <= b) of the reduced system
    const int* blk_off;    // NB + 1
    const int2* blk_pairs; // (edge whose pose is a, edge whose pose is b) sharing a landmark
    const int2* blk_ab;    // NB
    const int* blk_pair_l; // landmark of every pair (= e_point[pair.x]), in blk_pairs order
    // linear system
    double* W;     // E x 18: Hpl block (6x3, row-major) of every edge with free pose and free landmark, else 0
    double* lp_part;     // nP x lin_split x 27: partial pose blocks of k_ba_lin
    double* lm_max;      // one per landmark-side workgroup of k_ba_lin: its max |diagonal of Hll| (first iteration of an optimize() call)
    int lin_split;       // workgroups per free pose on the pose side of k_ba_lin (sv_ba_lin_split)
    int any_equirect;    // 1 = some pose carries the equirectangular model (fx == fy == 0): kernels with the atan2 / asin path
    double* sc_part;     // NB x nshare x 36: partial blocks of k_ba_schur_rhs
    double* rhs_part;    // nP x RHS_SPLIT x 6: partial sums of W Hll^-1 bl
    int nshare;          // shares per block of the reduced system (pairs per share ~200)
    // chunk-major units (global-BA sizes, ba_pairs.hip sv_ba_build_units): a unit = the pairs of ONE block whose landmarks lie in ONE chunk of
    // consecutive landmark ranks, executed chunk by chunk; null = the arithmetic shares above
    const int4* unit_rec;      // execution position -> {first pair, end pair, block, unit id (block-major)}
    const int* blk_unit_off;   // NB + 1: units of block b = [blk_unit_off[b], blk_unit_off[b + 1]) -- the rows of sc_part / rhs_unit k_ba_sys_fin sums
    double* rhs_unit;          // 6 per unit: W Hll^-1 bl summed over the pairs (i, i) of a diagonal block's unit
    int num_units;
    double* Hll;   // L x 6 (xx xy xz yy yz zz)
    double* bl;    // L x 3
    double* Hpp;   // nP x 36
    double* bp;    // nP x 6
    double* Sblk;  // NB x 36: the kept upper blocks (a <= b) of the reduced camera system, row-major 6x6 each, in blk_ab order
    double* g;     // n: right-hand side of the reduced system (directly behind Sblk: one all-reduce covers both)
    double* S;     // dense (n + 1) x n image in global memory (solver = dense: k_ba_chol_global)
    double* dp;    // n
    double* dl;    // L x 3
    double* red;   // reduction scratch / read-back: see offsets below
    int red_chi_off, red_chi_n;      // per-block partial sums of the robust chi2
    int red_scale_off, red_scale_n;  // per-block partial sums of delta^T (lambda delta + b)
    int red_flag_off;                // [2..5] cycle counters of the on-chip Cholesky
    int add_lambda;          // 1 = THIS rank adds the damping to the diagonal of S (rank 0 only in a sharded solve)
    const uint8_t* any_owner; // sharded: per landmark, some rank holds observations of it (the final exchange of the points)
    const int* lm_off_caller; // renumbered solve: the landmark offsets in the CALLER's numbering (which landmarks this rank holds observations of: k_ba_points_share); else null
    const int* lm_order;      // renumbered solve (ba_pairs.hip): caller's index of the landmark at rank r (k_ba_pack_out writes the positions back in the caller's order); else null
    const uint8_t* lam_slot; // nullable; keyframe-segment exchange of a sharded solve: per free pose, whether THIS rank adds the damping to its diagonal block
    const double* Hpp_full;  // pose blocks summed over all ranks (== Hpp when not sharded): lambda init
    const double* bp_full;   // same for bp: step-scale term
    int scale_pose;          // 1 = this rank contributes the pose part of delta^T(lambda delta + b)
    int chol_in_lds;
    int chol_mfma;           // 1 = the on-chip dense solve takes the blocked MFMA factorisation (k_ba_chol_mfma) where it applies
    int world, rank;         // sharded solve (world > 1): lambda init takes the max over the ranks' one-hot slots
    double* maxslots;        // world doubles (sharded)
    // block-row view of the kept blocks for the PCG solver: row a lists (block index | transposed << 30, column b)
    const int* prow_off;     // nP + 1
    const int2* prow_ent;
    const int* diag_blk;     // nP: index of block (a, a)
    double* pcg_Minv;        // nP x 36: inverse diagonal blocks (block-Jacobi preconditioner)
    double* pcg_rws;         // 2 sets x 3 vectors x n: r, w = S u, s (neighbour-visible, double-buffered)
    double* pcg_own;         // 2 x n: u, p of the own rows
    double* pcg_parts;       // 2 x nparts x 4: per-workgroup partial sums (gamma, delta, |r|^2)
    double* pcg_scal;        // 2 x 4: (gamma, alpha) of the previous iteration
    int pcg_nparts;
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
    // ---- tracked-frame chain (k_pose_opt<EQ, true>): the kernel gathers its own observations from the frame's keypoints, the matcher's
    //      result and the resident landmark table, and leaves its results where the host reads them without a copy (svgpu_track.hip)
    const double* pose_in_dev;     // nullable: the starting pose in device memory (the previous optimisation's result)
    const int32_t* trk_overflow;   // the candidate-list allocation counter: beyond trk_overflow_cap the matcher did not run -> do nothing
    int trk_overflow_cap;
    const int32_t* trk_match_q;    // trk_nq: keypoint a query was matched to or -1 (k_cand_replay*)
    const int32_t* trk_qid;        // trk_nq: landmark id of a query
    int trk_nq;
    int32_t* trk_cur_lm;           // per keypoint: landmark id held (-1 none); the matches are applied to it in increasing query order
    int trk_reset_cur;             // 1: the frame holds no landmark before the matches (curr_frm.erase_landmarks(), frame_tracker.cc:29)
    int32_t* trk_who;              // per keypoint scratch (last query that took it)
    const int32_t* trk_nt_dev;     // nullable: keypoint count in device memory
    int trk_nt;                    // keypoint count (capacity when trk_nt_dev)
    const void* trk_map;           // svgpu_landmark_record table
    int trk_map_cap;
    const float* trk_xy;           // undistorted keypoints
    const int32_t* trk_octave;
    const float* trk_xright;       // nullable
    float trk_inv_sigma_sq[16];
    float trk_huber;               // sqrt(5.99146) monocular, sqrt(7.81473) otherwise (pose_optimizer_g2o.cc:98-100)
    double* trk_pos;               // compacted observations (capacity trk_nt): written here, then read through pos_w / uvr / inv_sigma_sq / huber
    float* trk_uvr;
    float* trk_w;
    float* trk_h;
    int32_t* trk_kp_of;            // observation -> keypoint
    uint8_t* trk_outlier_kp;       // per keypoint (device)
    uint8_t* host_outlier_kp;      // per keypoint (page-locked)
    double* host_pose;             // 12 (page-locked)
    int* host_result;              // [0] num_valid [1] LM iterations [2] num_bad [3] observations (page-locked)
    int32_t* trk_counter_reset;    // nullable: a device word zeroed at the end (the next matcher's list allocation counter)
    unsigned long long* stamps;    // nullable (SVGPU_TRACK_STAMPS): wall_clock64 at the kernel's phase boundaries, [0] = how many
};
struct svgpu_ctx;
void sv_pose_opt(svgpu_ctx* ctx, hipStream_t s, const PoseOptDev& P);
void sv_ba_linearize(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int do_prepare);
void sv_ba_reduce(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);  // Schur complement blocks + right-hand side
void sv_ba_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);   // on-chip dense LL^T
size_t sv_ba_chol_bytes(int n);                                    // its dynamic LDS
size_t sv_ba_pcg_lds_bytes(const BaDev& D);
void sv_ba_solve_pcg_lds(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);  // PCG with the whole system in one workgroup's LDS
int sv_ba_lin_split(int E, int nP);
int sv_ba_lin_split_max();
int sv_ba_lm_blocks(int L);  // landmark-side workgroups of k_ba_lin
int sv_ba_rhs_split();
void sv_ba_chi2(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int use_trial, int store_cache, int guarded);
void sv_ba_gate(hipStream_t s, const BaDev& D, int set_levels, uint8_t* outlier_out);
void sv_ba_pack_out(hipStream_t s, const BaDev& D, double* out);  // poses | points of the current estimate, contiguous
void sv_ba_fold(hipStream_t s, const BaDev& D, double* out4, int with_scale);   // this rank's partial sums -> 4 doubles (sharded solve)
void sv_ba_activity(hipStream_t s, const BaDev& D, uint8_t* pt_free);               // after the gate: landmark activity, gated count, "pose numbering changed" on the device
void sv_ba_begin(hipStream_t s, const BaDev& D, int it_max, int stop_in);       // start of SparseOptimizer::optimize(it_max)
void sv_ba_prepare(hipStream_t s, const BaDev& D);                              // lambda init (iteration 0) + start of a trial
void sv_ba_decide(hipStream_t s, const BaDev& D);
bool sv_ba_tail_ok(const BaDev& D);                                             // local-BA sized, not sharded
void sv_ba_tail(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);                 // update + chi2 of the trial state in one launch                               // rho test, damping update, terminate_action
void sv_ba_solve_dense(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);          // dense image in global memory + one-workgroup LL^T (solver = dense)
void sv_ba_update(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);               // back-substitution, trial state
// block-Jacobi PCG on the block-sparse reduced camera system (ba_pcg.hip)
void sv_pcg_init(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);
void sv_pcg_iterate(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, int first_it, int count);
enum { SV_BA_SOLVER_AUTO = 0, SV_BA_SOLVER_CHOLESKY = 1, SV_BA_SOLVER_PCG = 2, SV_BA_SOLVER_DENSE = 3, SV_BA_SOLVER_PCG_MULTI = 4, SV_BA_SOLVER_PCG_LDS = 5 /* internal */,
       SV_BA_SOLVER_ENVELOPE = 6, SV_BA_SOLVER_CHOLESKY_MFMA = 7 };
// block envelope Cholesky of large reduced systems (ba_skyline.hip)
#ifdef __cplusplus
#include <vector>
int sv_sky_plan(svgpu_ctx* ctx, hipStream_t s, int nP, const std::vector<int2>& blk_ab, size_t max_bytes, bool* usable, int rank, int world);
#endif
int sv_sky_solve(svgpu_ctx* ctx, hipStream_t s, const BaDev& D);
void sv_sky_release(svgpu_ctx* ctx);
