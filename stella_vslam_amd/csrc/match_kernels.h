// Kernel-argument structs of the matcher kernels (internal).
#pragma once
#include <cstdint>

#include "svgpu.h"

struct svgpu_ctx;
#define BF_K 16     // per-query candidate prefix kept by k_bf_topk (candidates within dmax only; exact fallback when exhausted)
#define BF_LIST 16  // row stride of the per-query candidate lists (>= BF_K and >= MF_SLOTS of k_bf_mfma)

struct BfProblem {
    // side 1 = frame (scanned), side 2 = keyframe (queries); `pairs` independent problems, row p at p*cap
    const uint32_t* desc1;
    const uint32_t* desc2;
    const float* angle1;  // element (p*cap1 + i) * angle_stride
    const float* angle2;
    int angle_stride;     // 1 for plain float arrays, 7 when pointing at svgpu_keypoint::angle
    const int32_t* n1_dev;  // nullable: per-pair counts in device memory (index p * n_stride)
    const int32_t* n2_dev;
    int n_stride;
    int n1, n2;           // used when the *_dev pointers are null
    int cap1, cap2;
    int ring1;            // 0: side-1 row of pair p is row p; R > 0: it is row (p + 1) % R (frame t+1 against frame t in a ring of R frames)
    const uint8_t* valid2;  // nullable
    float lowe_ratio;
    int check_orientation;
    unsigned dmax;        // candidates farther than this are never listed (see k_bf_topk)
    int exhaustive;       // 1 when dmax >= 256: the list is a plain prefix of all candidates
    int list_k;           // entries a list row holds at most (BF_K from k_bf_topk, MF_SLOTS from k_bf_mfma), rows are sorted
    // angle-bin sorted copies (k_bf_binsort): descriptors, angles, original indices, bin starts (362 per pair and side)
    uint32_t *sd1, *sd2;
    float *sa1, *sa2;
    int *si1, *si2;
    int *bs1, *bs2;
    int* prune_ok;        // pairs * 2: every angle of the side lies in [0, 360]
    int shared_sort;      // ring mode with one sorted copy per frame (see bf_srow1)
    uint32_t* topk;       // pairs * cap2 * BF_LIST
    int32_t* cnt;         // pairs * cap2
    int32_t* matched;     // pairs * cap1
    int32_t* num;         // pairs
    unsigned long long* mfma_tiles;  // nullable (profiling only): k_bf_mfma adds the 64 x 32 patches its waves multiplied (16 MFMAs of 32x32x32 each)
};

__host__ __device__ inline int bf_row1(const BfProblem& P, int pair) { return P.ring1 ? (pair + 1 == P.ring1 ? 0 : pair + 1) : pair; }
// where the angle-sorted copy of side 1 of `pair` lives.  Ring mode (frame t + 1 against frame t inside ONE set of arrays): a frame is
// side 1 of one pair and side 2 of the next, so every frame is sorted once (as side 2 of its own pair) and sd1 / sa1 / si1 / bs1
// alias the side-2 arrays.
__host__ __device__ inline int bf_srow1(const BfProblem& P, int pair) { return P.shared_sort ? bf_row1(P, pair) : pair; }
__host__ __device__ inline int bf_prune1(const BfProblem& P, int pair) { return P.shared_sort ? P.prune_ok[bf_row1(P, pair) * 2 + 1] : P.prune_ok[pair * 2]; }

struct CandProblem {
    const uint32_t* qdesc;
    const uint32_t* tdesc;
    const int32_t* t_octave;  // nullable
    int nq, nt;
    const int32_t* cand_off;
    const int32_t* cand_idx;
    const uint8_t* cand_skip; // nullable, per CSR entry
    const uint8_t* q_valid;   // nullable
    const uint8_t* occupied;  // nullable
    const float* q_angle;
    const float* t_angle;
    int check_orientation;
    const float* q_xright;    // nullable trio
    const float* t_xright;
    const float* q_xr_tol;
    unsigned thr;
    float lowe_ratio;
    int mode;
    const uint8_t* q_blocks;  // nullable: 0 = an accepted query does NOT occupy its target for later queries (the landmark it
                              // carries has no observation: `curr_lm && curr_lm->has_observation()`, projection.cc:167-170)
    int no_claims;            // 1 = queries are independent (projection::match_keyframes_mutually, :498-510: no bookkeeping at all)
    // fuse::detect_duplication's reprojection gate (fuse.cc:92-119), evaluated per CSR entry
    int chi_gate;
    const double* q_reproj;   // nq x 2
    const float* q_reproj_xr; // nq
    const float* t_xy;        // nt x 2 undistorted keypoints
    const float* chi_t_xright;// nullable: stereo_x_right_ of the keyframe
    float inv_level_sigma_sq[16];
    uint32_t* dist;           // one per CSR entry: (distance << 22) | target index, 0xFFFFFFFF = gated out (targets < 2^22)
    int32_t* match_q;
    int32_t* num;
    int cap;                  // > 0: capacity of dist / cand_idx; the kernels do nothing when cand_off[nq] exceeds it (the host re-runs)
    // tracked-frame chain (track_kernels.hip): lists allocated in arbitrary order, counts known to the device only, results also written
    // where the host reads them without a copy
    const int32_t* cand_cnt;  // nullable: list q = [cand_off[q], cand_off[q] + cand_cnt[q])
    const int32_t* cand_total;// nullable: where the lists' total (or the allocation counter of the slot form) lives; default cand_off + nq
    const int32_t* nt_dev;    // nullable: the number of targets in device memory (at most nt, which then only sizes the tables)
    int32_t* match_host;      // nullable: page-locked copy of match_q
    int32_t* num_host;        // nullable: page-locked copy of *num, followed by the list total (cand_off[nq])
    int dbg_phase;            // timing experiments only (SVGPU_REPLAY_DBG): 1 = return after the set-up, 2 = one evaluation per chunk, no sweeps
};
void sv_launch_cand_replay(svgpu_ctx* ctx, hipStream_t s, const CandProblem& P, int* owner, int* match);  // the replay alone (lists + distances in place)

// Device-side candidate lists: data::assign_keypoints_to_grid + data::get_keypoints_in_cell (data/common.cc:83-190)
struct GridProblem {
    const float* t_xy;        // nt x 2 undistorted keypoint positions
    const int32_t* t_octave;  // nt
    int nt;
    float min_x, min_y;
    double inv_w, inv_h;      // (double)cols / (max_x - min_x), (double)rows / (max_y - min_y)
    int cols, rows;
    int32_t* cell_of;         // nt: col * rows + row, or -1 outside the grid
    int32_t* cell_off;        // cols * rows + 1
    int32_t* cell_items;      // nt
    const float* q_xy;        // nq x 2 reference points
    const float* q_margin;    // nq
    const int32_t* q_min_level;  // nq, < 0 = unbounded
    const int32_t* q_max_level;
    const uint8_t* q_valid;   // nullable
    int nq;
    int32_t* cand_off;        // nq + 1 (counts, then their exclusive scan)
    int32_t* cand_idx;        // filled by the second walk
    int cap;                  // > 0: capacity of cand_idx; the fill does nothing when the lists' total (cand_off[nq]) exceeds it (the host re-runs)
};
void sv_launch_grid_build(hipStream_t s, const GridProblem& G);                 // cell_of, cell_off, cell_items, cand_off (scanned)
void sv_launch_grid_frame(hipStream_t s, const GridProblem& G);                 // ... the keypoint side alone (a resident frame is binned once)
void sv_launch_grid_queries(hipStream_t s, const GridProblem& G);               // ... the query side over an already binned frame
void sv_launch_grid_fill(hipStream_t s, const GridProblem& G);                  // cand_idx

struct StereoProblem {
    const svgpu_keypoint* kl;
    const svgpu_keypoint* kr;
    const uint32_t* dl;
    const uint32_t* dr;
    int nl, nr, num_levels;
    const uint8_t* lev_l[16];  // pyramid level base pointers / pitches of the two extractors
    const uint8_t* lev_r[16];
    int pitch_l[16], pitch_r[16], w[16], h[16];
    float sf[16], isf[16];
    float fxb, min_disp, max_disp;
    unsigned thr;
    float* xr;    // nl
    float* depth; // nl
    float* corr;  // nl: best L1 correlation (integer valued) or -1
    // batched, device-resident form (svgpu_stereo_match_batch_device): pair p = blockIdx.y reads its keypoints / descriptors at
    // p * cap, its counts at n*_dev[p * n_stride], its pyramids at p * frame strides; all zero / null = one pair described above
    const int32_t* nl_dev;
    const int32_t* nr_dev;
    int n_stride, cap;
    size_t img_stride_l, img_stride_r;   // level 0 (the callers' images)
    size_t pyr_stride_l, pyr_stride_r;   // levels >= 1 (the extractors' pyramid blocks)
    // row index of the right keypoints (get_right_keypoint_indices_in_each_row, stereo.cc:34-60): per pair rows + 1 offsets, a fill cursor per
    // row and the right-keypoint indices of every row band (rows_per_kp entries per keypoint at most); built by sv_launch_stereo
    int rows;            // image rows of level 0
    int rows_per_kp;     // >= rows of the widest band: 2 * ceil(2 * scale_factor[top level]) + 3
    int32_t* row_off;    // pairs x (rows + 1)
    int32_t* row_fill;   // pairs x rows
    int32_t* row_items;  // pairs x nr (cap) x rows_per_kp
};
size_t sv_stereo_rows_bytes(int pairs, int rows, int nr_cap, int rows_per_kp);  // scratch for the three arrays above
void sv_launch_stereo(svgpu_ctx* ctx, hipStream_t s, const StereoProblem& P, int pairs = 1);
void sv_launch_stereo_median(hipStream_t s, const StereoProblem& P, int pairs);  // 2 x median correlation filter (stereo.cc:94-113) on the device

void sv_launch_hamming_pairs(hipStream_t s, const uint32_t* a, const uint32_t* b, int n, uint32_t* out);
void sv_launch_hamming_matrix(hipStream_t s, const uint32_t* d1, int n1, const uint32_t* d2, int n2, uint16_t* out);
void sv_launch_bf(svgpu_ctx* ctx, hipStream_t s, const BfProblem& P, int pairs, int* g_owner, int* g_match);
void sv_launch_cand(svgpu_ctx* ctx, hipStream_t s, const CandProblem& P, int* owner, int* match, unsigned* mdist);

// Bucketed candidate lists (bucket_kernels.hip): bow_tree::* and robust::match_for_triangulation
struct BucketProblem {
    int n1, n2;
    const uint32_t* desc1;     // n1 x 8
    const uint32_t* desc2;     // n2 x 8
    const float* angle1;       // nullable when check_orientation == 0
    const float* angle2;
    const uint8_t* valid1;     // nullable: 0 = side-1 keypoint is not a query
    const uint8_t* valid2;     // nullable: 0 = side-2 keypoint is never a candidate
    int check_orientation;
    // triangulation gates (match/robust.cc:56-117, bow_tree.cc:66-130)
    int tri;
    unsigned thr;              // Hamming cut-off applied in the scan (tri only)
    const int32_t* octave1;
    const double* bearings1;   // n1 x 3
    const double* bearings2;
    const float* xright1;      // nullable
    const float* xright2;
    double E12[9];
    double epipole[3];
    int valid_epipole;
    float scale_factors[16];
    float residual_rad_thr;
    // (node, index) orders and rows
    const unsigned* key1;      // n1 sorted node ids of side 1
    const int* q_idx;          // n1: row -> side-1 keypoint
    const unsigned* key2;      // n2 sorted node ids of side 2
    const int* t_sorted;       // n2: position in the sorted side 2 -> keypoint
    int* row_lo;
    int* row_hi;
    uint8_t* q_valid;          // n1 rows
    int32_t* cand_off;         // n1 + 1
    int32_t* cand_idx;
};
size_t sv_bucket_sort_bytes(int n);
int sv_bucket_sort(svgpu_ctx* ctx, hipStream_t s, const int32_t* node_dev, int n, void* scratch, size_t scratch_bytes, unsigned* keys_out, int* idx_out);
void sv_bucket_rows(hipStream_t s, const BucketProblem& B);
void sv_bucket_count(hipStream_t s, const BucketProblem& B);
void sv_bucket_fill(hipStream_t s, const BucketProblem& B);
void sv_bucket_gather_rows(hipStream_t s, const BucketProblem& B, uint32_t* qdesc_rows, float* qangle_rows);
void sv_bucket_scatter(hipStream_t s, const BucketProblem& B, const int32_t* match_rows, int32_t* match_q);
