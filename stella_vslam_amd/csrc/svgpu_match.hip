// Host side of the matchers: staging of host buffers into the context's scratch arena, launches.
#include <algorithm>
#include <cmath>
#include <utility>

#include "svgpu_match_common.h"

using namespace svm;

extern "C" {

int svgpu_hamming_distance(svgpu_ctx* ctx, const uint8_t* a, const uint8_t* b, int n, uint32_t* dist) {
    if (!ctx || n < 0 || (n > 0 && (!a || !b || !dist))) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_hamming_distance");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nb = (size_t)n * 32;
    int rc = sv_ensure_scratch(ctx, 2 * pad(nb) + pad((size_t)n * 4));
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    uint32_t* da = A.take<uint32_t>((size_t)n * 8);
    uint32_t* db = A.take<uint32_t>((size_t)n * 8);
    uint32_t* dd = A.take<uint32_t>(n);
    hipStream_t s = ctx->stream;
    SV_HIP(ctx, hipMemcpyAsync(da, a, nb, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(db, b, nb, hipMemcpyHostToDevice, s));
    sv_launch_hamming_pairs(s, da, db, n, dd);
    SV_HIP(ctx, hipMemcpyAsync(dist, dd, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_hamming_matrix(svgpu_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, uint16_t* out) {
    if (!ctx || n1 < 0 || n2 < 0) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_hamming_matrix");
    if (n1 == 0 || n2 == 0) return SVGPU_OK;
    if (!desc1 || !desc2 || !out) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_hamming_matrix: null pointer");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sv_ensure_scratch(ctx, pad((size_t)n1 * 32) + pad((size_t)n2 * 32) + pad((size_t)n1 * n2 * 2));
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    uint32_t* d1 = A.take<uint32_t>((size_t)n1 * 8);
    uint32_t* d2 = A.take<uint32_t>((size_t)n2 * 8);
    uint16_t* dm = A.take<uint16_t>((size_t)n1 * n2);
    hipStream_t s = ctx->stream;
    SV_HIP(ctx, hipMemcpyAsync(d1, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d2, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    sv_launch_hamming_matrix(s, d1, n1, d2, n2, dm);
    SV_HIP(ctx, hipMemcpyAsync(out, dm, (size_t)n1 * n2 * 2, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

}  // extern "C"

namespace {
int bf_batch_device(svgpu_ctx* ctx, int pairs, int ring1, const uint8_t* desc1_dev, const svgpu_keypoint* kps1_dev, const int32_t* n1_dev,
                    int cap1, const uint8_t* desc2_dev, const svgpu_keypoint* kps2_dev, const int32_t* n2_dev, int cap2, int n_stride,
                    const uint8_t* valid2_dev, float lowe_ratio, int check_orientation, int32_t* matched_dev, int32_t* num_dev,
                    void* stream, const float* angles1_dev = nullptr, const float* angles2_dev = nullptr) {
    if (!ctx || pairs < 1 || !desc1_dev || !kps1_dev || !n1_dev || !desc2_dev || !kps2_dev || !n2_dev || cap1 < 1 || cap2 < 1
        || cap1 > 65535 || cap2 > 65535 || !matched_dev || !num_dev)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_bruteforce_batch_device: bad arguments");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = pad((size_t)pairs * cap2 * BF_LIST * 4) + pad((size_t)pairs * cap2 * 4) + pad((size_t)pairs * cap1 * 4)
                        + pad((size_t)pairs * cap2 * 4) + sort_bytes(pairs, cap1, cap2);
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    BfProblem P{};
    P.desc1 = (const uint32_t*)desc1_dev;
    P.desc2 = (const uint32_t*)desc2_dev;
    if (angles1_dev && angles2_dev) {  // packed copies (svgpu_orb_extract_batch_device_angles): 4 bytes per keypoint instead of a 28-byte stride
        P.angle1 = angles1_dev;
        P.angle2 = angles2_dev;
        P.angle_stride = 1;
    }
    else {
        P.angle1 = &kps1_dev->angle;
        P.angle2 = &kps2_dev->angle;
        P.angle_stride = sizeof(svgpu_keypoint) / sizeof(float);
    }
    P.n1_dev = n1_dev;
    P.n2_dev = n2_dev;
    P.n_stride = n_stride;
    P.cap1 = cap1;
    P.cap2 = cap2;
    P.ring1 = ring1;
    P.valid2 = valid2_dev;
    P.lowe_ratio = lowe_ratio;
    P.check_orientation = check_orientation;
    P.topk = A.take<uint32_t>((size_t)pairs * cap2 * BF_LIST);
    P.cnt = A.take<int32_t>((size_t)pairs * cap2);
    int* g_owner = A.take<int>((size_t)pairs * cap1);
    int* g_match = A.take<int>((size_t)pairs * cap2);
    take_sort(A, P, pairs, cap1, cap2);
    P.matched = matched_dev;
    P.num = num_dev;
    sv_launch_bf(ctx, stream ? (hipStream_t)stream : ctx->stream, P, pairs, g_owner, g_match);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}
}  // namespace

extern "C" {

int svgpu_match_bruteforce_batch_device(svgpu_ctx* ctx, int pairs, const uint8_t* desc1_dev,
                                        const svgpu_keypoint* kps1_dev, const int32_t* n1_dev, int cap1,
                                        const uint8_t* desc2_dev, const svgpu_keypoint* kps2_dev,
                                        const int32_t* n2_dev, int cap2, int n_stride, const uint8_t* valid2_dev,
                                        float lowe_ratio, int check_orientation, int32_t* matched_dev,
                                        int32_t* num_dev, void* stream) {
    return bf_batch_device(ctx, pairs, 0, desc1_dev, kps1_dev, n1_dev, cap1, desc2_dev, kps2_dev, n2_dev, cap2, n_stride, valid2_dev, lowe_ratio,
                           check_orientation, matched_dev, num_dev, stream);
}

int svgpu_match_consecutive_batch_device(svgpu_ctx* ctx, int frames, const uint8_t* desc_dev, const svgpu_keypoint* kps_dev,
                                         const int32_t* n_dev, int cap, int n_stride, const uint8_t* valid_dev, float lowe_ratio,
                                         int check_orientation, int32_t* matched_dev, int32_t* num_dev, void* stream) {
    if (frames < 2) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_consecutive_batch_device: needs at least two frames");
    return bf_batch_device(ctx, frames, frames, desc_dev, kps_dev, n_dev, cap, desc_dev, kps_dev, n_dev, cap, n_stride, valid_dev, lowe_ratio,
                           check_orientation, matched_dev, num_dev, stream);
}

int svgpu_match_consecutive_batch_device_angles(svgpu_ctx* ctx, int frames, const uint8_t* desc_dev, const svgpu_keypoint* kps_dev, const float* angles_dev,
                                                const int32_t* n_dev, int cap, int n_stride, const uint8_t* valid_dev, float lowe_ratio,
                                                int check_orientation, int32_t* matched_dev, int32_t* num_dev, void* stream) {
    if (frames < 2) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_consecutive_batch_device: needs at least two frames");
    return bf_batch_device(ctx, frames, frames, desc_dev, kps_dev, n_dev, cap, desc_dev, kps_dev, n_dev, cap, n_stride, valid_dev, lowe_ratio,
                           check_orientation, matched_dev, num_dev, stream, angles_dev, angles_dev);
}

int svgpu_match_bruteforce(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2,
                           const float* angle2, const uint8_t* valid2, int n2, float lowe_ratio,
                           int check_orientation, int32_t* matched_2_in_1, int* num_matches) {
    if (!ctx || n1 < 0 || n2 < 0 || n1 > 65535 || n2 > 65535 || !num_matches)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_bruteforce: bad sizes");
    *num_matches = 0;
    if (n1 > 0 && !matched_2_in_1) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_bruteforce: null output");
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    if (n1 == 0 || n2 == 0) return SVGPU_OK;
    if (!desc1 || !desc2 || (check_orientation && (!angle1 || !angle2)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_bruteforce: null input");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = pad((size_t)n1 * 32) + pad((size_t)n2 * 32) + pad((size_t)n1 * 4) + pad((size_t)n2 * 4) + pad(n2)
                        + pad((size_t)n2 * BF_LIST * 4) + pad((size_t)n2 * 4) + 2 * pad((size_t)n1 * 4) + pad((size_t)n2 * 4) + 256
                        + sort_bytes(1, n1, n2);
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    uint32_t* d1 = A.take<uint32_t>((size_t)n1 * 8);
    uint32_t* d2 = A.take<uint32_t>((size_t)n2 * 8);
    float* a1 = A.take<float>(n1);
    float* a2 = A.take<float>(n2);
    uint8_t* v2 = A.take<uint8_t>(n2);
    BfProblem P{};
    P.topk = A.take<uint32_t>((size_t)n2 * BF_LIST);
    P.cnt = A.take<int32_t>(n2);
    P.matched = A.take<int32_t>(n1);
    int* g_owner = A.take<int>(n1);
    int* g_match = A.take<int>(n2);
    P.num = A.take<int32_t>(1);
    take_sort(A, P, 1, n1, n2);
    hipStream_t s = ctx->stream;
    // inputs through the page-locked mirror of the arena: one copy down (the five arrays are the first takes: one contiguous range), one up
    if ((rc = sv_ensure_stage(ctx, need))) return rc;
    A.mirror = ctx->h_stage;
    if ((rc = A.upload(ctx, s, d1, desc1, (size_t)n1 * 32))) return rc;
    if ((rc = A.upload(ctx, s, d2, desc2, (size_t)n2 * 32))) return rc;
    if (angle1) rc = A.upload(ctx, s, a1, angle1, (size_t)n1 * 4);
    else memset(A.mirror + ((char*)a1 - A.base), 0, (size_t)n1 * 4);
    if (rc) return rc;
    if (angle2) rc = A.upload(ctx, s, a2, angle2, (size_t)n2 * 4);
    else memset(A.mirror + ((char*)a2 - A.base), 0, (size_t)n2 * 4);
    if (rc) return rc;
    if (valid2 && (rc = A.upload(ctx, s, v2, valid2, n2))) return rc;
    if (!angle1 || !angle2) A.up_lo = 0, A.up_hi = std::max(A.up_hi, (size_t)((char*)(a2 + n2) - A.base));  // the zeroed angle rows travel with the range
    if ((rc = A.flush(ctx, s))) return rc;
    P.desc1 = d1;
    P.desc2 = d2;
    P.angle1 = a1;
    P.angle2 = a2;
    P.angle_stride = 1;
    P.n1 = n1;
    P.n2 = n2;
    P.cap1 = n1;
    P.cap2 = n2;
    P.valid2 = valid2 ? v2 : nullptr;
    P.lowe_ratio = lowe_ratio;
    P.check_orientation = check_orientation;
    sv_launch_bf(ctx, s, P, 1, g_owner, g_match);
    SV_HIP(ctx, hipGetLastError());
    int32_t num = 0;
    Downloads D;
    D.add(A, matched_2_in_1, P.matched, (size_t)n1 * 4);
    D.add(A, &num, P.num, 4);
    if ((rc = D.fetch(ctx, s, A))) return rc;
    SV_HIP(ctx, hipStreamSynchronize(s));
    D.scatter(A);
    *num_matches = num;
    return SVGPU_OK;
}

int svgpu_match_candidates(svgpu_ctx* ctx, const uint8_t* qdesc, int nq, const uint8_t* tdesc, const int32_t* t_octave,
                           int nt, const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* cand_skip,
                           const uint8_t* q_valid, const uint8_t* occupied, const float* q_angle, const float* t_angle, int check_orientation,
                           const float* q_xright, const float* t_xright, const float* q_xr_tol, unsigned thr,
                           float lowe_ratio, int mode, int32_t* match_q, int* num_matches) {
    if (!ctx || nq < 0 || nt < 0 || nt >= (1 << 22) || !num_matches || (mode < SVGPU_MATCH_BEST_ONLY || mode > SVGPU_MATCH_AREA))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_candidates: bad arguments");
    *num_matches = 0;
    if (nq == 0) return SVGPU_OK;
    if (!match_q || !cand_off) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_candidates: null pointer");
    for (int q = 0; q < nq; ++q) match_q[q] = -1;
    const int nc = cand_off[nq];
    if (nc == 0 || nt == 0) return SVGPU_OK;
    if (!qdesc || !tdesc || !cand_idx || (check_orientation && (!q_angle || !t_angle))
        || ((q_xright || t_xright || q_xr_tol) && !(q_xright && t_xright && q_xr_tol)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_candidates: inconsistent inputs");
    for (int q = 0; q < nq; ++q)
        if (cand_off[q + 1] < cand_off[q]) return sv_set_error(ctx, SVGPU_ERR_INVALID, "cand_off not monotone");
    for (int c = 0; c < nc; ++c)
        if (cand_idx[c] < 0 || cand_idx[c] >= nt) return sv_set_error(ctx, SVGPU_ERR_INVALID, "cand_idx out of range");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = pad((size_t)nq * 32) + pad((size_t)nt * 32) + 3 * pad((size_t)nt * 4) + pad(nt) + pad((size_t)(nq + 1) * 4)
                        + pad((size_t)nc * 4) + pad(nc) + pad(nq) + 3 * pad((size_t)nq * 4) + pad((size_t)nc * 4) + 2 * pad((size_t)nq * 4)
                        + 2 * pad((size_t)nt * 4) + 512;
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    hipStream_t s = ctx->stream;
    if ((rc = sv_ensure_stage(ctx, need))) return rc;
    A.mirror = ctx->h_stage;  // batched upload / read-back (Arena::upload, Downloads)
    CandProblem P{};
#define UP(dst, T, src, n)                                                                          \
    T* dst = nullptr;                                                                               \
    if (src) {                                                                                      \
        dst = A.take<T>(n);                                                                         \
        if ((rc = A.upload(ctx, s, dst, src, (size_t)(n) * sizeof(T)))) return rc;                  \
    }
    UP(d_q, uint8_t, qdesc, (size_t)nq * 32)
    UP(d_t, uint8_t, tdesc, (size_t)nt * 32)
    UP(d_toct, int32_t, t_octave, nt)
    UP(d_off, int32_t, cand_off, nq + 1)
    UP(d_idx, int32_t, cand_idx, nc)
    UP(d_skip, uint8_t, cand_skip, nc)
    UP(d_qv, uint8_t, q_valid, nq)
    UP(d_occ, uint8_t, occupied, nt)
    UP(d_qa, float, q_angle, nq)
    UP(d_ta, float, t_angle, nt)
    UP(d_qx, float, q_xright, nq)
    UP(d_tx, float, t_xright, nt)
    UP(d_qtol, float, q_xr_tol, nq)
#undef UP
    if ((rc = A.flush(ctx, s))) return rc;
    P.qdesc = (const uint32_t*)d_q;
    P.tdesc = (const uint32_t*)d_t;
    P.t_octave = d_toct;
    P.nq = nq;
    P.nt = nt;
    P.cand_off = d_off;
    P.cand_idx = d_idx;
    P.cand_skip = d_skip;
    P.q_valid = d_qv;
    P.occupied = d_occ;
    P.q_angle = d_qa;
    P.t_angle = d_ta;
    P.check_orientation = check_orientation;
    P.q_xright = d_qx;
    P.t_xright = d_tx;
    P.q_xr_tol = d_qtol;
    P.thr = thr;
    P.lowe_ratio = lowe_ratio;
    P.mode = mode;
    P.dist = A.take<uint32_t>(nc);
    P.match_q = A.take<int32_t>(nq);
    P.num = A.take<int32_t>(1);
    int* owner = A.take<int>(nt);
    int* match = A.take<int>(nq);
    unsigned* mdist = A.take<unsigned>(nt);
    sv_launch_cand(ctx, s, P, owner, match, mdist);
    SV_HIP(ctx, hipGetLastError());
    int32_t num = 0;
    Downloads D;
    D.add(A, match_q, P.match_q, (size_t)nq * 4);
    D.add(A, &num, P.num, 4);
    if ((rc = D.fetch(ctx, s, A))) return rc;
    SV_HIP(ctx, hipStreamSynchronize(s));
    D.scatter(A);
    *num_matches = num;
    return SVGPU_OK;
}

int svgpu_match_in_cells(svgpu_ctx* ctx, const uint8_t* qdesc, int nq, const float* q_xy, const float* q_margin,
                         const int32_t* q_min_level, const int32_t* q_max_level, const uint8_t* q_valid, const float* q_angle,
                         const float* q_xright, const float* q_xr_tol, const uint8_t* tdesc, const float* t_xy,
                         const int32_t* t_octave, int nt, const uint8_t* occupied, const float* t_angle, const float* t_xright,
                         float min_x, float max_x, float min_y, float max_y, int grid_cols, int grid_rows,
                         int check_orientation, unsigned thr, float lowe_ratio, int mode, int32_t* match_q, int* num_matches) {
    // both one-shots are consumed FIRST, whatever happens next: an early error return must not leave them armed for an unrelated call
    const svgpu_frame* rf = sv_take_bound_frame(ctx);
    const uint8_t* const q_blocks = ctx ? ctx->next_q_blocks : nullptr;  // svgpu_match_set_query_blocks
    if (ctx) ctx->next_q_blocks = nullptr;
    if (rf) {  // resident keypoint side: the frame's own device arrays, bounds and grid (svgpu_frame_bind)
        tdesc = rf->desc, t_xy = rf->xy, t_octave = rf->octave, nt = rf->n;
        t_angle = check_orientation ? rf->angle : nullptr;
        t_xright = (q_xright && rf->has_xright) ? rf->xright : nullptr;
        if (!t_xright) q_xright = nullptr, q_xr_tol = nullptr;  // (a monocular frame has no stereo gate: projection.cc:57)
        min_x = rf->min_x, max_x = rf->max_x, min_y = rf->min_y, max_y = rf->max_y, grid_cols = rf->grid_cols, grid_rows = rf->grid_rows;
    }
    if (!ctx || nq < 0 || nt < 0 || nt >= (1 << 22) || !num_matches || mode < SVGPU_MATCH_BEST_ONLY || mode > SVGPU_MATCH_AREA || grid_cols < 1 || grid_rows < 1
        || (size_t)grid_cols * grid_rows > (size_t(1) << 22) || !(min_x < max_x) || !(min_y < max_y))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_in_cells: bad arguments");
    *num_matches = 0;
    if (nq == 0) return SVGPU_OK;
    if (!match_q) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_in_cells: null pointer");
    for (int q = 0; q < nq; ++q) match_q[q] = -1;
    if (nt == 0) return SVGPU_OK;
    if (!qdesc || !q_xy || !q_margin || !tdesc || !t_xy || !t_octave || (check_orientation && (!q_angle || !t_angle))
        || ((q_xright || t_xright || q_xr_tol) && !(q_xright && t_xright && q_xr_tol)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_in_cells: inconsistent inputs");
    InCellsFrame F{tdesc, t_xy, t_octave, nt, occupied, t_angle, t_xright, min_x, max_x, min_y, max_y, grid_cols, grid_rows};
    F.res = rf;
    const size_t qbytes = pad((size_t)nq * 32) + pad((size_t)nq * 8) + 6 * pad((size_t)nq * 4) + 2 * pad(nq);
    hipStream_t s = ctx->stream;
    return in_cells_core(
        ctx, nq, F, qbytes, check_orientation, thr, lowe_ratio, mode,
        [&](Arena& A, bool fresh, CandProblem& P, GridProblem& G) -> int {
#define UP(dst, T, src, n)                                                                          \
    T* dst = nullptr;                                                                               \
    if (src) {                                                                                      \
        dst = A.take<T>(n);                                                                         \
        if (fresh) {                                                                                \
            const int rcu = A.upload(ctx, s, dst, src, (size_t)(n) * sizeof(T));                    \
            if (rcu) return rcu;                                                                    \
        }                                                                                           \
    }
            UP(d_q, uint8_t, qdesc, (size_t)nq * 32)
            UP(d_qxy, float, q_xy, (size_t)nq * 2)
            UP(d_qm, float, q_margin, nq)
            UP(d_qlo, int32_t, q_min_level, nq)
            UP(d_qhi, int32_t, q_max_level, nq)
            UP(d_qv, uint8_t, q_valid, nq)
            UP(d_qa, float, q_angle, nq)
            UP(d_qx, float, q_xright, nq)
            UP(d_qtol, float, q_xr_tol, nq)
            UP(d_qb, uint8_t, q_blocks, nq)
#undef UP
            G.q_xy = d_qxy;
            G.q_margin = d_qm;
            G.q_min_level = d_qlo;
            G.q_max_level = d_qhi;
            G.q_valid = d_qv;
            P.qdesc = (const uint32_t*)d_q;
            P.q_valid = d_qv;
            P.q_angle = d_qa;
            P.q_xright = d_qx;
            P.q_xr_tol = d_qtol;
            P.q_blocks = d_qb;
            return SVGPU_OK;
        },
        [&](const CandProblem&, const Arena&, Downloads&) -> int { return SVGPU_OK; }, match_q, num_matches);
}

int svgpu_camera_image_bounds(svgpu_ctx* ctx, svgpu_camera* cam) {
    if (!ctx || !cam || cam->model < SVGPU_CAM_PERSPECTIVE || cam->model > SVGPU_CAM_RADIAL_DIVISION)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_camera_image_bounds: bad arguments");
    const float cols = (float)cam->cols, rows = (float)cam->rows;
    const int nd = cam->model == SVGPU_CAM_PERSPECTIVE ? 5 : cam->model == SVGPU_CAM_FISHEYE ? 4 : cam->model == SVGPU_CAM_RADIAL_DIVISION ? 1 : 0;
    bool none = true;
    for (int i = 0; i < nd; ++i) none = none && cam->dist[i] == 0;
    if (none) {  // "any distortion does not exist"
        cam->min_x = 0.0f, cam->max_x = cols, cam->min_y = 0.0f, cam->max_y = rows;
        return SVGPU_OK;
    }
    bool wide = false;
    if (cam->model == SVGPU_CAM_FISHEYE) {  // fisheye.cc:76-83 (issue #83): corners beyond 90 degrees
        const double pwx = (0.0 - cam->cx) / cam->fx, pwy = (0.0 - cam->cy) / cam->fy;
        wide = std::sqrt(pwx * pwx + pwy * pwy) > M_PI / 2;
    }
    svgpu_keypoint q[4] = {}, u[4];
    if (wide) {
        q[0].x = (float)cam->cx, q[0].y = 0.f, q[1].x = cols, q[1].y = (float)cam->cy, q[2].x = 0.f, q[2].y = (float)cam->cy, q[3].x = (float)cam->cx, q[3].y = rows;
    }
    else {
        q[0].x = 0.f, q[0].y = 0.f, q[1].x = cols, q[1].y = 0.f, q[2].x = 0.f, q[2].y = rows, q[3].x = cols, q[3].y = rows;
    }
    int rc = svgpu_frame_observation(ctx, cam, q, 4, 1, 1, u, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (wide) {  // fisheye.cc:98-116
        constexpr float deg_thr = 5.0;
        const float tx = cam->fx / std::tan(deg_thr * M_PI / 180.0), ty = cam->fy / std::tan(deg_thr * M_PI / 180.0);
        const float mnx = -tx + cam->cx, mxx = tx + cam->cx, mny = -ty + cam->cy, mxy = ty + cam->cy;
        const float a = u[2].x, b = u[1].x, c = u[0].y, d = u[3].y;
        cam->min_x = (a < mnx || a > cam->cx) ? mnx : a;
        cam->max_x = (b > mxx || b < cam->cx) ? mxx : b;
        cam->min_y = (c < mny || c > cam->cy) ? mny : c;
        cam->max_y = (d > mxy || d < cam->cy) ? mxy : d;
    }
    else {
        cam->min_x = std::min(u[0].x, u[2].x);
        cam->max_x = std::max(u[1].x, u[3].x);
        cam->min_y = std::min(u[0].y, u[1].y);
        cam->max_y = std::max(u[2].y, u[3].y);
    }
    return SVGPU_OK;
}

int svgpu_frame_observation(svgpu_ctx* ctx, const svgpu_camera* cam, const svgpu_keypoint* kps, int n, int grid_cols,
                            int grid_rows, svgpu_keypoint* undist_kps, double* bearings, int32_t* cell_off, int32_t* cell_items) {
    if (!ctx || !cam || n < 0 || cam->model < SVGPU_CAM_PERSPECTIVE || cam->model > SVGPU_CAM_RADIAL_DIVISION || grid_cols < 1 || grid_rows < 1
        || (size_t)grid_cols * grid_rows > (size_t(1) << 22) || (n > 0 && !kps))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_observation: bad arguments");
    const bool want_grid = cell_off != nullptr;
    if (want_grid && (!(cam->min_x < cam->max_x) || !(cam->min_y < cam->max_y)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_observation: image bounds are not set (svgpu_camera_image_bounds)");
    const int ncell = grid_cols * grid_rows;
    if (n == 0) {
        if (want_grid) std::memset(cell_off, 0, (size_t)(ncell + 1) * sizeof(int32_t));
        return SVGPU_OK;
    }
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t need = 2 * pad((size_t)n * 28) + pad((size_t)n * 8) + pad((size_t)n * 24) + 2 * pad((size_t)n * 4) + pad((size_t)(ncell + 1) * 4) + 1024;
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    FrameObsProblem P{};
    P.cam = *cam;
    svgpu_keypoint* d_in = A.take<svgpu_keypoint>(n);
    P.kps = d_in;
    P.n = n;
    P.undist = undist_kps ? A.take<svgpu_keypoint>(n) : nullptr;
    P.undist_xy = A.take<float>((size_t)n * 2);
    P.bearings = bearings ? A.take<double>((size_t)n * 3) : nullptr;
    SV_HIP(ctx, hipMemcpyAsync(d_in, kps, (size_t)n * sizeof(svgpu_keypoint), hipMemcpyHostToDevice, s));
    sv_launch_frame_observation(s, P);
    if (undist_kps) SV_HIP(ctx, hipMemcpyAsync(undist_kps, P.undist, (size_t)n * sizeof(svgpu_keypoint), hipMemcpyDeviceToHost, s));
    if (bearings) SV_HIP(ctx, hipMemcpyAsync(bearings, P.bearings, (size_t)n * 24, hipMemcpyDeviceToHost, s));
    if (want_grid) {
        GridProblem G{};
        G.t_xy = P.undist_xy;
        G.nt = n;
        G.min_x = cam->min_x;
        G.min_y = cam->min_y;
        G.inv_w = (double)grid_cols / (cam->max_x - cam->min_x);
        G.inv_h = (double)grid_rows / (cam->max_y - cam->min_y);
        G.cols = grid_cols;
        G.rows = grid_rows;
        G.cell_of = A.take<int32_t>(n);
        G.cell_off = A.take<int32_t>(ncell + 1);
        G.cell_items = A.take<int32_t>(n);
        G.nq = 0;
        G.cand_off = A.take<int32_t>(1);
        sv_launch_grid_build(s, G);
        SV_HIP(ctx, hipMemcpyAsync(cell_off, G.cell_off, (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        if (cell_items && cell_off[ncell] > 0)
            SV_HIP(ctx, hipMemcpyAsync(cell_items, G.cell_items, (size_t)cell_off[ncell] * 4, hipMemcpyDeviceToHost, s));
    }
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_keypoints_to_bearings(svgpu_ctx* ctx, const svgpu_camera* cam, const svgpu_keypoint* undist_kps, int n, double* bearings) {
    if (!ctx || !cam || n < 0 || cam->model < SVGPU_CAM_PERSPECTIVE || cam->model > SVGPU_CAM_RADIAL_DIVISION || (n > 0 && (!undist_kps || !bearings)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_keypoints_to_bearings: bad arguments");
    if (n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = sv_ensure_scratch(ctx, pad((size_t)n * 28) + pad((size_t)n * 24) + 256);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    FrameObsProblem P{};
    P.cam = *cam;
    svgpu_keypoint* d_in = A.take<svgpu_keypoint>(n);
    P.kps = d_in;
    P.n = n;
    P.already_undistorted = 1;
    P.bearings = A.take<double>((size_t)n * 3);
    SV_HIP(ctx, hipMemcpyAsync(d_in, undist_kps, (size_t)n * sizeof(svgpu_keypoint), hipMemcpyHostToDevice, s));
    sv_launch_frame_observation(s, P);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(bearings, P.bearings, (size_t)n * 24, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

}  // extern "C"

namespace {
// Shared argument checks + ReprojProblem fill of the two reprojection entry points (host-side scalars only).
int fill_reproj(svgpu_ctx* ctx, const char* who, ReprojProblem& R, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw,
                const double* trans_wc, int n, const double* pos_w, const double* mean_normal, const float* min_valid_dist,
                const float* max_valid_dist, float ray_cos_thr, int num_levels, float log_scale_factor) {
    if (!ctx || !cam || n < 0 || cam->model < SVGPU_CAM_PERSPECTIVE || cam->model > SVGPU_CAM_RADIAL_DIVISION || !rot_cw || !trans_cw || !trans_wc
        || num_levels < 1 || num_levels > SV_MAX_LEVELS || !(log_scale_factor > 0.f)
        || (n > 0 && (!pos_w || !mean_normal || !min_valid_dist || !max_valid_dist)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    R.cam = *cam;
    std::memcpy(R.rot_cw, rot_cw, sizeof R.rot_cw);
    std::memcpy(R.trans_cw, trans_cw, sizeof R.trans_cw);
    std::memcpy(R.trans_wc, trans_wc, sizeof R.trans_wc);
    R.n = n;
    R.ray_cos_thr = ray_cos_thr;
    R.num_levels = (unsigned)num_levels;
    R.log_scale_factor = log_scale_factor;
    return SVGPU_OK;
}
}  // namespace

extern "C" {

int svgpu_reproject_landmarks(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw,
                              const double* trans_wc, int n, const double* pos_w, const double* mean_normal,
                              const float* min_valid_dist, const float* max_valid_dist, const uint8_t* skip, float ray_cos_thr,
                              int num_levels, float log_scale_factor, uint8_t* visible, double* reproj, float* x_right,
                              int32_t* pred_scale_level) {
    ReprojProblem R{};
    int rc = fill_reproj(ctx, "svgpu_reproject_landmarks: bad arguments", R, cam, rot_cw, trans_cw, trans_wc, n, pos_w, mean_normal, min_valid_dist,
                         max_valid_dist, ray_cos_thr, num_levels, log_scale_factor);
    if (rc) return rc;
    if (n == 0) return SVGPU_OK;
    if (!visible || !reproj || !x_right || !pred_scale_level) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_reproject_landmarks: null output");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t need = 2 * pad((size_t)n * 24) + 4 * pad((size_t)n * 4) + 2 * pad(n) + pad((size_t)n * 16) + 1024;
    rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    double* d_pw = A.take<double>((size_t)n * 3);
    double* d_nv = A.take<double>((size_t)n * 3);
    float* d_mn = A.take<float>(n);
    float* d_mx = A.take<float>(n);
    uint8_t* d_skip = skip ? A.take<uint8_t>(n) : nullptr;
    R.visible = A.take<uint8_t>(n);
    R.reproj = A.take<double>((size_t)n * 2);
    R.x_right = A.take<float>(n);
    R.pred_level = A.take<int32_t>(n);
    // one upload and one read-back through the page-locked mirror of the arena (Arena::upload / Downloads): nine separate copies from / to
    // pageable memory were nine staging kernels on the stream
    if ((rc = sv_ensure_stage(ctx, need))) return rc;
    A.mirror = ctx->h_stage;
    if ((rc = A.upload(ctx, s, d_pw, pos_w, (size_t)n * 24))) return rc;
    if ((rc = A.upload(ctx, s, d_nv, mean_normal, (size_t)n * 24))) return rc;
    if ((rc = A.upload(ctx, s, d_mn, min_valid_dist, (size_t)n * 4))) return rc;
    if ((rc = A.upload(ctx, s, d_mx, max_valid_dist, (size_t)n * 4))) return rc;
    if (skip && (rc = A.upload(ctx, s, d_skip, skip, n))) return rc;
    if ((rc = A.flush(ctx, s))) return rc;
    R.pos_w = d_pw, R.mean_normal = d_nv, R.min_valid_dist = d_mn, R.max_valid_dist = d_mx, R.skip = d_skip;
    sv_launch_reproject(s, R);
    SV_HIP(ctx, hipGetLastError());
    Downloads D;
    D.add(A, visible, R.visible, n);
    D.add(A, reproj, R.reproj, (size_t)n * 16);
    D.add(A, x_right, R.x_right, (size_t)n * 4);
    D.add(A, pred_scale_level, R.pred_level, (size_t)n * 4);
    if ((rc = D.fetch(ctx, s, A))) return rc;
    SV_HIP(ctx, hipStreamSynchronize(s));
    D.scatter(A);
    return SVGPU_OK;
}

int svgpu_match_frame_and_landmarks(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw,
                                    const double* trans_wc, int n, const double* pos_w, const double* mean_normal,
                                    const float* min_valid_dist, const float* max_valid_dist, const uint8_t* skip,
                                    const uint8_t* lm_desc, float ray_cos_thr, int num_levels, const float* scale_factors,
                                    float log_scale_factor, float margin, const uint8_t* tdesc, const float* t_xy,
                                    const int32_t* t_octave, int nt, const uint8_t* occupied, const float* t_xright,
                                    int grid_cols, int grid_rows, unsigned thr, float lowe_ratio, int32_t* match_lm,
                                    int* num_matches, uint8_t* visible, double* reproj, float* x_right,
                                    int32_t* pred_scale_level) {
    ReprojProblem R{};
    const svgpu_frame* rf = sv_take_bound_frame(ctx);
    if (rf) {  // resident keypoint side (svgpu_frame_bind)
        tdesc = rf->desc, t_xy = rf->xy, t_octave = rf->octave, nt = rf->n, grid_cols = rf->grid_cols, grid_rows = rf->grid_rows;
        t_xright = rf->has_xright ? rf->xright : nullptr;
    }
    int rc = fill_reproj(ctx, "svgpu_match_frame_and_landmarks: bad arguments", R, cam, rot_cw, trans_cw, trans_wc, n, pos_w, mean_normal,
                         min_valid_dist, max_valid_dist, ray_cos_thr, num_levels, log_scale_factor);
    if (rc) return rc;
    if (nt < 0 || nt >= (1 << 22) || !num_matches || !scale_factors || grid_cols < 1 || grid_rows < 1 || (size_t)grid_cols * grid_rows > (size_t(1) << 22)
        || !(cam->min_x < cam->max_x) || !(cam->min_y < cam->max_y) || (n > 0 && (!lm_desc || !match_lm)) || (nt > 0 && (!tdesc || !t_xy || !t_octave)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_frame_and_landmarks: bad arguments");
    *num_matches = 0;
    if (n == 0) return SVGPU_OK;
    for (int i = 0; i < n; ++i) match_lm[i] = -1;
    if (nt == 0)  // no keypoints: only the observability outputs are meaningful
        return (visible && reproj && x_right && pred_scale_level)
                   ? svgpu_reproject_landmarks(ctx, cam, rot_cw, trans_cw, trans_wc, n, pos_w, mean_normal, min_valid_dist, max_valid_dist, skip,
                                               ray_cos_thr, num_levels, log_scale_factor, visible, reproj, x_right, pred_scale_level)
                   : SVGPU_OK;
    R.margin = margin;
    for (int l = 0; l < num_levels; ++l) R.scale_factors[l] = scale_factors[l];
    InCellsFrame F{tdesc, t_xy, t_octave, nt, occupied, nullptr, t_xright, cam->min_x, cam->max_x, cam->min_y, cam->max_y, grid_cols, grid_rows};
    if (rf) F.min_x = rf->min_x, F.max_x = rf->max_x, F.min_y = rf->min_y, F.max_y = rf->max_y, F.res = rf;
    const size_t qbytes = pad((size_t)n * 32) + 2 * pad((size_t)n * 24) + 8 * pad((size_t)n * 4) + 2 * pad(n) + pad((size_t)n * 16) + pad((size_t)n * 8);
    hipStream_t s = ctx->stream;
    return in_cells_core(
        ctx, n, F, qbytes, 0, thr, lowe_ratio, SVGPU_MATCH_RATIO_SAME_OCTAVE,
        [&](Arena& A, bool fresh, CandProblem& P, GridProblem& G) -> int {
            uint8_t* d_q = A.take<uint8_t>((size_t)n * 32);
            double* d_pw = A.take<double>((size_t)n * 3);
            double* d_nv = A.take<double>((size_t)n * 3);
            float* d_mn = A.take<float>(n);
            float* d_mx = A.take<float>(n);
            uint8_t* d_skip = skip ? A.take<uint8_t>(n) : nullptr;
            R.visible = A.take<uint8_t>(n);
            R.reproj = A.take<double>((size_t)n * 2);
            R.x_right = A.take<float>(n);
            R.pred_level = A.take<int32_t>(n);
            R.q_xy = A.take<float>((size_t)n * 2);
            R.q_margin = A.take<float>(n);
            R.q_min_level = A.take<int32_t>(n);
            R.q_max_level = A.take<int32_t>(n);
            R.pos_w = d_pw, R.mean_normal = d_nv, R.min_valid_dist = d_mn, R.max_valid_dist = d_mx, R.skip = d_skip;
            if (fresh) {  // one batched upload (Arena::upload / flush), then the reprojection that consumes it
                int ru = A.upload(ctx, s, d_q, lm_desc, (size_t)n * 32);
                if (!ru) ru = A.upload(ctx, s, d_pw, pos_w, (size_t)n * 24);
                if (!ru) ru = A.upload(ctx, s, d_nv, mean_normal, (size_t)n * 24);
                if (!ru) ru = A.upload(ctx, s, d_mn, min_valid_dist, (size_t)n * 4);
                if (!ru) ru = A.upload(ctx, s, d_mx, max_valid_dist, (size_t)n * 4);
                if (!ru && skip) ru = A.upload(ctx, s, d_skip, skip, n);
                if (!ru) ru = A.flush(ctx, s);
                if (ru) return ru;
                sv_launch_reproject(s, R);
            }
            G.q_xy = R.q_xy;
            G.q_margin = R.q_margin;
            G.q_min_level = R.q_min_level;
            G.q_max_level = R.q_max_level;
            G.q_valid = R.visible;
            P.qdesc = (const uint32_t*)d_q;
            P.q_valid = R.visible;
            if (t_xright) {  // projection.cc:57-62: |lm_to_x_right - stereo_x_right| against margin * scale
                P.q_xright = R.x_right;
                P.q_xr_tol = R.q_margin;
            }
            return SVGPU_OK;
        },
        [&](const CandProblem&, const Arena& A, Downloads& D) -> int {
            D.add(A, visible, R.visible, n);
            D.add(A, reproj, R.reproj, (size_t)n * 16);
            D.add(A, x_right, R.x_right, (size_t)n * 4);
            D.add(A, pred_scale_level, R.pred_level, (size_t)n * 4);
            return SVGPU_OK;
        },
        match_lm, num_matches);
}

int svgpu_stereo_match(svgpu_ctx* ctx_left, svgpu_ctx* ctx_right, const svgpu_keypoint* kps_left, const uint8_t* desc_left,
                       int n_left, const svgpu_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                       float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths) {
    svgpu_ctx* ctx = ctx_left;
    if (!ctx_left || !ctx_right || n_left < 0 || n_right < 0 || n_right > 65535)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match: bad arguments");
    if (n_left == 0) return SVGPU_OK;
    if (!kps_left || !desc_left || !stereo_x_right || !depths || (n_right > 0 && (!kps_right || !desc_right)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match: null pointer");
    const OrbConfig& CL = ctx_left->orb;
    const OrbConfig& CR = ctx_right->orb;
    if (!CL.configured || !CR.configured || ctx_left->last_batch == 0 || ctx_right->last_batch == 0)
        return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_stereo_match: both contexts need a previous extract call");
    if (ctx_left->device != ctx_right->device || CL.width != CR.width || CL.height != CR.height || CL.num_levels != CR.num_levels
        || CL.scale_factor != CR.scale_factor)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match: the two extractors must share device, geometry and ORB parameters");
    for (int i = 0; i < n_left; ++i) {
        stereo_x_right[i] = -1.0f;
        depths[i] = -1.0f;
        if (kps_left[i].octave < 0 || kps_left[i].octave >= CL.num_levels) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match: octave out of range");
    }
    for (int i = 0; i < n_right; ++i)
        if (kps_right[i].octave < 0 || kps_right[i].octave >= CL.num_levels) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match: octave out of range");
    if (n_right == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipStreamSynchronize(ctx_right->stream));  // the right pyramid must be complete
    const int rows = CL.levels[0].h, rows_per_kp = 2 * (int)std::ceil(2.0 * std::pow((double)CL.scale_factor, CL.num_levels - 1)) + 3;  // rows of the widest band, one to spare
    const size_t need = pad((size_t)n_left * 28) + pad((size_t)n_right * 28) + pad((size_t)n_left * 32) + pad((size_t)n_right * 32)
                        + 3 * pad((size_t)n_left * 4) + sv_stereo_rows_bytes(1, rows, n_right, rows_per_kp) + 256;
    int rc = sv_ensure_scratch(ctx, need);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    StereoProblem P{};
    svgpu_keypoint* dkl = A.take<svgpu_keypoint>(n_left);
    svgpu_keypoint* dkr = A.take<svgpu_keypoint>(n_right);
    uint32_t* ddl = A.take<uint32_t>((size_t)n_left * 8);
    uint32_t* ddr = A.take<uint32_t>((size_t)n_right * 8);
    P.xr = A.take<float>(n_left);
    P.depth = A.take<float>(n_left);
    P.corr = A.take<float>(n_left);
    P.rows = rows, P.rows_per_kp = rows_per_kp;
    P.row_off = A.take<int32_t>((size_t)rows + 1);
    P.row_fill = A.take<int32_t>(rows);
    P.row_items = A.take<int32_t>((size_t)n_right * rows_per_kp);
    hipStream_t s = ctx->stream;
    SV_HIP(ctx, hipMemcpyAsync(dkl, kps_left, (size_t)n_left * 28, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(dkr, kps_right, (size_t)n_right * 28, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(ddl, desc_left, (size_t)n_left * 32, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(ddr, desc_right, (size_t)n_right * 32, hipMemcpyHostToDevice, s));
    P.kl = dkl;
    P.kr = dkr;
    P.dl = ddl;
    P.dr = ddr;
    P.nl = n_left;
    P.nr = n_right;
    P.num_levels = CL.num_levels;
    float inv[SV_MAX_LEVELS];
    svgpu_orb_scale_tables(CL.scale_factor, CL.num_levels, P.sf, inv, nullptr, nullptr);
    for (int l = 0; l < CL.num_levels; ++l) {
        P.isf[l] = inv[l];
        P.w[l] = CL.levels[l].w;
        P.h[l] = CL.levels[l].h;
        if (l == 0) {
            P.lev_l[0] = ctx_left->last_imgs;
            P.pitch_l[0] = ctx_left->last_row_stride;
            P.lev_r[0] = ctx_right->last_imgs;
            P.pitch_r[0] = ctx_right->last_row_stride;
        }
        else {
            P.lev_l[l] = ctx_left->d_pyr + CL.levels[l].pyr_off;
            P.pitch_l[l] = CL.levels[l].pitch;
            P.lev_r[l] = ctx_right->d_pyr + CR.levels[l].pyr_off;
            P.pitch_r[l] = CR.levels[l].pitch;
        }
    }
    P.fxb = focal_x_baseline;
    P.min_disp = 0.0f;                                // stereo.cc:18
    P.max_disp = focal_x_baseline / true_baseline;    // stereo.cc:18
    P.thr = 75;                                       // (HAMMING_DIST_THR_HIGH + HAMMING_DIST_THR_LOW) / 2, stereo.h:99
    sv_launch_stereo(ctx, s, P);
    sv_launch_stereo_median(s, P, 1);  // median filter of the correlations (stereo.cc:94-113) on the device
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(stereo_x_right, P.xr, (size_t)n_left * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(depths, P.depth, (size_t)n_left * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_stereo_match_batch_device(svgpu_ctx* ctx_left, svgpu_ctx* ctx_right, int pairs, const svgpu_keypoint* kps_left_dev, const uint8_t* desc_left_dev,
                                    const int32_t* n_left_dev, const svgpu_keypoint* kps_right_dev, const uint8_t* desc_right_dev,
                                    const int32_t* n_right_dev, int cap, int n_stride, float focal_x_baseline, float true_baseline,
                                    float* stereo_x_right_dev, float* depths_dev, void* stream) {
    svgpu_ctx* ctx = ctx_left;
    if (!ctx_left || !ctx_right || pairs < 0 || cap < 0 || cap > 65535 || n_stride < 1 || !(true_baseline > 0.f)
        || (pairs > 0 && cap > 0 && (!kps_left_dev || !desc_left_dev || !n_left_dev || !kps_right_dev || !desc_right_dev || !n_right_dev || !stereo_x_right_dev || !depths_dev)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match_batch_device: bad arguments");
    if (pairs == 0 || cap == 0) return SVGPU_OK;
    const OrbConfig& CL = ctx_left->orb;
    const OrbConfig& CR = ctx_right->orb;
    if (!CL.configured || !CR.configured || ctx_left->last_batch < pairs || ctx_right->last_batch < pairs)
        return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_stereo_match_batch_device: both contexts need a previous batch extraction of >= `pairs` frames");
    if (ctx_left->device != ctx_right->device || CL.width != CR.width || CL.height != CR.height || CL.num_levels != CR.num_levels
        || CL.scale_factor != CR.scale_factor)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_stereo_match_batch_device: the two extractors must share device, geometry and ORB parameters");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int rows = CL.levels[0].h, rows_per_kp = 2 * (int)std::ceil(2.0 * std::pow((double)CL.scale_factor, CL.num_levels - 1)) + 3;  // rows of the widest band, one to spare
    int rc = sv_ensure_scratch(ctx, pad((size_t)pairs * cap * 4) + sv_stereo_rows_bytes(pairs, rows, cap, rows_per_kp) + 256);
    if (rc) return rc;
    Arena A(ctx->d_scratch);
    StereoProblem P{};
    P.rows = rows, P.rows_per_kp = rows_per_kp;
    P.row_off = A.take<int32_t>((size_t)pairs * (rows + 1));
    P.row_fill = A.take<int32_t>((size_t)pairs * rows);
    P.row_items = A.take<int32_t>((size_t)pairs * cap * rows_per_kp);
    P.kl = kps_left_dev;
    P.kr = kps_right_dev;
    P.dl = (const uint32_t*)desc_left_dev;
    P.dr = (const uint32_t*)desc_right_dev;
    P.nl_dev = n_left_dev;
    P.nr_dev = n_right_dev;
    P.n_stride = n_stride;
    P.cap = cap;
    P.xr = stereo_x_right_dev;
    P.depth = depths_dev;
    P.corr = A.take<float>((size_t)pairs * cap);
    P.num_levels = CL.num_levels;
    float inv[SV_MAX_LEVELS];
    svgpu_orb_scale_tables(CL.scale_factor, CL.num_levels, P.sf, inv, nullptr, nullptr);
    for (int l = 0; l < CL.num_levels; ++l) {
        P.isf[l] = inv[l];
        P.w[l] = CL.levels[l].w;
        P.h[l] = CL.levels[l].h;
        if (l == 0) {
            P.lev_l[0] = ctx_left->last_imgs;
            P.pitch_l[0] = ctx_left->last_row_stride;
            P.lev_r[0] = ctx_right->last_imgs;
            P.pitch_r[0] = ctx_right->last_row_stride;
        }
        else {
            P.lev_l[l] = ctx_left->d_pyr + CL.levels[l].pyr_off;
            P.pitch_l[l] = CL.levels[l].pitch;
            P.lev_r[l] = ctx_right->d_pyr + CR.levels[l].pyr_off;
            P.pitch_r[l] = CR.levels[l].pitch;
        }
    }
    P.img_stride_l = ctx_left->last_frame_stride;
    P.img_stride_r = ctx_right->last_frame_stride;
    P.pyr_stride_l = CL.pyr_frame_bytes;
    P.pyr_stride_r = CR.pyr_frame_bytes;
    P.fxb = focal_x_baseline;
    P.min_disp = 0.0f;
    P.max_disp = focal_x_baseline / true_baseline;
    P.thr = 75;
    sv_launch_stereo(ctx, s, P, pairs);
    sv_launch_stereo_median(s, P, pairs);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}

}  // extern "C"
