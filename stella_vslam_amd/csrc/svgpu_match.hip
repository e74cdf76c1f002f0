// Host side of the matchers: staging of host buffers into the context's scratch arena, launches.
#include <algorithm>
#include <utility>

#include "svgpu_internal.h"
#include "match_kernels.h"

namespace {

// bump allocator over ctx->d_scratch (256-byte aligned pieces)
struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(void* p) : base((char*)p) {}
    template <class T>
    T* take(size_t n) {
        T* r = (T*)(base + off);
        off += (n * sizeof(T) + 255) & ~size_t(255);
        return r;
    }
};
inline size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// scratch for the angle-bin sorted copies of both sides (see k_bf_binsort)
inline size_t sort_bytes(int pairs, int cap1, int cap2) {
    const size_t p = (size_t)pairs;
    return pad(p * cap1 * 32) + pad(p * cap2 * 32) + 2 * pad(p * cap1 * 4) + 2 * pad(p * cap2 * 4) + 2 * pad(p * 362 * 4) + pad(p * 2 * 4);
}
inline void take_sort(Arena& A, BfProblem& P, int pairs, int cap1, int cap2) {
    const size_t p = (size_t)pairs;
    P.sd1 = A.take<uint32_t>(p * cap1 * 8);
    P.sd2 = A.take<uint32_t>(p * cap2 * 8);
    P.sa1 = A.take<float>(p * cap1);
    P.si1 = A.take<int>(p * cap1);
    P.sa2 = A.take<float>(p * cap2);
    P.si2 = A.take<int>(p * cap2);
    P.bs1 = A.take<int>(p * 362);
    P.bs2 = A.take<int>(p * 362);
    P.prune_ok = A.take<int>(p * 2);
}

}  // namespace

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

int svgpu_match_bruteforce_batch_device(svgpu_ctx* ctx, int pairs, const uint8_t* desc1_dev,
                                        const svgpu_keypoint* kps1_dev, const int32_t* n1_dev, int cap1,
                                        const uint8_t* desc2_dev, const svgpu_keypoint* kps2_dev,
                                        const int32_t* n2_dev, int cap2, int n_stride, const uint8_t* valid2_dev,
                                        float lowe_ratio, int check_orientation, int32_t* matched_dev,
                                        int32_t* num_dev, void* stream) {
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
    P.angle1 = &kps1_dev->angle;
    P.angle2 = &kps2_dev->angle;
    P.angle_stride = sizeof(svgpu_keypoint) / sizeof(float);
    P.n1_dev = n1_dev;
    P.n2_dev = n2_dev;
    P.n_stride = n_stride;
    P.cap1 = cap1;
    P.cap2 = cap2;
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
    SV_HIP(ctx, hipMemcpyAsync(d1, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d2, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    if (angle1) SV_HIP(ctx, hipMemcpyAsync(a1, angle1, (size_t)n1 * 4, hipMemcpyHostToDevice, s));
    else SV_HIP(ctx, hipMemsetAsync(a1, 0, (size_t)n1 * 4, s));
    if (angle2) SV_HIP(ctx, hipMemcpyAsync(a2, angle2, (size_t)n2 * 4, hipMemcpyHostToDevice, s));
    else SV_HIP(ctx, hipMemsetAsync(a2, 0, (size_t)n2 * 4, s));
    if (valid2) SV_HIP(ctx, hipMemcpyAsync(v2, valid2, n2, hipMemcpyHostToDevice, s));
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
    SV_HIP(ctx, hipMemcpyAsync(matched_2_in_1, P.matched, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(&num, P.num, 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
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
    CandProblem P{};
#define UP(dst, T, src, n)                                                                          \
    T* dst = nullptr;                                                                               \
    if (src) {                                                                                      \
        dst = A.take<T>(n);                                                                         \
        SV_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)(n) * sizeof(T), hipMemcpyHostToDevice, s)); \
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
    SV_HIP(ctx, hipMemcpyAsync(match_q, P.match_q, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(&num, P.num, 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    *num_matches = num;
    return SVGPU_OK;
}

int svgpu_match_in_cells(svgpu_ctx* ctx, const uint8_t* qdesc, int nq, const float* q_xy, const float* q_margin,
                         const int32_t* q_min_level, const int32_t* q_max_level, const uint8_t* q_valid, const float* q_angle,
                         const float* q_xright, const float* q_xr_tol, const uint8_t* tdesc, const float* t_xy,
                         const int32_t* t_octave, int nt, const uint8_t* occupied, const float* t_angle, const float* t_xright,
                         float min_x, float max_x, float min_y, float max_y, int grid_cols, int grid_rows,
                         int check_orientation, unsigned thr, float lowe_ratio, int mode, int32_t* match_q, int* num_matches) {
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
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int ncell = grid_cols * grid_rows;
    const size_t need1 = pad((size_t)nq * 32) + pad((size_t)nt * 32) + pad((size_t)nt * 8) + 4 * pad((size_t)nt * 4) + pad(nt) + pad((size_t)nq * 8)
                         + 6 * pad((size_t)nq * 4) + pad(nq) + pad((size_t)(ncell + 1) * 4) + pad((size_t)(nq + 1) * 4) + 2 * pad((size_t)nq * 4)
                         + 2 * pad((size_t)nt * 4) + 1024;
    int total = 0;
    // Pass 0 builds the grid and the list sizes and reads the total back; the scratch arena may then have to grow for the
    // lists, which discards its contents, so pass 1 repeats the (cheap) uploads and the grid build in the final arena.
    for (int pass = 0; pass < 2; ++pass) {
        const size_t need = need1 + (pass ? 2 * pad((size_t)total * 4) : 0);
        const bool regrow = need > ctx->scratch_bytes;  // pass 1 without regrowth: the arena of pass 0 is still valid, same layout
        int rc = sv_ensure_scratch(ctx, need);
        if (rc) return rc;
        Arena A(ctx->d_scratch);
        CandProblem P{};
        GridProblem G{};
#define UP(dst, T, src, n)                                                                          \
    T* dst = nullptr;                                                                               \
    if (src) {                                                                                      \
        dst = A.take<T>(n);                                                                         \
        if (!pass || regrow) SV_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)(n) * sizeof(T), hipMemcpyHostToDevice, s)); \
    }
        UP(d_q, uint8_t, qdesc, (size_t)nq * 32)
        UP(d_t, uint8_t, tdesc, (size_t)nt * 32)
        UP(d_txy, float, t_xy, (size_t)nt * 2)
        UP(d_toct, int32_t, t_octave, nt)
        UP(d_occ, uint8_t, occupied, nt)
        UP(d_ta, float, t_angle, nt)
        UP(d_tx, float, t_xright, nt)
        UP(d_qxy, float, q_xy, (size_t)nq * 2)
        UP(d_qm, float, q_margin, nq)
        UP(d_qlo, int32_t, q_min_level, nq)
        UP(d_qhi, int32_t, q_max_level, nq)
        UP(d_qv, uint8_t, q_valid, nq)
        UP(d_qa, float, q_angle, nq)
        UP(d_qx, float, q_xright, nq)
        UP(d_qtol, float, q_xr_tol, nq)
#undef UP
        G.t_xy = d_txy;
        G.t_octave = d_toct;
        G.nt = nt;
        G.min_x = min_x;
        G.min_y = min_y;
        G.inv_w = (double)grid_cols / (max_x - min_x);  // float difference, double quotient: data/common.cc:86-87 via camera::base
        G.inv_h = (double)grid_rows / (max_y - min_y);
        G.cols = grid_cols;
        G.rows = grid_rows;
        G.cell_of = A.take<int32_t>(nt);
        G.cell_off = A.take<int32_t>(ncell + 1);
        G.cell_items = A.take<int32_t>(nt);
        G.q_xy = d_qxy;
        G.q_margin = d_qm;
        G.q_min_level = d_qlo;
        G.q_max_level = d_qhi;
        G.q_valid = d_qv;
        G.nq = nq;
        G.cand_off = A.take<int32_t>(nq + 1);
        P.match_q = A.take<int32_t>(nq);
        P.num = A.take<int32_t>(1);
        int* owner = A.take<int>(nt);
        int* match = A.take<int>(nq);
        unsigned* mdist = A.take<unsigned>(nt);
        if (!pass || regrow) sv_launch_grid_build(s, G);
        if (!pass) {
            SV_HIP(ctx, hipMemcpyAsync(&total, G.cand_off + nq, 4, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
            if (total == 0) return SVGPU_OK;
            continue;
        }
        G.cand_idx = A.take<int32_t>(total);
        P.dist = A.take<uint32_t>(total);
        sv_launch_grid_fill(s, G);
        P.qdesc = (const uint32_t*)d_q;
        P.tdesc = (const uint32_t*)d_t;
        P.t_octave = d_toct;
        P.nq = nq;
        P.nt = nt;
        P.cand_off = G.cand_off;
        P.cand_idx = G.cand_idx;
        P.cand_skip = nullptr;
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
        sv_launch_cand(ctx, s, P, owner, match, mdist);
        SV_HIP(ctx, hipGetLastError());
        int32_t num = 0;
        SV_HIP(ctx, hipMemcpyAsync(match_q, P.match_q, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipMemcpyAsync(&num, P.num, 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        *num_matches = num;
    }
    return SVGPU_OK;
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
    const size_t need = pad((size_t)n_left * 28) + pad((size_t)n_right * 28) + pad((size_t)n_left * 32) + pad((size_t)n_right * 32)
                        + 3 * pad((size_t)n_left * 4) + 256;
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
    SV_HIP(ctx, hipGetLastError());
    std::vector<float> corr(n_left);
    SV_HIP(ctx, hipMemcpyAsync(stereo_x_right, P.xr, (size_t)n_left * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(depths, P.depth, (size_t)n_left * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(corr.data(), P.corr, (size_t)n_left * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    // median filter of the correlations (stereo.cc:94-113); std::pair<int, int> as in the reference
    std::vector<std::pair<int, int>> correlation_and_idx_left;
    for (int i = 0; i < n_left; ++i)
        if (stereo_x_right[i] != -1.0f || depths[i] != -1.0f) correlation_and_idx_left.emplace_back((int)corr[i], i);
    std::sort(correlation_and_idx_left.begin(), correlation_and_idx_left.end());
    const size_t median_i = correlation_and_idx_left.size() / 2;
    const float median_correlation = correlation_and_idx_left.empty() ? 0.0f : (float)correlation_and_idx_left[median_i].first;
    const float correlation_thr = (float)(2.0 * median_correlation);
    for (size_t i = median_i; i < correlation_and_idx_left.size(); ++i)
        if (correlation_thr < (float)correlation_and_idx_left[i].first) {
            stereo_x_right[correlation_and_idx_left[i].second] = -1;
            depths[correlation_and_idx_left[i].second] = -1;
        }
    return SVGPU_OK;
}

}  // extern "C"
