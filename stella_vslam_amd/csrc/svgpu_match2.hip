// Host side of the function-specific matchers (every match::* method with its own candidate source and gates):
//   projection::match_current_and_last_frames / match_frame_and_keyframe / match_by_Sim3_transform / match_keyframes_mutually
//   fuse::detect_duplication
//   robust::match_for_triangulation, bow_tree::match_for_triangulation / match_frame_and_keyframe / match_keyframes
// Reprojection, grid lookup, pair gates, Hamming distances and the exact greedy replay all run on the device; the host only
// stages the flat arrays and derives the handful of 3x3 quantities each method computes once per call.
#include "svgpu_match_common.h"

using namespace svm;

namespace {

// keypoint side of a projection-family call: the caller's host arrays, or the resident frame bound with svgpu_frame_bind
InCellsFrame frame_side(const svgpu_frame* rf, const svgpu_camera* cam, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, int nt,
                        const uint8_t* occupied, const float* t_angle, const float* t_xright, bool want_angle, bool want_xright, int grid_cols, int grid_rows) {
    if (!rf) return InCellsFrame{tdesc, t_xy, t_octave, nt, occupied, t_angle, t_xright, cam->min_x, cam->max_x, cam->min_y, cam->max_y, grid_cols, grid_rows};
    InCellsFrame F{rf->desc, rf->xy, rf->octave, rf->n, occupied, want_angle ? rf->angle : nullptr, want_xright && rf->has_xright ? rf->xright : nullptr,
                   rf->min_x, rf->max_x, rf->min_y, rf->max_y, rf->grid_cols, rf->grid_rows};
    F.res = rf;
    return F;
}

struct ProjQueries {  // host pointers: the landmarks that get reprojected (= the queries of the cell matcher)
    int n;
    const double* pos_w;           // n x 3
    const double* mean_normal;     // n x 3, nullable with normal_mode 2
    const float* min_valid_dist;   // nullable with q_level
    const float* max_valid_dist;
    const uint8_t* valid;          // nullable: 0 = landmark not offered (null / will_be_erased / already matched ...)
    const int32_t* q_level;        // nullable
    const uint8_t* desc;           // n x 32
    const float* q_angle;          // nullable: keypoint angle on the query side (orientation gate)
    const uint8_t* q_blocks;       // nullable
};
struct ProjOpts {
    int dist_mode, normal_mode, center_mode, window_mode;
    float margin;
    int check_orientation;
    unsigned thr;
    float lowe_ratio;
    int mode;
    int stereo_gate;  // |x_right - stereo_x_right| <= margin * scale for stereo keypoints (projection.cc:57-62, 172-177)
    int chi_gate;     // fuse.cc:92-119
    int no_claims;
    const float* inv_level_sigma_sq;
};

// landmark reprojection -> query windows -> device-built candidate lists -> candidate matcher; match_q[i] = keypoint or -1
int proj_match(svgpu_ctx* ctx, const char* who, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, const double* trans_wc,
               const ProjQueries& Q, int num_levels, const float* scale_factors, float log_scale_factor, const InCellsFrame& F, const ProjOpts& O,
               int32_t* match_q, int* num_matches, uint8_t* visible, double* reproj, float* x_right, int32_t* pred_level) {
    const int n = Q.n, nt = F.nt;
    if (!ctx || !cam || n < 0 || cam->model < SVGPU_CAM_PERSPECTIVE || cam->model > SVGPU_CAM_RADIAL_DIVISION || !rot_cw || !trans_cw || !trans_wc
        || num_levels < 1 || num_levels > SV_MAX_LEVELS || !scale_factors || (!Q.q_level && !(log_scale_factor > 0.f)) || !num_matches
        || nt < 0 || nt >= (1 << 22) || F.grid_cols < 1 || F.grid_rows < 1 || (size_t)F.grid_cols * F.grid_rows > (size_t(1) << 22)
        || !(F.min_x < F.max_x) || !(F.min_y < F.max_y)
        || (n > 0 && (!Q.pos_w || !Q.desc || !match_q || (O.normal_mode != 2 && !Q.mean_normal) || (!Q.q_level && (!Q.min_valid_dist || !Q.max_valid_dist))
                      || (O.check_orientation && !Q.q_angle)))
        || (nt > 0 && (!F.tdesc || !F.t_xy || !F.t_octave || (O.check_orientation && !F.t_angle))) || (O.chi_gate && !O.inv_level_sigma_sq))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    *num_matches = 0;
    if (n == 0) return SVGPU_OK;
    for (int i = 0; i < n; ++i) match_q[i] = -1;
    if (nt == 0) return SVGPU_OK;
    ReprojProblem R{};
    R.cam = *cam;
    std::memcpy(R.rot_cw, rot_cw, sizeof R.rot_cw);
    std::memcpy(R.trans_cw, trans_cw, sizeof R.trans_cw);
    std::memcpy(R.trans_wc, trans_wc, sizeof R.trans_wc);
    R.n = n;
    R.ray_cos_thr = 0.5f;
    R.num_levels = (unsigned)num_levels;
    R.log_scale_factor = log_scale_factor;
    R.margin = O.margin;
    for (int l = 0; l < num_levels; ++l) R.scale_factors[l] = scale_factors[l];
    R.dist_mode = O.dist_mode;
    R.normal_mode = O.normal_mode;
    R.center_mode = O.center_mode;
    R.window_mode = O.window_mode;
    std::vector<uint8_t> skip;
    if (Q.valid) {
        skip.resize(n);
        for (int i = 0; i < n; ++i) skip[i] = Q.valid[i] ? 0 : 1;
    }
    InCellsFrame Fq = F;
    if (!O.stereo_gate && !O.chi_gate) Fq.t_xright = nullptr;
    const size_t qbytes = pad((size_t)n * 32) + 2 * pad((size_t)n * 24) + 10 * pad((size_t)n * 4) + 3 * pad(n) + pad((size_t)n * 16) + pad((size_t)n * 8);
    hipStream_t s = ctx->stream;
    return in_cells_core(
        ctx, n, Fq, qbytes, O.check_orientation, O.thr, O.lowe_ratio, O.mode,
        [&](Arena& A, bool fresh, CandProblem& P, GridProblem& G) -> int {
            uint8_t* d_q = A.take<uint8_t>((size_t)n * 32);
            double* d_pw = A.take<double>((size_t)n * 3);
            double* d_nv = Q.mean_normal ? A.take<double>((size_t)n * 3) : nullptr;
            float* d_mn = Q.min_valid_dist ? A.take<float>(n) : nullptr;
            float* d_mx = Q.max_valid_dist ? A.take<float>(n) : nullptr;
            uint8_t* d_skip = Q.valid ? A.take<uint8_t>(n) : nullptr;
            int32_t* d_lvl = Q.q_level ? A.take<int32_t>(n) : nullptr;
            float* d_qa = Q.q_angle ? A.take<float>(n) : nullptr;
            uint8_t* d_qb = Q.q_blocks ? A.take<uint8_t>(n) : nullptr;
            R.visible = A.take<uint8_t>(n);
            R.reproj = A.take<double>((size_t)n * 2);
            R.x_right = A.take<float>(n);
            R.pred_level = A.take<int32_t>(n);
            R.q_xy = A.take<float>((size_t)n * 2);
            R.q_margin = A.take<float>(n);
            R.q_min_level = A.take<int32_t>(n);
            R.q_max_level = A.take<int32_t>(n);
            R.pos_w = d_pw, R.mean_normal = d_nv, R.min_valid_dist = d_mn, R.max_valid_dist = d_mx, R.skip = d_skip, R.q_level = d_lvl;
            if (fresh) {  // one batched upload (Arena::upload / flush), then the reprojection that consumes it
                int ru = A.upload(ctx, s, d_q, Q.desc, (size_t)n * 32);
                if (!ru) ru = A.upload(ctx, s, d_pw, Q.pos_w, (size_t)n * 24);
                if (!ru && d_nv) ru = A.upload(ctx, s, d_nv, Q.mean_normal, (size_t)n * 24);
                if (!ru && d_mn) ru = A.upload(ctx, s, d_mn, Q.min_valid_dist, (size_t)n * 4);
                if (!ru && d_mx) ru = A.upload(ctx, s, d_mx, Q.max_valid_dist, (size_t)n * 4);
                if (!ru && d_skip) ru = A.upload(ctx, s, d_skip, skip.data(), n);
                if (!ru && d_lvl) ru = A.upload(ctx, s, d_lvl, Q.q_level, (size_t)n * 4);
                if (!ru && d_qa) ru = A.upload(ctx, s, d_qa, Q.q_angle, (size_t)n * 4);
                if (!ru && d_qb) ru = A.upload(ctx, s, d_qb, Q.q_blocks, n);
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
            P.q_angle = d_qa;
            P.q_blocks = d_qb;
            P.no_claims = O.no_claims;
            if (O.stereo_gate && F.t_xright) {
                P.q_xright = R.x_right;
                P.q_xr_tol = R.q_margin;
            }
            if (O.chi_gate) {
                P.chi_gate = 1;
                P.q_reproj = R.reproj;
                P.q_reproj_xr = R.x_right;
                for (int l = 0; l < num_levels; ++l) P.inv_level_sigma_sq[l] = O.inv_level_sigma_sq[l];
            }
            return SVGPU_OK;
        },
        [&](const CandProblem&, const Arena& A, Downloads& D) -> int {
            D.add(A, visible, R.visible, n);
            D.add(A, reproj, R.reproj, (size_t)n * 16);
            D.add(A, x_right, R.x_right, (size_t)n * 4);
            D.add(A, pred_level, R.pred_level, (size_t)n * 4);
            return SVGPU_OK;
        },
        match_q, num_matches);
}

// -R^T t and R v + t as the reference writes them (Eigen 3-vectors: ((a0 b0 + a1 b1) + a2 b2); compiled without contraction)
inline void cam_center(const double* R, const double* t, double* c) {
    for (int i = 0; i < 3; ++i) c[i] = ((-R[i]) * t[0] + (-R[3 + i]) * t[1]) + (-R[6 + i]) * t[2];
}
inline void mat_vec(const double* M, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = (M[3 * i] * v[0] + M[3 * i + 1] * v[1]) + M[3 * i + 2] * v[2];
}
inline void mat_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}

// bucketed candidate lists (bucket_kernels.hip) + candidate matcher; match_1to2[i] = side-2 keypoint or -1
struct BucketSide {
    const uint8_t* desc;
    const float* angle;
    const uint8_t* valid;
    const int32_t* node;     // nullable on both sides = one bucket
    const int32_t* octave;   // side 1, triangulation
    const double* bearings;  // triangulation
    const float* xright;     // nullable
    int n;
};
int bucket_match(svgpu_ctx* ctx, const char* who, const BucketSide& S1, const BucketSide& S2, const uint8_t* occupied2, int tri, const double* E12,
                 const double* epipole, int valid_epipole, const float* scale_factors, int num_levels, float residual_rad_thr, unsigned thr,
                 float lowe_ratio, int check_orientation, int mode, int32_t* match_1to2, int* num_matches) {
    const int n1 = S1.n, n2 = S2.n;
    if (!ctx || n1 < 0 || n2 < 0 || n2 >= (1 << 22) || !num_matches || (n1 > 0 && (!S1.desc || !match_1to2)) || (n2 > 0 && !S2.desc)
        || ((S1.node == nullptr) != (S2.node == nullptr)) || (check_orientation && ((n1 > 0 && !S1.angle) || (n2 > 0 && !S2.angle)))
        || (tri && (!E12 || !epipole || !scale_factors || num_levels < 1 || num_levels > 16 || (n1 > 0 && (!S1.bearings || !S1.octave)) || (n2 > 0 && !S2.bearings))))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    *num_matches = 0;
    for (int i = 0; i < n1; ++i) match_1to2[i] = -1;
    if (n1 == 0 || n2 == 0) return SVGPU_OK;
    if (tri)
        for (int i = 0; i < n1; ++i)
            if (S1.octave[i] < 0 || S1.octave[i] >= num_levels) return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t sortb = std::max(sv_bucket_sort_bytes(n1), sv_bucket_sort_bytes(n2));
    const size_t need1 = 2 * pad((size_t)n1 * 32) + pad((size_t)n2 * 32) + 3 * pad((size_t)n1 * 4) + 2 * pad((size_t)n2 * 4) + 2 * pad(n1) + 3 * pad(n2)
                         + 2 * pad((size_t)n1 * 24) + 2 * pad((size_t)n2 * 24) + 16 * pad((size_t)n1 * 4) + 8 * pad((size_t)n2 * 4) + pad(sortb) + 4096;
    int total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const size_t need = need1 + (pass ? 2 * pad((size_t)total * 4) : 0);
        const bool regrow = need > ctx->scratch_bytes;
        int rc = sv_ensure_scratch(ctx, need);
        if (rc) return rc;
        const bool fresh = !pass || regrow;
        Arena A(ctx->d_scratch);
        BucketProblem B{};
#define UP(dst, T, src, cnt)                                                                          \
    T* dst = nullptr;                                                                                 \
    if (src) {                                                                                        \
        dst = A.take<T>(cnt);                                                                         \
        if (fresh) SV_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)(cnt) * sizeof(T), hipMemcpyHostToDevice, s)); \
    }
        UP(d_d1, uint8_t, S1.desc, (size_t)n1 * 32)
        UP(d_d2, uint8_t, S2.desc, (size_t)n2 * 32)
        UP(d_a1, float, S1.angle, n1)
        UP(d_a2, float, S2.angle, n2)
        UP(d_v1, uint8_t, S1.valid, n1)
        UP(d_v2, uint8_t, S2.valid, n2)
        UP(d_n1, int32_t, S1.node, n1)
        UP(d_n2, int32_t, S2.node, n2)
        UP(d_o1, int32_t, S1.octave, n1)
        UP(d_b1, double, S1.bearings, (size_t)n1 * 3)
        UP(d_b2, double, S2.bearings, (size_t)n2 * 3)
        UP(d_x1, float, S1.xright, n1)
        UP(d_x2, float, S2.xright, n2)
        UP(d_occ, uint8_t, occupied2, n2)
#undef UP
        unsigned* key1 = A.take<unsigned>(n1);
        int* q_idx = A.take<int>(n1);
        unsigned* key2 = A.take<unsigned>(n2);
        int* t_sorted = A.take<int>(n2);
        B.row_lo = A.take<int>(n1);
        B.row_hi = A.take<int>(n1);
        B.q_valid = A.take<uint8_t>(n1);
        B.cand_off = A.take<int32_t>(n1 + 1);
        uint32_t* qdesc_rows = A.take<uint32_t>((size_t)n1 * 8);
        float* qangle_rows = A.take<float>(n1);
        int32_t* match_rows = A.take<int32_t>(n1);
        int32_t* d_match = A.take<int32_t>(n1);
        int32_t* d_num = A.take<int32_t>(1);
        int* owner = A.take<int>(n2);
        int* match = A.take<int>(n1);
        unsigned* mdist = A.take<unsigned>(n2);
        void* sort_scratch = A.take<char>(sortb);
        B.n1 = n1;
        B.n2 = n2;
        B.desc1 = (const uint32_t*)d_d1;
        B.desc2 = (const uint32_t*)d_d2;
        B.angle1 = d_a1;
        B.angle2 = d_a2;
        B.valid1 = d_v1;
        B.valid2 = d_v2;
        B.check_orientation = check_orientation;
        B.tri = tri;
        B.thr = thr;
        B.octave1 = d_o1;
        B.bearings1 = d_b1;
        B.bearings2 = d_b2;
        B.xright1 = d_x1;
        B.xright2 = d_x2;
        if (tri) {
            std::memcpy(B.E12, E12, sizeof B.E12);
            std::memcpy(B.epipole, epipole, sizeof B.epipole);
            B.valid_epipole = valid_epipole;
            for (int l = 0; l < num_levels; ++l) B.scale_factors[l] = scale_factors[l];
            B.residual_rad_thr = residual_rad_thr;
        }
        B.key1 = key1;
        B.q_idx = q_idx;
        B.key2 = key2;
        B.t_sorted = t_sorted;
        if (fresh) {
            rc = sv_bucket_sort(ctx, s, d_n1, n1, sort_scratch, sortb, key1, q_idx);
            if (rc) return rc;
            rc = sv_bucket_sort(ctx, s, d_n2, n2, sort_scratch, sortb, key2, t_sorted);
            if (rc) return rc;
            sv_bucket_rows(s, B);
            sv_bucket_count(s, B);
        }
        if (!pass) {
            SV_HIP(ctx, hipMemcpyAsync(&total, B.cand_off + n1, 4, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
            if (total == 0) return SVGPU_OK;
            continue;
        }
        B.cand_idx = A.take<int32_t>(total);
        if (A.off + pad((size_t)total * 4) > ctx->scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "bucket matcher: internal arena overflow");
        sv_bucket_fill(s, B);
        sv_bucket_gather_rows(s, B, qdesc_rows, qangle_rows);
        CandProblem P{};
        P.qdesc = qdesc_rows;
        P.tdesc = (const uint32_t*)d_d2;
        P.nq = n1;
        P.nt = n2;
        P.cand_off = B.cand_off;
        P.cand_idx = B.cand_idx;
        P.q_valid = B.q_valid;
        P.occupied = d_occ;
        P.check_orientation = 0;  // gated in the scan
        P.thr = thr;
        P.lowe_ratio = lowe_ratio;
        P.mode = mode;
        P.dist = A.take<uint32_t>(total);
        P.match_q = match_rows;
        P.num = d_num;
        sv_launch_cand(ctx, s, P, owner, match, mdist);
        SV_HIP(ctx, hipMemsetAsync(d_match, 0xFF, (size_t)n1 * 4, s));
        sv_bucket_scatter(s, B, match_rows, d_match);
        SV_HIP(ctx, hipGetLastError());
        int32_t num = 0;
        SV_HIP(ctx, hipMemcpyAsync(match_1to2, d_match, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipMemcpyAsync(&num, d_num, 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        *num_matches = num;
    }
    return SVGPU_OK;
}

}  // namespace

extern "C" {

int svgpu_reproject_to_bearing(const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, const double* pos_w, double* bearing, int* valid) {
    if (!cam || !rot_cw || !trans_cw || !pos_w || !bearing || !valid) return SVGPU_ERR_INVALID;
    double p[3];
    for (int i = 0; i < 3; ++i) p[i] = ((rot_cw[3 * i] * pos_w[0] + rot_cw[3 * i + 1] * pos_w[1]) + rot_cw[3 * i + 2] * pos_w[2]) + trans_cw[i];
    auto normalize = [&]() {
        const double nrm = std::sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
        for (int i = 0; i < 3; ++i) bearing[i] = p[i] / nrm;
    };
    if (cam->model == SVGPU_CAM_EQUIRECTANGULAR) {  // equirectangular.cc:75-80
        normalize();
        *valid = 1;
        return SVGPU_OK;
    }
    if (p[2] <= 0.0) {  // perspective.cc:155-157: the un-normalised camera-frame point is what the caller is left with
        for (int i = 0; i < 3; ++i) bearing[i] = p[i];
        *valid = 0;
        return SVGPU_OK;
    }
    const double z_inv = 1.0 / p[2];
    const double x = cam->fx * p[0] * z_inv + cam->cx, y = cam->fy * p[1] * z_inv + cam->cy;
    if (cam->model == SVGPU_CAM_RADIAL_DIVISION) {  // radial_division.cc:146-155: inclusive bounds, normalised only when inside
        if (x < cam->min_x || x > cam->max_x || y < cam->min_y || y > cam->max_y) {
            for (int i = 0; i < 3; ++i) bearing[i] = p[i];
            *valid = 0;
            return SVGPU_OK;
        }
        normalize();
        *valid = 1;
        return SVGPU_OK;
    }
    normalize();  // perspective.cc:165-169, fisheye.cc:204-208
    *valid = (cam->min_x < x && x < cam->max_x && cam->min_y < y && y < cam->max_y) ? 1 : 0;
    return SVGPU_OK;
}

int svgpu_match_current_and_last_frames(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, const double* rot_lw,
                                        const double* trans_lw, int is_monocular, float true_baseline, int n_last, const double* pos_w,
                                        const uint8_t* valid, const uint8_t* lm_desc, const int32_t* octave_last, const float* angle_last,
                                        const uint8_t* lm_has_observation, int num_levels, const float* scale_factors, float margin,
                                        const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_angle, int nt,
                                        const uint8_t* occupied, const float* t_xright, int grid_cols, int grid_rows, int check_orientation,
                                        int32_t* match_last, int* num_matches) {
    const svgpu_frame* const bound = sv_take_bound_frame(ctx);  // one-shot: consumed before anything can fail
    if (!ctx || !cam || !rot_cw || !trans_cw || !rot_lw || !trans_lw || (n_last > 0 && !octave_last))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_current_and_last_frames: bad arguments");
    for (int i = 0; i < n_last; ++i)
        if (octave_last[i] < 0 || octave_last[i] >= num_levels) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_current_and_last_frames: octave out of range");
    double trans_wc[3], trans_lc[3];
    cam_center(rot_cw, trans_cw, trans_wc);            // projection.cc:101
    mat_vec(rot_lw, trans_wc, trans_lc);               // :106
    for (int i = 0; i < 3; ++i) trans_lc[i] += trans_lw[i];
    const bool fwd = is_monocular ? false : trans_lc[2] > (double)true_baseline;    // :110-116
    const bool bwd = is_monocular ? false : -trans_lc[2] > (double)true_baseline;
    ProjQueries Q{n_last, pos_w, nullptr, nullptr, nullptr, valid, octave_last, lm_desc, angle_last, lm_has_observation};
    ProjOpts O{2, 2, 0, fwd ? 1 : (bwd ? 2 : 0), margin, check_orientation, 100u, 0.f, SVGPU_MATCH_BEST_ONLY, 1, 0, 0, nullptr};
    const InCellsFrame F = frame_side(bound, cam, tdesc, t_xy, t_octave, nt, occupied, t_angle, t_xright, true, true, grid_cols, grid_rows);
    return proj_match(ctx, "svgpu_match_current_and_last_frames: bad arguments", cam, rot_cw, trans_cw, trans_wc, Q, num_levels, scale_factors, 1.f, F, O,
                      match_last, num_matches, nullptr, nullptr, nullptr, nullptr);
}

int svgpu_match_frame_and_keyframe_projection(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, int n_kf,
                                              const double* pos_w, const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist,
                                              const uint8_t* lm_desc, const float* angle_kf, int num_levels, const float* scale_factors,
                                              float log_scale_factor, float margin, unsigned hamm_dist_thr, const uint8_t* tdesc, const float* t_xy,
                                              const int32_t* t_octave, const float* t_angle, int nt, const uint8_t* occupied, int grid_cols,
                                              int grid_rows, int check_orientation, int32_t* match_kf, int* num_matches) {
    const svgpu_frame* const bound = sv_take_bound_frame(ctx);  // one-shot: consumed before anything can fail
    if (!ctx || !cam || !rot_cw || !trans_cw) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_frame_and_keyframe_projection: bad arguments");
    double center[3];
    cam_center(rot_cw, trans_cw, center);  // projection.cc:223
    ProjQueries Q{n_kf, pos_w, nullptr, min_valid_dist, max_valid_dist, valid, nullptr, lm_desc, angle_kf, nullptr};
    ProjOpts O{1, 2, 0, 0, margin, check_orientation, hamm_dist_thr, 0.f, SVGPU_MATCH_BEST_ONLY, 0, 0, 0, nullptr};
    const InCellsFrame F = frame_side(bound, cam, tdesc, t_xy, t_octave, nt, occupied, t_angle, nullptr, true, false, grid_cols, grid_rows);
    return proj_match(ctx, "svgpu_match_frame_and_keyframe_projection: bad arguments", cam, rot_cw, trans_cw, center, Q, num_levels, scale_factors,
                      log_scale_factor, F, O, match_kf, num_matches, nullptr, nullptr, nullptr, nullptr);
}

int svgpu_match_by_sim3_transform(svgpu_ctx* ctx, const svgpu_camera* cam, const double* sim3_cw, int n, const double* pos_w, const uint8_t* valid,
                                  const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc,
                                  int num_levels, const float* scale_factors, float log_scale_factor, float margin, const uint8_t* tdesc,
                                  const float* t_xy, const int32_t* t_octave, int nt, const uint8_t* occupied, int grid_cols, int grid_rows,
                                  int32_t* match_lm, int* num_matches) {
    const svgpu_frame* const bound = sv_take_bound_frame(ctx);  // one-shot: consumed before anything can fail
    if (!ctx || !cam || !sim3_cw) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_by_sim3_transform: bad arguments");
    // Sim3 -> SE3, projection.cc:326-330
    const double s_cw = std::sqrt((sim3_cw[0] * sim3_cw[0] + sim3_cw[1] * sim3_cw[1]) + sim3_cw[2] * sim3_cw[2]);
    double rot[9], trans[3], center[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot[3 * i + j] = sim3_cw[4 * i + j] / s_cw;
        trans[i] = sim3_cw[4 * i + 3] / s_cw;
    }
    cam_center(rot, trans, center);
    ProjQueries Q{n, pos_w, mean_normal, min_valid_dist, max_valid_dist, valid, nullptr, lm_desc, nullptr, nullptr};
    ProjOpts O{1, 1, 0, 0, margin, 0, 50u, 0.f, SVGPU_MATCH_BEST_ONLY, 0, 0, 0, nullptr};
    const InCellsFrame F = frame_side(bound, cam, tdesc, t_xy, t_octave, nt, occupied, nullptr, nullptr, false, false, grid_cols, grid_rows);
    return proj_match(ctx, "svgpu_match_by_sim3_transform: bad arguments", cam, rot, trans, center, Q, num_levels, scale_factors, log_scale_factor, F, O,
                      match_lm, num_matches, nullptr, nullptr, nullptr, nullptr);
}

int svgpu_match_keyframes_mutually(svgpu_ctx* ctx, const svgpu_camera* cam1, const svgpu_camera* cam2, const double* rot_1w, const double* trans_1w,
                                   const double* rot_2w, const double* trans_2w, float s_12, const double* rot_12, const double* trans_12,
                                   int n1, const double* pos_w1, const uint8_t* valid1, const float* min_valid1, const float* max_valid1,
                                   const uint8_t* lm_desc1, const uint8_t* desc1, const float* xy1, const int32_t* octave1,
                                   int n2, const double* pos_w2, const uint8_t* valid2, const float* min_valid2, const float* max_valid2,
                                   const uint8_t* lm_desc2, const uint8_t* desc2, const float* xy2, const int32_t* octave2,
                                   int num_levels, const float* scale_factors, float log_scale_factor, float margin, int grid_cols, int grid_rows,
                                   int32_t* matched_2_in_1, int32_t* matched_1_in_2, int32_t* mutual_2_in_1, int* num_matches) {
    if (!ctx || !cam1 || !cam2 || !rot_1w || !trans_1w || !rot_2w || !trans_2w || !rot_12 || !trans_12 || !num_matches
        || (n1 > 0 && (!matched_2_in_1 || !mutual_2_in_1)) || (n2 > 0 && !matched_1_in_2))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_match_keyframes_mutually: bad arguments");
    // similarity between the keyframes, projection.cc:428-431
    double s_rot_12[9], s_rot_21[9], trans_21[3];
    const double inv_s = 1.0 / (double)s_12;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            s_rot_12[3 * i + j] = (double)s_12 * rot_12[3 * i + j];
            s_rot_21[3 * i + j] = inv_s * rot_12[3 * j + i];
        }
    for (int i = 0; i < 3; ++i) trans_21[i] = ((-s_rot_21[3 * i]) * trans_12[0] + (-s_rot_21[3 * i + 1]) * trans_12[1]) + (-s_rot_21[3 * i + 2]) * trans_12[2];
    double s_rot_21w[9], trans_21w[3], s_rot_12w[9], trans_12w[3], zero[3] = {0, 0, 0};
    mat_mul(s_rot_21, rot_1w, s_rot_21w);  // :457-458
    mat_vec(s_rot_21, trans_1w, trans_21w);
    for (int i = 0; i < 3; ++i) trans_21w[i] += trans_21[i];
    mat_mul(s_rot_12, rot_2w, s_rot_12w);  // :529-530
    mat_vec(s_rot_12, trans_2w, trans_12w);
    for (int i = 0; i < 3; ++i) trans_12w[i] += trans_12[i];
    int num = 0;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = mutual_2_in_1[i] = -1;
    for (int i = 0; i < n2; ++i) matched_1_in_2[i] = -1;
    *num_matches = 0;
    {  // landmarks of keyframe 1 into keyframe 2 (:459-523); the image test uses keyframe 2's camera
        ProjQueries Q{n1, pos_w1, nullptr, min_valid1, max_valid1, valid1, nullptr, lm_desc1, nullptr, nullptr};
        ProjOpts O{1, 2, 1, 0, margin, 0, 100u, 0.f, SVGPU_MATCH_BEST_ONLY, 0, 0, 1, nullptr};
        InCellsFrame F{desc2, xy2, octave2, n2, nullptr, nullptr, nullptr, cam2->min_x, cam2->max_x, cam2->min_y, cam2->max_y, grid_cols, grid_rows};
        int rc = proj_match(ctx, "svgpu_match_keyframes_mutually: bad arguments", cam2, s_rot_21w, trans_21w, zero, Q, num_levels, scale_factors,
                            log_scale_factor, F, O, matched_2_in_1, &num, nullptr, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    {  // landmarks of keyframe 2 into keyframe 1 (:531-595); the reference reprojects with keyframe 2's camera here as well (:550) and
       // looks the keypoints up in keyframe 1's grid (:571)
        ProjQueries Q{n2, pos_w2, nullptr, min_valid2, max_valid2, valid2, nullptr, lm_desc2, nullptr, nullptr};
        ProjOpts O{1, 2, 1, 0, margin, 0, 100u, 0.f, SVGPU_MATCH_BEST_ONLY, 0, 0, 1, nullptr};
        InCellsFrame F{desc1, xy1, octave1, n1, nullptr, nullptr, nullptr, cam1->min_x, cam1->max_x, cam1->min_y, cam1->max_y, grid_cols, grid_rows};
        int rc = proj_match(ctx, "svgpu_match_keyframes_mutually: bad arguments", cam2, s_rot_12w, trans_12w, zero, Q, num_levels, scale_factors,
                            log_scale_factor, F, O, matched_1_in_2, &num, nullptr, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    num = 0;
    for (int i = 0; i < n1; ++i) {  // cross-check, :598-610
        const int idx_2 = matched_2_in_1[i];
        if (idx_2 < 0) continue;
        if (matched_1_in_2[idx_2] == i) {
            mutual_2_in_1[i] = idx_2;
            ++num;
        }
    }
    *num_matches = num;
    return SVGPU_OK;
}

int svgpu_fuse_detect_duplication(svgpu_ctx* ctx, const svgpu_camera* cam, const double* rot_cw, const double* trans_cw, int n, const double* pos_w,
                                  const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal,
                                  const uint8_t* lm_desc, int num_levels, const float* scale_factors, const float* inv_level_sigma_sq,
                                  float log_scale_factor, float margin, int do_reprojection_matching, const uint8_t* tdesc, const float* t_xy,
                                  const int32_t* t_octave, const float* t_xright, int nt, int grid_cols, int grid_rows, int32_t* best_idx,
                                  int* num_fused) {
    const svgpu_frame* const bound = sv_take_bound_frame(ctx);  // one-shot: consumed before anything can fail
    if (!ctx || !cam || !rot_cw || !trans_cw) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_fuse_detect_duplication: bad arguments");
    double center[3];
    cam_center(rot_cw, trans_cw, center);  // fuse.cc:20
    ProjQueries Q{n, pos_w, mean_normal, min_valid_dist, max_valid_dist, valid, nullptr, lm_desc, nullptr, nullptr};
    ProjOpts O{1, 1, 0, 0, margin, 0, 50u, 0.f, SVGPU_MATCH_BEST_ONLY, 0, do_reprojection_matching ? 1 : 0, 0, inv_level_sigma_sq};
    const InCellsFrame F = frame_side(bound, cam, tdesc, t_xy, t_octave, nt, nullptr, nullptr, t_xright, false, true, grid_cols, grid_rows);
    return proj_match(ctx, "svgpu_fuse_detect_duplication: bad arguments", cam, rot_cw, trans_cw, center, Q, num_levels, scale_factors, log_scale_factor, F,
                      O, best_idx, num_fused, nullptr, nullptr, nullptr, nullptr);
}

int svgpu_match_for_triangulation(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, const int32_t* octave1, const double* bearings1,
                                  const uint8_t* has_lm1, const float* xright1, int n1, const uint8_t* desc2, const float* angle2,
                                  const double* bearings2, const uint8_t* has_lm2, const float* xright2, int n2, const int32_t* node1,
                                  const int32_t* node2, const double* E_12, const double* epipole_in_2, int valid_epipole, const float* scale_factors,
                                  int num_levels, float residual_rad_thr, float lowe_ratio, int check_orientation, int32_t* matched_2_in_1,
                                  int* num_matches) {
    std::vector<uint8_t> v1(n1 > 0 ? n1 : 0), v2(n2 > 0 ? n2 : 0);  // queries / candidates = keypoints WITHOUT a landmark (robust.cc:47-50, 66-70)
    for (int i = 0; i < n1; ++i) v1[i] = has_lm1 ? !has_lm1[i] : 1;
    for (int i = 0; i < n2; ++i) v2[i] = has_lm2 ? !has_lm2[i] : 1;
    BucketSide S1{desc1, angle1, v1.data(), node1, octave1, bearings1, xright1, n1};
    BucketSide S2{desc2, angle2, v2.data(), node2, nullptr, bearings2, xright2, n2};
    return bucket_match(ctx, "svgpu_match_for_triangulation: bad arguments", S1, S2, nullptr, 1, E_12, epipole_in_2, valid_epipole, scale_factors, num_levels,
                        residual_rad_thr, 50u, lowe_ratio, check_orientation, SVGPU_MATCH_TRIANGULATION, matched_2_in_1, num_matches);
}

int svgpu_bow_match(svgpu_ctx* ctx, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const int32_t* node1, int n1,
                    const uint8_t* desc2, const float* angle2, const uint8_t* valid2, const int32_t* node2, int n2, const uint8_t* occupied2,
                    float lowe_ratio, int check_orientation, int32_t* match_1to2, int* num_matches) {
    if (!node1 || !node2) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_bow_match: node ids are required");
    BucketSide S1{desc1, angle1, valid1, node1, nullptr, nullptr, nullptr, n1};
    BucketSide S2{desc2, angle2, valid2, node2, nullptr, nullptr, nullptr, n2};
    return bucket_match(ctx, "svgpu_bow_match: bad arguments", S1, S2, occupied2, 0, nullptr, nullptr, 0, nullptr, 0, 0.f, 50u, lowe_ratio, check_orientation,
                        SVGPU_MATCH_RATIO, match_1to2, num_matches);
}

}  // extern "C"
