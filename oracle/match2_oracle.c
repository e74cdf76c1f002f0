/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Never linked or called by the product path (stella_vslam_amd/libsvgpu.so).
 *
 * One function per reference matcher method, each a line-by-line restatement of that method's own loops on flat arrays
 * (the object graph -- frame / keyframe / landmark -- is what the caller flattens; every `continue` that only looks at the
 * graph is folded into a `valid` / `occupied` / `has_lm` byte):
 *   orc_match_for_triangulation            robust::match_for_triangulation            match/robust.cc:14-146
 *                                          bow_tree::match_for_triangulation          match/bow_tree.cc:11-167   (node ids given)
 *   orc_bow_match                          bow_tree::match_frame_and_keyframe         match/bow_tree.cc:169-256
 *                                          bow_tree::match_keyframes                  match/bow_tree.cc:258-366
 *   orc_match_current_and_last_frames      projection::match_current_and_last_frames  match/projection.cc:95-207
 *   orc_match_frame_and_keyframe_projection projection::match_frame_and_keyframe      match/projection.cc:217-319
 *   orc_match_by_sim3_transform            projection::match_by_Sim3_transform        match/projection.cc:321-416
 *   orc_match_keyframes_mutually           projection::match_keyframes_mutually       match/projection.cc:418-629
 *   orc_fuse_detect_duplication            fuse::detect_duplication<T>                match/fuse.cc:11-154
 *   orc_reproject_to_bearing               camera::*::reproject_to_bearing            camera/perspective.cc:150-170, fisheye.cc:189-209,
 *                                                                                     equirectangular.cc:75-80, radial_division.cc:135-156
 * They deliberately do NOT go through the generic orc_match_candidates of match_oracle.c: the device path is built from generic
 * candidate-list kernels, so these independent literal loops are what it has to reproduce bit for bit.
 * PARITY: the reference's tests hold no vectors for any matcher class; every function here is pinned against the
 * reference's own compiled method instead (oracle/ref_local builds match/*.cc where they lie over stand-in data / Eigen headers;
 * tests/test_ref_local_match.py: identical match lists, mono and stereo).
 * Eigen expressions are taken as ((a0 b0 + a1 b1) + a2 b2) per 3-vector product, norm() = sqrt of that, normalize() = x / norm().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_HAMMING_DIST_THR_LOW 50u
#define ORC_HAMMING_DIST_THR_HIGH 100u
#define ORC_MAX_HAMMING_DIST 256u

typedef struct {
    int32_t model;
    int32_t pad_;
    double cols, rows;
    double fx, fy, cx, cy;
    double dist[5];
    double focal_x_baseline;
    float min_x, max_x, min_y, max_y;
} orc_camera;
enum { CAM_PERSPECTIVE = 0, CAM_FISHEYE = 1, CAM_EQUIRECT = 2, CAM_RADIAL_DIVISION = 3 };

unsigned orc_hamming_32(const uint8_t* a8, const uint8_t* b8);
float orc_angle_diff(float angle1, float angle2);
void orc_assign_keypoints_to_grid(const float* kx, const float* ky, int n, float min_x, float max_x, float min_y, float max_y, int cols, int rows,
                                  int32_t* cell_off, int32_t* cell_items);
int orc_get_keypoints_in_cell(const float* kx, const float* ky, const int32_t* octave, const int32_t* cell_off, const int32_t* cell_items,
                              float min_x, float max_x, float min_y, float max_y, int cols, int rows, float ref_x, float ref_y, float margin,
                              int min_level, int max_level, int32_t* out, int cap);
int orc_reproject_to_image(const orc_camera* c, const double* R, const double* t, const double* pw, double* reproj, float* x_right);

static void mat_vec(const double* M, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = (M[3 * i] * v[0] + M[3 * i + 1] * v[1]) + M[3 * i + 2] * v[2];
}
static void mat_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
static void cam_center(const double* R, const double* t, double* c) { /* -R^T t */
    for (int i = 0; i < 3; ++i) c[i] = ((-R[i]) * t[0] + (-R[3 + i]) * t[1]) + (-R[6 + i]) * t[2];
}

/* landmark::predict_scale_level (data/landmark.cc:336-353) */
static unsigned predict_scale_level(float max_valid_dist, float cam_to_lm_dist, float num_scale_levels, float log_scale_factor) {
    const float ratio = max_valid_dist / cam_to_lm_dist;
    const int pred = (int)ceilf(logf(ratio) / log_scale_factor);
    if (pred < 0) return 0;
    else if (num_scale_levels <= (float)(unsigned)pred) return (unsigned)(num_scale_levels - 1);
    else return (unsigned)pred;
}

/* the keypoint side of a frame / keyframe with its grid (data::assign_keypoints_to_grid over the camera's image bounds) */
typedef struct {
    int n;
    const uint8_t* desc;
    const float* xy; /* n x 2 */
    const int32_t* octave;
    float* kx;
    float* ky;
    int32_t* cell_off;
    int32_t* cell_items;
    int32_t* buf;
    const orc_camera* cam;
    int cols, rows;
} kp_side;
static void side_init(kp_side* S, const orc_camera* cam, const uint8_t* desc, const float* xy, const int32_t* octave, int n, int cols, int rows) {
    S->n = n;
    S->desc = desc;
    S->xy = xy;
    S->octave = octave;
    S->cam = cam;
    S->cols = cols;
    S->rows = rows;
    S->kx = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    S->ky = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        S->kx[i] = xy[2 * i];
        S->ky[i] = xy[2 * i + 1];
    }
    S->cell_off = (int32_t*)malloc(sizeof(int32_t) * ((size_t)cols * rows + 1));
    S->cell_items = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    S->buf = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    orc_assign_keypoints_to_grid(S->kx, S->ky, n, cam->min_x, cam->max_x, cam->min_y, cam->max_y, cols, rows, S->cell_off, S->cell_items);
}
static void side_free(kp_side* S) {
    free(S->kx);
    free(S->ky);
    free(S->cell_off);
    free(S->cell_items);
    free(S->buf);
}
static int side_cell(const kp_side* S, double ref_x, double ref_y, float margin, int min_level, int max_level) { /* get_keypoints_in_cell(float, float, ...) */
    return orc_get_keypoints_in_cell(S->kx, S->ky, S->octave, S->cell_off, S->cell_items, S->cam->min_x, S->cam->max_x, S->cam->min_y, S->cam->max_y,
                                     S->cols, S->rows, (float)ref_x, (float)ref_y, margin, min_level, max_level, S->buf, S->n);
}

int orc_reproject_to_bearing(const orc_camera* c, const double* rot_cw, const double* trans_cw, const double* pos_w, double* bearing) {
    double p[3];
    mat_vec(rot_cw, pos_w, p);
    for (int i = 0; i < 3; ++i) p[i] += trans_cw[i];
    const double nrm = sqrt((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
    if (c->model == CAM_EQUIRECT) {
        for (int i = 0; i < 3; ++i) bearing[i] = p[i] / nrm;
        return 1;
    }
    for (int i = 0; i < 3; ++i) bearing[i] = p[i];
    if (p[2] <= 0.0) return 0;
    const double z_inv = 1.0 / p[2];
    const double x = c->fx * p[0] * z_inv + c->cx, y = c->fy * p[1] * z_inv + c->cy;
    if (c->model == CAM_RADIAL_DIVISION) {
        if (x < c->min_x || x > c->max_x) return 0;
        if (y < c->min_y || y > c->max_y) return 0;
        for (int i = 0; i < 3; ++i) bearing[i] = p[i] / nrm;
        return 1;
    }
    for (int i = 0; i < 3; ++i) bearing[i] = p[i] / nrm;
    return c->min_x < x && x < c->max_x && c->min_y < y && y < c->max_y;
}

/* match/base.h:67-79 */
static int check_epipolar_constraint(const double* bearing_1, const double* bearing_2, const double* E_12, float residual_rad_thr,
                                     float bearing_1_scale_factor) {
    double epiplane_in_1[3];
    mat_vec(E_12, bearing_2, epiplane_in_1);
    const double dot = (epiplane_in_1[0] * bearing_1[0] + epiplane_in_1[1] * bearing_1[1]) + epiplane_in_1[2] * bearing_1[2];
    const double nrm = sqrt((epiplane_in_1[0] * epiplane_in_1[0] + epiplane_in_1[1] * epiplane_in_1[1]) + epiplane_in_1[2] * epiplane_in_1[2]);
    const double cos_residual = fmin(1.0, fmax(-1.0, dot / nrm));
    const double residual_rad = fabs(3.14159265358979323846 / 2.0 - acos(cos_residual));
    return residual_rad < residual_rad_thr * bearing_1_scale_factor;
}

/* one (idx_1, bucket of keyframe 2) step shared by the two triangulation matchers: robust.cc:43-133 == bow_tree.cc:52-139 */
typedef struct {
    const uint8_t *desc1, *desc2;
    const float *angle1, *angle2;
    const int32_t* octave1;
    const double *bearings1, *bearings2;
    const uint8_t *has_lm1, *has_lm2;
    const float *xright1, *xright2;
    const double* E_12;
    const double* epipole;
    int valid_epiplane;
    const float* scale_factors;
    float residual_rad_thr, lowe_ratio;
    int check_orientation;
    uint8_t* is_already_matched_in_keyfrm_2;
    int32_t* matched_indices_2_in_keyfrm_1;
    int num_matches;
} tri_t;
static void tri_step(tri_t* T, int idx_1, const int32_t* keyfrm_2_indices, int n_indices_2) {
    if (T->has_lm1 && T->has_lm1[idx_1]) return;
    const int is_stereo_keypt_1 = T->xright1 && 0 <= T->xright1[idx_1];
    const double* bearing_1 = T->bearings1 + 3 * (size_t)idx_1;
    const uint8_t* desc_1 = T->desc1 + 32 * (size_t)idx_1;
    unsigned best_hamm_dist = ORC_HAMMING_DIST_THR_LOW;
    int best_idx_2 = -1;
    unsigned second_best_hamm_dist = ORC_MAX_HAMMING_DIST;
    for (int k = 0; k < n_indices_2; ++k) {
        const int idx_2 = keyfrm_2_indices ? keyfrm_2_indices[k] : k;
        if (T->has_lm2 && T->has_lm2[idx_2]) continue;
        if (T->is_already_matched_in_keyfrm_2[idx_2]) continue;
        if (T->check_orientation && fabsf(orc_angle_diff(T->angle1[idx_1], T->angle2[idx_2])) > 30.0) continue;
        const int is_stereo_keypt_2 = T->xright2 && 0 <= T->xright2[idx_2];
        const double* bearing_2 = T->bearings2 + 3 * (size_t)idx_2;
        const unsigned hamm_dist = orc_hamming_32(desc_1, T->desc2 + 32 * (size_t)idx_2);
        if (ORC_HAMMING_DIST_THR_LOW < hamm_dist || best_hamm_dist < hamm_dist) continue;
        if (T->valid_epiplane && !is_stereo_keypt_1 && !is_stereo_keypt_2) {
            const double cos_dist = (T->epipole[0] * bearing_2[0] + T->epipole[1] * bearing_2[1]) + T->epipole[2] * bearing_2[2];
            const double cos_dist_thr = 0.99862953475;
            if (cos_dist_thr < cos_dist) continue;
        }
        const int is_inlier = check_epipolar_constraint(bearing_1, bearing_2, T->E_12, T->scale_factors[T->octave1[idx_1]], T->residual_rad_thr);
        if (is_inlier) {
            if (hamm_dist < best_hamm_dist) {
                second_best_hamm_dist = best_hamm_dist;
                best_hamm_dist = hamm_dist;
                best_idx_2 = idx_2;
            }
            else if (hamm_dist < second_best_hamm_dist) second_best_hamm_dist = hamm_dist;
        }
    }
    if (best_idx_2 < 0) return;
    if (T->lowe_ratio * second_best_hamm_dist < (float)best_hamm_dist) return;
    T->is_already_matched_in_keyfrm_2[best_idx_2] = 1;
    T->matched_indices_2_in_keyfrm_1[idx_1] = best_idx_2;
    ++T->num_matches;
}

/* std::map<node, std::vector<idx>> of one side as CSR: nodes ascending, a node's indices in push order (= index order) */
typedef struct {
    int n_nodes;
    int32_t* node_id;
    int32_t* off;
    int32_t* idx;
} feat_vec;
static int cmp_i64(const void* a, const void* b) {
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}
static void feat_vec_build(feat_vec* F, const int32_t* node, int n) {
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (node[i] >= 0) keys[m++] = ((int64_t)node[i] << 32) | (uint32_t)i;
    qsort(keys, m, sizeof(int64_t), cmp_i64);
    F->node_id = (int32_t*)malloc(sizeof(int32_t) * (m + 1));
    F->off = (int32_t*)malloc(sizeof(int32_t) * (m + 2));
    F->idx = (int32_t*)malloc(sizeof(int32_t) * (m + 1));
    F->n_nodes = 0;
    for (int k = 0; k < m; ++k) {
        const int32_t nd = (int32_t)(keys[k] >> 32);
        if (F->n_nodes == 0 || F->node_id[F->n_nodes - 1] != nd) {
            F->node_id[F->n_nodes] = nd;
            F->off[F->n_nodes] = k;
            ++F->n_nodes;
        }
        F->idx[k] = (int32_t)(keys[k] & 0xFFFFFFFF);
    }
    F->off[F->n_nodes] = m;
    free(keys);
}
static void feat_vec_free(feat_vec* F) {
    free(F->node_id);
    free(F->off);
    free(F->idx);
}

int orc_match_for_triangulation(const uint8_t* desc1, const float* angle1, const int32_t* octave1, const double* bearings1, const uint8_t* has_lm1,
                                const float* xright1, int n1, const uint8_t* desc2, const float* angle2, const double* bearings2,
                                const uint8_t* has_lm2, const float* xright2, int n2, const int32_t* node1, const int32_t* node2, const double* E_12,
                                const double* epipole_in_2, int valid_epipole, const float* scale_factors, float residual_rad_thr, float lowe_ratio,
                                int check_orientation, int32_t* matched_2_in_1) {
    tri_t T = {desc1, desc2, angle1, angle2, octave1, bearings1, bearings2, has_lm1, has_lm2, xright1, xright2, E_12, epipole_in_2, valid_epipole,
               scale_factors, residual_rad_thr, lowe_ratio, check_orientation, NULL, matched_2_in_1, 0};
    T.is_already_matched_in_keyfrm_2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    if (!node1) { /* robust.cc:43-133 */
        for (int idx_1 = 0; idx_1 < n1; ++idx_1) tri_step(&T, idx_1, NULL, n2);
    }
    else { /* bow_tree.cc:37-153: merge-join of the two bow_feat_vec_ maps */
        feat_vec F1, F2;
        feat_vec_build(&F1, node1, n1);
        feat_vec_build(&F2, node2, n2);
        int i1 = 0, i2 = 0;
        while (i1 < F1.n_nodes && i2 < F2.n_nodes) {
            if (F1.node_id[i1] == F2.node_id[i2]) {
                for (int a = F1.off[i1]; a < F1.off[i1 + 1]; ++a) tri_step(&T, F1.idx[a], F2.idx + F2.off[i2], F2.off[i2 + 1] - F2.off[i2]);
                ++i1;
                ++i2;
            }
            else if (F1.node_id[i1] < F2.node_id[i2]) {
                while (i1 < F1.n_nodes && F1.node_id[i1] < F2.node_id[i2]) ++i1; /* lower_bound */
            }
            else {
                while (i2 < F2.n_nodes && F2.node_id[i2] < F1.node_id[i1]) ++i2;
            }
        }
        feat_vec_free(&F1);
        feat_vec_free(&F2);
    }
    free(T.is_already_matched_in_keyfrm_2);
    return T.num_matches;
}

/* bow_tree::match_frame_and_keyframe (side 1 = keyframe, side 2 = frame, valid2 = NULL) and bow_tree::match_keyframes
 * (valid2 = keypoint of keyframe 2 holds a live landmark); bow_tree.cc:169-256, 258-366 */
int orc_bow_match(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, const int32_t* node1, int n1, const uint8_t* desc2,
                  const float* angle2, const uint8_t* valid2, const int32_t* node2, int n2, const uint8_t* occupied2, float lowe_ratio,
                  int check_orientation, int32_t* match_1to2) {
    int num_matches = 0;
    uint8_t* taken = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1); /* matched_lms_in_frm non-null / is_already_matched_in_keyfrm_2 */
    if (occupied2) memcpy(taken, occupied2, n2);
    for (int i = 0; i < n1; ++i) match_1to2[i] = -1;
    feat_vec F1, F2;
    feat_vec_build(&F1, node1, n1);
    feat_vec_build(&F2, node2, n2);
    int i1 = 0, i2 = 0;
    while (i1 < F1.n_nodes && i2 < F2.n_nodes) {
        if (F1.node_id[i1] == F2.node_id[i2]) {
            for (int a = F1.off[i1]; a < F1.off[i1 + 1]; ++a) {
                const int idx_1 = F1.idx[a];
                if (valid1 && !valid1[idx_1]) continue;
                const uint8_t* desc_1 = desc1 + 32 * (size_t)idx_1;
                unsigned best_hamm_dist = ORC_MAX_HAMMING_DIST;
                int best_idx_2 = -1;
                unsigned second_best_hamm_dist = ORC_MAX_HAMMING_DIST;
                for (int b = F2.off[i2]; b < F2.off[i2 + 1]; ++b) {
                    const int idx_2 = F2.idx[b];
                    if (valid2 && !valid2[idx_2]) continue;
                    if (taken[idx_2]) continue;
                    if (check_orientation && fabsf(orc_angle_diff(angle1[idx_1], angle2[idx_2])) > 30.0) continue;
                    const unsigned hamm_dist = orc_hamming_32(desc_1, desc2 + 32 * (size_t)idx_2);
                    if (hamm_dist < best_hamm_dist) {
                        second_best_hamm_dist = best_hamm_dist;
                        best_hamm_dist = hamm_dist;
                        best_idx_2 = idx_2;
                    }
                    else if (hamm_dist < second_best_hamm_dist) second_best_hamm_dist = hamm_dist;
                }
                if (ORC_HAMMING_DIST_THR_LOW < best_hamm_dist) continue;
                if (lowe_ratio * second_best_hamm_dist < (float)best_hamm_dist) continue;
                match_1to2[idx_1] = best_idx_2;
                taken[best_idx_2] = 1;
                ++num_matches;
            }
            ++i1;
            ++i2;
        }
        else if (F1.node_id[i1] < F2.node_id[i2]) {
            while (i1 < F1.n_nodes && F1.node_id[i1] < F2.node_id[i2]) ++i1;
        }
        else {
            while (i2 < F2.n_nodes && F2.node_id[i2] < F1.node_id[i1]) ++i2;
        }
    }
    feat_vec_free(&F1);
    feat_vec_free(&F2);
    free(taken);
    return num_matches;
}

/* projection::match_frame_and_landmarks, match/projection.cc:13-93, on the maps frame::can_observe filled (tracking_module.cc:554-594):
 * visible / reproj / x_right / pred_scale_level per local landmark (visible == 0: the landmark has no entry in lm_to_reproj, or will be
 * erased).  lm_has_observation (nullable = all 1): what `lm && lm->has_observation()` (:52-55) sees for a landmark this very loop added.
 * is_stereo_frame = !frm_obs.stereo_x_right_.empty().  match_lm[i] = frm.add_landmark(local_lm, best_idx) or -1, in landmark order. */
int orc_match_frame_and_landmarks(const orc_camera* cam, int n, const uint8_t* visible, const double* reproj, const float* x_right, const int32_t* pred_scale_level,
                                  const uint8_t* lm_desc, const uint8_t* lm_has_observation, int num_levels, const float* scale_factors, float margin,
                                  float lowe_ratio, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, int nt, const uint8_t* occupied,
                                  const float* t_xright, int grid_cols, int grid_rows, int32_t* match_lm) {
    unsigned num_matches = 0;
    kp_side S;
    side_init(&S, cam, tdesc, t_xy, t_octave, nt, grid_cols, grid_rows);
    uint8_t* frm_lm = (uint8_t*)calloc(nt > 0 ? nt : 1, 1); /* 0 none, 1 a landmark without observation, 2 a landmark with observations */
    for (int i = 0; i < nt; ++i) frm_lm[i] = (occupied && occupied[i]) ? 2 : 0;
    for (int i = 0; i < n; ++i) {
        match_lm[i] = -1;
        if (!visible[i]) continue; /* :23-29 */
        const unsigned pred = (unsigned)pred_scale_level[i];
        const int min_level = 0 > (int)pred - 1 ? 0 : (int)pred - 1;
        const int max_level = num_levels - 1 < (int)(pred + 1) ? num_levels - 1 : (int)(pred + 1);
        const int n_idx = side_cell(&S, reproj[2 * (size_t)i], reproj[2 * (size_t)i + 1], margin * scale_factors[pred], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)i;
        unsigned best_hamm_dist = ORC_MAX_HAMMING_DIST, second_best_hamm_dist = ORC_MAX_HAMMING_DIST;
        int best_scale_level = -1, second_best_scale_level = -1, best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int idx = S.buf[k];
            if (frm_lm[idx] == 2) continue; /* :52-55 */
            if (t_xright && 0 < t_xright[idx]) { /* :57-62 */
                const float reproj_error = fabsf(x_right[i] - t_xright[idx]);
                if (margin * scale_factors[pred] < reproj_error) continue;
            }
            const unsigned dist = orc_hamming_32(d, tdesc + 32 * (size_t)idx);
            if (dist < best_hamm_dist) {
                second_best_hamm_dist = best_hamm_dist;
                best_hamm_dist = dist;
                second_best_scale_level = best_scale_level;
                best_scale_level = t_octave[idx];
                best_idx = idx;
            }
            else if (dist < second_best_hamm_dist) {
                second_best_scale_level = t_octave[idx];
                second_best_hamm_dist = dist;
            }
        }
        if (best_hamm_dist <= ORC_HAMMING_DIST_THR_HIGH) {
            if (best_scale_level == second_best_scale_level && (float)best_hamm_dist > lowe_ratio * (float)second_best_hamm_dist) continue; /* :82 */
            frm_lm[best_idx] = (!lm_has_observation || lm_has_observation[i]) ? 2 : 1; /* :88 */
            match_lm[i] = best_idx;
            ++num_matches;
        }
    }
    side_free(&S);
    free(frm_lm);
    return (int)num_matches;
}

/* projection::match_current_and_last_frames, match/projection.cc:95-207 */
int orc_match_current_and_last_frames(const orc_camera* cam, const double* rot_cw, const double* trans_cw, const double* rot_lw, const double* trans_lw,
                                      int is_monocular, float true_baseline, int n_last, const double* pos_w, const uint8_t* valid,
                                      const uint8_t* lm_desc, const int32_t* octave_last, const float* angle_last, const uint8_t* lm_has_observation,
                                      int num_levels, const float* scale_factors, float margin, const uint8_t* tdesc, const float* t_xy,
                                      const int32_t* t_octave, const float* t_angle, int nt, const uint8_t* occupied, const float* t_xright,
                                      int grid_cols, int grid_rows, int check_orientation, int32_t* match_last) {
    unsigned num_matches = 0;
    double trans_wc[3], trans_lc[3];
    cam_center(rot_cw, trans_cw, trans_wc);
    mat_vec(rot_lw, trans_wc, trans_lc);
    for (int i = 0; i < 3; ++i) trans_lc[i] += trans_lw[i];
    const int assume_forward = is_monocular ? 0 : trans_lc[2] > true_baseline;
    const int assume_backward = is_monocular ? 0 : -trans_lc[2] > true_baseline;
    kp_side S;
    side_init(&S, cam, tdesc, t_xy, t_octave, nt, grid_cols, grid_rows);
    /* curr_frm landmarks as the loop sees them: 0 none, 1 a landmark without observation, 2 a landmark with observations */
    uint8_t* curr_lm = (uint8_t*)calloc(nt > 0 ? nt : 1, 1);
    for (int i = 0; i < nt; ++i) curr_lm[i] = (occupied && occupied[i]) ? 2 : 0;
    for (int idx_last = 0; idx_last < n_last; ++idx_last) {
        match_last[idx_last] = -1;
        if (valid && !valid[idx_last]) continue;
        double reproj[2];
        float x_right;
        const int in_image = orc_reproject_to_image(cam, rot_cw, trans_cw, pos_w + 3 * (size_t)idx_last, reproj, &x_right);
        if (!in_image) continue;
        const unsigned last_scale_level = (unsigned)octave_last[idx_last];
        int min_level, max_level;
        if (assume_forward) {
            min_level = (int)last_scale_level;
            max_level = num_levels - 1 < (int)(last_scale_level + 1) ? num_levels - 1 : (int)(last_scale_level + 1);
        }
        else if (assume_backward) {
            min_level = 0 > (int)last_scale_level - 1 ? 0 : (int)last_scale_level - 1;
            max_level = (int)last_scale_level;
        }
        else {
            min_level = 0 > (int)last_scale_level - 1 ? 0 : (int)last_scale_level - 1;
            max_level = num_levels - 1 < (int)(last_scale_level + 1) ? num_levels - 1 : (int)(last_scale_level + 1);
        }
        const int n_idx = side_cell(&S, reproj[0], reproj[1], margin * scale_factors[last_scale_level], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)idx_last;
        unsigned best_hamm_dist = ORC_MAX_HAMMING_DIST;
        int best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int curr_idx = S.buf[k];
            if (curr_lm[curr_idx] == 2) continue;
            if (t_xright && t_xright[curr_idx] > 0) {
                const float reproj_error = fabsf(x_right - t_xright[curr_idx]);
                if (margin * scale_factors[last_scale_level] < reproj_error) continue;
            }
            if (check_orientation && fabsf(orc_angle_diff(angle_last[idx_last], t_angle[curr_idx])) > 30.0) continue;
            const unsigned hamm_dist = orc_hamming_32(d, tdesc + 32 * (size_t)curr_idx);
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = curr_idx;
            }
        }
        if (ORC_HAMMING_DIST_THR_HIGH < best_hamm_dist) continue;
        curr_lm[best_idx] = (lm_has_observation && !lm_has_observation[idx_last]) ? 1 : 2; /* curr_frm.add_landmark(lm, best_idx) */
        match_last[idx_last] = best_idx;
        ++num_matches;
    }
    free(curr_lm);
    side_free(&S);
    return (int)num_matches;
}

/* the tests of projection.cc:228-241 / :357-372 / fuse.cc:47-69: returns 1 and the predicted level when the landmark passes */
static int scale_and_normal(const double* cam_to_lm_vec, float min_valid, float max_valid, const double* normal /* nullable */, float num_levels,
                            float log_scale_factor, unsigned* pred_scale_level, double* dist_out) {
    const double cam_to_lm_dist = sqrt((cam_to_lm_vec[0] * cam_to_lm_vec[0] + cam_to_lm_vec[1] * cam_to_lm_vec[1]) + cam_to_lm_vec[2] * cam_to_lm_vec[2]);
    const double margin_far = 1.3;
    const double margin_near = 1.0 / margin_far;
    const double max_cam_to_lm_dist = margin_far * max_valid;
    const double min_cam_to_lm_dist = margin_near * min_valid;
    if (cam_to_lm_dist < min_cam_to_lm_dist || max_cam_to_lm_dist < cam_to_lm_dist) return 0;
    if (normal) {
        const double dot = (cam_to_lm_vec[0] * normal[0] + cam_to_lm_vec[1] * normal[1]) + cam_to_lm_vec[2] * normal[2];
        if (dot < 0.5 * cam_to_lm_dist) return 0;
    }
    *pred_scale_level = predict_scale_level(max_valid, (float)cam_to_lm_dist, num_levels, log_scale_factor);
    if (dist_out) *dist_out = cam_to_lm_dist;
    return 1;
}

/* projection::match_frame_and_keyframe, match/projection.cc:217-319 */
int orc_match_frame_and_keyframe_projection(const orc_camera* cam, const double* rot_cw, const double* trans_cw, int n_kf, const double* pos_w,
                                            const uint8_t* valid, const float* min_valid_dist, const float* max_valid_dist, const uint8_t* lm_desc,
                                            const float* angle_kf, int num_levels, const float* scale_factors, float log_scale_factor, float margin,
                                            unsigned hamm_dist_thr, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave,
                                            const float* t_angle, int nt, const uint8_t* occupied, int grid_cols, int grid_rows,
                                            int check_orientation, int32_t* match_kf) {
    unsigned num_matches = 0;
    double cam_center_w[3];
    cam_center(rot_cw, trans_cw, cam_center_w);
    kp_side S;
    side_init(&S, cam, tdesc, t_xy, t_octave, nt, grid_cols, grid_rows);
    uint8_t* frm_landmarks = (uint8_t*)calloc(nt > 0 ? nt : 1, 1);
    if (occupied) memcpy(frm_landmarks, occupied, nt);
    for (int idx = 0; idx < n_kf; ++idx) {
        match_kf[idx] = -1;
        if (valid && !valid[idx]) continue;
        const double* pw = pos_w + 3 * (size_t)idx;
        double reproj[2];
        float x_right;
        if (!orc_reproject_to_image(cam, rot_cw, trans_cw, pw, reproj, &x_right)) continue;
        const double v[3] = {pw[0] - cam_center_w[0], pw[1] - cam_center_w[1], pw[2] - cam_center_w[2]};
        unsigned pred_scale_level;
        if (!scale_and_normal(v, min_valid_dist[idx], max_valid_dist[idx], NULL, (float)num_levels, log_scale_factor, &pred_scale_level, NULL)) continue;
        const int min_level = 0 > (int)pred_scale_level - 1 ? 0 : (int)pred_scale_level - 1;
        const int max_level = num_levels - 1 < (int)(pred_scale_level + 1) ? num_levels - 1 : (int)(pred_scale_level + 1);
        const int n_idx = side_cell(&S, reproj[0], reproj[1], margin * scale_factors[pred_scale_level], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)idx;
        unsigned best_hamm_dist = ORC_MAX_HAMMING_DIST;
        int best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int curr_idx = S.buf[k];
            if (frm_landmarks[curr_idx]) continue;
            if (check_orientation && fabsf(orc_angle_diff(angle_kf[idx], t_angle[curr_idx])) > 30.0) continue;
            const unsigned hamm_dist = orc_hamming_32(d, tdesc + 32 * (size_t)curr_idx);
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = curr_idx;
            }
        }
        if (hamm_dist_thr < best_hamm_dist) continue;
        frm_landmarks[best_idx] = 1;
        match_kf[idx] = best_idx;
        num_matches++;
    }
    free(frm_landmarks);
    side_free(&S);
    return (int)num_matches;
}

/* projection::match_by_Sim3_transform, match/projection.cc:321-416 */
int orc_match_by_sim3_transform(const orc_camera* cam, const double* sim3_cw, int n, const double* pos_w, const uint8_t* valid,
                                const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc,
                                int num_levels, const float* scale_factors, float log_scale_factor, float margin, const uint8_t* tdesc,
                                const float* t_xy, const int32_t* t_octave, int nt, const uint8_t* occupied, int grid_cols, int grid_rows,
                                int32_t* match_lm) {
    unsigned num_matches = 0;
    const double s_cw = sqrt((sim3_cw[0] * sim3_cw[0] + sim3_cw[1] * sim3_cw[1]) + sim3_cw[2] * sim3_cw[2]);
    double rot_cw[9], trans_cw[3], center[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) rot_cw[3 * i + j] = sim3_cw[4 * i + j] / s_cw;
        trans_cw[i] = sim3_cw[4 * i + 3] / s_cw;
    }
    cam_center(rot_cw, trans_cw, center);
    kp_side S;
    side_init(&S, cam, tdesc, t_xy, t_octave, nt, grid_cols, grid_rows);
    uint8_t* matched_lms_in_keyfrm = (uint8_t*)calloc(nt > 0 ? nt : 1, 1);
    if (occupied) memcpy(matched_lms_in_keyfrm, occupied, nt);
    for (int i = 0; i < n; ++i) {
        match_lm[i] = -1;
        if (valid && !valid[i]) continue;
        const double* pw = pos_w + 3 * (size_t)i;
        double reproj[2];
        float x_right;
        if (!orc_reproject_to_image(cam, rot_cw, trans_cw, pw, reproj, &x_right)) continue;
        const double v[3] = {pw[0] - center[0], pw[1] - center[1], pw[2] - center[2]};
        unsigned pred_scale_level;
        if (!scale_and_normal(v, min_valid_dist[i], max_valid_dist[i], mean_normal + 3 * (size_t)i, (float)num_levels, log_scale_factor, &pred_scale_level, NULL)) continue;
        const int min_level = 0 > (int)pred_scale_level - 1 ? 0 : (int)pred_scale_level - 1;
        const int max_level = num_levels - 1 < (int)(pred_scale_level + 1) ? num_levels - 1 : (int)(pred_scale_level + 1);
        const int n_idx = side_cell(&S, reproj[0], reproj[1], margin * scale_factors[pred_scale_level], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)i;
        unsigned best_dist = ORC_MAX_HAMMING_DIST;
        int best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int idx = S.buf[k];
            if (matched_lms_in_keyfrm[idx]) continue;
            const unsigned hamm_dist = orc_hamming_32(d, tdesc + 32 * (size_t)idx);
            if (hamm_dist < best_dist) {
                best_dist = hamm_dist;
                best_idx = idx;
            }
        }
        if (ORC_HAMMING_DIST_THR_LOW < best_dist) continue;
        matched_lms_in_keyfrm[best_idx] = 1;
        match_lm[i] = best_idx;
        ++num_matches;
    }
    free(matched_lms_in_keyfrm);
    side_free(&S);
    return (int)num_matches;
}

/* one pass of match_keyframes_mutually (projection.cc:459-523 / :531-595): landmarks through (s_rot, trans) into the other keyframe */
static void mutual_pass(const orc_camera* cam_image, const double* s_rot, const double* trans, int n, const double* pos_w, const uint8_t* valid,
                        const float* min_valid, const float* max_valid, const uint8_t* lm_desc, kp_side* S, int num_levels, const float* scale_factors,
                        float log_scale_factor, float margin, int32_t* matched) {
    for (int i = 0; i < n; ++i) {
        matched[i] = -1;
        if (valid && !valid[i]) continue;
        const double* pw = pos_w + 3 * (size_t)i;
        double pos_o[3];
        mat_vec(s_rot, pw, pos_o);
        for (int k = 0; k < 3; ++k) pos_o[k] += trans[k];
        double reproj[2];
        float x_right;
        if (!orc_reproject_to_image(cam_image, s_rot, trans, pw, reproj, &x_right)) continue;
        unsigned pred_scale_level;
        if (!scale_and_normal(pos_o, min_valid[i], max_valid[i], NULL, (float)num_levels, log_scale_factor, &pred_scale_level, NULL)) continue;
        const int min_level = 0 > (int)pred_scale_level - 1 ? 0 : (int)pred_scale_level - 1;
        const int max_level = num_levels - 1 < (int)(pred_scale_level + 1) ? num_levels - 1 : (int)(pred_scale_level + 1);
        const int n_idx = side_cell(S, reproj[0], reproj[1], margin * scale_factors[pred_scale_level], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)i;
        unsigned best_hamm_dist = ORC_MAX_HAMMING_DIST;
        int best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int idx = S->buf[k];
            const unsigned hamm_dist = orc_hamming_32(d, S->desc + 32 * (size_t)idx);
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = idx;
            }
        }
        if (best_hamm_dist <= ORC_HAMMING_DIST_THR_HIGH) matched[i] = best_idx;
    }
}

/* projection::match_keyframes_mutually, match/projection.cc:418-629 */
int orc_match_keyframes_mutually(const orc_camera* cam1, const orc_camera* cam2, const double* rot_1w, const double* trans_1w, const double* rot_2w,
                                 const double* trans_2w, float s_12, const double* rot_12, const double* trans_12, int n1, const double* pos_w1,
                                 const uint8_t* valid1, const float* min_valid1, const float* max_valid1, const uint8_t* lm_desc1, const uint8_t* desc1,
                                 const float* xy1, const int32_t* octave1, int n2, const double* pos_w2, const uint8_t* valid2, const float* min_valid2,
                                 const float* max_valid2, const uint8_t* lm_desc2, const uint8_t* desc2, const float* xy2, const int32_t* octave2,
                                 int num_levels, const float* scale_factors, float log_scale_factor, float margin, int grid_cols, int grid_rows,
                                 int32_t* matched_2_in_1, int32_t* matched_1_in_2, int32_t* mutual_2_in_1) {
    double s_rot_12[9], s_rot_21[9], trans_21[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            s_rot_12[3 * i + j] = s_12 * rot_12[3 * i + j];
            s_rot_21[3 * i + j] = (1.0 / s_12) * rot_12[3 * j + i];
        }
    for (int i = 0; i < 3; ++i) trans_21[i] = ((-s_rot_21[3 * i]) * trans_12[0] + (-s_rot_21[3 * i + 1]) * trans_12[1]) + (-s_rot_21[3 * i + 2]) * trans_12[2];
    double s_rot_21w[9], trans_21w[3], s_rot_12w[9], trans_12w[3];
    mat_mul(s_rot_21, rot_1w, s_rot_21w);
    mat_vec(s_rot_21, trans_1w, trans_21w);
    for (int i = 0; i < 3; ++i) trans_21w[i] += trans_21[i];
    mat_mul(s_rot_12, rot_2w, s_rot_12w);
    mat_vec(s_rot_12, trans_2w, trans_12w);
    for (int i = 0; i < 3; ++i) trans_12w[i] += trans_12[i];
    kp_side S1, S2;
    side_init(&S1, cam1, desc1, xy1, octave1, n1, grid_cols, grid_rows);
    side_init(&S2, cam2, desc2, xy2, octave2, n2, grid_cols, grid_rows);
    mutual_pass(cam2, s_rot_21w, trans_21w, n1, pos_w1, valid1, min_valid1, max_valid1, lm_desc1, &S2, num_levels, scale_factors, log_scale_factor, margin,
                matched_2_in_1);
    mutual_pass(cam2 /* :550 uses keyfrm_2->camera_ here too */, s_rot_12w, trans_12w, n2, pos_w2, valid2, min_valid2, max_valid2, lm_desc2, &S1, num_levels,
                scale_factors, log_scale_factor, margin, matched_1_in_2);
    int num_matches = 0;
    for (int i = 0; i < n1; ++i) {
        mutual_2_in_1[i] = -1;
        const int idx_2 = matched_2_in_1[i];
        if (idx_2 < 0) continue;
        if (matched_1_in_2[idx_2] == i) {
            mutual_2_in_1[i] = idx_2;
            ++num_matches;
        }
    }
    side_free(&S1);
    side_free(&S2);
    return num_matches;
}

/* fuse::detect_duplication<T>, match/fuse.cc:11-154 */
int orc_fuse_detect_duplication(const orc_camera* cam, const double* rot_cw, const double* trans_cw, int n, const double* pos_w, const uint8_t* valid,
                                const float* min_valid_dist, const float* max_valid_dist, const double* mean_normal, const uint8_t* lm_desc,
                                int num_levels, const float* scale_factors, const float* inv_level_sigma_sq, float log_scale_factor, float margin,
                                int do_reprojection_matching, const uint8_t* tdesc, const float* t_xy, const int32_t* t_octave, const float* t_xright,
                                int nt, int grid_cols, int grid_rows, int32_t* best_idx_out) {
    double trans_wc[3];
    cam_center(rot_cw, trans_cw, trans_wc);
    unsigned num_fused = 0;
    uint8_t* already_matched_idx_in_keyfrm = (uint8_t*)calloc(nt > 0 ? nt : 1, 1);
    kp_side S;
    side_init(&S, cam, tdesc, t_xy, t_octave, nt, grid_cols, grid_rows);
    for (int i = 0; i < n; ++i) {
        best_idx_out[i] = -1;
        if (valid && !valid[i]) continue;
        const double* pw = pos_w + 3 * (size_t)i;
        double reproj[2];
        float x_right;
        if (!orc_reproject_to_image(cam, rot_cw, trans_cw, pw, reproj, &x_right)) continue;
        const double v[3] = {pw[0] - trans_wc[0], pw[1] - trans_wc[1], pw[2] - trans_wc[2]};
        unsigned pred_scale_level;
        if (!scale_and_normal(v, min_valid_dist[i], max_valid_dist[i], mean_normal + 3 * (size_t)i, (float)num_levels, log_scale_factor, &pred_scale_level, NULL)) continue;
        const int min_level = 0 > (int)pred_scale_level - 1 ? 0 : (int)pred_scale_level - 1;
        const int max_level = num_levels - 1 < (int)(pred_scale_level + 1) ? num_levels - 1 : (int)(pred_scale_level + 1);
        const int n_idx = side_cell(&S, reproj[0], reproj[1], margin * scale_factors[pred_scale_level], min_level, max_level);
        if (n_idx == 0) continue;
        const uint8_t* d = lm_desc + 32 * (size_t)i;
        unsigned best_dist = ORC_MAX_HAMMING_DIST;
        int best_idx = -1;
        for (int k = 0; k < n_idx; ++k) {
            const int idx = S.buf[k];
            if (already_matched_idx_in_keyfrm[idx]) continue;
            if (do_reprojection_matching) {
                const unsigned scale_level = (unsigned)t_octave[idx];
                if (t_xright && t_xright[idx] >= 0) {
                    const double e_x = reproj[0] - t_xy[2 * idx];
                    const double e_y = reproj[1] - t_xy[2 * idx + 1];
                    const float e_x_right = x_right - t_xright[idx];
                    const double reproj_error_sq = e_x * e_x + e_y * e_y + e_x_right * e_x_right;
                    const float chi_sq_3D = 7.81473;
                    if (chi_sq_3D < reproj_error_sq * inv_level_sigma_sq[scale_level]) continue;
                }
                else {
                    const double e_x = reproj[0] - t_xy[2 * idx];
                    const double e_y = reproj[1] - t_xy[2 * idx + 1];
                    const double reproj_error_sq = e_x * e_x + e_y * e_y;
                    const float chi_sq_2D = 5.99146;
                    if (chi_sq_2D < reproj_error_sq * inv_level_sigma_sq[scale_level]) continue;
                }
            }
            const unsigned hamm_dist = orc_hamming_32(d, tdesc + 32 * (size_t)idx);
            if (hamm_dist < best_dist) {
                best_dist = hamm_dist;
                best_idx = idx;
            }
        }
        if (ORC_HAMMING_DIST_THR_LOW < best_dist) continue;
        already_matched_idx_in_keyfrm[best_idx] = 1;
        best_idx_out[i] = best_idx;
        ++num_fused;
    }
    free(already_matched_idx_in_keyfrm);
    side_free(&S);
    return (int)num_fused;
}
