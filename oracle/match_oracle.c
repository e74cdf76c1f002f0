/*
 * oracle/match_oracle.c -- CPU restatement of stella_vslam's descriptor matchers on flat arrays.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Checker for the HIP matcher kernels and the timed "port"
 * CPU baseline of bench.py; never linked into or called from the product.
 *
 * Restated (reference = /root/reference/src/stella_vslam):
 *   match/base.h:15-17      thresholds LOW=50 HIGH=100 MAX=256
 *   match/base.h:20-41      compute_descriptor_distance_32 (SWAR popcount, literal)
 *   match/base.h:44-65      compute_descriptor_distance_64
 *   util/angle.cc:7-16      util::angle::diff
 *   match/robust.cc:232-328 robust::brute_force_match (sequential greedy, already_matched_indices_1)
 *   match/projection.cc:13-93   match_frame_and_landmarks   -> candidate mode ORC_MODE_RATIO_SAME_OCTAVE
 *   match/projection.cc:95-207  match_current_and_last_frames -> candidate mode ORC_MODE_BEST_ONLY
 *   data/common.h:60-68, data/common.cc:83-108,127-190  grid assignment and get_keypoints_in_cell
 *
 * The reference walks shared_ptr object graphs (frame/keyframe/landmark).  The arithmetic and the
 * sequential bookkeeping are restated here over the flat arrays the C-ABI carries: descriptors
 * N x 32 bytes, angles, octaves, CSR candidate lists in the reference's scan order.
 *
 * PARITY STATUS: Hamming distance is pinned by the reference's known-answer tests
 * (test/stella_vslam/match/base.cc:11-57: 0 / 256 / 128).  No reference test exercises any matcher class; the matcher
 * restatements here (brute force, the candidate-list modes incl. area, stereo) are pinned against the reference's own
 * compiled match/*.cc instead (oracle/ref_local, tests/test_ref_local_match.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_HAMMING_DIST_THR_LOW 50u
#define ORC_HAMMING_DIST_THR_HIGH 100u
#define ORC_MAX_HAMMING_DIST 256u

unsigned orc_hamming_32(const uint8_t* a8, const uint8_t* b8) {
    const uint32_t m1 = 0x55555555U, m2 = 0x33333333U, m3 = 0x0F0F0F0FU, m4 = 0x01010101U;
    uint32_t pa[8], pb[8];
    memcpy(pa, a8, 32);
    memcpy(pb, b8, 32);
    unsigned dist = 0;
    for (unsigned i = 0; i < 8; ++i) {
        uint32_t v = pa[i] ^ pb[i];
        v -= ((v >> 1) & m1);
        v = (v & m2) + ((v >> 2) & m2);
        dist += (((v + (v >> 4)) & m3) * m4) >> 24;
    }
    return dist;
}

unsigned orc_hamming_64(const uint8_t* a8, const uint8_t* b8) {
    const uint64_t m1 = 0x5555555555555555ULL, m2 = 0x3333333333333333ULL, m3 = 0x0F0F0F0F0F0F0F0FULL,
                   m4 = 0x0101010101010101ULL;
    uint64_t pa[4], pb[4];
    memcpy(pa, a8, 32);
    memcpy(pb, b8, 32);
    unsigned dist = 0;
    for (unsigned i = 0; i < 4; ++i) {
        uint64_t v = pa[i] ^ pb[i];
        v -= (v >> 1) & m1;
        v = (v & m2) + ((v >> 2) & m2);
        dist += (unsigned)((((v + (v >> 4)) & m3) * m4) >> 56);
    }
    return dist;
}

float orc_angle_diff(float angle1, float angle2) {
    float ret = angle1 - angle2;
    if (ret <= -180.0) ret += 360.0;
    if (ret > 180.0) ret -= 360.0;
    return ret;
}

/* Full distance matrix (n2 x n1, u16) -- test helper for the tiled GPU kernel. */
void orc_hamming_matrix(const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, uint16_t* out) {
    for (int j = 0; j < n2; ++j)
        for (int i = 0; i < n1; ++i) out[(size_t)j * n1 + i] = (uint16_t)orc_hamming_32(desc2 + 32 * j, desc1 + 32 * i);
}

/* robust.cc:232-328.  Index 1 = frame (scanned, claimed), index 2 = keyframe (outer loop; only
 * entries with a live landmark, valid2[idx_2] != 0, are queried).
 * matched_2_in_1[idx_1] = idx_2 or -1.  Returns the number of matches. */
int orc_brute_force_match(const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2,
                          const uint8_t* valid2, int n2, float lowe_ratio, int check_orientation, int* matched_2_in_1) {
    int num_matches = 0;
    uint8_t* already = (uint8_t*)calloc(n1 > 0 ? n1 : 1, 1);
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    for (int idx_2 = 0; idx_2 < n2; ++idx_2) {
        if (valid2 && !valid2[idx_2]) continue;
        unsigned best = ORC_MAX_HAMMING_DIST, second = ORC_MAX_HAMMING_DIST;
        int best_idx_1 = -1;
        for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
            if (already[idx_1]) continue;
            if (check_orientation && fabsf(orc_angle_diff(angle1[idx_1], angle2[idx_2])) > 30.0) continue;
            const unsigned d = orc_hamming_32(desc2 + 32 * (size_t)idx_2, desc1 + 32 * (size_t)idx_1);
            if (d < best) {
                second = best;
                best = d;
                best_idx_1 = idx_1;
            }
            else if (d < second) {
                second = d;
            }
        }
        if (ORC_HAMMING_DIST_THR_LOW < best) continue;
        if (best_idx_1 < 0) continue;
        if (lowe_ratio * second < (float)best) continue;
        matched_2_in_1[best_idx_1] = idx_2;
        already[best_idx_1] = 1;
        ++num_matches;
    }
    free(already);
    return num_matches;
}

/* ---------------------------------------------------------------- candidate-list matchers */
enum { ORC_MODE_BEST_ONLY = 0, ORC_MODE_RATIO_SAME_OCTAVE = 1, ORC_MODE_RATIO = 2, ORC_MODE_TRIANGULATION = 3, ORC_MODE_AREA = 4 };

/* Queries are processed in index order; each scans its CSR candidate list in order.
 *   skip candidate if occupied[t]                       (projection.cc:52-55 / :167-170)
 *   skip if t_xright[t] > 0 and |q_xright[q]-t_xright[t]| > q_xr_tol[q]   (:57-62 / :172-177)
 *   skip if check_orientation and |diff(q_angle[q], t_angle[t])| > 30      (:179-181)
 * mode RATIO_SAME_OCTAVE keeps best/second and their octaves (:68-79), accepts if best <= thr and not
 * (best_octave == second_octave && best > ratio*second) (:82-90).  mode BEST_ONLY accepts if
 * best <= thr (:191-197).  mode RATIO = bow_tree::match_frame_and_keyframe (match/bow_tree.cc:200-237): best <= thr and not
 * lowe_ratio * second < best.  mode TRIANGULATION = bow_tree / robust ::match_for_triangulation (match/bow_tree.cc:66-140,
 * match/robust.cc:56-130): best starts AT thr, a candidate farther than thr or than the current best is skipped before the
 * (precomputed, cand_skip) epipolar gates, '<' updates, then the plain ratio test.  cand_skip[c] != 0 drops CSR entry c
 * (per-pair gates the caller evaluated: epipolar constraint, chi-square reprojection gate of fuse.cc:92-119, ...).
 * An accepted query occupies its target.  match_q[q] = t or -1.
 * mode AREA = area::match_in_consistent_area (match/area.cc:8-98): targets are never occupied; a candidate is skipped when the
 * target's current match is at least as close (matched_dists_in_frm_2, :47-50); best <= thr, ratio test (:63-70); an accepted
 * query TAKES the target from its previous holder, whose match is cleared (:77-88). */
int orc_match_candidates(const uint8_t* qdesc, int nq, const uint8_t* tdesc, const int32_t* t_octave, int nt,
                         const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* cand_skip, const uint8_t* q_valid,
                         const uint8_t* occupied_init, const float* q_angle, const float* t_angle, int check_orientation,
                         const float* q_xright, const float* t_xright, const float* q_xr_tol, unsigned thr,
                         float lowe_ratio, int mode, int32_t* match_q) {
    int num = 0;
    if (mode == ORC_MODE_AREA) {
        unsigned* mdist = (unsigned*)malloc(sizeof(unsigned) * (nt > 0 ? nt : 1));
        int* holder = (int*)malloc(sizeof(int) * (nt > 0 ? nt : 1));
        for (int t = 0; t < nt; ++t) {
            mdist[t] = ORC_MAX_HAMMING_DIST;
            holder[t] = -1;
        }
        for (int q = 0; q < nq; ++q) match_q[q] = -1;
        for (int q = 0; q < nq; ++q) {
            if (q_valid && !q_valid[q]) continue;
            unsigned best = ORC_MAX_HAMMING_DIST, second = ORC_MAX_HAMMING_DIST;
            int best_idx = -1;
            for (int c = cand_off[q]; c < cand_off[q + 1]; ++c) {
                const int t = cand_idx[c];
                if (cand_skip && cand_skip[c]) continue;
                if (check_orientation && fabsf(orc_angle_diff(q_angle[q], t_angle[t])) > 30.0) continue;
                const unsigned d = orc_hamming_32(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)t);
                if (mdist[t] <= d) continue;
                if (d < best) {
                    second = best;
                    best = d;
                    best_idx = t;
                }
                else if (d < second) second = d;
            }
            if (thr < best || best_idx < 0) continue;
            if ((float)second * lowe_ratio < (float)best) continue;
            if (0 <= holder[best_idx]) {
                match_q[holder[best_idx]] = -1;
                --num;
            }
            match_q[q] = best_idx;
            holder[best_idx] = q;
            mdist[best_idx] = best;
            ++num;
        }
        free(mdist);
        free(holder);
        return num;
    }
    uint8_t* occ = (uint8_t*)calloc(nt > 0 ? nt : 1, 1);
    if (occupied_init) memcpy(occ, occupied_init, nt);
    for (int q = 0; q < nq; ++q) {
        match_q[q] = -1;
        if (q_valid && !q_valid[q]) continue;
        if (cand_off[q + 1] == cand_off[q]) continue;
        unsigned best = mode == ORC_MODE_TRIANGULATION ? thr : ORC_MAX_HAMMING_DIST, second = ORC_MAX_HAMMING_DIST;
        int best_lvl = -1, second_lvl = -1, best_idx = -1;
        for (int c = cand_off[q]; c < cand_off[q + 1]; ++c) {
            const int t = cand_idx[c];
            if (occ[t]) continue;
            if (t_xright && 0 < t_xright[t]) {
                const float err = fabsf(q_xright[q] - t_xright[t]);
                if (q_xr_tol[q] < err) continue;
            }
            if (check_orientation && fabsf(orc_angle_diff(q_angle[q], t_angle[t])) > 30.0) continue;
            const unsigned d = orc_hamming_32(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)t);
            if (mode == ORC_MODE_TRIANGULATION && (thr < d || best < d)) continue;
            if (cand_skip && cand_skip[c]) continue;
            if (d < best) {
                second = best;
                best = d;
                second_lvl = best_lvl;
                best_lvl = t_octave ? t_octave[t] : 0;
                best_idx = t;
            }
            else if (d < second) {
                second_lvl = t_octave ? t_octave[t] : 0;
                second = d;
            }
        }
        if (mode == ORC_MODE_RATIO_SAME_OCTAVE) {
            if (best <= thr) {
                if (best_lvl == second_lvl && (float)best > lowe_ratio * second) continue;
                match_q[q] = best_idx;
                occ[best_idx] = 1;
                ++num;
            }
        }
        else if (mode == ORC_MODE_BEST_ONLY) {
            if (thr < best) continue;
            match_q[q] = best_idx;
            occ[best_idx] = 1;
            ++num;
        }
        else {  /* RATIO, TRIANGULATION */
            if (thr < best || best_idx < 0) continue;
            if (lowe_ratio * second < (float)best) continue;
            match_q[q] = best_idx;
            occ[best_idx] = 1;
            ++num;
        }
    }
    free(occ);
    return num;
}

/* ---------------------------------------------------------------- grid (data/common.*) */
static inline int cvfloor_d(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cvceil_d(double v) {
    int i = (int)v;
    return i + (i < v);
}

/* common.cc:83-108: bin keypoints (x-major [col][row] vectors, push order = keypoint index).
 * Output as CSR over cell = col * rows + row. cell_off has cols*rows+1 entries. */
void orc_assign_keypoints_to_grid(const float* kx, const float* ky, int n, float min_x, float max_x, float min_y,
                                  float max_y, int cols, int rows, int32_t* cell_off, int32_t* cell_items) {
    const double inv_w = (double)cols / (max_x - min_x), inv_h = (double)rows / (max_y - min_y);
    const int nc = cols * rows;
    int32_t* cnt = (int32_t*)calloc(nc + 1, sizeof(int32_t));
    int32_t* cell_of = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        const int cx = cvfloor_d((kx[i] - min_x) * inv_w), cy = cvfloor_d((ky[i] - min_y) * inv_h);
        if (0 <= cx && cx < cols && 0 <= cy && cy < rows) {
            cell_of[i] = cx * rows + cy;
            cnt[cell_of[i]]++;
        }
        else cell_of[i] = -1;
    }
    cell_off[0] = 0;
    for (int c = 0; c < nc; ++c) cell_off[c + 1] = cell_off[c] + cnt[c];
    memset(cnt, 0, sizeof(int32_t) * nc);
    for (int i = 0; i < n; ++i)
        if (cell_of[i] >= 0) cell_items[cell_off[cell_of[i]] + cnt[cell_of[i]]++] = i;
    free(cnt);
    free(cell_of);
}

/* common.cc:127-190.  Returns the number of indices written (<= cap). */
int orc_get_keypoints_in_cell(const float* kx, const float* ky, const int32_t* octave, const int32_t* cell_off,
                              const int32_t* cell_items, float min_x, float max_x, float min_y, float max_y, int cols,
                              int rows, float ref_x, float ref_y, float margin, int min_level, int max_level,
                              int32_t* out, int cap) {
    const double inv_w = (double)cols / (max_x - min_x), inv_h = (double)rows / (max_y - min_y);
    int n = 0;
    int lo_x = cvfloor_d((ref_x - min_x - margin) * inv_w);
    if (lo_x < 0) lo_x = 0;
    if (cols <= lo_x) return 0;
    int hi_x = cvceil_d((ref_x - min_x + margin) * inv_w);
    if (hi_x > cols - 1) hi_x = cols - 1;
    if (hi_x < 0) return 0;
    int lo_y = cvfloor_d((ref_y - min_y - margin) * inv_h);
    if (lo_y < 0) lo_y = 0;
    if (rows <= lo_y) return 0;
    int hi_y = cvceil_d((ref_y - min_y + margin) * inv_h);
    if (hi_y > rows - 1) hi_y = rows - 1;
    if (hi_y < 0) return 0;
    for (int cx = lo_x; cx <= hi_x; ++cx)
        for (int cy = lo_y; cy <= hi_y; ++cy) {
            const int c = cx * rows + cy;
            for (int k = cell_off[c]; k < cell_off[c + 1]; ++k) {
                const int idx = cell_items[k];
                if (0 <= min_level && octave[idx] < min_level) continue;
                if (0 <= max_level && max_level < octave[idx]) continue;
                const float dx = kx[idx] - ref_x, dy = ky[idx] - ref_y;
                if (fabsf(dx) < margin && fabsf(dy) < margin) {
                    if (n < cap) out[n] = idx;
                    ++n;
                }
            }
        }
    return n;
}

/* ---------------------------------------------------------------- match::stereo (match/stereo.cc:20-251)
 * kp arrays use the 28-byte cv::KeyPoint layout (x, y, size, angle, response, octave, class_id).
 * pyr_left / pyr_right: per level pointer, width, height, stride.  hamm_dist_thr = (100 + 50) / 2 (stereo.h:99). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_kp28;

static int cmp_pair_ii(const void* a, const void* b) {
    const int* p = (const int*)a;
    const int* q = (const int*)b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    return p[1] < q[1] ? -1 : (p[1] > q[1]);
}

void orc_stereo_match(const orc_kp28* kl, const uint8_t* dl, int nl, const orc_kp28* kr, const uint8_t* dr, int nr,
                      const uint8_t* const* pyr_left, const uint8_t* const* pyr_right, const int* lw, const int* lh, const int* ls_l,
                      const int* ls_r, const float* scale_factors, const float* inv_scale_factors, int num_levels,
                      float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths) {
    const float min_disp = 0.0f, max_disp = focal_x_baseline / true_baseline;
    const unsigned thr = (ORC_HAMMING_DIST_THR_HIGH + ORC_HAMMING_DIST_THR_LOW) / 2;
    const int rows0 = lh[0];
    (void)num_levels;
    /* get_right_keypoint_indices_in_each_row(2.0): CSR rows -> right indices, in index order */
    int* cnt = (int*)calloc(rows0 + 1, sizeof(int));
    for (int i = 0; i < nr; ++i) {
        const float r = 2.0f * scale_factors[kr[i].octave];
        int max_r = (int)ceil((double)(kr[i].y + r)), min_r = (int)floor((double)(kr[i].y - r));
        for (int row = min_r; row <= max_r; ++row)
            if (row >= 0 && row < rows0) cnt[row + 1]++; /* the reference's .at() would throw outside */
    }
    for (int r = 0; r < rows0; ++r) cnt[r + 1] += cnt[r];
    int* items = (int*)malloc(sizeof(int) * (cnt[rows0] + 1));
    int* fill = (int*)calloc(rows0 + 1, sizeof(int));
    for (int i = 0; i < nr; ++i) {
        const float r = 2.0f * scale_factors[kr[i].octave];
        int max_r = (int)ceil((double)(kr[i].y + r)), min_r = (int)floor((double)(kr[i].y - r));
        for (int row = min_r; row <= max_r; ++row)
            if (row >= 0 && row < rows0) items[cnt[row] + fill[row]++] = i;
    }
    free(fill);
    int* pairs = (int*)malloc(sizeof(int) * 2 * (nl + 1));
    int npairs = 0;
    for (int il = 0; il < nl; ++il) {
        stereo_x_right[il] = -1.0f;
        depths[il] = -1.0f;
    }
    for (int il = 0; il < nl; ++il) {
        const int lvl = kl[il].octave;
        const float y_left = kl[il].y, x_left = kl[il].x;
        const int row = (int)y_left;
        if (row < 0 || row >= rows0) continue;
        if (cnt[row + 1] == cnt[row]) continue;
        const float min_x_right = x_left - max_disp, max_x_right = x_left - min_disp;
        if (max_x_right < 0) continue;
        unsigned best_idx = 0, best = thr;
        for (int c = cnt[row]; c < cnt[row + 1]; ++c) {
            const int ir = items[c];
            if (kr[ir].octave < lvl - 1 || kr[ir].octave > lvl + 1) continue;
            const float xr = kr[ir].x;
            if (xr < min_x_right || max_x_right < xr) continue;
            const unsigned d = orc_hamming_32(dl + 32 * (size_t)il, dr + 32 * (size_t)ir);
            if (d < best) {
                best_idx = ir;
                best = d;
            }
        }
        if (thr <= best) continue;
        /* compute_subpixel_disparity (:180-251) */
        const float x_right = kr[best_idx].x;
        const float isf = inv_scale_factors[lvl];
        const int sxl = (int)lrintf(kl[il].x * isf), syl = (int)lrintf(kl[il].y * isf), sxr = (int)lrintf(x_right * isf);
        const int win = 5, slide = 5;
        const int ini_x = sxr - slide - win, end_x = sxr + slide + win;
        if (ini_x < 0 || lw[lvl] <= end_x) continue;
        if (syl - win < 0 || syl + win >= lh[lvl] || sxl - win < 0 || sxl + win >= lw[lvl]) continue; /* rowRange/colRange would assert */
        const uint8_t* PL = pyr_left[lvl];
        const uint8_t* PR = pyr_right[lvl];
        float best_corr = 3.402823466e+38f;
        int best_off = 0;
        float corr[11];
        const float lc = (float)PL[(size_t)syl * ls_l[lvl] + sxl];
        for (int off = -slide; off <= slide; ++off) {
            const float rc = (float)PR[(size_t)syl * ls_r[lvl] + sxr + off];
            double acc = 0; /* cv::norm(NORM_L1) of CV_32F accumulates in double */
            for (int dy = -win; dy <= win; ++dy)
                for (int dx = -win; dx <= win; ++dx) {
                    const float a = (float)PL[(size_t)(syl + dy) * ls_l[lvl] + sxl + dx] - lc;
                    const float b = (float)PR[(size_t)(syl + dy) * ls_r[lvl] + sxr + off + dx] - rc;
                    acc += fabs((double)(a - b));
                }
            const float c = (float)acc;
            if (c < best_corr) {
                best_corr = c;
                best_off = off;
            }
            corr[slide + off] = c;
        }
        if (best_off == -slide || best_off == slide) continue;
        const float c1 = corr[slide + best_off - 1], c2 = corr[slide + best_off], c3 = corr[slide + best_off + 1];
        const float x_delta = (float)((c1 - c3) / (2.0 * (c1 + c3) - 4.0 * c2));
        if (x_delta < -1.0 || 1.0 < x_delta) continue;
        float best_x_right = scale_factors[lvl] * (sxr + best_off + x_delta);
        float best_disp = kl[il].x - best_x_right;
        if (best_disp < min_disp || max_disp <= best_disp) continue;
        if (best_disp <= 0.0f) {
            best_disp = 0.01f;
            best_x_right = x_left - best_disp;
        }
        depths[il] = focal_x_baseline / best_disp;
        stereo_x_right[il] = best_x_right;
        pairs[2 * npairs] = (int)best_corr; /* std::pair<int, int>: the float correlation is narrowed to int */
        pairs[2 * npairs + 1] = il;
        ++npairs;
    }
    qsort(pairs, npairs, 2 * sizeof(int), cmp_pair_ii);
    const int median_i = npairs / 2;
    const float median_corr = npairs == 0 ? 0.0f : (float)pairs[2 * median_i];
    const float corr_thr = (float)(2.0 * median_corr);
    for (int i = median_i; i < npairs; ++i)
        if (corr_thr < (float)pairs[2 * i]) {
            stereo_x_right[pairs[2 * i + 1]] = -1;
            depths[pairs[2 * i + 1]] = -1;
        }
    free(pairs);
    free(items);
    free(cnt);
}
