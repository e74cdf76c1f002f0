/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Never linked or called by the product path (stella_vslam_amd/libsvgpu.so).
 *
 * Batched landmark refresh (second half), flat CSR form:
 *   data::landmark::compute_descriptor                          data/landmark.cc:199-254
 *   data::landmark::update_mean_normal_and_obs_scale_variance   data/landmark.cc:256-318
 * The observation lists arrive in the iteration order of the reference's observations_ map (ties and summation order follow it).
 * Eigen: v.normalized() = v / sqrt(v.squaredNorm()) when the squared norm is > 0, squaredNorm = (x*x + y*y) + z*z.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static unsigned hamming32(const uint8_t* a, const uint8_t* b) { /* match/base.h:20-41 */
    const uint32_t* pa = (const uint32_t*)a;
    const uint32_t* pb = (const uint32_t*)b;
    unsigned d = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pa[i] ^ pb[i];
        v -= (v >> 1) & 0x55555555u;
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        d += (((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24;
    }
    return d;
}
static int cmp_u(const void* a, const void* b) {
    const unsigned x = *(const unsigned*)a, y = *(const unsigned*)b;
    return x < y ? -1 : x > y;
}

/* obs_off: n + 1 (CSR over landmarks), obs_desc: obs_off[n] x 32.  best_obs[l] = index INSIDE landmark l's list,
 * descriptor: n x 32 = that row.  Every landmark needs >= 1 observation (the reference asserts it). */
void orc_landmarks_compute_descriptor(int n, const int32_t* obs_off, const uint8_t* obs_desc, int32_t* best_obs, uint8_t* descriptor) {
    for (int l = 0; l < n; ++l) {
        const int k = obs_off[l + 1] - obs_off[l];
        const uint8_t* D = obs_desc + (size_t)obs_off[l] * 32;
        unsigned* row = (unsigned*)malloc(sizeof(unsigned) * (k > 0 ? k : 1));
        unsigned best_median = 256; /* MAX_HAMMING_DIST */
        int best = 0;
        for (int i = 0; i < k; ++i) {
            for (int j = 0; j < k; ++j) row[j] = i == j ? 0 : hamming32(D + 32 * i, D + 32 * j);
            qsort(row, k, sizeof(unsigned), cmp_u);
            const unsigned median = row[(unsigned)(0.5 * (k - 1))];
            if (median < best_median) {
                best_median = median;
                best = i;
            }
        }
        free(row);
        best_obs[l] = best;
        if (k > 0) memcpy(descriptor + 32 * (size_t)l, D + 32 * best, 32);
    }
}

/* obs_trans_wc: obs_off[n] x 3 (camera centre of the observing keyframe, per observation); ref_trans_wc n x 3;
 * ref_scale_factor[l] = scale_factors_[octave of the landmark's keypoint in its reference keyframe]. */
void orc_landmarks_update_geometry(int n, const int32_t* obs_off, const double* obs_trans_wc, const double* pos_w, const double* ref_trans_wc,
                                   const float* ref_scale_factor, float inv_scale_factor_last, double* mean_normal, float* max_valid_dist,
                                   float* min_valid_dist) {
    for (int l = 0; l < n; ++l) {
        const double* p = pos_w + 3 * l;
        double m0 = 0.0, m1 = 0.0, m2 = 0.0;
        for (int o = obs_off[l]; o < obs_off[l + 1]; ++o) {
            const double* c = obs_trans_wc + 3 * (size_t)o;
            const double v0 = p[0] - c[0], v1 = p[1] - c[1], v2 = p[2] - c[2];
            const double sq = (v0 * v0 + v1 * v1) + v2 * v2;
            if (sq > 0.0) {
                const double nr = sqrt(sq);
                m0 = m0 + v0 / nr, m1 = m1 + v1 / nr, m2 = m2 + v2 / nr;
            }
            else m0 = m0 + v0, m1 = m1 + v1, m2 = m2 + v2;
        }
        const double sq = (m0 * m0 + m1 * m1) + m2 * m2;
        if (sq > 0.0) {
            const double nr = sqrt(sq);
            m0 = m0 / nr, m1 = m1 / nr, m2 = m2 / nr;
        }
        mean_normal[3 * l] = m0, mean_normal[3 * l + 1] = m1, mean_normal[3 * l + 2] = m2;
        const double* r = ref_trans_wc + 3 * l;
        const double w0 = p[0] - r[0], w1 = p[1] - r[1], w2 = p[2] - r[2];
        const double dist = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
        const float mx = (float)(dist * ref_scale_factor[l]); /* double * float -> double, stored to float& */
        max_valid_dist[l] = mx;
        min_valid_dist[l] = mx * inv_scale_factor_last;
    }
}
