// Batched landmark refresh (second half): data::landmark::compute_descriptor
// (data/landmark.cc:199-254) and update_mean_normal_and_obs_scale_variance (:256-318) over CSR observation lists.
// The reference runs these per landmark under a mutex after every local BA / triangulation (mapping_module.cc,
// local_bundle_adjuster_g2o.cc step 8); here all landmarks of the local map go in one call.
//
// compute_descriptor: thread per OBSERVATION (row of the k x k Hamming matrix).  The row's lower median is found without
// storing or sorting the row: distances are 0..256, so a 9-step binary search on the value v with "how many entries <= v"
// counted by recomputing the row (8 xor + 8 popcount per entry, descriptors stay in L2) gives the element at sorted index
// floor(0.5 (k-1)) exactly.  A second thread-per-landmark pass takes the first row with the smallest median (the reference's
// strict "<" scan) and does the fp64 geometry in the reference's operation order.
#include "svgpu_internal.h"

namespace {

inline size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }

__global__ void __launch_bounds__(256) k_lm_owner(int n, const int32_t* __restrict__ obs_off, int32_t* __restrict__ owner) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n) return;
    for (int o = obs_off[l]; o < obs_off[l + 1]; ++o) owner[o] = l;
}

__global__ void __launch_bounds__(256) k_lm_median(int total, const int32_t* __restrict__ obs_off, const int32_t* __restrict__ owner,
                                                   const uint4* __restrict__ desc, uint16_t* __restrict__ median) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int l = owner[o], beg = obs_off[l], k = obs_off[l + 1] - beg;
    const uint4 a0 = desc[2 * (size_t)o], a1 = desc[2 * (size_t)o + 1];
    const int m = (int)(unsigned)(0.5 * (k - 1));  // landmark.cc:240
    int lo = 0, hi = 256;                          // smallest v with #(d <= v) >= m + 1
    while (lo < hi) {
        const int v = (lo + hi) >> 1;
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            const uint4 b0 = desc[2 * (size_t)(beg + j)], b1 = desc[2 * (size_t)(beg + j) + 1];
            const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x)
                          + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            cnt += d <= v;
        }
        if (cnt >= m + 1) hi = v;
        else lo = v + 1;
    }
    median[o] = (uint16_t)lo;
}

__global__ void __launch_bounds__(256) k_lm_pick(int n, const int32_t* __restrict__ obs_off, const uint16_t* __restrict__ median,
                                                 const uint4* __restrict__ desc, int32_t* __restrict__ best_obs, uint4* __restrict__ out) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n) return;
    const int beg = obs_off[l], k = obs_off[l + 1] - beg;
    unsigned best_median = 256;  // match::MAX_HAMMING_DIST
    int best = 0;
    for (int i = 0; i < k; ++i) {
        const unsigned md = median[beg + i];
        if (md < best_median) {
            best_median = md;
            best = i;
        }
    }
    best_obs[l] = best;
    if (k > 0) {
        out[2 * (size_t)l] = desc[2 * (size_t)(beg + best)];
        out[2 * (size_t)l + 1] = desc[2 * (size_t)(beg + best) + 1];
    }
}

__global__ void __launch_bounds__(256) k_lm_geometry(int n, const int32_t* __restrict__ obs_off, const double* __restrict__ obs_twc,
                                                     const double* __restrict__ pos_w, const double* __restrict__ ref_twc,
                                                     const float* __restrict__ ref_sf, float inv_sf_last, double* __restrict__ mean_normal,
                                                     float* __restrict__ max_d, float* __restrict__ min_d) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n) return;
    const double p0 = pos_w[3 * l], p1 = pos_w[3 * l + 1], p2 = pos_w[3 * l + 2];
    double m0 = 0.0, m1 = 0.0, m2 = 0.0;
    for (int o = obs_off[l]; o < obs_off[l + 1]; ++o) {  // landmark.cc:256-266
        const double v0 = p0 - obs_twc[3 * (size_t)o], v1 = p1 - obs_twc[3 * (size_t)o + 1], v2 = p2 - obs_twc[3 * (size_t)o + 2];
        const double sq = (v0 * v0 + v1 * v1) + v2 * v2;
        if (sq > 0.0) {  // Eigen normalized(): n / sqrt(squaredNorm) when it is positive
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
    const double w0 = p0 - ref_twc[3 * l], w1 = p1 - ref_twc[3 * l + 1], w2 = p2 - ref_twc[3 * l + 2];  // landmark.cc:268-283
    const double dist = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    const float mx = (float)(dist * ref_sf[l]);
    max_d[l] = mx;
    min_d[l] = mx * inv_sf_last;
}

struct Bump {
    char* base;
    size_t off = 0;
    template <class T>
    T* take(size_t n) {
        T* r = (T*)(base + off);
        off += pad(n * sizeof(T));
        return r;
    }
};

int check_csr(svgpu_ctx* ctx, const char* who, int n, const int32_t* obs_off, bool need_one) {
    if (!ctx || n < 0 || (n > 0 && !obs_off)) return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    if (n == 0) return SVGPU_OK;
    if (obs_off[0] != 0) return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    for (int l = 0; l < n; ++l)
        if (obs_off[l + 1] < obs_off[l] + (need_one ? 1 : 0)) return sv_set_error(ctx, SVGPU_ERR_INVALID, who);
    return SVGPU_OK;
}

}  // namespace

extern "C" {

int svgpu_landmarks_compute_descriptor(svgpu_ctx* ctx, int n, const int32_t* obs_off, const uint8_t* obs_desc, int32_t* best_obs,
                                       uint8_t* descriptor) {
    int rc = check_csr(ctx, "svgpu_landmarks_compute_descriptor: bad arguments (every landmark needs >= 1 observation)", n, obs_off, true);
    if (rc || n == 0) return rc;
    if (!obs_desc || !best_obs || !descriptor) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_landmarks_compute_descriptor: null pointer");
    const int total = obs_off[n];
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    rc = sv_ensure_scratch(ctx, pad((size_t)(n + 1) * 4) + pad((size_t)total * 32) + pad((size_t)total * 4) + pad((size_t)total * 2) + pad((size_t)n * 4) + pad((size_t)n * 32) + 256);
    if (rc) return rc;
    Bump A{(char*)ctx->d_scratch};
    int32_t* d_off = A.take<int32_t>(n + 1);
    uint4* d_desc = A.take<uint4>((size_t)total * 2);
    int32_t* d_owner = A.take<int32_t>(total);
    uint16_t* d_med = A.take<uint16_t>(total);
    int32_t* d_best = A.take<int32_t>(n);
    uint4* d_out = A.take<uint4>((size_t)n * 2);
    SV_HIP(ctx, hipMemcpyAsync(d_off, obs_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_desc, obs_desc, (size_t)total * 32, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lm_owner, dim3((n + 255) / 256), dim3(256), 0, s, n, d_off, d_owner);
    hipLaunchKernelGGL(k_lm_median, dim3((total + 255) / 256), dim3(256), 0, s, total, d_off, d_owner, d_desc, d_med);
    hipLaunchKernelGGL(k_lm_pick, dim3((n + 255) / 256), dim3(256), 0, s, n, d_off, d_med, d_desc, d_best, d_out);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(best_obs, d_best, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(descriptor, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_landmarks_update_geometry(svgpu_ctx* ctx, int n, const int32_t* obs_off, const double* obs_trans_wc, const double* pos_w,
                                    const double* ref_trans_wc, const float* ref_scale_factor, float inv_scale_factor_last,
                                    double* mean_normal, float* max_valid_dist, float* min_valid_dist) {
    int rc = check_csr(ctx, "svgpu_landmarks_update_geometry: bad arguments", n, obs_off, false);
    if (rc || n == 0) return rc;
    const int total = obs_off[n];
    if ((total > 0 && !obs_trans_wc) || !pos_w || !ref_trans_wc || !ref_scale_factor || !mean_normal || !max_valid_dist || !min_valid_dist)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_landmarks_update_geometry: null pointer");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    rc = sv_ensure_scratch(ctx, pad((size_t)(n + 1) * 4) + pad((size_t)total * 24 + 8) + 3 * pad((size_t)n * 24) + 3 * pad((size_t)n * 4) + 256);
    if (rc) return rc;
    Bump A{(char*)ctx->d_scratch};
    int32_t* d_off = A.take<int32_t>(n + 1);
    double* d_c = A.take<double>((size_t)total * 3 + 1);
    double* d_p = A.take<double>((size_t)n * 3);
    double* d_r = A.take<double>((size_t)n * 3);
    double* d_m = A.take<double>((size_t)n * 3);
    float* d_sf = A.take<float>(n);
    float* d_mx = A.take<float>(n);
    float* d_mn = A.take<float>(n);
    SV_HIP(ctx, hipMemcpyAsync(d_off, obs_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));
    if (total > 0) SV_HIP(ctx, hipMemcpyAsync(d_c, obs_trans_wc, (size_t)total * 24, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_p, pos_w, (size_t)n * 24, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_r, ref_trans_wc, (size_t)n * 24, hipMemcpyHostToDevice, s));
    SV_HIP(ctx, hipMemcpyAsync(d_sf, ref_scale_factor, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lm_geometry, dim3((n + 255) / 256), dim3(256), 0, s, n, d_off, d_c, d_p, d_r, d_sf, inv_scale_factor_last, d_m, d_mx, d_mn);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemcpyAsync(mean_normal, d_m, (size_t)n * 24, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(max_valid_dist, d_mx, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipMemcpyAsync(min_valid_dist, d_mn, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

}  // extern "C"
