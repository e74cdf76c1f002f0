// Candidate lists of the bucketed matchers, built on the device:
//   match::bow_tree::match_frame_and_keyframe / match_keyframes / match_for_triangulation   (match/bow_tree.cc:11-366)
//   match::robust::match_for_triangulation                                                     (match/robust.cc:14-146)
//
// bow_tree walks two std::map<node_id, std::vector<idx>> (bow_feat_vec_) in step and, inside every node both sides share, pairs
// each keypoint of side 1 (in bucket order = index order) with every keypoint of side 2's bucket (index order).  Here both sides
// are STABLE-sorted by node id (rocPRIM radix sort: the order inside a node stays the index order), the sorted side 1 IS the
// reference's query order, and a query's bucket is the equal-range of its node in the sorted side 2 (binary search).  robust::
// match_for_triangulation is the same walk with one bucket holding everything (no node ids: identity order, full range).
//
// k_bucket_scan applies the PAIR gates that do not depend on the greedy bookkeeping -- side-2 validity, orientation, for the
// triangulation matchers the Hamming cut-off and the two epipolar tests of match/base.h:67-79 in fp64 -- and emits, per query, the
// surviving side-2 indices in scan order (count pass, scan, fill pass: one wave per query, ballot + prefix popcount keeps the
// order).  The sequential part (best / second best, already-matched targets) is the candidate replay of match_kernels.hip.
#include "sv_sort.h"

#include "svgpu_internal.h"
#include "match_kernels.h"

namespace {

__device__ __forceinline__ float angle_diff_b(float a1, float a2) {  // util/angle.cc:7-16
    float ret = a1 - a2;
    if (ret <= -180.0f) ret += 360.0f;
    if (ret > 180.0f) ret -= 360.0f;
    return ret;
}

__global__ void k_bucket_keys(const int32_t* __restrict__ node, int n, unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = node ? (node[i] < 0 ? 0xFFFFFFFFu : (unsigned)node[i]) : 0u;  // keypoints without a node sort behind every bucket
    vals[i] = i;
}

// row r = r-th keypoint of side 1 in (node, index) order: its bucket [lo, hi) in the sorted side 2
__global__ void k_bucket_rows(BucketProblem B) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n1) return;
    const int q = B.q_idx[r];
    const unsigned key = B.key1[r];
    bool live = !(B.valid1 && !B.valid1[q]) && key != 0xFFFFFFFFu;
    int lo = 0, hi = 0;
    if (live) {
        int a = 0, b = B.n2;
        while (a < b) {  // lower bound
            const int m = (a + b) >> 1;
            if (B.key2[m] < key) a = m + 1;
            else b = m;
        }
        lo = a;
        b = B.n2;
        while (a < b) {  // upper bound
            const int m = (a + b) >> 1;
            if (B.key2[m] <= key) a = m + 1;
            else b = m;
        }
        hi = a;
    }
    B.row_lo[r] = lo;
    B.row_hi[r] = hi;
    B.q_valid[r] = live && lo < hi;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_bucket_scan(BucketProblem B) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= B.n1) return;
    int total = 0;
    if (B.q_valid[r]) {
        const int q = B.q_idx[r];
        uint32_t qd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qd[k] = B.desc1[(size_t)q * 8 + k];
        const float qa = B.angle1 ? B.angle1[q] : 0.f;
        double b1[3] = {0, 0, 0};
        bool stereo1 = false;
        double thr_rad = 0.0;
        if (B.tri) {
            b1[0] = B.bearings1[3 * q];
            b1[1] = B.bearings1[3 * q + 1];
            b1[2] = B.bearings1[3 * q + 2];
            stereo1 = B.xright1 && 0.f <= B.xright1[q];
            // check_epipolar_constraint(bearing_1, bearing_2, E_12, scale_factors_.at(octave), residual_rad_thr): the two float
            // arguments are multiplied (in float), match/base.h:78
            thr_rad = (double)(B.scale_factors[B.octave1[q]] * B.residual_rad_thr);
        }
        const int lo = B.row_lo[r], hi = B.row_hi[r], base = FILL ? B.cand_off[r] : 0;
        for (int t0 = lo; t0 < hi; t0 += 64) {
            const int t = t0 + lane;
            bool pass = t < hi;
            int j = 0;
            if (pass) {
                j = B.t_sorted[t];
                if (B.valid2 && !B.valid2[j]) pass = false;
            }
            if (pass && B.check_orientation && fabsf(angle_diff_b(qa, B.angle2[j])) > 30.0f) pass = false;
            if (pass && B.tri) {
                unsigned d = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ B.desc2[(size_t)j * 8 + k]);
                if (B.thr < d) pass = false;
                if (pass) {
                    const double b20 = B.bearings2[3 * j], b21 = B.bearings2[3 * j + 1], b22 = B.bearings2[3 * j + 2];
                    const bool stereo2 = B.xright2 && 0.f <= B.xright2[j];
                    if (B.valid_epipole && !stereo1 && !stereo2) {  // robust.cc:93-104: keep clear of the epipole (3 degrees)
                        const double cos_dist = (B.epipole[0] * b20 + B.epipole[1] * b21) + B.epipole[2] * b22;
                        if (0.99862953475 < cos_dist) pass = false;
                    }
                    if (pass) {  // match/base.h:67-79
                        const double* E = B.E12;
                        const double e0 = (E[0] * b20 + E[1] * b21) + E[2] * b22, e1 = (E[3] * b20 + E[4] * b21) + E[5] * b22,
                                     e2 = (E[6] * b20 + E[7] * b21) + E[8] * b22;
                        const double dot = (e0 * b1[0] + e1 * b1[1]) + e2 * b1[2];
                        const double nrm = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
                        const double cos_residual = fmin(1.0, fmax(-1.0, dot / nrm));
                        const double residual_rad = fabs(3.14159265358979323846 / 2.0 - acos(cos_residual));
                        pass = residual_rad < thr_rad;
                    }
                }
            }
            const unsigned long long m = __ballot(pass);
            if (FILL && pass) B.cand_idx[base + total + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = j;
            total += __popcll(m);
        }
    }
    if (!FILL && lane == 0) B.cand_off[r] = total;
}

__global__ void k_bucket_gather_rows(BucketProblem B, uint32_t* __restrict__ qdesc_rows, float* __restrict__ qangle_rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n1) return;
    const int q = B.q_idx[r];
#pragma unroll
    for (int k = 0; k < 8; ++k) qdesc_rows[(size_t)r * 8 + k] = B.desc1[(size_t)q * 8 + k];
    if (qangle_rows) qangle_rows[r] = B.angle1 ? B.angle1[q] : 0.f;
}

__global__ void k_bucket_scatter(BucketProblem B, const int32_t* __restrict__ match_rows, int32_t* __restrict__ match_q) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B.n1) return;
    match_q[B.q_idx[r]] = match_rows[r];
}

// same single-workgroup scan as k_exclusive_scan of match_kernels.hip (kept local: anonymous namespaces)
__global__ __launch_bounds__(1024) void k_bucket_exscan(int32_t* __restrict__ data, int n) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? data[i] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const int carry = s_carry;
        if (i < n) data[i] = carry + s_part[tid] - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s_part[1023];
        __syncthreads();
    }
    if (tid == 0) data[n] = s_carry;
}

}  // namespace

size_t sv_bucket_sort_bytes(int n) {
    const size_t m = (size_t)(n > 0 ? n : 1), p4 = (m * 4 + 255) & ~size_t(255), p8 = (m * 8 + 255) & ~size_t(255);
    return 4 * p4 + 2 * p8 + ((sv_sort_hist_ints(m) * 4 + 255) & ~size_t(255)) + 1024;
}

__global__ void k_bucket_widen(const int* __restrict__ idx, int n, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned long long)(unsigned)idx[i];
}
__global__ void k_bucket_narrow(const unsigned long long* __restrict__ v, int n, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)(unsigned)v[i];
}

// (node, index) order of one side: keys_out / idx_out hold the sorted node ids and the keypoint indices (stable: equal nodes keep index order)
int sv_bucket_sort(svgpu_ctx* ctx, hipStream_t s, const int32_t* node_dev, int n, void* scratch, size_t scratch_bytes, unsigned* keys_out, int* idx_out) {
    if (n <= 0) return SVGPU_OK;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~size_t(255);
        return (void*)r;
    };
    unsigned* keys_in = (unsigned*)take((size_t)n * 4);
    int* vals_in = (int*)take((size_t)n * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "bucket sort scratch too small");
    hipLaunchKernelGGL(k_bucket_keys, dim3((n + 255) / 256), dim3(256), 0, s, node_dev, n, keys_in, vals_in);
    if (!node_dev) {  // one bucket: identity order
        SV_HIP(ctx, hipMemcpyAsync(keys_out, keys_in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        SV_HIP(ctx, hipMemcpyAsync(idx_out, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        return SVGPU_OK;
    }
    SV_HIP(ctx, hipGetLastError());
    if (n <= SV_SORT_SMALL_MAX && sv_sort_small(s, keys_in, vals_in, n, keys_out, idx_out)) return SVGPU_OK;  // every frame: one workgroup, the composites in LDS
    // beyond that: the general radix sort on widened values
    unsigned* keys_b = (unsigned*)take((size_t)n * 4);
    unsigned long long* va = (unsigned long long*)take((size_t)n * 8);
    unsigned long long* vb = (unsigned long long*)take((size_t)n * 8);
    int* hist = (int*)take(sv_sort_hist_ints(n) * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "bucket sort scratch too small");
    hipLaunchKernelGGL(k_bucket_widen, dim3((n + 255) / 256), dim3(256), 0, s, vals_in, n, va);
    unsigned* const keys[2] = {keys_in, keys_b};
    unsigned long long* const vals[2] = {va, vb};
    const int r = sv_sort_pairs(s, keys, vals, 0, n, 32, hist);
    SV_HIP(ctx, hipMemcpyAsync(keys_out, keys[r], (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_bucket_narrow, dim3((n + 255) / 256), dim3(256), 0, s, vals[r], n, idx_out);
    return SVGPU_OK;
}

void sv_bucket_rows(hipStream_t s, const BucketProblem& B) {
    if (B.n1 > 0) hipLaunchKernelGGL(k_bucket_rows, dim3((B.n1 + 255) / 256), dim3(256), 0, s, B);
}
void sv_bucket_count(hipStream_t s, const BucketProblem& B) {  // cand_off = exclusive scan of the per-row counts, total in cand_off[n1]
    if (B.n1 <= 0) return;
    hipLaunchKernelGGL(k_bucket_scan<false>, dim3((B.n1 + 3) / 4), dim3(256), 0, s, B);
    if (B.n1 <= 16384) sv_scan_i32(s, B.cand_off, B.n1, nullptr);  // (the shuffle-based one-workgroup scan; the kernel above beyond its size)
    else hipLaunchKernelGGL(k_bucket_exscan, dim3(1), dim3(1024), 0, s, B.cand_off, B.n1);
}
void sv_bucket_fill(hipStream_t s, const BucketProblem& B) {
    if (B.n1 > 0) hipLaunchKernelGGL(k_bucket_scan<true>, dim3((B.n1 + 3) / 4), dim3(256), 0, s, B);
}
void sv_bucket_gather_rows(hipStream_t s, const BucketProblem& B, uint32_t* qdesc_rows, float* qangle_rows) {
    if (B.n1 > 0) hipLaunchKernelGGL(k_bucket_gather_rows, dim3((B.n1 + 255) / 256), dim3(256), 0, s, B, qdesc_rows, qangle_rows);
}
void sv_bucket_scatter(hipStream_t s, const BucketProblem& B, const int32_t* match_rows, int32_t* match_q) {
    if (B.n1 > 0) hipLaunchKernelGGL(k_bucket_scatter, dim3((B.n1 + 255) / 256), dim3(256), 0, s, B, match_rows, match_q);
}
