// (edge, edge) pair lists of the reduced camera system, built on the device.
//
// S_ab -= sum over landmarks l and observation pairs (i, j) of l with pose slots (a, b):  Y_i W_j^T.  k_ba_schur walks, per
// upper block (a <= b), the list of such pairs in a FIXED order (fixed-order fp64 sums = bit-reproducible results).  The lists
// used to be built on the host (two passes over ~200 k pairs, 1.2 ms at 20 KF / 10 k landmarks); here every landmark emits its
// pairs in the same sequence (i <= j in edge order, swapped so that a <= b, the mirrored pair right after a same-pose pair),
// keyed by the dense block index, and a STABLE radix sort (hipcub / rocPRIM) groups them by block: the order inside a
// block is the landmark-major order the host pass produced.
#include <hipcub/hipcub.hpp>

#include "svgpu_internal.h"
#include "ba_kernels.h"

namespace {

__device__ __forceinline__ bool edge_live(const BaDev& D, int e) { return !D.e_level[e] && D.pose_slot[D.e_pose[e]] >= 0; }
__device__ __forceinline__ unsigned dense_block(int a, int b, int nP) { return (unsigned)(a * nP - a * (a - 1) / 2 + (b - a)); }  // a <= b

__global__ void k_pair_count(BaDev D, int* __restrict__ cnt) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= D.L) return;
    int n = 0;
    if (D.pt_free[l]) {
        const int lo = D.lm_off[l], hi = D.lm_off[l + 1];
        for (int i = lo; i < hi; ++i) {
            if (!edge_live(D, i)) continue;
            const int a = D.pose_slot[D.e_pose[i]];
            for (int j = i; j < hi; ++j) {
                if (!edge_live(D, j)) continue;
                n += (j != i && D.pose_slot[D.e_pose[j]] == a) ? 2 : 1;
            }
        }
    }
    cnt[l] = n;
}

__global__ void k_pair_emit(BaDev D, const int* __restrict__ off, unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= D.L || !D.pt_free[l]) return;
    int o = off[l];
    const int lo = D.lm_off[l], hi = D.lm_off[l + 1];
    for (int i = lo; i < hi; ++i) {
        if (!edge_live(D, i)) continue;
        for (int j = i; j < hi; ++j) {
            if (!edge_live(D, j)) continue;
            int e1 = i, e2 = j, a = D.pose_slot[D.e_pose[i]], b = D.pose_slot[D.e_pose[j]];
            if (a > b) {
                const int t = a;
                a = b;
                b = t;
                e1 = j;
                e2 = i;
            }
            const unsigned key = dense_block(a, b, D.nP);
            keys[o] = key;
            vals[o++] = (unsigned long long)(unsigned)e1 | ((unsigned long long)(unsigned)e2 << 32);
            if (a == b && e1 != e2) {  // two observations from one pose: both cross terms
                keys[o] = key;
                vals[o++] = (unsigned long long)(unsigned)e2 | ((unsigned long long)(unsigned)e1 << 32);
            }
        }
    }
}

// dense_off[k] = first sorted position with key >= k, k = 0 .. nb_dense
__global__ void k_pair_offsets(const unsigned* __restrict__ keys, int n, int nb_dense, int* __restrict__ dense_off) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nb_dense) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned)k) lo = mid + 1;
        else hi = mid;
    }
    dense_off[k] = lo;
}

}  // namespace

size_t sv_ba_pairs_scratch_bytes(size_t pair_cap, int L, size_t nb_cap) {
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (int*)nullptr, (int*)nullptr, L + 1);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, (int)pair_cap);
    const size_t t = t1 > t2 ? t1 : t2;
    return ((t + 255) & ~size_t(255)) + 2 * (((size_t)(L + 2) * 4 + 255) & ~size_t(255)) + (((pair_cap * 4) + 255) & ~size_t(255)) * 2
           + (((pair_cap * 8) + 255) & ~size_t(255)) + (((nb_cap + 1) * 4 + 255) & ~size_t(255)) + 1024;
}

// D.pose_slot / D.pt_free / D.e_level / D.nP must be current on the device.  Writes the sorted pairs to `pairs_out` (= D.blk_pairs
// storage) and the dense block offsets (nb_dense + 1 ints) to `dense_off_host`.  Synchronises the stream twice.
int sv_ba_build_pairs(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int2* pairs_out,
                      std::vector<int>& dense_off_host) {
    const int L = D.L, nb_dense = D.nP * (D.nP + 1) / 2;
    dense_off_host.assign((size_t)nb_dense + 1, 0);
    if (L == 0 || D.nP == 0) return SVGPU_OK;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~size_t(255);
        return (void*)r;
    };
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (int*)nullptr, (int*)nullptr, L + 1);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, (int)pair_cap);
    const size_t tbytes = t1 > t2 ? t1 : t2;
    void* temp = take(tbytes);
    int* cnt = (int*)take((size_t)(L + 2) * 4);
    int* off = (int*)take((size_t)(L + 2) * 4);
    unsigned* keys_in = (unsigned*)take(pair_cap * 4);
    unsigned* keys_out = (unsigned*)take(pair_cap * 4);
    unsigned long long* vals_in = (unsigned long long*)take(pair_cap * 8);
    int* dense_off = (int*)take(((size_t)nb_dense + 1) * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pair-list scratch too small");
    SV_HIP(ctx, hipGetLastError());  // anything pending from earlier launches is reported here, not by hipcub below
    hipLaunchKernelGGL(k_pair_count, dim3((L + 255) / 256), dim3(256), 0, s, D, cnt);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemsetAsync(cnt + L, 0, 4, s));
    size_t tb = tbytes;
    SV_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(temp, tb, cnt, off, L + 1, s));
    int total = 0;
    SV_HIP(ctx, hipMemcpyAsync(&total, off + L, 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    if ((size_t)total > pair_cap) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pair-list capacity exceeded");
    if (total > 0) {
        hipLaunchKernelGGL(k_pair_emit, dim3((L + 255) / 256), dim3(256), 0, s, D, off, keys_in, vals_in);
        int bits = 1;
        while ((1u << bits) < (unsigned)nb_dense + 1u && bits < 32) ++bits;
        tb = tbytes;
        SV_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(temp, tb, keys_in, keys_out, vals_in, reinterpret_cast<unsigned long long*>(pairs_out),
                                                       total, 0, bits, s));
    }
    hipLaunchKernelGGL(k_pair_offsets, dim3((nb_dense + 256) / 256), dim3(256), 0, s, keys_out, total, nb_dense, dense_off);
    SV_HIP(ctx, hipMemcpyAsync(dense_off_host.data(), dense_off, 4 * ((size_t)nb_dense + 1), hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

// Same pipeline with the pair total known to the caller (svgpu_ba.hip: host_pair_total): nothing is read back, nothing synchronises.
// The dense block offsets (nP (nP + 1) / 2 + 1 ints) are written to `dense_off_dev`.
int sv_ba_build_pairs_async(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int total,
                            int2* pairs_out, int* dense_off_dev) {
    const int L = D.L, nb_dense = D.nP * (D.nP + 1) / 2;
    if (L == 0 || D.nP == 0) {
        SV_HIP(ctx, hipMemsetAsync(dense_off_dev, 0, 4 * ((size_t)nb_dense + 1), s));
        return SVGPU_OK;
    }
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~size_t(255);
        return (void*)r;
    };
    size_t t1 = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (int*)nullptr, (int*)nullptr, L + 1);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned long long*)nullptr,
                                       (unsigned long long*)nullptr, (int)pair_cap);
    const size_t tbytes = t1 > t2 ? t1 : t2;
    void* temp = take(tbytes);
    int* cnt = (int*)take((size_t)(L + 2) * 4);
    int* off = (int*)take((size_t)(L + 2) * 4);
    unsigned* keys_in = (unsigned*)take(pair_cap * 4);
    unsigned* keys_out = (unsigned*)take(pair_cap * 4);
    unsigned long long* vals_in = (unsigned long long*)take(pair_cap * 8);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pair-list scratch too small");
    SV_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_pair_count, dim3((L + 255) / 256), dim3(256), 0, s, D, cnt);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipMemsetAsync(cnt + L, 0, 4, s));
    size_t tb = tbytes;
    SV_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(temp, tb, cnt, off, L + 1, s));
    if (total > 0) {
        hipLaunchKernelGGL(k_pair_emit, dim3((L + 255) / 256), dim3(256), 0, s, D, off, keys_in, vals_in);
        int bits = 1;
        while ((1u << bits) < (unsigned)nb_dense + 1u && bits < 32) ++bits;
        tb = tbytes;
        SV_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(temp, tb, keys_in, keys_out, vals_in, reinterpret_cast<unsigned long long*>(pairs_out),
                                                       total, 0, bits, s));
    }
    hipLaunchKernelGGL(k_pair_offsets, dim3((nb_dense + 256) / 256), dim3(256), 0, s, keys_out, total, nb_dense, dense_off_dev);
    return SVGPU_OK;
}
