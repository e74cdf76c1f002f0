// (edge, edge) pair lists of the reduced camera system, built on the device.
//
// S_ab -= sum over landmarks l and observation pairs (i, j) of l with pose slots (a, b):  Y_i W_j^T.  k_ba_schur walks, per
// upper block (a <= b), the list of such pairs in a FIXED order (fixed-order fp64 sums = bit-reproducible results).  The lists
// used to be built on the host (two passes over ~200 k pairs, 1.2 ms at 20 KF / 10 k landmarks); here every landmark emits its
// pairs in the same sequence (i <= j in edge order, swapped so that a <= b, the mirrored pair right after a same-pose pair),
// keyed by the dense block index, and a STABLE least-significant-digit radix sort (below: 6-bit digits, hand-written -- per-block digit
// histograms, one scan, a scatter with in-block stable ranks from wave ballots) groups them by block: the order inside a block is the
// landmark-major order the host pass produced.

#include <algorithm>
#include "svgpu_internal.h"
#include "ba_kernels.h"
#include "sv_sort.h"

namespace {

__device__ __forceinline__ bool edge_live(const BaDev& D, int e) { return !D.e_level[e] && D.pose_slot[D.e_pose[e]] >= 0; }
__device__ __forceinline__ unsigned dense_block(int a, int b, int nP) { return (unsigned)(a * nP - a * (a - 1) / 2 + (b - a)); }  // a <= b

// One thread per ROW of a landmark's pair triangle (= per observation i: its pairs (i, j), j >= i in edge order), counts then slots from a
// scan over the observations.  (Round 2 / 3 ran both kernels with one thread per LANDMARK: a chain of k (k + 1) / 2 ~ 21 dependent
// iterations per thread, 124 us for the emission at config 5 whatever its loads and stores were made of -- staged stores and a precomputed
// slot word per edge were both measured and changed nothing.)  The output order is unchanged: rows in edge order, a row's pairs in j order.
__global__ void k_pair_count(BaDev D, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.E) return;
    int n = 0;
    const int l = D.e_point[i];
    if (D.pt_free[l] && edge_live(D, i)) {
        const int hi = D.lm_off[l + 1], a = D.pose_slot[D.e_pose[i]];
        n = 1;  // (i, i)
        for (int j = i + 1; j < hi; ++j) {
            if (!edge_live(D, j)) continue;
            n += D.pose_slot[D.e_pose[j]] == a ? 2 : 1;
        }
    }
    cnt[i] = n;
}

__global__ void k_pair_emit(BaDev D, const int* __restrict__ off, unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.E) return;
    const int l = D.e_point[i];
    if (!D.pt_free[l] || !edge_live(D, i)) return;
    int o = off[i];
    const int hi = D.lm_off[l + 1], ai = D.pose_slot[D.e_pose[i]];
    for (int j = i; j < hi; ++j) {
        if (j != i && !edge_live(D, j)) continue;
        int e1 = i, e2 = j, a = ai, b = j == i ? ai : D.pose_slot[D.e_pose[j]];
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

// dense_off[k] = first sorted position with key >= k, k = 0 .. nb_dense
__global__ void k_pair_offsets(const unsigned* __restrict__ keys, const int* __restrict__ n_dev, int n_host, int nb_dense, int* __restrict__ dense_off) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, n = n_dev ? *n_dev : n_host;
    if (k > nb_dense) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned)k) lo = mid + 1;
        else hi = mid;
    }
    dense_off[k] = lo;
}

// landmark of every sorted pair: what the Schur kernel needs first (Hll), without the hop through e_point[pair.x]
__global__ void k_pair_landmark(const unsigned long long* __restrict__ vals, const int* __restrict__ n_dev, int n_host, const int* __restrict__ e_point, int* __restrict__ pair_l) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = n_dev ? *n_dev : n_host;
    if (i < n) pair_l[i] = e_point[(int)(unsigned)vals[i]];
}
__global__ void k_iota_u64(unsigned long long* __restrict__ v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (unsigned long long)(unsigned)i;
}
__global__ void k_copy_keys(const int* __restrict__ src, int n, unsigned* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (unsigned)src[i];
}
__global__ void k_narrow_u64(const unsigned long long* __restrict__ v, int n, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)(unsigned)v[i];
}
// everything the set-up does once per observation, in one launch: the robust-kernel flag, the (pose, index) sort records, level and cached
// chi2 cleared (five launches -- three kernels and two memsets -- at ~5 us each before)
__global__ void k_obs_prepare(const int* __restrict__ e_pose, int n, unsigned* __restrict__ keys,
                              unsigned long long* __restrict__ vals, uint8_t* __restrict__ e_level, double* __restrict__ e_chi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (unsigned)e_pose[i];
    vals[i] = (unsigned long long)(unsigned)i;
    e_level[i] = 0;
    e_chi[i] = 0.0;
}
// the pose-major copies straight from the sorted records (position q of the pose -> edge lists): pe_idx and the observation it points at
__global__ void k_pose_major_sorted(const unsigned long long* __restrict__ vals, const int* __restrict__ e_point, const float* __restrict__ e_uvr,
                                    const float* __restrict__ e_w, const float* __restrict__ e_hub, int E, int* __restrict__ pe_idx, int* __restrict__ pm_point,
                                    float* __restrict__ pm_uvr, float* __restrict__ pm_w, float* __restrict__ pm_hub, uint8_t* __restrict__ robust) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= E) return;
    const int e = (int)(unsigned)vals[q];
    pe_idx[q] = e;
    robust[e] = e_hub[e] > 0.f ? 1 : 0;  // (every edge is at exactly one position of the pose -> edge lists)
    pm_point[q] = e_point[e];
    pm_uvr[3 * (size_t)q] = e_uvr[3 * (size_t)e];
    pm_uvr[3 * (size_t)q + 1] = e_uvr[3 * (size_t)e + 1];
    pm_uvr[3 * (size_t)q + 2] = e_uvr[3 * (size_t)e + 2];
    pm_w[q] = e_w[e];
    pm_hub[q] = e_hub[e];
}
__global__ void k_flag_positive(const float* __restrict__ x, int n, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] > 0.f ? 1 : 0;
}

__global__ void k_pose_major(const int* __restrict__ pe_idx, const int* __restrict__ e_point, const float* __restrict__ e_uvr, const float* __restrict__ e_w,
                             const float* __restrict__ e_hub, int E, int* __restrict__ pm_point, float* __restrict__ pm_uvr, float* __restrict__ pm_w,
                             float* __restrict__ pm_hub) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= E) return;
    const int e = pe_idx[q];
    pm_point[q] = e_point[e];
    pm_uvr[3 * (size_t)q] = e_uvr[3 * (size_t)e];
    pm_uvr[3 * (size_t)q + 1] = e_uvr[3 * (size_t)e + 1];
    pm_uvr[3 * (size_t)q + 2] = e_uvr[3 * (size_t)e + 2];
    pm_w[q] = e_w[e];
    pm_hub[q] = e_hub[e];
}

inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }
}  // namespace

size_t sv_ba_pairs_scratch_bytes(size_t pair_cap, int E, size_t nb_cap) {  // E: observations (the counts / slots are per observation)
    return 2 * pad256((size_t)(E + 2) * 4) + pad256(sv_scan_scratch_ints((size_t)E + 1) * 4) + pad256(sv_sort_hist_ints(pair_cap) * 4) + 2 * pad256(pair_cap * 4) + pad256(pair_cap * 8) + pad256((nb_cap + 1) * 4) + 1024;
}

namespace {
// count -> scan -> emit -> radix passes -> dense offsets; total_host < 0: the pair total stays on the device (no read-back)
int build_pairs(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int total_host, int2* pairs_out,
                int* pair_l_out, int* dense_off_dev, bool read_total, int* total_out) {
    const int nb_dense = D.nP * (D.nP + 1) / 2;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += pad256(bytes);
        return (void*)r;
    };
    const int E = D.E;
    int* cnt = (int*)take((size_t)(E + 2) * 4);
    int* cnt_scan = (int*)take(sv_scan_scratch_ints((size_t)E + 1) * 4);
    int* hist = (int*)take(sv_sort_hist_ints(pair_cap) * 4);
    unsigned* keys[2] = {(unsigned*)take(pair_cap * 4), (unsigned*)take(pair_cap * 4)};
    unsigned long long* vals[2] = {(unsigned long long*)take(pair_cap * 8), reinterpret_cast<unsigned long long*>(pairs_out)};
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pair-list scratch too small");
    SV_HIP(ctx, hipGetLastError());
    if (E > 0) {
        hipLaunchKernelGGL(k_pair_count, dim3((E + 255) / 256), dim3(256), 0, s, D, cnt);
        sv_scan_i32(s, cnt, E, cnt_scan);  // cnt[i] = first pair of row (observation) i, cnt[E] = the total
    }
    else SV_HIP(ctx, hipMemsetAsync(cnt, 0, 8, s));  // (a rank of a sharded solve without observations)
    SV_HIP(ctx, hipGetLastError());
    int total = total_host;
    if (read_total) {
        SV_HIP(ctx, hipMemcpyAsync(&total, cnt + E, 4, hipMemcpyDeviceToHost, s));
        SV_HIP(ctx, hipStreamSynchronize(s));
        if ((size_t)total > pair_cap) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pair-list capacity exceeded");
        if (total_out) *total_out = total;
    }
    int bits = 1;
    while ((1u << bits) < (unsigned)nb_dense + 1u && bits < 32) ++bits;
    int cur = sv_sort_passes(bits) & 1 ? 0 : 1;  // so that the last pass lands in buffer 1 = pairs_out
    // total < 0 (and nothing read back): the pair total stays ON the device -- cnt[L], the scan's total -- and the launches are sized for the
    // capacity; the host pass that used to count the pairs of a local-BA sized problem left the device idle for ~0.1 ms
    const int* const n_dev = total < 0 ? cnt + E : nullptr;
    const int n_launch = total < 0 ? (int)std::min<size_t>(pair_cap, 0x7fffffff) : total;
    if (n_launch > 0) {
        if (E > 0) hipLaunchKernelGGL(k_pair_emit, dim3((E + 255) / 256), dim3(256), 0, s, D, cnt, keys[cur], vals[cur]);
        cur = sv_sort_pairs(s, keys, vals, cur, n_launch, bits, hist, n_dev);
        hipLaunchKernelGGL(k_pair_landmark, dim3((n_launch + 255) / 256), dim3(256), 0, s, vals[cur], n_dev, n_launch, D.e_point, pair_l_out);
    }
    hipLaunchKernelGGL(k_pair_offsets, dim3((nb_dense + 256) / 256), dim3(256), 0, s, keys[cur], n_dev, n_launch, nb_dense, dense_off_dev);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}
}  // namespace

// D.pose_slot / D.pt_free / D.e_level / D.nP must be current on the device.  Writes the sorted pairs to `pairs_out` (= D.blk_pairs
// storage) and the dense block offsets (nb_dense + 1 ints) to `dense_off_host`.  Synchronises the stream twice.
int sv_ba_build_pairs(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int2* pairs_out,
                      int* pair_l_out, std::vector<int>& dense_off_host) {
    const int L = D.L, nb_dense = D.nP * (D.nP + 1) / 2;
    dense_off_host.assign((size_t)nb_dense + 1, 0);
    if (L == 0 || D.nP == 0) return SVGPU_OK;
    // the dense offsets live at the END of the scratch block (behind what build_pairs takes)
    int* dense_off = (int*)((char*)scratch + scratch_bytes - pad256(((size_t)nb_dense + 1) * 4));
    const int rc = build_pairs(ctx, s, D, scratch, scratch_bytes - pad256(((size_t)nb_dense + 1) * 4), pair_cap, 0, pairs_out, pair_l_out, dense_off, true, nullptr);
    if (rc) return rc;
    SV_HIP(ctx, hipMemcpyAsync(dense_off_host.data(), dense_off, 4 * ((size_t)nb_dense + 1), hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

// Same pipeline with the pair total known to the caller (svgpu_ba.hip: host_pair_total): nothing is read back, nothing synchronises.
// The dense block offsets (nP (nP + 1) / 2 + 1 ints) are written to `dense_off_dev`.
int sv_ba_build_pairs_async(svgpu_ctx* ctx, hipStream_t s, const BaDev& D, void* scratch, size_t scratch_bytes, size_t pair_cap, int total,
                            int2* pairs_out, int* pair_l_out, int* dense_off_dev) {
    const int L = D.L, nb_dense = D.nP * (D.nP + 1) / 2;
    if (L == 0 || D.nP == 0) {
        SV_HIP(ctx, hipMemsetAsync(dense_off_dev, 0, 4 * ((size_t)nb_dense + 1), s));
        return SVGPU_OK;
    }
    return build_pairs(ctx, s, D, scratch, scratch_bytes, pair_cap, total, pairs_out, pair_l_out, dense_off_dev, false, nullptr);
}

// Pose -> edge lists (every pose's edges in increasing edge order, whatever their level) and the "has a robust kernel" flags, built on
// the device from the uploaded observations: a stable radix sort of (pose, edge index) + the offsets of the sorted keys.  The host used to
// do this with a two-pass counting sort over all observations (0.5 ms of a config-5 call on four threads).
size_t sv_ba_pose_lists_scratch_bytes(size_t E) { return 2 * pad256(E * 4) + 2 * pad256(E * 8) + pad256(sv_sort_hist_ints(E) * 4) + 1024; }
void sv_ba_build_pose_major(hipStream_t s, const int* pe_idx, const int* e_point, const float* e_uvr, const float* e_w, const float* e_hub, int E, int* pm_point,
                            float* pm_uvr, float* pm_w, float* pm_hub) {
    if (E > 0) hipLaunchKernelGGL(k_pose_major, dim3((E + 255) / 256), dim3(256), 0, s, pe_idx, e_point, e_uvr, e_w, e_hub, E, pm_point, pm_uvr, pm_w, pm_hub);
}
// The pose -> edge lists, the robust flags, the cleared level / chi2 arrays and the pose-major observation copies in 2 + 3 per radix pass + 1
// launches (sv_ba_build_pose_lists + sv_ba_build_pose_major + two memsets took 5 more).  Two halves, so that a global-BA sized call can
// run the first one -- it needs the observations' POSE INDICES only -- while the measurements are still on their way to the device:
//   sv_ba_prepare_lists       level / chi2 cleared, edges sorted by pose, pe_off; *sorted_out = the sorted (pose, edge) records, which stay
//                             in `scratch` until the second half has run
//   sv_ba_prepare_pose_major  pe_idx, the pose-major copies of point / measurement / information / kernel width, the robust flags
int sv_ba_prepare_lists(svgpu_ctx* ctx, hipStream_t s, const int* e_pose_dev, int E, int P, void* scratch, size_t scratch_bytes, int* pe_off_dev, uint8_t* e_level_dev,
                        double* e_chi_dev, const void** sorted_out) {
    *sorted_out = nullptr;
    if (E <= 0) {
        SV_HIP(ctx, hipMemsetAsync(pe_off_dev, 0, 4 * ((size_t)P + 1), s));
        return SVGPU_OK;
    }
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += pad256(bytes);
        return (void*)r;
    };
    unsigned* keys[2] = {(unsigned*)take((size_t)E * 4), (unsigned*)take((size_t)E * 4)};
    unsigned long long* vals[2] = {(unsigned long long*)take((size_t)E * 8), (unsigned long long*)take((size_t)E * 8)};
    int* hist = (int*)take(sv_sort_hist_ints(E) * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pose-list scratch too small");
    const dim3 g((E + 255) / 256), b(256);
    hipLaunchKernelGGL(k_obs_prepare, g, b, 0, s, e_pose_dev, E, keys[0], vals[0], e_level_dev, e_chi_dev);
    int bits = 1;
    while ((1u << bits) < (unsigned)P && bits < 31) ++bits;
    const int r = sv_sort_pairs(s, keys, vals, 0, E, bits, hist);
    hipLaunchKernelGGL(k_pair_offsets, dim3((P + 256) / 256), dim3(256), 0, s, keys[r], (const int*)nullptr, E, P, pe_off_dev);
    SV_HIP(ctx, hipGetLastError());
    *sorted_out = vals[r];
    return SVGPU_OK;
}
int sv_ba_prepare_pose_major(svgpu_ctx* ctx, hipStream_t s, const void* sorted, const int* e_point_dev, const float* e_uvr_dev, const float* e_w_dev, const float* e_huber_dev,
                             int E, int* pe_idx_dev, uint8_t* robust_dev, int* pm_point, float* pm_uvr, float* pm_w, float* pm_hub) {
    if (E <= 0 || !sorted) return SVGPU_OK;
    hipLaunchKernelGGL(k_pose_major_sorted, dim3((E + 255) / 256), dim3(256), 0, s, (const unsigned long long*)sorted, e_point_dev, e_uvr_dev, e_w_dev, e_huber_dev, E, pe_idx_dev,
                       pm_point, pm_uvr, pm_w, pm_hub, robust_dev);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}
int sv_ba_build_pose_lists(svgpu_ctx* ctx, hipStream_t s, const int* e_pose_dev, const float* e_huber_dev, int E, int P, void* scratch, size_t scratch_bytes,
                           int* pe_off_dev, int* pe_idx_dev, uint8_t* robust_dev) {
    if (E <= 0) {
        SV_HIP(ctx, hipMemsetAsync(pe_off_dev, 0, 4 * ((size_t)P + 1), s));
        return SVGPU_OK;
    }
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += pad256(bytes);
        return (void*)r;
    };
    unsigned* keys[2] = {(unsigned*)take((size_t)E * 4), (unsigned*)take((size_t)E * 4)};
    unsigned long long* vals[2] = {(unsigned long long*)take((size_t)E * 8), (unsigned long long*)take((size_t)E * 8)};
    int* hist = (int*)take(sv_sort_hist_ints(E) * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "pose-list scratch too small");
    const dim3 g((E + 255) / 256), b(256);
    hipLaunchKernelGGL(k_flag_positive, g, b, 0, s, e_huber_dev, E, robust_dev);
    hipLaunchKernelGGL(k_copy_keys, g, b, 0, s, e_pose_dev, E, keys[0]);
    hipLaunchKernelGGL(k_iota_u64, g, b, 0, s, vals[0], E);
    int bits = 1;
    while ((1u << bits) < (unsigned)P && bits < 31) ++bits;
    const int r = sv_sort_pairs(s, keys, vals, 0, E, bits, hist);
    hipLaunchKernelGGL(k_narrow_u64, g, b, 0, s, vals[r], E, pe_idx_dev);
    hipLaunchKernelGGL(k_pair_offsets, dim3((P + 256) / 256), dim3(256), 0, s, keys[r], (const int*)nullptr, E, P, pe_off_dev);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}

// ---- Landmark renumbering at global-BA sizes.  W, Hll and bl are stored in landmark order, and the Schur products of a block row (a, .)
// gather the records of the landmarks keyframe a sees: if the caller numbers its landmarks in the order a map creates them, those are a
// contiguous stretch of W that stays in the L2 / the Infinity Cache while the row is worked on; if it numbers them at random (the bench
// scene does; the reference's unordered_map walk does to a degree) every row gathers from all of W -- at 9.6 M observations the kernel
// then fetches 8.5 x its records from HBM.  The solve therefore runs on its OWN numbering: rank of the landmark in a stable sort by
// first observing keyframe (empty landmarks last).  Edges, measurements, positions and activity flags are permuted once at set-up;
// k_ba_pack_out writes the positions back in the caller's order.
__global__ void k_lm_key(const int* __restrict__ lm_off, const int* __restrict__ e_pose, int L, int P, unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    int k = P;
    for (int e = lm_off[l]; e < lm_off[l + 1]; ++e) k = min(k, e_pose[e]);
    keys[l] = (unsigned)k;
    vals[l] = (unsigned long long)(unsigned)l;
}
__global__ void k_lm_counts(const unsigned long long* __restrict__ vals, const int* __restrict__ lm_off, int L, int* __restrict__ order, int* __restrict__ cnt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= L) return;
    const int l = (int)(unsigned)vals[r];
    order[r] = l;
    cnt[r] = lm_off[l + 1] - lm_off[l];
}
__global__ void k_lm_permute_idx(const int* __restrict__ order, const int* __restrict__ lm_off_old, const int* __restrict__ lm_off_new, const int* __restrict__ e_pose_old,
                                 int L, int* __restrict__ e_pose_new, int* __restrict__ e_point_new, int* __restrict__ src_edge) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 3, sub = t & 7;
    if (r >= L) return;
    const int l = order[r], o0 = lm_off_old[l], n = lm_off_old[l + 1] - o0, n0 = lm_off_new[r];
    for (int k = sub; k < n; k += 8) {
        e_pose_new[n0 + k] = e_pose_old[o0 + k];
        e_point_new[n0 + k] = r;
        src_edge[n0 + k] = o0 + k;
    }
}
__global__ void k_lm_permute_meas(const int* __restrict__ src_edge, int E, const float* __restrict__ uvr_old, const float* __restrict__ w_old, const float* __restrict__ hub_old,
                                  float* __restrict__ uvr_new, float* __restrict__ w_new, float* __restrict__ hub_new) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int o = src_edge[e];
    uvr_new[3 * (size_t)e] = uvr_old[3 * (size_t)o];
    uvr_new[3 * (size_t)e + 1] = uvr_old[3 * (size_t)o + 1];
    uvr_new[3 * (size_t)e + 2] = uvr_old[3 * (size_t)o + 2];
    w_new[e] = w_old[o];
    hub_new[e] = hub_old[o];
}
__global__ void k_lm_permute_points(const int* __restrict__ order, int L, const double* __restrict__ pts_old, double* __restrict__ pts_new) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= L) return;
    const int l = order[r];
    pts_new[3 * (size_t)r] = pts_old[3 * (size_t)l];
    pts_new[3 * (size_t)r + 1] = pts_old[3 * (size_t)l + 1];
    pts_new[3 * (size_t)r + 2] = pts_old[3 * (size_t)l + 2];
}
__global__ void k_lm_permute_flags(const int* __restrict__ order, int L, const uint8_t* __restrict__ old_flags, uint8_t* __restrict__ new_flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < L) new_flags[r] = old_flags[order[r]];
}
size_t sv_ba_renumber_scratch_bytes(size_t L) { return 2 * pad256(L * 4) + 2 * pad256(L * 8) + pad256(sv_sort_hist_ints(L) * 4) + pad256(sv_scan_scratch_ints(L + 1) * 4 + 64) + 1024; }
// order_out[r] = caller's index of the landmark at rank r; lm_off_new (L + 1), e_pose_new / e_point_new / src_edge (E) in the new edge order
int sv_ba_renumber_landmarks(svgpu_ctx* ctx, hipStream_t s, const int* lm_off_old, const int* e_pose_old, int L, int P, int E, void* scratch, size_t scratch_bytes,
                             int* order_out, int* lm_off_new, int* e_pose_new, int* e_point_new, int* src_edge) {
    if (L <= 0 || E <= 0) return SVGPU_OK;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += pad256(bytes);
        return (void*)r;
    };
    unsigned* keys[2] = {(unsigned*)take((size_t)L * 4), (unsigned*)take((size_t)L * 4)};
    unsigned long long* vals[2] = {(unsigned long long*)take((size_t)L * 8), (unsigned long long*)take((size_t)L * 8)};
    int* hist = (int*)take(sv_sort_hist_ints(L) * 4);
    int* scan_scr = (int*)take(sv_scan_scratch_ints((size_t)L + 1) * 4 + 64);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "landmark renumbering scratch too small");
    const dim3 g((L + 255) / 256), b(256);
    hipLaunchKernelGGL(k_lm_key, g, b, 0, s, lm_off_old, e_pose_old, L, P, keys[0], vals[0]);
    int bits = 1;
    while ((1u << bits) < (unsigned)(P + 1) && bits < 31) ++bits;
    const int r = sv_sort_pairs(s, keys, vals, 0, L, bits, hist);
    hipLaunchKernelGGL(k_lm_counts, g, b, 0, s, vals[r], lm_off_old, L, order_out, lm_off_new);
    SV_HIP(ctx, hipMemsetAsync(lm_off_new + L, 0, 4, s));
    sv_scan_i32(s, lm_off_new, L, scan_scr);  // exclusive, total -> lm_off_new[L]
    hipLaunchKernelGGL(k_lm_permute_idx, dim3((unsigned)(((size_t)L * 8 + 255) / 256)), b, 0, s, order_out, lm_off_old, lm_off_new, e_pose_old, L, e_pose_new, e_point_new, src_edge);
    SV_HIP(ctx, hipGetLastError());
    return SVGPU_OK;
}
void sv_ba_permute_measurements(hipStream_t s, const int* src_edge, int E, const float* uvr_old, const float* w_old, const float* hub_old, float* uvr_new, float* w_new, float* hub_new) {
    if (E > 0) hipLaunchKernelGGL(k_lm_permute_meas, dim3((E + 255) / 256), dim3(256), 0, s, src_edge, E, uvr_old, w_old, hub_old, uvr_new, w_new, hub_new);
}
void sv_ba_permute_points(hipStream_t s, const int* order, int L, const double* pts_old, double* pts_new) {
    if (L > 0) hipLaunchKernelGGL(k_lm_permute_points, dim3((L + 255) / 256), dim3(256), 0, s, order, L, pts_old, pts_new);
}
void sv_ba_permute_flags(hipStream_t s, const int* order, int L, const uint8_t* old_flags, uint8_t* new_flags) {
    if (L > 0) hipLaunchKernelGGL(k_lm_permute_flags, dim3((L + 255) / 256), dim3(256), 0, s, order, L, old_flags, new_flags);
}

// ---- Chunk-major units of the Schur kernel (global-BA sizes, renumbered landmarks).  The pair list is sorted by (block, landmark rank); it is
// cut wherever the block or the landmark CHUNK (rank >> shift) changes.  A unit then gathers W records of one chunk only, and executing the
// units chunk by chunk keeps that stretch of W (~1 MB) in the XCD's L2 while all its pairs -- in every block they fall into -- are worked on:
// a record is fetched once per chunk instead of once per pair.  The block sums stay fixed-order: k_ba_sys_fin adds a block's units in chunk order.
__global__ void k_unit_mark_chunk(const int* __restrict__ pair_l, int n, int shift, int* __restrict__ flag) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) flag[q] = (q == 0 || (pair_l[q] >> shift) != (pair_l[q - 1] >> shift)) ? 1 : 0;
}
__global__ void k_unit_mark_blk(const int* __restrict__ blk_off, int NB, int n, int* __restrict__ flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < NB) {
        const int q = blk_off[b];
        if (q < n) flag[q] = 1;
    }
}
__global__ void k_unit_offsets(const int* __restrict__ uid, int n, int* __restrict__ unit_off) {  // uid = exclusive scan of the flags, total at [n]
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n && uid[q + 1] != uid[q]) unit_off[uid[q]] = q;
    if (q == 0) unit_off[uid[n]] = n;
}
__global__ void k_unit_blk_off(const int* __restrict__ blk_off, int NB, const int* __restrict__ uid, int* __restrict__ blk_unit_off) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= NB) blk_unit_off[b] = uid[blk_off[b]];
}
__global__ void k_unit_blk(const int* __restrict__ blk_unit_off, int NB, int* __restrict__ unit_blk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    for (int u = blk_unit_off[b]; u < blk_unit_off[b + 1]; ++u) unit_blk[u] = b;
}
__global__ void k_unit_keys(const int* __restrict__ unit_off, const int* __restrict__ pair_l, int U, int shift, unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    keys[u] = (unsigned)(pair_l[unit_off[u]] >> shift);
    vals[u] = (unsigned long long)(unsigned)u;
}
__global__ void k_unit_rec(const unsigned long long* __restrict__ vals, const int* __restrict__ unit_off, const int* __restrict__ unit_blk, int U, int4* __restrict__ rec) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= U) return;
    const int u = (int)(unsigned)vals[pos];
    rec[pos] = make_int4(unit_off[u], unit_off[u + 1], unit_blk[u], u);
}
size_t sv_ba_units_scratch_bytes(size_t num_pairs, size_t unit_cap) {
    return pad256((num_pairs + 2) * 4) + pad256(sv_scan_scratch_ints(num_pairs + 1) * 4 + 64) + pad256((unit_cap + 1) * 4) + pad256(unit_cap * 4) + 2 * pad256(unit_cap * 4) + 2 * pad256(unit_cap * 8)
           + pad256(sv_sort_hist_ints(unit_cap) * 4) + 2048;
}
// *num_units_out = 0 when the cut would need more than unit_cap units (the caller keeps the arithmetic shares).  One host synchronisation.
int sv_ba_build_units(svgpu_ctx* ctx, hipStream_t s, const int* blk_off_dev, int NB, const int* pair_l_dev, int num_pairs, int L, int chunk_shift, int unit_cap, void* scratch,
                      size_t scratch_bytes, int4* unit_rec_out, int* blk_unit_off_out, int* num_units_out) {
    *num_units_out = 0;
    if (num_pairs <= 0 || NB <= 0 || unit_cap <= 0) return SVGPU_OK;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += pad256(bytes);
        return (void*)r;
    };
    int* uid = (int*)take(((size_t)num_pairs + 2) * 4);
    int* scan_scr = (int*)take(sv_scan_scratch_ints((size_t)num_pairs + 1) * 4 + 64);
    int* unit_off = (int*)take(((size_t)unit_cap + 1) * 4);
    int* unit_blk = (int*)take((size_t)unit_cap * 4);
    unsigned* keys[2] = {(unsigned*)take((size_t)unit_cap * 4), (unsigned*)take((size_t)unit_cap * 4)};
    unsigned long long* vals[2] = {(unsigned long long*)take((size_t)unit_cap * 8), (unsigned long long*)take((size_t)unit_cap * 8)};
    int* hist = (int*)take(sv_sort_hist_ints((size_t)unit_cap) * 4);
    if ((size_t)(p - (char*)scratch) > scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "unit scratch too small");
    const dim3 b(256), gq((num_pairs + 255) / 256);
    hipLaunchKernelGGL(k_unit_mark_chunk, gq, b, 0, s, pair_l_dev, num_pairs, chunk_shift, uid);
    hipLaunchKernelGGL(k_unit_mark_blk, dim3((NB + 255) / 256), b, 0, s, blk_off_dev, NB, num_pairs, uid);
    SV_HIP(ctx, hipMemsetAsync(uid + num_pairs, 0, 4, s));
    sv_scan_i32(s, uid, num_pairs, scan_scr);
    int U = 0;
    SV_HIP(ctx, hipMemcpyAsync(&U, uid + num_pairs, 4, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    if (U <= 0 || U > unit_cap) return SVGPU_OK;
    hipLaunchKernelGGL(k_unit_offsets, gq, b, 0, s, uid, num_pairs, unit_off);
    hipLaunchKernelGGL(k_unit_blk_off, dim3((NB + 256) / 256), b, 0, s, blk_off_dev, NB, uid, blk_unit_off_out);
    hipLaunchKernelGGL(k_unit_blk, dim3((NB + 255) / 256), b, 0, s, blk_unit_off_out, NB, unit_blk);
    const dim3 gu((U + 255) / 256);
    hipLaunchKernelGGL(k_unit_keys, gu, b, 0, s, unit_off, pair_l_dev, U, chunk_shift, keys[0], vals[0]);
    int bits = 1;
    while ((1u << bits) < (unsigned)((L >> chunk_shift) + 1) && bits < 31) ++bits;
    const int r = sv_sort_pairs(s, keys, vals, 0, U, bits, hist);
    hipLaunchKernelGGL(k_unit_rec, gu, b, 0, s, vals[r], unit_off, unit_blk, U, unit_rec_out);
    SV_HIP(ctx, hipGetLastError());
    *num_units_out = U;
    return SVGPU_OK;
}

// ---- svgpu_selftest_scan_sort (include/svgpu.h): the scan and the radix sort on caller data
extern "C" int svgpu_selftest_scan_sort(svgpu_ctx* ctx, int n, const int32_t* values, int32_t* scan_out, const uint32_t* keys, int bits, int32_t* sorted_idx) {
    if (!ctx || n < 0 || bits < 1 || bits > 32 || (scan_out && !values) || (sorted_idx && !keys)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_selftest_scan_sort: bad arguments");
    if (n == 0) {
        if (scan_out) scan_out[0] = 0;
        return SVGPU_OK;
    }
    hipStream_t s = ctx->stream;
    const size_t N = (size_t)n;
    const size_t bytes = pad256((N + 2) * 4) + pad256(sv_scan_scratch_ints(N + 1) * 4 + 64) + 2 * pad256(N * 4) + 2 * pad256(N * 8) + pad256(sv_sort_hist_ints(N) * 4) + pad256(N * 4) + 4096;
    int rc = sv_ensure_scratch(ctx, bytes);
    if (rc) return rc;
    char* p = (char*)ctx->d_scratch;
    auto take = [&](size_t b) {
        char* r = p;
        p += pad256(b);
        return (void*)r;
    };
    int* d_scan = (int*)take((N + 2) * 4);
    int* d_scan_scr = (int*)take(sv_scan_scratch_ints(N + 1) * 4 + 64);
    unsigned* k2[2] = {(unsigned*)take(N * 4), (unsigned*)take(N * 4)};
    unsigned long long* v2[2] = {(unsigned long long*)take(N * 8), (unsigned long long*)take(N * 8)};
    int* hist = (int*)take(sv_sort_hist_ints(N) * 4);
    int* d_idx = (int*)take(N * 4);
    if (scan_out) {
        SV_HIP(ctx, hipMemcpyAsync(d_scan, values, N * 4, hipMemcpyHostToDevice, s));
        sv_scan_i32(s, d_scan, n, d_scan_scr);
        SV_HIP(ctx, hipMemcpyAsync(scan_out, d_scan, (N + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    if (sorted_idx) {
        SV_HIP(ctx, hipMemcpyAsync(k2[0], keys, N * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_iota_u64, dim3((n + 255) / 256), dim3(256), 0, s, v2[0], n);
        const int r = sv_sort_pairs(s, k2, v2, 0, n, bits, hist);
        hipLaunchKernelGGL(k_narrow_u64, dim3((n + 255) / 256), dim3(256), 0, s, v2[r], n, d_idx);
        SV_HIP(ctx, hipMemcpyAsync(sorted_idx, d_idx, N * 4, hipMemcpyDeviceToHost, s));
    }
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}
