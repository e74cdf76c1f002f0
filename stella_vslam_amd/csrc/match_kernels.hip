// HIP kernels of the Hamming matchers for gfx950 (wave64).
//   k_hamming_pairs / k_hamming_matrix   compute_descriptor_distance_32 (match/base.h:20-41); xor + v_bcnt_u32
//   k_bf_topk + k_bf_replay              robust::brute_force_match (match/robust.cc:232-328)
//   k_cand_dist + k_cand_replay          projection::match_frame_and_landmarks / match_current_and_last_frames
//                                        (match/projection.cc:13-207) on CSR candidate lists
//
// Greedy replay.  The reference loops are sequential: query k may not take a target that an earlier
// query already took (already_matched_indices_1 / frm.add_landmark).  The distance work is done in
// parallel; the sequential dependency is resolved exactly by a fixed-point iteration:
//     decision(q) = f(own candidates, { targets claimed by queries q' < q })
// is re-evaluated for all q in parallel with `owner[t] = min{ q' : claim(q') = t }` taken from the previous
// sweep.  After sweep s the first s queries are final, so any fixed point IS the serial result; real data
// converges in a handful of sweeps.  For brute force the candidates of a query are its K smallest
// (distance, index) pairs; the rare case where that prefix cannot decide falls back to a full scan.
#include "svgpu_internal.h"
#include "match_kernels.h"

namespace {

constexpr unsigned HAMMING_DIST_THR_LOW = 50;   // match/base.h:15
constexpr unsigned MAX_HAMMING_DIST = 256;      // match/base.h:17

__device__ __forceinline__ unsigned hamming256(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
    unsigned d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += __popc(a[i] ^ b[i]);
    return d;
}

__device__ __forceinline__ float angle_diff(float a1, float a2) {  // util/angle.cc:7-16
    float ret = a1 - a2;
    if (ret <= -180.0f) ret += 360.0f;
    if (ret > 180.0f) ret -= 360.0f;
    return ret;
}

__global__ void k_hamming_pairs(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, int n, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = a[(size_t)i * 8 + k];
    out[i] = hamming256(q, b + (size_t)i * 8);
}

// out[j * n1 + i] = d(desc2[j], desc1[i]); 256 rows of desc1 staged per LDS tile
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint32_t* __restrict__ d1, int n1, const uint32_t* __restrict__ d2,
                                                        int n2, uint16_t* __restrict__ out) {
    __shared__ uint32_t s_t[256 * 8];
    const int j = blockIdx.x * 256 + threadIdx.x;
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j < n2)
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = d2[(size_t)j * 8 + k];
    for (int base = 0; base < n1; base += 256) {
        const int m = min(256, n1 - base);
        __syncthreads();
        for (int i = threadIdx.x; i < m * 8; i += 256) s_t[i] = d1[(size_t)base * 8 + i];
        __syncthreads();
        if (j < n2)
            for (int i = 0; i < m; ++i) out[(size_t)j * n1 + base + i] = (uint16_t)hamming256(q, &s_t[i * 8]);
    }
}

// ------------------------------------------------------------------------------------------------ brute force
// Block = 64 queries (keyframe keypoints idx_2) x 4 candidate quarters: wave w scans quarter w of the frame
// keypoints idx_1 for its 64 queries.  The candidate index is wave-uniform, so candidate descriptors and
// angles arrive through the scalar cache (s_load) and never touch LDS; each lane keeps the BF_K smallest
// (dist << 16 | idx_1) keys of its quarter in registers, the four sorted lists are merged through LDS.
// Ascending key order = the reference's scan preference (strict '<' => lowest index wins ties).
__global__ __launch_bounds__(256) void k_bf_topk(BfProblem P) {
    __shared__ uint32_t s_l[4 * 64 * (BF_K + 1)];
    __shared__ int s_c[4 * 64];
    const int pair = blockIdx.y;
    const int n1 = P.n1_dev ? P.n1_dev[pair * P.n_stride] : P.n1;
    const int n2 = P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2;
    const int n1c = min(n1, P.cap1), n2c = min(n2, P.cap2);
    if (blockIdx.x * 64 >= n2c) return;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const uint32_t* __restrict__ D1 = P.desc1 + (size_t)pair * P.cap1 * 8;
    const uint32_t* __restrict__ D2 = P.desc2 + (size_t)pair * P.cap2 * 8;
    const float* __restrict__ A1 = P.angle1 + (size_t)pair * P.cap1 * P.angle_stride;
    const bool active = j < n2c && (!P.valid2 || P.valid2[(size_t)pair * P.cap2 + j]);
    uint32_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float qa = 0.f;
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = D2[(size_t)j * 8 + k];
        qa = P.angle2[((size_t)pair * P.cap2 + j) * P.angle_stride];
    }
    uint32_t L[BF_K];
#pragma unroll
    for (int k = 0; k < BF_K; ++k) L[k] = 0xFFFFFFFFu;
    int cnt = 0;
    const int chunk = (n1c + 3) / 4;
    const int i0 = __builtin_amdgcn_readfirstlane(part * chunk);
    const int i1 = min(n1c, i0 + chunk);
    const bool ori = P.check_orientation != 0;
    // Only candidates with dist <= dmax can influence a decision (best must be <= 50, and the ratio test
    // lowe_ratio * second < best can only reject when second < 50 / lowe_ratio): everything farther is never listed.
    const unsigned dmax = P.dmax;
    auto consider = [&](int i, unsigned d, float ca) {
        if (d <= dmax) {  // rare: wave-level branch is almost never taken for unrelated descriptors
            bool pass = active;
            if (ori) pass = pass && !(fabsf(angle_diff(ca, qa)) > 30.0f);
            if (pass) {
                ++cnt;
                const uint32_t key = (d << 16) | (uint32_t)i;
                if (key < L[BF_K - 1]) {
#pragma unroll
                    for (int k = BF_K - 1; k > 0; --k) L[k] = max(L[k - 1], min(L[k], key));
                    L[0] = min(L[0], key);
                }
            }
        }
    };
    int i = i0;
    for (; i + 4 <= i1; i += 4) {  // 4 candidates per trip: the scalar loads are issued back to back
        uint32_t c[4][8];
        float ca[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[u][k] = D1[(size_t)(i + u) * 8 + k];
            ca[u] = A1[(size_t)(i + u) * P.angle_stride];
        }
        unsigned d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d[u] += __popc(q[k] ^ c[u][k]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) consider(i + u, d[u], ca[u]);
    }
    for (; i < i1; ++i) {
        unsigned d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(q[k] ^ D1[(size_t)i * 8 + k]);
        consider(i, d, A1[(size_t)i * P.angle_stride]);
    }
    // merge the four quarter lists of each query (lane of wave 0 does a 4-way merge of sorted lists)
    uint32_t* mine = &s_l[(part * 64 + lane) * (BF_K + 1)];
#pragma unroll
    for (int k = 0; k < BF_K; ++k) mine[k] = L[k];
    mine[BF_K] = 0xFFFFFFFFu;  // sentinel
    s_c[part * 64 + lane] = cnt;
    __syncthreads();
    if (part == 0 && j < n2c) {
        const uint32_t* l0 = &s_l[(0 * 64 + lane) * (BF_K + 1)];
        const uint32_t* l1 = &s_l[(1 * 64 + lane) * (BF_K + 1)];
        const uint32_t* l2 = &s_l[(2 * 64 + lane) * (BF_K + 1)];
        const uint32_t* l3 = &s_l[(3 * 64 + lane) * (BF_K + 1)];
        int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        uint32_t* T = P.topk + ((size_t)pair * P.cap2 + j) * BF_K;
        for (int k = 0; k < BF_K; ++k) {
            const uint32_t v0 = l0[p0], v1 = l1[p1], v2 = l2[p2], v3 = l3[p3];
            const uint32_t m = min(min(v0, v1), min(v2, v3));
            if (m == 0xFFFFFFFFu) {
                T[k] = m;
                continue;
            }
            if (m == v0) ++p0;
            else if (m == v1) ++p1;
            else if (m == v2) ++p2;
            else ++p3;
            T[k] = m;
        }
        P.cnt[(size_t)pair * P.cap2 + j] = s_c[lane] + s_c[64 + lane] + s_c[128 + lane] + s_c[192 + lane];
    }
}

// decision of one query given the owner table; returns idx_1 or -1
__device__ int bf_decide(const BfProblem& P, int pair, int j, int n1c, const uint32_t* __restrict__ T, int cnt,
                         const int* owner) {
    if (cnt == 0) return -1;  // no candidate within dmax: best > dmax >= 50
    const int m = min(cnt, BF_K);
    const bool truncated = cnt > BF_K;
    uint32_t a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu;
    for (int k = 0; k < m; ++k) {
        const uint32_t key = T[k];
        if (owner[key & 0xFFFFu] < j) continue;  // claimed by an earlier query (already_matched_indices_1)
        if (a0 == 0xFFFFFFFFu) a0 = key;
        else {
            a1 = key;
            break;
        }
    }
    // The list holds the (at most BF_K) nearest of the candidates with dist <= dmax; if it is not truncated,
    // every unlisted candidate is farther than dmax.
    const unsigned d_beyond = truncated ? (T[m - 1] >> 16) : P.dmax + 1;  // lower bound on any unlisted distance
    if (a0 != 0xFFFFFFFFu) {
        const unsigned best = a0 >> 16;
        if (HAMMING_DIST_THR_LOW < best) return -1;
        if (a1 != 0xFFFFFFFFu) return (P.lowe_ratio * (float)(a1 >> 16) < (float)best) ? -1 : (int)(a0 & 0xFFFFu);
        if (P.exhaustive && !truncated) return (P.lowe_ratio * (float)MAX_HAMMING_DIST < (float)best) ? -1 : (int)(a0 & 0xFFFFu);
        // second best is unlisted: its distance is >= d_beyond
        if (P.lowe_ratio >= 0.f && !(P.lowe_ratio * (float)d_beyond < (float)best)) return (int)(a0 & 0xFFFFu);
    }
    else {
        if (d_beyond > HAMMING_DIST_THR_LOW) return -1;  // everything unlisted is at least that far
    }
    // ---- undecidable from the prefix: exact serial scan of this row (robust.cc:271-298)
    const uint32_t* D1 = P.desc1 + (size_t)pair * P.cap1 * 8;
    const uint32_t* D2 = P.desc2 + ((size_t)pair * P.cap2 + j) * 8;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = D2[k];
    const float qa = P.angle2[((size_t)pair * P.cap2 + j) * P.angle_stride];
    unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
    int best_idx = -1;
    for (int i = 0; i < n1c; ++i) {
        if (owner[i] < j) continue;
        if (P.check_orientation && fabsf(angle_diff(P.angle1[((size_t)pair * P.cap1 + i) * P.angle_stride], qa)) > 30.0f) continue;
        const unsigned d = hamming256(q, D1 + (size_t)i * 8);
        if (d < best) {
            second = best;
            best = d;
            best_idx = i;
        }
        else if (d < second) second = d;
    }
    if (HAMMING_DIST_THR_LOW < best || best_idx < 0) return -1;
    if (P.lowe_ratio * (float)second < (float)best) return -1;
    return best_idx;
}

// One workgroup per pair.  owner[n1] / match[n2] live in LDS when they fit, else in global scratch.
__global__ __launch_bounds__(256) void k_bf_replay(BfProblem P, int* __restrict__ g_owner, int* __restrict__ g_match, int use_lds) {
    extern __shared__ int s_mem[];
    __shared__ int s_changed;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n1 = P.n1_dev ? P.n1_dev[pair * P.n_stride] : P.n1;
    const int n2 = P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2;
    const int n1c = min(n1, P.cap1), n2c = min(n2, P.cap2);
    int* owner = use_lds ? s_mem : g_owner + (size_t)pair * P.cap1;
    int* match = use_lds ? s_mem + P.cap1 : g_match + (size_t)pair * P.cap2;
    const uint32_t* T = P.topk + (size_t)pair * P.cap2 * BF_K;
    const int* C = P.cnt + (size_t)pair * P.cap2;
    for (int i = tid; i < n1c; i += 256) owner[i] = 0x7FFFFFFF;
    for (int j = tid; j < n2c; j += 256) match[j] = -2;  // "unknown": forces at least one full sweep
    __syncthreads();
    for (int sweep = 0; sweep <= n2c; ++sweep) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        int local_changed = 0;
        // decisions against the owner table of the previous sweep; results parked in registers via match2 pass
        for (int j = tid; j < n2c; j += 256) {
            const int d = bf_decide(P, pair, j, n1c, T + (size_t)j * BF_K, C[j], owner);
            if (d != match[j]) local_changed = 1;
            // stash the new decision in the sign-safe upper half: decisions only read `owner`, not `match`
            match[j] = d;
        }
        if (local_changed) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;
        for (int i = tid; i < n1c; i += 256) owner[i] = 0x7FFFFFFF;
        __syncthreads();
        for (int j = tid; j < n2c; j += 256)
            if (match[j] >= 0) atomicMin(&owner[match[j]], j);
        __syncthreads();
    }
    // write-out: matched_2_in_1[idx_1] = idx_2 (robust.cc:317-325), unique by construction
    int32_t* out = P.matched + (size_t)pair * P.cap1;
    for (int i = tid; i < P.cap1; i += 256) out[i] = -1;
    __syncthreads();
    int local = 0;
    for (int j = tid; j < n2c; j += 256)
        if (match[j] >= 0) {
            out[match[j]] = j;
            ++local;
        }
    if (tid == 0) s_changed = 0;
    __syncthreads();
    if (local) atomicAdd(&s_changed, local);
    __syncthreads();
    if (tid == 0) P.num[pair] = s_changed;
}

// ------------------------------------------------------------------------------------------------ candidate lists
// dist[c] for every CSR entry; 0xFFFF = gated out (stereo / orientation gates of projection.cc:57-62,179-181)
__global__ void k_cand_dist(CandProblem P) {
    const int q = blockIdx.x;
    if (P.q_valid && !P.q_valid[q]) return;
    const int lo = P.cand_off[q], hi = P.cand_off[q + 1];
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = P.qdesc[(size_t)q * 8 + k];
    for (int c = lo + threadIdx.x; c < hi; c += blockDim.x) {
        const int t = P.cand_idx[c];
        bool gated = false;
        if (P.t_xright && 0.f < P.t_xright[t]) {
            const float err = fabsf(P.q_xright[q] - P.t_xright[t]);
            if (P.q_xr_tol[q] < err) gated = true;
        }
        if (!gated && P.check_orientation && fabsf(angle_diff(P.q_angle[q], P.t_angle[t])) > 30.0f) gated = true;
        P.dist[c] = gated ? (uint16_t)0xFFFF : (uint16_t)hamming256(qd, P.tdesc + (size_t)t * 8);
    }
}

__device__ int cand_decide(const CandProblem& P, int q, const int* owner) {
    if (P.q_valid && !P.q_valid[q]) return -1;
    const int lo = P.cand_off[q], hi = P.cand_off[q + 1];
    if (lo == hi) return -1;
    unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
    int best_lvl = -1, second_lvl = -1, best_idx = -1;
    for (int c = lo; c < hi; ++c) {
        const unsigned d = P.dist[c];
        if (d == 0xFFFFu) continue;
        const int t = P.cand_idx[c];
        if (owner[t] < q) continue;  // occupied before this query (initially, or by an earlier query)
        if (d < best) {
            second = best;
            best = d;
            second_lvl = best_lvl;
            best_lvl = P.t_octave ? P.t_octave[t] : 0;
            best_idx = t;
        }
        else if (d < second) {
            second_lvl = P.t_octave ? P.t_octave[t] : 0;
            second = d;
        }
    }
    if (P.mode == SVGPU_MATCH_RATIO_SAME_OCTAVE) {
        if (best <= P.thr) {
            if (best_lvl == second_lvl && (float)best > P.lowe_ratio * (float)second) return -1;
            return best_idx;
        }
        return -1;
    }
    if (P.thr < best) return -1;
    return best_idx;
}

__global__ __launch_bounds__(256) void k_cand_replay(CandProblem P, int* __restrict__ owner, int* __restrict__ match) {
    __shared__ int s_changed;
    const int tid = threadIdx.x;
    auto reset_owner = [&]() {
        for (int t = tid; t < P.nt; t += 256) owner[t] = (P.occupied && P.occupied[t]) ? -1 : 0x7FFFFFFF;
    };
    reset_owner();
    for (int q = tid; q < P.nq; q += 256) match[q] = -2;
    __syncthreads();
    for (int sweep = 0; sweep <= P.nq; ++sweep) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        int local_changed = 0;
        for (int q = tid; q < P.nq; q += 256) {
            const int d = cand_decide(P, q, owner);
            if (d != match[q]) local_changed = 1;
            match[q] = d;
        }
        if (local_changed) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;
        reset_owner();
        __syncthreads();
        for (int q = tid; q < P.nq; q += 256)
            if (match[q] >= 0) atomicMin(&owner[match[q]], q);
        __syncthreads();
    }
    int local = 0;
    for (int q = tid; q < P.nq; q += 256) {
        P.match_q[q] = match[q];
        local += match[q] >= 0;
    }
    if (tid == 0) s_changed = 0;
    __syncthreads();
    if (local) atomicAdd(&s_changed, local);
    __syncthreads();
    if (tid == 0) *P.num = s_changed;
}

}  // namespace

void sv_launch_hamming_pairs(hipStream_t s, const uint32_t* a, const uint32_t* b, int n, uint32_t* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, out);
}
void sv_launch_hamming_matrix(hipStream_t s, const uint32_t* d1, int n1, const uint32_t* d2, int n2, uint16_t* out) {
    if (n1 <= 0 || n2 <= 0) return;
    hipLaunchKernelGGL(k_hamming_matrix, dim3((n2 + 255) / 256), dim3(256), 0, s, d1, n1, d2, n2, out);
}
void sv_launch_bf(svgpu_ctx* ctx, hipStream_t s, const BfProblem& P0, int pairs, int* g_owner, int* g_match) {
    if (pairs <= 0) return;
    BfProblem P = P0;
    // dmax: smallest cutoff such that lowe_ratio * (dmax + 1) >= 50 certainly holds (two units of slack for fp32 rounding)
    if (P.lowe_ratio > 0.f && 50.0f / P.lowe_ratio + 2.0f < 256.0f) P.dmax = (unsigned)(50.0f / P.lowe_ratio) + 2u;
    else P.dmax = 256u;
    if (P.dmax < 50u) P.dmax = 50u;
    P.exhaustive = P.dmax >= 256u;
    {
        SvProfScope ps(ctx, s, "k_bf_topk");
        hipLaunchKernelGGL(k_bf_topk, dim3((P.cap2 + 63) / 64, pairs), dim3(256), 0, s, P);
    }
    SvProfScope ps(ctx, s, "k_bf_replay");
    const size_t lds = (size_t)(P.cap1 + P.cap2) * sizeof(int);
    const int use_lds = lds <= 96 * 1024;
    hipLaunchKernelGGL(k_bf_replay, dim3(pairs), dim3(256), use_lds ? lds : 0, s, P, g_owner, g_match, use_lds);
}
void sv_launch_cand(svgpu_ctx* ctx, hipStream_t s, const CandProblem& P, int* owner, int* match) {
    SvProfScope ps(ctx, s, "k_cand");
    if (P.nq > 0) hipLaunchKernelGGL(k_cand_dist, dim3(P.nq), dim3(64), 0, s, P);
    hipLaunchKernelGGL(k_cand_replay, dim3(1), dim3(256), 0, s, P, owner, match);
}
