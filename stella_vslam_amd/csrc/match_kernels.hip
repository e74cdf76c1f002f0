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
#include <algorithm>
#include <cstdlib>
#include "match_kernels.h"
#include "sv_sort.h"
#include "match_device.h"

namespace {
using namespace svmd;

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
// ---- orientation pruning.  v_xor + v_bcnt_u32_b32 cost ~9 cycles per wave and 32-bit word on gfx950, so a full
// 2.4k x 2.4k x 256-bit brute force is VALU-bound near 180 us per 64 pairs; the only real lever is to do fewer
// distance evaluations.  With check_orientation the reference rejects every pair whose angles differ by more
// than 30 degrees BEFORE looking at the descriptors (robust.cc:279), so both sides are first bucketed into 360
// one-degree angle bins (k_bf_binsort: LDS histogram + scan + scatter into angle-sorted copies); a block of
// queries then scans only the candidate bins within 31 bins of its own bin range (a superset of the pairs that pass
// the exact gate, which is still evaluated per pair).  Results are keyed by ORIGINAL indices and the per-query
// lists are sets, so the outcome is identical to the unpruned scan.
#define BF_BINS 360
#define BF_ANG_CACHE 4096
__global__ __launch_bounds__(256) void k_bf_binsort(BfProblem P) {
    __shared__ int s_hist[BF_BINS + 2];
    __shared__ int s_start[BF_BINS + 2];
    __shared__ int s_bad;
    __shared__ float s_ang[BF_ANG_CACHE];  // the angles of the first pass, so that the keypoint records (28-byte stride) are read once
    const int pair = blockIdx.x, side = P.shared_sort ? 1 : blockIdx.y, tid = threadIdx.x;
    const int cap = side == 0 ? P.cap1 : P.cap2;
    const int row = side == 0 ? bf_row1(P, pair) : pair;  // where this side of the pair lives in the caller's arrays
    const int nraw = side == 0 ? (P.n1_dev ? P.n1_dev[row * P.n_stride] : P.n1) : (P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2);
    const int n = min(nraw, cap);
    const uint32_t* D = (side == 0 ? P.desc1 : P.desc2) + (size_t)row * cap * 8;
    const float* A = (side == 0 ? P.angle1 : P.angle2) + (size_t)row * cap * P.angle_stride;
    uint32_t* SD = (side == 0 ? P.sd1 : P.sd2) + (size_t)pair * cap * 8;
    float* SA = (side == 0 ? P.sa1 : P.sa2) + (size_t)pair * cap;
    int* SI = (side == 0 ? P.si1 : P.si2) + (size_t)pair * cap;
    int* BS = (side == 0 ? P.bs1 : P.bs2) + (size_t)pair * (BF_BINS + 2);
    for (int b = tid; b < BF_BINS + 2; b += 256) s_hist[b] = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    auto bin_of = [&](float a, bool& ok) {
        ok = a >= 0.f && a <= 360.f;
        return ok ? min((int)a, BF_BINS - 1) : 0;
    };
    for (int i = tid; i < n; i += 256) {
        bool ok;
        const float a = A[(size_t)i * P.angle_stride];
        if (i < BF_ANG_CACHE) s_ang[i] = a;
        const int b = bin_of(a, ok);
        if (!ok) s_bad = 1;
        atomicAdd(&s_hist[b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b < BF_BINS; ++b) {
            s_start[b] = run;
            run += s_hist[b];
        }
        s_start[BF_BINS] = s_start[BF_BINS + 1] = run;
    }
    __syncthreads();
    for (int b = tid; b < BF_BINS + 2; b += 256) {
        BS[b] = s_start[b];
        s_hist[b] = 0;  // reused as the per-bin cursor
    }
    if (tid == 0) P.prune_ok[pair * 2 + side] = s_bad ? 0 : 1;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        bool ok;
        const float a = i < BF_ANG_CACHE ? s_ang[i] : A[(size_t)i * P.angle_stride];
        const int b = bin_of(a, ok);
        const int dst = s_start[b] + atomicAdd(&s_hist[b], 1);
        const uint4 lo = *reinterpret_cast<const uint4*>(D + (size_t)i * 8), hi = *reinterpret_cast<const uint4*>(D + (size_t)i * 8 + 4);
        *reinterpret_cast<uint4*>(SD + (size_t)dst * 8) = lo;
        *reinterpret_cast<uint4*>(SD + (size_t)dst * 8 + 4) = hi;
        SA[dst] = a;
        SI[dst] = i;
    }
}

// Block = 256 queries (keyframe keypoints idx_2, in angle-bin order), one per lane.  The candidate range (frame
// keypoints idx_1 whose bin is within 31 of the block's bins; everything when pruning is off) streams through
// double-buffered LDS tiles shared by the four waves; a candidate descriptor is two broadcast ds_read_b128
// (conflict-free: every lane reads the same address), compared with v_xor / v_bcnt_u32_b32.  Each query keeps its
// BF_K smallest (dist << 16 | original idx_1) keys among the candidates within dmax.
// Ascending key order = the reference's scan preference (strict '<' => lowest index wins ties).
#define BF_QB 256
#define BF_TILE 128
__global__ __launch_bounds__(256) void k_bf_topk(BfProblem P) {
    __shared__ __attribute__((aligned(16))) uint32_t s_d[2][BF_TILE * 8];
    __shared__ float s_a[2][BF_TILE];
    __shared__ int s_i[2][BF_TILE];
    const int pair = blockIdx.y;
    const int n1 = P.n1_dev ? P.n1_dev[bf_row1(P, pair) * P.n_stride] : P.n1;
    const int n2 = P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2;
    const int n1c = min(n1, P.cap1), n2c = min(n2, P.cap2);
    if (blockIdx.x * BF_QB >= n2c) return;
    const int tid = threadIdx.x;
    const uint32_t* __restrict__ D1 = P.sd1 + (size_t)bf_srow1(P, pair) * P.cap1 * 8;
    const float* __restrict__ A1 = P.sa1 + (size_t)bf_srow1(P, pair) * P.cap1;
    const int* __restrict__ I1 = P.si1 + (size_t)bf_srow1(P, pair) * P.cap1;
    const uint32_t* __restrict__ D2 = P.sd2 + (size_t)pair * P.cap2 * 8;
    const float* __restrict__ A2 = P.sa2 + (size_t)pair * P.cap2;
    const int* __restrict__ I2 = P.si2 + (size_t)pair * P.cap2;
    const int* __restrict__ BS1 = P.bs1 + (size_t)bf_srow1(P, pair) * (BF_BINS + 2);
    const bool ori = P.check_orientation != 0;
    // ---- this lane's query
    const int r = blockIdx.x * BF_QB + tid, rr = min(r, n2c - 1);
    const int j = I2[rr];
    const bool active = r < n2c && (!P.valid2 || P.valid2[(size_t)pair * P.cap2 + j]);
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = D2[(size_t)rr * 8 + k];
    const float qa = A2[rr];
    uint32_t L[BF_K];
#pragma unroll
    for (int k = 0; k < BF_K; ++k) L[k] = 0xFFFFFFFFu;
    int cnt = 0;
    // ---- candidate ranges of the block (block-uniform): up to two runs of the angle-sorted side 1
    int seg_lo[2] = {0, 0}, seg_hi[2] = {n1c, 0};
    if (ori && bf_prune1(P, pair) && P.prune_ok[pair * 2 + 1]) {
        const int r_lo = blockIdx.x * BF_QB, r_hi = min(r_lo + BF_QB - 1, n2c - 1);
        const int b_lo = min((int)A2[r_lo], BF_BINS - 1) - 31, b_hi = min((int)A2[r_hi], BF_BINS - 1) + 31;
        if (b_hi - b_lo + 1 < BF_BINS) {
            if (b_lo < 0) {
                seg_lo[0] = BS1[b_lo + BF_BINS];
                seg_hi[0] = n1c;
                seg_lo[1] = 0;
                seg_hi[1] = BS1[b_hi + 1];
            }
            else if (b_hi >= BF_BINS) {
                seg_lo[0] = BS1[b_lo];
                seg_hi[0] = n1c;
                seg_lo[1] = 0;
                seg_hi[1] = BS1[b_hi - BF_BINS + 1];
            }
            else {
                seg_lo[0] = BS1[b_lo];
                seg_hi[0] = BS1[b_hi + 1];
            }
        }
    }
    // Only candidates with dist <= dmax can influence a decision (best must be <= 50, and the ratio test
    // lowe_ratio * second < best can only reject when second < 50 / lowe_ratio): everything farther is never listed.
    const unsigned dmax = P.dmax;
    auto consider = [&](int idx, unsigned d, float ca) {
        if (d <= dmax) {
            bool pass = active;
            if (ori) pass = pass && !(fabsf(angle_diff(ca, qa)) > 30.0f);
            if (pass) {
                ++cnt;
                const uint32_t key = (d << 16) | (uint32_t)idx;
                if (key < L[BF_K - 1]) {
#pragma unroll
                    for (int k = BF_K - 1; k > 0; --k) L[k] = max(L[k - 1], min(L[k], key));
                    L[0] = min(L[0], key);
                }
            }
        }
    };
    // tile loader: 128 descriptors = 1024 dwords = one uint4 per thread (tail entries all-ones: far from everything)
    auto load_tile = [&](int buf, int base, int end) {
        const int m = min(BF_TILE, end - base);
        uint4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (tid * 4 < m * 8) v = *reinterpret_cast<const uint4*>(D1 + (size_t)base * 8 + tid * 4);
        reinterpret_cast<uint4*>(s_d[buf])[tid] = v;
        if (tid < BF_TILE) {
            s_a[buf][tid] = tid < m ? A1[base + tid] : 0.f;
            s_i[buf][tid] = tid < m ? I1[base + tid] : 0;
        }
    };
    int buf = 0;
    for (int sg = 0; sg < 2; ++sg) {
        const int lo = seg_lo[sg], hi = seg_hi[sg];
        if (lo >= hi) continue;
        __syncthreads();
        load_tile(buf, lo, hi);
        __syncthreads();
        for (int base = lo; base < hi; base += BF_TILE, buf ^= 1) {
            if (base + BF_TILE < hi) load_tile(buf ^ 1, base + BF_TILE, hi);  // next tile lands while this one is consumed
            const int m = min(BF_TILE, hi - base);
            const uint32_t* T = s_d[buf];
            for (int i = 0; i < m; i += 4) {  // 4 candidates per trip
                uint4 c[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u][0] = *reinterpret_cast<const uint4*>(T + (i + u) * 8);
                    c[u][1] = *reinterpret_cast<const uint4*>(T + (i + u) * 8 + 4);
                }
                unsigned d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    d[u] = __popc(q[0] ^ c[u][0].x) + __popc(q[1] ^ c[u][0].y) + __popc(q[2] ^ c[u][0].z) + __popc(q[3] ^ c[u][0].w)
                           + __popc(q[4] ^ c[u][1].x) + __popc(q[5] ^ c[u][1].y) + __popc(q[6] ^ c[u][1].z) + __popc(q[7] ^ c[u][1].w);
                if (min(min(d[0], d[1]), min(d[2], d[3])) <= dmax) {  // rarely-taken wave-level branch
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i + u < m) consider(s_i[buf][i + u], d[u], s_a[buf][i + u]);
                }
            }
            __syncthreads();
        }
    }
    if (r < n2c) {
        uint32_t* T = P.topk + ((size_t)pair * P.cap2 + j) * BF_LIST;
#pragma unroll
        for (int k = 0; k < BF_K; k += 4) *reinterpret_cast<uint4*>(T + k) = make_uint4(L[k], L[k + 1], L[k + 2], L[k + 3]);
        P.cnt[(size_t)pair * P.cap2 + j] = active ? cnt : 0;
    }
}

// ------------------------------------------------------------------------------------------------ brute force on the matrix cores
// All-pairs Hamming distance IS a matrix product: with both descriptors mapped bit -> {+1, -1} (int8),
//     sum_k q'_k t'_k = 256 - 2 hamming(q, t),
// so v_mfma_i32_32x32x32_i8 produces 1024 exact distances per 8 instructions (~0.3 cycles per distance and SIMD,
// against ~1.2 for v_xor / v_bcnt_u32_b32) and "distance <= dmax" is one uniform threshold on the accumulator.
// Both sides arrive angle-bin sorted (k_bf_binsort), so a block of 256 consecutive queries only visits the targets
// within 31 bins of its own bins, and each of its four waves (64 queries: two 32-row A tiles expanded to int8 once,
// 64 registers) skips the tiles outside its own, narrower window.  Targets are fetched 256 at a time into LDS (one
// global round trip per chunk) and expanded 32 at a time: 256 threads turn one dword each (8 nibbles -> 8 dwords of
// +-1 bytes) into the layout [k-step][lane] x 16 B, so that a B fragment is one conflict-free ds_read_b128 and serves
// both A tiles; sharing the expanded tile between the four waves keeps the operand traffic in LDS (a wave streaming
// its own 8 KB per patch from L2 is bandwidth-bound).  Both operands use the same (k-step, lane half, byte) -> bit
// map, which is all the MFMA needs (the dot product does not care how the hardware orders k inside the instruction).
// Only distances <= dmax can influence a decision (see k_bf_topk); well under 1 % of the pairs qualify, about a dozen
// per 64 x 32 patch.  Every accumulator register is tested against the threshold; the hit lanes append
// (row, dist, original idx_1, target angle) to a per-wave queue with ballot ranks (no atomics, no LDS reads on the
// way).  When the queue fills, the wave drains it with all lanes busy: exact orientation gate, then a slot in the
// query's row (rows are private to the wave); a full row keeps its MF_SLOTS smallest keys (wave-cooperative
// replace-the-maximum, rare).  Rows are sorted and written out once per block, so k_bf_replay gets exactly what
// k_bf_topk would give it: the nearest candidates in scan-preference order plus the count of all candidates.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define MF_QW 64       // queries per wave
#define MF_QB 256      // queries per block (4 waves)
#define MF_TT 32       // targets per MFMA tile
#define MF_CH 256      // targets per fetched chunk
#define MF_SLOTS 16    // list slots per query (= P.list_k of this path)
#define MF_WQ 256      // per-wave hit queue entries (row << 16 | column of the chunk); drained when fewer than 64 are free and at the end of every chunk
// Operand form (round 4).  Bits are expanded to 0 / 2 bytes on the query side and 0 / 1 bytes on the target side, and a NINTH k-step
// carries the two popcounts, so that the accumulator is the negated distance itself:
//     sum_k (2 q_k) t_k  -  pop(q)  -  pop(t)  =  -hamming(q, t),
// the popcounts riding as three int8 pieces each (<= 127 + 127 + 2) against ones on the other side.  "distance <= dmax" stays ONE uniform
// threshold on the accumulator, and the expansion of a nibble is two shift-ors and a mask (4 instructions with the field extract) where
// the +-1 form smeared every set bit over its byte (9): 75 -> 34 VALU instructions per thread and tile, for one more MFMA in nine.
__device__ __forceinline__ uint32_t mf_spread(uint32_t nibble) {  // bit b of the nibble -> bit 0 of byte b (other bits: copies of the nibble, masked by the caller)
    uint32_t m = nibble | (nibble << 7);
    return m | (m << 14);
}
__device__ __forceinline__ uint32_t mf_expand01(uint32_t nibble) { return mf_spread(nibble) & 0x01010101u; }        // targets: 0 / 1 (expanded per tile: the cheaper form)
__device__ __forceinline__ uint32_t mf_expand02(uint32_t nibble) { return (mf_spread(nibble) << 1) & 0x02020202u; }  // queries: 0 / 2 (expanded once per kernel)
// the ninth k-step: -(pop) as three int8 pieces in bytes 0..2 (own side) or 3..5 (other side's ones sit in the complementary bytes); bytes 6, 7
// carry the threshold: the query side holds dmax in two pieces (<= 127 each) against ones on the target side, so that the accumulator
// is dmax - distance and "hit" is its SIGN BIT (round 6: the hit mask of a lane is then one v_alignbit_b32 per accumulator register)
__device__ __forceinline__ uint2 mf_pop_pieces(int pop, bool own_first, int dmax) {
    const int a = min(pop, 127), b = min(pop - a, 127), c = pop - a - b;
    const uint32_t na = (uint32_t)(-a) & 0xFFu, nb = (uint32_t)(-b) & 0xFFu, nc = (uint32_t)(-c) & 0xFFu;
    const uint32_t d0 = (uint32_t)min(dmax, 127), d1 = (uint32_t)(dmax - (int)d0);
    // query side: bytes [-a -b -c  1 | 1 1 d0 d1]; target side: bytes [1 1 1 -a | -b -c 1 1]
    return own_first ? make_uint2(na | (nb << 8) | (nc << 16) | (1u << 24), 0x00000101u | (d0 << 16) | (d1 << 24))
                     : make_uint2(0x00010101u | (na << 24), nb | (nc << 8) | 0x01010000u);
}
// candidate window (in angle-sorted target positions) of the sorted queries [r_lo, r_hi]: up to two runs
__device__ __forceinline__ void mf_window(const float* __restrict__ A2s, const int* __restrict__ BS1, int r_lo, int r_hi, int n1c,
                                          bool prune, int (&lo)[2], int (&hi)[2]) {
    lo[0] = 0;
    hi[0] = n1c;
    lo[1] = hi[1] = 0;
    if (!prune) return;
    const int b_lo = min((int)A2s[r_lo], BF_BINS - 1) - 31, b_hi = min((int)A2s[r_hi], BF_BINS - 1) + 31;
    if (b_hi - b_lo + 1 >= BF_BINS) return;
    if (b_lo < 0) {
        lo[0] = BS1[b_lo + BF_BINS];
        hi[0] = n1c;
        lo[1] = 0;
        hi[1] = BS1[b_hi + 1];
    }
    else if (b_hi >= BF_BINS) {
        lo[0] = BS1[b_lo];
        hi[0] = n1c;
        lo[1] = 0;
        hi[1] = BS1[b_hi - BF_BINS + 1];
    }
    else {
        lo[0] = BS1[b_lo];
        hi[0] = BS1[b_hi + 1];
    }
}
struct MfShared {
    uint32_t raw[8][MF_CH];          // fetched chunk, dword-transposed: raw[k-step][target]
    float rang[MF_CH];               // target angles / original indices of the chunk
    int ridx[MF_CH];
    int rpop[MF_CH];                 // ... and their popcounts
    uint32_t b[2][9 * 64 * 4];       // expanded tile: [buffer][k-step 8 + the popcount step][lane 64][4 dwords]
    uint32_t list[MF_QB][MF_SLOTS];
    int cnt[MF_QB];
    uint32_t rowmax[MF_QB];          // largest key of a FULL row once it has been scanned (else all-ones): cheap reject of far candidates
    float qa[MF_QB];                 // query angle, or -1000 for rows that must never match (past n2, !valid2)
    uint32_t wq[4][MF_WQ];           // (query row of the wave << 16 | target column of the chunk): the drain recomputes the distance
#ifdef SV_MF_PAD
    uint32_t pad[SV_MF_PAD];
#endif
};
// max over every DPP row of 16 lanes, left in all 16 (rotations by 8, 4, 2, 1)
__device__ __forceinline__ uint32_t mf_row16_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));  // row_ror:8
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));  // row_ror:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false));  // row_ror:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false));  // row_ror:1
    return v;
}
// Drain the wave's hit queue into its rows.  Called by all 64 lanes of the wave, while the chunk the entries point into is still in LDS.
// An entry names (query row, target column); the distance is recomputed here (xor + popcount of the two descriptors: the same number the
// accumulator held), so the hit path of the multiply loop carries no payload and no accumulator has to be indexed by a run-time register.
__device__ __forceinline__ void mf_drain(MfShared& S, const uint32_t* __restrict__ D2, int q_first, int n2c, int wave, int lane, int n, bool ori) {
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        bool live = e < n;
        uint32_t key = 0;
        int ql = 0, slot = 0;
        if (live) {
            const uint32_t ent = S.wq[wave][e];
            const int row = (int)(ent >> 16), col = (int)(ent & 0xFFFFu);
            ql = row + wave * MF_QW;
            const uint32_t* q = D2 + (size_t)min(q_first + row, n2c - 1) * 8;
            const uint4 q0 = *reinterpret_cast<const uint4*>(q), q1 = *reinterpret_cast<const uint4*>(q + 4);
            const uint32_t dist = __popc(q0.x ^ S.raw[0][col]) + __popc(q0.y ^ S.raw[1][col]) + __popc(q0.z ^ S.raw[2][col]) + __popc(q0.w ^ S.raw[3][col])
                                  + __popc(q1.x ^ S.raw[4][col]) + __popc(q1.y ^ S.raw[5][col]) + __popc(q1.z ^ S.raw[6][col]) + __popc(q1.w ^ S.raw[7][col]);
            key = (dist << 16) | (uint32_t)S.ridx[col];
            const float qa = S.qa[ql];
            live = qa > -500.f;
            if (ori) live = live && !(fabsf(angle_diff(S.rang[col], qa)) > 30.0f);
            if (live) slot = atomicAdd(&S.cnt[ql], 1);  // several lanes may hold hits of the same row
            if (live && slot < MF_SLOTS) S.list[ql][slot] = key;
        }
        // rare in general, common for queries with dozens of near-identical candidates: the row is full -> keep its MF_SLOTS
        // smallest keys.  Keys not below the row's current maximum are dropped right away (counted, not listed).
        unsigned long long ov = __ballot(live && slot >= MF_SLOTS && key < S.rowmax[ql]);
        while (ov) {
            const int src = __ffsll((long long)ov) - 1;
            ov &= ov - 1;
            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, src);  // (src is wave-uniform: a scalar read, not an LDS shuffle)
            const int row = __builtin_amdgcn_readlane(ql, src);
            uint32_t* R = S.list[row];
            if (k >= S.rowmax[row]) continue;  // an earlier entry of this batch lowered the maximum
            // the row's maximum and where it sits, on DPP row rotations (the 16 slots are the 16 lanes of DPP row 0; keys of a row are
            // distinct -- they carry distinct target indices -- so the maximum has one owner): four VALU moves instead of the ten
            // ds_bpermute round trips of a shuffle tree
            const uint32_t v = lane < MF_SLOTS ? R[lane] : 0u;
            const uint32_t vmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)mf_row16_max(v));
            const int arg = __ffsll((long long)__ballot(lane < MF_SLOTS && v == vmax)) - 1;
            // replace the maximum, then the new maximum of the row = max(k, largest of the other slots)
            uint32_t w = lane == arg ? min(k, vmax) : v;
            if (k < vmax && lane == 0) R[arg] = k;
            w = mf_row16_max(w);
            if (lane == 0) S.rowmax[row] = w;
        }
    }
}
__global__ __launch_bounds__(256, 3) void k_bf_mfma(BfProblem P) {
    __shared__ __attribute__((aligned(16))) MfShared S;
    const int pair = blockIdx.y;
    const int n1 = P.n1_dev ? P.n1_dev[bf_row1(P, pair) * P.n_stride] : P.n1;
    const int n2 = P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2;
    const int n1c = min(n1, P.cap1), n2c = min(n2, P.cap2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = blockIdx.x * MF_QB, q0 = qb + wave * MF_QW;
    if (qb >= n2c) return;
    const uint32_t* __restrict__ D1 = P.sd1 + (size_t)bf_srow1(P, pair) * P.cap1 * 8;
    const float* __restrict__ A1 = P.sa1 + (size_t)bf_srow1(P, pair) * P.cap1;
    const int* __restrict__ I1 = P.si1 + (size_t)bf_srow1(P, pair) * P.cap1;
    const uint32_t* __restrict__ D2 = P.sd2 + (size_t)pair * P.cap2 * 8;
    const float* __restrict__ A2 = P.sa2 + (size_t)pair * P.cap2;
    const int* __restrict__ I2 = P.si2 + (size_t)pair * P.cap2;
    const int* __restrict__ BS1 = P.bs1 + (size_t)bf_srow1(P, pair) * (BF_BINS + 2);
    const bool ori = P.check_orientation != 0;
    const bool prune = ori && bf_prune1(P, pair) && P.prune_ok[pair * 2 + 1];
    S.cnt[tid] = 0;
    S.rowmax[tid] = 0xFFFFFFFFu;
    int my_j = 0;  // original idx_2 of this thread's query row (flush)
    {
        const int q = qb + tid;
        bool live = q < n2c;
        if (live) {
            my_j = I2[q];
            live = !P.valid2 || P.valid2[(size_t)pair * P.cap2 + my_j];
        }
        S.qa[tid] = live ? A2[q] : -1000.f;
    }
    // ---- A fragments: rows = sorted queries q0 + 32a + (lane & 31), k-step s, lane half h -> bits [32s + 16h, +16); step 8 = the popcount step
    const int h = lane >> 5;
    v4i A[2][9];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int q = min(q0 + 32 * a + (lane & 31), n2c - 1);
        int pq = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint32_t w = D2[(size_t)q * 8 + s];
            pq += __popc(w);
            const uint32_t half = (w >> (16 * h)) & 0xFFFFu;
            A[a][s] = v4i{(int)mf_expand02(half & 15u), (int)mf_expand02((half >> 4) & 15u), (int)mf_expand02((half >> 8) & 15u), (int)mf_expand02(half >> 12)};
        }
        const uint2 pp = mf_pop_pieces(pq, true, min((int)P.dmax, 254));
        A[a][8] = h == 0 ? v4i{(int)pp.x, (int)pp.y, 0, 0} : v4i{0, 0, 0, 0};
    }
    // ---- candidate windows: the block's (targets to stage) and this wave's (tiles to multiply)
    int blo[2], bhi[2], wlo[2], whi[2];
    mf_window(A2, BS1, qb, min(qb + MF_QB, n2c) - 1, n1c, prune, blo, bhi);
    const bool wave_live = q0 < n2c;
    mf_window(A2, BS1, min(q0, n2c - 1), min(q0 + MF_QW, n2c) - 1, n1c, prune, wlo, whi);
    const int lt = tid & 31, lw = tid >> 5;  // expansion role: target lt of the tile, descriptor dword lw (= k-step)
    auto expand = [&](int buf, int tile, int valid) {  // tile of the chunk in S.raw -> S.b[buf]; columns >= valid can never hit (popcount 381)
        uint4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (lt < valid) {
            const uint32_t w = S.raw[lw][tile * MF_TT + lt];
            lo.x = mf_expand01(w & 15u);
            lo.y = mf_expand01((w >> 4) & 15u);
            lo.z = mf_expand01((w >> 8) & 15u);
            lo.w = mf_expand01((w >> 12) & 15u);
            hi.x = mf_expand01((w >> 16) & 15u);
            hi.y = mf_expand01((w >> 20) & 15u);
            hi.z = mf_expand01((w >> 24) & 15u);
            hi.w = mf_expand01(w >> 28);
        }
        uint4* B = reinterpret_cast<uint4*>(S.b[buf]) + lw * 64;
        B[lt] = lo;        // lane half 0: bits [32 lw, +16)
        B[32 + lt] = hi;   // lane half 1: bits [32 lw + 16, +16)
        if (lw == 0) {     // the popcount step of the tile (one wave-half's worth of threads)
            const uint2 pp = lt < valid ? mf_pop_pieces(S.rpop[tile * MF_TT + lt], false, 0) : make_uint2(0x81010101u, 0x01018181u);
            uint4* B8 = reinterpret_cast<uint4*>(S.b[buf]) + 8 * 64;
            B8[lt] = make_uint4(pp.x, pp.y, 0, 0);
            B8[32 + lt] = make_uint4(0, 0, 0, 0);
        }
    };
    int wq_n = 0;                           // entries in this wave's hit queue (wave-uniform)
    unsigned n_mine = 0;  // patches this wave multiplied (profiling counter)
    // A wave's 64 x 32 patch of a tile: every B fragment is read from LDS ONCE and feeds both 32-query halves (eighteen MFMAs back to back).
    // The 32 accumulator registers of a lane (rows 32 a + (r & 3) + 8 (r >> 2) + 4 h of the C/D map, one target column per lane) then
    // leave a lane-LOCAL hit mask -- one v_alignbit_b32 per register (the ninth k-step carries + dmax, so a hit is a clear sign bit), no
    // compare, no ballot, no branch -- and the patch costs one wave-uniform
    // test; the few lanes that hold hits append (row, column) words to the wave's queue, one per lane and round.
    // (Round 5 tested every register with a ballot and a branch and appended row, distance, index and angle from inside the multiply loop:
    //  130 M scalar instructions per 1 024 pairs, the matrix pipe 17 % busy.)
    for (int sg = 0; sg < 2; ++sg) {
        const int lo = blo[sg], hi = bhi[sg];
        for (int cbase = lo; cbase < hi; cbase += MF_CH) {
            const int cn = min(MF_CH, hi - cbase), ntiles = (cn + MF_TT - 1) / MF_TT;
            __syncthreads();  // every wave is done with the previous chunk (its queue entries are drained)
            if (tid < cn) {   // one descriptor per thread: two 16-byte loads, scattered into the transposed layout
                const uint4 d0 = *reinterpret_cast<const uint4*>(D1 + (size_t)(cbase + tid) * 8);
                const uint4 d1 = *reinterpret_cast<const uint4*>(D1 + (size_t)(cbase + tid) * 8 + 4);
                S.rang[tid] = A1[cbase + tid];
                S.ridx[tid] = I1[cbase + tid];
                S.rpop[tid] = __popc(d0.x) + __popc(d0.y) + __popc(d0.z) + __popc(d0.w) + __popc(d1.x) + __popc(d1.y) + __popc(d1.z) + __popc(d1.w);
                S.raw[0][tid] = d0.x;
                S.raw[1][tid] = d0.y;
                S.raw[2][tid] = d0.z;
                S.raw[3][tid] = d0.w;
                S.raw[4][tid] = d1.x;
                S.raw[5][tid] = d1.y;
                S.raw[6][tid] = d1.z;
                S.raw[7][tid] = d1.w;
            }
            __syncthreads();
            expand(0, 0, min(MF_TT, cn));
            __syncthreads();
            for (int t = 0, buf = 0; t < ntiles; ++t, buf ^= 1) {
                const int base = cbase + t * MF_TT, tend = min(base + MF_TT, hi);
                const bool mine = wave_live && ((base < whi[0] && tend > wlo[0]) || (base < whi[1] && tend > wlo[1]));
                if (mine) {
                    const v4i* Bf = reinterpret_cast<const v4i*>(S.b[buf]) + lane;
                    v16i acc0 = {}, acc1 = {};
                    v4i b = Bf[0];
#pragma unroll
                    for (int s = 0; s < 9; ++s) {
                        const v4i bn = Bf[(s < 8 ? s + 1 : 0) * 64];  // next fragment in flight
                        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][s], b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][s], b, acc1, 0, 0, 0);
                        b = bn;
                    }
                    ++n_mine;
                    // the next tile's expansion is pure VALU / LDS work: it runs in the shadow of the MFMAs
                    if (t + 1 < ntiles) expand(buf ^ 1, t + 1, min(MF_TT, cn - (t + 1) * MF_TT));
                    uint32_t m = 0;  // the accumulator is dmax - distance: shift the sign bits in; then bit r: acc0[r] is a hit, bit 16 + r: acc1[r]
#pragma unroll
                    for (int r = 15; r >= 0; --r) m = __builtin_amdgcn_alignbit(m, (uint32_t)acc1[r], 31);
#pragma unroll
                    for (int r = 15; r >= 0; --r) m = __builtin_amdgcn_alignbit(m, (uint32_t)acc0[r], 31);
                    m = ~m;
                    unsigned long long any = __ballot(m != 0);
                    while (any) {  // one entry per lane and round (a dozen hits per patch: one or two rounds)
                        int wq = __builtin_amdgcn_readfirstlane(wq_n);
                        if (wq > MF_WQ - 64) {
                            mf_drain(S, D2, q0, n2c, wave, lane, wq, ori);
                            wq = 0;
                        }
                        if (m) {
                            const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                            m &= m - 1u;
                            const uint32_t row = ((bit >> 4) << 5) + (bit & 3u) + ((bit & 12u) << 1) + 4u * (uint32_t)h;
                            const int pq = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                            S.wq[wave][wq + pq] = (row << 16) | (uint32_t)(t * MF_TT + (lane & 31));
                        }
                        wq_n = wq + __popcll(any);
                        any = __ballot(m != 0);
                    }
                }
                else if (t + 1 < ntiles) expand(buf ^ 1, t + 1, min(MF_TT, cn - (t + 1) * MF_TT));
                __syncthreads();
            }
            // the entries point into this chunk's LDS copy: drain before the next chunk replaces it
            mf_drain(S, D2, q0, n2c, wave, lane, __builtin_amdgcn_readfirstlane(wq_n), ori);
            wq_n = 0;
        }
    }
    if (P.mfma_tiles && lane == 0 && n_mine) atomicAdd(P.mfma_tiles, (unsigned long long)n_mine);
    __syncthreads();
    // ---- flush: one thread per query: sort the row (ascending (dist, idx_1) = the reference's scan preference) in registers
    //      with a 16-input bitonic network (a data-dependent insertion sort in LDS costs a full row ~250 dependent LDS
    //      round trips, and the wave waits for its slowest lane), then four 16-byte stores.
    if (qb + tid < n2c) {
        const int cnt = S.cnt[tid], c = min(cnt, MF_SLOTS);
        uint32_t v[MF_SLOTS];
        const uint4* R = reinterpret_cast<const uint4*>(S.list[tid]);
#pragma unroll
        for (int i = 0; i < MF_SLOTS / 4; ++i) {
            const uint4 x = R[i];
            v[4 * i] = x.x;
            v[4 * i + 1] = x.y;
            v[4 * i + 2] = x.z;
            v[4 * i + 3] = x.w;
        }
#pragma unroll
        for (int i = 0; i < MF_SLOTS; ++i) v[i] = i < c ? v[i] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 2; k <= MF_SLOTS; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
                for (int i = 0; i < MF_SLOTS; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const uint32_t lo = min(v[i], v[l]), hi = max(v[i], v[l]);
                        const bool up = (i & k) == 0;
                        v[i] = up ? lo : hi;
                        v[l] = up ? hi : lo;
                    }
                }
        uint4* G = reinterpret_cast<uint4*>(P.topk + ((size_t)pair * P.cap2 + my_j) * BF_LIST);
#pragma unroll
        for (int i = 0; i < MF_SLOTS / 4; ++i) G[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        P.cnt[(size_t)pair * P.cap2 + my_j] = cnt;
    }
}

// decision of one query given the owner table; returns idx_1 or -1.  `head` = the first four keys of the row (kept in
// registers by the caller across sweeps: most rows hold one to three candidates), `tail` = keys 4.. of the row (the
// caller's LDS copy, or the row in global memory).
__device__ int bf_decide(const BfProblem& P, int pair, int j, int n1c, const uint4 head, const uint32_t* tail, int cnt, const int* owner) {
    if (cnt == 0) return -1;  // no candidate within dmax: best > dmax >= 50
    const int m = min(cnt, P.list_k);
    const bool truncated = cnt > P.list_k;
    auto key_at = [&](int k) -> uint32_t { return k == 0 ? head.x : k == 1 ? head.y : k == 2 ? head.z : k == 3 ? head.w : tail[k - 4]; };
    uint32_t a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu;
    for (int k = 0; k < m; ++k) {
        const uint32_t key = key_at(k);  // sorted ascending; usually m <= 3
        if (owner[key & 0xFFFFu] < j) continue;  // claimed by an earlier query (already_matched_indices_1)
        if (a0 == 0xFFFFFFFFu) a0 = key;
        else {
            a1 = key;
            break;
        }
    }
    // The list holds the (at most list_k) nearest of the candidates with dist <= dmax; if it is not truncated,
    // every unlisted candidate is farther than dmax.
    const unsigned d_beyond = truncated ? (key_at(m - 1) >> 16) : P.dmax + 1;  // lower bound on any unlisted distance
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
    return -3;  // undecidable from the prefix: the workgroup scans this row exactly (bf_full_row)
}

// Exact decision of one query by a cooperative scan of the whole row (robust.cc:271-314): every thread takes a
// strided share of the candidates, (best key, second distance) are combined with LDS atomicMin.
__device__ int bf_full_row(const BfProblem& P, int pair, int j, int n1c, const int* owner, uint32_t* s_best, uint32_t* s_second) {
    const uint32_t* D1 = P.desc1 + (size_t)bf_row1(P, pair) * P.cap1 * 8;
    const uint32_t* D2 = P.desc2 + ((size_t)pair * P.cap2 + j) * 8;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = D2[k];
    const float qa = P.angle2[((size_t)pair * P.cap2 + j) * P.angle_stride];
    if (threadIdx.x == 0) {
        *s_best = 0xFFFFFFFFu;
        *s_second = MAX_HAMMING_DIST;
    }
    __syncthreads();
    uint32_t k1 = 0xFFFFFFFFu;       // smallest (dist << 16 | idx) of this thread's share
    unsigned d2 = MAX_HAMMING_DIST;  // second smallest distance of this thread's share
    for (int i = threadIdx.x; i < n1c; i += blockDim.x) {
        if (owner[i] < j) continue;
        if (P.check_orientation && fabsf(angle_diff(P.angle1[((size_t)bf_row1(P, pair) * P.cap1 + i) * P.angle_stride], qa)) > 30.0f) continue;
        const uint32_t key = (hamming256(q, D1 + (size_t)i * 8) << 16) | (uint32_t)i;
        if (key < k1) {
            d2 = min(d2, k1 >> 16);
            k1 = key;
        }
        else d2 = min(d2, key >> 16);
    }
    if (k1 != 0xFFFFFFFFu) atomicMin(s_best, k1);
    __syncthreads();
    const uint32_t kb = *s_best;
    const unsigned mine = (k1 == kb) ? d2 : min(d2, k1 == 0xFFFFFFFFu ? MAX_HAMMING_DIST : (k1 >> 16));
    if (mine < MAX_HAMMING_DIST) atomicMin(s_second, mine);
    __syncthreads();
    const unsigned second = *s_second;
    __syncthreads();
    if (kb == 0xFFFFFFFFu) return -1;
    const unsigned best = kb >> 16;
    if (HAMMING_DIST_THR_LOW < best) return -1;
    if (P.lowe_ratio * (float)second < (float)best) return -1;
    return (int)(kb & 0xFFFFu);
}

// One workgroup per pair.  owner[n1] / match[n2] live in LDS when they fit, else in global scratch.  A thread owns the
// queries j = tid + u * blockDim; count and first four keys of the rows of its first BF_RC queries stay in registers, the rest of long rows in an LDS pool
// across the sweeps (enough for 4096 keypoints; further queries re-read their rows from global memory every sweep),
// so a sweep is LDS traffic and a handful of barriers.
// Two shapes: 1024 threads x 4 cached rows (up to 4096 keypoints in registers), and 512 threads x 5 rows for the usual <= 2560 keypoints:
// in the two-stream pipeline a 16-wave workgroup waits for a CU with 16 free wave slots while the extraction kernels hold them (2.1 ms
// elapsed for a 0.18 ms kernel at 1024 pairs); 8-wave workgroups find room four times as often.
template <int BF_RC, int NT>
__global__ __launch_bounds__(NT) void k_bf_replay(BfProblem P, int* __restrict__ g_owner, int* __restrict__ g_match, int use_lds, int pool_rows) {
    extern __shared__ int s_mem[];
    __shared__ int s_changed, s_nund, s_any_und, s_pool_n;
    __shared__ int s_und[1024];  // one slot per thread of a stride: can never overflow
    __shared__ uint32_t s_best, s_second;
    const int pair = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int n1 = P.n1_dev ? P.n1_dev[bf_row1(P, pair) * P.n_stride] : P.n1;
    const int n2 = P.n2_dev ? P.n2_dev[pair * P.n_stride] : P.n2;
    const int n1c = min(n1, P.cap1), n2c = min(n2, P.cap2);
    int* owner = use_lds ? s_mem : g_owner + (size_t)pair * P.cap1;
    int* match = use_lds ? s_mem + P.cap1 : g_match + (size_t)pair * P.cap2;
    uint32_t* pool = reinterpret_cast<uint32_t*>(s_mem + (use_lds ? ((P.cap1 + P.cap2 + 3) & ~3) : 0));  // pool_rows x (BF_LIST - 4) keys, 16-byte aligned
    const uint32_t* T = P.topk + (size_t)pair * P.cap2 * BF_LIST;
    const int* C = P.cnt + (size_t)pair * P.cap2;
    for (int i = tid; i < n1c; i += nthr) owner[i] = 0x7FFFFFFF;
    for (int j = tid; j < n2c; j += nthr) match[j] = -2;  // "unknown": forces at least one full sweep
    uint4 head[BF_RC];
    int hcnt[BF_RC];
    const uint32_t* tail[BF_RC];
    if (tid == 0) s_pool_n = 0;
    __syncthreads();
    // heads and counts of this thread's BF_RC rows first (independent loads, issued together), the long rows' tails afterwards
#pragma unroll
    for (int u = 0; u < BF_RC; ++u) {
        const int j = u * nthr + tid;
        hcnt[u] = 0;
        head[u] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        tail[u] = nullptr;
        if (j < n2c) {
            const uint32_t* R = T + (size_t)j * BF_LIST;
            hcnt[u] = C[j];
            head[u] = *reinterpret_cast<const uint4*>(R);
            tail[u] = R + 4;
        }
    }
#pragma unroll
    for (int u = 0; u < BF_RC; ++u)
        if (hcnt[u] > 4 && pool_rows > 0) {  // long rows (groups of look-alike keypoints) are walked deep every sweep: keep them in LDS
            const int slot = atomicAdd(&s_pool_n, 1);
            if (slot < pool_rows) {
                const uint32_t* R4 = tail[u];
                uint32_t* D = pool + (size_t)slot * (BF_LIST - 4);
#pragma unroll
                for (int i = 0; i < (BF_LIST - 4) / 4; ++i) reinterpret_cast<uint4*>(D)[i] = reinterpret_cast<const uint4*>(R4)[i];
                tail[u] = D;
            }
        }
    __syncthreads();
    for (int sweep = 0; sweep <= n2c; ++sweep) {
        if (tid == 0) {
            s_changed = 0;
            s_any_und = 0;
            s_nund = 0;
        }
        __syncthreads();
        // decisions against the owner table of the previous sweep (decisions read `owner`, never `match`)
        int dec[BF_RC];
        bool und = false;
#pragma unroll
        for (int u = 0; u < BF_RC; ++u) {
            const int j = u * nthr + tid;
            dec[u] = -1;
            if (j < n2c) {
                dec[u] = bf_decide(P, pair, j, n1c, head[u], tail[u], hcnt[u], owner);
                und = und || dec[u] == -3;
            }
        }
        if (und) s_any_und = 1;
        bool changed = false;
        for (int j = BF_RC * nthr + tid; j < n2c; j += nthr) {  // queries beyond the register-cached ones (more than 4096 keypoints)
            int d = bf_decide(P, pair, j, n1c, *reinterpret_cast<const uint4*>(T + (size_t)j * BF_LIST), T + (size_t)j * BF_LIST + 4, C[j], owner);
            if (d == -3) {  // resolved in the exact pass below; the previous decision old >= -2 is parked as -(old + 10) <= -8
                s_any_und = 1;
                match[j] = -(match[j] + 10);
            }
            else {
                changed = changed || d != match[j];
                match[j] = d;
            }
        }
        __syncthreads();
        if (s_any_und) {  // rare: some prefix was exhausted -> exact cooperative scans, one stride of queries at a time
            for (int j0 = 0; j0 < n2c; j0 += nthr) {
                const int j = j0 + tid, u = j0 / nthr;
                bool need = false;
                if (j < n2c) {
                    if (u < BF_RC) {
                        int du = dec[0];
#pragma unroll
                        for (int v = 1; v < BF_RC; ++v)
                            if (v == u) du = dec[v];
                        need = du == -3;
                    }
                    else need = match[j] <= -8;
                }
                if (need) s_und[atomicAdd(&s_nund, 1)] = j;
                __syncthreads();
                const int nund = s_nund;
                int mine = -3;
                for (int k = 0; k < nund; ++k) {
                    const int ju = s_und[k];
                    const int du = bf_full_row(P, pair, ju, n1c, owner, &s_best, &s_second);
                    if (ju == j) mine = du;
                }
                __syncthreads();
                if (tid == 0) s_nund = 0;
                if (need) {
                    if (u < BF_RC) {
#pragma unroll
                        for (int v = 0; v < BF_RC; ++v)
                            if (v == u) dec[v] = mine;
                    }
                    else {
                        changed = changed || mine != -match[j] - 10;
                        match[j] = mine;
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int u = 0; u < BF_RC; ++u) {
            const int j = u * nthr + tid;
            if (j < n2c) {
                changed = changed || dec[u] != match[j];
                match[j] = dec[u];
            }
        }
        if (changed) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;
        for (int i = tid; i < n1c; i += nthr) owner[i] = 0x7FFFFFFF;
        __syncthreads();
        for (int j = tid; j < n2c; j += nthr)
            if (match[j] >= 0) atomicMin(&owner[match[j]], j);
        __syncthreads();
    }
    // write-out: matched_2_in_1[idx_1] = idx_2 (robust.cc:317-325), unique by construction
    int32_t* out = P.matched + (size_t)pair * P.cap1;
    for (int i = tid; i < P.cap1; i += nthr) out[i] = -1;
    __syncthreads();
    int local = 0;
    for (int j = tid; j < n2c; j += nthr)
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
// are the entries of a list of n candidates kept sorted by (distance, scan position)?  Not beyond 1024 entries, and not in the modes whose outcome depends on the scan
// ORDER beyond ties: TRIANGULATION skips every candidate farther than the running best (bow_tree.cc:96-98 / robust.cc:89-91), so its
// "second" is the last superseded best, and AREA replays a non-monotone state.
#define CAND_SORT_MAX 1024
__device__ __forceinline__ int cand_total(const CandProblem& P) { return P.cand_total ? *P.cand_total : P.cand_off[P.nq]; }
__device__ __forceinline__ bool cand_sorted(const CandProblem& P, int n) { return n <= CAND_SORT_MAX && P.mode != SVGPU_MATCH_TRIANGULATION && P.mode != SVGPU_MATCH_AREA; }
__global__ void k_cand_dist(CandProblem P) {
    if (P.cap > 0 && cand_total(P) > P.cap) return;  // the lists did not fit the guessed capacity: the host re-runs with the exact size
    const int q = blockIdx.x;
    if (P.q_valid && !P.q_valid[q]) return;
    const int lo = P.cand_off[q], hi = P.cand_off[q + 1];
    __shared__ unsigned long long s_key[CAND_SORT_MAX];  // composites of a long list (more than one entry per lane)
    const bool long_sorted = hi - lo > 64 && cand_sorted(P, hi - lo);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = P.qdesc[(size_t)q * 8 + k];
    for (int c = lo + threadIdx.x; c < hi; c += blockDim.x) {
        const int t = P.cand_idx[c];
        // (a target that is occupied from the start is available to no query: gated here, it sorts behind every live entry instead of
        //  standing in front of them in every sweep of the replay)
        bool gated = (P.cand_skip && P.cand_skip[c]) || (P.occupied && P.occupied[t]);
        if (!gated && P.t_xright && 0.f < P.t_xright[t]) {
            const float err = fabsf(P.q_xright[q] - P.t_xright[t]);
            if (P.q_xr_tol[q] < err) gated = true;
        }
        if (!gated && P.check_orientation && fabsf(angle_diff(P.q_angle[q], P.t_angle[t])) > 30.0f) gated = true;
        if (!gated && P.chi_gate) {  // fuse.cc:92-119: chi-square test of the reprojection error at the keypoint's scale
            const double e_x = P.q_reproj[2 * q] - (double)P.t_xy[2 * t], e_y = P.q_reproj[2 * q + 1] - (double)P.t_xy[2 * t + 1];
            const double inv_sigma_sq = (double)P.inv_level_sigma_sq[(unsigned)P.t_octave[t]];
            if (P.chi_t_xright && P.chi_t_xright[t] >= 0.f) {
                const float e_xr = P.q_reproj_xr[q] - P.chi_t_xright[t];
                const double err = (e_x * e_x + e_y * e_y) + (double)(e_xr * e_xr);
                if ((double)7.81473f < err * inv_sigma_sq) gated = true;
            }
            else {
                const double err = e_x * e_x + e_y * e_y;
                if ((double)5.99146f < err * inv_sigma_sq) gated = true;
            }
        }
        uint32_t e = 0xFFFFFFFFu;
        if (!gated) {
            const unsigned d = hamming256(qd, P.tdesc + (size_t)t * 8);
            // a distance that can take part in no verdict (as a best only d <= thr counts, as a second only one that can fail the ratio test
            // of an acceptable best: lowe_ratio * d < thr) is as good as gated: it sorts behind the entries the replay has to look at.
            // Unrelated descriptors sit around 128 bits: nine in ten entries of a cell matcher's lists.
            bool useless = false;
            if (P.mode == SVGPU_MATCH_BEST_ONLY) useless = P.thr < d;
            else if (P.mode == SVGPU_MATCH_RATIO_SAME_OCTAVE) useless = !(P.lowe_ratio * (float)d < (float)P.thr) && P.thr < d;
            e = useless ? 0xFFFFFFFFu : ((uint32_t)d << 22) | (uint32_t)t;
        }
        P.dist[c] = e;
        if (long_sorted) s_key[c - lo] = e == 0xFFFFFFFFu ? ~0ull : ((unsigned long long)(e >> 22) << 32) | ((unsigned long long)(c - lo) << 22) | (e & 0x3FFFFFu);
    }
    // Lists of at most 64 entries are left SORTED by (distance, scan position) (cand_sorted): the replay's sweeps then look at the first
    // one or two still-available entries of a list instead of walking all of it -- the walk, one thread per query through ~50 scattered
    // entries per sweep, was 250 us of a tracked frame's match_current_and_last_frames.  One entry per lane, a bitonic network on the 64
    // composite keys (distance | position | target; gated entries and empty lanes sort last).
    const int n = hi - lo;
    if (!cand_sorted(P, n)) return;
    if (n <= 64) {
        const int c = lo + (int)threadIdx.x;
        const uint32_t e = c < hi ? P.dist[c] : 0xFFFFFFFFu;  // this lane's own store above
        unsigned long long key = e == 0xFFFFFFFFu ? ~0ull : ((unsigned long long)(e >> 22) << 32) | ((unsigned long long)threadIdx.x << 22) | (e & 0x3FFFFFu);
        const int lane = threadIdx.x;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j, 64);
                const bool up = (lane & k) == 0, low = (lane & j) == 0;  // ascending block, lower element of the pair
                const bool take_min = up == low;
                key = take_min ? (key < other ? key : other) : (key < other ? other : key);
            }
        if (c < hi) P.dist[c] = key == ~0ull ? 0xFFFFFFFFu : ((uint32_t)(key >> 32) << 22) | (uint32_t)(key & 0x3FFFFFu);
        return;
    }
    // longer lists (wide windows at the coarse levels; a handful per frame, but an unsorted one is walked to its end by ONE thread in every
    // sweep of the replay, and the slowest thread is what a sweep costs): the same network on the composites in LDS
    int npow2 = 128;
    while (npow2 < n) npow2 <<= 1;
    for (int i = n + threadIdx.x; i < npow2; i += 64) s_key[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += 64) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = s_key[i], b = s_key[p];
                    if ((a > b) == ((i & k) == 0)) {
                        s_key[i] = b;
                        s_key[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 64) {
        const unsigned long long key = s_key[i];
        P.dist[lo + i] = key == ~0ull ? 0xFFFFFFFFu : ((uint32_t)(key >> 32) << 22) | (uint32_t)(key & 0x3FFFFFu);
    }
}

__device__ __forceinline__ int cand_verdict(const CandProblem& P, unsigned best, unsigned second, int best_lvl, int second_lvl, int best_idx);
__device__ int cand_decide(const CandProblem& P, int q, const int* owner) {
    if (P.q_valid && !P.q_valid[q]) return -1;
    const int lo = P.cand_off[q], hi = P.cand_cnt ? lo + P.cand_cnt[q] : P.cand_off[q + 1];
    if (lo == hi) return -1;
    const bool tri = P.mode == SVGPU_MATCH_TRIANGULATION;
    unsigned best = tri ? P.thr : MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
    int best_lvl = -1, second_lvl = -1, best_idx = -1;
    if (cand_sorted(P, hi - lo)) {
        // sorted list: the scan below ends with best = the first available entry in (distance, position) order and second = the next one
        // (a superseded best is the (distance, position)-minimum so far, and an equal distance never replaces an earlier one), both only
        // when below MAX_HAMMING_DIST -- so the first two available entries decide
        for (int c = lo; c < hi; ++c) {
            const uint32_t e = P.dist[c];
            if (e == 0xFFFFFFFFu) break;  // gated entries sort last
            const unsigned d = e >> 22;
            if (d >= MAX_HAMMING_DIST) break;
            const int t = (int)(e & 0x3FFFFFu);
            if (owner[t] < q) continue;  // occupied before this query
            if (best_idx < 0) {
                best = d;
                best_idx = t;
                best_lvl = P.t_octave ? P.t_octave[t] : 0;
                if (P.mode == SVGPU_MATCH_BEST_ONLY) break;
            }
            else {
                second = d;
                second_lvl = P.t_octave ? P.t_octave[t] : 0;
                break;
            }
        }
    }
    else {
    auto visit = [&](uint32_t e) {
        if (e == 0xFFFFFFFFu) return;
        const unsigned d = e >> 22;
        const int t = (int)(e & 0x3FFFFFu);
        if (owner[t] < q) return;  // occupied before this query (initially, or by an earlier query)
        if (tri && (P.thr < d || best < d)) return;  // bow_tree.cc:96-98 / robust.cc:89-91
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
    };
    for (int c = lo; c < hi; c += 4) {  // four independent loads in flight per trip: the walk is latency-bound
        const uint32_t e0 = P.dist[c], e1 = c + 1 < hi ? P.dist[c + 1] : 0xFFFFFFFFu, e2 = c + 2 < hi ? P.dist[c + 2] : 0xFFFFFFFFu,
                       e3 = c + 3 < hi ? P.dist[c + 3] : 0xFFFFFFFFu;
        visit(e0);
        visit(e1);
        visit(e2);
        visit(e3);
    }
    }
    return cand_verdict(P, best, second, best_lvl, second_lvl, best_idx);  // (RATIO / TRIANGULATION: plain Lowe test, bow_tree.cc:226-233, :133-140)
}

// The final tests of cand_decide on (best, second) -- shared with the cached form below
__device__ __forceinline__ int cand_verdict(const CandProblem& P, unsigned best, unsigned second, int best_lvl, int second_lvl, int best_idx) {
    if (P.mode == SVGPU_MATCH_RATIO_SAME_OCTAVE) {
        if (best <= P.thr) {
            if (best_lvl == second_lvl && (float)best > P.lowe_ratio * (float)second) return -1;
            return best_idx;
        }
        return -1;
    }
    if (P.mode == SVGPU_MATCH_BEST_ONLY) {
        if (P.thr < best) return -1;
        return best_idx;
    }
    if (P.thr < best || best_idx < 0) return -1;
    if (P.lowe_ratio * (float)second < (float)best) return -1;
    return best_idx;
}

// Fixed-point replay of the reference's sequential greedy loop (a query takes its best target among those no EARLIER query holds): sweeps of
// "every query decides against the owner table of the previous sweep, then the winners claim" until nothing changes (8 sweeps for a tracked
// frame's match_current_and_last_frames).  What a sweep costs is the walk of every query through its list until two still-available entries
// are found -- late in the replay most of a list is taken, and a walk through global memory is one dependent round trip (~1 us) per entry:
// 12.7 us per sweep, 100 of the kernel's 123 us (time stamps inside the kernel).  So the whole state lives in the workgroup's LDS: owner and
// match tables, the list offsets, the initial occupancy and the levels as bytes, and the first K entries of every SORTED list (K = what
// fits: 12 for 2 400 queries, 5 for 4 800); only entries beyond K and unsorted lists (more than 64 candidates, order-dependent modes) are read
// from global memory, four at a time.
// (LDS pointers carry their address space: through a generic pointer every access is a FLAT instruction -- global-memory latency for
//  on-chip data; the replay's decisions are chains of six dependent table look-ups, and they were 12 us per sweep that way)
#define SV_LDS __attribute__((address_space(3)))
struct CandLds {
    SV_LDS int* owner;      // nt
    SV_LDS int* match;      // nq
    SV_LDS int* off;        // nq + 1
    SV_LDS int* cnt;        // nq (null: CSR offsets, list q ends where list q + 1 begins)
    SV_LDS uint32_t* head;  // nq x K
    SV_LDS uint8_t* occ;    // nt (null: nothing occupied initially)
    SV_LDS uint8_t* lvl;    // nt (null: no levels)
    SV_LDS uint8_t* qv;     // nq (null: every query is valid)
    int K;
};
__device__ __forceinline__ int cand_decide_lds(const CandProblem& P, int q, const CandLds& S) {
    const int lo = S.off[q], n = S.cnt ? S.cnt[q] : S.off[q + 1] - lo;
    if (n == 0 || (S.qv && !S.qv[q])) return -1;
    if (!cand_sorted(P, n)) return cand_decide(P, q, (const int*)S.owner);
    unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
    int best_lvl = -1, second_lvl = -1, best_idx = -1;
    bool done = false;
    auto take_loaded = [&](uint32_t e, int own, int lv) {  // the first two available entries in (distance, position) order decide (cand_decide)
        if (done) return;
        if (e == 0xFFFFFFFFu || (e >> 22) >= MAX_HAMMING_DIST) {
            done = true;
            return;
        }
        const int t = (int)(e & 0x3FFFFFu);
        if (own < q) return;
        if (best_idx < 0) {
            best = e >> 22;
            best_idx = t;
            best_lvl = lv;
            if (P.mode == SVGPU_MATCH_BEST_ONLY) done = true;
        }
        else {
            second = e >> 22;
            second_lvl = lv;
            done = true;
        }
    };
    auto take = [&](uint32_t e) {
        if (done || e == 0xFFFFFFFFu) {
            take_loaded(e, 0, 0);
            return;
        }
        const int t = (int)(e & 0x3FFFFFu);
        take_loaded(e, S.owner[t], S.lvl ? (int)S.lvl[t] : 0);
    };
    const int nk = min(n, S.K);
    const SV_LDS uint32_t* h = S.head + (size_t)q * S.K;
    // The staged head four entries at a time: the four entries, then their four owners and levels, are loaded TOGETHER and decided in
    // registers -- three LDS round trips per four entries.  Entry by entry the walk was two dependent LDS accesses per entry (the owner's
    // address comes out of the entry), and a wave walks as long as its slowest lane.  (Measured with stamps in the kernel: the decisions are
    // 3.8 of a sweep's 6 us at 2 400 queries, NO lane walks past its staged head in the tracked-frame workload -- holding the next four
    // entries of every list in registers changed nothing --, the rest is the sum of the per-chunk branches of 16 waves on one CU.)
    for (int k0 = 0; k0 < nk && !done; k0 += 4) {
        uint32_t e[4];
        int own[4], lv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = k0 + j < nk ? h[k0 + j] : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = e[j] == 0xFFFFFFFFu ? 0 : (int)(e[j] & 0x3FFFFFu);  // (a sentinel's loads are speculative: any valid slot)
            own[j] = S.owner[t];
            lv[j] = S.lvl ? (int)S.lvl[t] : 0;
        }
        const bool best_only = P.mode == SVGPU_MATCH_BEST_ONLY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // take_loaded without branches (selects): the four decisions of a chunk are straight-line code
            const unsigned d = e[j] >> 22;
            // (slots of the chunk beyond the staged head are neither an entry nor the list's end: the walk goes on in global memory.  Reading
            //  them as "gated entries sort last -> stop" ended the walk of a list LONGER than its staged head whose staged entries were all
            //  taken -- with one staged entry per list, 6 000 queries, a handful of wrong decisions per frame)
            const bool padding = k0 + j >= nk;
            const bool stop = !padding && (e[j] == 0xFFFFFFFFu || d >= MAX_HAMMING_DIST), active = !done;
            const bool avail = active && !padding && !stop && own[j] >= q, first = avail && best_idx < 0, sec = avail && best_idx >= 0;
            best = first ? d : best;
            best_lvl = first ? lv[j] : best_lvl;
            second = sec ? d : second;
            second_lvl = sec ? lv[j] : second_lvl;
            best_idx = first ? (int)(e[j] & 0x3FFFFFu) : best_idx;
            done = done || (active && stop) || sec || (first && best_only);
        }
    }
    for (int c = lo + nk; c < lo + n && !done; c += 4) {  // beyond the staged head: four independent loads per trip
        const int hi = lo + n;
        const uint32_t e0 = P.dist[c], e1 = c + 1 < hi ? P.dist[c + 1] : 0xFFFFFFFFu, e2 = c + 2 < hi ? P.dist[c + 2] : 0xFFFFFFFFu,
                       e3 = c + 3 < hi ? P.dist[c + 3] : 0xFFFFFFFFu;
        take(e0);
        take(e1);
        take(e2);
        take(e3);
    }
    return cand_verdict(P, best, second, best_lvl, second_lvl, best_idx);
}
#define CAND_LDS_BUDGET (150 * 1024)
// bytes of the LDS-resident form for (nq, nt) with K staged entries per list
__host__ __device__ inline size_t cand_lds_bytes(int nq, int nt, int K, bool with_cnt = false) {
    return (size_t)(nt + nq + nq + 1 + (with_cnt ? nq : 0)) * 4 + (size_t)nq * K * 4 + 2 * (((size_t)nt + 3) & ~size_t(3)) + (((size_t)nq + 3) & ~size_t(3)) + 16;
}
// Order of evaluation.  A query's decision depends only on EARLIER queries, so the queries are taken in chunks of increasing index and the
// fixed point is run chunk by chunk: while a chunk iterates, everything earlier is final (its claimed targets are simply closed, owner -1),
// and a sweep touches the chunk's queries only.  The sweeps a chunk needs are the depth of the dependency chains INSIDE it (two or three)
// instead of the depth over the whole frame (eight for a tracked frame's match_current_and_last_frames), and each costs a fraction of a
// sweep over everything: 67 -> about 30 us at 2 400 - 4 800 queries.  The result is the serial loop's, as before: within a chunk the
// iteration is the one described above, and across chunks the order is the serial order.
#define CAND_CHUNK_QPT 1  // queries per thread and chunk
__global__ __launch_bounds__(1024) void k_cand_replay_lds(CandProblem P, int K) {
    extern __shared__ int s_cand[];
    __shared__ int s_changed;
    if (P.cap > 0 && cand_total(P) > P.cap) {  // the lists did not fit: nothing was written, the host re-runs with the exact size
        if (P.num_host && threadIdx.x == 0) P.num_host[0] = 0, P.num_host[1] = cand_total(P);
        return;
    }
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nt = P.nt_dev ? min(*P.nt_dev, P.nt) : P.nt;  // (the tables are laid out for P.nt either way)
    CandLds S;
    S.owner = (SV_LDS int*)s_cand;
    S.match = S.owner + P.nt;
    S.off = S.match + P.nq;
    S.cnt = P.cand_cnt ? S.off + P.nq + 1 : nullptr;
    S.head = (SV_LDS uint32_t*)(S.off + P.nq + 1 + (P.cand_cnt ? P.nq : 0));
    SV_LDS uint8_t* bytes = (SV_LDS uint8_t*)(S.head + (size_t)P.nq * K);
    S.occ = P.occupied ? bytes : nullptr;
    S.lvl = P.t_octave ? bytes + ((P.nt + 3) & ~3) : nullptr;
    S.qv = P.q_valid ? bytes + 2 * ((P.nt + 3) & ~3) : nullptr;
    S.K = K;
    for (int q = tid; q <= P.nq; q += nthr) {
        S.off[q] = P.cand_off[q];
        if (S.cnt && q < P.nq) S.cnt[q] = P.cand_cnt[q];
        if (S.qv && q < P.nq) S.qv[q] = P.q_valid[q];
    }
    for (int t = tid; t < nt; t += nthr) {
        S.owner[t] = (P.occupied && P.occupied[t]) ? -1 : 0x7FFFFFFF;  // -1: closed for every query (initially occupied, or taken by a finished chunk)
        if (S.lvl) S.lvl[t] = (uint8_t)P.t_octave[t];
    }
    S.occ = nullptr;  // (the owner table carries it)
    __syncthreads();
    for (int i = tid; i < P.nq * K; i += nthr) {
        const int q = i / K, k = i - q * K, lo = S.off[q], n = S.cnt ? S.cnt[q] : S.off[q + 1] - lo;
        if (k < n) S.head[i] = P.dist[lo + k];
    }
    if (tid == 0) s_changed = 0;
    __syncthreads();
    if (P.dbg_phase == 1) return;
    int local = 0, sweeps_total = 0;
    const int chunk = nthr * CAND_CHUNK_QPT;
    for (int lo = 0; lo < P.nq; lo += chunk) {
        int mine[CAND_CHUNK_QPT], prev[CAND_CHUNK_QPT];
#pragma unroll
        for (int u = 0; u < CAND_CHUNK_QPT; ++u) mine[u] = -2, prev[u] = -1;
        for (int sweep = 0; sweep <= chunk; ++sweep) {  // three barriers per sweep
            ++sweeps_total;
            int local_changed = 0;
#pragma unroll
            for (int u = 0; u < CAND_CHUNK_QPT; ++u) {
                const int q = lo + u * nthr + tid;
                if (q < P.nq) {
                    const int d = cand_decide_lds(P, q, S);
                    if (d != mine[u]) local_changed = 1;
                    mine[u] = d;
                }
            }
            if (local_changed) s_changed = 1;
            __syncthreads();
            const bool again = s_changed != 0 && !P.no_claims && P.dbg_phase != 2;  // (without claims the queries are independent: one evaluation is the answer)
            if (!again) break;
            // the claims of the previous sweep go, the new ones come: owner[t] = the earliest query of the chunk that wants t
#pragma unroll
            for (int u = 0; u < CAND_CHUNK_QPT; ++u)
                if (prev[u] >= 0) S.owner[prev[u]] = 0x7FFFFFFF;
            __syncthreads();  // every thread has tested the flag and withdrawn: it can be cleared, and the new claims can land
            if (tid == 0) s_changed = 0;
#pragma unroll
            for (int u = 0; u < CAND_CHUNK_QPT; ++u) {
                const int q = lo + u * nthr + tid;
                const int m = mine[u];
                prev[u] = (q < P.nq && m >= 0 && (!P.q_blocks || P.q_blocks[q])) ? m : -1;
                if (prev[u] >= 0) __hip_atomic_fetch_min(S.owner + prev[u], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
        }
        // the chunk is final: its targets are closed for everything that follows, its matches go out
        __syncthreads();  // (every thread is past its last read of the flag and of the owner table)
        if (tid == 0) s_changed = 0;
#pragma unroll
        for (int u = 0; u < CAND_CHUNK_QPT; ++u) {
            const int q = lo + u * nthr + tid;
            if (q >= P.nq) continue;
            if (prev[u] >= 0) S.owner[prev[u]] = -1;
            const int m = mine[u];
            P.match_q[q] = m;
            if (P.match_host) P.match_host[q] = m;
            local += m >= 0;
        }
        __syncthreads();
    }
    if (local) atomicAdd(&s_changed, local);
    __syncthreads();
    if (tid == 0) {
        *P.num = s_changed;
        if (P.num_host) P.num_host[0] = s_changed, P.num_host[1] = cand_total(P), P.num_host[2] = sweeps_total;
    }
}
// the same replay with its tables in global memory (inputs beyond the LDS form)
__global__ __launch_bounds__(1024) void k_cand_replay(CandProblem P, int* __restrict__ owner, int* __restrict__ match) {
    __shared__ int s_changed;
    if (P.cap > 0 && cand_total(P) > P.cap) {
        if (P.num_host && threadIdx.x == 0) P.num_host[0] = 0, P.num_host[1] = cand_total(P);
        return;
    }
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nt = P.nt_dev ? min(*P.nt_dev, P.nt) : P.nt;
    auto reset_owner = [&]() {
        for (int t = tid; t < nt; t += nthr) owner[t] = (P.occupied && P.occupied[t]) ? -1 : 0x7FFFFFFF;
    };
    reset_owner();
    for (int q = tid; q < P.nq; q += nthr) match[q] = -2;
    if (tid == 0) s_changed = 0;
    __syncthreads();
    for (int sweep = 0; sweep <= P.nq; ++sweep) {
        int local_changed = 0;
        for (int q = tid; q < P.nq; q += nthr) {
            const int d = cand_decide(P, q, owner);
            if (d != match[q]) local_changed = 1;
            match[q] = d;
        }
        if (local_changed) s_changed = 1;
        __syncthreads();
        if (!s_changed) break;
        reset_owner();
        __syncthreads();
        if (tid == 0) s_changed = 0;
        if (!P.no_claims)
            for (int q = tid; q < P.nq; q += nthr)
                if (match[q] >= 0 && (!P.q_blocks || P.q_blocks[q])) atomicMin(&owner[match[q]], q);
        __syncthreads();
    }
    int local = 0;
    for (int q = tid; q < P.nq; q += nthr) {
        P.match_q[q] = match[q];
        if (P.match_host) P.match_host[q] = match[q];
        local += match[q] >= 0;
    }
    if (tid == 0) s_changed = 0;
    __syncthreads();
    if (local) atomicAdd(&s_changed, local);
    __syncthreads();
    if (tid == 0) {
        *P.num = s_changed;
        if (P.num_host) P.num_host[0] = s_changed, P.num_host[1] = cand_total(P);
    }
}

// ------------------------------------------------------------------------------------------------ grid candidate lists
// data::assign_keypoints_to_grid (data/common.cc:83-108) and data::get_keypoints_in_cell (:127-190) on the device, so that the
// projection-family matchers need no host-built CSR: cells are x-major (col * rows + row), a cell lists its keypoints in
// index order, a query scans cells col-major inside its window and keeps keypoints of the wanted levels strictly inside
// the margin square -- the reference's candidate ORDER, which decides ties downstream.
__global__ void k_grid_assign(GridProblem G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.nt) return;
    const int cx = (int)floor((double)(G.t_xy[2 * i] - G.min_x) * G.inv_w), cy = (int)floor((double)(G.t_xy[2 * i + 1] - G.min_y) * G.inv_h);
    int c = -1;
    if (0 <= cx && cx < G.cols && 0 <= cy && cy < G.rows) {
        c = cx * G.rows + cy;
        atomicAdd(&G.cell_off[c], 1);  // counts; scanned in place afterwards
    }
    G.cell_of[i] = c;
}
// in-place exclusive scan of data[0..n) with the total written to data[n]; single workgroup
__global__ __launch_bounds__(1024) void k_exclusive_scan(int32_t* __restrict__ data, int n) {
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
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
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
// stable placement: rank inside the cell = number of earlier keypoints of the same cell (cell ids stream through LDS tiles)
__global__ __launch_bounds__(256) void k_grid_place(GridProblem G) {
    __shared__ int s_cell[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = i < G.nt ? G.cell_of[i] : -1;
    const int last = min(blockIdx.x * 256 + 255, G.nt - 1);  // tiles beyond the block's last keypoint hold no earlier keypoint
    int rank = 0;
    for (int base = 0; base <= last; base += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024; k += 256) s_cell[k] = base + k < G.nt ? G.cell_of[base + k] : -2;
        __syncthreads();
        const int m = min(1024, i - base);  // earlier keypoints only
        for (int k = 0; k < m; ++k) rank += s_cell[k] == c;
    }
    if (c >= 0) G.cell_items[G.cell_off[c] + rank] = i;
}
// The keypoint side of the matcher grid in ONE launch of one workgroup (grids up to GRID_ONE_CELLS cells; the reference's is 64 x 48):
// cell of every keypoint + its arrival number in the cell (LDS counters), exclusive scan of the counters (-> cell_off), unordered placement,
// then every cell with more than one keypoint puts its items into increasing order -- the same arrays as the four launches below
// (memset, k_grid_assign, scan, k_grid_place), whose last one counted, for every keypoint, ALL earlier keypoints of its cell by walking
// every earlier keypoint: 26 us at 2 400 keypoints.
__global__ __launch_bounds__(1024) void k_grid_frame_one(GridProblem G) {
    grid_frame_one(G, G.nt);
}
// one wave per query: the lanes take the cells of the window (column-major = the reference's scan order), count their
// keypoints that pass the level and margin tests, and a wave prefix sum gives every cell its place in the query's list
template <bool FILL>
__global__ __launch_bounds__(256) void k_grid_walk(GridProblem G) {
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // scalar: per-query data via scalar loads
    if (q >= G.nq) return;
    if (FILL && G.cap > 0 && G.cand_off[G.nq] > G.cap) return;  // capacity guess too small: nothing is written, the host re-runs
    int total = 0;
    const bool live = !G.q_valid || G.q_valid[q];
    if (live) {
        const float ref_x = G.q_xy[2 * q], ref_y = G.q_xy[2 * q + 1], margin = G.q_margin[q];
        const int min_level = G.q_min_level ? G.q_min_level[q] : -1, max_level = G.q_max_level ? G.q_max_level[q] : -1;
        int lo_x = (int)floor((double)(ref_x - G.min_x - margin) * G.inv_w), hi_x = (int)ceil((double)(ref_x - G.min_x + margin) * G.inv_w);
        int lo_y = (int)floor((double)(ref_y - G.min_y - margin) * G.inv_h), hi_y = (int)ceil((double)(ref_y - G.min_y + margin) * G.inv_h);
        lo_x = max(lo_x, 0);
        lo_y = max(lo_y, 0);
        hi_x = min(hi_x, G.cols - 1);
        hi_y = min(hi_y, G.rows - 1);
        if (lo_x < G.cols && 0 <= hi_x && lo_y < G.rows && 0 <= hi_y && lo_x <= hi_x && lo_y <= hi_y) {
            const int ny = hi_y - lo_y + 1, ncell = (hi_x - lo_x + 1) * ny;
            int32_t* out = FILL ? G.cand_idx + G.cand_off[q] : nullptr;
            for (int base = 0; base < ncell; base += 64) {
                const int k = base + lane;
                int n = 0, c = 0;
                if (k < ncell) {
                    c = (lo_x + k / ny) * G.rows + lo_y + k % ny;
                    for (int it = G.cell_off[c]; it < G.cell_off[c + 1]; ++it) {
                        const int idx = G.cell_items[it];
                        const int oct = G.t_octave[idx];
                        if (0 <= min_level && oct < min_level) continue;
                        if (0 <= max_level && max_level < oct) continue;
                        const float dx = G.t_xy[2 * idx] - ref_x, dy = G.t_xy[2 * idx + 1] - ref_y;
                        if (fabsf(dx) < margin && fabsf(dy) < margin) ++n;
                    }
                }
                // inclusive prefix sum over the lanes: DPP row scans (VALU; six __shfl_up were six LDS-crossbar round trips on every query's chain)
                int incl = n;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8: scans inside the four rows of 16
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
                if (FILL && n > 0) {
                    int o = total + incl - n;
                    for (int it = G.cell_off[c]; it < G.cell_off[c + 1]; ++it) {
                        const int idx = G.cell_items[it];
                        const int oct = G.t_octave[idx];
                        if (0 <= min_level && oct < min_level) continue;
                        if (0 <= max_level && max_level < oct) continue;
                        const float dx = G.t_xy[2 * idx] - ref_x, dy = G.t_xy[2 * idx + 1] - ref_y;
                        if (fabsf(dx) < margin && fabsf(dy) < margin) out[o++] = idx;
                    }
                }
                total += __builtin_amdgcn_readlane(incl, 63);
            }
        }
    }
    if (!FILL && lane == 0) G.cand_off[q] = total;
}

// area::match_in_consistent_area (match/area.cc:8-98) on the same CSR lists and distances (mode SVGPU_MATCH_AREA).
// Its state is not monotone (a later, closer query takes a target away from its holder), so the loop over the queries
// stays sequential: ONE wave walks the queries in order, its lanes stride over the candidates of the current query
// (distances are precomputed), wave-reduce (best (dist, scan position), second distance), lane 0 applies the update.
// The initialiser calls this once per frame pair on level-0 keypoints only.
__global__ __launch_bounds__(64) void k_area_replay(CandProblem P, int* __restrict__ g_holder, int* __restrict__ g_match,
                                                    unsigned* __restrict__ g_mdist, int use_lds) {
    extern __shared__ int s_area[];
    const int lane = threadIdx.x;
    int* holder = use_lds ? s_area : g_holder;
    unsigned* mdist = use_lds ? reinterpret_cast<unsigned*>(s_area + P.nt) : g_mdist;
    int* match = use_lds ? s_area + 2 * P.nt : g_match;
    for (int t = lane; t < P.nt; t += 64) {
        holder[t] = -1;
        mdist[t] = MAX_HAMMING_DIST;
    }
    for (int q = lane; q < P.nq; q += 64) match[q] = -1;
    __syncthreads();
    // state in global memory (inputs too large for LDS): device-scope atomics keep lane 0's updates visible to the other lanes
    auto ld = [&](const unsigned* p) -> unsigned {
        return use_lds ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto st = [&](unsigned* p, unsigned v) {
        if (use_lds) *p = v;
        else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (int q = 0; q < P.nq; ++q) {
        if (P.q_valid && !P.q_valid[q]) continue;
        const int lo = P.cand_off[q], hi = P.cand_off[q + 1];
        if (lo == hi) continue;
        uint32_t k1 = 0xFFFFFFFFu;       // smallest (dist << 20 | scan position) of this lane's share: strict '<' keeps the first
        unsigned d2 = MAX_HAMMING_DIST;  // second smallest distance of this lane's share
        for (int c = lo + lane; c < hi; c += 64) {
            const uint32_t ent = P.dist[c];
            if (ent == 0xFFFFFFFFu) continue;           // cand_skip / orientation gate (:40-42)
            const unsigned d = ent >> 22;
            if (ld(&mdist[ent & 0x3FFFFFu]) <= d) continue;  // the current holder of that target is at least as close (:47-50)
            const uint32_t key = (d << 20) | (uint32_t)min(c - lo, 0xFFFFF);
            if (key < k1) {
                d2 = min(d2, k1 >> 20);
                k1 = key;
            }
            else d2 = min(d2, d);
        }
        uint32_t kb = k1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kb = min(kb, (uint32_t)__shfl_xor(kb, off, 64));
        unsigned second = (k1 == kb) ? d2 : min(d2, k1 == 0xFFFFFFFFu ? MAX_HAMMING_DIST : (k1 >> 20));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) second = min(second, (unsigned)__shfl_xor(second, off, 64));
        if (kb == 0xFFFFFFFFu) continue;
        const unsigned best = kb >> 20;
        if (P.thr < best) continue;                                   // HAMMING_DIST_THR_LOW (:58-60)
        if ((float)second * P.lowe_ratio < (float)best) continue;     // ratio test (:63-65)
        const int t = P.cand_idx[lo + (int)(kb & 0xFFFFFu)];
        if (lane == 0) {                                              // take the target from its previous holder (:72-86)
            const int prev = (int)ld(reinterpret_cast<unsigned*>(&holder[t]));
            if (0 <= prev) st(reinterpret_cast<unsigned*>(&match[prev]), 0xFFFFFFFFu);
            st(reinterpret_cast<unsigned*>(&match[q]), (unsigned)t);
            st(reinterpret_cast<unsigned*>(&holder[t]), (unsigned)q);
            st(&mdist[t], best);
        }
        __syncthreads();  // single wave: orders lane 0's LDS / global writes before the next query's reads
    }
    int local = 0;
    __syncthreads();
    for (int q = lane; q < P.nq; q += 64) {
        const int m = (int)ld(reinterpret_cast<const unsigned*>(&match[q]));
        P.match_q[q] = m;
        local += m >= 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if (lane == 0) *P.num = local;
}

// ------------------------------------------------------------------------------------------------ stereo
// match::stereo::compute (match/stereo.cc:20-251), one wave per left keypoint.
//   phase 1: lanes stride over the right keypoints: row-band membership (get_right_keypoint_indices_in_each_row, margin 2),
//            octave +-1, disparity range, Hamming; wave-min of (dist << 16 | idx) = the reference's first strict minimum
//   phase 2: 11 patch offsets x 121 pixels, integer L1 sums (exact: cv::norm accumulates integer-valued floats in double),
//            parabola in double as the reference's mixed float/double expression.
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// The row index of the right keypoints, as the reference builds it (stereo.cc:34-60): keypoint ir is listed in every row of
// [floor(y - 2 sf), ceil(y + 2 sf)] of the level-0 image.  k_stereo used to test EVERY right keypoint against every left one (3 700 x 3 700
// band tests per pair, 0.64 ms for 8 pairs); with the index a left keypoint walks the ~60 entries of its own row.  Count -> scan -> fill;
// the order inside a row is whatever the atomics give, which is harmless: the matcher takes the minimum of (distance, index).
struct StereoPair {  // one pair's slice of the batched arrays (the kernel argument itself is never modified: that would send it to scratch)
    const svgpu_keypoint *kl, *kr;
    const uint32_t *dl, *dr;
    int nl, nr;
    float *xr, *depth, *corr;
    int32_t *row_off, *row_fill, *row_items;
};
__device__ __forceinline__ StereoPair stereo_pair(const StereoProblem& P, int pair) {
    StereoPair Q;
    const size_t o = P.nl_dev ? (size_t)pair * P.cap : 0;
    Q.nl = P.nl_dev ? min(P.nl_dev[(size_t)pair * P.n_stride], P.cap) : P.nl;
    Q.nr = P.nr_dev ? min(P.nr_dev[(size_t)pair * P.n_stride], P.cap) : P.nr;
    Q.kl = P.kl + o, Q.kr = P.kr + o, Q.dl = P.dl + o * 8, Q.dr = P.dr + o * 8;
    Q.xr = P.xr + o, Q.depth = P.depth + o, Q.corr = P.corr + o;
    const int nr_cap = P.nl_dev ? P.cap : P.nr;
    Q.row_off = P.row_off + (size_t)pair * (P.rows + 1);
    Q.row_fill = P.row_fill + (size_t)pair * P.rows;
    Q.row_items = P.row_items + (size_t)pair * nr_cap * P.rows_per_kp;
    return Q;
}
__device__ __forceinline__ void stereo_band(const StereoProblem& P, const svgpu_keypoint& r, int& min_r, int& max_r) {
    const float rad = 2.0f * P.sf[r.octave];
    max_r = min((int)ceil((double)(r.y + rad)), P.rows - 1);
    min_r = max((int)floor((double)(r.y - rad)), 0);
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_stereo_rows(StereoProblem P) {
    const StereoPair Q = stereo_pair(P, blockIdx.y);
    const int ir = blockIdx.x * 256 + threadIdx.x;
    if (ir >= Q.nr) return;
    int min_r, max_r;
    stereo_band(P, Q.kr[ir], min_r, max_r);
    for (int row = min_r; row <= max_r; ++row) {
        if (FILL) Q.row_items[Q.row_off[row] + atomicAdd(&Q.row_fill[row], 1)] = ir;
        else atomicAdd(&Q.row_off[row], 1);
    }
}
__global__ __launch_bounds__(1024) void k_stereo_rows_scan(StereoProblem P) {  // one workgroup per pair: counts -> offsets (rows + 1 entries)
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    int32_t* off = P.row_off + (size_t)blockIdx.x * (P.rows + 1);
    const int tid = threadIdx.x, n = P.rows;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? off[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d) incl += t;
        }
        if ((tid & 63) == 63) s_part[tid >> 6] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < (tid >> 6); ++w) before += s_part[w];
        const int carry = s_carry;
        if (i < n) off[i] = carry + before + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + before + incl;
        __syncthreads();
    }
    if (tid == 0) off[n] = s_carry;
}
__global__ __launch_bounds__(256) void k_stereo(StereoProblem P) {
    const int il = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // scalar: per-keypoint data via scalar loads
    const int pair = blockIdx.y;
    const StereoPair Q = stereo_pair(P, pair);
    if (il >= Q.nl) return;
    const svgpu_keypoint k = Q.kl[il];
    const int lvl = k.octave;
    float out_xr = -1.0f, out_depth = -1.0f, out_corr = -1.0f;
    const int row = (int)k.y;
    const float min_x_right = k.x - P.max_disp, max_x_right = k.x - P.min_disp;
    uint32_t best = 0xFFFFFFFFu;
    if (!(max_x_right < 0) && row >= 0 && row < P.rows) {
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = Q.dl[(size_t)il * 8 + w];
        for (int c = Q.row_off[row] + lane; c < Q.row_off[row + 1]; c += 64) {  // the right keypoints whose band holds this row
            const int ir = Q.row_items[c];
            const svgpu_keypoint r = Q.kr[ir];
            if (r.octave < lvl - 1 || r.octave > lvl + 1) continue;
            if (r.x < min_x_right || max_x_right < r.x) continue;
            const unsigned d = hamming256(q, Q.dr + (size_t)ir * 8);
            if (d < P.thr) best = min(best, (d << 16) | (uint32_t)ir);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
    if (best != 0xFFFFFFFFu) {
        const float x_right = Q.kr[best & 0xFFFFu].x;
        const float isf = P.isf[lvl];
        const int sxl = __float2int_rn(k.x * isf), syl = __float2int_rn(k.y * isf), sxr = __float2int_rn(x_right * isf);
        const int ini_x = sxr - 10, end_x = sxr + 10;
        const bool ok = !(ini_x < 0 || P.w[lvl] <= end_x) && syl - 5 >= 0 && syl + 5 < P.h[lvl] && sxl - 5 >= 0 && sxl + 5 < P.w[lvl];
        if (ok) {
            const uint8_t* PL = P.lev_l[lvl] + (size_t)pair * (lvl == 0 ? P.img_stride_l : P.pyr_stride_l);
            const uint8_t* PR = P.lev_r[lvl] + (size_t)pair * (lvl == 0 ? P.img_stride_r : P.pyr_stride_r);
            const int pl = P.pitch_l[lvl], pr = P.pitch_r[lvl];
            const int lc = PL[__umul24(syl, pl) + sxl];
            // this lane's (up to) two patch pixels
            int a[2], dy[2], dx[2];
            bool has[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int p = lane + 64 * t;
                has[t] = p < 121;
                dy[t] = has[t] ? p / 11 - 5 : 0;
                dx[t] = has[t] ? p % 11 - 5 : 0;
                a[t] = has[t] ? (int)PL[__umul24(syl + dy[t], pl) + sxl + dx[t]] - lc : 0;
            }
            float corr[11];
            float best_corr = 3.402823466e+38f;
            int best_off = 0;
#pragma unroll
            for (int o = 0; o < 11; ++o) {
                const int off = o - 5;
                const int rc = PR[__umul24(syl, pr) + sxr + off];
                int acc = 0;
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (has[t]) {
                        const int b = (int)PR[__umul24(syl + dy[t], pr) + sxr + off + dx[t]] - rc;
                        acc += abs(a[t] - b);
                    }
                const float c = (float)wave_sum_i(acc);
                corr[o] = c;
                if (c < best_corr) {
                    best_corr = c;
                    best_off = off;
                }
            }
            if (best_off != -5 && best_off != 5) {
                float c1 = corr[0], c2 = corr[1], c3 = corr[2];
#pragma unroll
                for (int o = 1; o < 10; ++o)
                    if (o == best_off + 5) {
                        c1 = corr[o - 1];
                        c2 = corr[o];
                        c3 = corr[o + 1];
                    }
                const float x_delta = (float)((double)(c1 - c3) / (2.0 * (double)(c1 + c3) - 4.0 * (double)c2));
                if (!((double)x_delta < -1.0 || 1.0 < (double)x_delta)) {
                    float best_x_right = P.sf[lvl] * ((float)(sxr + best_off) + x_delta);
                    float best_disp = k.x - best_x_right;
                    if (!(best_disp < P.min_disp || P.max_disp <= best_disp)) {
                        if (best_disp <= 0.0f) {
                            best_disp = 0.01f;
                            best_x_right = k.x - best_disp;
                        }
                        out_depth = P.fxb / best_disp;
                        out_xr = best_x_right;
                        out_corr = best_corr;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        Q.xr[il] = out_xr;
        Q.depth[il] = out_depth;
        Q.corr[il] = out_corr;
    }
}

// stereo.cc:94-113 per pair: median of the kept correlations = the element of rank size / 2 in ascending order; matches whose
// correlation exceeds twice the median are dropped.  Correlations are integers below 2^16 (121 absolute differences of bytes, twice),
// so the rank-size/2 VALUE is found by bisection on the value with a counting pass per step: no sort, one workgroup per pair.
#define STEREO_MEDIAN_LDS 32768  // kept correlations staged as 16-bit values: pairs of up to 32 k left keypoints
__global__ __launch_bounds__(1024) void k_stereo_median(StereoProblem P) {
    // The 17 counting passes of the bisection used to re-read xr / depth / corr of every keypoint from global memory (8 us per pass, 123 us per
    // launch); the kept correlations are staged ONCE in LDS (0xFFFF = not kept: correlations stay below 2 * 121 * 255 = 61 710).
    __shared__ int s_cnt[16];
    __shared__ unsigned short s_c[STEREO_MEDIAN_LDS];
    const int pair = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int nl = P.nl_dev ? min(P.nl_dev[(size_t)pair * P.n_stride], P.cap) : P.nl;
    const size_t o = P.nl_dev ? (size_t)pair * P.cap : 0;
    float* xr = P.xr + o;
    float* depth = P.depth + o;
    const float* corr = P.corr + o;
    const bool staged = nl <= STEREO_MEDIAN_LDS;
    if (staged)
        for (int i = tid; i < nl; i += nthr) s_c[i] = (xr[i] != -1.0f || depth[i] != -1.0f) ? (unsigned short)(int)corr[i] : (unsigned short)0xFFFF;
    __syncthreads();
    auto count_le = [&](int v) -> int {  // kept matches with correlation <= v (v < 0: all kept matches)
        int c = 0;
        if (staged) {
            for (int i = tid; i < nl; i += nthr) {
                const int x = s_c[i];
                c += (x != 0xFFFF && (v < 0 || x <= v)) ? 1 : 0;
            }
        }
        else {
            for (int i = tid; i < nl; i += nthr)
                if (xr[i] != -1.0f || depth[i] != -1.0f) c += (v < 0 || (int)corr[i] <= v) ? 1 : 0;
        }
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_cnt[tid >> 6] = c;
        __syncthreads();
        int t = 0;
        for (int w = 0; w < (nthr >> 6); ++w) t += s_cnt[w];
        return t;
    };
    const int kept = count_le(-1);
    if (kept == 0) return;
    const int rank = kept / 2;  // 0-based rank of the median in ascending order
    int lo = 0, hi = 1 << 17;   // smallest v with count_le(v) > rank
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (count_le(mid) > rank) hi = mid;
        else lo = mid + 1;
    }
    const float thr = (float)(2.0 * (double)(float)lo);
    // The reference walks the sorted list from the median on, so only elements at or behind the median rank are tested; an element
    // in front of it has a correlation <= the median and can never exceed twice the (non-negative) median.
    for (int i = tid; i < nl; i += nthr)
        if ((xr[i] != -1.0f || depth[i] != -1.0f) && thr < (float)(int)corr[i]) {
            xr[i] = -1.0f;
            depth[i] = -1.0f;
        }
}

}  // namespace

size_t sv_stereo_rows_bytes(int pairs, int rows, int nr_cap, int rows_per_kp) {
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    return pad((size_t)pairs * (rows + 1) * 4) + pad((size_t)pairs * rows * 4) + pad((size_t)pairs * nr_cap * rows_per_kp * 4) + 256;
}
void sv_launch_stereo(svgpu_ctx* ctx, hipStream_t s, const StereoProblem& P, int pairs) {
    SvProfScope ps(ctx, s, "k_stereo");
    const int nl = P.nl_dev ? P.cap : P.nl, nr = P.nl_dev ? P.cap : P.nr;
    if (nl <= 0 || pairs <= 0) return;
    (void)hipMemsetAsync(P.row_off, 0, (size_t)pairs * (P.rows + 1) * 4, s);
    (void)hipMemsetAsync(P.row_fill, 0, (size_t)pairs * P.rows * 4, s);
    if (nr > 0) hipLaunchKernelGGL(k_stereo_rows<false>, dim3((nr + 255) / 256, pairs), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_stereo_rows_scan, dim3(pairs), dim3(1024), 0, s, P);
    if (nr > 0) hipLaunchKernelGGL(k_stereo_rows<true>, dim3((nr + 255) / 256, pairs), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_stereo, dim3((nl + 3) / 4, pairs), dim3(256), 0, s, P);
}
void sv_launch_stereo_median(hipStream_t s, const StereoProblem& P, int pairs) {
    if (pairs > 0) hipLaunchKernelGGL(k_stereo_median, dim3(pairs), dim3(1024), 0, s, P);
}

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
    // Candidate lists.  Normal case: distances on the matrix cores (k_bf_mfma).  The VALU kernel (k_bf_topk, keeps the
    // BF_K nearest per query) serves cutoffs of 128 and more, where all-zero padding columns could pass the MFMA
    // threshold and the lists would be dense anyway (lowe_ratio < 0.4; dmax = 256 means every pair is a candidate).
    static const bool force_valu = getenv("SVGPU_BF_VALU") != nullptr;
    {
        SvProfScope ps(ctx, s, "k_bf_binsort");
        // ring mode over one set of arrays: one sorted copy per frame serves both of its roles
        P.shared_sort = P.ring1 == pairs && P.desc1 == P.desc2 && P.angle1 == P.angle2 && P.n1_dev == P.n2_dev && P.cap1 == P.cap2;
        if (P.shared_sort) {
            P.sd1 = P.sd2;
            P.sa1 = P.sa2;
            P.si1 = P.si2;
            P.bs1 = P.bs2;
        }
        hipLaunchKernelGGL(k_bf_binsort, dim3(pairs, P.shared_sort ? 1 : 2), dim3(256), 0, s, P);
    }
    {
        SvProfScope ps(ctx, s, "k_bf_topk");
        if (P.dmax >= 128u || force_valu) {
            P.list_k = BF_K;
            hipLaunchKernelGGL(k_bf_topk, dim3((P.cap2 + BF_QB - 1) / BF_QB, pairs), dim3(256), 0, s, P);
        }
        else {
            P.list_k = MF_SLOTS;
            P.mfma_tiles = sv_prof_counter(ctx, "k_bf_topk");
            hipLaunchKernelGGL(k_bf_mfma, dim3((P.cap2 + MF_QB - 1) / MF_QB, pairs), dim3(256), 0, s, P);
        }
    }
    SvProfScope ps(ctx, s, "k_bf_replay");
    size_t lds = (size_t)((P.cap1 + P.cap2 + 3) & ~3) * sizeof(int);
    const int use_lds = lds <= 96 * 1024;
    if (!use_lds) lds = 0;
    const int pool_rows = (int)std::min<size_t>(1024, (128 * 1024 - lds) / ((BF_LIST - 4) * sizeof(uint32_t)));
    lds += (size_t)pool_rows * (BF_LIST - 4) * sizeof(uint32_t);
    if (P.cap2 <= 5 * 512) {
        (void)sv_allow_dynamic_lds(reinterpret_cast<const void*>(k_bf_replay<5, 512>), 144 * 1024);
        hipLaunchKernelGGL((k_bf_replay<5, 512>), dim3(pairs), dim3(512), lds, s, P, g_owner, g_match, use_lds, pool_rows);
    }
    else {
        (void)sv_allow_dynamic_lds(reinterpret_cast<const void*>(k_bf_replay<4, 1024>), 144 * 1024);
        hipLaunchKernelGGL((k_bf_replay<4, 1024>), dim3(pairs), dim3(1024), lds, s, P, g_owner, g_match, use_lds, pool_rows);
    }
}
// exclusive scan of the grid's counters: the shuffle-based one-workgroup scan of sv_sort.hip (4.5 us) up to its 16 k elements, the
// Hillis-Steele kernel above (9 us per 1 024 elements, needs no scratch) beyond
static void grid_scan(hipStream_t s, int32_t* data, int n) {
    if (n <= 16384) sv_scan_i32(s, data, n, nullptr);
    else hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, data, n);
}
void sv_launch_grid_frame(hipStream_t s, const GridProblem& G) {  // the keypoint side: cell_of, cell_off, cell_items
    const int nc = G.cols * G.rows;
    if (nc <= GRID_ONE_CELLS && G.nt <= 1024 * GRID_ONE_KPT_ROUNDS && !std::getenv("SVGPU_GRID_FOUR_LAUNCHES")) {
        hipLaunchKernelGGL(k_grid_frame_one, dim3(1), dim3(1024), 0, s, G);
        return;
    }
    (void)hipMemsetAsync(G.cell_off, 0, (size_t)(nc + 1) * sizeof(int32_t), s);
    if (G.nt > 0) hipLaunchKernelGGL(k_grid_assign, dim3((G.nt + 255) / 256), dim3(256), 0, s, G);
    grid_scan(s, G.cell_off, nc);
    if (G.nt > 0) hipLaunchKernelGGL(k_grid_place, dim3((G.nt + 255) / 256), dim3(256), 0, s, G);
}
void sv_launch_grid_queries(hipStream_t s, const GridProblem& G) {  // the query side over a binned frame: list sizes + their scan
    if (G.nq > 0) hipLaunchKernelGGL(k_grid_walk<false>, dim3((G.nq + 3) / 4), dim3(256), 0, s, G);
    grid_scan(s, G.cand_off, G.nq);
}
void sv_launch_grid_build(hipStream_t s, const GridProblem& G) {
    sv_launch_grid_frame(s, G);
    sv_launch_grid_queries(s, G);
}
void sv_launch_grid_fill(hipStream_t s, const GridProblem& G) {
    if (G.nq > 0) hipLaunchKernelGGL(k_grid_walk<true>, dim3((G.nq + 3) / 4), dim3(256), 0, s, G);
}
void sv_launch_cand(svgpu_ctx* ctx, hipStream_t s, const CandProblem& P, int* owner, int* match, unsigned* mdist) {
    SvProfScope ps(ctx, s, "k_cand");
    if (P.nq > 0) hipLaunchKernelGGL(k_cand_dist, dim3(P.nq), dim3(64), 0, s, P);
    if (P.mode == SVGPU_MATCH_AREA) {
        const size_t lds = (size_t)(2 * P.nt + P.nq) * sizeof(int);
        const int use_lds = lds <= 60 * 1024;
        hipLaunchKernelGGL(k_area_replay, dim3(1), dim3(64), use_lds ? lds : 0, s, P, owner, match, mdist, use_lds);
        return;
    }
    sv_launch_cand_replay(ctx, s, P, owner, match);
}
void sv_launch_cand_replay(svgpu_ctx* ctx, hipStream_t s, const CandProblem& P0, int* owner, int* match) {
    CandProblem P = P0;
    static const int dbg = std::getenv("SVGPU_REPLAY_DBG") ? std::atoi(std::getenv("SVGPU_REPLAY_DBG")) : 0;
    P.dbg_phase = dbg;
    // the LDS-resident form with as many staged entries per list as fit (at most 64: longer lists are not sorted)
    const bool wc = P.cand_cnt != nullptr;
    if (cand_lds_bytes(P.nq, P.nt, 0, wc) <= CAND_LDS_BUDGET) {
        const int K = P.nq > 0 ? (int)std::min<size_t>(64, (CAND_LDS_BUDGET - cand_lds_bytes(P.nq, P.nt, 0, wc)) / ((size_t)P.nq * 4)) : 0;
        (void)sv_allow_dynamic_lds((const void*)k_cand_replay_lds, CAND_LDS_BUDGET);
        hipLaunchKernelGGL(k_cand_replay_lds, dim3(1), dim3(1024), cand_lds_bytes(P.nq, P.nt, K, wc), s, P, K);
    }
    else hipLaunchKernelGGL(k_cand_replay, dim3(1), dim3(1024), 0, s, P, owner, match);
}
