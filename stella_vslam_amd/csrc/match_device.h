// Device functions shared by match_kernels.hip and track_kernels.hip (the tracked-frame chain runs the same steps fused).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "match_kernels.h"

namespace svmd {

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

#define GRID_ONE_CELLS 4096
#define GRID_ONE_KPT_ROUNDS 8  // keypoints per thread whose arrival numbers stay in registers (8 192 keypoints)
#define GRID_ONE_SMALL 24      // a cell of at most this many keypoints is ordered by one thread (insertion sort); a larger one by the workgroup
// (body of k_grid_frame_one; `nt` = the number of keypoints, which the fused tracked-frame kernel reads from device memory)
__device__ __forceinline__ void grid_frame_one(const GridProblem& G, const int nt) {
    __shared__ int s_cnt[GRID_ONE_CELLS + 1];
    __shared__ int s_wsum[16];
    __shared__ int s_items[1024 * GRID_ONE_KPT_ROUNDS];  // the items of one crowded cell
    __shared__ int s_big[1024 * GRID_ONE_KPT_ROUNDS / GRID_ONE_SMALL + 1], s_nbig;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nc = G.cols * G.rows;
    for (int c = tid; c <= nc; c += 1024) s_cnt[c] = 0;
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    int cell[GRID_ONE_KPT_ROUNDS], pos[GRID_ONE_KPT_ROUNDS];
#pragma unroll
    for (int r = 0; r < GRID_ONE_KPT_ROUNDS; ++r) {
        const int i = tid + r * 1024;
        cell[r] = -1, pos[r] = 0;
        if (i < nt) {
            const int cx = (int)floor((double)(G.t_xy[2 * i] - G.min_x) * G.inv_w), cy = (int)floor((double)(G.t_xy[2 * i + 1] - G.min_y) * G.inv_h);
            if (0 <= cx && cx < G.cols && 0 <= cy && cy < G.rows) {
                cell[r] = cx * G.rows + cy;
                pos[r] = atomicAdd(&s_cnt[cell[r]], 1);  // arrival order inside the cell: arbitrary, put right at the end
            }
            G.cell_of[i] = cell[r];
        }
    }
    __syncthreads();
    // exclusive scan of the counters: every thread owns a contiguous run of cells, wave scan of the run sums, then the waves' totals
    const int per = (nc + 1023) / 1024, c0 = tid * per, c1 = min(c0 + per, nc);
    int run = 0;
    for (int c = c0; c < c1; ++c) run += s_cnt[c];
    int incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - run;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    __syncthreads();
    for (int c = c0; c < c1; ++c) {  // counters -> offsets, in place (a thread's own run only)
        const int k = s_cnt[c];
        s_cnt[c] = base;
        base += k;
    }
    if (tid == 1023) s_cnt[nc] = base;  // (the last thread's run ends at nc or is empty: its base is the total either way)
    __syncthreads();
    for (int c = tid; c <= nc; c += 1024) G.cell_off[c] = s_cnt[c];
#pragma unroll
    for (int r = 0; r < GRID_ONE_KPT_ROUNDS; ++r)
        if (cell[r] >= 0) G.cell_items[s_cnt[cell[r]] + pos[r]] = tid + r * 1024;
    __syncthreads();  // (workgroup-scope: the placements above are visible to the threads below)
    for (int c = tid; c < nc; c += 1024) {  // increasing keypoint index inside every cell (= the stable placement)
        const int lo = s_cnt[c], k = s_cnt[c + 1] - lo;
        if (k > GRID_ONE_SMALL) {  // crowded: left to the whole workgroup below (at most nt / GRID_ONE_SMALL such cells)
            s_big[atomicAdd(&s_nbig, 1)] = c;
            continue;
        }
        for (int a = 1; a < k; ++a) {  // insertion sort: cells hold a handful of keypoints
            const int v = G.cell_items[lo + a];
            int b = a - 1;
            while (b >= 0 && G.cell_items[lo + b] > v) {
                G.cell_items[lo + b + 1] = G.cell_items[lo + b];
                --b;
            }
            G.cell_items[lo + b + 1] = v;
        }
    }
    __syncthreads();
    // crowded cells (a dense patch under a coarse grid; in the limit every keypoint in one cell): rank sort by the workgroup -- the cell's
    // items staged in LDS, every thread counts the items below its own: k^2 / 1 024 LDS reads per thread, 16 k at the 8 192-keypoint limit
    // (one thread's insertion sort would be k^2 / 4 global round trips)
    const int nbig = s_nbig;
    for (int bi = 0; bi < nbig; ++bi) {
        const int c = s_big[bi], lo = s_cnt[c], k = s_cnt[c + 1] - lo;
        for (int a = tid; a < k; a += 1024) s_items[a] = G.cell_items[lo + a];
        __syncthreads();
        for (int a = tid; a < k; a += 1024) {
            const int v = s_items[a];
            int rank = 0;
            for (int j = 0; j < k; ++j) rank += s_items[j] < v;  // (indices are distinct)
            G.cell_items[lo + rank] = v;
        }
        __syncthreads();
    }
}

}  // namespace svmd
