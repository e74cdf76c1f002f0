// HIP kernels of the ORB front end for gfx950 (wave64).  One kernel per reference routine:
//   k_resize    <- cv::resize INTER_LINEAR chain       (feature/orb_extractor.cc:153-162)
//   k_blur      <- cv::GaussianBlur 7x7 sigma 2        (feature/orb_extractor.cc:103)
//   k_fast      <- per-cell cv::FAST + NMS + retry + selection-grid arg-max
//                                                      (feature/orb_extractor.cc:164-287, 289-329)
//   k_select    <- emission of the selected keypoints in grid-cell order (:314-326)
//   k_describe  <- ic_angle + compute_orb_descriptor + correct_keypoint_scale
//                                                      (feature/orb_impl.cc:68-154, orb_extractor.cc:337-345)
// All integer stages are bit-exact by construction; the fp32 stages replicate the reference's
// operation order with contraction disabled (-ffp-contract=off, checked in the disassembly).
#include "svgpu_internal.h"
#include <cstdlib>
#include <utility>

#pragma clang fp contract(off)

namespace {

// the rBRIEF pattern (small integers) stored as floats: k_describe keeps its pairs as floats, so one 16-byte load per pair replaces a byte
// unpack + 4 converts per wave
__device__ __constant__ float c_pattern_f[1024] = {
#include "orb_pattern_i8.inc"
};

// Which level does work item `idx` belong to?  `first` = per-level first index (non-decreasing);
// linear scan over <= 16 levels, block-uniform.
__device__ __forceinline__ int find_level(const OrbLevel* __restrict__ L, int num_levels, int idx, int OrbLevel::*first,
                                          int* local) {
    int lv = 0;
    for (int l = 0; l < num_levels; ++l) {
        const int f = L[l].*first;
        if (idx >= f) lv = l;
    }
    *local = idx - L[lv].*first;
    return lv;
}

__device__ __forceinline__ int reflect101(int p, int len) {  // cv::BORDER_REFLECT_101
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// XCD-aware work mapping.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and every XCD has its own L2:
// with the plain (x = work item, y = frame) grid each XCD touches every frame and the same image lines are fetched
// from HBM up to eight times.  This remaps the linear id so that all workgroups of a frame run on one XCD (frame f ->
// XCD f % 8); batches that are not a multiple of 8 keep the plain order for the remainder frames.
__device__ __forceinline__ void xcd_frame_map(int per_frame, int batch, int& item, int& frame) {
    const int lin = blockIdx.x + blockIdx.y * gridDim.x;
    const int full = (batch / 8) * 8, lim = full * per_frame;
    if (lin < lim) {
        const int xcd = lin & 7, idx = lin >> 3;
        frame = (idx / per_frame) * 8 + xcd;
        item = idx % per_frame;
    }
    else {
        const int r = lin - lim;
        frame = full + r / per_frame;
        item = r % per_frame;
    }
}

// ------------------------------------------------------------------------------------------------ resize
// OpenCV 8-bit bilinear: horizontal int32 with 11-bit coefficients, vertical
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.  Coefficient tables are built on the host.
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                int sw, uint8_t* __restrict__ dst, size_t dst_frame_stride, int dst_pitch,
                                                int dw, int dh, const short* __restrict__ xofs,
                                                const short2* __restrict__ xa, const short2* __restrict__ yofs,
                                                const short2* __restrict__ yb) {
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= dh || dx0 >= dw) return;
    const uint8_t* S = src + (size_t)blockIdx.z * src_frame_stride;
    const short2 yo = yofs[dy];
    const short2 bb = yb[dy];
    const uint8_t* S0 = S + (size_t)yo.x * src_pitch;
    const uint8_t* S1 = S + (size_t)yo.y * src_pitch;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = dx0 + i;
        if (dx < dw) {
            const int sx = xofs[dx];
            const short2 a = xa[dx];
            const int sx1 = min(sx + 1, sw - 1);
            const int r0 = S0[sx] * a.x + S0[sx1] * a.y;
            const int r1 = S1[sx] * a.x + S1[sx1] * a.y;
            const int v = ((((int)bb.x * (r0 >> 4)) >> 16) + (((int)bb.y * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    uint8_t* D = dst + (size_t)blockIdx.z * dst_frame_stride + (size_t)dy * dst_pitch + dx0;
    *reinterpret_cast<uint32_t*>(D) = packed;  // pitch is a multiple of 64: in-bounds and aligned
}

// Whole pyramid in ONE launch.  A workgroup owns a horizontal band of one frame through all levels: for level
// l = 1..L-1 it computes the rows of that level it owns plus the few halo rows its own next level reads
// (rows[band][l] = [lo, hi), built on the host), with a workgroup barrier between levels.  Halo rows are
// recomputed by neighbouring bands with identical arithmetic, so the duplicate stores write identical bytes.
// This removes the six dependent kernel boundaries of the chained cv::resize calls.
#define PYR_THREADS 1024
#define PYR_MAXW 4096  // widest level whose column tables are staged in LDS
__global__ __launch_bounds__(PYR_THREADS) void k_pyramid(const OrbLevel* __restrict__ L, int num_levels, const int2* __restrict__ band_rows,
                                                         const uint8_t* __restrict__ img0, size_t img0_frame_stride, int img0_pitch,
                                                         uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                         const short* __restrict__ xofs, const short2* __restrict__ xa,
                                                         const short2* __restrict__ yofs, const short2* __restrict__ yb) {
    __shared__ short s_xo[PYR_MAXW];
    __shared__ short2 s_xa[PYR_MAXW];
    const int band = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    uint8_t* P = pyr + (size_t)b * pyr_frame_bytes;
    for (int l = 1; l < num_levels; ++l) {
        const OrbLevel lev = L[l], prev = L[l - 1];
        const uint8_t* S = l == 1 ? img0 + (size_t)b * img0_frame_stride : P + prev.pyr_off;
        const int sp = l == 1 ? img0_pitch : prev.pitch;
        uint8_t* Dst = P + lev.pyr_off;
        const int2 rr = band_rows[band * num_levels + l];
        const int groups = (lev.w + 3) >> 2, ntask = (rr.y - rr.x) * groups;
        // column tables of this level -> LDS (one coalesced pass), so that the per-pixel chain is a single global round trip
        const bool in_lds = lev.w <= PYR_MAXW;
        if (in_lds)
            for (int x = tid; x < lev.w; x += PYR_THREADS) {
                s_xo[x] = xofs[lev.xtab_off + x];
                s_xa[x] = xa[lev.xtab_off + x];
            }
        __syncthreads();
        for (int t = tid; t < ntask; t += PYR_THREADS) {
            const int row = t / groups, g = t - row * groups;
            const int dy = rr.x + row, dx0 = g * 4;
            const short2 yo = yofs[lev.ytab_off + dy], bb = yb[lev.ytab_off + dy];
            const uint8_t* S0 = S + (size_t)yo.x * sp;
            const uint8_t* S1 = S + (size_t)yo.y * sp;
            int sx[4], sx1[4];
            short2 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dx = min(dx0 + i, lev.w - 1);
                sx[i] = in_lds ? s_xo[dx] : xofs[lev.xtab_off + dx];
                a[i] = in_lds ? s_xa[dx] : xa[lev.xtab_off + dx];
                sx1[i] = min(sx[i] + 1, prev.w - 1);
            }
            int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // 16 independent byte loads
                p00[i] = S0[sx[i]];
                p01[i] = S0[sx1[i]];
                p10[i] = S1[sx[i]];
                p11[i] = S1[sx1[i]];
            }
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r0 = p00[i] * a[i].x + p01[i] * a[i].y;
                const int r1 = p10[i] * a[i].x + p11[i] * a[i].y;
                const int v = ((((int)bb.x * (r0 >> 4)) >> 16) + (((int)bb.y * (r1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 255) << (8 * i);
            }
            *reinterpret_cast<uint32_t*>(Dst + (size_t)dy * lev.pitch + dx0) = packed;
        }
        __syncthreads();  // level l of this band (and its halo) is complete and visible to the whole workgroup
    }
}

// LDS-resident variant (the one normally launched).  A workgroup owns a horizontal band of one frame through ALL levels, entirely in LDS:
//   * the band's level-0 rows (with the halo its level-1 rows read) are staged into LDS with wide coalesced loads;
//   * level l is computed from level l - 1 in LDS and stored to global memory once, by the band that owns the row;
//   * a thread owns ONE column group (4 adjacent output pixels) and WALKS a chunk of consecutive output rows downwards.  cv::resize's
//     bilinear kernel is separable -- H(s) = S[s][sx] * a0 + S[s][sx + 1] * a1 per source row s, then the vertical blend of H(sy) and
//     H(sy + 1) -- and consecutive output rows share a source row five times out of six at the 1 / 1.2 step, so the walk keeps H(sy + 1)
//     in registers and forms ~1.3 instead of 2 horizontal rows per output row; the column record (window positions, byte selectors,
//     coefficient pairs) is loop-invariant and lives in registers, read once per level straight from global memory.
//   Horizontal: two 8-byte windows per source row (ds_read2_b32), one v_perm_b32 per pixel lifts (S[sx], S[sx + 1]) into two 16-bit
//   halves, one v_dot2_u32_u16 forms H.  Vertical: ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) as two v_mul_hi_u32_u24 of
//   (b << 12) and (H & ~15): (b * 2^12) * ((H >> 4) * 2^4) >> 32 == (b * (H >> 4)) >> 16, both factors below 2^24.
// LDS map (dynamic): [rows of the odd levels (one region, reused)][rows of the even levels, level 0 included (one region, reused)], pitch = w
// rounded up to 4, [8-byte row records of the band's rows].  The host picks the band count so that two workgroups fit a CU (svgpu_orb_configure).
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b) {  // (a * b) >> 32 for a, b < 2^24: full-rate v_mul_hi_u32_u24
    uint32_t d;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__global__ __launch_bounds__(PYR_THREADS) void k_pyramid_lds(const OrbLevel* __restrict__ L, int num_levels, const int2* __restrict__ band_rows,
                                                             int bands, const uint8_t* __restrict__ img0, size_t img0_frame_stride,
                                                             int img0_pitch, uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                             const uint32_t* __restrict__ xg, const short4* __restrict__ yrow) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    __shared__ int s_img[SV_MAX_LEVELS], s_yt[SV_MAX_LEVELS + 1];
    const int band = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int2* BR = band_rows + (size_t)band * num_levels;
    if (tid == 0) {
        // level l is computed from level l - 1 only: two image regions alternate (odd levels in the first, even levels in the second),
        // each as large as its largest tenant
        int size_a = 0, size_b = 0;
        for (int l = 0; l < num_levels; ++l) {
            const int bytes = (BR[l].y - BR[l].x) * ((L[l].w + 3) & ~3);
            if (l & 1) size_a = max(size_a, bytes);
            else size_b = max(size_b, bytes);
        }
        size_a = (size_a + 15) & ~15;
        size_b = (size_b + 15) & ~15;
        for (int l = 0; l < num_levels; ++l) s_img[l] = (l & 1) ? 0 : size_a;
        int off = size_a + size_b;
        s_yt[0] = off;
        for (int l = 1; l < num_levels; ++l) {
            s_yt[l] = off;
            off += (BR[l].y - BR[l].x) * 8;
        }
        s_yt[num_levels] = off;
    }
    __syncthreads();
    const uint8_t* I0 = img0 + (size_t)b * img0_frame_stride;
    {
        // ---- level-0 rows of the band -> LDS (pitch = w rounded up to 4), row records of the band's rows
        const int w0 = L[0].w, p0 = (w0 + 3) & ~3, lo0 = BR[0].x, n0 = BR[0].y - lo0;
        uint8_t* D0 = s_mem + s_img[0];
        if (((((size_t)I0) | (size_t)img0_pitch) & 15) == 0 && (p0 & 15) == 0) {
            const int cpr = p0 >> 4, total = n0 * cpr;  // 16-byte chunks (the row pitch covers the rounded-up width)
            const float inv = 1.0f / (float)cpr;
            for (int i = tid; i < total; i += PYR_THREADS) {
                const int r = (int)(((float)i + 0.5f) * inv), c = i - __mul24(r, cpr);
                reinterpret_cast<uint4*>(D0)[i] = *reinterpret_cast<const uint4*>(I0 + (__umul24(lo0 + r, img0_pitch) + 16 * c));
            }
        }
        else if (((((size_t)I0) | (size_t)img0_pitch) & 3) == 0) {
            const int wpr = p0 >> 2, total = n0 * wpr;
            const float inv = 1.0f / (float)wpr;
            for (int i = tid; i < total; i += PYR_THREADS) {
                const int r = (int)(((float)i + 0.5f) * inv), c = i - __mul24(r, wpr);
                reinterpret_cast<uint32_t*>(D0)[i] = *reinterpret_cast<const uint32_t*>(I0 + (__umul24(lo0 + r, img0_pitch) + 4 * c));
            }
        }
        else {
            const int total = n0 * p0;
            for (int i = tid; i < total; i += PYR_THREADS) {
                const int r = i / p0, c = i - r * p0;
                D0[i] = c < w0 ? I0[(size_t)(lo0 + r) * img0_pitch + c] : (uint8_t)0;
            }
        }
        const int rows = (s_yt[num_levels] - s_yt[0]) >> 3;
        for (int i = tid; i < rows; i += PYR_THREADS) {
            int l = 1;
            while (l + 1 < num_levels && s_yt[0] + 8 * i >= s_yt[l + 1]) ++l;
            const int r = i - ((s_yt[l] - s_yt[0]) >> 3);
            reinterpret_cast<short4*>(s_mem + s_yt[0])[i] = yrow[L[l].ytab_off + BR[l].x + r];
        }
    }
    __syncthreads();
    uint8_t* P = pyr + (size_t)b * pyr_frame_bytes;
    for (int l = 1; l < num_levels; ++l) {
        const OrbLevel lev = L[l];
        const int spw = ((L[l - 1].w + 3) & ~3) >> 2, dpw = ((lev.w + 3) & ~3) >> 2;  // LDS pitches in dwords
        const int src_lo = BR[l - 1].x, lo = BR[l].x, hi = BR[l].y;
        const int own_lo = (int)((long long)band * lev.h / bands), own_hi = (int)((long long)(band + 1) * lev.h / bands);
        const uint32_t* S = reinterpret_cast<const uint32_t*>(s_mem + s_img[l - 1]);
        uint32_t* Dl = reinterpret_cast<uint32_t*>(s_mem + s_img[l]);
        const short4* yt = reinterpret_cast<const short4*>(s_mem + s_yt[l]);
        uint8_t* Dg = P + lev.pyr_off;
        // thread -> (chunk of consecutive rows, column group)
        const int groups = (lev.w + 3) >> 2, nrows = hi - lo;
        const int chunks = max(min(PYR_THREADS / groups, nrows), 1), rc = (nrows + chunks - 1) / chunks;
        const int c = (int)(((float)tid + 0.5f) * (1.0f / (float)groups)), g = tid - __mul24(c, groups);
        const int r0 = __mul24(c, rc), r1 = min(r0 + rc, nrows);
        if (r0 < r1) {
            const uint4 e = reinterpret_cast<const uint4*>(xg)[2 * (lev.xg_off + g)];
            const uint2 e2 = reinterpret_cast<const uint2*>(xg)[4 * (lev.xg_off + g) + 2];
            const int i0 = e.x & 0xffff, i2 = e.x >> 16;
            const uint32_t coef[4] = {e.z, e.w, e2.x, e2.y};
            uint32_t sel[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) sel[i] = 0x0c010c00u + ((e.y >> (8 * i)) & 255u) * 0x00010001u;  // bytes (k, zero, k + 1, zero) of the 8-byte window
            // row sy of the source level for this thread's first window: S0 + sy * spw (the second window d2 dwords further)
            const uint32_t* S0 = S + (i0 - __mul24(src_lo, spw));
            const int d2 = i2 - i0;
            auto hrow = [&](int sy, uint32_t (&h)[4]) {
                const uint32_t* R = S0 + __mul24(sy, spw);
                const uint32_t a0 = R[0], a1 = R[1], c0 = R[d2], c1 = R[d2 + 1];
                h[0] = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(a1, a0, sel[0])), as_u16x2(coef[0]), 0u, false) & ~15u;
                h[1] = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(a1, a0, sel[1])), as_u16x2(coef[1]), 0u, false) & ~15u;
                h[2] = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(c1, c0, sel[2])), as_u16x2(coef[2]), 0u, false) & ~15u;
                h[3] = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(c1, c0, sel[3])), as_u16x2(coef[3]), 0u, false) & ~15u;
            };
            uint32_t* dl = Dl + (__mul24(r0, dpw) + g);
            uint32_t goff = __umul24(lo + r0, lev.pitch) + 4 * g;  // byte offset of (row, group) in the level's global image
            const uint32_t own_n = (uint32_t)(own_hi - own_lo);
            uint32_t dyo = (uint32_t)(lo + r0 - own_lo);            // row - own_lo: owned <=> dyo < own_n (unsigned)
            int prev = -1;                                           // the source row the bottom registers of the previous step hold
            // one output row: `top` already holds H(prev) -- reused when the row's first tap is that source row --, the second tap goes to `bot`;
            // the two register sets swap roles from row to row (the loop is unrolled by two), so the reuse costs no moves
            auto step = [&](int row, uint32_t (&top)[4], uint32_t (&bot)[4]) {
                const short4 ye = yt[row];
                if (ye.x != prev) hrow(ye.x, top);
                hrow(ye.y, bot);
                prev = ye.y;
                const uint32_t b0 = (uint32_t)ye.z << 12, b1 = (uint32_t)ye.w << 12;  // 0 .. 2048 each
                uint32_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (mul_hi_u24(b0, top[i]) + mul_hi_u24(b1, bot[i]) + 2u) >> 2;
                const uint32_t packed = (v[0] & 255u) | ((v[1] & 255u) << 8) | ((v[2] & 255u) << 16) | (v[3] << 24);
                *dl = packed;
                if (dyo < own_n) *reinterpret_cast<uint32_t*>(Dg + goff) = packed;
                dl += dpw;
                goff += (uint32_t)lev.pitch;
                ++dyo;
            };
            uint32_t hX[4], hY[4] = {0, 0, 0, 0};
            int row = r0;
            for (; row + 1 < r1; row += 2) {
                step(row, hY, hX);
                step(row + 1, hX, hY);
            }
            if (row < r1) step(row, hY, hX);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ blur
// Fixed-point separable 7x7, taps {18,34,48,56,48,34,18}/256, out = (sum + 32768) >> 16, reflect-101.
// Streaming form, no LDS: a thread owns 4 adjacent columns and walks BLUR_ROWS rows downwards, keeping the last
// seven rows of horizontal sums in registers.  A wave reads/writes 256 contiguous bytes per row.
#define BLUR_EDGE_ROWS 8            // rows per thread in the (slow, gather-based) edge tiles
struct BlurRow {  // the 12 source bytes of pixels [x0-4, x0+8) of one row, image borders already reflected
    uint32_t w0, w1, w2;
};
// How a thread fetches its rows.  BLUR_INTERIOR: three aligned words.  BLUR_EDGE: the column group touches the left or
// right image border of an aligned level; the reflected bytes are picked out of the neighbouring aligned words with
// v_perm_b32 selectors computed once per thread.  BLUR_GATHER: byte gather (caller images with an odd pitch / base,
// levels narrower than 16 px).
enum { BLUR_INTERIOR = 0, BLUR_EDGE = 1, BLUR_GATHER = 2 };
struct BlurEdge {
    bool left, hi_p1;
    uint32_t sel1, sel2;
};
__device__ __forceinline__ uint32_t blur_selector(int xbase, int lo_base, int w) {
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = xbase + k, sx = x < w ? x : 2 * (w - 1) - x;
        s |= (uint32_t)min(max(sx - lo_base, 0), 7) << (8 * k);  // clamped entries belong to output columns >= w (row padding)
    }
    return s;
}
__device__ __forceinline__ BlurEdge blur_edge_ctx(int x0, int w) {
    BlurEdge e;
    e.left = x0 == 0;
    e.hi_p1 = x0 + 4 < w;
    e.sel1 = blur_selector(x0, x0 - 4, w);
    e.sel2 = blur_selector(x0 + 4, e.hi_p1 ? x0 : x0 - 4, w);
    return e;
}
template <int MODE>
__device__ __forceinline__ BlurRow blur_load(const uint8_t* __restrict__ row, int x0, int w, const BlurEdge& e) {
    BlurRow r;
    if (MODE == BLUR_INTERIOR) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(row + x0 - 4);
        r.w0 = p[0];
        r.w1 = p[1];
        r.w2 = p[2];
    }
    else if (MODE == BLUR_EDGE) {
        if (e.left) {  // pixels -4..-1 are pixels 4,3,2,1
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row);
            r.w1 = p[0];
            r.w2 = p[1];
            r.w0 = __builtin_amdgcn_perm(r.w2, r.w1, 0x01020304u);
        }
        else {  // pixels >= w are pixels 2(w-1)-x, all inside the words at x0-4, x0 (and x0+4 when that word starts inside the row)
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row + x0 - 4);
            const uint32_t pm1 = p[0], p0 = p[1], p1 = e.hi_p1 ? p[2] : 0u;
            r.w0 = pm1;
            r.w1 = __builtin_amdgcn_perm(p0, pm1, e.sel1);
            r.w2 = e.hi_p1 ? __builtin_amdgcn_perm(p1, p0, e.sel2) : __builtin_amdgcn_perm(p0, pm1, e.sel2);
        }
    }
    else {
        uint32_t b[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = row[reflect101(x0 - 4 + k, w)];
        r.w0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        r.w1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        r.w2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    }
    return r;
}
// Horizontal 7-tap sums of the 4 owned columns for TWO image rows at once: pixel k of row a and of row b travel as the
// two 16-bit halves of one register (v_perm_b32 gathers them), so every packed multiply-add serves both rows.
// 256 * 255 = 65280 fits 16 bits exactly.
__device__ __forceinline__ void blur_hpair(const BlurRow& a, const BlurRow& b, u16x2 (&hp)[4]) {
    u16x2 q[10];  // pixels x0-3 .. x0+6
#define SV_PK(WA, WB, K) as_u16x2(__builtin_amdgcn_perm(WB, WA, 0x0c000c00u | ((4u + K) << 16) | K))
    q[0] = SV_PK(a.w0, b.w0, 1u);
    q[1] = SV_PK(a.w0, b.w0, 2u);
    q[2] = SV_PK(a.w0, b.w0, 3u);
    q[3] = SV_PK(a.w1, b.w1, 0u);
    q[4] = SV_PK(a.w1, b.w1, 1u);
    q[5] = SV_PK(a.w1, b.w1, 2u);
    q[6] = SV_PK(a.w1, b.w1, 3u);
    q[7] = SV_PK(a.w2, b.w2, 0u);
    q[8] = SV_PK(a.w2, b.w2, 1u);
    q[9] = SV_PK(a.w2, b.w2, 2u);
#undef SV_PK
    const u16x2 t18 = {18, 18}, t34 = {34, 34}, t48 = {48, 48}, t56 = {56, 56};
#pragma unroll
    for (int j = 0; j < 4; ++j) hp[j] = t18 * (q[j] + q[j + 6]) + t34 * (q[j + 1] + q[j + 5]) + t48 * (q[j + 2] + q[j + 4]) + t56 * q[j + 3];
}
__device__ __forceinline__ uint32_t blur_dot(u16x2 h, uint32_t taps, uint32_t acc) { return __builtin_amdgcn_udot2(h, as_u16x2(taps), acc, false); }
__device__ __forceinline__ uint32_t blur_pack(const uint32_t (&acc)[4]) {  // byte 2 of each accumulator = (acc >> 16) & 255
    const uint32_t lo = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
    return lo | hi;
}

// one thread: columns x0..x0+3, rows ys..ye-1 (ys even).  Rows are handled in pairs (ys-4, ys-3), (ys-2, ys-1), ...; four pairs
// of horizontal sums stay in registers and every iteration adds one pair and emits two output rows, each as four
// 2-element dot products (v_dot2_u32_u16) against the vertically paired taps.
template <int MODE>
__device__ __forceinline__ void blur_strip(const uint8_t* __restrict__ src, int spitch, int w, int h, uint8_t* __restrict__ dst,
                                           int dpitch, int x0, int ys, int ye) {
    auto row_ptr = [&](int y) { return src + __umul24(reflect101(y, h), spitch); };  // < 2^24 each: full-rate multiply, 32-bit offset
    BlurEdge edge = {};
    if (MODE == BLUR_EDGE) edge = blur_edge_ctx(x0, w);
    auto load_pair = [&](int y, BlurRow& a, BlurRow& b) {
        a = blur_load<MODE>(row_ptr(y), x0, w, edge);
        b = blur_load<MODE>(row_ptr(y + 1), x0, w, edge);
    };
    u16x2 A[4], B[4], C[4], D[4], E[4];
    BlurRow n0, n1;
    {
        // the ten rows the first output pair needs: all loads issued before the first use (the empty asm ties every loaded word
        // to one point, so the compiler cannot wait for one row pair before requesting the next)
        BlurRow r[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load_pair(ys - 4 + 2 * k, r[2 * k], r[2 * k + 1]);
        load_pair(ys + 4, n0, n1);
        if (MODE == BLUR_INTERIOR) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(r[k].w0), "+v"(r[k].w1), "+v"(r[k].w2));
            asm volatile("" : "+v"(n0.w0), "+v"(n0.w1), "+v"(n0.w2), "+v"(n1.w0), "+v"(n1.w1), "+v"(n1.w2));
        }
        blur_hpair(r[0], r[1], A);
        blur_hpair(r[2], r[3], B);
        blur_hpair(r[4], r[5], C);
        blur_hpair(r[6], r[7], D);
    }
    uint8_t* Dp = dst + x0;
    for (int y = ys; y < ye; y += 2) {
        const BlurRow c0 = n0, c1 = n1;
        load_pair(y + 6, n0, n1);  // one pair of rows of loads stays in flight ahead of the arithmetic
        blur_hpair(c0, c1, E);
        uint32_t acc[4];
        // row y: taps 18 34 48 56 48 34 18 over rows y-3 .. y+3 = A.hi | B | C | D
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = blur_dot(D[j], 34u | (18u << 16), blur_dot(C[j], 56u | (48u << 16), blur_dot(B[j], 34u | (48u << 16), blur_dot(A[j], 18u << 16, 32768u))));
        *reinterpret_cast<uint32_t*>(Dp + __umul24(y, dpitch)) = blur_pack(acc);
        if (y + 1 < ye) {  // row y+1: rows y-2 .. y+4 = B | C | D | E.lo
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = blur_dot(E[j], 18u, blur_dot(D[j], 48u | (34u << 16), blur_dot(C[j], 48u | (56u << 16), blur_dot(B[j], 18u | (34u << 16), 32768u))));
            *reinterpret_cast<uint32_t*>(Dp + __umul24(y + 1, dpitch)) = blur_pack(acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A[j] = B[j];
            B[j] = C[j];
            C[j] = D[j];
            D[j] = E[j];
        }
    }
}

// Tiles [0, btiles_x*btiles_y) of a level are interior tiles (aligned 12-byte windows, branch-free); the remaining tiles
// of the level are EDGE tiles whose threads do the column groups that touch the left/right image border.  Keeping the
// two roles in different workgroups keeps every wave on a single code path.  Sources that are not 4-byte aligned
// (caller images) or narrower than 16 px go through k_blur_gather instead (host decision, sv_launch_blur).
__device__ __forceinline__ void blur_locate(const OrbLevel* __restrict__ L, int num_levels, const uint8_t* __restrict__ img0,
                                            size_t img0_frame_stride, int img0_pitch, const uint8_t* __restrict__ pyr,
                                            size_t pyr_frame_bytes, uint8_t* __restrict__ blur, size_t blur_frame_bytes, OrbLevel& lev,
                                            int& tile, const uint8_t*& src, int& spitch, uint8_t*& dst) {
    int b, item;
    xcd_frame_map(gridDim.x, gridDim.y, item, b);
    const int lv = find_level(L, num_levels, item, &OrbLevel::btile_first, &tile);
    lev = L[lv];
    if (lv == 0) {
        src = img0 + (size_t)b * img0_frame_stride;
        spitch = img0_pitch;
    }
    else {
        src = pyr + (size_t)b * pyr_frame_bytes + lev.pyr_off;
        spitch = lev.pitch;
    }
    dst = blur + (size_t)b * blur_frame_bytes + lev.blur_off;
}
__device__ __forceinline__ bool blur_streamable(const uint8_t* src, int spitch, int w) {
    return ((((size_t)src) | (size_t)spitch) & 3) == 0 && w >= 16;
}
template <int ROWS>  // rows a thread walks: 64 for batches (least halo traffic), 16 for a context configured for a few frames (4 x the threads:
                     // one 640x480 frame is 1 280 strips of 64 rows, a serial walk of 25 us on a chip that holds 500 k threads)
__global__ __launch_bounds__(256) void k_blur(const OrbLevel* __restrict__ L, int num_levels, const uint8_t* __restrict__ img0,
                                              size_t img0_frame_stride, int img0_pitch, const uint8_t* __restrict__ pyr,
                                              size_t pyr_frame_bytes, uint8_t* __restrict__ blur, size_t blur_frame_bytes) {
    OrbLevel lev;
    int tile, spitch;
    const uint8_t* src;
    uint8_t* dst;
    blur_locate(L, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes, lev, tile, src, spitch, dst);
    if (!blur_streamable(src, spitch, lev.w)) return;  // k_blur_gather's level
    const int main_tiles = lev.btiles_x * lev.btiles_y;
    if (tile < main_tiles) {
        const int tx = tile % lev.btiles_x, ty = tile / lev.btiles_x;
        // The last column tile of a level is rarely full (640 px: 32 of its 64 column groups, 533 px: 6, 370 px: 29, 309 px: 14): with one strip
        // per wave the level widths fill 77 % of the lanes of this kernel, which is bound by instruction issue.  When the tile's interior groups fit
        // `g` <= 32 lanes, a wave takes 64 / g STRIPS at once (lane = strip_in_wave * g + group): the strip's first row is then per lane and
        // blur_strip's row pointers become vector registers (a few more instructions in these waves, which do 2-8 x the work) -- 95 % of the lanes.
        const int rem_groups = tx == lev.btiles_x - 1 ? (min(lev.w - 6, (tx + 1) * BLUR_TW) - tx * BLUR_TW + 3) / 4 : 64;  // groups x0 with x0 + 6 < w
        int g = 64;
        while (g > 1 && (g >> 1) >= rem_groups) g >>= 1;
#ifdef BLUR_NO_MULTISTRIP
        g = 64;
#endif
        if (g <= 32 && rem_groups > 0) {
            const int spw = 64 / g, lane = threadIdx.x & 63;
            if (ty * spw >= lev.btiles_y) return;  // (this tile's strips are carried by an earlier tile's waves)
            const int strip = (ty * 4 + (int)(threadIdx.x >> 6)) * spw + lane / g;
            const int x0 = tx * BLUR_TW + (lane % g) * 4, ys = strip * ROWS;
            if (x0 >= lev.w || ys >= lev.h) return;
            if (x0 >= 4 && x0 + 6 < lev.w) blur_strip<BLUR_INTERIOR>(src, spitch, lev.w, lev.h, dst, lev.pitch, x0, ys, min(ys + ROWS, lev.h));
            return;
        }
        const int x0 = tx * BLUR_TW + (threadIdx.x & 63) * 4;
        // the strip's first row is the same for the whole wave: say so, and the row pointers of blur_strip become scalar
        const int ys = __builtin_amdgcn_readfirstlane(ty * (4 * ROWS) + (int)(threadIdx.x >> 6) * ROWS);
        if (x0 >= lev.w || ys >= lev.h) return;
        if (x0 >= 4 && x0 + 6 < lev.w) blur_strip<BLUR_INTERIOR>(src, spitch, lev.w, lev.h, dst, lev.pitch, x0, ys, min(ys + ROWS, lev.h));
    }
    else {
        // edge groups: x0 = 0 and every group with x0 + 6 >= w (at most two); thread = (strip, which)
        const int which = threadIdx.x & 3, strip = (tile - main_tiles) * 64 + (threadIdx.x >> 2);
        const int ys = strip * BLUR_EDGE_ROWS;
        const int last = ((lev.w - 1) >> 2) << 2;
        int x0;
        if (which == 0) x0 = 0;
        else if (which == 1) x0 = last;
        else if (which == 2) x0 = last - 4;
        else return;
        if (ys >= lev.h || x0 < 0 || (which != 0 && (x0 == 0 || x0 + 6 < lev.w))) return;
        blur_strip<BLUR_EDGE>(src, spitch, lev.w, lev.h, dst, lev.pitch, x0, ys, min(ys + BLUR_EDGE_ROWS, lev.h));
    }
}
template <int ROWS>
__global__ __launch_bounds__(256) void k_blur_gather(const OrbLevel* __restrict__ L, int num_levels, const uint8_t* __restrict__ img0,
                                                     size_t img0_frame_stride, int img0_pitch, const uint8_t* __restrict__ pyr,
                                                     size_t pyr_frame_bytes, uint8_t* __restrict__ blur, size_t blur_frame_bytes) {
    OrbLevel lev;
    int tile, spitch;
    const uint8_t* src;
    uint8_t* dst;
    blur_locate(L, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes, lev, tile, src, spitch, dst);
    if (blur_streamable(src, spitch, lev.w) || tile >= lev.btiles_x * lev.btiles_y) return;
    const int x0 = (tile % lev.btiles_x) * BLUR_TW + (threadIdx.x & 63) * 4;
    const int ys = (tile / lev.btiles_x) * (4 * ROWS) + (threadIdx.x >> 6) * ROWS;
    if (x0 >= lev.w || ys >= lev.h) return;
    blur_strip<BLUR_GATHER>(src, spitch, lev.w, lev.h, dst, lev.pitch, x0, ys, min(ys + ROWS, lev.h));
}

// ------------------------------------------------------------------------------------------------ FAST
// Arc score A(p) = max over the 16 arcs of 9 contiguous ring pixels of min(|v - p_k|) with consistent sign.
// cv::FAST: p is a corner at threshold t  <=>  A(p) > t, and cornerScore = A(p) - 1
// (closed form of fast.cpp / fast_score.cpp; proven equal to the literal row-buffer algorithm by
// tests/test_oracle_orb.py::test_fast_matches_closed_form_definition).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int arc_score16(int v, const int (&p)[16]) {
    // A(v) = max(0, max over the 16 arcs of 9 of min_k (v - p_k), max over arcs of min_k (p_k - v)).
    // min / max commute with adding a constant, so the centre value stays out of the packed part: every ring pixel becomes the pair
    // E_k = (p_k, ~p_k) = (p_k, -p_k - 1) in the two int16 halves of one register -- a single 24-bit multiply-add from the byte --
    // and one v_pk_min_i16 per window serves the "centre darker" (min p) and the "centre brighter" (-max p - 1) arcs at once:
    //   brighter arcs: v - max p = v + 1 + hi,   darker arcs: min p - v = lo - v.
    // The 16 arcs of 9 are the 8 windows of 8 starting at odd k, each extended by its left or its right neighbour.
    s16x2 e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = __builtin_bit_cast(s16x2, __mul24(p[k], -65535) + (int)0xFFFF0000);  // (0xFFFF - p) << 16 | p
    s16x2 m2[8], m4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m2[j] = __builtin_elementwise_min(e[2 * j + 1], e[(2 * j + 2) & 15]);  // window 2 at k = 2j+1
#pragma unroll
    for (int j = 0; j < 8; ++j) m4[j] = __builtin_elementwise_min(m2[j], m2[(j + 1) & 7]);             // window 4 at k = 2j+1
    s16x2 best = s16x2{(short)-32768, (short)-32768};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const s16x2 m8 = __builtin_elementwise_min(m4[j], m4[(j + 2) & 7]);                            // window 8 at k = 2j+1
        // arcs 2j .. 2j+8 and 2j+1 .. 2j+9: max(min(m8, a), min(m8, b)) = min(m8, max(a, b))
        best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8, __builtin_elementwise_max(e[2 * j], e[(2 * j + 9) & 15])));
    }
    return max(max((int)best.x - v, v + 1 + (int)best.y), 0);
}

// one LDS-DMA instruction: lane i copies the 16 bytes at base + voff[i] (any byte address: profiles/r06_ubench_glds.json) to LDS byte
// lds_dst + 16 i.  M0 carries the LDS address and is written in the statement that reads it (the compiler does not preserve it); the two
// moves + s_nop 2 are also the five wait states a VMEM instruction needs behind a VALU instruction that produced its scalar base
// (hipcc does not look for hazards inside an asm statement).
__device__ __forceinline__ void sv_glds16(unsigned long long base, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}

#define FAST_KT 12  // selection-grid cells per dimension cached in LDS (a 70-px ROI spans at most ~10 at the coarsest level)
#define FP 80  // LDS pitch of the ROI arrays: 3 skew bytes + 70, rounded up to whole 16-byte chunks
// One workgroup works through `cpw` consecutive cells of one frame (persistent over cells: the frame / lane-role set-up, and the row
// addresses, band masks and clamps of the quick test -- a third of the instructions of a one-cell workgroup -- are paid once per
// workgroup; the patch rows of the quick test are immediate offsets from one LDS address).
__global__ __launch_bounds__(256) void k_fast(const OrbLevel* __restrict__ L, int num_levels, const FastCell* __restrict__ cells, int num_cells, int cpw,
                                              const uint8_t* __restrict__ img0, size_t img0_frame_stride, int img0_pitch,
                                              const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                              const unsigned short* __restrict__ gtab, unsigned long long* __restrict__ keys,
                                              int total_grid, int ini_thr, int min_thr, const uint8_t* __restrict__ mask,
                                              size_t mask_frame_stride, int mask_pitch, int mask_w, int mask_h) {
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[SV_ROI_MAX * FP + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_a[SV_ROI_MAX * FP];
    __shared__ unsigned short s_q[SV_CELL * SV_CELL];
    __shared__ unsigned long long s_key[FAST_KT * FAST_KT];  // per-block arg-max of the selection-grid cells the ROI touches
    __shared__ unsigned short s_gx[SV_ROI_MAX], s_gy[SV_ROI_MAX];  // selection-grid column / row of every ROI column / row
    __shared__ int s_count;
    int b, grp;
    xcd_frame_map(gridDim.x, gridDim.y, grp, b);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t* M = mask ? mask + (size_t)b * mask_frame_stride : nullptr;
    const uint8_t* const I0 = img0 + (size_t)b * img0_frame_stride;
    const uint8_t* const PY = pyr + (size_t)b * pyr_frame_bytes;
    unsigned long long* const KF = keys + (size_t)b * total_grid;
    uint8_t* const s_img = s_raw + 3;
    unsigned short* const my_q = s_q + wv * (SV_CELL * SV_CELL / 4);
    // ---- lane roles of the quick test (pass A below): a lane owns a 4 x 4 patch; 16 lanes span the 64 scored columns, the four lane
    //      groups of a wave and the four waves take the sixteen bands of four rows
    const int qrow = lane >> 4, qk = lane & 15;
    const int x0 = 3 + 4 * qk;                  // the lane's columns x0 .. x0 + 3 (ROI coordinates)
    const int ly0 = 3 + 4 * (4 * qrow + wv);    // ... and rows ly0 .. ly0 + 3
    const int e0 = (ly0 << 7) | x0;             // queue entry of (row 0, column 0) of the patch
    // pixel (ly, x) sits at s_raw[ly * FP + x + 3]; qbase = the aligned dword that holds the lane's first pixel of patch row 0 at byte 2.
    // Rows ly0 + g +- 3 <= 69 stay inside the 70-row LDS image whatever the cell's height (rows past the cell are zero-filled and their
    // results masked), so every row of the patch is an immediate offset from this one address.
    const uint32_t* const qbase = reinterpret_cast<const uint32_t*>(s_raw + ly0 * FP + 4 * qk);

    auto load_ring = [&](const uint8_t* c, int (&p)[16]) {
        p[0] = c[3 * FP];
        p[1] = c[3 * FP + 1];
        p[2] = c[2 * FP + 2];
        p[3] = c[FP + 3];
        p[4] = c[3];
        p[5] = c[-FP + 3];
        p[6] = c[-2 * FP + 2];
        p[7] = c[-3 * FP + 1];
        p[8] = c[-3 * FP];
        p[9] = c[-3 * FP - 1];
        p[10] = c[-2 * FP - 2];
        p[11] = c[-FP - 3];
        p[12] = c[-3];
        p[13] = c[FP - 3];
        p[14] = c[2 * FP - 2];
        p[15] = c[3 * FP - 1];
    };

    const int c_first = grp * cpw, c_last = min(c_first + cpw, num_cells);
    for (int ci = c_first; ci < c_last; ++ci) {
    const FastCell cell = cells[ci];
    const OrbLevel lev = L[cell.lv];
    auto masked = [&](int y, int x) -> bool {  // is_in_mask (orb_extractor.cc:168-170): (int)(y * scale), (int)(x * scale)
        int my = (int)((float)y * lev.scale), mx = (int)((float)x * lev.scale);
        my = min(my, mask_h - 1);
        mx = min(mx, mask_w - 1);
        return M[(size_t)my * mask_pitch + mx] == 0;
    };
    if (M) {  // skip the cell if one of its corners is masked (:219-225)
        const int y0 = cell.min_y, y1 = cell.min_y + cell.h, xa = cell.min_x, xb = cell.min_x + cell.w;
        if (masked(y0, xa) || masked(y1, xa) || masked(y0, xb) || masked(y1, xb)) continue;
    }
    const uint8_t* src;
    int spitch;
    if (cell.lv == 0) {
        src = I0;
        spitch = img0_pitch;
    }
    else {
        src = PY + lev.pyr_off;
        spitch = lev.pitch;
    }
    const int w = cell.w, h = cell.h;
    if (ci != c_first) __syncthreads();  // the previous cell's flush has read s_key, its last pass s_a / s_raw
    // ROI -> LDS by LDS-DMA (round 6: global_load_lds_dwordx4, no staging registers, no ds_write pass, any source alignment -- the three
    // load variants of rounds 1-5 were one per alignment class).  The ROI starts at x = 19 + 64j, i.e. 3 bytes past a 16-byte boundary of an
    // aligned level: the pieces start 3 bytes early (real pixels of the border band) and the LDS copy is addressed with a +3 skew.  Piece q = row * 5 +
    // piece-of-row goes to LDS byte 16 q (the rows are FP = 80 bytes apart); a thread owns pieces tid and tid + 256, so its row and column are the same
    // for every cell.  Rows past the cell and pieces past its width are NOT fetched: what the previous cell left there is never used (the quick test masks
    // those positions, the arc score and the NMS only visit queued pixels, whose rings lie inside the cell).
    const uint8_t* rsrc = src + (size_t)cell.min_y * spitch + (cell.min_x - 3);
    {
        const unsigned long long rbase = (unsigned long long)(uintptr_t)rsrc;
        const uint32_t lds_raw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)s_raw) + (uint32_t)wv * 1024u;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = tid + 256 * it, r = q / 5, c = q - 5 * r;  // (compile-time divisor)
            if (it == 0 || wv < 2) {                                   // pieces 256..349 belong to waves 0 and 1
                if (q < SV_ROI_MAX * (FP / 16) && r < h && 16 * c < w + 3) sv_glds16(rbase, (uint32_t)(__umul24(r, spitch) + 16 * c), lds_raw + 4096u * it);
            }
            if (q < SV_ROI_MAX * (FP / 16)) reinterpret_cast<uint4*>(s_a)[q] = make_uint4(0, 0, 0, 0);
        }
    }
    if (tid == 0) {
        s_count = 0;
    }
    if (tid < FAST_KT * FAST_KT) s_key[tid] = 0ull;
    if (tid < w) s_gx[tid] = gtab[lev.gtab_x_off + cell.min_x + tid - SV_PATCH_RADIUS];
    else if (tid >= 128 && tid - 128 < h) s_gy[tid - 128] = gtab[lev.gtab_y_off + cell.min_y + (tid - 128) - SV_PATCH_RADIUS];
    uint32_t band80 = 0;  // 0x80 per column of the patch inside the scored band
#pragma unroll
    for (int j = 0; j < 4; ++j) band80 |= (x0 + j < w - 3) ? (0x80u << (8 * j)) : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's own DMA pieces have landed; the barrier publishes them to the other waves
    __syncthreads();

    // cv::FAST at ini_thr; if the cell stays empty, once more at min_thr (:228-235).  The quick test, the queue and the arc scores are
    // rebuilt for the retry (rare: textured cells are never empty), so the common case queues and scores only what can exceed ini_thr.
    unsigned long long* K = KF + lev.grid_first;
    const int gx0 = s_gx[3], gy0 = s_gy[3];  // grid cell of the first scored pixel
    for (int pass = 0; pass < 2; ++pass) {
    const int t = pass == 0 ? ini_thr : min_thr, tq = t;
    // --- pass A: every pixel of the scored band [3, w-3) x [3, h-3) takes a 5-pixel quick test at this pass's threshold.
    //     Nine contiguous ring pixels always contain at least one pixel of every opposite pair (k, k + 8), so a
    //     9-arc brighter than v + t needs (p0 | p8) and (p4 | p12) brighter (same for darker): a necessary condition
    //     that ~90 % of the pixels fail.  Survivors are compacted into an LDS queue so that the expensive arc score
    //     below runs on full waves; it is exact, so queueing a superset of the corners is harmless.
    //       brighter on both opposite pairs  <=>  min(max(p0, p8), max(p4, p12)) > v + t
    //       darker   on both opposite pairs  <=>  max(min(p0, p8), min(p4, p12)) < v - t
    //     A lane tests FOUR horizontally adjacent pixels at once and owns a 4 x 4 patch: 16 lanes span the 64 scored columns, four
    //     steps walk the patch's rows, the four lane groups of a wave take four different bands of four rows.  The 10 bytes of the centre row and the 4 bytes of rows +-3 arrive as aligned
    //     dwords (ds_read2_b32), one v_perm_b32 per pixel pair lifts them straight into packed 16-bit lanes (no byte loads, no
    //     separate alignment step), and the whole test runs on v_pk_min / max / sub: ~9 instructions per pixel instead of ~16.
    //     Every wave owns a quarter of the queue.
    int wq = 0;
#if defined(FAST_ABLATE) && FAST_ABLATE >= 2  // profiling build only: no quick test either (staging, set-up and the empty passes remain)
    if (mask_h == -12345)
#endif
    {
        const uint32_t tqq = (uint32_t)(tq + 1) * 0x10001u;
        uint32_t m = 0;                           // bit 8 j + 4 + g: pixel (row g, column j) of the patch passed
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t* const rc = qbase + g * (FP / 4);  // W0 = [B-4, B), W1, W2, W3 of the centre row; rows +3 / -3
            const uint32_t w0 = rc[0], w1 = rc[1], w2 = rc[2], w3 = rc[3];
            const uint32_t d1 = rc[3 * (FP / 4) + 1], d2 = rc[3 * (FP / 4) + 2], u1 = rc[-3 * (FP / 4) + 1], u2 = rc[-3 * (FP / 4) + 2];
            auto pk = [](uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(hi, lo, sel)); };
            uint32_t e2[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // pixels (0, 1) and (2, 3)
                const s16x2 v = pk(w2, w1, half ? 0x0c050c04u : 0x0c030c02u);
                const s16x2 p0 = pk(d2, d1, half ? 0x0c050c04u : 0x0c030c02u);
                const s16x2 p8 = pk(u2, u1, half ? 0x0c050c04u : 0x0c030c02u);
                const s16x2 p12 = pk(w1, w0, half ? 0x0c060c05u : 0x0c040c03u);
                const s16x2 p4 = pk(w3, w2, half ? 0x0c040c03u : 0x0c020c01u);
                const s16x2 up = __builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12));
                const s16x2 dn = __builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12));
                const s16x2 mm = __builtin_elementwise_max(up - v, v - dn);
                e2[half] = __builtin_bit_cast(uint32_t, mm - __builtin_bit_cast(s16x2, tqq));  // >= 0 per half <=> max(..) > tq
            }
            // sign bytes of the four results -> one byte per pixel; a clear sign inside the band is a hit
            const uint32_t sgn = __builtin_amdgcn_perm(e2[1], e2[0], 0x07050301u);
            m = (m >> 1) | (~sgn & (ly0 + g < h - 3 ? band80 : 0u));
        }
        // Compaction, once per wave: exclusive prefix of the lanes' hit counts (row-scan adds), then every lane stores the hits of
        // its patch.  The queue order is patch by patch along a band of four rows -- neighbouring lanes of pass B then work on
        // neighbouring pixels, which its ring reads need (a queue in which a lane's entries were 16 rows apart cost +8 %).
        const int cnt = __popc(m);
        int incl = cnt;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);  // row_bcast:15
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);  // row_bcast:31
        wq = __builtin_amdgcn_readlane(incl, 63);
        int pos = incl - cnt;
        while (m) {
            const int bit = __ffs((int)m) - 1;  // 8 j + 4 + g
            m &= m - 1;
            my_q[pos++] = (unsigned short)(e0 + (((bit & 7) - 4) << 7) + (bit >> 3));
        }
    }
#if defined(FAST_ABLATE) && FAST_ABLATE >= 1  // profiling build only (tools/build_variant.sh NAME -DFAST_ABLATE=1): nothing survives the quick test (opaque to the compiler)
    if (mask_h != -12345) wq = 0;
#endif
    // every wave scores and filters ITS quarter of the queue: no index mapping
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the queue entries were written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // --- pass B: arc score of the candidates
    for (int i = lane; i < wq; i += 64) {
        const int e = my_q[i], ly = e >> 7, qx = e & 127;
        const uint8_t* c = &s_img[ly * FP + qx];
        int p[16];
        load_ring(c, p);
        s_a[ly * FP + qx] = (uint8_t)arc_score16(c[0], p);
    }
    __syncthreads();

    // --- per-cell NMS at this pass's threshold
    {
        int found = 0;
        for (int i = lane; i < wq; i += 64) {  // only queued pixels can have A > t
            const int e = my_q[i], ly = e >> 7, qx = e & 127;
            const uint8_t* a = &s_a[ly * FP + qx];
            const int A = a[0];
            if (A <= t) continue;
            const int s = A - 1;
            // s > (n > t ? n - 1 : 0) for all 8 neighbours  <=>  every neighbour's A is below this one's (A > t >= 1 makes the
            // "n <= t" branch always true and n <= t < A)
            const int n0 = max(max((int)a[-FP - 1], (int)a[-FP]), (int)a[-FP + 1]);
            const int n1 = max(max((int)a[-1], (int)a[1]), (int)a[FP - 1]);
            const int n2 = max(max((int)a[FP], (int)a[FP + 1]), n0);
            const bool keep = max(n1, n2) < A;
            if (!keep) continue;
            ++found;
            const int x_level = cell.min_x + qx, y_level = cell.min_y + ly;
            if (M && masked(y_level, x_level)) continue;  // keypoint filter (:246-256), after the retry decision
            const int gx = s_gx[qx], gy = s_gy[ly];
            const uint32_t order = (uint32_t)cell.order_base | ((uint32_t)ly << 7) | (uint32_t)qx;
            const unsigned long long key = ((unsigned long long)(uint32_t)s << 32) | (0xFFFFFFFFu - order);
            // arg-max per selection-grid cell: first inside the block (LDS), one global atomic per touched cell afterwards
            const int kx = gx - gx0, ky = gy - gy0;
            if ((unsigned)kx < FAST_KT && (unsigned)ky < FAST_KT) atomicMax(&s_key[ky * FAST_KT + kx], key);
            else atomicMax(&K[gy * lev.grid_x + gx], key);
        }
        if (found) atomicAdd(&s_count, found);
        __syncthreads();
        if (s_count > 0) break;
        __syncthreads();
    }
    }
    if (tid < FAST_KT * FAST_KT) {
        const unsigned long long key = s_key[tid];
        if (key) atomicMax(&K[(gy0 + tid / FAST_KT) * lev.grid_x + gx0 + tid % FAST_KT], key);
    }
    }
}

// ------------------------------------------------------------------------------------------------ select
// One block per frame: walk the selection grids level by level in cell-index order, compact the
// non-empty cells (this IS the output order of distribute_keypoints + extract), clear the keys.
__global__ __launch_bounds__(1024) void k_select(const OrbLevel* __restrict__ L, int num_levels, unsigned long long* __restrict__ keys,
                                                 int total_grid, int4* __restrict__ sel, int32_t* __restrict__ counts, int32_t* __restrict__ cellpos) {
    __shared__ int s_wave[16], s_lvcnt[SV_MAX_LEVELS];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* K = keys + (size_t)b * total_grid;
    int4* S = sel + (size_t)b * total_grid;
    if (tid < SV_MAX_LEVELS) s_lvcnt[tid] = 0;
    int running = 0;
    // the grids of all levels are contiguous (grid_first ascending): one flat pass in cell-index order, 1024 cells per trip
    for (int base = 0; base < total_grid; base += 1024) {
        const int idx = base + tid;
        unsigned long long key = 0;
        if (idx < total_grid) {
            key = K[idx];
            if (key) K[idx] = 0;
        }
        const unsigned long long bal = __ballot(key != 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();  // s_wave of the previous trip has been consumed; s_lvcnt is initialised
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < wave) woff += s_wave[k];
            total += s_wave[k];
        }
        // position of every grid cell in the emission order (= keypoints in the cells before it): k_describe_bands finds the run of a band with it
        if (cellpos && idx < total_grid) cellpos[(size_t)b * (total_grid + 1) + idx] = running + woff + before;
        if (key) {
            int lv = 0;
            while (lv + 1 < num_levels && idx >= L[lv + 1].grid_first) ++lv;
            const uint32_t order = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
            const int lx = order & 127, ly = (order >> 7) & 127, cid = order >> 14;
            const int cells_x = L[lv].cells_x;
            const int ci = cid / cells_x, cj = cid - ci * cells_x;
            int4 r;
            r.x = SV_PATCH_RADIUS + cj * SV_CELL + lx;
            r.y = SV_PATCH_RADIUS + ci * SV_CELL + ly;
            r.z = lv;
            r.w = (int)(key >> 32);
            S[running + woff + before] = r;
            atomicAdd(&s_lvcnt[lv], 1);
        }
        running += total;
    }
    __syncthreads();
    if (tid < num_levels) counts[b * (1 + num_levels) + 1 + tid] = s_lvcnt[tid];
    if (tid == 0) {
        counts[b * (1 + num_levels)] = running;
        if (cellpos) cellpos[(size_t)b * (total_grid + 1) + total_grid] = running;
    }
}

// ------------------------------------------------------------------------------------------------ describe
__device__ __forceinline__ float dev_fast_atan2(float y, float x) {  // cv::fastAtan2 (atan_f32), degrees
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ float dev_cos_poly(float v) {  // util/trigonometric.h:17-24
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}
__device__ __forceinline__ float dev_util_cos(float v) {  // util/trigonometric.h:26-42
    constexpr float PI_ = 3.14159265358979f;
    constexpr float PI_2 = PI_ / 2.0f;
    constexpr float TWO_PI = 2.0f * PI_;
    constexpr float INV_TWO_PI = 1.0f / TWO_PI;
    constexpr float THREE_PI_2 = 3.0f * PI_2;
    v = v - (float)(int)floorf(v * INV_TWO_PI) * TWO_PI;
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return dev_cos_poly(v);
    else if (v < PI_) return -dev_cos_poly(PI_ - v);
    else if (v < THREE_PI_2) return -dev_cos_poly(v - PI_);
    else return dev_cos_poly(TWO_PI - v);
}
__device__ __forceinline__ float dev_util_sin(float v) {
    constexpr float PI_2 = 3.14159265358979f / 2.0f;
    return dev_util_cos(PI_2 - v);
}

__device__ __forceinline__ int wave_sum(int v) {
    // row scans with DPP adds (one VALU instruction each, no LDS round trip), then the row totals across rows; the wave total
    // lands in lane 63 and is broadcast through a scalar register.  Integer adds: any order gives the same sum.
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8   -> lane 15 of every row holds the row sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- orientation + descriptors.  One wave per keypoint, DESC_KPW keypoints per wave one after the other (the rBRIEF pattern
// and the disc weights stay in registers across them).  Per keypoint:
//   * the 31 x 31 un-blurred and 37 x 37 blurred patches are copied to LDS with a handful of 8-byte loads issued at the
//     patch's own (arbitrary) byte address -- global memory takes unaligned accesses -- so that patch column 0 sits at
//     byte 0 of every LDS row;
//   * intensity centroid (orb_impl.cc:68-91) on dwords: a lane takes (row, 4 columns) items, v_dot4_u32_u8 against the
//     disc mask and against (u + 15) x mask gives sum I and sum (u + 15) I of its 4 pixels; m10 = sum u I, m01 = sum v I
//     are exact integers, so the order of summation is free;
//   * fastAtan2, util::cos / sin, 4 ballot rounds x 64 rotated pairs read from the blurred patch in LDS (orb_impl.cc:93-154).
// (A three-phase variant that computes the angle / cos / sin of 64 keypoints lane-parallel issues half the instructions
// per keypoint but measured slower, 121-136 us against 109: the longer per-wave chains of dependent patch fetches cost
// more than the saved issue slots.)
#ifndef DESC_KPW
#define DESC_KPW 5  // keypoints per wave.  Stand-alone per 1 024 frames (round 5): 3 / 4 / 5 / 6 -> 1.29 / 1.25 / 1.20 / 1.22 ms; round 4, pipeline: 2 / 3 / 4 / 5 / 6 / 8 -> 207 / 207.5 / 208.9 / 210.3 / 208.9 / 206 k frames/s: flat (fewer
                    // keypoints mean more waves per SIMD, more of them amortise the orientation pass: neither is what limits the kernel)
#endif
#define DESC_IP 32                     // LDS row pitch of the 31 x 31 patch (8 dwords)
#define DESC_BP 40                     // LDS row pitch of the 37 x 37 patch
#define DESC_R 18                      // largest |rounded rotated pattern coordinate| (pattern radius 18.38)
#define DESC_SLAB (DESC_KPW * 37 * DESC_BP + 16)
struct IcWeights {
    uint32_t w1[256], wu[256];  // per (row, dword) item: disc mask bytes, (u + 15) x mask bytes
};
constexpr IcWeights make_ic_weights() {
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // orb_impl.cc:51-66
    IcWeights t{};
    for (int row = 0; row < 31; ++row)
        for (int j = 0; j < 8; ++j) {
            uint32_t w1 = 0, wu = 0;
            const int v = row - 15, av = v < 0 ? -v : v;
            for (int k = 0; k < 4; ++k) {
                const int col = 4 * j + k, u = col - 15, au = u < 0 ? -u : u;
                if (col < 31 && au <= umax[av]) {
                    w1 |= 1u << (8 * k);
                    wu |= (uint32_t)(u + 15) << (8 * k);
                }
            }
            t.w1[row * 8 + j] = w1;
            t.wu[row * 8 + j] = wu;
        }
    return t;
}
__device__ __constant__ IcWeights c_icw = make_ic_weights();

__global__ __launch_bounds__(256) void k_describe(const OrbLevel* __restrict__ L, int num_levels, const int4* __restrict__ sel,
                                                  int total_grid, const int32_t* __restrict__ counts,
                                                  const uint8_t* __restrict__ img0, size_t img0_frame_stride, int img0_pitch,
                                                  const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                  const uint8_t* __restrict__ blur, size_t blur_frame_bytes,
                                                  svgpu_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, int cap, float* __restrict__ angles) {
    __shared__ __attribute__((aligned(16))) uint8_t s_slab[4][DESC_SLAB];
    int b, blk;
    xcd_frame_map(gridDim.x, gridDim.y, blk, b);
    // the wave index as a SCALAR: everything addressed through it (selection entries, level records) then comes in through
    // scalar loads issued together, instead of one dependent vector-load round trip per keypoint and field
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = min(counts[b * (1 + num_levels)], cap);
    const int i0 = (blk * 4 + wave) * DESC_KPW;
    if (i0 >= n) return;
    uint8_t* const slab = s_slab[wave];  // DESC_KPW un-blurred 31 x 32 patches, later overwritten by DESC_KPW blurred 37 x 40 patches
    // rBRIEF pattern of this lane's four pairs as floats; disc weights of this lane's (row, dword) items
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 q = reinterpret_cast<const float4*>(c_pattern_f)[r * 64 + lane];
        px0[r] = q.x;
        py0[r] = q.y;
        px1[r] = q.z;
        py1[r] = q.w;
    }
    uint32_t w1[4], wu[4];
    int rowv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int item = lane + 64 * m;  // items 248..255 carry zero weights
        w1[m] = c_icw.w1[item];
        wu[m] = c_icw.wu[item];
        rowv[m] = (item >> 3) - 15;
    }
    const int nk = min(DESC_KPW, n - i0);  // keypoints of this wave (wave-uniform)
    const int4* const S = sel + (size_t)b * total_grid + i0;
    const uint8_t* const I0 = img0 + (size_t)b * img0_frame_stride;
    const uint8_t* const PY = pyr + (size_t)b * pyr_frame_bytes;
    const uint8_t* const BL = blur + (size_t)b * blur_frame_bytes;
    const int part = lane & 3, prow = lane >> 2;          // 31 x 31 patch: 4 x 8 bytes per row, 16 rows per pass
    const int br = lane / 5, bpart = lane - 5 * br;       // 37 x 37 patch: 5 x 8 bytes per row, 12 rows per pass
    // ---- phase 1: the un-blurred patches of ALL keypoints of the wave -> LDS (loads issued back to back), then their moments.
    // Keypoints keep 19 px to every border: the 32- / 40-byte rows stay inside the image rows (the blurred rows may run 3 bytes
    // into the row padding / next row, never past the buffer: 256 bytes of slack).
    // selection entries and level records of all keypoints first (scalar loads, unconditional so that they are issued together;
    // slots beyond the wave's last keypoint repeat it and are never used)
    int4 sv[DESC_KPW];
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk) sv[kk] = S[kk];  // the selection buffer is padded by DESC_KPW entries (svgpu_orb.hip)
    const uint8_t* gi[DESC_KPW];
    const uint8_t* gbl[DESC_KPW];
    int pit_i[DESC_KPW], pit_b[DESC_KPW];
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk) {
        const int lv = kk < nk ? sv[kk].z : 0;  // slots past the wave's last keypoint hold stale entries: keep their level index in range
        pit_b[kk] = L[lv].pitch;
        pit_i[kk] = lv == 0 ? img0_pitch : pit_b[kk];
        gi[kk] = (lv == 0 ? I0 : PY + L[lv].pyr_off) + (size_t)(sv[kk].y - 15) * pit_i[kk] + (sv[kk].x - 15);
        gbl[kk] = BL + L[lv].blur_off + (size_t)(sv[kk].y - DESC_R) * pit_b[kk] + (sv[kk].x - DESC_R);
    }
    uint2 vi[DESC_KPW][2];
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (16 * h + prow < 31) __builtin_memcpy(&vi[kk][h], gi[kk] + (__umul24(16 * h + prow, pit_i[kk]) + 8 * part), 8);  // unaligned 8-byte load
        }
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (16 * h + prow < 31) *reinterpret_cast<uint2*>(slab + kk * (31 * DESC_IP) + (16 * h + prow) * DESC_IP + 8 * part) = vi[kk][h];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the blurred patches are requested now and land while the moments and the angle are computed
    uint2 vb[DESC_KPW][4];
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (br < 12 && 12 * h + br < 37) __builtin_memcpy(&vb[kk][h], gbl[kk] + (__umul24(12 * h + br, pit_b[kk]) + 8 * bpart), 8);
        }
    int my10 = 0, my01 = 0;  // lane kk keeps the moments of keypoint kk
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
            int m10 = 0, m01 = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int item = lane + 64 * m;
                const uint32_t px = item < 248 ? reinterpret_cast<const uint32_t*>(slab + kk * (31 * DESC_IP))[item] : 0u;
                const int s1 = (int)__builtin_amdgcn_udot4(px, w1[m], 0u, false), su = (int)__builtin_amdgcn_udot4(px, wu[m], 0u, false);
                m10 += su - __mul24(15, s1);  // 24-bit multiplies: v_mul_lo_u32 is quarter rate
                m01 += __mul24(rowv[m], s1);
            }
            m10 = wave_sum(m10);
            m01 = wave_sum(m01);
            if (lane == kk) {
                my10 = m10;
                my01 = m01;
            }
        }
    // ---- phase 2: orientation of all keypoints at once (lane kk works for keypoint kk; the other lanes idle along)
    const float angle = dev_fast_atan2((float)my01, (float)my10);
    const float rad = (float)((double)angle * 3.14159265358979323846 / 180.0);
    const float ca_l = dev_util_cos(rad), sa_l = dev_util_sin(rad);
    // ---- phase 3: blurred patches -> LDS (same region), rotated BRIEF (orb_impl.cc:93-154)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (br < 12 && 12 * h + br < 37) *reinterpret_cast<uint2*>(slab + kk * (37 * DESC_BP) + __mul24(12 * h + br, DESC_BP) + 8 * bpart) = vb[kk][h];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < DESC_KPW; ++kk)
        if (kk < nk) {
            const float ca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ca_l), kk));
            const float sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa_l), kk));
            const uint8_t* B = slab + kk * (37 * DESC_BP) + DESC_R * DESC_BP + DESC_R;
            unsigned long long bits[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x0 = px0[r], y0 = py0[r], x1 = px1[r], y1 = py1[r];
                // cvRound = round half to even: adding 1.5 * 2^23 leaves the rounded integer in the low mantissa bits (|coordinate| < 2^22),
                // one add instead of v_rndne + v_cvt; the 24-bit multiply sees 0x400000 + row, the constant folds into the LDS base
                constexpr float RN = 12582912.0f;
                constexpr int RK = 0x400000 * DESC_BP + 0x4B400000;
                // (row, column) of a point as ONE packed pair: v_pk_mul_f32 / v_pk_add_f32 do both halves per instruction with the same IEEE
                // roundings as the scalar forms (4 instructions per point instead of 8: the rotation is half of this kernel's VALU work)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 sc = {sa, ca}, cs = {ca, -sa}, rn = {RN, RN};
                const f32x2 q0 = (f32x2{x0, x0} * sc + f32x2{y0, y0} * cs) + rn;  // (x sa + y ca, x ca - y sa): a - b == a + (-b) exactly
                const f32x2 q1 = (f32x2{x1, x1} * sc + f32x2{y1, y1} * cs) + rn;
                const int r0 = __float_as_int(q0.x), c0 = __float_as_int(q0.y);
                const int r1 = __float_as_int(q1.x), c1 = __float_as_int(q1.y);
                const int a = B[__mul24(r0, DESC_BP) + c0 - RK];
                const int bb = B[__mul24(r1, DESC_BP) + c1 - RK];
                bits[r] = __builtin_amdgcn_ballot_w64(a < bb);
            }
            uint8_t* D = desc + ((size_t)b * cap + i0 + kk) * 32;
            if (lane < 4) reinterpret_cast<unsigned long long*>(D)[lane] = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        }
    if (lane < nk) {  // lane kk writes the record of keypoint kk
        const int4 s = S[lane];  // its own (vector) load: keeps the wave-uniform copies above in scalar registers
        const int lv = s.z;
        svgpu_keypoint k;
        k.x = (float)s.x;
        k.y = (float)s.y;
        if (lv != 0) {  // correct_keypoint_scale (orb_extractor.cc:337-345)
            k.x = k.x * L[lv].scale;
            k.y = k.y * L[lv].scale;
        }
        k.size = L[lv].kp_size;
        k.angle = angle;
        k.response = (float)s.w;
        k.octave = lv;
        k.class_id = -1;
        kps[(size_t)b * cap + i0 + lane] = k;
        if (angles) angles[(size_t)b * cap + i0 + lane] = angle;  // the packed copy the batched matcher's angle-bin sort reads (4 instead of 28 bytes per keypoint)
    }
}


// ---- orientation + descriptors from LDS-resident BANDS (round 6).  Measured on MI355X (profiles/r06_ubench_tcp_patterns.json,
// tools/pmc_kernel.sh): the per-keypoint patch fetches of k_describe above are bound by the vector-memory front end -- a wave-level
// load whose lanes touch 16-32 different image rows costs 28-100 cycles in the TA / TCP whatever its width, and every 32-48-byte
// patch row pulls a whole 128-byte line from L2 (4.5 cycles each): 300 cycles per keypoint per CU, of which the arithmetic is 110.
// LDS-DMA per patch (global_load_lds_dwordx4 into a per-wave ring, built and measured: 1.39 ms against 1.20) does not change that count.
// So the image goes to LDS in full rows instead, once per group of neighbouring keypoints:
//   * a BAND = a few consecutive rows of the selection grid of one level (svgpu_orb.hip builds the table); its keypoints are one
//     contiguous run of the selection order (k_select writes every cell's position), its pixels the level's rows
//     [y_min - 18, y_max + 18], full width: one workgroup of 8 waves copies them with 1 KB `global_load_lds_dwordx4` instructions
//     (lane-linear: LDS pitch = 16 x pieces per row, chosen = 32 mod 64 so that eight rows of dword reads hit 64 different banks);
//   * phase U: un-blurred rows -> moments of every keypoint of the band (three accumulating v_dot4_u32_u8 per dword), one wave per
//     keypoint, results to LDS;  phase A: thread t computes keypoint t's orientation / cos / sin / record (lane-parallel: the chain
//     runs once per 64 keypoints) while the blurred rows replace the un-blurred ones;  phase B: rotated BRIEF, one wave per keypoint,
//     byte gathers straight from the band.
// Same arithmetic as k_describe (orb_impl.cc:68-154); k_describe stays for configurations whose bands do not fit (very wide images).
#ifndef DB_THREADS
#define DB_THREADS 512
#endif
#define DB_WAVES (DB_THREADS / 64)
#define DB_MAX_KP 128  // keypoints per band (svgpu_orb.hip keeps bands below it): phase A gives each keypoint one thread of waves 0-1
struct IcWeights3 {
    uint32_t w1[256], wu[256], wv[256];  // per (row, dword) item: disc mask bytes, (u + 15) x mask, (v + 15) x mask
};
constexpr IcWeights3 make_ic_weights3() {
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // orb_impl.cc:51-66
    IcWeights3 t{};
    for (int row = 0; row < 31; ++row)
        for (int j = 0; j < 8; ++j) {
            uint32_t w1 = 0, wu = 0, wv = 0;
            const int v = row - 15, av = v < 0 ? -v : v;
            for (int k = 0; k < 4; ++k) {
                const int col = 4 * j + k, u = col - 15, au = u < 0 ? -u : u;
                if (col < 31 && au <= umax[av]) {
                    w1 |= 1u << (8 * k);
                    wu |= (uint32_t)(u + 15) << (8 * k);
                    wv |= (uint32_t)row << (8 * k);
                }
            }
            t.w1[row * 8 + j] = w1;
            t.wu[row * 8 + j] = wu;
            t.wv[row * 8 + j] = wv;
        }
    return t;
}
__device__ __constant__ IcWeights3 c_icw3 = make_ic_weights3();

// v_writelane_b32 has no builtin in this compiler.  gfx950 wants two wait states between a VALU instruction that writes an SGPR / VCC
// (v_cmp, v_readlane) and a VALU instruction that reads it; hipcc inserts them for its own instructions but does not look inside an asm
// statement, so the statement opens with s_nop 1 (without it the lanes written right behind a v_cmp took the OLD mask).
template <int LANE>
__device__ __forceinline__ void sv_writelane8(uint32_t& lo, uint32_t& hi, const unsigned long long (&m)[4]) {  // (lo, hi)[LANE + r] = m[r], r = 0..3
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\tv_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
        "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\tv_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)m[0]), "s"((uint32_t)(m[0] >> 32)), "s"((uint32_t)m[1]), "s"((uint32_t)(m[1] >> 32)), "s"((uint32_t)m[2]), "s"((uint32_t)(m[2] >> 32)),
          "s"((uint32_t)m[3]), "s"((uint32_t)(m[3] >> 32)), "n"(LANE), "n"(LANE + 1), "n"(LANE + 2), "n"(LANE + 3));
}

// rows [y0, y0 + nrows) of an image (row pitch gp, any alignment) -> LDS at lds0 with row pitch 16 * cpr, by the whole workgroup.
// An instruction carries `rpi` = 64 / cpr whole rows (lane = row_in_group * cpr + piece, so that LDS byte 16 * lane continues the linear
// image; the lanes behind rpi * cpr rest): the lane's source offset is loop-invariant and the group's base address advances in scalar
// registers -- no vector instruction per DMA.  Wave w takes the row groups w, w + 8, ...  (Rows wider than 1 KB: see db_stage_rows_wide.)
__device__ __forceinline__ void db_stage_rows(const uint8_t* img, int gp, int y0, int nrows, int cpr, uint32_t lds0, int lane, int wave) {
    const int rpi = 64 / cpr;                       // rows per instruction (>= 1: cpr <= 64)
    const int rsub = lane / cpr, csub = lane - rsub * cpr;
    const uint32_t voff = (uint32_t)(rsub * gp + csub * 16);
    const int groups = (nrows + rpi - 1) / rpi;
    const unsigned long long base = (unsigned long long)(uintptr_t)(img + (size_t)y0 * gp);
    // (the last group may hold fewer than rpi rows: its surplus lanes stay out -- the rows behind a level's last one belong to somebody else, and
    //  behind a caller's image to nobody)
    if (rsub < rpi)
        for (int g = wave; g < groups; g += DB_WAVES)
            if (g * rpi + rsub < nrows) sv_glds16(base + (unsigned long long)(g * rpi) * (unsigned)gp, voff, lds0 + (uint32_t)(g * rpi * cpr) * 16);
}
// the general form (pieces of a row spread over several instructions): piece q = row * cpr + c goes to LDS byte 16 q, instruction i covers
// pieces [64 i, 64 i + 64)
__device__ __forceinline__ void db_stage_rows_wide(const uint8_t* img, int gp, int y0, int nrows, int cpr, uint32_t lds0, int lane, int wave) {
    const int total = nrows * cpr;
    int q = wave * 64 + lane;
    int row = q / cpr, c = q - row * cpr;
    const int drow = (DB_WAVES * 64) / cpr, dc = (DB_WAVES * 64) - drow * cpr;
    const unsigned long long base = (unsigned long long)(uintptr_t)(img + (size_t)y0 * gp);
    uint32_t dst = lds0 + wave * 1024;
    for (int i = wave * 64; i < total; i += DB_WAVES * 64) {  // wave-uniform trip count
        if (q < total) sv_glds16(base, (uint32_t)(row * gp + c * 16), dst);
        q += DB_WAVES * 64;
        row += drow;
        c += dc;
        if (c >= cpr) {
            c -= cpr;
            ++row;
        }
        dst += DB_WAVES * 1024;
    }
}

__global__ __launch_bounds__(DB_THREADS) void k_describe_bands(const OrbLevel* __restrict__ L, int num_levels, const DescBand* __restrict__ bands, int num_bands,
                                                               const int4* __restrict__ sel, int total_grid, const int32_t* __restrict__ cellpos,
                                                               const int32_t* __restrict__ counts,
                                                               const uint8_t* __restrict__ img0, size_t img0_frame_stride, int img0_pitch,
                                                               const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                               const uint8_t* __restrict__ blur, size_t blur_frame_bytes,
                                                               svgpu_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, int cap, float* __restrict__ angles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_band[];  // the band's rows; behind them the per-keypoint arrays
    int b, bi;
    xcd_frame_map(gridDim.x, gridDim.y, bi, b);
    const DescBand bd = bands[bi];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t* const P = cellpos + (size_t)b * (total_grid + 1);
    const int n = min(counts[b * (1 + num_levels)], cap);
    const int p0 = P[bd.cell0], nkp = min(P[bd.cell1], n) - p0;  // the band's keypoints: selection entries [p0, p0 + nkp)
    if (nkp <= 0) return;
    const int lv = bd.lv, lp = bd.lp, cpr = lp >> 4;
    int2* const s_xy = reinterpret_cast<int2*>(s_band + bd.img_bytes);  // (x, y) of keypoint j
    int2* const s_aux = s_xy + DB_MAX_KP;                               // (m10, m01), later (cos, sin) as float bits
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)s_band);
    const int4* const S = sel + (size_t)b * total_grid + p0;
    int4 sv = make_int4(0, 0, 0, 0);
    if (tid < nkp) {
        sv = S[tid];
        s_xy[tid] = make_int2(sv.x, sv.y);
    }
    // disc weights of this lane's (row, dword) items (lane = 8 * row + dword, four row groups); rBRIEF pattern of its four pairs
    uint32_t w1[4], wu[4], wv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int item = lane + 64 * m;  // items 248..255 (patch row 31) carry zero weights
        w1[m] = c_icw3.w1[item];
        wu[m] = c_icw3.wu[item];
        wv[m] = c_icw3.wv[item];
    }
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 q = reinterpret_cast<const float4*>(c_pattern_f)[r * 64 + lane];
        px0[r] = q.x;
        py0[r] = q.y;
        px1[r] = q.z;
        py1[r] = q.w;
    }
    // ---- phase U: un-blurred rows -> LDS, moments (orb_impl.cc:68-91): m10 = SU - 15 S1, m01 = SV - 15 S1, exact integers
    {
        const uint8_t* img = lv == 0 ? img0 + (size_t)b * img0_frame_stride : pyr + (size_t)b * pyr_frame_bytes + L[lv].pyr_off;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the loads above are not counted against the DMA below
        if (cpr <= 64) db_stage_rows(img, lv == 0 ? img0_pitch : L[lv].pitch, bd.yu0, bd.nru, cpr, lds0, lane, wave);
        else db_stage_rows_wide(img, lv == 0 ? img0_pitch : L[lv].pitch, bd.yu0, bd.nru, cpr, lds0, lane, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    {
        const int lane_off = (lane >> 3) * lp + 4 * (lane & 7), step = 8 * lp;
        for (int j = wave; j < nkp; j += DB_WAVES) {
            const int2 xy = s_xy[j];
            const int x = __builtin_amdgcn_readfirstlane(xy.x), y = __builtin_amdgcn_readfirstlane(xy.y);
            // the patch starts at any byte; misaligned ds_read_b32 is served several times slower by the LDS (SQ_LDS_IDX_ACTIVE: 17 cycles per
            // LDS instruction with them), so: the two aligned dwords around each item (ds_read2_b32) + one v_alignbit
            const int org = (y - 15 - bd.yu0) * lp + (x - 15);
            const uint32_t* U = reinterpret_cast<const uint32_t*>(s_band + ((org & ~3) + lane_off));
            const uint32_t sh = (org & 3) * 8;
            uint32_t s1 = 0, su = 0, sw = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t lo = U[m * (step >> 2)], hi = U[m * (step >> 2) + 1];
                const uint32_t px = __builtin_amdgcn_alignbit(hi, lo, sh);
                s1 = __builtin_amdgcn_udot4(px, w1[m], s1, false);
                su = __builtin_amdgcn_udot4(px, wu[m], su, false);
                sw = __builtin_amdgcn_udot4(px, wv[m], sw, false);
            }
            const int m10 = wave_sum((int)su - __mul24(15, (int)s1)), m01 = wave_sum((int)sw - __mul24(15, (int)s1));
            if (lane == 0) s_aux[j] = make_int2(m10, m01);
        }
    }
    __syncthreads();  // every wave is done with the un-blurred rows; the moments are in LDS
    // ---- phase A: blurred rows -> LDS (same region); meanwhile thread t: orientation, cos / sin and the record of keypoint t
    if (cpr <= 64) db_stage_rows(blur + (size_t)b * blur_frame_bytes + L[lv].blur_off, L[lv].pitch, bd.yb0, bd.nrb, cpr, lds0, lane, wave);
    else db_stage_rows_wide(blur + (size_t)b * blur_frame_bytes + L[lv].blur_off, L[lv].pitch, bd.yb0, bd.nrb, cpr, lds0, lane, wave);
    if (tid < nkp) {
        const int2 mm = s_aux[tid];
        const float angle = dev_fast_atan2((float)mm.y, (float)mm.x);
        const float rad = (float)((double)angle * 3.14159265358979323846 / 180.0);
        s_aux[tid] = make_int2(__float_as_int(dev_util_cos(rad)), __float_as_int(dev_util_sin(rad)));
        svgpu_keypoint k;
        k.x = (float)sv.x;
        k.y = (float)sv.y;
        if (lv != 0) {  // correct_keypoint_scale (orb_extractor.cc:337-345)
            k.x = k.x * L[lv].scale;
            k.y = k.y * L[lv].scale;
        }
        k.size = L[lv].kp_size;
        k.angle = angle;
        k.response = (float)sv.w;
        k.octave = lv;
        k.class_id = -1;
        kps[(size_t)b * cap + p0 + tid] = k;
        if (angles) angles[(size_t)b * cap + p0 + tid] = angle;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- phase B: rotated BRIEF on the blurred rows (orb_impl.cc:93-154)
    {
        constexpr float RN = 12582912.0f;  // cvRound = round half to even: adding 1.5 * 2^23 leaves the integer in the low mantissa bits
        const int rk = 0x400000 * lp + 0x4B400000;  // what the biased (row, column) pair adds to row * lp + column
        for (int j = wave; j < nkp; j += DB_WAVES) {
            const int2 xy = s_xy[j], cs2 = s_aux[j];
            const int x = __builtin_amdgcn_readfirstlane(xy.x), y = __builtin_amdgcn_readfirstlane(xy.y);
            const float ca = __int_as_float(__builtin_amdgcn_readfirstlane(cs2.x)), sa = __int_as_float(__builtin_amdgcn_readfirstlane(cs2.y));
            const uint8_t* B = s_band + ((y - bd.yb0) * lp + x - rk);
            unsigned long long bits[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 sc = {sa, ca}, cs = {ca, -sa}, rn = {RN, RN};
                const f32x2 q0 = (f32x2{px0[r], px0[r]} * sc + f32x2{py0[r], py0[r]} * cs) + rn;  // (x sa + y ca, x ca - y sa)
                const f32x2 q1 = (f32x2{px1[r], px1[r]} * sc + f32x2{py1[r], py1[r]} * cs) + rn;
                const int a = B[__mul24(__float_as_int(q0.x), lp) + __float_as_int(q0.y)];
                const int bb = B[__mul24(__float_as_int(q1.x), lp) + __float_as_int(q1.y)];
                bits[r] = __builtin_amdgcn_ballot_w64(a < bb);
            }
            uint32_t d_lo = 0, d_hi = 0;  // lane r: bits 64 r .. 64 r + 63
            sv_writelane8<0>(d_lo, d_hi, bits);
            if (lane < 4) reinterpret_cast<uint2*>(desc + ((size_t)b * cap + p0 + j) * 32)[lane] = make_uint2(d_lo, d_hi);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers
void sv_launch_resize(hipStream_t s, const uint8_t* src, size_t src_frame_stride, int src_pitch, int sw, int sh,
                      uint8_t* dst, size_t dst_frame_stride, int dst_pitch, int dw, int dh, const short* xofs,
                      const short2* xa, const short2* yofs, const short2* yb, int batch) {
    (void)sh;
    dim3 block(64, 4), grid((dw + 255) / 256, (dh + 3) / 4, batch);
    hipLaunchKernelGGL(k_resize, grid, block, 0, s, src, src_frame_stride, src_pitch, sw, dst, dst_frame_stride, dst_pitch, dw,
                       dh, xofs, xa, yofs, yb);
}

hipError_t sv_pyramid_prepare() {  // once per device: allow the LDS-resident pyramid its large dynamic allocation
    return sv_allow_dynamic_lds(reinterpret_cast<const void*>(k_pyramid_lds), SV_PYR_LDS_MAX);
}
void sv_launch_pyramid(hipStream_t s, const OrbLevel* levels, int num_levels, const int2* band_rows, int bands, const uint8_t* img0,
                       size_t img0_frame_stride, int img0_pitch, uint8_t* pyr, size_t pyr_frame_bytes, const short* xofs,
                       const short2* xa, const short2* yofs, const short2* yb, const uint32_t* xg, const short4* yrow, int batch, size_t lds_bytes) {
    if (lds_bytes > 0) {
        hipLaunchKernelGGL(k_pyramid_lds, dim3(bands, batch), dim3(PYR_THREADS), lds_bytes, s, levels, num_levels, band_rows, bands, img0,
                           img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, xg, yrow);
        return;
    }
    hipLaunchKernelGGL(k_pyramid, dim3(bands, batch), dim3(PYR_THREADS), 0, s, levels, num_levels, band_rows, img0, img0_frame_stride,
                       img0_pitch, pyr, pyr_frame_bytes, xofs, xa, yofs, yb);
}

void sv_launch_blur(hipStream_t s, const OrbLevel* levels, int num_levels, int total_tiles, const uint8_t* img0,
                    size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr, size_t pyr_frame_bytes, uint8_t* blur,
                    size_t blur_frame_bytes, int batch, bool need_gather, int rows) {
    const dim3 grid(total_tiles, batch), block(256);
    if (rows == BLUR_ROWS_SMALL) {
        hipLaunchKernelGGL(k_blur<BLUR_ROWS_SMALL>, grid, block, 0, s, levels, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes);
        if (need_gather)
            hipLaunchKernelGGL(k_blur_gather<BLUR_ROWS_SMALL>, grid, block, 0, s, levels, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes);
    }
    else {
        hipLaunchKernelGGL(k_blur<BLUR_ROWS>, grid, block, 0, s, levels, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes);
        if (need_gather)
            hipLaunchKernelGGL(k_blur_gather<BLUR_ROWS>, grid, block, 0, s, levels, num_levels, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes);
    }
}

void sv_launch_fast(hipStream_t s, const OrbLevel* levels, int num_levels, const FastCell* cells, int num_cells,
                    const uint8_t* img0, size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr,
                    size_t pyr_frame_bytes, const unsigned short* gtab, unsigned long long* keys, int total_grid,
                    int ini_thr, int min_thr, const uint8_t* mask, size_t mask_frame_stride, int mask_pitch, int mask_w,
                    int mask_h, int batch) {
    if (num_cells == 0) return;
    // cells per workgroup: four once the batch alone fills the chip many times over, one for a few frames (latency: more workgroups)
    int cpw = (long long)num_cells * batch >= 16384 ? 4 : 1;
    if (const char* e = getenv("SVGPU_FAST_CPW")) cpw = std::max(1, atoi(e));
    hipLaunchKernelGGL(k_fast, dim3((num_cells + cpw - 1) / cpw, batch), dim3(256), 0, s, levels, num_levels, cells, num_cells, cpw, img0, img0_frame_stride,
                       img0_pitch, pyr, pyr_frame_bytes, gtab, keys, total_grid, ini_thr, min_thr, mask, mask_frame_stride,
                       mask_pitch, mask_w, mask_h);
}

void sv_launch_select(hipStream_t s, const OrbLevel* levels, int num_levels, unsigned long long* keys, int total_grid,
                      int4* sel, int32_t* counts, int32_t* cellpos, int batch) {
    hipLaunchKernelGGL(k_select, dim3(batch), dim3(1024), 0, s, levels, num_levels, keys, total_grid, sel, counts, cellpos);
}

void sv_launch_describe(hipStream_t s, const OrbLevel* levels, int num_levels, const int4* sel, int total_grid,
                        const int32_t* counts, const uint8_t* img0, size_t img0_frame_stride, int img0_pitch,
                        const uint8_t* pyr, size_t pyr_frame_bytes, const uint8_t* blur, size_t blur_frame_bytes,
                        svgpu_keypoint* kps, uint8_t* desc, int cap, int batch, float* angles) {
    if (total_grid == 0) return;
    hipLaunchKernelGGL(k_describe, dim3((total_grid + 4 * DESC_KPW - 1) / (4 * DESC_KPW), batch), dim3(256), 0, s, levels, num_levels, sel, total_grid,
                       counts, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes, kps, desc, cap, angles);
}

hipError_t sv_describe_bands_prepare(size_t lds_bytes) {  // once per device: allow the band kernel its dynamic LDS
    return sv_allow_dynamic_lds(reinterpret_cast<const void*>(k_describe_bands), lds_bytes);
}
void sv_launch_describe_bands(hipStream_t s, const OrbLevel* levels, int num_levels, const DescBand* bands, int num_bands, size_t lds_bytes,
                              const int4* sel, int total_grid, const int32_t* cellpos, const int32_t* counts, const uint8_t* img0,
                              size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr, size_t pyr_frame_bytes, const uint8_t* blur,
                              size_t blur_frame_bytes, svgpu_keypoint* kps, uint8_t* desc, int cap, int batch, float* angles) {
    if (num_bands == 0) return;
    hipLaunchKernelGGL(k_describe_bands, dim3(num_bands, batch), dim3(DB_THREADS), lds_bytes, s, levels, num_levels, bands, num_bands, sel, total_grid,
                       cellpos, counts, img0, img0_frame_stride, img0_pitch, pyr, pyr_frame_bytes, blur, blur_frame_bytes, kps, desc, cap, angles);
}
