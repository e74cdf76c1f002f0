// Shared host-side pieces of the matcher entry points (svgpu_match.hip, svgpu_match2.hip): scratch arena, the frame side of the
// cell matchers, and the two-pass "grid build -> candidate lists -> candidate matcher" driver.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "svgpu_internal.h"
#include "match_kernels.h"
#include "frame_kernels.h"

namespace svm {

// bump allocator over ctx->d_scratch (256-byte aligned pieces)
struct Arena {
    char* base;
    size_t off = 0;
    // Batched uploads: with a page-locked mirror of the arena (ctx->h_stage) every host array is copied to the mirror at its arena offset and
    // ONE host-to-device copy of the touched range follows (flush) -- the runtime turns every small copy from pageable memory into a staging
    // kernel of its own (~5 us each on the stream: ten of them per cell-matcher call were half of what the call waited for).  Device-only
    // pieces inside the range receive stale bytes, harmlessly: the kernels that produce them run behind the copy.
    char* mirror = nullptr;
    size_t up_lo = ~size_t(0), up_hi = 0;
    explicit Arena(void* p) : base((char*)p) {}
    int upload(svgpu_ctx* ctx, hipStream_t s, void* dst, const void* src, size_t bytes) {
        if (!mirror) {
            SV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
            return SVGPU_OK;
        }
        const size_t o = (size_t)((char*)dst - base);
        memcpy(mirror + o, src, bytes);
        up_lo = std::min(up_lo, o);
        up_hi = std::max(up_hi, o + bytes);
        return SVGPU_OK;
    }
    int flush(svgpu_ctx* ctx, hipStream_t s) {
        if (mirror && up_hi > up_lo) SV_HIP(ctx, hipMemcpyAsync(base + up_lo, mirror + up_lo, up_hi - up_lo, hipMemcpyHostToDevice, s));
        up_lo = ~size_t(0), up_hi = 0;
        return SVGPU_OK;
    }
    template <class T>
    T* take(size_t n) {
        T* r = (T*)(base + off);
        off += (n * sizeof(T) + 255) & ~size_t(255);
        return r;
    }
};
inline size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }

// Batched read-backs, the counterpart of Arena::upload: results that live in the arena are requested with add(), fetch() copies the
// range(s) that cover them into the page-locked mirror (requests closer than 32 KB share one copy), and after the stream has been
// synchronised scatter() hands them to the caller's arrays.
struct Downloads {
    struct Item {
        void* dst;
        size_t off, bytes;
    };
    std::vector<Item> items;
    void add(const Arena& A, void* dst, const void* src, size_t bytes) {
        if (dst && bytes) items.push_back({dst, (size_t)((const char*)src - A.base), bytes});
    }
    int fetch(svgpu_ctx* ctx, hipStream_t s, const Arena& A) {
        std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.off < b.off; });
        size_t k = 0;
        while (k < items.size()) {
            size_t lo = items[k].off, hi = lo + items[k].bytes;
            ++k;
            while (k < items.size() && items[k].off <= hi + 32768) {
                hi = std::max(hi, items[k].off + items[k].bytes);
                ++k;
            }
            SV_HIP(ctx, hipMemcpyAsync(A.mirror + lo, A.base + lo, hi - lo, hipMemcpyDeviceToHost, s));
        }
        return SVGPU_OK;
    }
    void scatter(const Arena& A) const {
        for (const Item& it : items) memcpy(it.dst, A.mirror + it.off, it.bytes);
    }
};

// scratch for the angle-bin sorted copies of both sides (see k_bf_binsort)
inline size_t sort_bytes(int pairs, int cap1, int cap2) {
    const size_t p = (size_t)pairs;
    return pad(p * cap1 * 32) + pad(p * cap2 * 32) + 2 * pad(p * cap1 * 4) + 2 * pad(p * cap2 * 4) + 2 * pad(p * 362 * 4) + pad(p * 2 * 4);
}
inline void take_sort(Arena& A, BfProblem& P, int pairs, int cap1, int cap2) {
    const size_t p = (size_t)pairs;
    P.sd1 = A.take<uint32_t>(p * cap1 * 8);
    P.sd2 = A.take<uint32_t>(p * cap2 * 8);
    P.sa1 = A.take<float>(p * cap1);
    P.si1 = A.take<int>(p * cap1);
    P.sa2 = A.take<float>(p * cap2);
    P.si2 = A.take<int>(p * cap2);
    P.bs1 = A.take<int>(p * 362);
    P.bs2 = A.take<int>(p * 362);
    P.prune_ok = A.take<int>(p * 2);
}

// The frame side of the cell matcher (keypoints that get binned) -- host pointers.
struct InCellsFrame {
    const uint8_t* tdesc;
    const float* t_xy;
    const int32_t* t_octave;
    int nt;
    const uint8_t* occupied;
    const float* t_angle;
    const float* t_xright;
    float min_x, max_x, min_y, max_y;
    int grid_cols, grid_rows;
    const svgpu_frame* res = nullptr;  // resident frame (svgpu_frame_bind): tdesc / t_xy / t_octave / t_angle / t_xright are then ITS device arrays,
                                       // nothing of the keypoint side is uploaded and its grid is not rebuilt; `occupied` stays a host array
};
// svgpu_frame_bind: the keypoint-side arguments of the entry point that calls this are replaced by the bound frame's (one-shot)
inline const svgpu_frame* sv_take_bound_frame(svgpu_ctx* ctx) {
    if (!ctx) return nullptr;
    const svgpu_frame* f = ctx->bound_frame;
    ctx->bound_frame = nullptr;
    return f;
}

// Candidate lists built on the device + candidate matcher.  `stage(A, fresh, P, G)` places the query-side arrays in the
// arena (uploading or generating them when `fresh`) and points P / G at them; `finish(P)` enqueues extra read-backs.
// Pass 0 builds the grid and the list sizes and reads the total back; the scratch arena may then have to grow for the
// lists, which discards its contents, so pass 1 repeats the (cheap) staging and the grid build in the final arena.
template <class Stage, class Finish>
int in_cells_core(svgpu_ctx* ctx, int nq, const InCellsFrame& F, size_t query_bytes, int check_orientation, unsigned thr, float lowe_ratio,
                  int mode, Stage&& stage, Finish&& finish, int32_t* match_q, int* num_matches) {
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    static const bool mtrace = std::getenv("SVGPU_MATCH_TRACE") != nullptr;  // host-side phase times of a call, to stderr
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = tnow();
    auto lap = [&](const char* what) {
        if (!mtrace) return;
        const auto t = tnow();
        std::fprintf(stderr, "[match] %-22s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
        t_prev = t;
    };
    const int nt = F.nt, ncell = F.grid_cols * F.grid_rows;
    // target side: descriptors, xy, seven nt x 4 arrays (t_octave, t_angle, t_xright, cell_of, cell_items, owner, mdist), occupied;
    // query side: cand_off, match_q, match; P.num; slack for the alignment of each take
    const size_t need1 = query_bytes + pad((size_t)nt * 32) + pad((size_t)nt * 8) + 7 * pad((size_t)nt * 4) + pad(nt)
                         + pad((size_t)(ncell + 1) * 4) + pad((size_t)(nq + 1) * 4) + 2 * pad((size_t)nq * 4) + pad(4) + 2048;
    int total = 0;
    // Sizing of the candidate lists.  Exact: pass 0 builds the grid and the list sizes and reads their total back (a synchronisation in the
    // middle of the call), pass 1 fills and matches.  Speculative (a capacity that sufficed for this context before, ctx->cand_cap_hint):
    // everything is enqueued in pass 0 against that capacity, the kernels do nothing if the total exceeds it, and the total comes back with
    // the results -- one synchronisation per call; a miss falls through to the exact pass 1.
    const size_t guess = std::getenv("SVGPU_MATCH_EXACT_SIZING") ? 0 : ctx->cand_cap_hint;
    for (int pass = 0; pass < 2; ++pass) {
        const bool speculative = !pass && guess > 0;
        const size_t cap = pass ? (size_t)total : guess;
        const size_t need = need1 + ((pass || speculative) ? 2 * pad(cap * 4) : 0);
        const bool regrow = need > ctx->scratch_bytes;  // pass 1 without regrowth: the arena of pass 0 is still valid, same layout
        int rc = sv_ensure_scratch(ctx, need);
        if (rc) return rc;
        const bool fresh = !pass || regrow;
        Arena A(ctx->d_scratch);
        rc = sv_ensure_stage(ctx, need);  // page-locked mirror of the arena: batched uploads (fresh passes) and read-backs
        if (rc) return rc;
        A.mirror = ctx->h_stage;
        CandProblem P{};
        GridProblem G{};
#define UP(dst, T, src, n)                                                                          \
    T* dst = nullptr;                                                                               \
    if (src) {                                                                                      \
        dst = A.take<T>(n);                                                                         \
        if (fresh && (rc = A.upload(ctx, s, dst, src, (size_t)(n) * sizeof(T)))) return rc; \
    }
#define UPR(dst, T, src, n)                       \
    T* dst = nullptr;                             \
    if (F.res) dst = const_cast<T*>(src);         \
    else if (src) {                               \
        dst = A.take<T>(n);                       \
        if (fresh && (rc = A.upload(ctx, s, dst, src, (size_t)(n) * sizeof(T)))) return rc; \
    }
        UPR(d_t, uint8_t, F.tdesc, (size_t)nt * 32)
        UPR(d_txy, float, F.t_xy, (size_t)nt * 2)
        UPR(d_toct, int32_t, F.t_octave, nt)
        UP(d_occ, uint8_t, F.occupied, nt)
        UPR(d_ta, float, F.t_angle, nt)
        UPR(d_tx, float, F.t_xright, nt)
#undef UP
#undef UPR
        lap(pass ? "target side (pass 1)" : "target side uploads");
        rc = stage(A, fresh, P, G);
        if (rc) return rc;
        if ((rc = A.flush(ctx, s))) return rc;
        lap(pass ? "stage (pass 1)" : "stage: query uploads");
        G.t_xy = d_txy;
        G.t_octave = d_toct;
        G.nt = nt;
        G.min_x = F.min_x;
        G.min_y = F.min_y;
        G.inv_w = (double)F.grid_cols / (F.max_x - F.min_x);  // float difference, double quotient: data/common.cc:86-87 via camera::base
        G.inv_h = (double)F.grid_rows / (F.max_y - F.min_y);
        G.cols = F.grid_cols;
        G.rows = F.grid_rows;
        if (F.res) {  // binned when the frame was created
            G.cell_of = F.res->cell_of;
            G.cell_off = F.res->cell_off;
            G.cell_items = F.res->cell_items;
        }
        else {
            G.cell_of = A.take<int32_t>(nt);
            G.cell_off = A.take<int32_t>(ncell + 1);
            G.cell_items = A.take<int32_t>(nt);
        }
        G.nq = nq;
        G.cand_off = A.take<int32_t>(nq + 1);
        P.match_q = A.take<int32_t>(nq);
        P.num = A.take<int32_t>(1);
        int* owner = A.take<int>(nt);
        int* match = A.take<int>(nq);
        unsigned* mdist = A.take<unsigned>(nt);
        if (fresh) {
            if (F.res) sv_launch_grid_queries(s, G);
            else sv_launch_grid_build(s, G);
        }
        if (!pass && !speculative) {
            SV_HIP(ctx, hipMemcpyAsync(&total, G.cand_off + nq, 4, hipMemcpyDeviceToHost, s));
            SV_HIP(ctx, hipStreamSynchronize(s));
            lap("grid + total read-back");
            if (mtrace) std::fprintf(stderr, "[match] nq %d nt %d candidates %d (%.1f per query)\n", nq, nt, total, nq ? (double)total / nq : 0.0);
            if (total == 0) {
                Downloads D;
                rc = finish(P, A, D);
                if (!rc) rc = D.fetch(ctx, s, A);
                if (rc) return rc;
                SV_HIP(ctx, hipStreamSynchronize(s));
                D.scatter(A);
                return SVGPU_OK;
            }
            continue;
        }
        G.cand_idx = A.take<int32_t>(cap);
        P.dist = A.take<uint32_t>(cap);
        G.cap = P.cap = speculative ? (int)cap : 0;
        if (A.off > ctx->scratch_bytes) return sv_set_error(ctx, SVGPU_ERR_INVALID, "cell matcher: internal arena overflow");
        sv_launch_grid_fill(s, G);
        P.tdesc = (const uint32_t*)d_t;
        P.t_octave = d_toct;
        P.nq = nq;
        P.nt = nt;
        P.cand_off = G.cand_off;
        P.cand_idx = G.cand_idx;
        P.cand_skip = nullptr;
        P.occupied = d_occ;
        P.t_angle = d_ta;
        P.check_orientation = check_orientation;
        P.t_xright = P.q_xright ? d_tx : nullptr;
        P.t_xy = d_txy;
        P.chi_t_xright = P.chi_gate ? d_tx : nullptr;
        P.thr = thr;
        P.lowe_ratio = lowe_ratio;
        P.mode = mode;
        sv_launch_cand(ctx, s, P, owner, match, mdist);
        SV_HIP(ctx, hipGetLastError());
        int32_t num = 0, total_dev = 0;
        Downloads D;
        D.add(A, match_q, P.match_q, (size_t)nq * 4);
        D.add(A, &num, P.num, 4);
        if (speculative) D.add(A, &total_dev, G.cand_off + nq, 4);
        rc = finish(P, A, D);
        if (!rc) rc = D.fetch(ctx, s, A);
        if (rc) return rc;
        lap("lists + matcher enqueued");
        SV_HIP(ctx, hipStreamSynchronize(s));
        D.scatter(A);
        lap("final sync");
        if (speculative) {
            total = total_dev;
            if ((size_t)total > cap) {  // the guess was too small: nothing was written; size exactly
                ctx->cand_cap_hint = (size_t)total + (size_t)total / 4 + 4096;
                continue;
            }
        }
        ctx->cand_cap_hint = std::max(ctx->cand_cap_hint, (size_t)total + (size_t)total / 4 + 4096);
        *num_matches = num;
        return SVGPU_OK;
    }
    return SVGPU_OK;
}


}  // namespace svm
