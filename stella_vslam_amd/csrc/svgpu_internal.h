// Internal declarations shared by the svgpu translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "svgpu.h"

#define SV_MAX_LEVELS 16
#define SV_PATCH_RADIUS 19  // orb_extractor.h:107 orb_patch_radius_
#define SV_CELL 64          // orb_extractor.cc:173 cell_size
#define SV_OVERLAP 6        // orb_extractor.cc:172 overlap
#define SV_ROI_MAX 70       // SV_CELL + SV_OVERLAP
// k_blur tiling (shared by the kernel and the host-side tile count): a thread walks BLUR_ROWS rows of 4 columns
#ifndef BLUR_ROWS
#define BLUR_ROWS 64
#endif
#define BLUR_ROWS_SMALL 16          // contexts configured for at most BLUR_SMALL_BATCH frames: latency, not halo traffic, is what counts
#define BLUR_SMALL_BATCH 4
#define BLUR_TW 256                 // tile width  = 64 threads x 4 px; tile height = 4 strips of OrbConfig::blur_rows rows

// ---- per-level geometry, read by every ORB kernel (lives in device memory, one array per context)
struct OrbLevel {
    int w, h;               // level size in pixels
    int pitch;              // row pitch (bytes) of the stored pyramid / blurred level
    int has_cells;          // 0 if the level is too small for a 19-px border
    long long pyr_off;      // byte offset of this level inside one frame's pyramid block (levels >= 1)
    long long blur_off;     // byte offset inside one frame's blurred block (all levels)
    int xtab_off, ytab_off; // resize coefficient tables (level produced from level-1)
    int xg_off;             // first 32-byte column-group record of this level in the packed table of k_pyramid_lds
    int cell_first, cell_count;     // FAST cells of this level in the cell table
    int cells_x;                    // num_cols of the FAST cell lattice (orb_extractor.cc:186)
    int grid_x, grid_y, grid_first; // selection grid (distribute_keypoints) and its offset in the key array
    int gtab_x_off, gtab_y_off;     // region coordinate -> grid index lookup tables
    int btile_first, btiles_x, btiles_y;  // blur tiles
    float scale;            // scale_factors_[level]
    float kp_size;          // (float)(unsigned)(31 * scale)
};

struct FastCell {
    short min_x, min_y;  // ROI origin in level coordinates
    short w, h;          // ROI size (<= 70)
    short lv, cj;        // pyramid level; cell column (j in orb_extractor.cc:199-217)
    int order_base;      // (ci * num_cols + cj) << 14 : emission order prefix
};

// k_describe_bands: a band = consecutive rows of one level's selection grid whose pixels one workgroup stages in LDS (orb_kernels.hip)
struct DescBand {
    int cell0, cell1;  // selection-grid cells [cell0, cell1) (indices into the per-frame key / position arrays)
    int img_bytes;     // LDS bytes of the staged rows (the larger of the two phases), a multiple of 16
    short lv, lp;      // level; LDS row pitch (multiple of 16, = 32 mod 64)
    short yu0, nru;    // un-blurred rows [yu0, yu0 + nru)  (every keypoint's y - 15 .. y + 16)
    short yb0, nrb;    // blurred rows    [yb0, yb0 + nrb)  (every keypoint's y - 18 .. y + 18)
};

struct OrbConfig {
    int width = 0, height = 0, max_batch = 0, num_levels = 0;
    float scale_factor = 0;
    int ini_thr = 0, min_thr = 0;
    unsigned min_area_sqrt = 0;
    float scale_factors[SV_MAX_LEVELS];
    OrbLevel levels[SV_MAX_LEVELS];
    std::vector<FastCell> cells;
    int total_grid = 0;    // sum of grid cells over levels = max keypoints per frame
    int total_btiles = 0;
    int blur_rows = BLUR_ROWS;  // rows per k_blur thread (BLUR_ROWS, or BLUR_ROWS_SMALL for a context of a few frames)
    size_t pyr_frame_bytes = 0, blur_frame_bytes = 0;
    std::vector<DescBand> dbands;  // empty: the configuration does not fit the band kernel, k_describe takes it
    size_t dband_lds_bytes = 0;    // dynamic LDS of k_describe_bands
    bool configured = false;
};

// Optional per-kernel timing with HIP events on the launch stream (svgpu_profile_select / _read / _read_class).
// One accumulator per kernel class; the selection "*" brackets EVERY class, so that one timed region yields the mean launch time of
// each kernel as it runs inside the caller's pipeline (bench.py: roofline.kernels[] comes from the timed region itself).
struct SvProfClass {
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t used = 0;
    double total_ms = 0;
    long long launches = 0;
};
struct SvProf {
    std::string name;  // kernel class being bracketed; "*" = all; empty = off
    std::map<std::string, SvProfClass> cls;
    unsigned long long* d_counter = nullptr;  // device word a profiled k_bf_mfma adds its multiplied 64 x 32 patches to
};

// A frame observation resident on the device (include/svgpu.h svgpu_frame_*): what every projection-family matcher reads of a
// data::frame / data::keyframe -- descriptors, undistorted keypoints split into the arrays the kernels take, stereo x_right, and the
// keypoint grid of data::assign_keypoints_to_grid as CSR.  Plain device allocations (grow-only), usable from any context of the device.
struct svgpu_frame {
    int device = 0;
    int n = 0, cap = 0;
    int grid_cols = 0, grid_rows = 0, cells_cap = 0;
    float min_x = 0, max_x = 0, min_y = 0, max_y = 0;  // image bounds the grid was binned over
    bool has_xright = false;
    // ONE device allocation (`slab`, `slab_bytes`), carved in this order, so that what the host wants back of a freshly extracted frame
    // (kps_raw | desc | undist | bearings [| xright | depth for a stereo pair]) is one contiguous copy (svgpu_track_motion)
    char* slab = nullptr;
    size_t slab_bytes = 0;
    svgpu_keypoint* kps_raw = nullptr;  // n distorted keypoints as the extractor wrote them (fused extraction only)
    uint8_t* desc = nullptr;            // n x 32
    svgpu_keypoint* undist = nullptr;   // n undistorted keypoint records
    double* bearings = nullptr;         // n x 3
    float* xy = nullptr;                // n x 2
    int32_t* octave = nullptr;
    float* angle = nullptr;
    float* xright = nullptr;            // stereo_x_right_ (valid when has_xright)
    float* depth = nullptr;             // depths_ (written by the fused stereo chain only: svgpu_track_motion_stereo)
    int32_t* cell_of = nullptr;         // n
    int32_t* cell_items = nullptr;      // n
    int32_t* counts = nullptr;          // 1 + SV_MAX_LEVELS ints: the extractor's counts (fused extraction: [0] = n as the device knows it)
    int32_t* cell_off = nullptr;        // grid_cols * grid_rows + 1 (own allocation: sized by the grid)
    int32_t* dummy = nullptr;           // 1 int (the query-side scan of a grid build without queries)
};

// The landmark table of the tracked-frame chain (include/svgpu.h svgpu_map_*): records indexed by data::landmark::id_.
// Writers (upsert / erase: the mapping thread after BA, the tracking thread's flush) and readers (the tracker's chains) may sit on
// different contexts = streams: `mtx` serialises the host side, ev_write / the pending read events order the streams (a reader's stream
// waits for the last write, a writer's for EVERY read since the previous writer), and a growth waits for both before the old table is freed.
struct svgpu_map {
    int device = 0;
    int cap = 0;                         // records allocated (ids < cap are addressable)
    svgpu_landmark_record* rec = nullptr;
    std::mutex mtx;
    hipEvent_t ev_write = nullptr;
    bool wrote = false;
    // one event PER READ that no writer has waited for yet: readers run on any number of streams (two trackers of a stereo rig, a
    // relocaliser beside the tracker, svgpu_map_download on another context), and a writer -- above all a growth, which frees the old
    // allocation -- has to wait for every one of them, not for whichever recorded last
    std::vector<hipEvent_t> reads_pending, reads_pool;
};
int sv_map_reader_begin(svgpu_ctx* ctx, svgpu_map* map, hipStream_t s);  // the reader's stream waits for the last write (call with map->mtx held)
int sv_map_reader_end(svgpu_ctx* ctx, svgpu_map* map, hipStream_t s);    // records this read for the next writer (mutex still held)
// Scope of a read of the table on stream s (constructed with map->mtx held, destroyed before it is released): whatever path leaves the
// scope -- every SV_HIP early return included -- the read is recorded for the writers; a path that did not reach end() also drains the stream,
// so nothing it enqueued still reads the table when the caller sees the error.
struct SvMapReadScope {
    svgpu_ctx* ctx;
    svgpu_map* map;
    hipStream_t s;
    bool open = false;
    SvMapReadScope(svgpu_ctx* c, svgpu_map* m, hipStream_t st) : ctx(c), map(m), s(st) {}
    int begin() {
        const int rc = sv_map_reader_begin(ctx, map, s);
        open = rc == 0;
        return rc;
    }
    int end() {
        open = false;
        return sv_map_reader_end(ctx, map, s);
    }
    ~SvMapReadScope() {
        if (open) {
            (void)sv_map_reader_end(ctx, map, s);
            (void)hipStreamSynchronize(s);
        }
    }
    SvMapReadScope(const SvMapReadScope&) = delete;
    SvMapReadScope& operator=(const SvMapReadScope&) = delete;
};

struct svgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_aux = nullptr;  // blur of a batch runs here beside FAST + selection on `stream`
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};  // recorded inside svgpu_orb_extract_batch_device: [0] before k_fast, [1] before k_describe (svgpu_orb_stream_wait_stage)
    bool stage_recorded = false;
    std::string last_error;
    SvProf prof;
    OrbConfig orb;
    // device resources of the ORB path
    OrbLevel* d_levels = nullptr;
    FastCell* d_cells = nullptr;
    short* d_xofs = nullptr;        // per level >= 1: source column
    short2* d_xa = nullptr;         // (a0, a1) 11-bit coefficients
    short2* d_yofs = nullptr;       // (row0, row1) clamped
    short2* d_yb = nullptr;         // (b0, b1)
    uint32_t* d_xg = nullptr;       // k_pyramid_lds: 8 words per group of 4 output columns (see build in svgpu_orb.hip)
    short4* d_yrow = nullptr;       // k_pyramid_lds: (row0, row1, b0, b1) per output row
    int2* d_band_rows = nullptr;    // [bands][levels]: rows of each level a pyramid band computes
    int pyr_bands = 0;
    size_t pyr_lds_bytes = 0;       // dynamic LDS of k_pyramid_lds for this configuration; 0 = use the global-memory variant
    unsigned short* d_gtab = nullptr;
    uint8_t* d_pyr = nullptr;       // max_batch * pyr_frame_bytes
    uint8_t* d_blur = nullptr;      // max_batch * blur_frame_bytes
    unsigned long long* d_keys = nullptr;  // max_batch * total_grid
    int4* d_sel = nullptr;          // max_batch * total_grid  (x, y, level, response)
    int32_t* d_cellpos = nullptr;   // max_batch * (total_grid + 1): emission position of every selection-grid cell
    DescBand* d_dbands = nullptr;
    // staging for the host-buffer entry points
    uint8_t* d_img = nullptr;
    uint8_t* d_mask = nullptr;
    svgpu_keypoint* d_kps = nullptr;
    uint8_t* d_desc = nullptr;
    int32_t* d_counts = nullptr;
    int last_batch = 0;
    int last_extract_n = -1;             // keypoints of the last svgpu_orb_extract, still in d_kps / d_desc (svgpu_frame_adopt_extraction)
    const svgpu_frame* bound_frame = nullptr;  // svgpu_frame_bind: keypoint side of the NEXT matcher call (one-shot)
    size_t cand_cap_hint = 0;                  // cell matchers: candidate-list capacity that sufficed last time (0: unknown -> exact two-pass sizing)
    const uint8_t* next_q_blocks = nullptr;    // svgpu_match_set_query_blocks: per-query "an accepted match occupies its target" of the NEXT svgpu_match_in_cells
    const uint8_t* last_imgs = nullptr;  // level-0 of the last call (device)
    size_t last_frame_stride = 0;
    int last_row_stride = 0;
    // generic scratch (matchers, BA)
    void* d_scratch = nullptr;
    double* h_pinned = nullptr;     // page-locked read-back buffer of the BA loops (small per-trial partial sums)
    size_t pinned_doubles = 0;
    char* h_stage = nullptr;        // page-locked staging of the BA entry points: inputs are written here in device layout, one copy each way
    size_t stage_bytes = 0;
    size_t scratch_bytes = 0;
    // bundle adjustment
    hipStream_t ba_copy_stream = nullptr;  // second stream of a global-BA sized call: the measurements' upload runs beside the structure kernels
    hipEvent_t ev_ba_copy = nullptr;
    hipEvent_t ev_ba = nullptr;     // completion event the BA host loop polls (while mirroring the caller's stop flag)
    int ba_solver = 0;              // svgpu_ba_solver
    double pcg_tol = 1e-10;         // relative residual of the reduced-system PCG
    int pcg_max_it = 0;             // 0 = max(2000, 4 n)
    void* comm = nullptr;           // ncclComm_t of svgpu_comm_init (RCCL, loaded with dlopen)
    void* ba_sky = nullptr;         // plan + buffers of the envelope Cholesky (ba_skyline.hip)
    svgpu_allreduce_fn ba_ar_fn = nullptr;  // the all-reduce of the sharded solve in progress (the segmented envelope solve exchanges through it); else null
    void* ba_ar_user = nullptr;
    long long ba_xch[10] = {0};    // svgpu_ba_last_exchange: what the last sharded solve all-reduced (mode | bytes by category | calls | trials | linearisations)
    int comm_rank = 0, comm_world = 1;
};
void sv_comm_release(svgpu_ctx* ctx);

int sv_set_error(svgpu_ctx* ctx, int status, const char* what, hipError_t e = hipSuccess);
void sv_prof_begin(svgpu_ctx* ctx, hipStream_t s, const char* name);
void sv_prof_end(svgpu_ctx* ctx, hipStream_t s, const char* name);
unsigned long long* sv_prof_counter(svgpu_ctx* ctx, const char* name);  // the device counter when `name` is being profiled, else null
struct SvProfScope {  // brackets the launches issued inside its lifetime when `name` is the selected kernel class
    svgpu_ctx* c;
    hipStream_t s;
    const char* n;
    SvProfScope(svgpu_ctx* ctx, hipStream_t st, const char* name) : c(ctx), s(st), n(name) {
        if (!c->prof.name.empty()) sv_prof_begin(c, s, n);
    }
    ~SvProfScope() {
        if (!c->prof.name.empty()) sv_prof_end(c, s, n);
    }
};
int sv_ensure_scratch(svgpu_ctx* ctx, size_t bytes);
void sv_sky_release(svgpu_ctx* ctx);  // plan + buffers of the envelope Cholesky (ba_skyline.hip)
int sv_ensure_stage(svgpu_ctx* ctx, size_t bytes);  // grow-only page-locked host buffer (ctx->h_stage)
hipError_t sv_allow_dynamic_lds(const void* kernel, size_t bytes);  // per (device, kernel), thread-safe
void sv_orb_release(svgpu_ctx* ctx);
int sv_frame_reserve(svgpu_ctx* ctx, svgpu_frame* f, int n, int ncell);  // grow-only slab of a resident frame (svgpu_frame.hip)

#define SV_HIP(ctx, call)                                                   \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return sv_set_error((ctx), SVGPU_ERR_HIP, #call, _e); \
    } while (0)

// ---- kernel launchers (orb_kernels.hip)
void sv_launch_resize(hipStream_t s, const uint8_t* src, size_t src_frame_stride, int src_pitch, int sw, int sh,
                      uint8_t* dst, size_t dst_frame_stride, int dst_pitch, int dw, int dh, const short* xofs,
                      const short2* xa, const short2* yofs, const short2* yb, int batch);
#define SV_PYR_LDS_MAX (156 * 1024)  // dynamic LDS budget of k_pyramid_lds (160 KB per CU minus its static tables)
#define SV_PYR_LDS_HALF (78 * 1024)  // ... of which two fit a CU
hipError_t sv_pyramid_prepare();
void sv_launch_pyramid(hipStream_t s, const OrbLevel* levels, int num_levels, const int2* band_rows, int bands, const uint8_t* img0,
                       size_t img0_frame_stride, int img0_pitch, uint8_t* pyr, size_t pyr_frame_bytes, const short* xofs,
                       const short2* xa, const short2* yofs, const short2* yb, const uint32_t* xg, const short4* yrow, int batch, size_t lds_bytes);
void sv_launch_blur(hipStream_t s, const OrbLevel* levels, int num_levels, int total_tiles, const uint8_t* img0,
                    size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr, size_t pyr_frame_bytes, uint8_t* blur,
                    size_t blur_frame_bytes, int batch, bool need_gather, int rows);
void sv_launch_fast(hipStream_t s, const OrbLevel* levels, int num_levels, const FastCell* cells, int num_cells,
                    const uint8_t* img0, size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr,
                    size_t pyr_frame_bytes, const unsigned short* gtab, unsigned long long* keys, int total_grid,
                    int ini_thr, int min_thr, const uint8_t* mask, size_t mask_frame_stride, int mask_pitch, int mask_w,
                    int mask_h, int batch);
void sv_launch_select(hipStream_t s, const OrbLevel* levels, int num_levels, unsigned long long* keys, int total_grid,
                      int4* sel, int32_t* counts, int32_t* cellpos, int batch);
hipError_t sv_describe_bands_prepare(size_t lds_bytes);
void sv_launch_describe_bands(hipStream_t s, const OrbLevel* levels, int num_levels, const DescBand* bands, int num_bands, size_t lds_bytes,
                              const int4* sel, int total_grid, const int32_t* cellpos, const int32_t* counts, const uint8_t* img0,
                              size_t img0_frame_stride, int img0_pitch, const uint8_t* pyr, size_t pyr_frame_bytes, const uint8_t* blur,
                              size_t blur_frame_bytes, svgpu_keypoint* kps, uint8_t* desc, int cap, int batch, float* angles);
void sv_launch_describe(hipStream_t s, const OrbLevel* levels, int num_levels, const int4* sel, int total_grid,
                        const int32_t* counts, const uint8_t* img0, size_t img0_frame_stride, int img0_pitch,
                        const uint8_t* pyr, size_t pyr_frame_bytes, const uint8_t* blur, size_t blur_frame_bytes,
                        svgpu_keypoint* kps, uint8_t* desc, int cap, int batch, float* angles);
