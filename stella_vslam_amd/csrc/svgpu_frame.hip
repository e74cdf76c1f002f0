// Device-resident frame observations ("removes the D2H / H2D round trip"): data::frame_observation
// (data/frame_observation.h:12-38) of a frame or keyframe kept on the device -- descriptors, undistorted keypoints, stereo x_right,
// bearings and the keypoint grid of data::assign_keypoints_to_grid (data/common.cc:83-108) -- so that the matchers a tracked frame goes
// through (tracking_module.cc:533-608: match_current_and_last_frames, match_frame_and_landmarks, ...) read it where the extractor left it.
//   svgpu_frame_adopt_extraction   system.cc:380-395: the keypoints / descriptors of the context's last svgpu_orb_extract, never re-uploaded
//   svgpu_frame_upload             an observation that exists on the host (keyframes of the map database)
//   svgpu_frame_bind               the NEXT matcher call of a context takes its keypoint side from the frame
#include "frame_kernels.h"
#include "match_kernels.h"

namespace {
inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

__global__ void k_frame_split(const svgpu_keypoint* __restrict__ k, int n, float* __restrict__ xy, int32_t* __restrict__ octave, float* __restrict__ angle) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const svgpu_keypoint p = k[i];
    if (xy) xy[2 * i] = p.x, xy[2 * i + 1] = p.y;
    octave[i] = p.octave;
    angle[i] = p.angle;
}

}  // namespace
// (also used by the tracked-frame chain, svgpu_track.hip)
int sv_frame_reserve(svgpu_ctx* ctx, svgpu_frame* f, int n, int ncell) {
    if (n > f->cap) {
        const int cap = n + n / 4 + 64;
        if (f->slab) SV_HIP(ctx, hipFree(f->slab));
        f->slab = nullptr;
        f->cap = 0;
        size_t off = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = off;
            off += pad256(bytes);
            return o;
        };
        const size_t o_kps = carve((size_t)cap * sizeof(svgpu_keypoint)), o_desc = carve((size_t)cap * 32), o_und = carve((size_t)cap * sizeof(svgpu_keypoint)),
                     o_brg = carve((size_t)cap * 24), o_xr = carve((size_t)cap * 4), o_dep = carve((size_t)cap * 4), o_xy = carve((size_t)cap * 8),
                     o_oct = carve((size_t)cap * 4), o_ang = carve((size_t)cap * 4), o_cof = carve((size_t)cap * 4), o_cit = carve((size_t)cap * 4), o_cnt = carve((1 + SV_MAX_LEVELS) * 4);
        SV_HIP(ctx, hipMalloc((void**)&f->slab, off));
        f->slab_bytes = off;
        f->kps_raw = (svgpu_keypoint*)(f->slab + o_kps);
        f->desc = (uint8_t*)(f->slab + o_desc);
        f->undist = (svgpu_keypoint*)(f->slab + o_und);
        f->bearings = (double*)(f->slab + o_brg);
        f->xy = (float*)(f->slab + o_xy);
        f->octave = (int32_t*)(f->slab + o_oct);
        f->angle = (float*)(f->slab + o_ang);
        f->xright = (float*)(f->slab + o_xr);
        f->depth = (float*)(f->slab + o_dep);
        f->cell_of = (int32_t*)(f->slab + o_cof);
        f->cell_items = (int32_t*)(f->slab + o_cit);
        f->counts = (int32_t*)(f->slab + o_cnt);
        f->cap = cap;
    }
    if (ncell + 1 > f->cells_cap) {
        if (f->cell_off) SV_HIP(ctx, hipFree(f->cell_off));
        f->cell_off = nullptr;
        f->cells_cap = 0;
        SV_HIP(ctx, hipMalloc((void**)&f->cell_off, (size_t)(ncell + 1) * 4));
        f->cells_cap = ncell + 1;
    }
    if (!f->dummy) SV_HIP(ctx, hipMalloc((void**)&f->dummy, 256));
    return SVGPU_OK;
}
namespace {
inline int frame_reserve(svgpu_ctx* ctx, svgpu_frame* f, int n, int ncell) { return sv_frame_reserve(ctx, f, n, ncell); }

// bins the frame's undistorted keypoints (f->xy, f->octave already in place) over the camera's image bounds
void frame_bin(hipStream_t s, svgpu_frame* f, const svgpu_camera* cam, int grid_cols, int grid_rows) {
    f->grid_cols = grid_cols, f->grid_rows = grid_rows;
    f->min_x = cam->min_x, f->max_x = cam->max_x, f->min_y = cam->min_y, f->max_y = cam->max_y;
    GridProblem G{};
    G.t_xy = f->xy;
    G.t_octave = f->octave;
    G.nt = f->n;
    G.min_x = cam->min_x;
    G.min_y = cam->min_y;
    G.inv_w = (double)grid_cols / (cam->max_x - cam->min_x);  // float difference, double quotient: data/common.cc:86-87 via camera::base
    G.inv_h = (double)grid_rows / (cam->max_y - cam->min_y);
    G.cols = grid_cols;
    G.rows = grid_rows;
    G.cell_of = f->cell_of;
    G.cell_off = f->cell_off;
    G.cell_items = f->cell_items;
    G.nq = 0;
    G.cand_off = f->dummy;
    sv_launch_grid_frame(s, G);
}
bool grid_args_ok(const svgpu_camera* cam, int grid_cols, int grid_rows) {
    return cam && cam->model >= SVGPU_CAM_PERSPECTIVE && cam->model <= SVGPU_CAM_RADIAL_DIVISION && grid_cols >= 1 && grid_rows >= 1
           && (size_t)grid_cols * grid_rows <= (size_t(1) << 22) && cam->min_x < cam->max_x && cam->min_y < cam->max_y;
}
}  // namespace

extern "C" {

int svgpu_frame_create(svgpu_ctx* ctx, svgpu_frame** out) {
    if (!ctx || !out) return SVGPU_ERR_INVALID;
    *out = new (std::nothrow) svgpu_frame();
    if (!*out) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_create: out of memory");
    (*out)->device = ctx->device;
    return SVGPU_OK;
}

void svgpu_frame_destroy(svgpu_frame* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    void* p[] = {f->slab, f->cell_off, f->dummy};
    for (void* q : p)
        if (q) (void)hipFree(q);
    delete f;
}

int svgpu_frame_size(const svgpu_frame* f) { return f ? f->n : -1; }

int svgpu_frame_adopt_extraction(svgpu_ctx* ctx, svgpu_frame* f, const svgpu_camera* cam, int grid_cols, int grid_rows, svgpu_keypoint* undist_kps,
                                 double* bearings) {
    if (!ctx || !f || !grid_args_ok(cam, grid_cols, grid_rows)) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_adopt_extraction: bad arguments");
    if (ctx->last_extract_n < 0) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_frame_adopt_extraction: no svgpu_orb_extract on this context yet");
    if (f->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_adopt_extraction: the frame lives on another device");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n = ctx->last_extract_n;
    int rc = frame_reserve(ctx, f, n, grid_cols * grid_rows);
    if (rc) return rc;
    f->n = n;
    f->has_xright = false;
    if (n > 0) {
        SV_HIP(ctx, hipMemcpyAsync(f->desc, ctx->d_desc, (size_t)n * 32, hipMemcpyDeviceToDevice, s));
        FrameObsProblem P{};
        P.cam = *cam;
        P.kps = ctx->d_kps;
        P.n = n;
        P.undist = f->undist;
        P.undist_xy = f->xy;
        P.bearings = f->bearings;
        sv_launch_frame_observation(s, P);
        hipLaunchKernelGGL(k_frame_split, dim3((n + 255) / 256), dim3(256), 0, s, f->undist, n, (float*)nullptr, f->octave, f->angle);
    }
    frame_bin(s, f, cam, grid_cols, grid_rows);
    SV_HIP(ctx, hipGetLastError());
    if (n > 0 && undist_kps) SV_HIP(ctx, hipMemcpyAsync(undist_kps, f->undist, (size_t)n * sizeof(svgpu_keypoint), hipMemcpyDeviceToHost, s));
    if (n > 0 && bearings) SV_HIP(ctx, hipMemcpyAsync(bearings, f->bearings, (size_t)n * 24, hipMemcpyDeviceToHost, s));
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_frame_upload(svgpu_ctx* ctx, svgpu_frame* f, const svgpu_camera* cam, const svgpu_keypoint* undist_kps, const uint8_t* desc, const float* x_right,
                       int n, int grid_cols, int grid_rows) {
    if (!ctx || !f || n < 0 || !grid_args_ok(cam, grid_cols, grid_rows) || (n > 0 && (!undist_kps || !desc)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_upload: bad arguments");
    if (f->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_upload: the frame lives on another device");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc = frame_reserve(ctx, f, n, grid_cols * grid_rows);
    if (rc) return rc;
    f->n = n;
    f->has_xright = x_right != nullptr;
    if (n > 0) {
        SV_HIP(ctx, hipMemcpyAsync(f->desc, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
        SV_HIP(ctx, hipMemcpyAsync(f->undist, undist_kps, (size_t)n * sizeof(svgpu_keypoint), hipMemcpyHostToDevice, s));
        if (x_right) SV_HIP(ctx, hipMemcpyAsync(f->xright, x_right, (size_t)n * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_frame_split, dim3((n + 255) / 256), dim3(256), 0, s, f->undist, n, f->xy, f->octave, f->angle);
    }
    frame_bin(s, f, cam, grid_cols, grid_rows);
    SV_HIP(ctx, hipGetLastError());
    SV_HIP(ctx, hipStreamSynchronize(s));
    return SVGPU_OK;
}

int svgpu_frame_set_stereo(svgpu_ctx* ctx, svgpu_frame* f, const float* x_right) {
    if (!ctx || !f) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_set_stereo: bad arguments");
    f->has_xright = x_right != nullptr;
    if (!x_right || f->n == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    SV_HIP(ctx, hipMemcpyAsync(f->xright, x_right, (size_t)f->n * 4, hipMemcpyHostToDevice, ctx->stream));
    SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVGPU_OK;
}

int svgpu_frame_bind(svgpu_ctx* ctx, const svgpu_frame* f) {
    if (!ctx) return SVGPU_ERR_INVALID;
    if (f && f->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_frame_bind: the frame lives on another device");
    ctx->bound_frame = f;
    return SVGPU_OK;
}

int svgpu_match_set_query_blocks(svgpu_ctx* ctx, const uint8_t* q_blocks) {
    if (!ctx) return SVGPU_ERR_INVALID;
    ctx->next_q_blocks = q_blocks;
    return SVGPU_OK;
}

}  // extern "C"
