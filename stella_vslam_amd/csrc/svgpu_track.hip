// The tracked-frame chain (include/svgpu.h svgpu_tracker_* / svgpu_track_*): what tracking_module does per image
// (tracking_module.cc:253-275, 333-355, 533-608; module/frame_tracker.cc:22-60) as TWO submissions of a few launches each on one stream,
// every one followed by ONE synchronisation:
//   svgpu_track_motion      [H2D image + ids] -> (k_pyramid, k_blur, k_fast, k_select, k_describe -> k_track_frame) -> k_track_cand<0>
//                           -> k_cand_replay_lds -> k_pose_opt<.., true> [-> D2H of the new observation]
//   svgpu_track_local_map   [H2D ids] -> k_track_cand<1> -> k_cand_replay_lds -> k_pose_opt<.., true>
// Landmarks are read from the resident table (svgpu_map) by id; the pose the first optimisation finds stays on the device for the second
// half; matches, outlier flags, pose and counts are written by the kernels straight into page-locked host memory, so the only copies are
// the inputs and the freshly extracted observation.  Between the halves the host runs update_local_map (object graph: the reference's).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "svgpu_internal.h"
#include "ba_kernels.h"
#include "track_kernels.h"

struct svgpu_tracker {
    svgpu_ctx* ctx = nullptr;
    svgpu_map* map = nullptr;
    int map_device = 0;
    svgpu_camera cam{};
    svgpu_track_config cfg{};
    // capacities
    int cap_kp = 0;       // keypoints of the current frame
    int cap_q = 0;        // queries (last frame's keypoints / local landmarks)
    size_t cap_cand = 0;  // candidate-list entries BEHIND the per-query slots (lists longer than TRACK_SLOT)
    size_t img_bytes = 0; // image part of the input block
    // device
    char* d_in = nullptr;    // input block: [image] | ids ...
    size_t in_bytes = 0;
    char* d_work = nullptr;  // everything the kernels hand to each other
    size_t work_bytes = 0;
    double* d_pose = nullptr;  // 12: the pose of the last optimisation
    // page-locked host
    char* h_in = nullptr;
    size_t h_in_bytes = 0;
    char* h_out = nullptr;   // small results written by the kernels + the read-back of a fresh observation
    size_t h_out_bytes = 0;
    // layout of d_work (set by reserve)
    int32_t *cand_off = nullptr, *cand_cnt = nullptr, *match_q = nullptr, *num = nullptr, *who = nullptr, *kp_of = nullptr, *pred_level = nullptr;
    uint32_t* dist = nullptr;
    uint8_t *q_valid = nullptr, *q_blocks = nullptr, *occupied = nullptr, *visible = nullptr, *outlier_kp = nullptr, *po_outlier = nullptr, *po_level = nullptr,
            *po_robust = nullptr;
    double *reproj = nullptr, *po_pos = nullptr;
    float *x_right = nullptr, *po_uvr = nullptr, *po_w = nullptr, *po_h = nullptr;
    int* po_result = nullptr;
    int32_t* cur_lm_motion = nullptr;  // chain 1's landmark ids per keypoint (chain 2's come with its input block)
    int *g_owner = nullptr, *g_match = nullptr;  // tables of the replay's global-memory form (inputs beyond its LDS form)
    // layout of h_out
    int32_t *h_n = nullptr, *h_num = nullptr, *h_match = nullptr;
    int* h_result = nullptr;
    double* h_pose = nullptr;
    uint8_t *h_outlier = nullptr, *h_visible = nullptr;
    unsigned long long* h_stamps = nullptr;  // debug (SVGPU_TRACK_STAMPS): phase stamps of the last k_pose_opt
    char* h_obs = nullptr;  // read-back of the fresh observation (prefix of the frame's slab)
    size_t h_obs_bytes = 0;
    int last_n_local = -1;
    int obs_n = 0;          // the observation in h_obs (svgpu_tracker_observation)
    size_t obs_off[6] = {0, 0, 0, 0, 0, 0};  // kps | desc | undist | bearings | x_right | depth inside h_obs
    bool obs_stereo = false;
    long long launches = 0, syncs = 0;
    // the RIGHT camera of a stereo rig (svgpu_track_motion_stereo): its image goes through its own context and stream beside the left one's
    svgpu_ctx* right_ctx = nullptr;
    char* d_in_r = nullptr;
    char* h_in_r = nullptr;
    size_t in_r_bytes = 0;
    svgpu_keypoint* r_kps = nullptr;
    uint8_t* r_desc = nullptr;
    int32_t* r_counts = nullptr;
    int r_cap = 0;
    hipEvent_t ev_right = nullptr;
};

namespace {
inline size_t pad256(size_t b) { return (b + 255) & ~size_t(255); }

void release_right(svgpu_tracker* t) {
    if (t->d_in_r) (void)hipFree(t->d_in_r);
    if (t->h_in_r) (void)hipHostFree(t->h_in_r);
    if (t->r_kps) (void)hipFree(t->r_kps);
    if (t->r_desc) (void)hipFree(t->r_desc);
    if (t->r_counts) (void)hipFree(t->r_counts);
    if (t->ev_right) (void)hipEventDestroy(t->ev_right);
    t->d_in_r = t->h_in_r = nullptr;
    t->r_kps = nullptr, t->r_desc = nullptr, t->r_counts = nullptr, t->ev_right = nullptr;
    t->in_r_bytes = 0, t->r_cap = 0;
}
// buffers of the right image's extraction (grow-only)
int reserve_right(svgpu_tracker* t, size_t img_bytes, int cap) {
    svgpu_ctx* ctx = t->ctx;
    if (img_bytes <= t->in_r_bytes && cap <= t->r_cap && t->ev_right) return SVGPU_OK;
    release_right(t);
    SV_HIP(ctx, hipMalloc((void**)&t->d_in_r, pad256(img_bytes)));
    SV_HIP(ctx, hipHostMalloc((void**)&t->h_in_r, pad256(img_bytes), hipHostMallocDefault));
    SV_HIP(ctx, hipMalloc((void**)&t->r_kps, (size_t)cap * sizeof(svgpu_keypoint)));
    SV_HIP(ctx, hipMalloc((void**)&t->r_desc, (size_t)cap * 32));
    SV_HIP(ctx, hipMalloc((void**)&t->r_counts, (1 + SV_MAX_LEVELS) * 4));
    SV_HIP(ctx, hipEventCreateWithFlags(&t->ev_right, hipEventDisableTiming));
    t->in_r_bytes = pad256(img_bytes), t->r_cap = cap;
    return SVGPU_OK;
}

void release(svgpu_tracker* t) {
    if (t->d_in) (void)hipFree(t->d_in);
    if (t->d_work) (void)hipFree(t->d_work);
    if (t->h_in) (void)hipHostFree(t->h_in);
    if (t->h_out) (void)hipHostFree(t->h_out);
    t->d_in = t->d_work = t->h_in = t->h_out = nullptr;
    t->in_bytes = t->work_bytes = t->h_in_bytes = t->h_out_bytes = 0;
}

// (re)allocates the tracker's buffers for the given capacities (grow-only; the tracker is idle between its synchronous calls)
int reserve(svgpu_tracker* t, int kp, int q, size_t cand, size_t img_bytes, size_t obs_bytes) {
    svgpu_ctx* ctx = t->ctx;
    const bool grow = kp > t->cap_kp || q > t->cap_q || cand > t->cap_cand || img_bytes > t->img_bytes || obs_bytes > t->h_obs_bytes;
    if (!grow && t->d_work) return SVGPU_OK;
    const int ckp = std::max(std::max(kp, t->cap_kp), 64), cq = std::max(std::max(q + q / 4, t->cap_q), 64);
    const size_t cc = std::max(std::max(cand, t->cap_cand), (size_t)65536), ib = std::max(img_bytes, t->img_bytes), ob = std::max(obs_bytes, t->h_obs_bytes);
    double pose_keep[12] = {0};
    const bool had_pose = t->d_work != nullptr;
    if (had_pose) SV_HIP(ctx, hipMemcpy(pose_keep, t->d_pose, sizeof pose_keep, hipMemcpyDeviceToHost));  // the last optimisation's pose survives a growth
    std::vector<char> obs_keep;  // ... and so does the observation svgpu_tracker_observation hands out (a growth between an extraction and its matcher's re-run)
    if (t->obs_n > 0 && t->h_obs) obs_keep.assign(t->h_obs, t->h_obs + t->h_obs_bytes);
    release(t);
    // input block: image | ids of the frame's keypoints (ckp) | ids of the queries (cq) | pose
    t->in_bytes = pad256(ib) + pad256((size_t)ckp * 4) + pad256((size_t)cq * 4) + 256;
    SV_HIP(ctx, hipMalloc((void**)&t->d_in, t->in_bytes));
    SV_HIP(ctx, hipHostMalloc((void**)&t->h_in, t->in_bytes, hipHostMallocDefault));
    t->h_in_bytes = t->in_bytes;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += pad256(bytes);
        return o;
    };
    const size_t o_off = take((size_t)(cq + 1) * 4), o_cnt = take((size_t)cq * 4), o_mq = take((size_t)cq * 4), o_num = take(16), o_who = take((size_t)ckp * 4),
                 o_kpof = take((size_t)ckp * 4), o_pl = take((size_t)cq * 4), o_dist = take(((size_t)cq * TRACK_SLOT + cc) * 4), o_qv = take(cq), o_qb = take(cq), o_occ = take(ckp),
                 o_vis = take(cq), o_okp = take(ckp), o_poo = take(ckp), o_pol = take(ckp), o_por = take(ckp), o_rp = take((size_t)cq * 16),
                 o_pos = take((size_t)ckp * 24), o_xr = take((size_t)cq * 4), o_uvr = take((size_t)ckp * 12), o_w = take((size_t)ckp * 4), o_h = take((size_t)ckp * 4),
                 o_res = take(16), o_clm = take((size_t)ckp * 4), o_pose = take(96), o_gown = take((size_t)ckp * 4), o_gmat = take((size_t)cq * 4);
    t->work_bytes = off;
    SV_HIP(ctx, hipMalloc((void**)&t->d_work, off));
    SV_HIP(ctx, hipMemset(t->d_work, 0, off));  // (the list allocation counter starts at zero; every chain leaves it at zero again)
    char* w = t->d_work;
    t->cand_off = (int32_t*)(w + o_off), t->cand_cnt = (int32_t*)(w + o_cnt), t->match_q = (int32_t*)(w + o_mq), t->num = (int32_t*)(w + o_num);
    t->who = (int32_t*)(w + o_who), t->kp_of = (int32_t*)(w + o_kpof), t->pred_level = (int32_t*)(w + o_pl), t->dist = (uint32_t*)(w + o_dist);
    t->q_valid = (uint8_t*)(w + o_qv), t->q_blocks = (uint8_t*)(w + o_qb), t->occupied = (uint8_t*)(w + o_occ), t->visible = (uint8_t*)(w + o_vis);
    t->outlier_kp = (uint8_t*)(w + o_okp), t->po_outlier = (uint8_t*)(w + o_poo), t->po_level = (uint8_t*)(w + o_pol), t->po_robust = (uint8_t*)(w + o_por);
    t->reproj = (double*)(w + o_rp), t->po_pos = (double*)(w + o_pos), t->x_right = (float*)(w + o_xr), t->po_uvr = (float*)(w + o_uvr);
    t->po_w = (float*)(w + o_w), t->po_h = (float*)(w + o_h), t->po_result = (int*)(w + o_res), t->cur_lm_motion = (int32_t*)(w + o_clm);
    t->d_pose = (double*)(w + o_pose);
    t->g_owner = (int*)(w + o_gown), t->g_match = (int*)(w + o_gmat);
    if (had_pose) SV_HIP(ctx, hipMemcpy(t->d_pose, pose_keep, sizeof pose_keep, hipMemcpyHostToDevice));
    // host results
    off = 0;
    const size_t h_st = take(64 * 8), h_n = take(16), h_num = take(16), h_res = take(16), h_pose = take(96), h_out = take(ckp), h_match = take((size_t)cq * 4), h_vis = take(cq),
                 h_obs = take(ob);
    t->h_out_bytes = off;
    SV_HIP(ctx, hipHostMalloc((void**)&t->h_out, off, hipHostMallocDefault));
    memset(t->h_out, 0, off);
    char* h = t->h_out;
    t->h_stamps = (unsigned long long*)(h + h_st);
    t->h_n = (int32_t*)(h + h_n), t->h_num = (int32_t*)(h + h_num), t->h_result = (int*)(h + h_res), t->h_pose = (double*)(h + h_pose);
    t->h_outlier = (uint8_t*)(h + h_out), t->h_match = (int32_t*)(h + h_match), t->h_visible = (uint8_t*)(h + h_vis), t->h_obs = h + h_obs;
    t->h_obs_bytes = ob;
    t->cap_kp = ckp, t->cap_q = cq, t->cap_cand = cc, t->img_bytes = ib;
    t->last_n_local = -1;
    if (!obs_keep.empty()) memcpy(t->h_obs, obs_keep.data(), std::min(obs_keep.size(), ob));
    return SVGPU_OK;
}

void fill_reproj(const svgpu_tracker* t, ReprojProblem& R, const double* pose, float margin) {
    memset(&R, 0, sizeof R);
    R.cam = t->cam;
    if (pose) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) R.rot_cw[3 * i + j] = pose[4 * i + j];
            R.trans_cw[i] = pose[4 * i + 3];
        }
        for (int i = 0; i < 3; ++i) R.trans_wc[i] = ((-R.rot_cw[i]) * R.trans_cw[0] + (-R.rot_cw[3 + i]) * R.trans_cw[1]) + (-R.rot_cw[6 + i]) * R.trans_cw[2];
    }
    R.num_levels = (unsigned)t->cfg.num_levels;
    R.log_scale_factor = t->cfg.log_scale_factor;
    R.margin = margin;
    for (int l = 0; l < t->cfg.num_levels; ++l) R.scale_factors[l] = t->cfg.scale_factors[l];
}

void fill_cand_frame(TrackCandProblem& C, const svgpu_frame* f) {
    C.tdesc = (const uint32_t*)f->desc;
    C.t_xy = f->xy;
    C.t_octave = f->octave;
    C.t_angle = f->angle;
    C.t_xright = f->has_xright ? f->xright : nullptr;
    C.cell_off = f->cell_off;
    C.cell_items = f->cell_items;
    C.min_x = f->min_x;
    C.min_y = f->min_y;
    C.inv_w = (double)f->grid_cols / (f->max_x - f->min_x);  // float difference, double quotient: data/common.cc:86-87 via camera::base
    C.inv_h = (double)f->grid_rows / (f->max_y - f->min_y);
    C.cols = f->grid_cols;
    C.rows = f->grid_rows;
}

// matcher (lists + replay) and optimisation of one half; `nt` = keypoints of the frame (capacity when nt_dev), `cur_lm` = its landmark ids
int enqueue_match_and_optimize(svgpu_tracker* t, hipStream_t s, TrackCandProblem& C, const svgpu_frame* cur, int nt, const int32_t* nt_dev, int32_t* cur_lm,
                               int reset_cur, const int32_t* q_ids, int nq, unsigned thr, float lowe_ratio, int mode, const double* pose_by_value) {
    svgpu_ctx* ctx = t->ctx;
    C.nq = nq;
    C.q_ids = q_ids;
    C.thr = thr;
    C.lowe_ratio = lowe_ratio;
    C.map = t->map->rec;
    C.map_cap = t->map->cap;
    fill_cand_frame(C, cur);
    C.nt = nt;
    C.nt_dev = nt_dev;
    C.occupied = t->occupied;
    C.cand_off = t->cand_off;
    C.counter = t->num + 2;  // (t->num: [0] matches, [2] the list allocation counter)
    C.cand_cnt = t->cand_cnt;
    C.dist = t->dist;
    C.cap = (int)std::min<size_t>(t->cap_cand, 0x7FFFFFFF);
    C.q_valid = t->q_valid;
    C.q_blocks = t->q_blocks;
    C.visible = t->visible;
    C.reproj = t->reproj;
    C.x_right = t->x_right;
    C.pred_level = t->pred_level;
    sv_launch_track_cand(ctx, s, C);
    CandProblem P{};
    P.t_octave = cur->octave;
    P.nq = nq;
    P.nt = nt;
    P.nt_dev = nt_dev;
    P.cand_off = t->cand_off;
    P.cand_cnt = t->cand_cnt;
    P.cand_total = t->num + 2;
    P.q_valid = t->q_valid;
    P.occupied = C.cur_lm ? t->occupied : nullptr;
    P.q_blocks = t->q_blocks;
    P.thr = thr;
    P.lowe_ratio = lowe_ratio;
    P.mode = mode;
    P.dist = t->dist;
    P.match_q = t->match_q;
    P.num = t->num;
    P.cap = C.cap;
    P.match_host = t->h_match;
    P.num_host = t->h_num;
    {
        SvProfScope ps(ctx, s, "k_cand");
        sv_launch_cand_replay(ctx, s, P, t->g_owner, t->g_match);
    }
    PoseOptDev O;
    memset(&O, 0, sizeof O);
    O.n = 0;
    O.pos_w = t->po_pos, O.uvr = t->po_uvr, O.inv_sigma_sq = t->po_w, O.huber = t->po_h;
    if (t->cam.model == SVGPU_CAM_EQUIRECTANGULAR) O.intr[0] = 0, O.intr[1] = 0, O.intr[2] = t->cam.cols, O.intr[3] = t->cam.rows, O.intr[4] = 0;
    else O.intr[0] = t->cam.fx, O.intr[1] = t->cam.fy, O.intr[2] = t->cam.cx, O.intr[3] = t->cam.cy, O.intr[4] = t->cam.focal_x_baseline;
    if (pose_by_value) memcpy(O.pose_in, pose_by_value, sizeof O.pose_in);
    else O.pose_in_dev = t->d_pose;
    O.num_trials_robust = t->cfg.po_num_trials_robust, O.num_trials = t->cfg.po_num_trials, O.num_each_iter = t->cfg.po_num_each_iter;
    O.reset_flag_each_round = t->cfg.po_reset_stop_flag_each_round;
    O.gain_thr = 1e-3;  // terminateAction->setGainThreshold(1e-3), pose_optimizer_g2o.cc:55
    O.pose_out = t->d_pose;
    O.outlier = t->po_outlier, O.result = t->po_result, O.level = t->po_level, O.robust = t->po_robust;
    O.trk_overflow = t->num + 2, O.trk_overflow_cap = C.cap;
    O.trk_match_q = t->match_q, O.trk_qid = q_ids, O.trk_nq = nq;
    O.trk_cur_lm = cur_lm, O.trk_reset_cur = reset_cur, O.trk_who = t->who;
    O.trk_nt_dev = nt_dev, O.trk_nt = nt;
    O.trk_map = t->map->rec, O.trk_map_cap = t->map->cap;
    O.trk_xy = cur->xy, O.trk_octave = cur->octave, O.trk_xright = cur->has_xright ? cur->xright : nullptr;
    for (int l = 0; l < 16; ++l) O.trk_inv_sigma_sq[l] = l < t->cfg.num_levels ? t->cfg.inv_level_sigma_sq[l] : 0.f;
    constexpr float chi_sq_2D = 5.99146, chi_sq_3D = 7.81473;  // pose_optimizer_g2o.cc:74-79
    O.trk_huber = t->cfg.is_monocular ? std::sqrt(chi_sq_2D) : std::sqrt(chi_sq_3D);
    O.trk_pos = t->po_pos, O.trk_uvr = t->po_uvr, O.trk_w = t->po_w, O.trk_h = t->po_h, O.trk_kp_of = t->kp_of;
    O.trk_outlier_kp = t->outlier_kp, O.host_outlier_kp = t->h_outlier, O.host_pose = t->h_pose, O.host_result = t->h_result;
    O.trk_counter_reset = t->num + 2;
    static const bool want_stamps = std::getenv("SVGPU_TRACK_STAMPS") != nullptr;
    if (want_stamps) {
        O.stamps = t->h_stamps;
        t->h_stamps[0] = 0;
    }
    sv_pose_opt(ctx, s, O);
    SV_HIP(ctx, hipGetLastError());
    t->launches += 3;
    return SVGPU_OK;
}

void fill_result(const svgpu_tracker* t, int n_kp, svgpu_track_result* r) {
    r->n_keypoints = n_kp;
    r->num_matches = t->h_num[0];
    r->num_candidates = t->h_num[1];
    r->replay_sweeps = t->h_num[2];
    r->reserved = 0;
    r->num_valid = t->h_result[0];
    r->lm_iterations = t->h_result[1];
    r->num_observations = t->h_result[3];
    memcpy(r->pose_cw, t->h_pose, sizeof r->pose_cw);
}
}  // namespace

extern "C" {

int svgpu_tracker_create(svgpu_ctx* ctx, svgpu_map* map, const svgpu_camera* cam, const svgpu_track_config* cfg, svgpu_tracker** out) {
    if (!ctx || !map || !cam || !cfg || !out || cfg->num_levels < 1 || cfg->num_levels > 16 || cfg->grid_cols < 1 || cfg->grid_rows < 1
        || (size_t)cfg->grid_cols * cfg->grid_rows > 4096 /* GRID_ONE_CELLS: the one-workgroup grid build */ || cam->model < SVGPU_CAM_PERSPECTIVE
        || cam->model > SVGPU_CAM_RADIAL_DIVISION || !(cam->min_x < cam->max_x) || !(cam->min_y < cam->max_y) || cfg->po_num_trials_robust < 0
        || cfg->po_num_trials < 0 || cfg->po_num_each_iter < 0 || !(cfg->log_scale_factor > 0.f))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_tracker_create: bad arguments");
    if (map->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_tracker_create: the map lives on another device");
    svgpu_tracker* t = new (std::nothrow) svgpu_tracker();
    if (!t) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_tracker_create: out of memory");
    t->ctx = ctx, t->map = map, t->map_device = ctx->device, t->cam = *cam, t->cfg = *cfg;
    *out = t;
    return SVGPU_OK;
}

void svgpu_tracker_destroy(svgpu_tracker* t) {
    if (!t) return;
    // (every call of the tracker ends with its own synchronisation: nothing is in flight; the context may already be gone)
    (void)hipSetDevice(t->map_device);
    release(t);
    release_right(t);
    delete t;
}

int svgpu_tracker_observation(const svgpu_tracker* t, const svgpu_keypoint** kps, const uint8_t** desc, const svgpu_keypoint** undist_kps, const double** bearings) {
    if (!t || !t->h_obs || t->obs_n <= 0) return 0;
    if (kps) *kps = (const svgpu_keypoint*)(t->h_obs + t->obs_off[0]);
    if (desc) *desc = (const uint8_t*)(t->h_obs + t->obs_off[1]);
    if (undist_kps) *undist_kps = (const svgpu_keypoint*)(t->h_obs + t->obs_off[2]);
    if (bearings) *bearings = (const double*)(t->h_obs + t->obs_off[3]);
    return t->obs_n;
}

const unsigned long long* svgpu_tracker_debug_stamps(const svgpu_tracker* t) { return t ? t->h_stamps : nullptr; }

int svgpu_tracker_counters(const svgpu_tracker* t, long long* launches, long long* host_syncs) {
    if (!t) return SVGPU_ERR_INVALID;
    if (launches) *launches = t->launches;
    if (host_syncs) *host_syncs = t->syncs;
    return SVGPU_OK;
}

}  // extern "C"

namespace {
// svgpu_track_motion / svgpu_track_motion_stereo: `ctx_right` + `img_right` make the submission a stereo frame's (system.cc:406-447: the right
// image's extraction on its own context and stream beside the left one's, match::stereo::compute behind both, then the frame observation)
int track_motion(svgpu_tracker* t, svgpu_ctx* ctx_right, svgpu_frame* cur, const uint8_t* img, int stride, const uint8_t* img_right, int stride_right,
                 const float* depth_img, int depth_stride, const svgpu_frame* last, const int32_t* last_lm_ids, const double* pose_guess_cw, const double* pose_last_cw, float margin, int check_orientation,
                 svgpu_keypoint* kps, uint8_t* desc, svgpu_keypoint* undist_kps, double* bearings, int cap, int32_t* match_last, uint8_t* outlier,
                 svgpu_track_result* result) {
    if (!t) return SVGPU_ERR_INVALID;
    svgpu_ctx* ctx = t->ctx;
    if (!cur || !last || !pose_guess_cw || !pose_last_cw || !result || cur == last || (last->n > 0 && (!last_lm_ids || !match_last)) || !outlier || cap < 0
        || (img && (kps || desc || undist_kps || bearings) && (!kps || !desc || !undist_kps || !bearings)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion: bad arguments");
    if (cur->device != ctx->device || last->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion: a frame lives on another device");
    const OrbConfig& C = ctx->orb;
    if (img && (!C.configured || stride < C.width)) return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_track_motion: fused extraction needs svgpu_orb_configure on the tracker's context");
    const bool stereo = img && img_right, rgbd = img && depth_img;
    if (rgbd && (stereo || depth_stride < C.width || t->cfg.is_monocular))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion_rgbd: needs a depth image of the frame's size and a tracker created with is_monocular = 0");
    if (stereo) {
        if (!ctx_right || ctx_right == ctx || ctx_right->device != ctx->device)
            return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion_stereo: the right image needs a context of its own on the tracker's device");
        if (!ctx_right->orb.configured) {  // the right extractor has not run yet: configure its context like the left one
            const int rcc = svgpu_orb_configure(ctx_right, C.width, C.height, C.max_batch, C.scale_factor, C.num_levels, C.ini_thr, C.min_thr, C.min_area_sqrt * C.min_area_sqrt);
            if (rcc) return sv_set_error(ctx, rcc, "svgpu_track_motion_stereo: configuring the right context failed");
        }
        const OrbConfig& CR = ctx_right->orb;
        if (!CR.configured || CR.width != C.width || CR.height != C.height || CR.num_levels != C.num_levels || CR.total_grid != C.total_grid || stride_right < C.width)
            return sv_set_error(ctx, SVGPU_ERR_NOT_CONFIGURED, "svgpu_track_motion_stereo: the right context must be configured like the left one");
        if (t->cfg.is_monocular) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion_stereo: the tracker was created for a monocular set-up");
    }
    if (!img && (cur->grid_cols != t->cfg.grid_cols || cur->grid_rows != t->cfg.grid_rows))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion: the current frame was binned over another grid");
    // (landmark ids are validated where they are used: k_track_cand treats an id outside [0, map capacity) as "no landmark")
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n_last = last->n, ncell = t->cfg.grid_cols * t->cfg.grid_rows;
    const int nt_cap = img ? std::max(1, C.total_grid) : cur->n;
    if (nt_cap > 1024 * 8) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_track_motion: more than 8192 keypoints per frame");
    int rc;
    if (img && (rc = sv_frame_reserve(ctx, cur, nt_cap, ncell))) return rc;
    const int pitch = img ? C.levels[0].pitch : 0;
    const size_t img_bytes = img ? (size_t)pitch * C.height : 0;
    // what comes back of a fresh observation: the slab's prefix kps_raw | desc | undist | bearings
    const size_t obs_bytes = !img ? 0 : (stereo || rgbd) ? (size_t)((char*)cur->depth - cur->slab) + (size_t)cur->cap * 4 : (size_t)((char*)cur->bearings - cur->slab) + (size_t)cur->cap * 24;
    const size_t depth_bytes = rgbd ? (size_t)C.width * C.height * sizeof(float) : 0;
    if ((stereo || rgbd) && (rc = reserve_right(t, std::max(img_bytes, depth_bytes), nt_cap))) return rc;
    if ((rc = reserve(t, nt_cap, n_last, t->cap_cand, img_bytes, obs_bytes))) return rc;
    // assume_forward / assume_backward (projection.cc:101-116)
    double Rg[9], tg[3], twc[3], tlc[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Rg[3 * i + j] = pose_guess_cw[4 * i + j];
        tg[i] = pose_guess_cw[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i) twc[i] = ((-Rg[i]) * tg[0] + (-Rg[3 + i]) * tg[1]) + (-Rg[6 + i]) * tg[2];
    for (int i = 0; i < 3; ++i)
        tlc[i] = ((pose_last_cw[4 * i] * twc[0] + pose_last_cw[4 * i + 1] * twc[1]) + pose_last_cw[4 * i + 2] * twc[2]) + pose_last_cw[4 * i + 3];
    const bool fwd = t->cfg.is_monocular ? false : tlc[2] > (double)t->cfg.true_baseline;
    const bool bwd = t->cfg.is_monocular ? false : -tlc[2] > (double)t->cfg.true_baseline;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool extract = img && attempt == 0;
        // ---- input block: [image] | last frame's landmark ids, one copy
        const size_t o_ids = pad256(t->img_bytes);
        if (extract) {
            if (stride == pitch) memcpy(t->h_in, img, img_bytes);
            else
                for (int y = 0; y < C.height; ++y) memcpy(t->h_in + (size_t)y * pitch, img + (size_t)y * stride, C.width);
        }
        if (n_last > 0) memcpy(t->h_in + o_ids, last_lm_ids, (size_t)n_last * 4);
        if (extract && stereo) {  // the right image: its own copy, its own extraction, on the right context's stream -- beside everything below
            if (stride_right == pitch) memcpy(t->h_in_r, img_right, img_bytes);
            else
                for (int y = 0; y < C.height; ++y) memcpy(t->h_in_r + (size_t)y * pitch, img_right + (size_t)y * stride_right, C.width);
            hipStream_t sr = ctx_right->stream;
            SV_HIP(ctx, hipMemcpyAsync(t->d_in_r, t->h_in_r, img_bytes, hipMemcpyHostToDevice, sr));
            rc = svgpu_orb_extract_batch_device(ctx_right, (const uint8_t*)t->d_in_r, 1, img_bytes, pitch, nullptr, 0, pitch, t->r_kps, t->r_desc, nt_cap, t->r_counts, sr);
            if (rc) {
                (void)hipStreamSynchronize(sr);
                return sv_set_error(ctx, rc, "svgpu_track_motion_stereo: extraction of the right image failed");
            }
            SV_HIP(ctx, hipEventRecord(t->ev_right, sr));
            t->launches += 6;
        }
        if (extract && rgbd) {  // the depth image rides down on the main stream behind the colour image (dense rows of C.width floats)
            for (int y = 0; y < C.height; ++y) memcpy(t->h_in_r + (size_t)y * C.width * 4, depth_img + (size_t)y * depth_stride, (size_t)C.width * 4);
        }
        std::unique_lock<std::mutex> lock(t->map->mtx);
        if (t->map->cap == 0) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_motion: the map is empty");
        SvMapReadScope reading(ctx, t->map, s);  // (declared behind the lock: every exit path records the read before the mutex is released)
        if ((rc = reading.begin())) return rc;
        if (extract) SV_HIP(ctx, hipMemcpyAsync(t->d_in, t->h_in, o_ids + (size_t)n_last * 4, hipMemcpyHostToDevice, s));
        else if (n_last > 0) SV_HIP(ctx, hipMemcpyAsync(t->d_in + o_ids, t->h_in + o_ids, (size_t)n_last * 4, hipMemcpyHostToDevice, s));
        t->launches += 1;
        const int32_t* d_last_ids = (const int32_t*)(t->d_in + o_ids);
        const int32_t* nt_dev = nullptr;
        if (img) {
            nt_dev = cur->counts;
            if (extract) {
                rc = svgpu_orb_extract_batch_device(ctx, (const uint8_t*)t->d_in, 1, img_bytes, pitch, nullptr, 0, pitch, cur->kps_raw, cur->desc, nt_cap, cur->counts, s);
                if (rc) return rc;
                cur->grid_cols = t->cfg.grid_cols, cur->grid_rows = t->cfg.grid_rows;
                cur->min_x = t->cam.min_x, cur->max_x = t->cam.max_x, cur->min_y = t->cam.min_y, cur->max_y = t->cam.max_y;
                cur->has_xright = false;
                if (stereo) {  // match::stereo::compute (match/stereo.cc:20-114) on the two extractions' device-resident outputs and pyramids
                    SV_HIP(ctx, hipStreamWaitEvent(s, t->ev_right, 0));
                    rc = svgpu_stereo_match_batch_device(ctx, ctx_right, 1, cur->kps_raw, cur->desc, cur->counts, t->r_kps, t->r_desc, t->r_counts, nt_cap,
                                                         1 + C.num_levels, (float)t->cam.focal_x_baseline, t->cfg.true_baseline, cur->xright, cur->depth, s);
                    if (rc) return rc;
                    cur->has_xright = true;
                    t->launches += 4;
                }
                TrackFrameProblem F{};
                F.cam = t->cam;
                F.kps = cur->kps_raw;
                F.n_dev = cur->counts;
                F.cap = nt_cap;
                F.undist = cur->undist, F.xy = cur->xy, F.octave = cur->octave, F.angle = cur->angle, F.bearings = cur->bearings;
                F.G.t_xy = cur->xy;
                F.G.t_octave = cur->octave;
                F.G.min_x = t->cam.min_x;
                F.G.min_y = t->cam.min_y;
                F.G.inv_w = (double)t->cfg.grid_cols / (t->cam.max_x - t->cam.min_x);
                F.G.inv_h = (double)t->cfg.grid_rows / (t->cam.max_y - t->cam.min_y);
                F.G.cols = t->cfg.grid_cols;
                F.G.rows = t->cfg.grid_rows;
                F.G.cell_of = cur->cell_of, F.G.cell_off = cur->cell_off, F.G.cell_items = cur->cell_items;
                F.n_host = t->h_n;
                if (rgbd) {
                    SV_HIP(ctx, hipMemcpyAsync(t->d_in_r, t->h_in_r, depth_bytes, hipMemcpyHostToDevice, s));
                    t->launches += 1;
                    F.depth_img = (const float*)t->d_in_r, F.depth_pitch = C.width, F.focal_x_baseline = t->cam.focal_x_baseline;
                    F.xright = cur->xright, F.depth_out = cur->depth;
                    cur->has_xright = true;
                }
                sv_launch_track_frame(ctx, s, F);
                t->launches += 6;
                // the observation goes home on the auxiliary stream, beside the matcher and the optimiser (joined before the synchronisation)
                if (ctx->stream_aux) {
                    SV_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
                    SV_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->ev_fork, 0));
                    SV_HIP(ctx, hipMemcpyAsync(t->h_obs, cur->slab, obs_bytes, hipMemcpyDeviceToHost, ctx->stream_aux));
                    SV_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream_aux));
                    t->launches += 1;
                }
            }
        }
        TrackCandProblem Cd{};
        fill_reproj(t, Cd.R, pose_guess_cw, margin);
        Cd.R.dist_mode = 2, Cd.R.normal_mode = 2, Cd.R.center_mode = 0, Cd.R.window_mode = fwd ? 1 : (bwd ? 2 : 0);
        Cd.mode = 0;
        Cd.q_octave = last->octave;
        Cd.q_angle = last->angle;
        Cd.check_orientation = check_orientation;
        rc = enqueue_match_and_optimize(t, s, Cd, cur, nt_cap, nt_dev, t->cur_lm_motion, 1, d_last_ids, n_last, 100u /* HAMMING_DIST_THR_HIGH */, 0.f,
                                        SVGPU_MATCH_BEST_ONLY, pose_guess_cw);
        if (rc) return rc;
        if (extract) {
            if (ctx->stream_aux) SV_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
            else {
                SV_HIP(ctx, hipMemcpyAsync(t->h_obs, cur->slab, obs_bytes, hipMemcpyDeviceToHost, s));
                t->launches += 1;
            }
        }
        if ((rc = reading.end())) return rc;
        lock.unlock();
        SV_HIP(ctx, hipStreamSynchronize(s));
        t->syncs += 1;
        if (extract) {  // host copies of the fresh observation (data::frame_observation), out of the slab prefix
            cur->n = t->h_n[0];
            t->obs_n = cur->n;
            t->obs_off[0] = (size_t)((char*)cur->kps_raw - cur->slab), t->obs_off[1] = (size_t)((char*)cur->desc - cur->slab);
            t->obs_off[2] = (size_t)((char*)cur->undist - cur->slab), t->obs_off[3] = (size_t)((char*)cur->bearings - cur->slab);
            t->obs_off[4] = (size_t)((char*)cur->xright - cur->slab), t->obs_off[5] = (size_t)((char*)cur->depth - cur->slab);
            t->obs_stereo = stereo || rgbd;
            if (kps) {
                const int m = std::min(cur->n, cap);
                const char* o = t->h_obs;
                memcpy(kps, o + t->obs_off[0], (size_t)m * sizeof(svgpu_keypoint));
                memcpy(desc, o + t->obs_off[1], (size_t)m * 32);
                memcpy(undist_kps, o + t->obs_off[2], (size_t)m * sizeof(svgpu_keypoint));
                memcpy(bearings, o + t->obs_off[3], (size_t)m * 24);
            }
        }
        if ((size_t)t->h_num[1] <= t->cap_cand) break;
        // the candidate lists did not fit the capacity: nothing ran behind the list kernel.  Grow (fresh buffers, counter at zero) and enqueue
        // the matcher again -- the extraction stands.
        if (attempt == 1) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_track_motion: candidate lists keep overflowing");
        if ((rc = reserve(t, nt_cap, n_last, (size_t)t->h_num[1] + (size_t)t->h_num[1] / 4 + 4096, t->img_bytes, t->h_obs_bytes))) return rc;
    }
    const int n_kp = cur->n;
    if (n_last > 0) memcpy(match_last, t->h_match, (size_t)n_last * 4);
    memcpy(outlier, t->h_outlier, (size_t)(img ? std::min(n_kp, cap) : n_kp));
    fill_result(t, n_kp, result);
    if (img && n_kp > cap) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_track_motion: more keypoints than cap");
    return SVGPU_OK;
}
}  // namespace

extern "C" {

int svgpu_track_motion(svgpu_tracker* t, svgpu_frame* cur, const uint8_t* img, int stride, const svgpu_frame* last, const int32_t* last_lm_ids,
                       const double* pose_guess_cw, const double* pose_last_cw, float margin, int check_orientation, svgpu_keypoint* kps, uint8_t* desc,
                       svgpu_keypoint* undist_kps, double* bearings, int cap, int32_t* match_last, uint8_t* outlier, svgpu_track_result* result) {
    return track_motion(t, nullptr, cur, img, stride, nullptr, 0, nullptr, 0, last, last_lm_ids, pose_guess_cw, pose_last_cw, margin, check_orientation, kps, desc, undist_kps,
                        bearings, cap, match_last, outlier, result);
}

int svgpu_track_motion_stereo(svgpu_tracker* t, svgpu_ctx* ctx_right, svgpu_frame* cur, const uint8_t* img_left, int stride_left, const uint8_t* img_right,
                              int stride_right, const svgpu_frame* last, const int32_t* last_lm_ids, const double* pose_guess_cw, const double* pose_last_cw,
                              float margin, int check_orientation, int cap, int32_t* match_last, uint8_t* outlier, svgpu_track_result* result) {
    if (!t) return SVGPU_ERR_INVALID;
    if (!img_left || !img_right) return sv_set_error(t->ctx, SVGPU_ERR_INVALID, "svgpu_track_motion_stereo: both images are required");
    return track_motion(t, ctx_right, cur, img_left, stride_left, img_right, stride_right, nullptr, 0, last, last_lm_ids, pose_guess_cw, pose_last_cw, margin,
                        check_orientation, nullptr, nullptr, nullptr, nullptr, cap, match_last, outlier, result);
}

int svgpu_track_motion_rgbd(svgpu_tracker* t, svgpu_frame* cur, const uint8_t* img, int stride, const float* depth, int depth_stride, const svgpu_frame* last,
                            const int32_t* last_lm_ids, const double* pose_guess_cw, const double* pose_last_cw, float margin, int check_orientation, int cap,
                            int32_t* match_last, uint8_t* outlier, svgpu_track_result* result) {
    if (!t) return SVGPU_ERR_INVALID;
    if (!img || !depth) return sv_set_error(t->ctx, SVGPU_ERR_INVALID, "svgpu_track_motion_rgbd: the image and the depth image are required");
    return track_motion(t, nullptr, cur, img, stride, nullptr, 0, depth, depth_stride, last, last_lm_ids, pose_guess_cw, pose_last_cw, margin, check_orientation,
                        nullptr, nullptr, nullptr, nullptr, cap, match_last, outlier, result);
}

int svgpu_tracker_observation_stereo(const svgpu_tracker* t, const float** x_right, const float** depths) {
    if (!t || !t->h_obs || t->obs_n <= 0 || !t->obs_stereo) return 0;
    if (x_right) *x_right = (const float*)(t->h_obs + t->obs_off[4]);
    if (depths) *depths = (const float*)(t->h_obs + t->obs_off[5]);
    return t->obs_n;
}

int svgpu_track_local_map(svgpu_tracker* t, const svgpu_frame* cur, const int32_t* cur_lm_ids, int n_local, const int32_t* local_ids, const double* pose_cw,
                          float margin, float lowe_ratio, float ray_cos_thr, int32_t* match_local, uint8_t* visible, uint8_t* outlier,
                          svgpu_track_result* result) {
    if (!t) return SVGPU_ERR_INVALID;
    svgpu_ctx* ctx = t->ctx;
    if (!cur || n_local < 0 || !result || !outlier || (cur->n > 0 && !cur_lm_ids) || (n_local > 0 && (!local_ids || !match_local)))
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map: bad arguments");
    if (cur->device != ctx->device) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map: the frame lives on another device");
    if (cur->grid_cols != t->cfg.grid_cols || cur->grid_rows != t->cfg.grid_rows)
        return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map: the frame was binned over another grid");
    if (!pose_cw && !t->d_work) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map: no pose on the device yet (pass pose_cw)");
    SV_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int nt = cur->n;
    if (nt > 1024 * 8) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_track_local_map: more than 8192 keypoints per frame");
    int rc;
    if ((rc = reserve(t, nt, n_local, t->cap_cand, t->img_bytes, t->h_obs_bytes))) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- input block (behind the image area): the frame's landmark ids | the local landmarks' ids, one copy
        const size_t o_cur = pad256(t->img_bytes), o_loc = o_cur + pad256((size_t)t->cap_kp * 4);
        if (nt > 0) memcpy(t->h_in + o_cur, cur_lm_ids, (size_t)nt * 4);
        if (n_local > 0) memcpy(t->h_in + o_loc, local_ids, (size_t)n_local * 4);
        std::unique_lock<std::mutex> lock(t->map->mtx);
        if (t->map->cap == 0) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map: the map is empty");
        SvMapReadScope reading(ctx, t->map, s);
        if ((rc = reading.begin())) return rc;
        SV_HIP(ctx, hipMemcpyAsync(t->d_in + o_cur, t->h_in + o_cur, (o_loc - o_cur) + (size_t)n_local * 4, hipMemcpyHostToDevice, s));
        t->launches += 1;
        TrackCandProblem Cd{};
        fill_reproj(t, Cd.R, pose_cw, margin);
        Cd.R.ray_cos_thr = ray_cos_thr;
        Cd.R.dist_mode = 0, Cd.R.normal_mode = 0, Cd.R.center_mode = 0, Cd.R.window_mode = 0;
        Cd.pose_dev = pose_cw ? nullptr : t->d_pose;
        Cd.mode = 1;
        Cd.cur_lm = (const int32_t*)(t->d_in + o_cur);
        Cd.visible_host = t->h_visible;
        rc = enqueue_match_and_optimize(t, s, Cd, cur, nt, nullptr, (int32_t*)(t->d_in + o_cur), 0, (const int32_t*)(t->d_in + o_loc), n_local,
                                        100u /* HAMMING_DIST_THR_HIGH */, lowe_ratio, SVGPU_MATCH_RATIO_SAME_OCTAVE, pose_cw);
        if (rc) return rc;
        if ((rc = reading.end())) return rc;
        lock.unlock();
        SV_HIP(ctx, hipStreamSynchronize(s));
        t->syncs += 1;
        if ((size_t)t->h_num[1] <= t->cap_cand) break;
        if (attempt == 1) return sv_set_error(ctx, SVGPU_ERR_CAPACITY, "svgpu_track_local_map: candidate lists keep overflowing");
        if ((rc = reserve(t, nt, n_local, (size_t)t->h_num[1] + (size_t)t->h_num[1] / 4 + 4096, t->img_bytes, t->h_obs_bytes))) return rc;  // (keeps the device pose)
    }
    if (n_local > 0) memcpy(match_local, t->h_match, (size_t)n_local * 4);
    if (visible && n_local > 0) memcpy(visible, t->h_visible, n_local);
    memcpy(outlier, t->h_outlier, nt);
    fill_result(t, nt, result);
    t->last_n_local = n_local;
    return SVGPU_OK;
}

int svgpu_track_local_map_observability(svgpu_tracker* t, int n_local, double* reproj, float* x_right, int32_t* pred_scale_level) {
    if (!t) return SVGPU_ERR_INVALID;
    svgpu_ctx* ctx = t->ctx;
    if (n_local < 0 || n_local != t->last_n_local) return sv_set_error(ctx, SVGPU_ERR_INVALID, "svgpu_track_local_map_observability: no matching svgpu_track_local_map call");
    if (n_local == 0) return SVGPU_OK;
    SV_HIP(ctx, hipSetDevice(ctx->device));
    if (reproj) SV_HIP(ctx, hipMemcpyAsync(reproj, t->reproj, (size_t)n_local * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (x_right) SV_HIP(ctx, hipMemcpyAsync(x_right, t->x_right, (size_t)n_local * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pred_scale_level) SV_HIP(ctx, hipMemcpyAsync(pred_scale_level, t->pred_level, (size_t)n_local * 4, hipMemcpyDeviceToHost, ctx->stream));
    SV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SVGPU_OK;
}

}  // extern "C"
