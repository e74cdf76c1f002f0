// Host side of the tracked-frame chain: the landmark table's shadow + dirty list, and the two reference call sites (tracking_hip.h).
#include "tracking_hip.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace stella_vslam {
namespace hip {

// ------------------------------------------------------------------------------------------------------- the landmark table's host shadow
namespace {
struct mirror_state {
    std::mutex mtx;
    std::vector<svgpu_landmark_record> shadow;  // by landmark id
    std::vector<uint8_t> is_dirty;
    std::vector<uint32_t> dirty;
    svgpu_map* map = nullptr;
    svgpu_ctx* map_ctx = nullptr;  // (the context the table was created on: only its device matters)
    std::mutex upload_mtx;         // one flush uploads at a time: batches reach the table in the order their records were copied
    std::vector<uint32_t> up_ids;  // flush scratch (under upload_mtx)
    std::vector<svgpu_landmark_record> up_rec;
};
// Leaked on purpose (as frame_pool is): a function-local static would destroy the table -- hipEventSynchronize, hipFree -- during static
// destruction, possibly after the HIP runtime has been torn down.  release_map() is the explicit shutdown.
mirror_state& mirror() {
    static mirror_state* s = new mirror_state();
    return *s;
}
// (called with the mutex held)
svgpu_landmark_record& touch(mirror_state& M, unsigned int id) {
    if (id >= M.shadow.size()) {
        const size_t n = std::max<size_t>((size_t)id + 1, M.shadow.size() * 2 + 1024);
        svgpu_landmark_record zero;
        std::memset(&zero, 0, sizeof zero);
        M.shadow.resize(n, zero);
        M.is_dirty.resize(n, 0);
    }
    if (!M.is_dirty[id]) {
        M.is_dirty[id] = 1;
        M.dirty.push_back(id);
    }
    return M.shadow[id];
}
}  // namespace

namespace map_mirror {
void landmark_created(unsigned int id, const double* pos_w) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    svgpu_landmark_record& r = touch(M, id);
    std::memset(&r, 0, sizeof r);
    for (int k = 0; k < 3; ++k) r.pos_w[k] = pos_w[k];
    r.flags = SVGPU_LM_PRESENT;
}
void set_position(unsigned int id, const double* pos_w) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    svgpu_landmark_record& r = touch(M, id);
    for (int k = 0; k < 3; ++k) r.pos_w[k] = pos_w[k];
}
void set_geometry(unsigned int id, const double* mean_normal, float min_valid_dist, float max_valid_dist) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    svgpu_landmark_record& r = touch(M, id);
    for (int k = 0; k < 3; ++k) r.mean_normal[k] = mean_normal[k];
    r.min_valid_dist = min_valid_dist;
    r.max_valid_dist = max_valid_dist;
}
void set_descriptor(unsigned int id, const unsigned char* descriptor32) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    svgpu_landmark_record& r = touch(M, id);
    if (descriptor32) {
        std::memcpy(r.descriptor, descriptor32, 32);
        r.flags |= SVGPU_LM_HAS_DESCRIPTOR;
    }
    else r.flags &= ~(uint32_t)SVGPU_LM_HAS_DESCRIPTOR;
}
void set_has_observation(unsigned int id, bool has_observation) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    svgpu_landmark_record& r = touch(M, id);
    if (has_observation) r.flags |= SVGPU_LM_HAS_OBSERVATION;
    else r.flags &= ~(uint32_t)SVGPU_LM_HAS_OBSERVATION;
}
void landmark_erased(unsigned int id) {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    touch(M, id).flags = 0;
}
}  // namespace map_mirror

svgpu_map* flush_map(svgpu_ctx* ctx) {
    mirror_state& M = mirror();
    // Two locks: the mirror's is held only while the dirty records are COPIED (every data::landmark mutator takes it under its own landmark
    // mutex: an upload of thousands of records after a BA write-back -- a copy, a kernel and a stream synchronisation -- must not stall the
    // mapping, loop-closing and tracking threads behind it); the upload lock keeps the batches in copy order, so a record changed again
    // during the upload is dirty again and its newer copy reaches the table with the NEXT flush, never before this one.
    std::lock_guard<std::mutex> uploading(M.upload_mtx);
    size_t n = 0;
    svgpu_map* map = nullptr;
    {
        std::lock_guard<std::mutex> lock(M.mtx);
        if (!M.map) {
            check(svgpu_map_create(ctx, &M.map), "svgpu_map_create");
            M.map_ctx = ctx;
        }
        map = M.map;
        n = M.dirty.size();
        M.up_ids.resize(n);
        M.up_rec.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t id = M.dirty[i];
            M.up_ids[i] = id;
            M.up_rec[i] = M.shadow[id];
            M.is_dirty[id] = 0;
        }
        M.dirty.clear();
    }
    if (n > 0) check(svgpu_map_upsert(ctx, map, (int)n, M.up_ids.data(), M.up_rec.data()), "svgpu_map_upsert");
    return map;
}
void release_map() {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> uploading(M.upload_mtx);
    std::lock_guard<std::mutex> lock(M.mtx);
    if (M.map) svgpu_map_destroy(M.map);
    M.map = nullptr;
    M.map_ctx = nullptr;
    // every record the shadow holds is dirty again for a table created later
    M.dirty.clear();
    for (size_t id = 0; id < M.shadow.size(); ++id) {
        M.is_dirty[id] = M.shadow[id].flags ? 1 : 0;
        if (M.is_dirty[id]) M.dirty.push_back((uint32_t)id);
    }
}
size_t pending_map_updates() {
    mirror_state& M = mirror();
    std::lock_guard<std::mutex> lock(M.mtx);
    return M.dirty.size();
}

// ------------------------------------------------------------------------------------------------------------------- the chain
namespace {
// SVGPU_TRACK_TRACE: host-side phase times of the two calls, to stderr
struct lap_timer {
    bool on;
    std::chrono::steady_clock::time_point t;
    const char* who;
    explicit lap_timer(const char* w) : on(std::getenv("SVGPU_TRACK_TRACE") != nullptr), t(std::chrono::steady_clock::now()), who(w) {}
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[track host] %-10s %-34s %7.1f us\n", who, what, std::chrono::duration<double, std::micro>(n - t).count());
        t = n;
    }
};
void pose12(const Mat44_t& T, double* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = T(i, j);
}
Mat44_t pose44(const double* p) {
    Mat44_t T = Mat44_t::Identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) T(i, j) = p[4 * i + j];
    return T;
}
}  // namespace

tracked_frame_chain::tracked_frame_chain(svgpu_ctx* ctx, const camera::base* camera, const feature::orb_params* orb_params, unsigned int num_grid_cols,
                                         unsigned int num_grid_rows, unsigned int num_trials_robust, unsigned int num_trials, unsigned int num_each_iter)
    : ctx_(ctx), camera_(camera), orb_params_(orb_params), num_grid_cols_(num_grid_cols), num_grid_rows_(num_grid_rows) {
    map_ = flush_map(ctx_);
    svgpu_track_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.num_levels = (int)orb_params->num_levels_;
    if (cfg.num_levels < 1 || cfg.num_levels > 16) throw std::runtime_error("tracked_frame_chain: unsupported number of pyramid levels");
    for (int l = 0; l < cfg.num_levels; ++l) {
        cfg.scale_factors[l] = orb_params->scale_factors_.at(l);
        cfg.inv_level_sigma_sq[l] = orb_params->inv_level_sigma_sq_.at(l);
    }
    cfg.log_scale_factor = orb_params->log_scale_factor_;
    cfg.grid_cols = (int)num_grid_cols, cfg.grid_rows = (int)num_grid_rows;
    cfg.is_monocular = camera->setup_type_ == camera::setup_type_t::Monocular ? 1 : 0;
    cfg.true_baseline = (float)camera->true_baseline_;
    cfg.po_num_trials_robust = (int)num_trials_robust, cfg.po_num_trials = (int)num_trials, cfg.po_num_each_iter = (int)num_each_iter;
    cfg.po_reset_stop_flag_each_round = 0;
    const svgpu_camera cam = to_svgpu_camera(camera);
    check(svgpu_tracker_create(ctx_, map_, &cam, &cfg, &tracker_), "svgpu_tracker_create");
}

tracked_frame_chain::~tracked_frame_chain() { svgpu_tracker_destroy(tracker_); }

void tracked_frame_chain::counters(long long& launches, long long& host_syncs) const { svgpu_tracker_counters(tracker_, &launches, &host_syncs); }

frame_handle tracked_frame_chain::handle_of(const data::frame& frm) {
    for (const memo& m : memo_)
        if (m.h && m.id == frm.id_ && (size_t)svgpu_frame_size(m.h.get()) == frm.frm_obs_.undist_keypts_.size()) return m.h;
    frame_handle h = resident(frm);
    if (!h) throw std::runtime_error("tracked_frame_chain: resident frames are disabled (SVGPU_NO_RESIDENT_FRAMES)");
    remember(frm.id_, h);
    return h;
}
void tracked_frame_chain::remember(unsigned int frame_id, const frame_handle& h) {
    for (memo& m : memo_)
        if (m.h && m.id == frame_id) {
            m.h = h;
            return;
        }
    memo_[1] = memo_[0];  // (two frames: the current one and the one before)
    memo_[0].id = frame_id;
    memo_[0].h = h;
}

namespace {
// system.cc:403-404: the frame of a freshly built observation.  data::frame sizes its (private) landmark vector in its constructor -- from
// the keypoint count alone, so it is constructed from a stub with that many keypoints (the constructor copies its observation twice:
// 400 KB for a 2 400-keypoint frame) and the observation itself is MOVED into the public frm_obs_ afterwards.
// What the constructor does not take is carried over: the reference keyframe (tracking_module.cc:202 sets it BEFORE track_current_frame; the
// BoW fallback of a failed motion track dereferences it, :346), the BoW vectors and a pose somebody already set.
void rebuild_frame(data::frame& frm, data::frame_observation& frm_obs) {
    data::frame_observation stub;
    stub.undist_keypts_.resize(frm_obs.undist_keypts_.size());
    auto ref_keyfrm = frm.ref_keyfrm_;
    auto bow_feat_vec = std::move(frm.bow_feat_vec_);
#ifdef SVGPU_WITH_STELLA_VSLAM
    auto bow_vec = std::move(frm.bow_vec_);
    const bool had_pose = frm.pose_is_valid();
    const Mat44_t pose = had_pose ? frm.get_pose_cw() : Mat44_t::Identity();
    frm = data::frame(frm.id_, frm.timestamp_, frm.camera_, const_cast<feature::orb_params*>(frm.orb_params_), stub, frm.markers_2d_);
    frm.bow_vec_ = std::move(bow_vec);
    if (had_pose) frm.set_pose_cw(pose);
#else
    const Mat44_t pose = frm.get_pose_cw();
    frm = data::frame(frm.id_, frm.camera_, frm.orb_params_, stub);
    frm.set_pose_cw(pose);
#endif
    frm.ref_keyfrm_ = ref_keyfrm;
    frm.bow_feat_vec_ = std::move(bow_feat_vec);
    frm.frm_obs_ = std::move(frm_obs);
}
}  // namespace

bool tracked_frame_chain::motion_based_track(data::frame& curr_frm, const data::frame& last_frm, const Mat44_t& velocity, unsigned int num_matches_thr, float margin,
                                             const cv::Mat* img, std::vector<cv::KeyPoint>* keypts, const cv::Mat* img_right, const cv::Mat* img_depth) {
    if (img_right && (!img || !ctx_right_)) throw std::runtime_error("tracked_frame_chain: a stereo frame needs the left image and set_right_context()");
    if (img_depth && (!img || img_right)) throw std::runtime_error("tracked_frame_chain: an RGB-D frame is an image plus its depth map");
    // (svgpu_track_motion_rgbd reads height x width floats behind the pointer: the raw CV_16U map before util::convert_to_true_depth, or a smaller
    //  map, would be read out of bounds -- ADVICE r5)
#ifdef CV_32FC1
    if (img_depth && (img_depth->type() != CV_32FC1 || img_depth->rows != img->rows || img_depth->cols != img->cols))
#else  // (the stand-in cv::Mat of host/standin/ is untyped bytes: every row must at least hold the image's width in floats)
    if (img_depth && (img_depth->rows != img->rows || img_depth->step < (size_t)img->cols * sizeof(float)))
#endif
        throw std::runtime_error("tracked_frame_chain: the depth map must be CV_32FC1 (metres, util::convert_to_true_depth applied) of the image's size");
    lap_timer T("motion");
    // Set the initial pose by using the motion model (frame_tracker.cc:25-26)
    const Mat44_t guess = velocity * last_frm.get_pose_cw();
    double guess12[12], last12[12];
    pose12(guess, guess12);
    pose12(last_frm.get_pose_cw(), last12);
    // the last frame: its resident observation, and per keypoint the id of the landmark it holds (projection.cc:119-127)
    const frame_handle last_h = handle_of(last_frm);
    const auto last_lms = last_frm.get_landmarks();
    const int n_last = (int)last_frm.frm_obs_.undist_keypts_.size();
    last_ids_.resize(n_last);
    for (int i = 0; i < n_last; ++i) {
        const auto& lm = last_lms[i];
        last_ids_[i] = (lm && !lm->will_be_erased()) ? (int32_t)lm->id_ : -1;
    }
    match_.resize(std::max(n_last, 1));
    T.lap("last frame: resident + ids");
    flush_map(ctx_);
    T.lap("flush_map");
    frame_handle cur_h;
    int cap = 0;
    if (img) {
        cap = std::max(1, svgpu_orb_max_keypoints(ctx_));
        cur_h = new_frame(ctx_);
        outlier_.resize(cap);
    }
    else {
        cur_h = handle_of(curr_frm);
        outlier_.resize(std::max<size_t>(curr_frm.frm_obs_.undist_keypts_.size(), 1));
        cap = (int)outlier_.size();
    }
    unsigned int num_matches = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fused = img && attempt == 0;
        if (fused && img_depth)
            check(svgpu_track_motion_rgbd(tracker_, cur_h.get(), img->ptr(0), (int)img->step, reinterpret_cast<const float*>(img_depth->ptr(0)),
                                          (int)(img_depth->step / sizeof(float)), last_h.get(), last_ids_.data(), guess12, last12, margin, 1, cap, match_.data(),
                                          outlier_.data(), &last_motion_),
                  "svgpu_track_motion_rgbd");
        else if (fused && img_right)
            check(svgpu_track_motion_stereo(tracker_, ctx_right_, cur_h.get(), img->ptr(0), (int)img->step, img_right->ptr(0), (int)img_right->step, last_h.get(),
                                            last_ids_.data(), guess12, last12, margin, 1, cap, match_.data(), outlier_.data(), &last_motion_),
                  "svgpu_track_motion_stereo");
        else
            check(svgpu_track_motion(tracker_, cur_h.get(), fused ? img->ptr(0) : nullptr, fused ? (int)img->step : 0, last_h.get(), last_ids_.data(), guess12, last12,
                                     attempt == 0 ? margin : 2 * margin, 1 /* projection_matcher(0.9, true) */, nullptr, nullptr, nullptr, nullptr, cap, match_.data(),
                                     outlier_.data(), &last_motion_),
                  "svgpu_track_motion");
        T.lap("svgpu_track_motion");
        if (fused) {  // system.cc:380-395: the host copies data::frame_observation holds, straight out of the tracker's page-locked buffer
            const svgpu_keypoint *kps = nullptr, *und = nullptr;
            const uint8_t* desc = nullptr;
            const double* brg = nullptr;
            const int n = svgpu_tracker_observation(tracker_, &kps, &desc, &und, &brg);
            data::frame_observation o;
            o.num_grid_cols_ = num_grid_cols_, o.num_grid_rows_ = num_grid_rows_;
            o.descriptors_.create(n, 32, CV_8U);
            if (n > 0) std::memcpy(o.descriptors_.ptr(0), desc, (size_t)n * 32);
            o.undist_keypts_.resize(n);
            if (n > 0) std::memcpy(static_cast<void*>(o.undist_keypts_.data()), und, (size_t)n * sizeof(svgpu_keypoint));
            o.bearings_.resize(n);
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) o.bearings_[i](k) = brg[3 * (size_t)i + k];
            if (img_right || img_depth) {  // system.cc:443-447 / :492-510
                const float *xr = nullptr, *dp = nullptr;
                if (svgpu_tracker_observation_stereo(tracker_, &xr, &dp) != n) throw std::runtime_error("tracked_frame_chain: the stereo observation is missing");
                o.stereo_x_right_.assign(xr, xr + n);
                o.depths_.assign(dp, dp + n);
            }
            if (keypts) {
                keypts->resize(n);
                if (n > 0) std::memcpy(static_cast<void*>(keypts->data()), kps, (size_t)n * sizeof(svgpu_keypoint));
            }
            rebuild_frame(curr_frm, o);
            register_adopted(curr_frm.id_, cur_h, curr_frm.frm_obs_.undist_keypts_);
            remember(curr_frm.id_, cur_h);
            T.lap("observation -> frm_obs_, register");
        }
        // Initialize the 2D-3D matches, then replay them in the reference's order (frame_tracker.cc:29, projection.cc:202); cur_ids_ mirrors
        // the ids the frame holds so that the loops below do not copy a shared_ptr per keypoint just to look at it
        const int n_cur = (int)curr_frm.frm_obs_.undist_keypts_.size();
        curr_frm.erase_landmarks();
        cur_ids_.assign(std::max(n_cur, 1), -1);
        for (int i = 0; i < n_last; ++i)
            if (0 <= match_[i]) {
                curr_frm.add_landmark(last_lms[i], (unsigned int)match_[i]);
                cur_ids_[match_[i]] = last_ids_[i];
            }
        num_matches = (unsigned int)last_motion_.num_matches;
        T.lap("add_landmark replay");
        if (num_matches >= num_matches_thr) break;  // else: increment the margin, and search again (:36-40)
    }
    curr_frm.set_pose_cw(guess);
    device_pose_valid_ = false;
    if (num_matches < num_matches_thr) return false;
    // Pose optimization (:48-51) -- already done behind the matcher, on the device
    curr_frm.set_pose_cw(pose44(last_motion_.pose_cw));
    device_pose_valid_ = true;  // the tracker's device pose is this frame's pose -- as long as nobody sets another one (track_local_map checks)
    device_pose_frame_ = curr_frm.id_;
    // Discard the outliers (:54, :133-151)
    unsigned int num_valid_matches = 0;
    const unsigned int n = (unsigned int)curr_frm.frm_obs_.undist_keypts_.size();
    for (unsigned int idx = 0; idx < n; ++idx) {
        if (cur_ids_[idx] < 0) continue;
        if (outlier_[idx]) {
            curr_frm.erase_landmark_with_index(idx);
            cur_ids_[idx] = -1;
        }
        else ++num_valid_matches;
    }
    T.lap("discard_outliers");
    return num_valid_matches >= num_matches_thr;
}

bool tracked_frame_chain::track_local_map(data::frame& curr_frm, const std::vector<std::shared_ptr<data::landmark>>& local_landmarks,
                                          unsigned int fixed_keyframe_id_threshold, float margin, float lowe_ratio) {
    lap_timer T("local map");
    const frame_handle cur_h = handle_of(curr_frm);
    T.lap("resident(curr_frm)");
    // select the landmarks which can be reprojected from the ones observed in the current frame (tracking_module.cc:536-551): the frame's
    // own landmarks are stamped in a table by id instead of collected in a hash set
    const unsigned int n = (unsigned int)curr_frm.frm_obs_.undist_keypts_.size();
    const auto cur_lms = curr_frm.get_landmarks();
    cur_ids_.resize(std::max(n, 1u));
    ++frame_serial_;
    for (unsigned int idx = 0; idx < n; ++idx) {
        const auto& lm = cur_lms[idx];
        cur_ids_[idx] = lm ? (int32_t)lm->id_ : -1;
        if (!lm || lm->will_be_erased()) continue;
        if (lm->id_ >= held_stamp_.size()) held_stamp_.resize((size_t)lm->id_ * 2 + 1024, 0);
        held_stamp_[lm->id_] = frame_serial_;
        lm->increase_num_observable();  // :549
    }
    T.lap("frame's landmark ids");
    const int n_local = (int)local_landmarks.size();
    local_ids_.resize(std::max(n_local, 1));
    last_local_lm_ids_.resize(n_local);
    for (int i = 0; i < n_local; ++i) {
        const auto& lm = local_landmarks[i];
        last_local_lm_ids_[i] = lm->id_;
        bool offered = !(lm->id_ < held_stamp_.size() && held_stamp_[lm->id_] == frame_serial_) && !lm->will_be_erased();  // :560-565
        if (offered && fixed_keyframe_id_threshold > 0) {  // :566-580
            const auto observations = lm->get_observations();
            unsigned int temporal_observations = 0;
            for (const auto& obs : observations) {
                const auto keyfrm = obs.first.lock();
                if (keyfrm && keyfrm->id_ >= fixed_keyframe_id_threshold) ++temporal_observations;
            }
            const double temporal_ratio_thr = 0.5;
            const double temporal_ratio = static_cast<double>(temporal_observations) / observations.size();
            if (temporal_ratio > temporal_ratio_thr) offered = false;
        }
        local_ids_[i] = offered ? (int32_t)lm->id_ : -1;
    }
    match_.resize(std::max(n_local, 1));
    visible_.resize(std::max(n_local, 1));
    outlier_.resize(std::max(n, 1u));
    T.lap("local landmark ids");
    flush_map(ctx_);
    T.lap("flush_map");
    // The pose: the one the first half left on the device ONLY if it is still this frame's pose.  When motion_based_track failed (or was
    // skipped: no valid motion model, the first frame after initialisation) the reference falls back to the BoW / robust trackers, which set
    // the pose through the per-call pose optimizer; the device then holds a stale one (or none at all) and the frame's own pose goes down.
    double pose12_cw[12];
    pose12(curr_frm.get_pose_cw(), pose12_cw);
    const bool device_pose_is_current = device_pose_valid_ && device_pose_frame_ == curr_frm.id_ && std::memcmp(pose12_cw, last_motion_.pose_cw, sizeof pose12_cw) == 0;
    check(svgpu_track_local_map(tracker_, cur_h.get(), cur_ids_.data(), n_local, local_ids_.data(), device_pose_is_current ? nullptr : pose12_cw, margin,
                                lowe_ratio, 0.5f, match_.data(), visible_.data(), outlier_.data(), &last_local_),
          "svgpu_track_local_map");
    device_pose_valid_ = false;  // (the second half's optimisation overwrote it)
    T.lap("svgpu_track_local_map");
    bool found_proj_candidate = false;
    for (int i = 0; i < n_local; ++i)
        if (visible_[i]) {
            local_landmarks[i]->increase_num_observable();  // :588
            found_proj_candidate = true;
        }
    if (!found_proj_candidate) return false;  // :596-599 "projection candidate not found"
    for (int i = 0; i < n_local; ++i)
        if (0 <= match_[i]) {
            curr_frm.add_landmark(local_landmarks[i], (unsigned int)match_[i]);  // projection.cc:88
            cur_ids_[match_[i]] = (int32_t)local_landmarks[i]->id_;
        }
    // optimize_current_frame_with_local_map (:441-455): the pose, then the outliers
    curr_frm.set_pose_cw(pose44(last_local_.pose_cw));
    for (unsigned int idx = 0; idx < n; ++idx)
        if (outlier_[idx] && cur_ids_[idx] >= 0) {
            curr_frm.erase_landmark_with_index(idx);
            cur_ids_[idx] = -1;
        }
    T.lap("observable marks, add_landmark, outliers");
    return true;
}

void tracked_frame_chain::last_observability(eigen_alloc_unord_map<unsigned int, Vec2_t>& lm_to_reproj, std::unordered_map<unsigned int, float>& lm_to_x_right,
                                             std::unordered_map<unsigned int, unsigned int>& lm_to_scale) {
    const int n = (int)last_local_lm_ids_.size();
    std::vector<double> rp((size_t)std::max(n, 1) * 2);
    std::vector<float> xr(std::max(n, 1));
    std::vector<int32_t> lv(std::max(n, 1));
    check(svgpu_track_local_map_observability(tracker_, n, rp.data(), xr.data(), lv.data()), "svgpu_track_local_map_observability");
    lm_to_reproj.clear(), lm_to_x_right.clear(), lm_to_scale.clear();
    for (int i = 0; i < n; ++i)
        if (visible_[i]) {
            Vec2_t q;
            q(0) = rp[2 * (size_t)i], q(1) = rp[2 * (size_t)i + 1];
            lm_to_reproj[last_local_lm_ids_[i]] = q;
            lm_to_x_right[last_local_lm_ids_[i]] = xr[i];
            lm_to_scale[last_local_lm_ids_[i]] = (unsigned int)lv[i];
        }
}

}  // namespace hip
}  // namespace stella_vslam
