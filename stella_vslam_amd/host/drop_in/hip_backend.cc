// Implementation of the drop-in classes: flatten -> one libsvgpu call -> replay on the object graph (hip_backend.h).
#include "hip_backend.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace stella_vslam {

namespace hip {

namespace {
struct ctx_holder {
    svgpu_ctx* ctx = nullptr;
    ~ctx_holder() {
        if (ctx) svgpu_destroy(ctx);
    }
};
}  // namespace

svgpu_ctx* context() {
    thread_local ctx_holder h;
    if (!h.ctx) {
        const char* dev = std::getenv("SVGPU_DEVICE");
        const int rc = svgpu_create(dev ? std::atoi(dev) : 0, &h.ctx);
        if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_create: ") + svgpu_status_string(rc));
    }
    return h.ctx;
}

void check(int status, const char* where) {
    if (status == SVGPU_OK) return;
    throw std::runtime_error(std::string(where) + ": " + svgpu_status_string(status) + " (" + svgpu_last_error(context()) + ")");
}

svgpu_camera to_svgpu_camera(const camera::base* camera) {
    svgpu_camera c;
    std::memset(&c, 0, sizeof(c));
    c.model = static_cast<int32_t>(camera->model_type_);
    c.cols = camera->cols_;
    c.rows = camera->rows_;
    c.focal_x_baseline = camera->focal_x_baseline_;
    switch (camera->model_type_) {
        case camera::model_type_t::Perspective: {
            const auto* p = static_cast<const camera::perspective*>(camera);
            c.fx = p->fx_, c.fy = p->fy_, c.cx = p->cx_, c.cy = p->cy_;
            c.dist[0] = p->k1_, c.dist[1] = p->k2_, c.dist[2] = p->p1_, c.dist[3] = p->p2_, c.dist[4] = p->k3_;
            break;
        }
        case camera::model_type_t::Fisheye: {
            const auto* p = static_cast<const camera::fisheye*>(camera);
            c.fx = p->fx_, c.fy = p->fy_, c.cx = p->cx_, c.cy = p->cy_;
            c.dist[0] = p->k1_, c.dist[1] = p->k2_, c.dist[2] = p->k3_, c.dist[3] = p->k4_;
            break;
        }
        case camera::model_type_t::RadialDivision: {
            const auto* p = static_cast<const camera::radial_division*>(camera);
            c.fx = p->fx_, c.fy = p->fy_, c.cx = p->cx_, c.cy = p->cy_;
            c.dist[0] = p->distortion_;
            break;
        }
        default: break;
    }
    c.min_x = camera->img_bounds_.min_x_;
    c.max_x = camera->img_bounds_.max_x_;
    c.min_y = camera->img_bounds_.min_y_;
    c.max_y = camera->img_bounds_.max_y_;
    return c;
}

#ifndef SVGPU_DROP_IN_OPTIMIZE_ONLY
// ---- resident frame observations
// Thread model (tracking, mapping and loop-closing threads share this cache): an entry is a REFERENCE-COUNTED handle.  A caller keeps its
// handle for the duration of its matcher call, so eviction, forget_*() or a replacement on another thread only drop the cache's own
// reference -- the device arrays live until the last in-flight call has returned.  A stale entry is never re-uploaded in place: a new
// frame is built (outside the lock: an upload is a synchronous copy) and swapped in.
namespace {
static_assert(sizeof(cv::KeyPoint) == sizeof(svgpu_keypoint), "cv::KeyPoint and svgpu_keypoint share the 28-byte layout");
struct resident_entry {
    frame_handle f;
    uint64_t full_hash = 0, kp_hash = 0, stamp = 0;
    const void *desc_ptr = nullptr, *kp_ptr = nullptr;  // host buffers the hash was taken over: the same buffers + size + samples = verified before
    uint64_t sample = 0;
    size_t n = 0;
    bool adopted = false;  // built on the device from an extraction: the first matcher call completes the identity (and the stereo part)
};
struct resident_store {
    std::mutex mtx;
    std::map<std::pair<int, unsigned int>, resident_entry> entries;  // (0 frame | 1 keyframe, id)
    uint64_t clock = 0;
};
resident_store& store() {
    static resident_store s;
    return s;
}
constexpr size_t RESIDENT_CAPACITY = 192;  // ~0.3 MB each: the tracker's frames and the local-map keyframes of a while
bool resident_enabled() {
    static const bool off = std::getenv("SVGPU_NO_RESIDENT_FRAMES") != nullptr;
    return !off;
}
// 8 bytes per step (the arrays are ~150 KB per frame: a byte-wise FNV would cost more than the upload it saves)
uint64_t mix(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    uint64_t tail = 0;
    if (i < n) std::memcpy(&tail, b + i, n - i);
    h = (h ^ tail ^ (uint64_t)n) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 32);
}
uint64_t keypoint_hash(const std::vector<cv::KeyPoint>& k) { return mix(0x243F6A8885A308D3ull, k.data(), k.size() * sizeof(cv::KeyPoint)); }
// identity of an observation under an id: EVERY descriptor and keypoint, the stereo column, the camera bounds and the grid
uint64_t full_hash(const data::frame_observation& o, const camera::base* cam) {
    const size_t n = o.undist_keypts_.size();
    uint64_t h = keypoint_hash(o.undist_keypts_);
    if (n > 0) {
        if (o.descriptors_.isContinuous()) h = mix(h, o.descriptors_.ptr(0), n * 32);
        else
            for (size_t i = 0; i < n; ++i) h = mix(h, o.descriptors_.ptr((int)i), 32);
    }
    h = mix(h, o.stereo_x_right_.data(), o.stereo_x_right_.size() * sizeof(float));
    h = mix(h, &cam->img_bounds_, sizeof cam->img_bounds_);
    h = mix(h, &o.num_grid_cols_, sizeof o.num_grid_cols_);
    return mix(h, &o.num_grid_rows_, sizeof o.num_grid_rows_);
}
// a few samples: together with unchanged buffer addresses and size, "nothing was edited since the full hash was taken"
uint64_t sample_hash(const data::frame_observation& o) {
    const size_t n = o.undist_keypts_.size();
    uint64_t h = 0x13198A2E03707344ull ^ n ^ ((uint64_t)o.stereo_x_right_.size() << 32);
    for (size_t k = 0; k < 4 && n > 0; ++k) {
        const size_t i = (n - 1) * k / 3;
        h = mix(h, o.descriptors_.ptr((int)i), 32);
        h = mix(h, &o.undist_keypts_[i], sizeof(cv::KeyPoint));
    }
    return h;
}
// Frames are recycled: a tracked frame lives for one image, and building a svgpu_frame is three device allocations (and releasing it three
// frees, each a device-wide synchronisation) -- more than the frame's whole matcher chain costs.  The deleter of a handle parks the frame
// in a small pool (grow-only slabs: a recycled frame is as good as a new one), new_frame() takes from it.
struct frame_pool {
    std::mutex mtx;
    std::vector<svgpu_frame*> parked;
};
frame_pool& pool() {
    // never destroyed: handles held by other statics (the resident cache) are released during static destruction and park their frames
    // here, in an order nobody controls; the device memory goes with the process
    static frame_pool* const p = new frame_pool();
    return *p;
}
constexpr size_t FRAME_POOL_CAPACITY = 16;
void park_frame(svgpu_frame* f) {
    if (!f) return;
    frame_pool& P = pool();
    {
        std::lock_guard<std::mutex> lock(P.mtx);
        if (P.parked.size() < FRAME_POOL_CAPACITY) {
            P.parked.push_back(f);
            return;
        }
    }
    svgpu_frame_destroy(f);
}
}  // namespace
frame_handle new_frame(svgpu_ctx* ctx) {
    svgpu_frame* raw = nullptr;
    {
        frame_pool& P = pool();
        std::lock_guard<std::mutex> lock(P.mtx);
        if (!P.parked.empty()) {
            raw = P.parked.back();
            P.parked.pop_back();
        }
    }
    if (!raw) check(svgpu_frame_create(ctx, &raw), "svgpu_frame_create");
    return frame_handle(raw, park_frame);
}
namespace {
frame_handle make_frame(svgpu_ctx* ctx) { return new_frame(ctx); }
void evict_if_full(resident_store& S) {
    while (S.entries.size() > RESIDENT_CAPACITY) {
        auto oldest = S.entries.begin();
        for (auto it = S.entries.begin(); it != S.entries.end(); ++it)
            if (it->second.stamp < oldest->second.stamp) oldest = it;
        S.entries.erase(oldest);  // (drops the cache's reference only)
    }
}
frame_handle resident_of(int kind, unsigned int id, const data::frame_observation& o, const camera::base* cam) {
    if (!resident_enabled() || !cam) return nullptr;
    resident_store& S = store();
    const size_t n = o.undist_keypts_.size();
    const void* const dp = n > 0 ? static_cast<const void*>(o.descriptors_.ptr(0)) : nullptr;
    const void* const kp = static_cast<const void*>(o.undist_keypts_.data());
    const uint64_t smp = sample_hash(o);
    const auto key = std::make_pair(kind, id);
    resident_entry seen;
    {
        std::lock_guard<std::mutex> lock(S.mtx);
        auto it = S.entries.find(key);
        if (it != S.entries.end()) {
            resident_entry& e = it->second;
            e.stamp = ++S.clock;
            if (e.f && !e.adopted && e.n == n && e.desc_ptr == dp && e.kp_ptr == kp && e.sample == smp) return e.f;
            seen = e;
        }
    }
    // slow path, outside the lock: the full identity, then (if it is a different observation) a NEW resident frame
    const uint64_t full = full_hash(o, cam);
    frame_handle f;
    if (seen.f && !seen.adopted && seen.full_hash == full) f = seen.f;  // the same observation in other host buffers (a frame copy)
    else if (seen.f && seen.adopted && (size_t)svgpu_frame_size(seen.f.get()) == n && seen.kp_hash == keypoint_hash(o.undist_keypts_)) {
        // adopted from the extractor: the keypoints are the ones handed back at adoption; the stereo column arrives now
        if (!o.stereo_x_right_.empty()) check(svgpu_frame_set_stereo(context(), seen.f.get(), o.stereo_x_right_.data()), "svgpu_frame_set_stereo");
        f = seen.f;
    }
    else {
        f = make_frame(context());
        const svgpu_camera c = to_svgpu_camera(cam);
        std::vector<uint8_t> desc(n * 32);
        for (size_t i = 0; i < n; ++i) std::memcpy(&desc[i * 32], o.descriptors_.ptr((int)i), 32);
        check(svgpu_frame_upload(context(), f.get(), &c, reinterpret_cast<const svgpu_keypoint*>(o.undist_keypts_.data()), desc.data(),
                                 o.stereo_x_right_.empty() ? nullptr : o.stereo_x_right_.data(), (int)n, (int)o.num_grid_cols_, (int)o.num_grid_rows_),
              "svgpu_frame_upload");
    }
    std::lock_guard<std::mutex> lock(S.mtx);
    resident_entry& e = S.entries[key];
    e.f = f;
    e.full_hash = full, e.sample = smp, e.desc_ptr = dp, e.kp_ptr = kp, e.n = n;
    e.adopted = false;
    e.stamp = ++S.clock;
    evict_if_full(S);
    return f;
}
}  // namespace

frame_handle resident(const data::frame& frm) { return resident_of(0, frm.id_, frm.frm_obs_, frm.camera_); }
frame_handle resident(const std::shared_ptr<data::keyframe>& keyfrm) { return resident_of(1, keyfrm->id_, keyfrm->frm_obs_, keyfrm->camera_); }

void adopt_extraction(unsigned int frame_id, svgpu_ctx* extractor_ctx, const camera::base* camera, unsigned int num_grid_cols, unsigned int num_grid_rows,
                      std::vector<cv::KeyPoint>& undist_keypts, eigen_alloc_vector<Vec3_t>& bearings) {
    const svgpu_camera c = to_svgpu_camera(camera);
    frame_handle f = make_frame(extractor_ctx);
    // worst-case sized host buffers: the extractor's count is only known to the device side here
    const int cap = std::max(1, svgpu_orb_max_keypoints(extractor_ctx));
    std::vector<svgpu_keypoint> und(cap);
    std::vector<double> brg((size_t)cap * 3);
    const int rc = svgpu_frame_adopt_extraction(extractor_ctx, f.get(), &c, (int)num_grid_cols, (int)num_grid_rows, und.data(), brg.data());
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_frame_adopt_extraction: ") + svgpu_last_error(extractor_ctx));
    const int n = svgpu_frame_size(f.get());
    undist_keypts.resize(n);
    std::memcpy(static_cast<void*>(undist_keypts.data()), und.data(), (size_t)n * sizeof(svgpu_keypoint));
    bearings.resize(n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) bearings[i](k) = brg[3 * (size_t)i + k];
    register_adopted(frame_id, f, undist_keypts);
}

void register_adopted(unsigned int frame_id, const frame_handle& f, const std::vector<cv::KeyPoint>& undist_keypts) {
    resident_store& S = store();
    std::lock_guard<std::mutex> lock(S.mtx);
    resident_entry& e = S.entries[std::make_pair(0, frame_id)];
    e = resident_entry();
    e.f = f;
    e.kp_hash = keypoint_hash(undist_keypts);  // the first matcher call sees the caller's host copy of the observation: it must be THIS one
    e.n = undist_keypts.size();
    e.adopted = true;
    e.stamp = ++S.clock;
    evict_if_full(S);
}

void forget_frame(unsigned int frame_id) {
    resident_store& S = store();
    std::lock_guard<std::mutex> lock(S.mtx);
    S.entries.erase(std::make_pair(0, frame_id));
}
void forget_keyframe(unsigned int keyframe_id) {
    resident_store& S = store();
    std::lock_guard<std::mutex> lock(S.mtx);
    S.entries.erase(std::make_pair(1, keyframe_id));
}
#endif  // SVGPU_DROP_IN_OPTIMIZE_ONLY

}  // namespace hip

#if !defined(SVGPU_DROP_IN_OPTIMIZE_ONLY) && !defined(SVGPU_DROP_IN_MATCH_ONLY)
namespace hip {
svgpu_map* flush_map(svgpu_ctx* ctx);  // drop_in/tracking_hip.h
}
#endif

namespace {
using lm_ptr = std::shared_ptr<data::landmark>;
using kf_ptr = std::shared_ptr<data::keyframe>;
}  // namespace

#ifndef SVGPU_DROP_IN_OPTIMIZE_ONLY
namespace {

// ---- flat views ---------------------------------------------------------------------------------------------------------------
struct kp_side {  // the keypoint side of a frame / keyframe (data::frame_observation)
    std::vector<uint8_t> desc;
    std::vector<float> xy, angle, xright;
    std::vector<int32_t> octave;
    std::vector<double> bearings;
    int n = 0;
    const float* xr() const { return xright.empty() ? nullptr : xright.data(); }
};
kp_side flatten(const data::frame_observation& o, bool with_bearings = false) {
    kp_side s;
    s.n = (int)o.undist_keypts_.size();
    s.desc.resize((size_t)s.n * 32);
    s.xy.resize((size_t)s.n * 2);
    s.angle.resize(s.n);
    s.octave.resize(s.n);
    for (int i = 0; i < s.n; ++i) {
        std::memcpy(&s.desc[(size_t)i * 32], o.descriptors_.ptr(i), 32);
        const auto& kp = o.undist_keypts_[i];
        s.xy[2 * i] = kp.pt.x;
        s.xy[2 * i + 1] = kp.pt.y;
        s.angle[i] = kp.angle;
        s.octave[i] = kp.octave;
    }
    if (!o.stereo_x_right_.empty()) s.xright.assign(o.stereo_x_right_.begin(), o.stereo_x_right_.end());
    if (with_bearings) {
        s.bearings.resize((size_t)s.n * 3);
        for (int i = 0; i < s.n; ++i)
            for (int k = 0; k < 3; ++k) s.bearings[3 * (size_t)i + k] = o.bearings_[i](k);
    }
    return s;
}

struct lm_set {  // landmarks as the projection-family entry points take them
    std::vector<double> pos_w, normal;
    std::vector<float> min_d, max_d;
    std::vector<uint8_t> valid, desc;
    int n = 0;
};
template <class Pred>
lm_set flatten(const std::vector<lm_ptr>& lms, Pred&& offered) {
    lm_set s;
    s.n = (int)lms.size();
    s.pos_w.assign((size_t)s.n * 3, 0.0);
    s.normal.assign((size_t)s.n * 3, 0.0);
    s.min_d.assign(s.n, 0.f);
    s.max_d.assign(s.n, 0.f);
    s.valid.assign(s.n, 0);
    s.desc.assign((size_t)s.n * 32, 0);
    for (int i = 0; i < s.n; ++i) {
        const auto& lm = lms[i];
        if (!lm || lm->will_be_erased() || !offered(lm, i)) continue;
        s.valid[i] = 1;
        const Vec3_t p = lm->get_pos_in_world(), nv = lm->get_obs_mean_normal();
        for (int k = 0; k < 3; ++k) {
            s.pos_w[3 * (size_t)i + k] = p(k);
            s.normal[3 * (size_t)i + k] = nv(k);
        }
        s.min_d[i] = lm->get_min_valid_distance();
        s.max_d[i] = lm->get_max_valid_distance();
        const cv::Mat d = lm->get_descriptor();
        if (!d.empty()) std::memcpy(&s.desc[(size_t)i * 32], d.ptr(0), 32);
        else s.valid[i] = 0;
    }
    return s;
}

void rot_rowmajor(const Mat33_t& R, double* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[3 * i + j] = R(i, j);
}
void vec3(const Vec3_t& v, double* out) {
    for (int i = 0; i < 3; ++i) out[i] = v(i);
}

// node id of every keypoint from a bow_feat_vec_ (-1 = in no node)
std::vector<int32_t> node_ids(const data::bow_feature_vector& fv, int n) {
    std::vector<int32_t> node(n, -1);
    for (const auto& kv : fv)
        for (const auto idx : kv.second)
            if ((int)idx < n) node[idx] = (int32_t)kv.first;
    return node;
}

unsigned tri_common(const kf_ptr& keyfrm_1, const kf_ptr& keyfrm_2, const Mat33_t& E_12, std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs,
                    float residual_rad_thr, float lowe_ratio, bool check_orientation, bool with_nodes) {
    svgpu_ctx* ctx = hip::context();
    // epipole of keyframe 1 in keyframe 2 (robust.cc:22-27)
    const svgpu_camera cam2 = hip::to_svgpu_camera(keyfrm_2->camera_);
    double R2[9], t2[3], c1[3], epi[3], E[9];
    rot_rowmajor(keyfrm_2->get_rot_cw(), R2);
    vec3(keyfrm_2->get_trans_cw(), t2);
    vec3(keyfrm_1->get_trans_wc(), c1);
    int valid_epi = 0;
    hip::check(svgpu_reproject_to_bearing(&cam2, R2, t2, c1, epi, &valid_epi), "svgpu_reproject_to_bearing");
    rot_rowmajor(E_12, E);
    const kp_side s1 = flatten(keyfrm_1->frm_obs_, true), s2 = flatten(keyfrm_2->frm_obs_, true);
    const auto lms1 = keyfrm_1->get_landmarks(), lms2 = keyfrm_2->get_landmarks();
    std::vector<uint8_t> has1(s1.n), has2(s2.n);
    for (int i = 0; i < s1.n; ++i) has1[i] = lms1.at(i) ? 1 : 0;
    for (int i = 0; i < s2.n; ++i) has2[i] = lms2.at(i) ? 1 : 0;
    std::vector<int32_t> n1, n2;
    if (with_nodes) {
        n1 = node_ids(keyfrm_1->bow_feat_vec_, s1.n);
        n2 = node_ids(keyfrm_2->bow_feat_vec_, s2.n);
    }
    const auto& sf = keyfrm_1->orb_params_->scale_factors_;
    std::vector<int32_t> m(s1.n, -1);
    int num = 0;
    hip::check(svgpu_match_for_triangulation(ctx, s1.desc.data(), s1.angle.data(), s1.octave.data(), s1.bearings.data(), has1.data(), s1.xr(), s1.n, s2.desc.data(),
                                             s2.angle.data(), s2.bearings.data(), has2.data(), s2.xr(), s2.n, with_nodes ? n1.data() : nullptr,
                                             with_nodes ? n2.data() : nullptr, E, epi, valid_epi, sf.data(), (int)sf.size(), residual_rad_thr, lowe_ratio,
                                             check_orientation ? 1 : 0, m.data(), &num),
               "svgpu_match_for_triangulation");
    matched_idx_pairs.clear();
    matched_idx_pairs.reserve(num);
    for (int i = 0; i < s1.n; ++i)
        if (0 <= m[i]) matched_idx_pairs.emplace_back(std::make_pair((unsigned)i, (unsigned)m[i]));
    return (unsigned)num;
}

}  // namespace

namespace match {
namespace hip {

using stella_vslam::hip::check;
using stella_vslam::hip::context;
using stella_vslam::hip::to_svgpu_camera;

// ------------------------------------------------------------------------------------------------------------------------ robust
unsigned int robust::match_for_triangulation(const kf_ptr& keyfrm_1, const kf_ptr& keyfrm_2, const Mat33_t& E_12,
                                             std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs, const float residual_rad_thr) const {
    return tri_common(keyfrm_1, keyfrm_2, E_12, matched_idx_pairs, residual_rad_thr, lowe_ratio_, check_orientation_, false);
}

unsigned int robust::brute_force_match(const data::frame_observation& frm_obs, const kf_ptr& keyfrm, std::vector<std::pair<int, int>>& matches) const {
    const kp_side s1 = flatten(frm_obs), s2 = flatten(keyfrm->frm_obs_);
    const auto lms_2 = keyfrm->get_landmarks();
    std::vector<uint8_t> valid2(s2.n);
    for (int i = 0; i < s2.n; ++i) valid2[i] = (lms_2.at(i) && !lms_2.at(i)->will_be_erased()) ? 1 : 0;  // robust.cc:258-263
    std::vector<int32_t> m(s1.n, -1);
    int num = 0;
    check(svgpu_match_bruteforce(context(), s1.desc.data(), s1.angle.data(), s1.n, s2.desc.data(), s2.angle.data(), valid2.data(), s2.n, lowe_ratio_,
                                 check_orientation_ ? 1 : 0, m.data(), &num),
          "svgpu_match_bruteforce");
    matches.clear();
    matches.reserve(num);
    for (int idx_1 = 0; idx_1 < s1.n; ++idx_1)
        if (0 <= m[idx_1]) matches.emplace_back(std::make_pair(idx_1, (int)m[idx_1]));  // :316-325
    return (unsigned)num;
}

// The two RANSAC-validated wrappers (match/robust.cc:148-230).  Only the all-pairs distance work is device work; the essential-matrix
// RANSAC is the reference's own solve::essential_solver on the host (seeded as the caller asks: util/random_array.cc:12-23), fed with the
// same (idx_1, idx_2) list in the same order as the reference's brute_force_match produces, so its random 8-point sets pick the same matches.
namespace {
unsigned int assign_inliers(const std::vector<std::pair<int, int>>& matches, const std::vector<bool>* is_inlier, const std::vector<lm_ptr>& keyfrm_lms,
                            std::vector<lm_ptr>& matched_lms_in_frm) {
    unsigned int num_inlier_matches = 0;
    for (unsigned int i = 0; i < matches.size(); ++i) {
        if (is_inlier && !is_inlier->at(i)) continue;
        matched_lms_in_frm.at(matches.at(i).first) = keyfrm_lms.at(matches.at(i).second);  // robust.cc:176-186, :220-226
        ++num_inlier_matches;
    }
    return num_inlier_matches;
}
}  // namespace

unsigned int robust::match_keyframes(const kf_ptr& keyfrm1, const kf_ptr& keyfrm2, std::vector<lm_ptr>& matched_lms_in_frm, bool validate_with_essential_solver,
                                     bool use_fixed_seed) const {
    const auto num_frm_keypts = keyfrm1->frm_obs_.undist_keypts_.size();
    const auto keyfrm_lms = keyfrm2->get_landmarks();
    matched_lms_in_frm = std::vector<lm_ptr>(num_frm_keypts, nullptr);
    std::vector<std::pair<int, int>> matches;
    brute_force_match(keyfrm1->frm_obs_, keyfrm2, matches);
    if (!validate_with_essential_solver) return assign_inliers(matches, nullptr, keyfrm_lms, matched_lms_in_frm);
    solve::essential_solver solver(keyfrm1->frm_obs_.bearings_, keyfrm2->frm_obs_.bearings_, matches, use_fixed_seed);
    solver.find_via_ransac(50, false);  // robust.cc:163
    if (!solver.solution_is_valid()) return 0;
    const auto is_inlier_matches = solver.get_inlier_matches();
    return assign_inliers(matches, &is_inlier_matches, keyfrm_lms, matched_lms_in_frm);
}

unsigned int robust::match_frame_and_keyframe(data::frame& frm, const kf_ptr& keyfrm, std::vector<lm_ptr>& matched_lms_in_frm, bool use_fixed_seed) const {
    const auto num_frm_keypts = frm.frm_obs_.undist_keypts_.size();
    const auto keyfrm_lms = keyfrm->get_landmarks();
    matched_lms_in_frm = std::vector<lm_ptr>(num_frm_keypts, nullptr);
    std::vector<std::pair<int, int>> matches;
    brute_force_match(frm.frm_obs_, keyfrm, matches);
    solve::essential_solver solver(frm.frm_obs_.bearings_, keyfrm->frm_obs_.bearings_, matches, use_fixed_seed);
    solver.find_via_ransac(1000, true);  // robust.cc:209
    if (!solver.solution_is_valid()) return 0;
    const auto is_inlier_matches = solver.get_inlier_matches();
    return assign_inliers(matches, &is_inlier_matches, keyfrm_lms, matched_lms_in_frm);
}

// ---------------------------------------------------------------------------------------------------------------------- bow_tree
unsigned int bow_tree::match_for_triangulation(const kf_ptr& keyfrm_1, const kf_ptr& keyfrm_2, const Mat33_t& E_12,
                                               std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs, const float residual_rad_thr) const {
    return tri_common(keyfrm_1, keyfrm_2, E_12, matched_idx_pairs, residual_rad_thr, lowe_ratio_, check_orientation_, true);
}

unsigned int bow_tree::match_frame_and_keyframe(const kf_ptr& keyfrm, data::frame& frm, std::vector<lm_ptr>& matched_lms_in_frm) const {
    const kp_side s1 = flatten(keyfrm->frm_obs_), s2 = flatten(frm.frm_obs_);
    matched_lms_in_frm = std::vector<lm_ptr>(s2.n, nullptr);
    const auto keyfrm_lms = keyfrm->get_landmarks();
    std::vector<uint8_t> valid1(s1.n);
    for (int i = 0; i < s1.n; ++i) valid1[i] = (keyfrm_lms.at(i) && !keyfrm_lms.at(i)->will_be_erased()) ? 1 : 0;  // bow_tree.cc:191-198
    const auto n1 = node_ids(keyfrm->bow_feat_vec_, s1.n), n2 = node_ids(frm.bow_feat_vec_, s2.n);
    std::vector<int32_t> m(s1.n, -1);
    int num = 0;
    check(svgpu_bow_match(context(), s1.desc.data(), s1.angle.data(), valid1.data(), n1.data(), s1.n, s2.desc.data(), s2.angle.data(), nullptr, n2.data(), s2.n,
                          nullptr, lowe_ratio_, check_orientation_ ? 1 : 0, m.data(), &num),
          "svgpu_bow_match");
    for (int i = 0; i < s1.n; ++i)
        if (0 <= m[i]) matched_lms_in_frm.at(m[i]) = keyfrm_lms.at(i);  // :235
    return (unsigned)num;
}

unsigned int bow_tree::match_keyframes(const kf_ptr& keyfrm_1, const kf_ptr& keyfrm_2, std::vector<lm_ptr>& matched_lms_in_keyfrm_1) const {
    const kp_side s1 = flatten(keyfrm_1->frm_obs_), s2 = flatten(keyfrm_2->frm_obs_);
    const auto lms_1 = keyfrm_1->get_landmarks(), lms_2 = keyfrm_2->get_landmarks();
    matched_lms_in_keyfrm_1 = std::vector<lm_ptr>(lms_1.size(), nullptr);
    std::vector<uint8_t> valid1(s1.n), valid2(s2.n);
    for (int i = 0; i < s1.n; ++i) valid1[i] = (lms_1.at(i) && !lms_1.at(i)->will_be_erased()) ? 1 : 0;  // bow_tree.cc:286-294
    for (int i = 0; i < s2.n; ++i) valid2[i] = (lms_2.at(i) && !lms_2.at(i)->will_be_erased()) ? 1 : 0;  // :305-312
    const auto n1 = node_ids(keyfrm_1->bow_feat_vec_, s1.n), n2 = node_ids(keyfrm_2->bow_feat_vec_, s2.n);
    std::vector<int32_t> m(s1.n, -1);
    int num = 0;
    check(svgpu_bow_match(context(), s1.desc.data(), s1.angle.data(), valid1.data(), n1.data(), s1.n, s2.desc.data(), s2.angle.data(), valid2.data(), n2.data(),
                          s2.n, nullptr, lowe_ratio_, check_orientation_ ? 1 : 0, m.data(), &num),
          "svgpu_bow_match");
    for (int i = 0; i < s1.n; ++i)
        if (0 <= m[i]) matched_lms_in_keyfrm_1.at(i) = lms_2.at(m[i]);  // :343
    return (unsigned)num;
}

// -------------------------------------------------------------------------------------------------------------------- projection
unsigned int projection::match_frame_and_landmarks(data::frame& frm, const std::vector<lm_ptr>& local_landmarks,
                                                   eigen_alloc_unord_map<unsigned int, Vec2_t>& lm_to_reproj, std::unordered_map<unsigned int, float>& lm_to_x_right,
                                                   std::unordered_map<unsigned int, unsigned int>& lm_to_scale, const float margin) const {
    // the caller has already run frame::can_observe (tracking_module.cc:554-594): the queries are its reprojections
    const int n = (int)local_landmarks.size();
    const auto rfh = stella_vslam::hip::resident(frm);  // the frame's keypoint side stays on the device between the matchers of a tracked frame
    const svgpu_frame* const rf = rfh.get();  // (the handle keeps the device arrays alive for the duration of this call, whatever other threads do to the cache)
    kp_side s;
    if (rf) s.n = (int)frm.frm_obs_.undist_keypts_.size();
    else s = flatten(frm.frm_obs_);
    const bool frame_is_stereo = !frm.frm_obs_.stereo_x_right_.empty();
    std::vector<uint8_t> qdesc((size_t)n * 32, 0), qvalid(n, 0), occupied(s.n, 0), qblocks(n, 1);
    std::vector<float> qxy((size_t)n * 2, 0.f), qmargin(n, 0.f), qxr(n, 0.f);
    std::vector<int32_t> qlo(n, 0), qhi(n, 0);
    const auto& sf = frm.orb_params_->scale_factors_;
    for (int i = 0; i < n; ++i) {
        const auto& lm = local_landmarks[i];
        if (!lm_to_reproj.count(lm->id_) || lm->will_be_erased()) continue;  // projection.cc:24-29
        const Vec2_t reproj = lm_to_reproj.at(lm->id_);
        const auto pred = lm_to_scale.at(lm->id_);
        qvalid[i] = 1;
        qxy[2 * i] = (float)reproj(0);
        qxy[2 * i + 1] = (float)reproj(1);
        qmargin[i] = margin * sf.at(pred);
        qlo[i] = std::max(0, static_cast<int>(pred) - 1);
        qhi[i] = std::min((int)frm.orb_params_->num_levels_ - 1, (int)pred + 1);
        qxr[i] = lm_to_x_right.count(lm->id_) ? lm_to_x_right.at(lm->id_) : 0.f;
        qblocks[i] = lm->has_observation() ? 1 : 0;  // a landmark added without observations does not close its keypoint (:52-55 re-reads the frame)
        const cv::Mat d = lm->get_descriptor();
        if (d.empty()) {  // a landmark whose representative descriptor has not been computed yet offers nothing to compare
            qvalid[i] = 0;
            continue;
        }
        std::memcpy(&qdesc[(size_t)i * 32], d.ptr(0), 32);
    }
    for (int k = 0; k < s.n; ++k) {
        const auto lm = frm.get_landmark(k);
        occupied[k] = (lm && lm->has_observation()) ? 1 : 0;  // :52-55
    }
    const auto& b = frm.camera_->img_bounds_;
    std::vector<int32_t> m(n, -1);
    int num = 0;
    check(svgpu_match_set_query_blocks(context(), qblocks.data()), "svgpu_match_set_query_blocks");
    if (rf) check(svgpu_frame_bind(context(), rf), "svgpu_frame_bind");
    check(svgpu_match_in_cells(context(), qdesc.data(), n, qxy.data(), qmargin.data(), qlo.data(), qhi.data(), qvalid.data(), nullptr,
                               frame_is_stereo ? qxr.data() : nullptr, frame_is_stereo ? qmargin.data() : nullptr, rf ? nullptr : s.desc.data(),
                               rf ? nullptr : s.xy.data(), rf ? nullptr : s.octave.data(), s.n, occupied.data(), nullptr, rf ? nullptr : s.xr(), b.min_x_,
                               b.max_x_, b.min_y_, b.max_y_, (int)frm.frm_obs_.num_grid_cols_, (int)frm.frm_obs_.num_grid_rows_, 0, HAMMING_DIST_THR_HIGH,
                               lowe_ratio_, SVGPU_MATCH_RATIO_SAME_OCTAVE, m.data(), &num),
          "svgpu_match_in_cells");
    for (int i = 0; i < n; ++i)
        if (0 <= m[i]) frm.add_landmark(local_landmarks[i], m[i]);  // :88
    return (unsigned)num;
}

unsigned int projection::match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm, const float margin) const {
    const auto rfh = stella_vslam::hip::resident(curr_frm);
    const svgpu_frame* const rf = rfh.get();
    const kp_side sl = flatten(last_frm.frm_obs_);
    kp_side sc;
    if (rf) sc.n = (int)curr_frm.frm_obs_.undist_keypts_.size();
    else sc = flatten(curr_frm.frm_obs_);
    const auto last_lms = last_frm.get_landmarks();
    lm_set L = flatten(last_lms, [](const lm_ptr&, int) { return true; });
    std::vector<uint8_t> has_obs(sl.n, 1), occupied(sc.n, 0);
    for (int i = 0; i < sl.n; ++i)
        if (last_lms[i]) has_obs[i] = last_lms[i]->has_observation() ? 1 : 0;
    for (int k = 0; k < sc.n; ++k) {
        const auto lm = curr_frm.get_landmark(k);
        occupied[k] = (lm && lm->has_observation()) ? 1 : 0;  // projection.cc:167-170
    }
    const svgpu_camera cam = to_svgpu_camera(curr_frm.camera_);
    double Rc[9], tc[3], Rl[9], tl[3];
    rot_rowmajor(curr_frm.get_rot_cw(), Rc);
    vec3(curr_frm.get_trans_cw(), tc);
    rot_rowmajor(last_frm.get_rot_cw(), Rl);
    vec3(last_frm.get_trans_cw(), tl);
    const auto& sf = curr_frm.orb_params_->scale_factors_;
    std::vector<int32_t> m(sl.n, -1);
    int num = 0;
    if (rf) check(svgpu_frame_bind(context(), rf), "svgpu_frame_bind");
    check(svgpu_match_current_and_last_frames(context(), &cam, Rc, tc, Rl, tl, curr_frm.camera_->setup_type_ == camera::setup_type_t::Monocular ? 1 : 0,
                                              (float)curr_frm.camera_->true_baseline_, sl.n, L.pos_w.data(), L.valid.data(), L.desc.data(), sl.octave.data(),
                                              sl.angle.data(), has_obs.data(), (int)sf.size(), sf.data(), margin, sc.desc.data(), sc.xy.data(), sc.octave.data(),
                                              sc.angle.data(), sc.n, occupied.data(), sc.xr(), (int)curr_frm.frm_obs_.num_grid_cols_,
                                              (int)curr_frm.frm_obs_.num_grid_rows_, check_orientation_ ? 1 : 0, m.data(), &num),
          "svgpu_match_current_and_last_frames");
    for (int i = 0; i < sl.n; ++i)
        if (0 <= m[i]) curr_frm.add_landmark(last_lms[i], m[i]);  // :202, in the reference's order (a later landmark may overwrite)
    return (unsigned)num;
}

unsigned int projection::match_frame_and_keyframe(data::frame& curr_frm, const kf_ptr& keyfrm, const std::set<lm_ptr>& already_matched_lms, const float margin,
                                                  const unsigned int hamm_dist_thr) const {
    auto lms = curr_frm.get_landmarks();  // projection.cc:209-215
    const auto rfh = stella_vslam::hip::resident(curr_frm);  // (kept until the matcher below has returned)
    if (rfh) check(svgpu_frame_bind(context(), rfh.get()), "svgpu_frame_bind");
    auto num_matches = match_frame_and_keyframe(curr_frm.get_pose_cw(), curr_frm.camera_, curr_frm.frm_obs_, curr_frm.orb_params_, lms, keyfrm, already_matched_lms,
                                                margin, hamm_dist_thr);
    curr_frm.set_landmarks(lms);
    return num_matches;
}

unsigned int projection::match_frame_and_keyframe(const Mat44_t& cam_pose_cw, const camera::base* camera, const data::frame_observation& frm_obs,
                                                  const feature::orb_params* orb_params, std::vector<lm_ptr>& frm_landmarks, const kf_ptr& keyfrm,
                                                  const std::set<lm_ptr>& already_matched_lms, const float margin, const unsigned int hamm_dist_thr) const {
    const auto landmarks = keyfrm->get_landmarks();
    const lm_set L = flatten(landmarks, [&](const lm_ptr& lm, int) { return already_matched_lms.count(lm) == 0; });  // :237-245
    const kp_side sk = flatten(keyfrm->frm_obs_), sf_ = flatten(frm_obs);
    std::vector<uint8_t> occupied(sf_.n, 0);
    for (int k = 0; k < sf_.n; ++k) occupied[k] = frm_landmarks.at(k) ? 1 : 0;  // :290-292
    const svgpu_camera cam = to_svgpu_camera(camera);
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[3 * i + j] = cam_pose_cw(i, j);
        t[i] = cam_pose_cw(i, 3);
    }
    std::vector<int32_t> m(L.n, -1);
    int num = 0;
    check(svgpu_match_frame_and_keyframe_projection(context(), &cam, R, t, L.n, L.pos_w.data(), L.valid.data(), L.min_d.data(), L.max_d.data(), L.desc.data(),
                                                    sk.angle.data(), (int)orb_params->num_levels_, orb_params->scale_factors_.data(), orb_params->log_scale_factor_,
                                                    margin, hamm_dist_thr, sf_.desc.data(), sf_.xy.data(), sf_.octave.data(), sf_.angle.data(), sf_.n, occupied.data(),
                                                    (int)frm_obs.num_grid_cols_, (int)frm_obs.num_grid_rows_, check_orientation_ ? 1 : 0, m.data(), &num),
          "svgpu_match_frame_and_keyframe_projection");
    for (int i = 0; i < L.n; ++i)
        if (0 <= m[i]) frm_landmarks.at(m[i]) = landmarks.at(i);  // :313
    return (unsigned)num;
}

unsigned int projection::match_by_Sim3_transform(const kf_ptr& keyfrm, const Mat44_t& Sim3_cw, const std::vector<lm_ptr>& landmarks,
                                                 std::vector<lm_ptr>& matched_lms_in_keyfrm, const float margin) const {
    std::set<lm_ptr> already_matched(matched_lms_in_keyfrm.begin(), matched_lms_in_keyfrm.end());  // projection.cc:332-333
    already_matched.erase(nullptr);
    const lm_set L = flatten(landmarks, [&](const lm_ptr& lm, int) { return already_matched.count(lm) == 0; });
    const kp_side s = flatten(keyfrm->frm_obs_);
    const auto rfh = stella_vslam::hip::resident(keyfrm);  // (kept until the matcher below has returned)
    std::vector<uint8_t> occupied(s.n, 0);
    for (int k = 0; k < s.n; ++k) occupied[k] = matched_lms_in_keyfrm.at(k) ? 1 : 0;  // :391-393
    const svgpu_camera cam = to_svgpu_camera(keyfrm->camera_);
    double S[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) S[4 * i + j] = Sim3_cw(i, j);
    const auto* op = keyfrm->orb_params_;
    std::vector<int32_t> m(L.n, -1);
    int num = 0;
    if (rfh) check(svgpu_frame_bind(context(), rfh.get()), "svgpu_frame_bind");  // bound immediately in front of the call that consumes it
    check(svgpu_match_by_sim3_transform(context(), &cam, S, L.n, L.pos_w.data(), L.valid.data(), L.min_d.data(), L.max_d.data(), L.normal.data(), L.desc.data(),
                                        (int)op->num_levels_, op->scale_factors_.data(), op->log_scale_factor_, margin, s.desc.data(), s.xy.data(), s.octave.data(),
                                        s.n, occupied.data(), (int)keyfrm->frm_obs_.num_grid_cols_, (int)keyfrm->frm_obs_.num_grid_rows_, m.data(), &num),
          "svgpu_match_by_sim3_transform");
    for (int i = 0; i < L.n; ++i)
        if (0 <= m[i]) matched_lms_in_keyfrm.at(m[i]) = landmarks[i];  // :410
    return (unsigned)num;
}

unsigned int projection::match_keyframes_mutually(const kf_ptr& keyfrm_1, const kf_ptr& keyfrm_2, std::vector<lm_ptr>& matched_lms_in_keyfrm_1, const float& s_12,
                                                  const Mat33_t& rot_12, const Vec3_t& trans_12, const float margin) const {
    const auto landmarks_1 = keyfrm_1->get_landmarks(), landmarks_2 = keyfrm_2->get_landmarks();
    // matches that already exist between the two keyframes (projection.cc:436-450)
    std::vector<bool> is_already_matched_in_keyfrm_1(landmarks_1.size(), false), is_already_matched_in_keyfrm_2(landmarks_2.size(), false);
    for (unsigned int idx_1 = 0; idx_1 < landmarks_1.size(); ++idx_1) {
        auto& lm = matched_lms_in_keyfrm_1.at(idx_1);
        if (!lm) continue;
        const auto idx_2 = lm->get_index_in_keyframe(keyfrm_2);
        if (0 <= idx_2 && idx_2 < static_cast<int>(landmarks_2.size())) {
            is_already_matched_in_keyfrm_1.at(idx_1) = true;
            is_already_matched_in_keyfrm_2.at(idx_2) = true;
        }
    }
    const lm_set L1 = flatten(landmarks_1, [&](const lm_ptr&, int i) { return !is_already_matched_in_keyfrm_1[i]; });
    const lm_set L2 = flatten(landmarks_2, [&](const lm_ptr&, int i) { return !is_already_matched_in_keyfrm_2[i]; });
    const kp_side s1 = flatten(keyfrm_1->frm_obs_), s2 = flatten(keyfrm_2->frm_obs_);
    const svgpu_camera cam1 = to_svgpu_camera(keyfrm_1->camera_), cam2 = to_svgpu_camera(keyfrm_2->camera_);
    double R1[9], t1[3], R2[9], t2[3], R12[9], t12[3];
    rot_rowmajor(keyfrm_1->get_rot_cw(), R1);
    vec3(keyfrm_1->get_trans_cw(), t1);
    rot_rowmajor(keyfrm_2->get_rot_cw(), R2);
    vec3(keyfrm_2->get_trans_cw(), t2);
    rot_rowmajor(rot_12, R12);
    vec3(trans_12, t12);
    const auto* op = keyfrm_2->orb_params_;
    std::vector<int32_t> m21(L1.n, -1), m12(L2.n, -1), mut(L1.n, -1);
    int num = 0;
    check(svgpu_match_keyframes_mutually(context(), &cam1, &cam2, R1, t1, R2, t2, s_12, R12, t12, L1.n, L1.pos_w.data(), L1.valid.data(), L1.min_d.data(),
                                         L1.max_d.data(), L1.desc.data(), s1.desc.data(), s1.xy.data(), s1.octave.data(), L2.n, L2.pos_w.data(), L2.valid.data(),
                                         L2.min_d.data(), L2.max_d.data(), L2.desc.data(), s2.desc.data(), s2.xy.data(), s2.octave.data(), (int)op->num_levels_,
                                         op->scale_factors_.data(), op->log_scale_factor_, margin, (int)keyfrm_2->frm_obs_.num_grid_cols_,
                                         (int)keyfrm_2->frm_obs_.num_grid_rows_, m21.data(), m12.data(), mut.data(), &num),
          "svgpu_match_keyframes_mutually");
    for (int i = 0; i < L1.n; ++i)
        if (0 <= mut[i]) matched_lms_in_keyfrm_1.at(i) = landmarks_2.at(mut[i]);  // :606
    return (unsigned)num;
}

// -------------------------------------------------------------------------------------------------------------------------- fuse
template <typename T>
unsigned int fuse::detect_duplication(const kf_ptr& keyfrm, const Mat33_t& rot_cw, const Vec3_t& trans_cw, const T& landmarks_to_check, const float margin,
                                      std::unordered_map<lm_ptr, lm_ptr>& duplicated_lms_in_keyfrm, std::unordered_map<unsigned int, lm_ptr>& new_connections,
                                      bool do_reprojection_matching) const {
    duplicated_lms_in_keyfrm.clear();
    const std::vector<lm_ptr> lms(landmarks_to_check.begin(), landmarks_to_check.end());  // the container's own iteration order
    const lm_set L = flatten(lms, [&](const lm_ptr& lm, int) { return !lm->is_observed_in_keyframe(keyfrm); });  // fuse.cc:27-35
    const kp_side s = flatten(keyfrm->frm_obs_);
    const svgpu_camera cam = to_svgpu_camera(keyfrm->camera_);
    double R[9], t[3];
    rot_rowmajor(rot_cw, R);
    vec3(trans_cw, t);
    const auto* op = keyfrm->orb_params_;
    std::vector<int32_t> best(L.n, -1);
    int num = 0;
    check(svgpu_fuse_detect_duplication(context(), &cam, R, t, L.n, L.pos_w.data(), L.valid.data(), L.min_d.data(), L.max_d.data(), L.normal.data(), L.desc.data(),
                                        (int)op->num_levels_, op->scale_factors_.data(), op->inv_level_sigma_sq_.data(), op->log_scale_factor_, margin,
                                        do_reprojection_matching ? 1 : 0, s.desc.data(), s.xy.data(), s.octave.data(), s.xr(), s.n,
                                        (int)keyfrm->frm_obs_.num_grid_cols_, (int)keyfrm->frm_obs_.num_grid_rows_, best.data(), &num),
          "svgpu_fuse_detect_duplication");
    for (int i = 0; i < L.n; ++i) {  // :130-147
        if (best[i] < 0) continue;
        auto lm_in_keyfrm = keyfrm->get_landmark(best[i]);
        if (lm_in_keyfrm) {
            if (!lm_in_keyfrm->will_be_erased()) duplicated_lms_in_keyfrm[lms[i]] = lm_in_keyfrm;
        }
        else new_connections.emplace((unsigned)best[i], lms[i]);
    }
    return (unsigned)num;
}
template unsigned int fuse::detect_duplication(const kf_ptr&, const Mat33_t&, const Vec3_t&, const std::vector<lm_ptr>&, const float, std::unordered_map<lm_ptr, lm_ptr>&,
                                               std::unordered_map<unsigned int, lm_ptr>&, bool) const;
template unsigned int fuse::detect_duplication(const kf_ptr&, const Mat33_t&, const Vec3_t&, const std::unordered_set<lm_ptr>&, const float,
                                               std::unordered_map<lm_ptr, lm_ptr>&, std::unordered_map<unsigned int, lm_ptr>&, bool) const;
#ifdef SVGPU_WITH_STELLA_VSLAM
template unsigned int fuse::detect_duplication(const kf_ptr&, const Mat33_t&, const Vec3_t&, const id_ordered_set<lm_ptr>&, const float,
                                               std::unordered_map<lm_ptr, lm_ptr>&, std::unordered_map<unsigned int, lm_ptr>&, bool) const;
#endif

// -------------------------------------------------------------------------------------------------------------------------- area
unsigned int area::match_in_consistent_area(data::frame& frm_1, data::frame& frm_2, std::vector<cv::Point2f>& prev_matched_pts,
                                            std::vector<int>& matched_indices_2_in_frm_1, int margin) {
    const kp_side s1 = flatten(frm_1.frm_obs_), s2 = flatten(frm_2.frm_obs_);
    matched_indices_2_in_frm_1 = std::vector<int>(s1.n, -1);
    std::vector<uint8_t> qvalid(s1.n);
    std::vector<float> qxy((size_t)s1.n * 2), qmargin(s1.n, (float)margin);
    std::vector<int32_t> lvl(s1.n, 0);
    for (int i = 0; i < s1.n; ++i) {
        qvalid[i] = s1.octave[i] > 0 ? 0 : 1;  // area.cc:21-25: level-0 keypoints only
        qxy[2 * i] = prev_matched_pts.at(i).x;
        qxy[2 * i + 1] = prev_matched_pts.at(i).y;
        lvl[i] = s1.octave[i];
    }
    const auto& b = frm_2.camera_->img_bounds_;
    std::vector<int32_t> m(s1.n, -1);
    int num = 0;
    check(svgpu_match_in_cells(context(), s1.desc.data(), s1.n, qxy.data(), qmargin.data(), lvl.data(), lvl.data(), qvalid.data(), s1.angle.data(), nullptr, nullptr,
                               s2.desc.data(), s2.xy.data(), s2.octave.data(), s2.n, nullptr, s2.angle.data(), nullptr, b.min_x_, b.max_x_, b.min_y_, b.max_y_,
                               (int)frm_2.frm_obs_.num_grid_cols_, (int)frm_2.frm_obs_.num_grid_rows_, check_orientation_ ? 1 : 0, HAMMING_DIST_THR_LOW, lowe_ratio_,
                               SVGPU_MATCH_AREA, m.data(), &num),
          "svgpu_match_in_cells");
    for (int i = 0; i < s1.n; ++i) {
        matched_indices_2_in_frm_1[i] = m[i];
        if (0 <= m[i]) prev_matched_pts.at(i) = frm_2.frm_obs_.undist_keypts_.at(m[i]).pt;  // :91-95
    }
    return (unsigned)num;
}

}  // namespace hip
}  // namespace match
#endif  // SVGPU_DROP_IN_OPTIMIZE_ONLY

// ====================================================================================================================== local BA
#ifndef SVGPU_DROP_IN_MATCH_ONLY
namespace optimize {

local_bundle_adjuster_hip::local_bundle_adjuster_hip(const YAML::Node& yaml_node, const unsigned int num_first_iter, const unsigned int num_second_iter)
    : num_first_iter_(num_first_iter), num_second_iter_(num_second_iter),
      use_additional_keyframes_for_monocular_(yaml_node["use_additional_keyframes_for_monocular"].as<bool>(false)) {}

void local_bundle_adjuster_hip::optimize(data::map_database* map_db, const kf_ptr& curr_keyfrm, bool* const force_stop_flag) const {
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](int phase) {
        const auto n = std::chrono::steady_clock::now();
        last_phase_ms_[phase] = std::chrono::duration<double, std::milli>(n - lap_t).count();
        lap_t = n;
    };
    for (double& v : last_phase_ms_) v = 0.0;
    // 1. Aggregate the local and fixed keyframes, and local landmarks (local_bundle_adjuster_g2o.cc:38-147; id-ordered maps: the
    //    reference's unordered_map order is unspecified, so only the mathematical result is comparable)
    std::map<unsigned int, kf_ptr> local_keyfrms;
    bool has_scale = false;
    local_keyfrms[curr_keyfrm->id_] = curr_keyfrm;
    for (const auto& local_keyfrm : curr_keyfrm->graph_node_->get_covisibilities()) {
        if (!local_keyfrm || local_keyfrm->will_be_erased() || local_keyfrm->graph_node_->is_spanning_root()) continue;
        if (local_keyfrm->id_ < map_db->get_fixed_keyframe_id_threshold()) continue;
        local_keyfrms[local_keyfrm->id_] = local_keyfrm;
        if (local_keyfrm->camera_->setup_type_ != camera::setup_type_t::Monocular) has_scale = true;
    }
    // (id-ordered like the map above, but as a sorted vector: ten thousand tree insertions cost a millisecond of the mapping thread; the
    //  observations of every landmark are copied ONCE here -- get_observations() returns the map by value -- and reused by the flattening below)
    std::vector<std::pair<unsigned int, lm_ptr>> local_lms;
    for (const auto& id_kf : local_keyfrms)
        for (const auto& local_lm : id_kf.second->get_landmarks()) {
            if (!local_lm || local_lm->will_be_erased()) continue;
            local_lms.emplace_back(local_lm->id_, local_lm);
        }
    std::sort(local_lms.begin(), local_lms.end(), [](const std::pair<unsigned int, lm_ptr>& a, const std::pair<unsigned int, lm_ptr>& b) { return a.first < b.first; });
    local_lms.erase(std::unique(local_lms.begin(), local_lms.end(), [](const std::pair<unsigned int, lm_ptr>& a, const std::pair<unsigned int, lm_ptr>& b) { return a.first == b.first; }),
                    local_lms.end());
    // every landmark's observations, flat: the keyframe (locked once, nullptr if it has expired) and the keypoint index of observation o of
    // landmark k at all_obs[lm_obs_first[k] .. lm_obs_first[k + 1]), in the order of the landmark's observations_ map.  get_observations() returns
    // the map BY VALUE; the copy lives for one loop trip (round 5 kept ten thousand of them for the flattening below and walked them twice).
    struct obs_rec {
        kf_ptr keyfrm;
        unsigned int idx;
    };
    std::vector<obs_rec> all_obs;
    std::vector<int> lm_obs_first(local_lms.size() + 1, 0);
    all_obs.reserve(8 * local_lms.size());
    std::map<unsigned int, std::shared_ptr<data::marker>> local_mkrs;  // :86-102
    for (const auto& id_kf : local_keyfrms)
        for (const auto& local_mkr : id_kf.second->get_markers()) {
            if (!local_mkr) continue;
            local_mkrs.emplace(local_mkr->id_, local_mkr);
        }
    std::map<unsigned int, kf_ptr> fixed_keyfrms;
    for (size_t k = 0; k < local_lms.size(); ++k) {
        lm_obs_first[k] = (int)all_obs.size();
        for (const auto& obs : local_lms[k].second->get_observations()) {
            all_obs.push_back(obs_rec{obs.first.lock(), obs.second});
            const auto& fixed_keyfrm = all_obs.back().keyfrm;
            if (!fixed_keyfrm || fixed_keyfrm->will_be_erased()) continue;
            if (local_keyfrms.count(fixed_keyfrm->id_)) continue;
            fixed_keyfrms.emplace(fixed_keyfrm->id_, fixed_keyfrm);
        }
    }
    lm_obs_first[local_lms.size()] = (int)all_obs.size();
    if (use_additional_keyframes_for_monocular_) {  // :135-147
        auto additional_keyfrms_size = 2 - fixed_keyfrms.size();
        if (!has_scale && fixed_keyfrms.size() < 2 && local_keyfrms.size() > additional_keyfrms_size) {
            for (unsigned int i = 0; i < additional_keyfrms_size; ++i) {
                auto itr = local_keyfrms.begin();
                auto keyfrm_id = itr->first;
                auto keyfrm = itr->second;
                local_keyfrms.erase(keyfrm_id);
                fixed_keyfrms[keyfrm_id] = keyfrm;
            }
        }
    }
    if (force_stop_flag && *force_stop_flag) return;  // :308-310 (nothing has been touched yet)
    lap(0);

    // 2.-4. vertices and reprojection edges as flat arrays (:165-246)
    std::vector<kf_ptr> poses;
    std::unordered_map<unsigned int, int> pose_index;
    std::vector<uint8_t> pose_fixed;
    for (const auto& kv : local_keyfrms) {
        pose_index[kv.first] = (int)poses.size();
        poses.push_back(kv.second);
        pose_fixed.push_back(0);
    }
    for (const auto& kv : fixed_keyfrms) {
        pose_index[kv.first] = (int)poses.size();
        poses.push_back(kv.second);
        pose_fixed.push_back(1);
    }
    const int P = (int)poses.size();
    std::vector<double> pose_cw((size_t)P * 12), intr((size_t)P * 5);
    for (int p = 0; p < P; ++p) {
        const Mat44_t T = poses[p]->get_pose_cw();
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) pose_cw[(size_t)p * 12 + 4 * i + j] = T(i, j);
        const svgpu_camera c = hip::to_svgpu_camera(poses[p]->camera_);
        double* K = &intr[(size_t)p * 5];
        if (c.model == SVGPU_CAM_EQUIRECTANGULAR) K[0] = 0, K[1] = 0, K[2] = c.cols, K[3] = c.rows, K[4] = 0;  // equirectangular_reproj_edge.h
        else K[0] = c.fx, K[1] = c.fy, K[2] = c.cx, K[3] = c.cy, K[4] = c.focal_x_baseline;
    }
    constexpr float chi_sq_2D = 5.99146;
    const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D);
    constexpr float chi_sq_3D = 7.81473;
    const float sqrt_chi_sq_3D = std::sqrt(chi_sq_3D);
    std::vector<lm_ptr> points;
    std::vector<double> pts;
    std::vector<int32_t> obs_pose, obs_point;
    std::vector<float> obs_uvr, obs_w, obs_huber;
    std::vector<std::pair<kf_ptr, lm_ptr>> obs_objects;
    std::vector<unsigned int> obs_kpidx;      // keypoint index of the observation in its keyframe
    std::vector<int> lm_edge_first;           // edges of landmark l = [lm_edge_first[l], lm_edge_first[l + 1]) (appended landmark by landmark, in observations_ order)
    std::vector<uint8_t> lm_partial;          // an observation of l was NOT turned into an edge (expired / erased keyframe): the batched refresh leaves l to the object's own methods
    {
        const size_t n_obs = all_obs.size();
        points.reserve(local_lms.size()), pts.reserve(3 * local_lms.size());
        obs_pose.reserve(n_obs), obs_point.reserve(n_obs), obs_uvr.reserve(3 * n_obs), obs_w.reserve(n_obs), obs_huber.reserve(n_obs), obs_objects.reserve(n_obs);
        obs_kpidx.reserve(n_obs), lm_edge_first.reserve(local_lms.size() + 1), lm_partial.reserve(local_lms.size());
    }
    for (size_t k_lm = 0; k_lm < local_lms.size(); ++k_lm) {
        const auto& local_lm = local_lms[k_lm].second;
        if (lm_obs_first[k_lm] == lm_obs_first[k_lm + 1]) continue;
        const int l = (int)points.size();
        points.push_back(local_lm);
        lm_edge_first.push_back((int)obs_pose.size());
        lm_partial.push_back(0);
        const Vec3_t pw = local_lm->get_pos_in_world();
        for (int k = 0; k < 3; ++k) pts.push_back(pw(k));
        for (int o = lm_obs_first[k_lm]; o < lm_obs_first[k_lm + 1]; ++o) {
            const auto& keyfrm = all_obs[o].keyfrm;
            const auto idx = all_obs[o].idx;
            if (!keyfrm || keyfrm->will_be_erased()) {
                lm_partial.back() = 1;
                continue;
            }
            const auto it = pose_index.find(keyfrm->id_);
            if (it == pose_index.end()) {
                lm_partial.back() = 1;
                continue;
            }
            const auto& undist_keypt = keyfrm->frm_obs_.undist_keypts_.at(idx);
            const float x_right = keyfrm->frm_obs_.stereo_x_right_.empty() ? -1.0f : keyfrm->frm_obs_.stereo_x_right_.at(idx);
            obs_pose.push_back(it->second);
            obs_point.push_back(l);
            obs_uvr.push_back(undist_keypt.pt.x);
            obs_uvr.push_back(undist_keypt.pt.y);
            obs_uvr.push_back(x_right);
            obs_w.push_back(keyfrm->orb_params_->inv_level_sigma_sq_.at(undist_keypt.octave));
            obs_huber.push_back(keyfrm->camera_->setup_type_ == camera::setup_type_t::Monocular ? sqrt_chi_sq_2D : sqrt_chi_sq_3D);
            obs_objects.emplace_back(keyfrm, local_lm);
            obs_kpidx.push_back(idx);
        }
    }
    lm_edge_first.push_back((int)obs_pose.size());
    // marker corners (:246-304): four points per marker that was initialised before or is kept fixed (then fixed vertices), one edge per
    // observing keyframe of the graph and corner, information 1, no kernel -- and outside the gate / outlier list (negative width, svgpu.h)
    const int L_lm = (int)points.size();
    std::vector<uint8_t> point_fixed((size_t)L_lm, 0);
    std::vector<std::pair<std::shared_ptr<data::marker>, int>> marker_slots;
    for (const auto& id_mkr : local_mkrs) {
        const auto& mkr = id_mkr.second;
        if (!mkr->keep_fixed_ && !mkr->initialized_before_) continue;
        const int first = (int)point_fixed.size();
        marker_slots.emplace_back(mkr, first);
        for (unsigned int corner_idx = 0; corner_idx < 4; ++corner_idx) {
            point_fixed.push_back(mkr->keep_fixed_ ? 1 : 0);
            const Vec3_t pw = mkr->corners_pos_w_.at(corner_idx);
            for (int k = 0; k < 3; ++k) pts.push_back(pw(k));
            for (const auto& id_keyfrm : mkr->observations_) {
                const auto& keyfrm = id_keyfrm.second;
                if (!keyfrm || keyfrm->will_be_erased()) continue;
                const auto it = pose_index.find(keyfrm->id_);
                if (it == pose_index.end()) continue;
                const auto& undist_pt = keyfrm->markers_2d_.at(mkr->id_).undist_corners_.at(corner_idx);
                obs_pose.push_back(it->second);
                obs_point.push_back(first + (int)corner_idx);
                obs_uvr.push_back(undist_pt.x);
                obs_uvr.push_back(undist_pt.y);
                obs_uvr.push_back(-1.0f);
                obs_w.push_back(1.0f);
                obs_huber.push_back(-1.0f);
            }
        }
    }
    const int L = (int)point_fixed.size(), E = (int)obs_pose.size();

    // 5.-6. the two-stage Levenberg-Marquardt schedule on the device (:306-348)
    svgpu_ba_problem pr;
    std::memset(&pr, 0, sizeof(pr));
    pr.num_poses = P, pr.num_points = L, pr.num_obs = E;
    pr.pose_cw = pose_cw.data(), pr.pose_fixed = pose_fixed.data(), pr.points = pts.data(), pr.point_fixed = point_fixed.data();
    pr.obs_pose = obs_pose.data(), pr.obs_point = obs_point.data(), pr.obs_uvr = obs_uvr.data(), pr.obs_inv_sigma_sq = obs_w.data(), pr.obs_huber_delta = obs_huber.data();
    pr.intrinsics = intr.data();
    pr.num_first_iter = (int)num_first_iter_, pr.num_second_iter = (int)num_second_iter_;
    pr.gain_threshold = 1e-3;
    std::vector<double> pose_out((size_t)P * 12), pts_out((size_t)L * 3);
    std::vector<uint8_t> outlier(E > 0 ? E : 1, 0);
    static_assert(sizeof(bool) == 1, "force_stop_flag is polled as one byte");
    lap(1);
    last_status_ = svgpu_local_ba(hip::context(), &pr, reinterpret_cast<volatile uint8_t*>(force_stop_flag), pose_out.data(), pts_out.data(), outlier.data(), &last_stats_);
    lap(2);
    if (last_status_ == SVGPU_STOPPED) return;
    hip::check(last_status_, "svgpu_local_ba");

    // 7.-8. outlier observations, then poses and positions under the map mutex (:352-411)
    {
        std::lock_guard<std::mutex> lock(data::map_database::mtx_database_);
#ifdef SVGPU_LANDMARK_HAS_BATCH_SETTERS
        // The reference refreshes every touched landmark through its own methods (compute_descriptor after an erased observation,
        // update_mean_normal_and_obs_scale_variance after a moved position: data/landmark.cc:199-318) -- ten thousand calls that each copy
        // the observation map and lock every observing keyframe.  Here the graph mutations stay per object, the refreshes are ONE call of
        // svgpu_landmarks_update_geometry (+ one of svgpu_landmarks_compute_descriptor for the landmarks that lost an observation) on the flat
        // observation lists the flattening above already holds, and the results are stored through two setters.
        std::vector<uint8_t> edge_erased(obs_objects.size(), 0), desc_dirty((size_t)L_lm, 0);
        for (int e = 0; e < (int)obs_objects.size(); ++e) {
            if (!outlier[e]) continue;
            const auto& keyfrm = obs_objects[e].first;
            const auto& lm = obs_objects[e].second;
            if (lm->will_be_erased()) continue;  // :358-361
            keyfrm->erase_landmark(lm);
            lm->erase_observation(map_db, keyfrm);
            edge_erased[e] = 1;
            desc_dirty[obs_point[e]] = 1;
        }
        std::vector<double> centre((size_t)P * 3);  // camera centres -R^T t of the window's keyframes after the solve (fixed ones: unchanged)
        for (int p = 0; p < P; ++p) {
            const double* T = pose_fixed[p] ? &pose_cw[(size_t)p * 12] : &pose_out[(size_t)p * 12];
            for (int k = 0; k < 3; ++k) centre[(size_t)p * 3 + k] = -(T[k] * T[3] + T[4 + k] * T[7] + T[8 + k] * T[11]);
        }
        for (const auto& kv : local_keyfrms) {
            const int p = pose_index.at(kv.first);
            Mat44_t T = Mat44_t::Identity();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) T(i, j) = pose_out[(size_t)p * 12 + 4 * i + j];
            kv.second->set_pose_cw(T);
        }
        const auto* orb0 = curr_keyfrm->orb_params_;
        const float inv_last = orb0->inv_scale_factors_.at(orb0->num_levels_ - 1);
        std::vector<int> g_lm, d_lm;                 // landmarks of the geometry batch / of the descriptor batch
        std::vector<int32_t> g_off(1, 0), d_off(1, 0);
        std::vector<double> g_trans, g_pos, g_ref;
        std::vector<float> g_scale;
        std::vector<uint8_t> d_rows;
        g_lm.reserve(L_lm), g_off.reserve(L_lm + 1), g_trans.reserve(3 * obs_objects.size()), g_pos.reserve(3 * (size_t)L_lm), g_ref.reserve(3 * (size_t)L_lm), g_scale.reserve(L_lm);
        for (int l = 0; l < L_lm; ++l) {
            const auto& local_lm = points[l];
            if (local_lm->will_be_erased()) continue;
            Vec3_t pw;
            for (int k = 0; k < 3; ++k) pw(k) = pts_out[(size_t)l * 3 + k];
            local_lm->set_pos_in_world(pw);
            // the landmark's observation list as it is NOW = its edges minus the erased ones; its reference keyframe must be among them
            const auto ref = local_lm->get_ref_keyframe();
            int n_kept = 0, ref_edge = -1;
            for (int e = lm_edge_first[l]; e < lm_edge_first[l + 1]; ++e) {
                if (edge_erased[e]) continue;
                ++n_kept;
                if (obs_objects[e].first == ref) ref_edge = e;
            }
            if (lm_partial[l] || n_kept == 0 || ref_edge < 0 || ref->orb_params_->inv_scale_factors_.at(ref->orb_params_->num_levels_ - 1) != inv_last) {
                if (desc_dirty[l]) local_lm->compute_descriptor();  // the object's own methods for the odd ones
                local_lm->update_mean_normal_and_obs_scale_variance();
                continue;
            }
            for (int e = lm_edge_first[l]; e < lm_edge_first[l + 1]; ++e)
                if (!edge_erased[e])
                    for (int k = 0; k < 3; ++k) g_trans.push_back(centre[(size_t)obs_pose[e] * 3 + k]);
            g_off.push_back(g_off.back() + n_kept);
            for (int k = 0; k < 3; ++k) g_pos.push_back(pw(k)), g_ref.push_back(centre[(size_t)obs_pose[ref_edge] * 3 + k]);
            g_scale.push_back(ref->orb_params_->scale_factors_.at(ref->frm_obs_.undist_keypts_.at(obs_kpidx[ref_edge]).octave));
            g_lm.push_back(l);
            if (desc_dirty[l]) {
                for (int e = lm_edge_first[l]; e < lm_edge_first[l + 1]; ++e)
                    if (!edge_erased[e]) {
                        const unsigned char* row = obs_objects[e].first->frm_obs_.descriptors_.ptr((int)obs_kpidx[e]);
                        d_rows.insert(d_rows.end(), row, row + 32);
                    }
                d_off.push_back(d_off.back() + n_kept);
                d_lm.push_back(l);
            }
        }
        if (!g_lm.empty()) {
            const int n = (int)g_lm.size();
            std::vector<double> mean_normal((size_t)n * 3);
            std::vector<float> max_d(n), min_d(n);
            hip::check(svgpu_landmarks_update_geometry(hip::context(), n, g_off.data(), g_trans.data(), g_pos.data(), g_ref.data(), g_scale.data(), inv_last,
                                                       mean_normal.data(), max_d.data(), min_d.data()),
                       "svgpu_landmarks_update_geometry");
            for (int i = 0; i < n; ++i) {
                Vec3_t mn;
                for (int k = 0; k < 3; ++k) mn(k) = mean_normal[(size_t)i * 3 + k];
                points[g_lm[i]]->set_prediction_parameters(mn, min_d[i], max_d[i]);
            }
        }
        if (!d_lm.empty()) {
            const int n = (int)d_lm.size();
            std::vector<int32_t> best(n);
            std::vector<uint8_t> rep((size_t)n * 32);
            hip::check(svgpu_landmarks_compute_descriptor(hip::context(), n, d_off.data(), d_rows.data(), best.data(), rep.data()), "svgpu_landmarks_compute_descriptor");
            for (int i = 0; i < n; ++i) points[d_lm[i]]->set_representative_descriptor(&rep[(size_t)i * 32]);
        }
#else
        for (int e = 0; e < (int)obs_objects.size(); ++e) {  // (landmark edges come first; marker edges are never outliers)
            if (!outlier[e]) continue;
            const auto& keyfrm = obs_objects[e].first;
            const auto& lm = obs_objects[e].second;
            if (lm->will_be_erased()) continue;  // :358-361
            keyfrm->erase_landmark(lm);
            lm->erase_observation(map_db, keyfrm);
            if (!lm->will_be_erased()) {
                lm->compute_descriptor();
                lm->update_mean_normal_and_obs_scale_variance();
            }
        }
        for (const auto& kv : local_keyfrms) {
            const int p = pose_index.at(kv.first);
            Mat44_t T = Mat44_t::Identity();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) T(i, j) = pose_out[(size_t)p * 12 + 4 * i + j];
            kv.second->set_pose_cw(T);
        }
        for (int l = 0; l < L_lm; ++l) {
            const auto& local_lm = points[l];
            if (local_lm->will_be_erased()) continue;
            Vec3_t pw;
            for (int k = 0; k < 3; ++k) pw(k) = pts_out[(size_t)l * 3 + k];
            local_lm->set_pos_in_world(pw);
            local_lm->update_mean_normal_and_obs_scale_variance();
        }
#endif
        for (const auto& mk_slot : marker_slots) {  // :411-428
            const auto& mkr = mk_slot.first;
            if (mkr->keep_fixed_ || !mkr->initialized_before_) continue;
            for (int corner_idx = 0; corner_idx < 4; ++corner_idx)
                for (int k = 0; k < 3; ++k) mkr->corners_pos_w_[corner_idx](k) = pts_out[(size_t)(mk_slot.second + corner_idx) * 3 + k];
        }
    }
    lap(3);
#if !defined(SVGPU_DROP_IN_OPTIMIZE_ONLY) && !defined(SVGPU_DROP_IN_MATCH_ONLY)
    // the write-back has moved a few thousand landmarks (set_pos_in_world, update_mean_normal_and_obs_scale_variance, compute_descriptor:
    // drop_in/map_mirror.h): the device-resident landmark table is brought up to date HERE, on the mapping thread, so that the tracking
    // thread's next frame finds nothing left to upload
    hip::flush_map(hip::context());
    lap(4);
#endif
}

namespace hip_backend {
std::unique_ptr<local_bundle_adjuster> create_local_bundle_adjuster(const YAML::Node& yaml_node) {
    const auto& backend = yaml_node["backend"].as<std::string>("g2o");
    if (backend == "hip") return std::unique_ptr<local_bundle_adjuster>(new local_bundle_adjuster_hip(yaml_node));
    return nullptr;
}
}  // namespace hip_backend

}  // namespace optimize
#endif  // SVGPU_DROP_IN_MATCH_ONLY
}  // namespace stella_vslam
