// Drop-in classes with the reference's own signatures, backed by libsvgpu (C ABI, include/svgpu.h).
//   stella_vslam::match::hip::{robust, bow_tree, projection, fuse, area}      match/robust.h, bow_tree.h, projection.h, fuse.h, area.h
//   stella_vslam::optimize::local_bundle_adjuster_hip                          optimize/local_bundle_adjuster.h:15-24 (abstract base)
//   stella_vslam::optimize::hip_backend::create_local_bundle_adjuster          the line local_bundle_adjuster_factory::create gains
// They take the reference's object graph (data::frame&, std::shared_ptr<data::keyframe>, data::landmark, data::map_database*),
// flatten exactly what the arithmetic reads, call ONE device entry point per method and replay the results onto the objects in the
// reference's order -- so tracking_module / mapping_module / the loop detector call them unchanged.
// Compiles against the reference tree with -DSVGPU_WITH_STELLA_VSLAM, or stand-alone against host/standin/stella_standin.h.
#pragma once
#ifdef SVGPU_WITH_STELLA_VSLAM
#include "stella_vslam/camera/base.h"
#include "stella_vslam/camera/equirectangular.h"
#include "stella_vslam/camera/fisheye.h"
#include "stella_vslam/camera/perspective.h"
#include "stella_vslam/camera/radial_division.h"
#include "stella_vslam/data/frame.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"
#ifndef SVGPU_DROP_IN_OPTIMIZE_ONLY
#include "stella_vslam/solve/essential_solver.h"  // the RANSAC behind robust::match_frame_and_keyframe / match_keyframes stays the reference's (host)
#endif
#ifndef SVGPU_DROP_IN_MATCH_ONLY
#include "stella_vslam/data/map_database.h"
#include "stella_vslam/data/marker.h"
#include "stella_vslam/optimize/local_bundle_adjuster.h"
#include <yaml-cpp/yaml.h>
#endif
#else
#include "standin/stella_standin.h"
#endif
#include <set>
#include <stdexcept>
#include <utility>

#include "svgpu.h"

namespace stella_vslam {

namespace hip {
//! The calling thread's device context (created on first use on device $SVGPU_DEVICE, default 0; tracking and mapping threads get
//! their own, so their calls overlap on the GPU).  Throws std::runtime_error when no HIP device is present: there is NO CPU
//! fallback inside these classes -- a deployment that wants one keeps the reference's own match::* / g2o objects next to these
//! and switches on the exception ("errors").
svgpu_ctx* context();
//! camera::base -> svgpu_camera (model parameters through the concrete class, img_bounds_ copied)
svgpu_camera to_svgpu_camera(const camera::base* camera);
//! throws std::runtime_error("<where>: <svgpu_last_error>") when status != SVGPU_OK
void check(int status, const char* where);

#ifndef SVGPU_DROP_IN_OPTIMIZE_ONLY
//! Device-resident copy of a frame's / keyframe's observation (include/svgpu.h svgpu_frame_*), kept in a process-wide cache keyed by
//! data::frame::id_ / data::keyframe::id_ and verified against a hash of the WHOLE observation (every descriptor and keypoint): descriptors,
//! undistorted keypoints, stereo x_right and the keypoint grid are uploaded ONCE per frame, every later matcher call binds the resident
//! copy instead of re-uploading them (tracking_module.cc:533-608 runs three to four matchers per frame).  The cache hands out
//! REFERENCE-COUNTED handles: keep the handle for as long as a call that reads the frame is in flight -- eviction, forget_*() or a
//! replacement on another thread then only drop the cache's own reference.  Empty handle when SVGPU_NO_RESIDENT_FRAMES is set.
using frame_handle = std::shared_ptr<svgpu_frame>;
frame_handle resident(const data::frame& frm);
frame_handle resident(const std::shared_ptr<data::keyframe>& keyfrm);
//! system.cc:384-395 on the device, for a frame whose keypoints come from the HIP extractor: undistort_keypoints, convert_keypoints_to_bearings
//! and assign_keypoints_to_grid run on what `extractor_ctx`'s last extract() LEFT ON THE DEVICE (stella_vslam_hip::feature::orb_extractor::
//! context()), the host copies data::frame_observation holds come back in one transfer, and the resident frame is registered under
//! `frame_id`: the descriptors of a tracked frame then cross PCIe once.  (The one line system::create_*_frame gains: INTEGRATION.md.)
void adopt_extraction(unsigned int frame_id, svgpu_ctx* extractor_ctx, const camera::base* camera, unsigned int num_grid_cols, unsigned int num_grid_rows,
                      std::vector<cv::KeyPoint>& undist_keypts, eigen_alloc_vector<Vec3_t>& bearings);
//! an empty resident frame (recycled from a small pool when one is parked there); the handle's deleter parks it again
frame_handle new_frame(svgpu_ctx* ctx);
//! registers a resident frame that was built on the device elsewhere (the tracked-frame chain's fused extraction) under `frame_id`
void register_adopted(unsigned int frame_id, const frame_handle& frame, const std::vector<cv::KeyPoint>& undist_keypts);
//! drops the cached copies (a frame / keyframe that has been destroyed); the cache also evicts least-recently-used entries by itself
void forget_frame(unsigned int frame_id);
void forget_keyframe(unsigned int keyframe_id);
#endif
}  // namespace hip

#ifndef SVGPU_DROP_IN_OPTIMIZE_ONLY  // (the parity fixture of oracle/ref_local compiles only the optimiser class against its stand-in headers)
namespace match {
namespace hip {

static constexpr unsigned int HAMMING_DIST_THR_LOW = 50;    // match/base.h:15-17
static constexpr unsigned int HAMMING_DIST_THR_HIGH = 100;
static constexpr unsigned int MAX_HAMMING_DIST = 256;

class base {  // match/base.h:81-91
public:
    base(const float lowe_ratio, const bool check_orientation) : lowe_ratio_(lowe_ratio), check_orientation_(check_orientation) {}
    virtual ~base() = default;

protected:
    const float lowe_ratio_;
    const bool check_orientation_;
};

class robust final : public base {  // match/robust.h:20-44
public:
    explicit robust(const float lowe_ratio, const bool check_orientation) : base(lowe_ratio, check_orientation) {}
    //! match/robust.cc:148-192 (loop_detector.cc:366-388): brute force on the device, then the reference's own solve::essential_solver
    //! (8-point RANSAC, 50 iterations, no recompute) on the host, then the landmark assignment
    unsigned int match_keyframes(const std::shared_ptr<data::keyframe>& keyfrm1, const std::shared_ptr<data::keyframe>& keyfrm2,
                                 std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_frm, bool validate_with_essential_solver = true,
                                 bool use_fixed_seed = false) const;
    //! match/robust.cc:194-230 (frame_tracker.cc:98-103, relocalizer.cc:27): the same with 1000 iterations and the recompute
    unsigned int match_frame_and_keyframe(data::frame& frm, const std::shared_ptr<data::keyframe>& keyfrm,
                                          std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_frm, bool use_fixed_seed = false) const;
    unsigned int match_for_triangulation(const std::shared_ptr<data::keyframe>& keyfrm_1, const std::shared_ptr<data::keyframe>& keyfrm_2,
                                         const Mat33_t& E_12, std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs,
                                         const float residual_rad_thr) const;
    unsigned int brute_force_match(const data::frame_observation& frm_obs, const std::shared_ptr<data::keyframe>& keyfrm,
                                   std::vector<std::pair<int, int>>& matches) const;
};

class bow_tree final : public base {  // match/bow_tree.h:18-44
public:
    explicit bow_tree(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    unsigned int match_for_triangulation(const std::shared_ptr<data::keyframe>& keyfrm_1, const std::shared_ptr<data::keyframe>& keyfrm_2,
                                         const Mat33_t& E_12, std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs,
                                         const float residual_rad_thr) const;
    unsigned int match_frame_and_keyframe(const std::shared_ptr<data::keyframe>& keyfrm, data::frame& frm,
                                          std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_frm) const;
    unsigned int match_keyframes(const std::shared_ptr<data::keyframe>& keyfrm_1, const std::shared_ptr<data::keyframe>& keyfrm_2,
                                 std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_keyfrm_1) const;
};

class projection final : public base {  // match/projection.h:20-64
public:
    explicit projection(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    unsigned int match_frame_and_landmarks(data::frame& frm, const std::vector<std::shared_ptr<data::landmark>>& local_landmarks,
                                           eigen_alloc_unord_map<unsigned int, Vec2_t>& lm_to_reproj, std::unordered_map<unsigned int, float>& lm_to_x_right,
                                           std::unordered_map<unsigned int, unsigned int>& lm_to_scale, const float margin = 5.0) const;
    unsigned int match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm, const float margin) const;
    unsigned int match_frame_and_keyframe(data::frame& curr_frm, const std::shared_ptr<data::keyframe>& keyfrm,
                                          const std::set<std::shared_ptr<data::landmark>>& already_matched_lms, const float margin,
                                          const unsigned int hamm_dist_thr) const;
    unsigned int match_frame_and_keyframe(const Mat44_t& cam_pose_cw, const camera::base* camera, const data::frame_observation& frm_obs,
                                          const feature::orb_params* orb_params, std::vector<std::shared_ptr<data::landmark>>& frm_landmarks,
                                          const std::shared_ptr<data::keyframe>& keyfrm, const std::set<std::shared_ptr<data::landmark>>& already_matched_lms,
                                          const float margin, const unsigned int hamm_dist_thr) const;
    unsigned int match_by_Sim3_transform(const std::shared_ptr<data::keyframe>& keyfrm, const Mat44_t& Sim3_cw,
                                         const std::vector<std::shared_ptr<data::landmark>>& landmarks,
                                         std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_keyfrm, const float margin) const;
    unsigned int match_keyframes_mutually(const std::shared_ptr<data::keyframe>& keyfrm_1, const std::shared_ptr<data::keyframe>& keyfrm_2,
                                          std::vector<std::shared_ptr<data::landmark>>& matched_lms_in_keyfrm_1, const float& s_12, const Mat33_t& rot_12,
                                          const Vec3_t& trans_12, const float margin) const;
};

class fuse final {  // match/fuse.h:20-40
public:
    explicit fuse(float lowe_ratio) : lowe_ratio_(lowe_ratio) {}
    virtual ~fuse() = default;
    template <typename T>
    unsigned int detect_duplication(const std::shared_ptr<data::keyframe>& keyfrm, const Mat33_t& rot_cw, const Vec3_t& trans_cw, const T& landmarks_to_check,
                                    const float margin, std::unordered_map<std::shared_ptr<data::landmark>, std::shared_ptr<data::landmark>>& duplicated_lms_in_keyfrm,
                                    std::unordered_map<unsigned int, std::shared_ptr<data::landmark>>& new_connections, bool do_reprojection_matching = false) const;

protected:
    const float lowe_ratio_;
};

class area final : public base {  // match/area.h:8-24
public:
    area(const float lowe_ratio, const bool check_orientation) : base(lowe_ratio, check_orientation) {}
    unsigned int match_in_consistent_area(data::frame& frm_1, data::frame& frm_2, std::vector<cv::Point2f>& prev_matched_pts,
                                          std::vector<int>& matched_indices_2_in_frm_1, int margin = 10);
};

}  // namespace hip
}  // namespace match
#endif  // SVGPU_DROP_IN_OPTIMIZE_ONLY

#ifndef SVGPU_DROP_IN_MATCH_ONLY  // (... and only the matcher classes against its matcher stand-ins)
namespace optimize {

#ifndef SVGPU_WITH_STELLA_VSLAM
class local_bundle_adjuster {  // optimize/local_bundle_adjuster.h:15-24
public:
    virtual ~local_bundle_adjuster() = default;
    virtual void optimize(data::map_database* map_db, const std::shared_ptr<data::keyframe>& curr_keyfrm, bool* const force_stop_flag) const = 0;
};
#endif

//! `backend: "hip"` of the LocalBundleAdjuster YAML node: the sibling of local_bundle_adjuster_g2o (optimize/local_bundle_adjuster_g2o.h)
//! and local_bundle_adjuster_gtsam.  Gather (local / fixed keyframes, local landmarks, marker corners) and write-back are the host
//! steps 1 and 7-8 of local_bundle_adjuster_g2o.cc:38-147, 352-430; steps 2-6 are one svgpu_local_ba call.
class local_bundle_adjuster_hip : public local_bundle_adjuster {
public:
    explicit local_bundle_adjuster_hip(const YAML::Node& yaml_node, const unsigned int num_first_iter = 5, const unsigned int num_second_iter = 10);
    virtual ~local_bundle_adjuster_hip() = default;
    void optimize(data::map_database* map_db, const std::shared_ptr<data::keyframe>& curr_keyfrm, bool* const force_stop_flag) const override;
    //! statistics of the last call (diagnostics; the reference logs nothing here)
    mutable svgpu_ba_stats last_stats_{};
    mutable int last_status_ = 0;
    //! host wall-clock of the last call's phases, milliseconds: gather (steps 1), flatten (2-4), solve (svgpu_local_ba: staging, device
    //! loop, read-back), write-back (7-8, under the map mutex), flush of the device-resident landmark table
    mutable double last_phase_ms_[5] = {0, 0, 0, 0, 0};

private:
    const unsigned int num_first_iter_, num_second_iter_;
    const bool use_additional_keyframes_for_monocular_;
};

namespace hip_backend {
//! What local_bundle_adjuster_factory::create (optimize/local_bundle_adjuster_factory.h:17-32) returns for `backend: "hip"`;
//! nullptr for any other backend string so that the reference's own branches stay in charge of theirs.
std::unique_ptr<local_bundle_adjuster> create_local_bundle_adjuster(const YAML::Node& yaml_node);
}  // namespace hip_backend

}  // namespace optimize
#endif  // SVGPU_DROP_IN_MATCH_ONLY
}  // namespace stella_vslam
