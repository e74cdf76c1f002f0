// optimize::global_bundle_adjuster_hip -- the reference's optimize::global_bundle_adjuster (optimize/global_bundle_adjuster.h:20-73: a
// concrete class, constructed by module/loop_bundle_adjuster.cc and the initializer) with the same constructor and the same two methods,
// over svgpu_global_ba.  The gather is optimize_impl (global_bundle_adjuster.cc:26-192): every live keyframe a pose (the spanning root
// fixed), every live landmark a point with one edge per observation in a keyframe of the set (a landmark left without an edge is not
// optimised), the four corners of every eligible marker as points (fixed when the marker is kept fixed or markers are fixed for the
// call) with information 1 and no kernel; ONE Levenberg-Marquardt run of num_iter iterations under the gain rule; then the reference's
// post-conditions and result containers (:202-277, :279-412).  Compiles against the reference tree with -DSVGPU_WITH_STELLA_VSLAM, or
// against host/standin/stella_standin.h.
#pragma once
#ifdef SVGPU_WITH_STELLA_VSLAM
#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/keyframe.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/data/marker.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/type.h"
#else
#include "standin/stella_standin.h"
#endif
#include <array>
#include <memory>
#include <unordered_set>
#include <vector>

#include "svgpu.h"

namespace stella_vslam {
namespace hip {  // (hip_backend.h / hip_backend.cc)
svgpu_ctx* context();
svgpu_camera to_svgpu_camera(const camera::base* camera);
void check(int status, const char* where);
}  // namespace hip

namespace optimize {

class global_bundle_adjuster_hip {
public:
    explicit global_bundle_adjuster_hip(unsigned int num_iter = 10, bool use_huber_kernel = true, bool verbose = false);
    virtual ~global_bundle_adjuster_hip() = default;

    void optimize_for_initialization(const std::vector<std::shared_ptr<data::keyframe>>& keyfrms, const std::vector<std::shared_ptr<data::landmark>>& lms,
                                     const std::vector<std::shared_ptr<data::marker>>& markers, float gain_threshold, bool fix_markers,
                                     bool* const force_stop_flag = nullptr) const;

    bool optimize(const std::vector<std::shared_ptr<data::keyframe>>& keyfrms, std::unordered_set<unsigned int>& optimized_keyfrm_ids,
                  std::unordered_set<unsigned int>& optimized_landmark_ids, std::unordered_set<unsigned int>& optimized_marker_ids,
                  eigen_alloc_unord_map<unsigned int, Vec3_t>& lm_to_pos_w_after_global_BA,
                  eigen_alloc_unord_map<unsigned int, Mat44_t>& keyfrm_to_pose_cw_after_global_BA,
                  eigen_alloc_unord_map<unsigned int, std::array<Vec3_t, 4>>& marker_to_pos_w_after_global_BA, bool* const force_stop_flag = nullptr) const;

    //! statistics / status of the last call (diagnostics; the reference logs only with verbose)
    mutable svgpu_ba_stats last_stats_{};
    mutable int last_status_ = 0;

private:
    const unsigned int num_iter_;
    const bool use_huber_kernel_;
    const bool verbose_;
};

}  // namespace optimize
}  // namespace stella_vslam
