// Stand-in for data/frame.h: the members optimize/pose_optimizer_g2o.cc reads (frame.h:60-75, 200-230 of the reference).
#ifndef SVREF_OPT_DATA_FRAME_H
#define SVREF_OPT_DATA_FRAME_H
#include <memory>
#include <vector>

#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/frame_observation.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/type.h"

namespace stella_vslam {
namespace data {
class frame {
public:
    Mat44_t get_pose_cw() const { return pose_cw_; }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    Mat44_t pose_cw_;
    frame_observation frm_obs_;
    const feature::orb_params* orb_params_ = nullptr;
    camera::base* camera_ = nullptr;
    std::vector<std::shared_ptr<landmark>> landmarks_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
