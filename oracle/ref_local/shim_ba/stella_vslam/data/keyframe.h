// Stand-in for data/keyframe.h and data/graph_node.h: the members optimize/local_bundle_adjuster_g2o.cc and the vertex containers read or
// call; erase_landmark / set_pose_cw are recorded for the fixture.
#ifndef SVREF_BA_DATA_KEYFRAME_H
#define SVREF_BA_DATA_KEYFRAME_H
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/frame_observation.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/data/marker.h"
#include "stella_vslam/feature/orb_params.h"
#include "stella_vslam/type.h"
#include "stella_vslam/util/converter.h"  // the reference's keyframe.h brings it in for the vertex containers
namespace stella_vslam {
namespace data {
class keyframe;
class graph_node {
public:
    std::vector<std::shared_ptr<keyframe>> get_covisibilities() const { return covisibilities_; }
    bool is_spanning_root() const { return is_spanning_root_; }
    std::vector<std::shared_ptr<keyframe>> covisibilities_;
    bool is_spanning_root_ = false;
};
class keyframe {
public:
    unsigned int id_ = 0;
    std::unique_ptr<graph_node> graph_node_{new graph_node()};
    camera::base* camera_ = nullptr;
    const feature::orb_params* orb_params_ = nullptr;
    frame_observation frm_obs_;
    std::unordered_map<unsigned int, marker2d> markers_2d_;
    bool will_be_erased() const { return will_be_erased_; }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    std::vector<std::shared_ptr<marker>> get_markers() const { return {}; }
    Mat44_t get_pose_cw() const { return pose_cw_; }
    void set_pose_cw(const Mat44_t& T) {
        pose_cw_ = T;
        ++num_set_pose_;
    }
    void erase_landmark(const std::shared_ptr<landmark>& lm) {
        for (auto& l : landmarks_)
            if (l == lm) l = nullptr;
        erased_landmarks_.push_back(lm->id_);
    }
    Mat44_t pose_cw_;
    bool will_be_erased_ = false;
    std::vector<std::shared_ptr<landmark>> landmarks_;  // per keypoint index
    std::vector<unsigned int> erased_landmarks_;
    int num_set_pose_ = 0;
};
}  // namespace data
}  // namespace stella_vslam
#endif
