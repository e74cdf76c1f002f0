// Stand-in for data/frame.h: what optimize/internal/se3/shot_vertex_container.h reads of a frame.
#ifndef SVREF_BA_DATA_FRAME_H
#define SVREF_BA_DATA_FRAME_H
#include "stella_vslam/type.h"
namespace stella_vslam {
namespace data {
class frame {
public:
    unsigned int id_ = 0;
    Mat44_t get_pose_cw() const { return pose_cw_; }
    Mat44_t pose_cw_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
