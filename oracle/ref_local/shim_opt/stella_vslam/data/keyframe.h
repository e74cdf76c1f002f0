// Stand-in for data/keyframe.h: the members optimize/pose_optimizer_g2o.cc reads -- the same five as of a frame.
#ifndef SVREF_OPT_DATA_KEYFRAME_H
#define SVREF_OPT_DATA_KEYFRAME_H
#include "stella_vslam/data/frame.h"

namespace stella_vslam {
namespace data {
class keyframe : public frame {};
}  // namespace data
}  // namespace stella_vslam
#endif
