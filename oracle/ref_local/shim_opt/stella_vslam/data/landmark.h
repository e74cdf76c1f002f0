// Stand-in for data/landmark.h: what optimize/pose_optimizer_g2o.cc reads of a landmark.
#ifndef SVREF_OPT_DATA_LANDMARK_H
#define SVREF_OPT_DATA_LANDMARK_H
#include "stella_vslam/type.h"

namespace stella_vslam {
namespace data {
class landmark {
public:
    landmark() : will_be_erased_(false) {}
    landmark(const Vec3_t& pos_w, bool erased) : pos_w_(pos_w), will_be_erased_(erased) {}
    Vec3_t get_pos_in_world() const { return pos_w_; }
    bool will_be_erased() const { return will_be_erased_; }

private:
    Vec3_t pos_w_;
    bool will_be_erased_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
