// Stand-in for util/converter.h as far as optimize/pose_optimizer_g2o.cc uses it: the two SE3 conversions (util/converter.cc:17-25),
// restated because the real header pulls in Eigen/Dense and g2o's sim3 types.
#ifndef SVREF_UTIL_CONVERTER_H
#define SVREF_UTIL_CONVERTER_H
#include <g2o/types/slam3d/se3quat.h>

#include "stella_vslam/type.h"

namespace stella_vslam {
namespace util {
struct converter {
    static g2o::SE3Quat to_g2o_SE3(const Mat44_t& pose) {
        const Mat33_t rot = pose.block<3, 3>(0, 0);
        const Vec3_t trans = pose.block<3, 1>(0, 3);
        return g2o::SE3Quat{rot, trans};
    }
    static Mat44_t to_eigen_mat(const g2o::SE3Quat& g2o_SE3) { return g2o_SE3.to_homogeneous_matrix(); }
};
}  // namespace util
}  // namespace stella_vslam
#endif
