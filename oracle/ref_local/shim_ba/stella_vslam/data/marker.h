// Stand-in for data/marker.h and data/marker2d.h: members local_bundle_adjuster_g2o.cc / marker_vertex_container.h read (the fixtures hold no markers).
#ifndef SVREF_BA_DATA_MARKER_H
#define SVREF_BA_DATA_MARKER_H
#include <map>
#include <memory>
#include <vector>

#include <opencv2/core/types.hpp>

#include "stella_vslam/type.h"
namespace stella_vslam {
namespace data {
class keyframe;
class marker2d {
public:
    std::vector<cv::Point2f> undist_corners_;
};
class marker {
public:
    unsigned int id_ = 0;
    bool keep_fixed_ = false, initialized_before_ = false;
    std::map<unsigned int, std::shared_ptr<keyframe>> observations_;
    eigen_alloc_vector<Vec3_t> corners_pos_w_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
