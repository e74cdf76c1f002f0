// Stand-in for the four concrete camera headers in the matcher drop-in fixture (libsvref_mdropin.so): ONE class carrying the public
// parameter members of camera/perspective.h:60-71, fisheye.h:63-72, radial_division.h:66-71 under their own names, so that the product's
// hip::to_svgpu_camera (static_cast by model_type_) reads them from the fixture's camera object; the reprojections stay pure virtual here
// and are the oracle's in svref::camera_fixture (ref_support.h).
#ifndef SVGPU_SHIM_STELLA_CAMERA_ANY_MODEL_H
#define SVGPU_SHIM_STELLA_CAMERA_ANY_MODEL_H
#include "stella_vslam/camera/base.h"
namespace stella_vslam {
namespace camera {
class any_model : public base {
public:
    using base::base;
    double fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
    double k1_ = 0, k2_ = 0, p1_ = 0, p2_ = 0, k3_ = 0, k4_ = 0;
    double distortion_ = 0;
};
using perspective = any_model;
using fisheye = any_model;
using equirectangular = any_model;
using radial_division = any_model;
}  // namespace camera
}  // namespace stella_vslam
#endif
