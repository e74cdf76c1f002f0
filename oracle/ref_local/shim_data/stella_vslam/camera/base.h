// Stand-in (see ../../README.md) for camera/base.h: the members and the two virtual reprojections the matcher / grid sources use.
// The concrete camera of the fixtures (camera::svref_camera) forwards the reprojections to the oracle's camera functions.
#ifndef SVGPU_SHIM_STELLA_CAMERA_BASE_H
#define SVGPU_SHIM_STELLA_CAMERA_BASE_H
#include "stella_vslam/type.h"
namespace stella_vslam {
namespace camera {
enum class setup_type_t { Monocular = 0, Stereo = 1, RGBD = 2 };
enum class model_type_t { Perspective = 0, Fisheye = 1, Equirectangular = 2, RadialDivision = 3 };
struct image_bounds {
    float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;
};
class base {
public:
    base(setup_type_t setup, model_type_t model, unsigned int cols, unsigned int rows, double focal_x_baseline, double true_baseline)
        : setup_type_(setup), model_type_(model), cols_(cols), rows_(rows), focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline) {}
    virtual ~base() = default;
    const setup_type_t setup_type_;
    const model_type_t model_type_;
    const unsigned int cols_, rows_;
    const double focal_x_baseline_, true_baseline_;
    image_bounds img_bounds_;
    // camera/base.h:123-129 (set by the concrete cameras from img_bounds_ and the grid size of the observation)
    float inv_cell_width_ = 0.f, inv_cell_height_ = 0.f;
    virtual bool reproject_to_image(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec2_t& reproj, float& x_right) const = 0;
    virtual bool reproject_to_bearing(const Mat33_t& rot_cw, const Vec3_t& trans_cw, const Vec3_t& pos_w, Vec3_t& reproj) const = 0;
};
}  // namespace camera
}  // namespace stella_vslam
#endif
