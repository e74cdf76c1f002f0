// Stand-in (see ../../README.md) for data/common.h: the grid lookup the matcher sources call.  Its body (oracle/ref_local/data_shim.cc)
// forwards to the oracle's restatement of data/common.cc:127-190, which the reference's own test vectors pin
// (test/stella_vslam/data/common_get_cell_indices.cc -> tests/test_oracle_frame.py).
#ifndef SVGPU_SHIM_STELLA_DATA_COMMON_H
#define SVGPU_SHIM_STELLA_DATA_COMMON_H
#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/frame_observation.h"
namespace stella_vslam {
namespace data {
std::vector<unsigned int> get_keypoints_in_cell(const camera::base* camera, const frame_observation& frm_obs, const float ref_x, const float ref_y,
                                                const float margin, const int min_level = -1, const int max_level = -1);
}  // namespace data
}  // namespace stella_vslam
#endif
