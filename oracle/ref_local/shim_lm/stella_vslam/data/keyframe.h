// Stand-in (see ../../../shim/README.md) for data/keyframe.h as far as the reference's data/landmark.cc touches a keyframe (third
// library, oracle/_ref/libsvref_lm.so: it uses the reference's REAL data/landmark.h).
#ifndef SVGPU_SHIMLM_KEYFRAME_H
#define SVGPU_SHIMLM_KEYFRAME_H
#include <memory>
#include "stella_vslam/data/frame_observation.h"
#include "stella_vslam/feature/orb_params.h"
namespace stella_vslam {
namespace data {
class landmark;
class keyframe : public std::enable_shared_from_this<keyframe> {
public:
    keyframe(unsigned int id, const feature::orb_params* orb_params) : id_(id), orb_params_(orb_params) {}
    Vec3_t get_trans_wc() const { return trans_wc_; }
    bool will_be_erased() { return will_be_erased_; }
    void add_landmark(std::shared_ptr<landmark>, const unsigned int) {}
    void erase_landmark_with_index(const unsigned int) {}
    unsigned int id_;
    const feature::orb_params* orb_params_;
    frame_observation frm_obs_;
    Vec3_t trans_wc_;
    bool will_be_erased_ = false;
};
}  // namespace data
}  // namespace stella_vslam
#endif
