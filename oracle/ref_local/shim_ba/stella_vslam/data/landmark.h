// Stand-in for data/landmark.h: the members optimize/local_bundle_adjuster_g2o.cc calls; the write-back calls are recorded for the fixture.
#ifndef SVREF_BA_DATA_LANDMARK_H
#define SVREF_BA_DATA_LANDMARK_H
#include <map>
#include <memory>

#include "stella_vslam/type.h"
namespace stella_vslam {
namespace data {
class keyframe;
class map_database;
class landmark {
public:
    using observations_t = std::map<std::weak_ptr<keyframe>, unsigned int, id_less<std::weak_ptr<keyframe>>>;
    landmark() {}
    landmark(unsigned int id, const Vec3_t& pos_w, bool erased) : id_(id), pos_w_(pos_w), will_be_erased_(erased) {}
    unsigned int id_ = 0;
    Vec3_t get_pos_in_world() const { return pos_w_; }
    void set_pos_in_world(const Vec3_t& p) {
        pos_w_ = p;
        ++num_set_pos_;
    }
    bool will_be_erased() const { return will_be_erased_; }
    observations_t get_observations() const { return observations_; }
    void erase_observation(map_database*, const std::shared_ptr<keyframe>& keyfrm) {
        observations_.erase(keyfrm);
        ++num_erase_observation_;
    }
    void compute_descriptor() { ++num_compute_descriptor_; }
    void update_mean_normal_and_obs_scale_variance() { ++num_update_geometry_; }
    Vec3_t pos_w_;
    bool will_be_erased_ = false;
    observations_t observations_;
    int num_set_pos_ = 0, num_erase_observation_ = 0, num_compute_descriptor_ = 0, num_update_geometry_ = 0;
};
}  // namespace data
}  // namespace stella_vslam
#endif
