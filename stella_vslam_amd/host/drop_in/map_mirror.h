// Notifications that keep the device-resident landmark table (include/svgpu.h svgpu_map_*) current: one call per mutator of
// data::landmark, made under the lock the mutator already holds, passing only what it has just written (INTEGRATION.md section 3c lists
// the six lines data/landmark.cc gains).  Dependency-free on purpose: data/landmark.cc includes nothing else of this backend.
// Each call updates a host shadow record and marks the landmark dirty (~50 ns); stella_vslam::hip::flush_map uploads the dirty records
// in one copy -- the tracked-frame chain does that before it reads the table, the HIP bundle adjusters after their write-back.
#pragma once
#include <cstdint>

namespace stella_vslam {
namespace hip {
namespace map_mirror {
//! landmark::landmark (data/landmark.cc:17-34, both constructors)
void landmark_created(unsigned int id, const double* pos_w);
//! landmark::set_pos_in_world (:60-64)
void set_position(unsigned int id, const double* pos_w);
//! landmark::update_mean_normal_and_obs_scale_variance (:256-318), after mean_normal_ / min_valid_dist_ / max_valid_dist_ are stored
void set_geometry(unsigned int id, const double* mean_normal, float min_valid_dist, float max_valid_dist);
//! landmark::compute_descriptor (:199-254), after descriptor_ is stored (nullptr: the descriptor is empty)
void set_descriptor(unsigned int id, const unsigned char* descriptor32);
//! landmark::add_observation / erase_observation (:83-140), with !observations_.empty()
void set_has_observation(unsigned int id, bool has_observation);
//! landmark::prepare_for_erasing (:320-334) -- will_be_erased() from here on
void landmark_erased(unsigned int id);
}  // namespace map_mirror
}  // namespace hip
}  // namespace stella_vslam
