// Stand-in for data/map_database.h: the two members optimize/local_bundle_adjuster_g2o.cc touches.
#ifndef SVREF_BA_DATA_MAP_DATABASE_H
#define SVREF_BA_DATA_MAP_DATABASE_H
#include <mutex>
namespace stella_vslam {
namespace data {
class map_database {
public:
    unsigned int get_fixed_keyframe_id_threshold() const { return fixed_keyframe_id_threshold_; }
    unsigned int fixed_keyframe_id_threshold_ = 0;
    static std::mutex mtx_database_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
