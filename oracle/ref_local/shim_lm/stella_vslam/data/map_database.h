// Stand-in: the one map_database call of data/landmark.cc (prepare_for_erasing), not exercised by the fixtures.
#ifndef SVGPU_SHIMLM_MAP_DATABASE_H
#define SVGPU_SHIMLM_MAP_DATABASE_H
#include <memory>
namespace stella_vslam {
namespace data {
class landmark;
class map_database {
public:
    void erase_landmark(unsigned int) {}
};
}  // namespace data
}  // namespace stella_vslam
#endif
