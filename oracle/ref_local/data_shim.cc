// Bodies behind the stand-in data / camera headers of shim/ (test infrastructure only): everything that is NOT matcher code forwards to
// the oracle's restatements, so that a difference between this library and the oracle can only come from the reference's matcher code.
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "ref_support.h"

extern "C" {
void orc_assign_keypoints_to_grid(const float* kx, const float* ky, int n, float min_x, float max_x, float min_y, float max_y, int cols, int rows,
                                  int32_t* cell_off, int32_t* cell_items);
int orc_get_keypoints_in_cell(const float* kx, const float* ky, const int32_t* octave, const int32_t* cell_off, const int32_t* cell_items, float min_x,
                              float max_x, float min_y, float max_y, int cols, int rows, float ref_x, float ref_y, float margin, int min_level,
                              int max_level, int32_t* out, int cap);
}

namespace {
struct Grid {
    std::vector<float> kx, ky;
    std::vector<int32_t> octave, cell_off, cell_items;
};
// one grid per observation, built at first use (the fixtures keep an observation alive for one call)
std::unordered_map<const stella_vslam::data::frame_observation*, Grid> g_grids;
}  // namespace

namespace svref {
void forget_grids() { g_grids.clear(); }
}  // namespace svref

namespace stella_vslam {
namespace data {
std::vector<unsigned int> get_keypoints_in_cell(const camera::base* camera, const frame_observation& fo, const float ref_x, const float ref_y,
                                                const float margin, const int min_level, const int max_level) {
    const auto& b = camera->img_bounds_;
    auto it = g_grids.find(&fo);
    if (it == g_grids.end()) {
        Grid g;
        const int n = (int)fo.undist_keypts_.size(), nc = (int)(fo.num_grid_cols_ * fo.num_grid_rows_);
        g.kx.resize(n + 1), g.ky.resize(n + 1), g.octave.resize(n + 1), g.cell_off.resize(nc + 1), g.cell_items.resize(n + 1);
        for (int i = 0; i < n; ++i) {
            g.kx[i] = fo.undist_keypts_[i].pt.x, g.ky[i] = fo.undist_keypts_[i].pt.y;
            g.octave[i] = fo.undist_keypts_[i].octave;
        }
        orc_assign_keypoints_to_grid(g.kx.data(), g.ky.data(), n, b.min_x_, b.max_x_, b.min_y_, b.max_y_, (int)fo.num_grid_cols_, (int)fo.num_grid_rows_,
                                     g.cell_off.data(), g.cell_items.data());
        it = g_grids.emplace(&fo, std::move(g)).first;
    }
    const Grid& g = it->second;
    std::vector<int32_t> out(fo.undist_keypts_.size() + 1);
    const int n = orc_get_keypoints_in_cell(g.kx.data(), g.ky.data(), g.octave.data(), g.cell_off.data(), g.cell_items.data(), b.min_x_, b.max_x_, b.min_y_,
                                            b.max_y_, (int)fo.num_grid_cols_, (int)fo.num_grid_rows_, ref_x, ref_y, margin, min_level, max_level, out.data(),
                                            (int)out.size());
    return std::vector<unsigned int>(out.begin(), out.begin() + n);
}
}  // namespace data
}  // namespace stella_vslam
