// Stand-in (see ../../README.md): robust.cc's two RANSAC-validated wrappers need this class to compile; the fixtures do not run them
// (the essential-matrix solver is host-side control flow outside the hot path).
#ifndef SVGPU_SHIM_STELLA_ESSENTIAL_SOLVER_H
#define SVGPU_SHIM_STELLA_ESSENTIAL_SOLVER_H
#include <vector>
#include "stella_vslam/type.h"
namespace stella_vslam {
namespace solve {
class essential_solver {
public:
    essential_solver(const eigen_alloc_vector<Vec3_t>&, const eigen_alloc_vector<Vec3_t>&, const std::vector<std::pair<int, int>>& matches, bool = false)
        : n_(matches.size()) {}
    void find_via_ransac(const unsigned int, const bool = true, const unsigned int = 5) {}
    bool solution_is_valid() const { return false; }
    std::vector<bool> get_inlier_matches() const { return std::vector<bool>(n_, false); }

private:
    size_t n_;
};
}  // namespace solve
}  // namespace stella_vslam
#endif
