// Stand-in (see ../../README.md) for solve/essential_solver.h:12-80.  The real class needs Eigen's JacobiSVD / EigenSolver / FullPivLU,
// which this container does not have; the RANSAC is host-side control flow outside the hot path, so the fixtures of
// ref_match_exports.cc SCRIPT it: the outcome is a fixed function of the matches it is handed, identical for the reference's
// robust::match_frame_and_keyframe / match_keyframes (match/robust.cc:148-230) and for the product's drop-in wrappers, and every call
// is recorded so that the tests can check what each class asked the solver for (iterations, recompute, fixed seed).
#ifndef SVGPU_SHIM_STELLA_ESSENTIAL_SOLVER_H
#define SVGPU_SHIM_STELLA_ESSENTIAL_SOLVER_H
#include <utility>
#include <vector>
#include "stella_vslam/type.h"
namespace stella_vslam {
namespace solve {
struct essential_solver_script {
    // script
    bool valid = true;          // what solution_is_valid() reports after a RANSAC over enough matches
    int inlier_mod = 3;         // match (i1, i2) is an inlier iff (7 i1 + 3 i2) % inlier_mod != 0; <= 1: every match is an inlier
    // record of the calls
    int calls = 0;
    unsigned last_max_num_iter = 0, last_min_set_size = 0;
    bool last_recompute = false, last_use_fixed_seed = false;
    size_t last_num_matches = 0, last_bearings_1 = 0, last_bearings_2 = 0;
};
inline essential_solver_script& essential_script() {
    static essential_solver_script s;
    return s;
}
class essential_solver {
public:
    essential_solver(const eigen_alloc_vector<Vec3_t>& bearings_1, const eigen_alloc_vector<Vec3_t>& bearings_2,
                     const std::vector<std::pair<int, int>>& matches_12, bool use_fixed_seed = false)
        : matches_12_(matches_12) {
        auto& S = essential_script();
        S.last_use_fixed_seed = use_fixed_seed;
        S.last_num_matches = matches_12.size();
        S.last_bearings_1 = bearings_1.size();
        S.last_bearings_2 = bearings_2.size();
    }
    void find_via_ransac(const unsigned int max_num_iter, const bool recompute = true, const unsigned int min_set_size = 5) {
        auto& S = essential_script();
        ++S.calls;
        S.last_max_num_iter = max_num_iter;
        S.last_recompute = recompute;
        S.last_min_set_size = min_set_size;
        is_inlier_match_.assign(matches_12_.size(), false);
        if (matches_12_.size() < min_set_size) {  // essential_solver.cc:19-23
            solution_is_valid_ = false;
            return;
        }
        for (size_t i = 0; i < matches_12_.size(); ++i)
            is_inlier_match_[i] = S.inlier_mod <= 1 || (7 * matches_12_[i].first + 3 * matches_12_[i].second) % S.inlier_mod != 0;
        solution_is_valid_ = S.valid;
    }
    bool solution_is_valid() const { return solution_is_valid_; }
    std::vector<bool> get_inlier_matches() const { return is_inlier_match_; }

private:
    const std::vector<std::pair<int, int>>& matches_12_;
    bool solution_is_valid_ = false;
    std::vector<bool> is_inlier_match_;
};
}  // namespace solve
}  // namespace stella_vslam
#endif
