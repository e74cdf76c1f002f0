// Stand-in (see ../../README.md): the two sparse BoW containers as the matcher sources use them (std::map semantics of
// fbow::BoWFeatVector: node id -> keypoint indices in ascending index order).
#ifndef SVGPU_SHIM_STELLA_BOW_VOCABULARY_H
#define SVGPU_SHIM_STELLA_BOW_VOCABULARY_H
#include <map>
#include <vector>
namespace stella_vslam {
namespace data {
typedef std::map<unsigned int, float> bow_vector;
typedef std::map<unsigned int, std::vector<unsigned int>> bow_feature_vector;
class bow_vocabulary_stub {};
typedef bow_vocabulary_stub bow_vocabulary;
}  // namespace data
}  // namespace stella_vslam
#endif
