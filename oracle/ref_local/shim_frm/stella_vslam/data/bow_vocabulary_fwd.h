// Stand-in for data/bow_vocabulary_fwd.h / bow_vocabulary.h: FBoW / DBoW2 are absent; data/frame.h only stores the two sparse containers.
#ifndef SVREF_FRM_BOW_VOCABULARY_FWD_H
#define SVREF_FRM_BOW_VOCABULARY_FWD_H
#include <map>
#include <vector>

#include <opencv2/core/mat.hpp>
namespace stella_vslam {
namespace data {
typedef std::map<unsigned int, float> bow_vector;
typedef std::map<unsigned int, std::vector<unsigned int>> bow_feature_vector;
class bow_vocabulary {};
namespace bow_vocabulary_util {
inline void compute_bow(bow_vocabulary*, const cv::Mat&, bow_vector&, bow_feature_vector&) {}
}  // namespace bow_vocabulary_util
}  // namespace data
}  // namespace stella_vslam
#endif
