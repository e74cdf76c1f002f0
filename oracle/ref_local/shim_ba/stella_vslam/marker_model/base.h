// Stand-in for marker_model/base.h: local_bundle_adjuster_g2o.cc includes it and uses nothing of it.
#ifndef SVREF_BA_MARKER_MODEL_BASE_H
#define SVREF_BA_MARKER_MODEL_BASE_H
namespace stella_vslam {
namespace marker_model {
class base {};
}  // namespace marker_model
}  // namespace stella_vslam
#endif
