// Stand-in: data/landmark.cc includes data/frame.h and uses nothing of it.
#ifndef SVGPU_SHIMLM_FRAME_H
#define SVGPU_SHIMLM_FRAME_H
namespace stella_vslam {
namespace data {
class frame;
}
}  // namespace stella_vslam
#endif
