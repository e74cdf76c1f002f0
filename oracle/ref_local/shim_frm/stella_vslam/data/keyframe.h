// Stand-in: see shim_lm (data/landmark.cc and data/frame.cc are compiled against the same keyframe stand-in)
#include "../../../shim_lm/stella_vslam/data/keyframe.h"
