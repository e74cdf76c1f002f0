// Stand-in for util/converter.h: data/frame.h includes it and data/frame.cc uses nothing of it
#include "stella_vslam/type.h"
