// Stand-in: local_bundle_adjuster_g2o.cc includes g2o/types/sba/types_six_dof_expmap.h and uses nothing of it beyond SE3Quat
#include "g2o/types/slam3d/se3quat.h"
