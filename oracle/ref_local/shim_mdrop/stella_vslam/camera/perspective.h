// see any_model.h
#include "stella_vslam/camera/any_model.h"
