// Stand-in: see bow_vocabulary_fwd.h
#include "stella_vslam/data/bow_vocabulary_fwd.h"
