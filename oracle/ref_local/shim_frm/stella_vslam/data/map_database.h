// Stand-in: see shim_lm
#include "../../../shim_lm/stella_vslam/data/map_database.h"
