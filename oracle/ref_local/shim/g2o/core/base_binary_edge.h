// Stand-in: see optimizable_graph.h
#include "g2o/core/optimizable_graph.h"
