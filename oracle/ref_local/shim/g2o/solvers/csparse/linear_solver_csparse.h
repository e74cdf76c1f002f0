// Stand-in: see ../../core/hyper_graph_action.h
#include "g2o/core/hyper_graph_action.h"
