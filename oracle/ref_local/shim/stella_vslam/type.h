// Stand-in (see ../README.md) for the reference's Eigen typedef header: feature/orb_extractor.cc includes it and uses nothing of it.
#ifndef SVGPU_SHIM_STELLA_TYPE_H
#define SVGPU_SHIM_STELLA_TYPE_H
#endif
