"""stella_vslam_amd -- MI355X-native ORB front end, Hamming matchers and local BA for stella_vslam.

The product is the C-ABI library (include/svgpu.h, stella_vslam_amd/libsvgpu.so).  This package holds its
HIP sources (csrc/), the C++ adaptor classes with the reference's signatures (host/), and a thin Python
mirror of the same interfaces used by the tests and the benchmark.
"""
__all__ = ["feature", "match", "optimize", "synthetic"]
