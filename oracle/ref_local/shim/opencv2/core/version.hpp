// Stand-in (see ../../README.md): the OpenCV version the reference pins (Dockerfile.desktop:110).
#ifndef SVGPU_SHIM_OPENCV_VERSION_HPP
#define SVGPU_SHIM_OPENCV_VERSION_HPP
#define CV_VERSION_MAJOR 4
#define CV_VERSION_MINOR 7
#define CV_VERSION_REVISION 0
#define CV_MAJOR_VERSION CV_VERSION_MAJOR
#define CV_MINOR_VERSION CV_VERSION_MINOR
#endif
