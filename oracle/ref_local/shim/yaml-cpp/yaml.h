// Stand-in (see ../README.md): orb_params.cc compiles a constructor from a YAML node that the fixture driver never calls.
#ifndef SVGPU_SHIM_YAML_H
#define SVGPU_SHIM_YAML_H
#include <string>
namespace YAML {
class Node {
public:
    Node operator[](const char*) const { return Node(); }
    template <class T>
    T as() const { return T(); }
    template <class T>
    T as(const T& fallback) const { return fallback; }
    bool operator!() const { return true; }
    explicit operator bool() const { return false; }
};
}  // namespace YAML
#endif
