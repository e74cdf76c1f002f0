// Stand-in (see ../README.md): orb_params.cc compiles a constructor from a YAML node that the fixture driver never calls.
#ifndef SVGPU_SHIM_YAML_H
#define SVGPU_SHIM_YAML_H
#include <string>
#include <type_traits>
namespace YAML {
class Node {
public:
    Node operator[](const char*) const { return Node(); }
    template <class T>
    T as() const { return T(); }
    // every key is absent -> the fallback; a fixture can force the value of the boolean keys (-1 = absent)
    static int& forced_bool() {
        static int v = -1;
        return v;
    }
    template <class T>
    T as(const T& fallback) const {
        if constexpr (std::is_same<T, bool>::value) {
            if (forced_bool() >= 0) return forced_bool() != 0;
        }
        return fallback;
    }
    bool operator!() const { return true; }
    explicit operator bool() const { return false; }
};
}  // namespace YAML
#endif
