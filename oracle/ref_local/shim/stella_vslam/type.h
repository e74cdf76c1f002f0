// Stand-in (see ../README.md) for the reference's type.h: its Eigen typedefs over a minimal fixed-size matrix template with exactly the
// operations the reference's matcher / grid sources use.  Coefficient products are summed left to right ((a0 b0 + a1 b1) + a2 b2), which
// is what Eigen's fixed-size kernels evaluate for these shapes; -ffp-contract=off keeps them unfused as in the reference's default build.
#ifndef SVGPU_SHIM_STELLA_TYPE_H
#define SVGPU_SHIM_STELLA_TYPE_H
#include <cmath>
#include <cstddef>
#include <map>
#include <memory>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <type_traits>

#include <opencv2/core/types.hpp>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace svref_eigen {
template <int R, int C>
struct Matrix {
    typedef double Scalar;
    static int rows() { return R; }
    static int cols() { return C; }
    double v[R * C];  // column-major, as Eigen's default
    Matrix() {
        for (int i = 0; i < R * C; ++i) v[i] = 0.0;
    }
    Matrix(double a, double b, double c) {
        static_assert(R * C == 3, "3-vector");
        v[0] = a, v[1] = b, v[2] = c;
    }
    Matrix(double a, double b) {
        static_assert(R * C == 2, "2-vector");
        v[0] = a, v[1] = b;
    }
    explicit Matrix(const double* p) {  // Eigen: coefficients from a column-major array
        for (int i = 0; i < R * C; ++i) v[i] = p[i];
    }
    double& operator()(int i, int j) { return v[j * R + i]; }
    double operator()(int i, int j) const { return v[j * R + i]; }
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double* data() { return v; }
    const double* data() const { return v; }
    static Matrix Identity() {
        Matrix m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
        return m;
    }
    static Matrix Zero() { return Matrix(); }
    Matrix<C, R> transpose() const {
        Matrix<C, R> t;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) t(j, i) = (*this)(i, j);
        return t;
    }
    template <int BR, int BC>
    Matrix<BR, BC> block(int r0, int c0) const {
        Matrix<BR, BC> b;
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j) b(i, j) = (*this)(r0 + i, c0 + j);
        return b;
    }
    // writable block of a non-const matrix: a copy of the coefficients (usable in any expression) that writes back on assignment
    template <int BR, int BC>
    struct BlockRef : Matrix<BR, BC> {
        Matrix& parent;
        int r0, c0;
        BlockRef(Matrix& p, int r, int c) : Matrix<BR, BC>(static_cast<const Matrix&>(p).template block<BR, BC>(r, c)), parent(p), r0(r), c0(c) {}
        BlockRef& operator=(const Matrix<BR, BC>& o) {
            for (int i = 0; i < BR; ++i)
                for (int j = 0; j < BC; ++j) parent(r0 + i, c0 + j) = o(i, j);
            static_cast<Matrix<BR, BC>&>(*this) = o;
            return *this;
        }
        // Eigen lets a column vector be assigned to a row-vector block (and the reverse): the coefficients in order
        template <int OR, int OC, typename = typename std::enable_if<(OR == BC && OC == BR && (BR == 1 || BC == 1) && BR != BC)>::type>
        BlockRef& operator=(const Matrix<OR, OC>& o) {
            Matrix<BR, BC> t;
            for (int i = 0; i < BR * BC; ++i) t.v[i] = o.v[i];
            return (*this = t);
        }
    };
    template <int BR, int BC>
    BlockRef<BR, BC> block(int r0, int c0) {
        return BlockRef<BR, BC>(*this, r0, c0);
    }
    bool operator==(const Matrix& o) const {
        for (int i = 0; i < R * C; ++i)
            if (v[i] != o.v[i]) return false;
        return true;
    }
    bool operator!=(const Matrix& o) const { return !(*this == o); }
    void fill(double x) {
        for (int i = 0; i < R * C; ++i) v[i] = x;
    }
    Matrix& operator+=(const Matrix& o) {
        for (int i = 0; i < R * C; ++i) v[i] += o.v[i];
        return *this;
    }
    template <int OR, int OC>
    double dot(const Matrix<OR, OC>& o) const {
        static_assert(OR * OC == R * C, "dot of equal-length vectors");
        double s = v[0] * o.v[0];
        for (int i = 1; i < R * C; ++i) s += v[i] * o.v[i];
        return s;
    }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    void normalize() {  // Eigen: *this /= norm()
        const double n = norm();
        for (int i = 0; i < R * C; ++i) v[i] /= n;
    }
    Matrix normalized() const {
        Matrix m(*this);
        const double n = norm();
        if (n > 0.0) m.normalize();
        return m;
    }
    // Eigen's comma initialiser: coefficients in ROW-major order
    struct CommaInit {
        Matrix& m;
        int k;
        CommaInit& operator,(double x) {
            m(k / C, k % C) = x;
            ++k;
            return *this;
        }
    };
    CommaInit operator<<(double x) {
        (*this)(0, 0) = x;
        return CommaInit{*this, 1};
    }
    Matrix operator-() const {
        Matrix m;
        for (int i = 0; i < R * C; ++i) m.v[i] = -v[i];
        return m;
    }
    Matrix operator+(const Matrix& o) const {
        Matrix m;
        for (int i = 0; i < R * C; ++i) m.v[i] = v[i] + o.v[i];
        return m;
    }
    Matrix operator-(const Matrix& o) const {
        Matrix m;
        for (int i = 0; i < R * C; ++i) m.v[i] = v[i] - o.v[i];
        return m;
    }
    Matrix operator*(double s) const {
        Matrix m;
        for (int i = 0; i < R * C; ++i) m.v[i] = v[i] * s;
        return m;
    }
    Matrix operator/(double s) const {
        Matrix m;
        for (int i = 0; i < R * C; ++i) m.v[i] = v[i] / s;
        return m;
    }
    template <int K>
    Matrix<R, K> operator*(const Matrix<C, K>& o) const {
        Matrix<R, K> m;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < K; ++j) {
                double s = (*this)(i, 0) * o(0, j);
                for (int k = 1; k < C; ++k) s += (*this)(i, k) * o(k, j);
                m(i, j) = s;
            }
        return m;
    }
};
template <int R, int C>
static inline Matrix<R, C> operator*(double s, const Matrix<R, C>& a) {
    Matrix<R, C> m;
    for (int i = 0; i < R * C; ++i) m.v[i] = s * a.v[i];
    return m;
}
// Eigen::Quaterniond as far as data/common.cc's JSON helpers need it to compile (not exercised by the fixtures)
struct Quaternion {
    double q[4] = {0, 0, 0, 1};  // x y z w, Eigen's coefficient order
    Quaternion() {}
    explicit Quaternion(const double* xyzw) {
        for (int i = 0; i < 4; ++i) q[i] = xyzw[i];
    }
    explicit Quaternion(const Matrix<3, 3>& R) {
        const double tr = R(0, 0) + R(1, 1) + R(2, 2);
        if (tr > 0) {
            const double s = std::sqrt(tr + 1.0) * 2;
            q[3] = 0.25 * s, q[0] = (R(2, 1) - R(1, 2)) / s, q[1] = (R(0, 2) - R(2, 0)) / s, q[2] = (R(1, 0) - R(0, 1)) / s;
        }
    }
    double x() const { return q[0]; }
    double y() const { return q[1]; }
    double z() const { return q[2]; }
    double w() const { return q[3]; }
    Quaternion normalized() const {
        Quaternion r(*this);
        const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int i = 0; i < 4; ++i) r.q[i] /= n;
        return r;
    }
    Matrix<3, 3> toRotationMatrix() const {
        Matrix<3, 3> R;
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        R(0, 0) = 1 - 2 * (y * y + z * z), R(0, 1) = 2 * (x * y - z * w), R(0, 2) = 2 * (x * z + y * w);
        R(1, 0) = 2 * (x * y + z * w), R(1, 1) = 1 - 2 * (x * x + z * z), R(1, 2) = 2 * (y * z - x * w);
        R(2, 0) = 2 * (x * z - y * w), R(2, 1) = 2 * (y * z + x * w), R(2, 2) = 1 - 2 * (x * x + y * y);
        return R;
    }
};
}  // namespace svref_eigen

namespace Eigen {  // Eigen::Map<const VecN_t>(ptr) as the g2o vertices use it: the coefficients behind the pointer
template <class M>
struct Map;
template <int R, int C>
struct Map<const svref_eigen::Matrix<R, C>> : svref_eigen::Matrix<R, C> {
    explicit Map(const double* p) : svref_eigen::Matrix<R, C>(p) {}
};
}  // namespace Eigen

namespace stella_vslam {
template <typename T, typename... ArgTs>
std::unique_ptr<T> make_unique(ArgTs&&... args) {
    return std::unique_ptr<T>(new T(std::forward<ArgTs>(args)...));
}
typedef float real_t;
template <size_t R, size_t C>
using MatRC_t = svref_eigen::Matrix<(int)R, (int)C>;
using Mat22_t = svref_eigen::Matrix<2, 2>;
using Mat33_t = svref_eigen::Matrix<3, 3>;
using Mat44_t = svref_eigen::Matrix<4, 4>;
using Mat34_t = svref_eigen::Matrix<3, 4>;
template <size_t R>
using VecR_t = svref_eigen::Matrix<(int)R, 1>;
using Vec2_t = svref_eigen::Matrix<2, 1>;
using Vec3_t = svref_eigen::Matrix<3, 1>;
using Vec4_t = svref_eigen::Matrix<4, 1>;
using Vec5_t = svref_eigen::Matrix<5, 1>;
using Vec6_t = svref_eigen::Matrix<6, 1>;
using Vec7_t = svref_eigen::Matrix<7, 1>;
using Quat_t = svref_eigen::Quaternion;
template <typename T>
using eigen_alloc_vector = std::vector<T>;
template <typename T, typename U>
using eigen_alloc_map = std::map<T, U>;
template <typename T>
using eigen_alloc_set = std::set<T>;
template <typename T, typename U>
using eigen_alloc_unord_map = std::unordered_map<T, U>;
template <typename T>
using eigen_alloc_unord_set = std::unordered_set<T>;

// the id comparators of the reference's type.h:118-170, which the matcher sources use through id_ordered_set / nondeterministic::
template <class T>
struct id_less;
template <class T>
struct id_less<std::shared_ptr<T>> {
    bool operator()(const std::shared_ptr<T>& a, const std::shared_ptr<T>& b) const { return a != nullptr && (b == nullptr || a->id_ < b->id_); }
};
template <class T>
struct id_less<std::weak_ptr<T>> {
    bool operator()(const std::weak_ptr<T>& a, const std::weak_ptr<T>& b) const { return !a.expired() && (b.expired() || a.lock()->id_ < b.lock()->id_); }
};
template <class T>
using id_ordered_set = std::set<T, id_less<T>>;
template <class T, class U>
using id_ordered_map = std::map<T, U, id_less<T>>;
namespace nondeterministic {
template <class T>
using unordered_set = std::set<T, id_less<T>>;
template <class T, class U>
using unordered_map = std::map<T, U, id_less<T>>;
}  // namespace nondeterministic
}  // namespace stella_vslam
#endif
