// Stand-in (see ../../README.md) for the part of g2o's graph classes that the reference's edge / vertex / wrapper headers
// (optimize/internal/se3/*.h, landmark_vertex.h) touch: vertex slots, level, robust kernel, measurement, information, error, the Jacobian
// members and chi2().  No solver, no graph: the fixtures call computeError / linearizeOplus / oplusImpl directly.
#ifndef SVREF_G2O_OPTIMIZABLE_GRAPH_H
#define SVREF_G2O_OPTIMIZABLE_GRAPH_H
#include <cmath>
#include <istream>
#include <ostream>
#include <vector>

#include "stella_vslam/type.h"

namespace g2o {
class RobustKernel {
public:
    virtual ~RobustKernel() = default;
    virtual void setDelta(double d) { _delta = d; }
    double delta() const { return _delta; }
    // rho[0] = rho(e2), rho[1] = rho'(e2), rho[2] = rho''(e2)
    virtual void robustify(double e2, double* rho) const = 0;

protected:
    double _delta = 1.0;
};
// g2o 's RobustKernelHuber::robustify (core/robust_kernel_impl.cpp, pinned release 20230223_git)
class RobustKernelHuber : public RobustKernel {
public:
    void robustify(double e, double* rho) const override {
        const double dsqr = _delta * _delta;
        if (e <= dsqr) {
            rho[0] = e, rho[1] = 1.0, rho[2] = 0.0;
        }
        else {
            const double sqrte = std::sqrt(e);
            rho[0] = 2 * sqrte * _delta - dsqr;
            rho[1] = _delta / sqrte;
            rho[2] = -0.5 * rho[1] / e;
        }
    }
};

struct OptimizableGraph {
    class Vertex {
    public:
        virtual ~Vertex() = default;
        virtual bool read(std::istream&) = 0;
        virtual bool write(std::ostream&) const = 0;
        virtual void setToOriginImpl() = 0;
        virtual void oplusImpl(const double*) = 0;
        void oplus(const double* u) { oplusImpl(u); }
        void setId(int id) { _id = id; }
        int id() const { return _id; }
        void setFixed(bool f) { _fixed = f; }
        bool fixed() const { return _fixed; }
        void setMarginalized(bool m) { _marg = m; }
        bool marginalized() const { return _marg; }

    protected:
        int _id = 0;
        bool _fixed = false, _marg = false;
    };
    class Edge {
    public:
        virtual ~Edge() { delete _robustKernel; }
        virtual bool read(std::istream&) = 0;
        virtual bool write(std::ostream&) const = 0;
        virtual void computeError() = 0;
        virtual void linearizeOplus() = 0;
        virtual double chi2() const = 0;
        void setVertex(size_t i, Vertex* v) { _vertices.at(i) = v; }
        Vertex* vertex(size_t i) const { return _vertices.at(i); }
        int level() const { return _level; }
        void setLevel(int l) { _level = l; }
        void setRobustKernel(RobustKernel* k) {
            delete _robustKernel;
            _robustKernel = k;
        }
        RobustKernel* robustKernel() const { return _robustKernel; }

    protected:
        std::vector<Vertex*> _vertices;
        int _level = 0;
        RobustKernel* _robustKernel = nullptr;
    };
};

template <int D, typename T>
class BaseVertex : public OptimizableGraph::Vertex {
public:
    static const int Dimension = D;
    const T& estimate() const { return _estimate; }
    void setEstimate(const T& e) { _estimate = e; }

protected:
    T _estimate;
};

template <int D, typename E>
class BaseEdge : public OptimizableGraph::Edge {
public:
    typedef svref_eigen::Matrix<D, D> InformationType;
    typedef svref_eigen::Matrix<D, 1> ErrorVector;
    BaseEdge() { _information = InformationType::Identity(); }
    const E& measurement() const { return _measurement; }
    void setMeasurement(const E& m) { _measurement = m; }
    const InformationType& information() const { return _information; }
    InformationType& information() { return _information; }
    void setInformation(const InformationType& i) { _information = i; }
    const ErrorVector& error() const { return _error; }
    double chi2() const override { return _error.dot(_information * _error); }

protected:
    E _measurement;
    InformationType _information;
    ErrorVector _error;
};

template <int D, typename E, typename VertexXi>
class BaseUnaryEdge : public BaseEdge<D, E> {
public:
    BaseUnaryEdge() { this->_vertices.resize(1, nullptr); }
    const svref_eigen::Matrix<D, VertexXi::Dimension>& jacobianOplusXi() const { return _jacobianOplusXi; }

protected:
    svref_eigen::Matrix<D, VertexXi::Dimension> _jacobianOplusXi;
};

template <int D, typename E, typename VertexXi, typename VertexXj>
class BaseBinaryEdge : public BaseEdge<D, E> {
public:
    BaseBinaryEdge() { this->_vertices.resize(2, nullptr); }
    const svref_eigen::Matrix<D, VertexXi::Dimension>& jacobianOplusXi() const { return _jacobianOplusXi; }
    const svref_eigen::Matrix<D, VertexXj::Dimension>& jacobianOplusXj() const { return _jacobianOplusXj; }

protected:
    svref_eigen::Matrix<D, VertexXi::Dimension> _jacobianOplusXi;
    svref_eigen::Matrix<D, VertexXj::Dimension> _jacobianOplusXj;
};
}  // namespace g2o
#endif
