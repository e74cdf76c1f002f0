// Stand-in (see ../../README.md) for g2o's HyperGraph / HyperGraphAction / SparseOptimizer / SparseOptimizerTerminateAction as far as the
// reference's optimize/terminate_action.{h,cc} touches them: the action's parameters, the optimizer's active robust chi2 and force-stop
// pointer.  The optimizer here is scripted: the fixture sets the chi2 it reports for an iteration.
#ifndef SVREF_G2O_HYPER_GRAPH_ACTION_H
#define SVREF_G2O_HYPER_GRAPH_ACTION_H
#include <cassert>
#include <functional>
#include <limits>
#include <memory>
#include <vector>

#include "g2o/core/optimizable_graph.h"

namespace g2o {
class HyperGraph {
public:
    virtual ~HyperGraph() = default;
};
class HyperGraphAction {
public:
    class Parameters {
    public:
        virtual ~Parameters() = default;
    };
    class ParametersIteration : public Parameters {
    public:
        explicit ParametersIteration(int iter) : iteration(iter) {}
        int iteration;
    };
    virtual ~HyperGraphAction() = default;
    virtual HyperGraphAction* operator()(const HyperGraph* graph, Parameters* parameters = nullptr) = 0;
};
class OptimizationAlgorithm {
public:
    virtual ~OptimizationAlgorithm() = default;
};
// The optimizer as a container: it owns what is added to it, as g2o's does, and hands optimize() to a hook the fixture installs
// (ref_opt_exports.cc runs the oracle's Levenberg-Marquardt there).  For the terminate_action fixture it is "scripted": it reports the
// chi2 the fixture sets.
class SparseOptimizer : public HyperGraph {
public:
    ~SparseOptimizer() override {
        for (auto* e : _edges) delete e;
        for (auto* v : _vertices) delete v;
        delete _algorithm;
    }
    void computeActiveErrors() { ++num_error_computations; }
    double activeRobustChi2() const { return scripted_chi2; }
    void setForceStopFlag(bool* flag) { _forceStopFlag = flag; }
    bool* forceStopFlag() const { return _forceStopFlag; }
    bool terminate() const { return _forceStopFlag && *_forceStopFlag; }
    void setAlgorithm(OptimizationAlgorithm* a) { _algorithm = a; }
    bool addPostIterationAction(HyperGraphAction* a) {
        post_iteration_actions.push_back(a);
        return true;
    }
    bool addVertex(OptimizableGraph::Vertex* v) {
        _vertices.push_back(v);
        return true;
    }
    bool addEdge(OptimizableGraph::Edge* e) {
        _edges.push_back(e);
        return true;
    }
    bool removeVertex(OptimizableGraph::Vertex* v) {
        for (size_t i = 0; i < _vertices.size(); ++i)
            if (_vertices[i] == v) {
                _vertices.erase(_vertices.begin() + (long)i);
                delete v;
                return true;
            }
        return false;
    }
    void setVerbose(bool) {}
    bool initializeOptimization(int level = 0) {
        active_level = level;
        ++num_initializations;
        return true;
    }
    int optimize(int iterations) { return optimize_hook ? optimize_hook(*this, iterations) : 0; }
    const std::vector<OptimizableGraph::Vertex*>& vertices() const { return _vertices; }
    const std::vector<OptimizableGraph::Edge*>& edges() const { return _edges; }
    std::vector<HyperGraphAction*> post_iteration_actions;
    // Set before the optimizer under test is constructed (it is a local of the reference's function): every new optimizer copies it
    static std::function<int(SparseOptimizer&, int)>& default_hook() {
        static std::function<int(SparseOptimizer&, int)> h;
        return h;
    }
    std::function<int(SparseOptimizer&, int)> optimize_hook = default_hook();
    double scripted_chi2 = 0.0;
    int num_error_computations = 0, num_initializations = 0, active_level = 0;
    // state of the stand-in's Levenberg-Marquardt that outlives one optimize(): the terminate action's _lastChi and its own stop flag
    double lm_last_chi = 0.0;
    unsigned char lm_stop = 0;

protected:
    bool* _forceStopFlag = nullptr;
    OptimizationAlgorithm* _algorithm = nullptr;
    std::vector<OptimizableGraph::Vertex*> _vertices;
    std::vector<OptimizableGraph::Edge*> _edges;
};
// solver stack types of optimize/*_g2o.cc: only constructed and handed over
template <int P, int L>
struct BlockSolverTraits {
    struct PoseMatrixType {};
};
template <typename M>
class LinearSolverEigen {
public:
    virtual ~LinearSolverEigen() = default;
};
template <typename M>
class LinearSolverCSparse : public LinearSolverEigen<M> {};
class BlockSolverBase {
public:
    virtual ~BlockSolverBase() = default;
};
class BlockSolver_6_3 : public BlockSolverBase {
public:
    typedef BlockSolverTraits<6, 3>::PoseMatrixType PoseMatrixType;
    explicit BlockSolver_6_3(std::unique_ptr<LinearSolverEigen<PoseMatrixType>> s) : solver(std::move(s)) {}
    std::unique_ptr<LinearSolverEigen<PoseMatrixType>> solver;
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
public:
    template <class S>
    explicit OptimizationAlgorithmLevenberg(std::unique_ptr<S> s) : solver(std::move(s)) {}
    std::unique_ptr<BlockSolverBase> solver;
};
// g2o/core/sparse_optimizer_terminate_action.{h,cpp} (pinned release 20230223_git): defaults and setOptimizerStopFlag
class SparseOptimizerTerminateAction : public HyperGraphAction {
public:
    SparseOptimizerTerminateAction() : _gainThreshold(1e-6), _lastChi(0.0), _auxTerminateFlag(false), _maxIterations(std::numeric_limits<int>::max()) {}
    double gainThreshold() const { return _gainThreshold; }
    void setGainThreshold(double g) { _gainThreshold = g; }
    int maxIterations() const { return _maxIterations; }
    void setMaxIterations(int m) { _maxIterations = m; }
    double lastChi() const { return _lastChi; }
    bool auxTerminateFlag() const { return _auxTerminateFlag; }

protected:
    void setOptimizerStopFlag(const SparseOptimizer* optimizer, bool stop) {
        if (optimizer->forceStopFlag()) *(optimizer->forceStopFlag()) = stop;
        else {  // g2o installs its own flag when the optimizer has none
            _auxTerminateFlag = stop;
            const_cast<SparseOptimizer*>(optimizer)->setForceStopFlag(&_auxTerminateFlag);
        }
    }
    double _gainThreshold;
    double _lastChi;
    bool _auxTerminateFlag;
    int _maxIterations;
};
}  // namespace g2o
#endif
