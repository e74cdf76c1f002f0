// Stand-in (see ../../README.md) for g2o's HyperGraph / HyperGraphAction / SparseOptimizer / SparseOptimizerTerminateAction as far as the
// reference's optimize/terminate_action.{h,cc} touches them: the action's parameters, the optimizer's active robust chi2 and force-stop
// pointer.  The optimizer here is scripted: the fixture sets the chi2 it reports for an iteration.
#ifndef SVREF_G2O_HYPER_GRAPH_ACTION_H
#define SVREF_G2O_HYPER_GRAPH_ACTION_H
#include <cassert>
#include <limits>

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
class SparseOptimizer : public HyperGraph {
public:
    void computeActiveErrors() { ++num_error_computations; }
    double activeRobustChi2() const { return scripted_chi2; }
    void setForceStopFlag(bool* flag) { _forceStopFlag = flag; }
    bool* forceStopFlag() const { return _forceStopFlag; }
    double scripted_chi2 = 0.0;
    int num_error_computations = 0;

protected:
    bool* _forceStopFlag = nullptr;
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
