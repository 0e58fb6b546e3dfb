// compile-hygiene stand-in (tests/adapter_stubs/README.md): g2o::Solver's interface as core/solver.h:43-137 declares it, and the
// graph classes it names
#pragma once
#include <cstddef>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include <Eigen/Core>

namespace g2o {
using Eigen::MatrixXd;
class RobustKernel;
class HyperGraph {      // core/hyper_graph.h:93-140
 public:
  class Vertex { public: virtual ~Vertex(); int id() const; };
  typedef std::vector<Vertex*> VertexContainer;
  class Edge {
   public:
    virtual ~Edge();
    const VertexContainer& vertices() const;
    Vertex* vertex(size_t i);
    const Vertex* vertex(size_t i) const;
  };
  typedef std::set<Edge*> EdgeSet;
};
class JacobianWorkspace {   // core/jacobian_workspace.h:49-96
 public:
  bool allocate();
  void updateSize(const HyperGraph::Edge* e);
  double* workspaceForVertex(int vertexIndex);
};
class OptimizableGraph : public HyperGraph {   // core/optimizable_graph.h:127-525, 659
 public:
  class Vertex : public HyperGraph::Vertex {
   public:
    bool fixed() const;
    bool marginalized() const;
    int dimension() const;
    int hessianIndex() const;
    int colInHessian() const;
    virtual void mapHessianMemory(double* d) = 0;
    virtual const double& b(int i) const = 0;
    virtual double& b(int i) = 0;
    virtual void clearQuadraticForm() = 0;
  };
  class Edge : public HyperGraph::Edge {
   public:
    RobustKernel* robustKernel() const;
    int dimension() const;
    virtual void computeError() = 0;
    virtual double chi2() const = 0;
    virtual void constructQuadraticForm() = 0;
    virtual void mapHessianMemory(double* d, int i, int j, bool rowMajor) = 0;
    virtual void linearizeOplus(JacobianWorkspace& jacobianWorkspace) = 0;
  };
  JacobianWorkspace& jacobianWorkspace();
};
template <class M> class SparseBlockMatrix;      // (core/solver.h includes sparse_block_matrix.h; the stand-in for it is a file of its own)
class SparseOptimizer;
class Solver {
 public:
  Solver();
  virtual ~Solver();
  virtual bool init(SparseOptimizer* optimizer, bool online = false) = 0;
  virtual bool buildStructure(bool zeroBlocks = false) = 0;
  virtual bool updateStructure(const std::vector<HyperGraph::Vertex*>& vset, const HyperGraph::EdgeSet& edges) = 0;
  virtual bool buildSystem() = 0;
  virtual bool solve() = 0;
  virtual bool computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) = 0;
  virtual bool setLambda(double lambda, bool backup = false) = 0;
  virtual void restoreDiagonal() = 0;
  double* x();
  double* b();
  size_t vectorSize() const;
  SparseOptimizer* optimizer() const;
  virtual bool supportsSchur();
  virtual bool schur() = 0;
  virtual void setSchur(bool s) = 0;
  virtual void setWriteDebug(bool) = 0;
  virtual bool writeDebug() const = 0;
  virtual bool saveHessian(const std::string& fileName) const = 0;

 protected:
  SparseOptimizer* _optimizer;
  double* _x;
  double* _b;
  size_t _xSize, _maxXSize;
  void resizeVector(size_t sx);
};
}  // namespace g2o
