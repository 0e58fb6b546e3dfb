// compile-hygiene stand-in (tests/adapter_stubs/README.md): core/sparse_optimizer.h:191-195
#pragma once
#include "solver.h"
namespace g2o {
class SparseOptimizer : public OptimizableGraph {
 public:
  typedef std::vector<OptimizableGraph::Vertex*> VertexContainer;
  typedef std::vector<OptimizableGraph::Edge*> EdgeContainer;
  const VertexContainer& indexMapping() const;
  const VertexContainer& activeVertices() const;
  const EdgeContainer& activeEdges() const;
};
}  // namespace g2o
