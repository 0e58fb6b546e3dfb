// compile-hygiene stand-in (tests/adapter_stubs/README.md): the vertex / edge classes of types/types_six_dof_expmap.h and their bases
#pragma once
#include "../core/solver.h"
namespace g2o {
typedef Eigen::Matrix<double, 7, 1> Vector7d;
using Eigen::Vector2d;
using Eigen::Vector3d;
using Eigen::Vector4d;
using Eigen::Matrix3d;
class SE3Quat { public: Vector7d toVector() const; };
template <int D, class T>
class BaseVertex : public OptimizableGraph::Vertex {
 public:
  const T& estimate() const;
  virtual void mapHessianMemory(double* d);
  virtual const double& b(int i) const;
  virtual double& b(int i);
  virtual void clearQuadraticForm();
};
template <int D, class E, class VertexXi, class VertexXj>
class BaseBinaryEdge : public OptimizableGraph::Edge {
 public:
  const E& measurement() const;
  const Eigen::Matrix<double, D, D>& information() const;
  virtual void computeError();
  virtual double chi2() const;
  virtual void constructQuadraticForm();
  virtual void mapHessianMemory(double* d, int i, int j, bool rowMajor);
  virtual void linearizeOplus(JacobianWorkspace& jacobianWorkspace);
};
class VertexSE3Expmap : public BaseVertex<6, SE3Quat> {};
class VertexSBAPointXYZ : public BaseVertex<3, Vector3d> {};
class EdgeSE3ProjectXYZ : public BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap> { public: double fx, fy, cx, cy; };
class EdgeSE3Expmap : public BaseBinaryEdge<6, SE3Quat, VertexSE3Expmap, VertexSE3Expmap> {};
}  // namespace g2o
