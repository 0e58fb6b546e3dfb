// compile-hygiene stand-in (tests/adapter_stubs/README.md): include/object_slam/g2o_Object.h:16,23,145,202,235,264,291
#pragma once
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
typedef Eigen::Matrix<double, 10, 1> Vector10d;
namespace g2o {
class cuboid { public: Vector10d toVector() const; };
class VertexCuboid : public BaseVertex<9, cuboid> {};
class EdgeSE3Cuboid : public BaseBinaryEdge<9, cuboid, VertexSE3Expmap, VertexCuboid> {};
class EdgeSE3CuboidProj : public BaseBinaryEdge<4, Vector4d, VertexSE3Expmap, VertexCuboid> { public: Matrix3d Kalib; };
}  // namespace g2o
