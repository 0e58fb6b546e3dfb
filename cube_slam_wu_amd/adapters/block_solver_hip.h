// block_solver_hip.h -- header-only g2o::Solver that runs buildSystem / solve on an MI355X through
// libcubeslam_hip.so.  It plugs into the seam the reference constructs at object_slam/src/main_obj.cpp:510-519:
//
//     g2o::BlockSolverX* solver_ptr = new g2o::BlockSolverX(linearSolver);          // reference
//     g2o::Solver*       solver_ptr = new cubeslam::BlockSolverHIP();               // this adapter
//     g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
//
// OptimizationAlgorithmLevenberg keeps driving the iteration exactly as before
// (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-163): it calls buildStructure(), buildSystem(),
// setLambda(), solve(), restoreDiagonal(), reads x()/b()/vectorSize() (core/solver.h:95-103), and calls the
// optimizer's computeActiveErrors()/activeRobustChi2()/update()/push()/pop() -- which stay on the CPU unless the
// caller swaps the whole loop for cs_ba_optimize().  It also reads v->hessian(j, j) of every vertex for
// lambda_0 (:166-180): buildStructure() maps memory behind the vertices' diagonal blocks and buildSystem() fills it.
// tests/test_ba_gpu.py::test_stepwise_abi_driven_like_g2o_levenberg drives the same call sequence from a host-side LM loop.  Vertex/edge types handled on the device:
// VertexSE3Expmap, VertexSBAPointXYZ, VertexCuboid, EdgeSE3ProjectXYZ, EdgeSE3Cuboid, EdgeSE3CuboidProj, EdgeSE3Expmap; any other
// active edge makes init() fail loudly (no silent CPU fallback).
//
// Needs g2o + Eigen headers; not compiled in the build container (neither is installed there).
#pragma once

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "Thirdparty/g2o/g2o/core/solver.h"
#include "Thirdparty/g2o/g2o/core/sparse_optimizer.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
#include "object_slam/g2o_Object.h"

#include "cubeslam_hip.h"

namespace cubeslam {

class BlockSolverHIP : public g2o::Solver {
 public:
  explicit BlockSolverHIP(int device = 0) { if (cs_ba_create(device, &ba_) != CS_OK) throw std::runtime_error(cs_last_error()); }
  virtual ~BlockSolverHIP() { cs_ba_destroy(ba_); }

  virtual bool init(g2o::SparseOptimizer* optimizer, bool /*online*/ = false) { _optimizer = optimizer; return true; }

  // core/block_solver.hpp:142-295: index mapping -> flat arrays.  g2o orders non-marginalised vertices by id; the
  // library orders (cameras, cuboids) or (cuboids, cameras), so the caller must give one class the lower ids.
  virtual bool buildStructure(bool /*zeroBlocks*/ = false) {
    cams_.clear(); cubs_.clear(); pts_.clear();
    std::vector<double> cam7, cub10, pt3;
    std::vector<int> cam_fixed, cub_fixed, pt_fixed;
    int min_cam_id = 1 << 30, min_cub_id = 1 << 30;
    for (auto* v : _optimizer->activeVertices()) {
      if (auto* c = dynamic_cast<g2o::VertexSE3Expmap*>(v)) {
        index_[v] = (int)cams_.size(); cams_.push_back(c); cam_fixed.push_back(c->fixed());
        g2o::Vector7d e = c->estimate().toVector(); cam7.insert(cam7.end(), e.data(), e.data() + 7);
        min_cam_id = std::min(min_cam_id, v->id());
      } else if (auto* o = dynamic_cast<g2o::VertexCuboid*>(v)) {
        index_[v] = (int)cubs_.size(); cubs_.push_back(o); cub_fixed.push_back(o->fixed());
        Vector10d e = o->estimate().toVector(); cub10.insert(cub10.end(), e.data(), e.data() + 10);
        min_cub_id = std::min(min_cub_id, v->id());
      } else if (auto* p = dynamic_cast<g2o::VertexSBAPointXYZ*>(v)) {
        index_[v] = (int)pts_.size(); pts_.push_back(p); pt_fixed.push_back(p->fixed());
        pt3.insert(pt3.end(), p->estimate().data(), p->estimate().data() + 3);
      } else {
        throw std::runtime_error("BlockSolverHIP: unsupported vertex type");
      }
    }
    if (cs_ba_set_vertices(ba_, cam7.data(), cam_fixed.data(), (int)cams_.size(), cub10.data(), cub_fixed.data(), (int)cubs_.size(),
                           pt3.data(), pt_fixed.data(), (int)pts_.size(), min_cub_id < min_cam_id) != CS_OK) return false;
    std::vector<int> e_pt, e_cam, ce_cam, ce_cub, pe_cam, pe_cub, oe_i, oe_j;
    std::vector<double> uv, info4, intr4, huber, meas10, info81, meas7, info36, meas4, info16, K9;
    for (auto* e : _optimizer->activeEdges()) {
      if (auto* pe = dynamic_cast<g2o::EdgeSE3ProjectXYZ*>(e)) {
        e_pt.push_back(index_[pe->vertex(0)]); e_cam.push_back(index_[pe->vertex(1)]);
        uv.push_back(pe->measurement()[0]); uv.push_back(pe->measurement()[1]);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) info4.push_back(pe->information()(i, j));
        intr4.push_back(pe->fx); intr4.push_back(pe->fy); intr4.push_back(pe->cx); intr4.push_back(pe->cy);
        auto* hk = dynamic_cast<g2o::RobustKernelHuber*>(pe->robustKernel());
        huber.push_back(hk ? hk->delta() : 0.0);
      } else if (auto* ce = dynamic_cast<g2o::EdgeSE3Cuboid*>(e)) {
        ce_cam.push_back(index_[ce->vertex(0)]); ce_cub.push_back(index_[ce->vertex(1)]);
        Vector10d m = ce->measurement().toVector(); meas10.insert(meas10.end(), m.data(), m.data() + 10);
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) info81.push_back(ce->information()(i, j));
      } else if (auto* qe = dynamic_cast<g2o::EdgeSE3CuboidProj*>(e)) {
        pe_cam.push_back(index_[qe->vertex(0)]); pe_cub.push_back(index_[qe->vertex(1)]);
        for (int i = 0; i < 4; i++) meas4.push_back(qe->measurement()[i]);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) info16.push_back(qe->information()(i, j));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K9.push_back(qe->Kalib(i, j));
      } else if (auto* oe = dynamic_cast<g2o::EdgeSE3Expmap*>(e)) {
        oe_i.push_back(index_[oe->vertex(0)]); oe_j.push_back(index_[oe->vertex(1)]);
        g2o::Vector7d m = oe->measurement().toVector(); meas7.insert(meas7.end(), m.data(), m.data() + 7);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) info36.push_back(oe->information()(i, j));
      } else {
        throw std::runtime_error("BlockSolverHIP: unsupported edge type (no CPU fallback by design)");
      }
    }
    cs_ba_set_edges_proj(ba_, (int)e_pt.size(), e_pt.data(), e_cam.data(), uv.data(), info4.data(), intr4.data(), huber.data());
    cs_ba_set_edges_cuboid(ba_, (int)ce_cam.size(), ce_cam.data(), ce_cub.data(), meas10.data(), info81.data());
    cs_ba_set_edges_cuboid_proj(ba_, (int)pe_cam.size(), pe_cam.data(), pe_cub.data(), meas4.data(), info16.data(), K9.data());
    cs_ba_set_edges_odom(ba_, (int)oe_i.size(), oe_i.data(), oe_j.data(), meas7.data(), info36.data());
    int sp = 0, sl = 0;
    if (cs_ba_sizes(ba_, &sp, &sl) != CS_OK) return false;
    resizeVector(sp + sl);
    // BlockSolver::buildStructure maps real memory behind every active vertex's A_ii (block_solver.hpp:185,191): a vertex is
    // born with _hessian(0, D, D) (core/base_vertex.hpp:30) and OptimizationAlgorithmLevenberg::computeLambdaInit()
    // dereferences v->hessian(j, j) of every vertex of indexMapping() on iteration 0
    // (optimization_algorithm_levenberg.cpp:166-180).  The blocks live in diag_ and are refreshed by buildSystem().
    size_t tot = 0;
    for (auto* v : _optimizer->indexMapping()) tot += (size_t)v->dimension() * v->dimension();
    diag_.assign(tot, 0.0);
    size_t off = 0;
    for (auto* v : _optimizer->indexMapping()) { v->mapHessianMemory(diag_.data() + off); off += (size_t)v->dimension() * v->dimension(); }
    // x()/b() follow indexMapping() (SparseOptimizer::update walks it, sparse_optimizer.cpp:422-435); the library lays its
    // vectors out as [cameras | cuboids] or [cuboids | cameras], then the points, each class in caller order.  That is g2o's
    // order exactly when ids are class-contiguous, which is how the reference numbers its vertices (main_obj.cpp:541,599).
    {
      int prev_class = -1, prev_idx = -1, seen = 0;
      for (auto* v : _optimizer->indexMapping()) {
        const int cls = dynamic_cast<g2o::VertexSBAPointXYZ*>(v) ? 2 : (dynamic_cast<g2o::VertexCuboid*>(v) ? (min_cub_id < min_cam_id ? 0 : 1) : (min_cub_id < min_cam_id ? 1 : 0));
        const int idx = index_[v];
        if (cls < prev_class || (cls == prev_class && idx < prev_idx)) throw std::runtime_error("BlockSolverHIP: vertex ids must be contiguous per class (cameras, cuboids) and points marginalised");
        if (cls != prev_class) seen++;
        prev_class = cls; prev_idx = idx;
      }
      (void)seen;
    }
    return true;
  }
  // core/block_solver.hpp:297-350 (online processing: the graph grew by `vset` / `edges`).  New cameras, points and edges are appended
  // behind the existing ones (cs_ba_append_*: the library keeps what it has and re-runs its structure phase on the next solve); a
  // new cuboid would break the class-contiguous vertex order x() / b() rely on, so that case packs the graph again from scratch.
  virtual bool updateStructure(const std::vector<g2o::HyperGraph::Vertex*>& vset, const g2o::HyperGraph::EdgeSet& edges) {
    for (auto* v : vset) if (dynamic_cast<g2o::VertexCuboid*>(v)) return buildStructure();
    std::vector<double> cam7, pt3;
    std::vector<int> cam_fixed, pt_fixed;
    for (auto* hv : vset) {
      if (auto* c = dynamic_cast<g2o::VertexSE3Expmap*>(hv)) {
        index_[hv] = (int)cams_.size(); cams_.push_back(c); cam_fixed.push_back(c->fixed());
        g2o::Vector7d e = c->estimate().toVector(); cam7.insert(cam7.end(), e.data(), e.data() + 7);
      } else if (auto* p = dynamic_cast<g2o::VertexSBAPointXYZ*>(hv)) {
        index_[hv] = (int)pts_.size(); pts_.push_back(p); pt_fixed.push_back(p->fixed());
        pt3.insert(pt3.end(), p->estimate().data(), p->estimate().data() + 3);
      } else {
        throw std::runtime_error("BlockSolverHIP: unsupported vertex type");
      }
    }
    if (cs_ba_append_vertices(ba_, cam7.data(), cam_fixed.data(), (int)cam_fixed.size(), nullptr, nullptr, 0, pt3.data(), pt_fixed.data(), (int)pt_fixed.size()) != CS_OK) return false;
    std::vector<int> e_pt, e_cam, ce_cam, ce_cub, pe_cam, pe_cub, oe_i, oe_j;
    std::vector<double> uv, info4, intr4, huber, meas10, info81, meas7, info36, meas4, info16, K9;
    for (auto* he : edges) {
      if (auto* pe = dynamic_cast<g2o::EdgeSE3ProjectXYZ*>(he)) {
        e_pt.push_back(index_[pe->vertex(0)]); e_cam.push_back(index_[pe->vertex(1)]);
        uv.push_back(pe->measurement()[0]); uv.push_back(pe->measurement()[1]);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) info4.push_back(pe->information()(i, j));
        intr4.push_back(pe->fx); intr4.push_back(pe->fy); intr4.push_back(pe->cx); intr4.push_back(pe->cy);
        auto* hk = dynamic_cast<g2o::RobustKernelHuber*>(pe->robustKernel());
        huber.push_back(hk ? hk->delta() : 0.0);
      } else if (auto* ce = dynamic_cast<g2o::EdgeSE3Cuboid*>(he)) {
        ce_cam.push_back(index_[ce->vertex(0)]); ce_cub.push_back(index_[ce->vertex(1)]);
        Vector10d m = ce->measurement().toVector(); meas10.insert(meas10.end(), m.data(), m.data() + 10);
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) info81.push_back(ce->information()(i, j));
      } else if (auto* qe = dynamic_cast<g2o::EdgeSE3CuboidProj*>(he)) {
        pe_cam.push_back(index_[qe->vertex(0)]); pe_cub.push_back(index_[qe->vertex(1)]);
        for (int i = 0; i < 4; i++) meas4.push_back(qe->measurement()[i]);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) info16.push_back(qe->information()(i, j));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K9.push_back(qe->Kalib(i, j));
      } else if (auto* oe = dynamic_cast<g2o::EdgeSE3Expmap*>(he)) {
        oe_i.push_back(index_[oe->vertex(0)]); oe_j.push_back(index_[oe->vertex(1)]);
        g2o::Vector7d m = oe->measurement().toVector(); meas7.insert(meas7.end(), m.data(), m.data() + 7);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) info36.push_back(oe->information()(i, j));
      } else {
        throw std::runtime_error("BlockSolverHIP: unsupported edge type (no CPU fallback by design)");
      }
    }
    if (cs_ba_append_edges_proj(ba_, (int)e_pt.size(), e_pt.data(), e_cam.data(), uv.data(), info4.data(), intr4.data(), huber.data()) != CS_OK) return false;
    if (cs_ba_append_edges_cuboid(ba_, (int)ce_cam.size(), ce_cam.data(), ce_cub.data(), meas10.data(), info81.data()) != CS_OK) return false;
    if (cs_ba_append_edges_cuboid_proj(ba_, (int)pe_cam.size(), pe_cam.data(), pe_cub.data(), meas4.data(), info16.data(), K9.data()) != CS_OK) return false;
    if (cs_ba_append_edges_odom(ba_, (int)oe_i.size(), oe_i.data(), oe_j.data(), meas7.data(), info36.data()) != CS_OK) return false;
    int sp = 0, sl = 0;
    if (cs_ba_sizes(ba_, &sp, &sl) != CS_OK) return false;
    resizeVector(sp + sl);
    size_t tot = 0;
    for (auto* v : _optimizer->indexMapping()) tot += (size_t)v->dimension() * v->dimension();
    diag_.assign(tot, 0.0);
    size_t off = 0;
    for (auto* v : _optimizer->indexMapping()) { v->mapHessianMemory(diag_.data() + off); off += (size_t)v->dimension() * v->dimension(); }
    return true;
  }

  // core/block_solver.hpp:501-560.  Estimates may have changed on the CPU side (update/pop): push them first.
  virtual bool buildSystem() {
    upload_estimates();
    double chi;
    if (cs_ba_compute_errors(ba_, &chi) != CS_OK || cs_ba_build_system(ba_) != CS_OK) return false;
    if (cs_ba_get_system(ba_, nullptr, nullptr, nullptr, _b, nullptr) != CS_OK) return false;
    // refresh the vertices' mapped A_ii (symmetric blocks: Eigen's column-major view reads the same numbers)
    hc_.resize(36 * cams_.size()); ho_.resize(81 * cubs_.size()); hp_.resize(9 * pts_.size());
    if (cs_ba_get_vertex_hessians(ba_, hc_.data(), ho_.data(), hp_.data()) != CS_OK) return false;
    size_t off = 0;
    for (auto* v : _optimizer->indexMapping()) {
      const int d = v->dimension(), i = index_[v];
      const double* src = d == 6 ? &hc_[36 * (size_t)i] : (d == 9 ? &ho_[81 * (size_t)i] : &hp_[9 * (size_t)i]);
      std::copy(src, src + (size_t)d * d, diag_.begin() + off);
      off += (size_t)d * d;
    }
    return true;
  }
  virtual bool setLambda(double lambda, bool /*backup*/ = false) { lambda_ = lambda; return true; }  // :563-589
  virtual void restoreDiagonal() {}                                                                   // :591-604: nothing was modified
  virtual bool solve() {                                                                              // :353-486
    int pd = 0;
    if (cs_ba_solve(ba_, lambda_, &pd) != CS_OK || !pd) return false;
    return cs_ba_get_system(ba_, nullptr, nullptr, nullptr, nullptr, _x) == CS_OK;
  }
  virtual bool computeMarginals(g2o::SparseBlockMatrix<g2o::MatrixXd>&, const std::vector<std::pair<int, int> >&) { return false; }
  virtual bool schur() { return true; }
  virtual void setSchur(bool) {}
  virtual bool supportsSchur() { return true; }
  virtual void setWriteDebug(bool) {}
  virtual bool writeDebug() const { return false; }
  virtual bool saveHessian(const std::string&) const { return false; }

 private:
  void upload_estimates() {  // estimates may have been changed by g2o's update()/pop() since the last call
    std::vector<double> cam7, cub10, pt3;
    for (auto* c : cams_) { g2o::Vector7d e = c->estimate().toVector(); cam7.insert(cam7.end(), e.data(), e.data() + 7); }
    for (auto* o : cubs_) { Vector10d e = o->estimate().toVector(); cub10.insert(cub10.end(), e.data(), e.data() + 10); }
    for (auto* p : pts_) pt3.insert(pt3.end(), p->estimate().data(), p->estimate().data() + 3);
    if (cs_ba_set_estimates(ba_, cam7.data(), cub10.data(), pt3.data()) != CS_OK) throw std::runtime_error(cs_last_error());
  }
  cs_ba* ba_ = nullptr;
  double lambda_ = 0;
  std::vector<g2o::VertexSE3Expmap*> cams_;
  std::vector<g2o::VertexCuboid*> cubs_;
  std::vector<g2o::VertexSBAPointXYZ*> pts_;
  std::map<g2o::HyperGraph::Vertex*, int> index_;
  std::vector<double> diag_, hc_, ho_, hp_;   // the vertices' mapped diagonal blocks (indexMapping order) and their staging copies
};

}  // namespace cubeslam
