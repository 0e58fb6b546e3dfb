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
// tests/test_ba_gpu.py::test_stepwise_abi_driven_like_g2o_levenberg drives the same call sequence from a host-side LM loop.
//
// What runs where:
//   * VertexSE3Expmap, VertexSBAPointXYZ (marginalised), VertexCuboid with EdgeSE3ProjectXYZ, EdgeSE3Cuboid, EdgeSE3CuboidProj,
//     EdgeSE3Expmap are evaluated on the device, with the robust kernels the library implements (Huber, PseudoHuber, Cauchy,
//     Saturated, DCS) on any of them.
//   * Any other active edge between those vertices -- or one of the four with a kernel the device cannot take (RobustKernelTukey keeps
//     its widths in private members, RobustKernelScaleDelta wraps another kernel, user-defined classes) -- goes through its OWN
//     virtuals on the CPU, as in BlockSolver::buildSystem (core/block_solver.hpp:519-528): linearizeOplus(jacobianWorkspace) +
//     constructQuadraticForm(), into memory this adapter maps (the vertices' A_ii / b_i, the edge's off-diagonal block); the sums
//     are handed to the library (cs_ba_set_external_edges / cs_ba_set_external_terms) and join the device's system before the Schur
//     complement.  Unary edges on any vertex and binary edges between cameras / cuboids qualify; an unknown edge that touches a
//     marginalised point together with another vertex, has more than two vertices, or touches an unknown vertex type is refused.
//   * Refusals follow g2o's conventions (core/solver.h:53-132 declares bool-returning, exception-free virtuals): a message on
//     std::cerr and `false`, never a throw.
//   * Vertex ids need no particular order: x() / b() are permuted between g2o's indexMapping() order (SparseOptimizer::update walks
//     it, sparse_optimizer.cpp:422-435) and the library's per-class layout.
//
// Needs g2o + Eigen headers; not compiled in the build container (neither is installed there) beyond the syntax check of
// tests/test_adapters_compile.py against declaration-only stand-ins.
#pragma once

#include <algorithm>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "Thirdparty/g2o/g2o/core/solver.h"
#include "Thirdparty/g2o/g2o/core/sparse_block_matrix.h"
#include "Thirdparty/g2o/g2o/core/sparse_optimizer.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"
#include "object_slam/g2o_Object.h"

#include "cubeslam_hip.h"

namespace cubeslam {

class BlockSolverHIP : public g2o::Solver {
 public:
  explicit BlockSolverHIP(int device = 0) : device_(device) {}
  virtual ~BlockSolverHIP() { if (ba_) cs_ba_destroy(ba_); }

  virtual bool init(g2o::SparseOptimizer* optimizer, bool /*online*/ = false) {
    _optimizer = optimizer;
    if (!ba_ && cs_ba_create(device_, &ba_) != CS_OK) return fail("cs_ba_create");
    return true;
  }

  // core/block_solver.hpp:142-295: index mapping -> flat arrays.
  virtual bool buildStructure(bool /*zeroBlocks*/ = false) {
    if (!ba_ && cs_ba_create(device_, &ba_) != CS_OK) return fail("cs_ba_create");
    cams_.clear(); cubs_.clear(); pts_.clear(); index_.clear(); ext_edges_.clear();
    for (int c = 0; c < 4; c++) { rk_[c].clear(); rd_[c].clear(); }
    VertexPack V;
    for (auto* v : _optimizer->activeVertices()) if (!add_vertex(v, V)) return false;
    if (cs_ba_set_vertices(ba_, V.cam7.data(), V.cam_fixed.data(), (int)V.cam_fixed.size(), V.cub10.data(), V.cub_fixed.data(), (int)V.cub_fixed.size(),
                           V.pt3.data(), V.pt_fixed.data(), (int)V.pt_fixed.size(), 0) != CS_OK) return fail("cs_ba_set_vertices");
    EdgePack E;
    for (auto* e : _optimizer->activeEdges()) if (!add_edge(e, E)) return false;
    if (cs_ba_set_edges_proj(ba_, (int)E.e_pt.size(), E.e_pt.data(), E.e_cam.data(), E.uv.data(), E.info4.data(), E.intr4.data(), E.huber.data()) != CS_OK ||
        cs_ba_set_edges_cuboid(ba_, (int)E.ce_cam.size(), E.ce_cam.data(), E.ce_cub.data(), E.meas10.data(), E.info81.data()) != CS_OK ||
        cs_ba_set_edges_cuboid_proj(ba_, (int)E.pe_cam.size(), E.pe_cam.data(), E.pe_cub.data(), E.meas4.data(), E.info16.data(), E.K9.data()) != CS_OK ||
        cs_ba_set_edges_odom(ba_, (int)E.oe_i.size(), E.oe_i.data(), E.oe_j.data(), E.meas7.data(), E.info36.data()) != CS_OK) return fail("cs_ba_set_edges_*");
    return finish_structure();
  }

  // core/block_solver.hpp:297-350 (online processing: the graph grew by `vset` / `edges`).  SparseOptimizer::updateInitialization
  // passes only the NEW, NON-FIXED, non-marginalised vertices (sparse_optimizer.cpp:459-479): a new fixed camera, a new point or a new
  // fixed point referenced by a new edge arrives through the edge's end points alone.  Everything new is appended behind what the
  // library holds (cs_ba_append_*: it keeps its estimates and re-runs its structure phase on the next solve); an end point that can
  // neither be found nor appended packs the graph again from scratch.
  virtual bool updateStructure(const std::vector<g2o::HyperGraph::Vertex*>& vset, const g2o::HyperGraph::EdgeSet& edges) {
    if (!ba_) return buildStructure();
    VertexPack V;
    for (auto* hv : vset) if (index_.find(hv) == index_.end() && !add_vertex(static_cast<g2o::OptimizableGraph::Vertex*>(hv), V)) return false;
    for (auto* he : edges)
      for (auto* hv : he->vertices())
        if (hv && index_.find(hv) == index_.end() && !add_vertex(static_cast<g2o::OptimizableGraph::Vertex*>(hv), V)) return false;
    if (cs_ba_append_vertices(ba_, V.cam7.data(), V.cam_fixed.data(), (int)V.cam_fixed.size(), V.cub10.data(), V.cub_fixed.data(), (int)V.cub_fixed.size(),
                              V.pt3.data(), V.pt_fixed.data(), (int)V.pt_fixed.size()) != CS_OK) return fail("cs_ba_append_vertices");
    EdgePack E;
    for (auto* he : edges) {
      auto* e = static_cast<g2o::OptimizableGraph::Edge*>(he);
      bool known = true;
      for (auto* hv : e->vertices()) known = known && hv && index_.find(hv) != index_.end();
      if (!known) return buildStructure();
      if (!add_edge(e, E)) return false;
    }
    if (cs_ba_append_edges_proj(ba_, (int)E.e_pt.size(), E.e_pt.data(), E.e_cam.data(), E.uv.data(), E.info4.data(), E.intr4.data(), E.huber.data()) != CS_OK ||
        cs_ba_append_edges_cuboid(ba_, (int)E.ce_cam.size(), E.ce_cam.data(), E.ce_cub.data(), E.meas10.data(), E.info81.data()) != CS_OK ||
        cs_ba_append_edges_cuboid_proj(ba_, (int)E.pe_cam.size(), E.pe_cam.data(), E.pe_cub.data(), E.meas4.data(), E.info16.data(), E.K9.data()) != CS_OK ||
        cs_ba_append_edges_odom(ba_, (int)E.oe_i.size(), E.oe_i.data(), E.oe_j.data(), E.meas7.data(), E.info36.data()) != CS_OK) return fail("cs_ba_append_edges_*");
    return finish_structure();
  }

  // core/block_solver.hpp:501-560.  Estimates may have changed on the CPU side (update/pop): push them first.
  virtual bool buildSystem() {
    if (!upload_estimates()) return false;
    if (!ext_edges_.empty() && !evaluate_external_edges()) return false;
    double chi;
    if (cs_ba_compute_errors(ba_, &chi) != CS_OK || cs_ba_build_system(ba_) != CS_OK) return fail("cs_ba_build_system");
    lib_.resize(_xSize);
    if (cs_ba_get_system(ba_, nullptr, nullptr, nullptr, lib_.data(), nullptr) != CS_OK) return fail("cs_ba_get_system");
    for (const Perm& p : perm_) std::copy(lib_.begin() + p.lib, lib_.begin() + p.lib + p.dim, _b + p.g2o);
    // refresh the vertices' mapped A_ii (symmetric blocks: Eigen's column-major view reads the same numbers)
    hc_.resize(36 * cams_.size()); ho_.resize(81 * cubs_.size()); hp_.resize(9 * pts_.size());
    if (cs_ba_get_vertex_hessians(ba_, hc_.data(), ho_.data(), hp_.data()) != CS_OK) return fail("cs_ba_get_vertex_hessians");
    size_t off = 0;
    for (auto* v : _optimizer->indexMapping()) {
      const int d = v->dimension();
      const Slot s = index_[v];
      const double* src = s.cls == CS_VERTEX_CAM ? &hc_[36 * (size_t)s.idx] : (s.cls == CS_VERTEX_CUBOID ? &ho_[81 * (size_t)s.idx] : &hp_[9 * (size_t)s.idx]);
      std::copy(src, src + (size_t)d * d, diag_.begin() + off);
      off += (size_t)d * d;
    }
    return true;
  }
  virtual bool setLambda(double lambda, bool /*backup*/ = false) { lambda_ = lambda; return true; }  // :563-589
  virtual void restoreDiagonal() {}                                                                   // :591-604: nothing was modified
  virtual bool solve() {                                                                              // :353-486
    int pd = 0;
    if (cs_ba_solve(ba_, lambda_, &pd) != CS_OK) return fail("cs_ba_solve");
    if (!pd) return false;                                    // "Cholesky failure": LM raises lambda and retries (:580-584)
    lib_.resize(_xSize);
    if (cs_ba_get_system(ba_, nullptr, nullptr, nullptr, nullptr, lib_.data()) != CS_OK) return fail("cs_ba_get_system");
    for (const Perm& p : perm_) std::copy(lib_.begin() + p.lib, lib_.begin() + p.lib + p.dim, _x + p.g2o);
    return true;
  }
  // core/block_solver.hpp:488-499 -> LinearSolver::solvePattern(spinv, blockIndices, *_Hpp) -> MarginalCovarianceCholesky::computeCovariance
  // (core/marginal_covariance_cholesky.cpp:154-222): spinv is re-created over the pose blocks' row layout and receives block (first, second) of
  // the inverse of H_pp for every requested pair of hessian indices.  cs_ba_pose_marginals does the factorisation and the solves on the device.
  // (Pairs that name a marginalised vertex -- a point -- are outside _Hpp in the reference too: refused.)
  virtual bool computeMarginals(g2o::SparseBlockMatrix<g2o::MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) {
    if (!ba_) return refuse("computeMarginals before buildStructure");
    const auto& im = _optimizer->indexMapping();
    std::vector<int> rbi;                                  // end column of every non-marginalised vertex's block, in hessian-index order
    std::vector<Slot> slot_of;
    int end = 0;
    for (auto* v : im) {
      if (v->marginalized()) break;                        // (indexMapping: the non-marginalised vertices first, sparse_optimizer.cpp:166-190)
      end += v->dimension();
      rbi.push_back(end);
      slot_of.push_back(index_[v]);
    }
    if (rbi.empty()) return refuse("computeMarginals: no pose vertex in the graph");
    std::vector<int> ci, ii, cj, ij;
    size_t total = 0;
    for (const auto& pr : blockIndices) {
      if (pr.first < 0 || pr.second < 0 || pr.first >= (int)rbi.size() || pr.second >= (int)rbi.size()) return refuse("computeMarginals: a block index outside the pose block (a marginalised vertex?)");
      const Slot a = slot_of[pr.first], b = slot_of[pr.second];
      ci.push_back(a.cls); ii.push_back(a.idx); cj.push_back(b.cls); ij.push_back(b.idx);
      total += (size_t)(a.cls == CS_VERTEX_CAM ? 6 : 9) * (size_t)(b.cls == CS_VERTEX_CAM ? 6 : 9);
    }
    std::vector<double> blocks(total + 1);
    int pd = 1;
    if (cs_ba_pose_marginals(ba_, (int)blockIndices.size(), ci.data(), ii.data(), cj.data(), ij.data(), blocks.data(), &pd) != CS_OK) return fail("cs_ba_pose_marginals");
    if (!pd) return false;                                 // (Cholesky failure: the reference's solvePattern returns false)
    spinv = g2o::SparseBlockMatrix<g2o::MatrixXd>(&rbi[0], &rbi[0], (int)rbi.size(), (int)rbi.size(), true);
    size_t o = 0;
    for (size_t k = 0; k < blockIndices.size(); k++) {
      g2o::MatrixXd* blk = spinv.block(blockIndices[k].first, blockIndices[k].second, true);
      if (!blk) return refuse("computeMarginals: SparseBlockMatrix::block returned null");
      const int dr = ci[k] == CS_VERTEX_CAM ? 6 : 9, dc = cj[k] == CS_VERTEX_CAM ? 6 : 9;
      for (int r = 0; r < dr; r++) for (int c = 0; c < dc; c++) (*blk)(r, c) = blocks[o + (size_t)r * dc + c];
      o += (size_t)dr * dc;
    }
    return true;
  }
  virtual bool schur() { return true; }
  virtual void setSchur(bool) {}
  virtual bool supportsSchur() { return true; }
  virtual void setWriteDebug(bool) {}
  virtual bool writeDebug() const { return false; }
  virtual bool saveHessian(const std::string&) const { return false; }

 private:
  struct Slot { int cls, idx; };                       // library-side identity of a vertex: cs_vertex_class, index in its class
  struct Perm { size_t g2o, lib; int dim; };           // one vertex's entries in x() / b(): g2o's offset, the library's
  struct VertexPack { std::vector<double> cam7, cub10, pt3; std::vector<int> cam_fixed, cub_fixed, pt_fixed; };
  struct EdgePack {
    std::vector<int> e_pt, e_cam, ce_cam, ce_cub, pe_cam, pe_cub, oe_i, oe_j;
    std::vector<double> uv, info4, intr4, huber, meas10, info81, meas7, info36, meas4, info16, K9;
  };

  static bool fail(const char* what) { std::cerr << "BlockSolverHIP: " << what << ": " << cs_last_error() << std::endl; return false; }
  static bool refuse(const std::string& why) { std::cerr << "BlockSolverHIP: " << why << std::endl; return false; }

  bool add_vertex(g2o::OptimizableGraph::Vertex* v, VertexPack& V) {
    if (auto* c = dynamic_cast<g2o::VertexSE3Expmap*>(v)) {
      if (c->marginalized()) return refuse("a marginalised camera vertex is not supported");
      index_[v] = Slot{CS_VERTEX_CAM, (int)cams_.size()}; cams_.push_back(c); V.cam_fixed.push_back(c->fixed());
      g2o::Vector7d e = c->estimate().toVector(); V.cam7.insert(V.cam7.end(), e.data(), e.data() + 7);
    } else if (auto* o = dynamic_cast<g2o::VertexCuboid*>(v)) {
      if (o->marginalized()) return refuse("a marginalised cuboid vertex is not supported");
      index_[v] = Slot{CS_VERTEX_CUBOID, (int)cubs_.size()}; cubs_.push_back(o); V.cub_fixed.push_back(o->fixed());
      Vector10d e = o->estimate().toVector(); V.cub10.insert(V.cub10.end(), e.data(), e.data() + 10);
    } else if (auto* p = dynamic_cast<g2o::VertexSBAPointXYZ*>(v)) {
      if (!p->fixed() && !p->marginalized()) return refuse("point vertices must be marginalised (setMarginalized(true)): the library solves the pose block through the Schur complement");
      index_[v] = Slot{CS_VERTEX_POINT, (int)pts_.size()}; pts_.push_back(p); V.pt_fixed.push_back(p->fixed());
      V.pt3.insert(V.pt3.end(), p->estimate().data(), p->estimate().data() + 3);
    } else {
      return refuse("unsupported vertex type (neither VertexSE3Expmap, VertexCuboid nor VertexSBAPointXYZ)");
    }
    return true;
  }

  // RobustKernel -> (cs_robust_kernel, delta); false: a kernel the device cannot evaluate (the edge then takes the CPU path, whose
  // constructQuadraticForm applies whatever kernel the edge carries)
  // KERNEL_OTHER: a kernel class the device does not know; KERNEL_BAD_DELTA: one it knows, with delta <= 0 -- the reference's formulas
  // (robust_kernel_impl.cpp:78-135) divide by delta^2 or give zero / negative weights there, nothing the device reproduces; such an edge
  // takes the CPU path like KERNEL_OTHER, and a camera-point edge is refused with a message that names delta.
  // Restriction: the device's Huber threshold is float(delta * delta), what RobustKernelHuber::setDelta stores (robust_kernel_impl.cpp:65-69,
  // `float dsqr`).  A kernel configured through setDeltaSqr(delta, dsqr) with dsqr != delta^2 (robust_kernel_impl.cpp:72-76) cannot be
  // told apart from outside (dsqr is private) and is evaluated with delta^2: configure such edges' kernels through setDelta.
  enum KernelClass { KERNEL_DEVICE = 0, KERNEL_OTHER = 1, KERNEL_BAD_DELTA = 2 };
  static KernelClass kernel_of(const g2o::OptimizableGraph::Edge* e, int* kind, double* delta) {
    g2o::RobustKernel* rk = e->robustKernel();
    *kind = CS_RK_NONE; *delta = 0.0;
    if (!rk) return KERNEL_DEVICE;
    *delta = rk->delta();
    if (dynamic_cast<g2o::RobustKernelHuber*>(rk)) *kind = CS_RK_HUBER;
    else if (dynamic_cast<g2o::RobustKernelPseudoHuber*>(rk)) *kind = CS_RK_PSEUDO_HUBER;
    else if (dynamic_cast<g2o::RobustKernelCauchy*>(rk)) *kind = CS_RK_CAUCHY;
    else if (dynamic_cast<g2o::RobustKernelSaturated*>(rk)) *kind = CS_RK_SATURATED;
    else if (dynamic_cast<g2o::RobustKernelDCS*>(rk)) *kind = CS_RK_DCS;
    else return KERNEL_OTHER;
    return *delta > 0.0 ? KERNEL_DEVICE : KERNEL_BAD_DELTA;   // (NaN: bad)
  }

  bool add_edge(g2o::OptimizableGraph::Edge* e, EdgePack& E) {
    int kind = 0; double delta = 0;
    const KernelClass kclass = kernel_of(e, &kind, &delta);
    const bool dev_kernel = kclass == KERNEL_DEVICE;
    for (auto* hv : e->vertices()) if (!hv || index_.find(hv) == index_.end()) return refuse("an active edge ends in a vertex that is not an active vertex");
    auto slot = [&](size_t k) { return index_[e->vertex(k)]; };
    if (auto* pe = dynamic_cast<g2o::EdgeSE3ProjectXYZ*>(e)) {
      if (kclass == KERNEL_BAD_DELTA) return refuse("EdgeSE3ProjectXYZ with a robust kernel whose delta is not positive: set a delta > 0 or remove the kernel (a camera-point edge cannot take the CPU path, it is part of the Schur structure)");
      if (!dev_kernel) return refuse("EdgeSE3ProjectXYZ with a robust kernel other than Huber / PseudoHuber / Cauchy / Saturated / DCS: a camera-point edge cannot take the CPU path (it is part of the Schur structure)");
      E.e_pt.push_back(slot(0).idx); E.e_cam.push_back(slot(1).idx);
      E.uv.push_back(pe->measurement()[0]); E.uv.push_back(pe->measurement()[1]);
      for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) E.info4.push_back(pe->information()(i, j));
      E.intr4.push_back(pe->fx); E.intr4.push_back(pe->fy); E.intr4.push_back(pe->cx); E.intr4.push_back(pe->cy);
      E.huber.push_back(kind == CS_RK_HUBER ? delta : 0.0);
      rk_[CS_EDGE_PROJ].push_back(kind); rd_[CS_EDGE_PROJ].push_back(delta);
      return true;
    }
    if (dev_kernel) {
      if (auto* ce = dynamic_cast<g2o::EdgeSE3Cuboid*>(e)) {
        E.ce_cam.push_back(slot(0).idx); E.ce_cub.push_back(slot(1).idx);
        Vector10d m = ce->measurement().toVector(); E.meas10.insert(E.meas10.end(), m.data(), m.data() + 10);
        for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) E.info81.push_back(ce->information()(i, j));
        rk_[CS_EDGE_CUBOID].push_back(kind); rd_[CS_EDGE_CUBOID].push_back(delta);
        return true;
      }
      if (auto* qe = dynamic_cast<g2o::EdgeSE3CuboidProj*>(e)) {
        E.pe_cam.push_back(slot(0).idx); E.pe_cub.push_back(slot(1).idx);
        for (int i = 0; i < 4; i++) E.meas4.push_back(qe->measurement()[i]);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) E.info16.push_back(qe->information()(i, j));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E.K9.push_back(qe->Kalib(i, j));
        rk_[CS_EDGE_CUBOID_PROJ].push_back(kind); rd_[CS_EDGE_CUBOID_PROJ].push_back(delta);
        return true;
      }
      if (auto* oe = dynamic_cast<g2o::EdgeSE3Expmap*>(e)) {
        E.oe_i.push_back(slot(0).idx); E.oe_j.push_back(slot(1).idx);
        g2o::Vector7d m = oe->measurement().toVector(); E.meas7.insert(E.meas7.end(), m.data(), m.data() + 7);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) E.info36.push_back(oe->information()(i, j));
        rk_[CS_EDGE_ODOM].push_back(kind); rd_[CS_EDGE_ODOM].push_back(delta);
        return true;
      }
    }
    // the CPU path: the edge's own linearizeOplus / constructQuadraticForm (evaluate_external_edges)
    const size_t nv = e->vertices().size();
    if (nv < 1 || nv > 2) return refuse("an edge type the device does not evaluate with more than two vertices");
    const Slot a = slot(0), b = nv == 2 ? slot(1) : Slot{0, -1};
    if (nv == 2 && (a.cls == CS_VERTEX_POINT || b.cls == CS_VERTEX_POINT))
      return refuse("an edge type the device does not evaluate that couples a marginalised point to another vertex (it would change the Schur structure)");
    ext_edges_.push_back(e);
    return true;
  }

  // after the vertex / edge lists changed: kernels, the external edges' pattern and mapped memory, x() / b() size and permutation
  bool finish_structure() {
    // Huber-or-none projection edges travelled in the `huber` argument; anything else needs the kind array
    bool proj_generic = false;
    for (int k : rk_[CS_EDGE_PROJ]) proj_generic = proj_generic || (k != CS_RK_NONE && k != CS_RK_HUBER);
    for (int c = 0; c < 4; c++) {
      bool any = false;
      for (int k : rk_[c]) any = any || k != CS_RK_NONE;
      if (c == CS_EDGE_PROJ ? proj_generic : any)
        if (cs_ba_set_robust_kernels(ba_, c, (int)rk_[c].size(), rk_[c].data(), rd_[c].data()) != CS_OK) return fail("cs_ba_set_robust_kernels");
    }
    std::vector<int> ci, ii, cj, ij;
    ext_blk_.assign(81 * ext_edges_.size(), 0.0);
    for (size_t k = 0; k < ext_edges_.size(); k++) {
      g2o::OptimizableGraph::Edge* e = ext_edges_[k];
      const Slot a = index_[e->vertex(0)], b = e->vertices().size() == 2 ? index_[e->vertex(1)] : Slot{0, -1};
      ci.push_back(a.cls); ii.push_back(a.idx); cj.push_back(b.cls); ij.push_back(b.idx);
      if (b.idx >= 0) {
        auto* v0 = static_cast<g2o::OptimizableGraph::Vertex*>(e->vertex(0));
        auto* v1 = static_cast<g2o::OptimizableGraph::Vertex*>(e->vertex(1));
        // (block_solver.hpp:228-233 maps the block of two free vertices; column-major dim_0 x dim_1, not transposed)
        if (!v0->fixed() && !v1->fixed()) e->mapHessianMemory(&ext_blk_[81 * k], 0, 1, false);
      }
    }
    if (cs_ba_set_external_edges(ba_, (int)ext_edges_.size(), ci.data(), ii.data(), cj.data(), ij.data()) != CS_OK) return fail("cs_ba_set_external_edges");
    int sp = 0, sl = 0;
    if (cs_ba_sizes(ba_, &sp, &sl) != CS_OK) return fail("cs_ba_sizes");
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
    // x() / b(): g2o's order is indexMapping() (poses, then the marginalised points); the library's is [free cameras | free cuboids]
    // in its own class order, then the free points
    std::vector<size_t> lib_cam(cams_.size(), 0), lib_cub(cubs_.size(), 0), lib_pt(pts_.size(), 0);
    size_t col = 0;
    for (size_t i = 0; i < cams_.size(); i++) if (!cams_[i]->fixed()) { lib_cam[i] = col; col += 6; }
    for (size_t i = 0; i < cubs_.size(); i++) if (!cubs_[i]->fixed()) { lib_cub[i] = col; col += 9; }
    if ((int)col != sp) return refuse("internal: pose block size disagrees with the library's");
    for (size_t i = 0; i < pts_.size(); i++) if (!pts_[i]->fixed()) { lib_pt[i] = col; col += 3; }
    if (col != (size_t)(sp + sl)) return refuse("internal: landmark block size disagrees with the library's");
    perm_.clear();
    size_t g = 0;
    for (auto* v : _optimizer->indexMapping()) {
      auto it = index_.find(v);
      if (it == index_.end()) return refuse("a vertex of indexMapping() is not part of the packed graph");
      const Slot s = it->second;
      perm_.push_back(Perm{g, s.cls == CS_VERTEX_CAM ? lib_cam[s.idx] : (s.cls == CS_VERTEX_CUBOID ? lib_cub[s.idx] : lib_pt[s.idx]), v->dimension()});
      g += (size_t)v->dimension();
    }
    if (g != (size_t)(sp + sl)) return refuse("indexMapping() and the packed graph disagree on the number of unknowns");
    return true;
  }

  // The CPU path of the edges in ext_edges_: what BlockSolver::buildSystem does for every edge (block_solver.hpp:505-528), for these
  // alone -- clear the vertices' b and the mapped blocks, linearizeOplus + constructQuadraticForm, collect.  The errors are current:
  // OptimizationAlgorithmLevenberg calls computeActiveErrors() before buildSystem() (optimization_algorithm_levenberg.cpp:67-87).
  bool evaluate_external_edges() {
    for (auto* v : _optimizer->indexMapping()) v->clearQuadraticForm();
    std::fill(diag_.begin(), diag_.end(), 0.0);
    std::fill(ext_blk_.begin(), ext_blk_.end(), 0.0);
    g2o::JacobianWorkspace& jw = _optimizer->jacobianWorkspace();
    for (auto* e : ext_edges_) { e->linearizeOplus(jw); e->constructQuadraticForm(); }
    x36_.assign(36 * cams_.size(), 0.0); x6_.assign(6 * cams_.size(), 0.0); x81_.assign(81 * cubs_.size(), 0.0); x9_.assign(9 * cubs_.size(), 0.0);
    y9_.assign(9 * pts_.size(), 0.0); y3_.assign(3 * pts_.size(), 0.0);
    size_t off = 0;
    for (auto* v : _optimizer->indexMapping()) {
      const int d = v->dimension();
      const Slot s = index_[v];
      double* A = s.cls == CS_VERTEX_CAM ? &x36_[36 * (size_t)s.idx] : (s.cls == CS_VERTEX_CUBOID ? &x81_[81 * (size_t)s.idx] : &y9_[9 * (size_t)s.idx]);
      double* bb = s.cls == CS_VERTEX_CAM ? &x6_[6 * (size_t)s.idx] : (s.cls == CS_VERTEX_CUBOID ? &x9_[9 * (size_t)s.idx] : &y3_[3 * (size_t)s.idx]);
      std::copy(diag_.begin() + off, diag_.begin() + off + (size_t)d * d, A);       // symmetric: column-major == row-major
      for (int k = 0; k < d; k++) bb[k] = v->b(k);
      off += (size_t)d * d;
    }
    // the edges' blocks: Eigen column-major dim_0 x dim_1 -> row-major
    xij_.assign(81 * ext_edges_.size(), 0.0);
    for (size_t k = 0; k < ext_edges_.size(); k++) {
      g2o::OptimizableGraph::Edge* e = ext_edges_[k];
      if (e->vertices().size() != 2) continue;
      const int d0 = static_cast<g2o::OptimizableGraph::Vertex*>(e->vertex(0))->dimension(), d1 = static_cast<g2o::OptimizableGraph::Vertex*>(e->vertex(1))->dimension();
      for (int r = 0; r < d0; r++) for (int c = 0; c < d1; c++) xij_[81 * k + (size_t)r * d1 + c] = ext_blk_[81 * k + (size_t)c * d0 + r];
    }
    // chi2 stays with the optimizer in this flow (activeRobustChi2 runs on the CPU over all edges), so the library's share is 0
    if (cs_ba_set_external_terms(ba_, x36_.data(), x6_.data(), cubs_.empty() ? nullptr : x81_.data(), cubs_.empty() ? nullptr : x9_.data(),
                                 pts_.empty() ? nullptr : y9_.data(), pts_.empty() ? nullptr : y3_.data(), xij_.data(), 0.0) != CS_OK) return fail("cs_ba_set_external_terms");
    return true;
  }

  bool upload_estimates() {  // estimates may have been changed by g2o's update()/pop() since the last call
    std::vector<double> cam7, cub10, pt3;
    for (auto* c : cams_) { g2o::Vector7d e = c->estimate().toVector(); cam7.insert(cam7.end(), e.data(), e.data() + 7); }
    for (auto* o : cubs_) { Vector10d e = o->estimate().toVector(); cub10.insert(cub10.end(), e.data(), e.data() + 10); }
    for (auto* p : pts_) pt3.insert(pt3.end(), p->estimate().data(), p->estimate().data() + 3);
    if (cs_ba_set_estimates(ba_, cam7.data(), cub10.data(), pt3.data()) != CS_OK) return fail("cs_ba_set_estimates");
    return true;
  }

  int device_ = 0;
  cs_ba* ba_ = nullptr;
  double lambda_ = 0;
  std::vector<g2o::VertexSE3Expmap*> cams_;
  std::vector<g2o::VertexCuboid*> cubs_;
  std::vector<g2o::VertexSBAPointXYZ*> pts_;
  std::map<const g2o::HyperGraph::Vertex*, Slot> index_;
  std::vector<Perm> perm_;
  std::vector<int> rk_[4];                            // kernels per edge class (cs_edge_class), in the library's edge order
  std::vector<double> rd_[4];
  std::vector<g2o::OptimizableGraph::Edge*> ext_edges_;   // edges on the CPU path
  std::vector<double> ext_blk_;                           // their mapped off-diagonal blocks (81 per edge)
  std::vector<double> diag_, hc_, ho_, hp_, lib_;   // the vertices' mapped diagonal blocks (indexMapping order), staging copies, x / b in the library's order
  std::vector<double> x36_, x6_, x81_, x9_, y9_, y3_, xij_;
};

}  // namespace cubeslam
