// g2o_text.h -- the graph of the object-SLAM driver as g2o's text format (core/optimizable_graph.cpp:370-470 load, :586-610 save,
// :817-860 saveVertex / saveEdge: one element per line, `TAG id fields...` for a vertex, `FIX id` after a fixed one, `TAG id_0 id_1
// fields...` for an edge; `#` starts a comment).  Fields follow the vertex / edge classes' own read / write:
//   VERTEX_SE3:EXPMAP id  x y z qx qy qz qw                 VertexSE3Expmap::write (types/types_six_dof_expmap.cpp:55-73): the
//                                                           CAMERA-TO-WORLD pose, i.e. the inverse of the estimate the vertex holds
//   VERTEX_CUBOID id      x y z roll pitch yaw sx sy sz     VertexCuboid::write (object_slam/include/object_slam/g2o_Object.h:215-231):
//                                                           cuboid::toMinimalVector
//   EDGE_SE3:EXPMAP i j   x y z qx qy qz qw  I(0,0) I(0,1) ... I(5,5)     EdgeSE3Expmap::write (types_six_dof_expmap.cpp:75-103): the inverse
//                                                           of the measurement, then the upper triangle of the information matrix, row by row
//   EDGE_SE3_CUBOID c o   x y z roll pitch yaw sx sy sz  I(0,0) ... I(8,8)
//   VERTEX_XYZ id         x y z                             VertexSBAPointXYZ::write (types/types_sba.cpp:47-55)
//   EDGE_SE3_PROJECT_XYZ:EXPMAP p c   u v  I(0,0) I(0,1) I(1,1)   EdgeSE3ProjectXYZ::write (types/types_six_dof_expmap.cpp:134-146): vertex 0 the point,
//                                                           vertex 1 the camera; the measurement, then the information's upper triangle.  The class
//                                                           keeps fx fy cx cy as public MEMBERS set by the caller's code (types_six_dof_expmap.h:
//                                                           166-173) and never writes them, nor its robust kernel: so that a saved graph is the whole
//                                                           problem, two state lines of this library's own precede the edges they apply to,
//                                                             CS_INTRINSICS fx fy cx cy        and        CS_ROBUST_HUBER delta     (delta <= 0: none)
//                                                           -- g2o's loader would skip both as unknown tags, as this one skips any other unknown tag
// The reference's vendored g2o registers NO type with its Factory (no G2O_REGISTER_TYPE under object_slam/Thirdparty/g2o), so its own
// save() writes nothing for this graph, and EdgeSE3Cuboid::read / write are empty stubs (g2o_Object.h:241-248).  The two SE3 tags are
// upstream g2o's, as are VERTEX_XYZ and EDGE_SE3_PROJECT_XYZ:EXPMAP [tag names from upstream g2o's G2O_REGISTER_TYPE lines: general knowledge, not in the
// reference]; VERTEX_CUBOID / EDGE_SE3_CUBOID are this library's names, and the cuboid edge's fields are chosen the way the classes
// around it write theirs (measurement as the vertex writes its estimate, then the information's upper triangle).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "cs_se3.h"

namespace g2o_text {

struct Graph {
  // vertices in the layout of cs_ba_set_vertices: cameras world-to-camera (x y z qx qy qz qw), cuboids pose + half sizes (10)
  std::vector<int> cam_id, cub_id;
  std::vector<double> cam_Tcw, cuboids;
  std::vector<int> cam_fixed, cub_fixed;
  // edges by INDEX into the lists above (cs_ba_set_edges_*): measurement 10 + information 81, measurement 7 + information 36
  std::vector<int> ce_cam, ce_cub, oe_i, oe_j;
  std::vector<double> ce_meas, ce_info, oe_meas, oe_info;
  // landmarks (cs_ba_set_vertices: points3 / pt_fixed) and camera-point edges (cs_ba_set_edges_proj: uv 2, information 4 row-major,
  // intrinsics fx fy cx cy, Huber delta or <= 0)
  std::vector<int> pt_id, pt_fixed;
  std::vector<double> points;
  std::vector<int> pe_pt, pe_cam;
  std::vector<double> pe_uv, pe_info, pe_intr, pe_huber;
};

inline cs::Cube cuboid_from_minimal(const double* v) {     // cuboid::fromMinimalVector (g2o_Object.h:37-42), zyx_euler_to_quat (matrix_utils.cpp:19-33)
  const double roll = v[3], pitch = v[4], yaw = v[5];
  const double sy = std::sin(yaw * 0.5), cy = std::cos(yaw * 0.5), sp = std::sin(pitch * 0.5), cp = std::cos(pitch * 0.5);
  const double sr = std::sin(roll * 0.5), cr = std::cos(roll * 0.5);
  cs::Cube c;
  c.pose.qw = cr * cp * cy + sr * sp * sy;
  c.pose.qx = sr * cp * cy - cr * sp * sy;
  c.pose.qy = cr * sp * cy + sr * cp * sy;
  c.pose.qz = cr * cp * sy - sr * sp * cy;
  for (int d = 0; d < 3; d++) { c.pose.t[d] = v[d]; c.scale[d] = v[6 + d]; }
  cs::pose_normalize(c.pose);
  return c;
}
inline void cuboid_to_minimal(const cs::Cube& c, double* o) {   // toMinimalVector (g2o_Object.h:137-143; SE3Quat::toXYZPRYVector se3quat.h:196-222)
  const double qx = c.pose.qx, qy = c.pose.qy, qz = c.pose.qz, qw = c.pose.qw;
  for (int d = 0; d < 3; d++) { o[d] = c.pose.t[d]; o[6 + d] = c.scale[d]; }
  o[3] = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
  o[4] = std::asin(2 * (qw * qy - qz * qx));
  o[5] = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}

inline bool save(const std::string& path, const Graph& g, int digits = 17) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  auto put = [&](const double* v, int n) { for (int i = 0; i < n; i++) std::fprintf(f, " %.*g", digits, v[i]); };
  // vertices by ascending id, as OptimizableGraph::save does (it sorts the vertex set by id)
  std::map<int, std::pair<int, int>> order;     // id -> (kind, index)
  for (size_t i = 0; i < g.cam_id.size(); i++) order[g.cam_id[i]] = {0, (int)i};
  for (size_t i = 0; i < g.cub_id.size(); i++) order[g.cub_id[i]] = {1, (int)i};
  for (size_t i = 0; i < g.pt_id.size(); i++) order[g.pt_id[i]] = {2, (int)i};
  for (const auto& kv : order) {
    const int id = kv.first, i = kv.second.second;
    if (kv.second.first == 2) {
      std::fprintf(f, "VERTEX_XYZ %d", id); put(&g.points[3 * (size_t)i], 3); std::fprintf(f, "\n");
      if (g.pt_fixed[i]) std::fprintf(f, "FIX %d\n", id);
    } else if (kv.second.first == 0) {
      double twc[7];
      cs::pose_store(cs::pose_inv(cs::pose_load(&g.cam_Tcw[7 * (size_t)i])), twc);
      std::fprintf(f, "VERTEX_SE3:EXPMAP %d", id); put(twc, 7); std::fprintf(f, "\n");
      if (g.cam_fixed[i]) std::fprintf(f, "FIX %d\n", id);
    } else {
      double m[9];
      cuboid_to_minimal(cs::cube_load(&g.cuboids[10 * (size_t)i]), m);
      std::fprintf(f, "VERTEX_CUBOID %d", id); put(m, 9); std::fprintf(f, "\n");
      if (g.cub_fixed[i]) std::fprintf(f, "FIX %d\n", id);
    }
  }
  for (size_t k = 0; k < g.ce_cam.size(); k++) {
    double m[9];
    cuboid_to_minimal(cs::cube_load(&g.ce_meas[10 * k]), m);
    std::fprintf(f, "EDGE_SE3_CUBOID %d %d", g.cam_id[g.ce_cam[k]], g.cub_id[g.ce_cub[k]]); put(m, 9);
    for (int r = 0; r < 9; r++) put(&g.ce_info[81 * k + 9 * r + r], 9 - r);
    std::fprintf(f, "\n");
  }
  for (size_t k = 0; k < g.oe_i.size(); k++) {
    double inv[7];
    cs::pose_store(cs::pose_inv(cs::pose_load(&g.oe_meas[7 * k])), inv);
    std::fprintf(f, "EDGE_SE3:EXPMAP %d %d", g.cam_id[g.oe_i[k]], g.cam_id[g.oe_j[k]]); put(inv, 7);
    for (int r = 0; r < 6; r++) put(&g.oe_info[36 * k + 6 * r + r], 6 - r);
    std::fprintf(f, "\n");
  }
  {
    double intr[4] = {0, 0, 0, 0}, huber = 0;
    bool have_intr = false, have_huber = false;
    for (size_t k = 0; k < g.pe_pt.size(); k++) {
      const double* in = &g.pe_intr[4 * k];
      if (!have_intr || in[0] != intr[0] || in[1] != intr[1] || in[2] != intr[2] || in[3] != intr[3]) {
        std::fprintf(f, "CS_INTRINSICS"); put(in, 4); std::fprintf(f, "\n");
        for (int q = 0; q < 4; q++) intr[q] = in[q];
        have_intr = true;
      }
      const double hb = g.pe_huber.empty() ? 0.0 : g.pe_huber[k];
      if (!have_huber || hb != huber) { std::fprintf(f, "CS_ROBUST_HUBER"); put(&hb, 1); std::fprintf(f, "\n"); huber = hb; have_huber = true; }
      std::fprintf(f, "EDGE_SE3_PROJECT_XYZ:EXPMAP %d %d", g.pt_id[g.pe_pt[k]], g.cam_id[g.pe_cam[k]]); put(&g.pe_uv[2 * k], 2);
      std::fprintf(f, " "); put(&g.pe_info[4 * k], 2); put(&g.pe_info[4 * k + 3], 1);      // I(0,0) I(0,1) I(1,1)
      std::fprintf(f, "\n");
    }
  }
  const bool ok = std::ferror(f) == 0;
  std::fclose(f);
  return ok;
}

// Unknown tags are skipped with one warning each, a FIX of an unknown vertex warns, an edge on an unknown vertex is dropped with a
// warning -- OptimizableGraph::load's behaviour (optimizable_graph.cpp:390-470).  Returns false only if the file cannot be read.
inline bool load(const std::string& path, Graph& g, std::string* warnings = nullptr) {
  std::ifstream f(path.c_str());
  if (!f) return false;
  g = Graph();
  std::map<int, int> cam_of, cub_of, pt_of;
  std::map<std::string, int> unknown;
  double cur_intr[4] = {0, 0, 0, 0}, cur_huber = 0;
  bool have_intr = false;
  auto warn = [&](const std::string& m) { if (warnings) *warnings += m + "\n"; };
  std::string line;
  while (std::getline(f, line)) {
    std::stringstream ss(line);
    std::string tag;
    if (!(ss >> tag) || tag[0] == '#') continue;
    if (tag == "FIX") {
      int id;
      while (ss >> id) {
        if (cam_of.count(id)) g.cam_fixed[cam_of[id]] = 1;
        else if (cub_of.count(id)) g.cub_fixed[cub_of[id]] = 1;
        else if (pt_of.count(id)) g.pt_fixed[pt_of[id]] = 1;
        else warn("Warning: Unable to fix vertex with id " + std::to_string(id) + ". Not found in the graph.");
      }
    } else if (tag == "VERTEX_SE3:EXPMAP") {
      int id; double v[7];
      ss >> id;
      for (int i = 0; i < 7; i++) ss >> v[i];
      if (!ss || cam_of.count(id) || cub_of.count(id) || pt_of.count(id)) { warn("Failure adding Vertex, " + tag + " " + std::to_string(id)); continue; }
      cs::Pose twc = cs::pose_load(v);
      cs::pose_normalize(twc);                       // SE3Quat::fromVector normalises (se3quat.h:84-100)
      double tcw[7];
      cs::pose_store(cs::pose_inv(twc), tcw);        // setEstimate(cam2world.inverse())
      cam_of[id] = (int)g.cam_id.size();
      g.cam_id.push_back(id); g.cam_fixed.push_back(0); g.cam_Tcw.insert(g.cam_Tcw.end(), tcw, tcw + 7);
    } else if (tag == "VERTEX_CUBOID") {
      int id; double v[9], c10[10];
      ss >> id;
      for (int i = 0; i < 9; i++) ss >> v[i];
      if (!ss || cam_of.count(id) || cub_of.count(id) || pt_of.count(id)) { warn("Failure adding Vertex, " + tag + " " + std::to_string(id)); continue; }
      cs::cube_store(cuboid_from_minimal(v), c10);
      cub_of[id] = (int)g.cub_id.size();
      g.cub_id.push_back(id); g.cub_fixed.push_back(0); g.cuboids.insert(g.cuboids.end(), c10, c10 + 10);
    } else if (tag == "EDGE_SE3:EXPMAP") {
      int a, b; double v[7], info[36] = {0}, m[7];
      ss >> a >> b;
      for (int i = 0; i < 7; i++) ss >> v[i];
      for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { ss >> info[6 * r + c]; info[6 * c + r] = info[6 * r + c]; }
      if (!ss || !cam_of.count(a) || !cam_of.count(b)) { warn("Unable to find vertices for edge " + tag + " " + std::to_string(a) + " " + std::to_string(b)); continue; }
      cs::Pose twc = cs::pose_load(v);
      cs::pose_normalize(twc);
      cs::pose_store(cs::pose_inv(twc), m);          // setMeasurement(cam2world.inverse())
      g.oe_i.push_back(cam_of[a]); g.oe_j.push_back(cam_of[b]);
      g.oe_meas.insert(g.oe_meas.end(), m, m + 7); g.oe_info.insert(g.oe_info.end(), info, info + 36);
    } else if (tag == "EDGE_SE3_CUBOID") {
      int a, b; double v[9], info[81] = {0}, m[10];
      ss >> a >> b;
      for (int i = 0; i < 9; i++) ss >> v[i];
      for (int r = 0; r < 9; r++) for (int c = r; c < 9; c++) { ss >> info[9 * r + c]; info[9 * c + r] = info[9 * r + c]; }
      if (!ss || !cam_of.count(a) || !cub_of.count(b)) { warn("Unable to find vertices for edge " + tag + " " + std::to_string(a) + " " + std::to_string(b)); continue; }
      cs::cube_store(cuboid_from_minimal(v), m);
      g.ce_cam.push_back(cam_of[a]); g.ce_cub.push_back(cub_of[b]);
      g.ce_meas.insert(g.ce_meas.end(), m, m + 10); g.ce_info.insert(g.ce_info.end(), info, info + 81);
    } else if (tag == "VERTEX_XYZ") {
      int id; double v[3];
      ss >> id;
      for (int i = 0; i < 3; i++) ss >> v[i];
      if (!ss || cam_of.count(id) || cub_of.count(id) || pt_of.count(id)) { warn("Failure adding Vertex, " + tag + " " + std::to_string(id)); continue; }
      pt_of[id] = (int)g.pt_id.size();
      g.pt_id.push_back(id); g.pt_fixed.push_back(0); g.points.insert(g.points.end(), v, v + 3);
    } else if (tag == "CS_INTRINSICS") {
      double v[4];
      for (int i = 0; i < 4; i++) ss >> v[i];
      if (!ss) { warn("malformed CS_INTRINSICS line"); continue; }
      for (int i = 0; i < 4; i++) cur_intr[i] = v[i];
      have_intr = true;
    } else if (tag == "CS_ROBUST_HUBER") {
      double v;
      if (!(ss >> v)) { warn("malformed CS_ROBUST_HUBER line"); continue; }
      cur_huber = v;
    } else if (tag == "EDGE_SE3_PROJECT_XYZ:EXPMAP") {
      int a, b; double uv[2], i00, i01, i11;
      ss >> a >> b >> uv[0] >> uv[1] >> i00 >> i01 >> i11;
      if (!ss || !pt_of.count(a) || !cam_of.count(b)) { warn("Unable to find vertices for edge " + tag + " " + std::to_string(a) + " " + std::to_string(b)); continue; }
      if (!have_intr) { warn("edge " + tag + " " + std::to_string(a) + " " + std::to_string(b) + " dropped: fx fy cx cy are members the class does not write -- put a CS_INTRINSICS line before the edges"); continue; }
      const double info[4] = {i00, i01, i01, i11};
      g.pe_pt.push_back(pt_of[a]); g.pe_cam.push_back(cam_of[b]);
      g.pe_uv.insert(g.pe_uv.end(), uv, uv + 2); g.pe_info.insert(g.pe_info.end(), info, info + 4);
      g.pe_intr.insert(g.pe_intr.end(), cur_intr, cur_intr + 4); g.pe_huber.push_back(cur_huber);
    } else if (unknown[tag]++ == 0) {
      warn("unknown type: " + tag);
    }
  }
  return true;
}

}  // namespace g2o_text
