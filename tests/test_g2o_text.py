"""g2o text IO of the graph driver (examples/g2o_text.h; OptimizableGraph::load / save, core/optimizable_graph.h:594-606), no GPU needed:
`object_slam_main --g2o in out 0` parses a file and writes it again without touching the device."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    exe = os.path.join(ROOT, "build_tmp", "object_slam_main")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    return exe


def _rows(path):
    out = []
    for line in open(path):
        tok = line.split()
        if tok and not tok[0].startswith("#"):
            out.append((tok[0], [float(t) for t in tok[1:]]))
    return out


def test_load_follows_g2os_rules_and_save_uses_the_classes_field_order(tmp_path):
    """Comments and unknown tags are skipped (one warning per tag), FIX marks a vertex that already exists, an edge on a missing vertex is
    dropped with a warning, vertices come out sorted by id with their FIX lines behind them, edges in input order; the information
    matrices' upper triangles survive; poses are written camera-to-world (the inverse of the estimate, VertexSE3Expmap::write) and
    cuboids as toMinimalVector."""
    src = tmp_path / "in.g2o"
    q = np.array([0.1, -0.2, 0.3, 0.9])
    q /= np.linalg.norm(q)
    tri6 = " ".join(str(1.0 + 0.01 * k) for k in range(21))
    tri9 = " ".join(str(2.0 + 0.001 * k) for k in range(45))
    src.write_text(
        "# a comment\n"
        "VERTEX_SE3:EXPMAP 2 1.5 -0.25 0.75 %r %r %r %r\n" % tuple(float(v) for v in q) +
        "VERTEX_SE3:EXPMAP 1 0 0 0 0 0 0 1\n"
        "FIX 1\n"
        "VERTEX_CUBOID 0 1 2 0.5 0.01 -0.02 0.7 0.4 0.3 0.2\n"
        "VERTEX_TRACKXYZ 7 1 2 3\n"
        "VERTEX_TRACKXYZ 8 1 2 3\n"
        "EDGE_SE3:EXPMAP 1 2 0.1 0.2 0.3 0 0 0 1 " + tri6 + "\n"
        "EDGE_SE3:EXPMAP 1 9 0.1 0.2 0.3 0 0 0 1 " + tri6 + "\n"
        "EDGE_SE3_CUBOID 2 0 0.9 1.9 0.4 0 0 0.65 0.4 0.3 0.2 " + tri9 + "\n"
        "FIX 12\n")
    out = subprocess.run([_exe(), "--g2o", str(src), str(tmp_path / "out.g2o"), "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "loaded 2 cameras, 1 cuboids, 1 camera-cuboid edges, 1 odometry edges" in out.stdout
    assert out.stderr.count("unknown type: VERTEX_TRACKXYZ") == 1 and "Unable to fix vertex with id 12" in out.stderr and "edge EDGE_SE3:EXPMAP 1 9" in out.stderr
    rows = _rows(tmp_path / "out.g2o")
    assert [r[0] for r in rows] == ["VERTEX_CUBOID", "VERTEX_SE3:EXPMAP", "FIX", "VERTEX_SE3:EXPMAP", "EDGE_SE3_CUBOID", "EDGE_SE3:EXPMAP"]
    assert rows[0][1][0] == 0 and np.allclose(rows[0][1][1:], [1, 2, 0.5, 0.01, -0.02, 0.7, 0.4, 0.3, 0.2], atol=1e-14)
    assert rows[1][1][0] == 1 and rows[2][1] == [1.0] and rows[3][1][0] == 2
    assert np.allclose(rows[3][1][1:], [1.5, -0.25, 0.75, *q], atol=1e-14)
    assert np.allclose(rows[4][1][2:11], [0.9, 1.9, 0.4, 0, 0, 0.65, 0.4, 0.3, 0.2], atol=1e-14) and np.allclose(rows[4][1][11:], [2.0 + 0.001 * k for k in range(45)])
    assert np.allclose(rows[5][1][2:9], [0.1, 0.2, 0.3, 0, 0, 0, 1], atol=1e-14) and np.allclose(rows[5][1][9:], [1.0 + 0.01 * k for k in range(21)])
    # a second pass changes nothing
    out = subprocess.run([_exe(), "--g2o", str(tmp_path / "out.g2o"), str(tmp_path / "out2.g2o"), "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for a, b in zip(rows, _rows(tmp_path / "out2.g2o")):
        assert a[0] == b[0] and np.allclose(a[1], b[1], atol=1e-14)


def test_points_and_projection_edges_round_trip(tmp_path):
    """VERTEX_XYZ (VertexSBAPointXYZ::write, types/types_sba.cpp:47-55) and EDGE_SE3_PROJECT_XYZ:EXPMAP (EdgeSE3ProjectXYZ::write,
    types/types_six_dof_expmap.cpp:134-146: u v, then I(0,0) I(0,1) I(1,1); vertex 0 the point, vertex 1 the camera).  The class never writes
    fx fy cx cy or its robust kernel: the CS_INTRINSICS / CS_ROBUST_HUBER state lines carry them, an edge in front of any CS_INTRINSICS line
    is dropped with a warning, a FIX on a point sticks, and a second pass through the parser changes nothing."""
    src = tmp_path / "in.g2o"
    src.write_text(
        "VERTEX_SE3:EXPMAP 1 0 0 0 0 0 0 1\n"
        "FIX 1\n"
        "VERTEX_SE3:EXPMAP 2 0.8 0 0 0 0 0 1\n"
        "VERTEX_XYZ 10 0.5 -0.25 6.5\n"
        "VERTEX_XYZ 11 -1.5 0.75 9.25\n"
        "FIX 11\n"
        "EDGE_SE3_PROJECT_XYZ:EXPMAP 10 1 600.5 200.25  1 0 1\n"
        "CS_INTRINSICS 718.856 718.856 607.19 185.22\n"
        "CS_ROBUST_HUBER 2.4477\n"
        "EDGE_SE3_PROJECT_XYZ:EXPMAP 10 1 662.5 157.5  1 0 1\n"
        "EDGE_SE3_PROJECT_XYZ:EXPMAP 10 2 574 157.75  0.25 0.01 0.5\n"
        "CS_ROBUST_HUBER 0\n"
        "CS_INTRINSICS 700 710 600 180\n"
        "EDGE_SE3_PROJECT_XYZ:EXPMAP 11 2 420.5 243  1 0 1\n"
        "EDGE_SE3_PROJECT_XYZ:EXPMAP 12 2 1 2  1 0 1\n")
    out = subprocess.run([_exe(), "--g2o", str(src), str(tmp_path / "out.g2o"), "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2 points, 3 camera-point edges" in out.stdout
    assert "put a CS_INTRINSICS line before the edges" in out.stderr and "EDGE_SE3_PROJECT_XYZ:EXPMAP 12 2" in out.stderr
    rows = _rows(tmp_path / "out.g2o")
    assert [r[0] for r in rows] == ["VERTEX_SE3:EXPMAP", "FIX", "VERTEX_SE3:EXPMAP", "VERTEX_XYZ", "VERTEX_XYZ", "FIX",
                                    "CS_INTRINSICS", "CS_ROBUST_HUBER", "EDGE_SE3_PROJECT_XYZ:EXPMAP", "EDGE_SE3_PROJECT_XYZ:EXPMAP",
                                    "CS_INTRINSICS", "CS_ROBUST_HUBER", "EDGE_SE3_PROJECT_XYZ:EXPMAP"]
    assert rows[3][1] == [10, 0.5, -0.25, 6.5] and rows[4][1] == [11, -1.5, 0.75, 9.25] and rows[5][1] == [11.0]
    assert rows[6][1] == [718.856, 718.856, 607.19, 185.22] and rows[7][1] == [2.4477]
    assert rows[8][1] == [10, 1, 662.5, 157.5, 1, 0, 1] and rows[9][1] == [10, 2, 574, 157.75, 0.25, 0.01, 0.5]
    assert rows[10][1] == [700, 710, 600, 180] and rows[11][1] == [0.0] and rows[12][1] == [11, 2, 420.5, 243, 1, 0, 1]
    out = subprocess.run([_exe(), "--g2o", str(tmp_path / "out.g2o"), str(tmp_path / "out2.g2o"), "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stderr.strip() == ""
    assert _rows(tmp_path / "out.g2o") == _rows(tmp_path / "out2.g2o")
