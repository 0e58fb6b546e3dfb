"""GPU parity of the bundle-adjustment path (through the C ABI) against the CPU oracle.

Floating point: FP64 on both sides, but sums are associated differently (per-camera shuffle reductions vs
g2o's edge-order accumulation) and sin/cos/acos/tan come from ocml vs glibc, so the bar is the one
BASELINE.json states: states within 1e-5 relative after the same number of LM iterations, with the same
accept/reject sequence.  Intermediate quantities are held to tighter bounds where they are analytic
(1e-9 relative to the matrix scale) and to 1e-5 where they go through 1e-9-step numeric Jacobians.
"""
import os

import numpy as np
import pytest

from cube_slam_wu_amd import capi, synth_ba
from oracle import ba_oracle_py as O

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "golden", "object_slam_data")


def _oracle(pr, cuboids_first=False):
    P = O.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"], cuboids_first=cuboids_first)
    if len(pr["e_pt"]):
        P.set_edges_proj(pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
    if len(pr["ce_cam"]):
        P.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
    if len(pr.get("pe_cam", [])):
        P.set_edges_cuboid_proj(pr["pe_cam"], pr["pe_cub"], pr["pe_meas"], pr["pe_info"], pr["pe_K"])
    if len(pr["oe_i"]):
        P.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
    for cls, (kind, delta) in pr.get("robust", {}).items():
        P.set_robust_kernels(cls, kind, delta)
    return P


def _with_kernels(pr, seed, kinds=(0, 1, 2, 3, 4, 5, 6)):
    """Every edge class gets a random mix of the given kernel kinds, with widths around the edges' typical chi (so that inliers and
    outliers of every kernel occur)."""
    rng = np.random.default_rng(seed)
    pr = dict(pr)
    rob = {}
    for cls, n, typ in ((0, len(pr["e_pt"]), 2.5), (1, len(pr["ce_cam"]), 3.0), (2, len(pr.get("pe_cam", [])), 8.0), (3, len(pr["oe_i"]), 0.3)):
        if n == 0:
            continue
        kind = rng.choice(np.asarray(kinds), n)
        delta = typ * rng.uniform(0.3, 2.0, n)
        rob[cls] = (kind, delta)
    pr["robust"] = rob
    return pr


@pytest.mark.parametrize("bbox", [False, True])
def test_robust_kernels_on_every_edge_class(bbox):
    """g2o's robust kernels (robust_kernel_impl.cpp:78-165: Huber, PseudoHuber, Cauchy, Saturated, DCS, Tukey) on projection,
    EdgeSE3Cuboid, EdgeSE3CuboidProj and EdgeSE3Expmap edges: chi2, linear system, one damped solve and a 6-iteration LM run
    against the oracle (base_binary_edge.hpp:88-111 weights Omega and -Omega e with rho')."""
    pr = _with_kernels(synth_ba.make_problem(n_cams=40, n_points=2000, n_cuboids=6, seed=17, bbox_edges=bbox), 5)
    assert set(pr["robust"]) == ({0, 1, 2, 3} if bbox else {0, 1, 3})
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    chi_r = R.compute_errors()[0]
    assert abs(G.compute_errors() - chi_r) < 1e-9 * chi_r
    # the kernels must matter: without them chi2 differs by far more than the tolerance
    pr0 = dict(pr); pr0["robust"] = {}
    G0 = capi.ba_from_dict(pr0)
    assert abs(G0.compute_errors() - chi_r) > 1e-3 * chi_r
    Hpp_g, Hll_g, Hpl_g, b_g = G.build_system()
    Hpp_r, Hll_r, Hpl_r, b_r = R.build_system()
    assert _rel(Hpp_g, Hpp_r) < 1e-5 and _rel(b_g, b_r) < 1e-5 and _rel(Hll_g, Hll_r) < 1e-11 and _rel(Hpl_g, Hpl_r) < 1e-11
    ok_g, x_g = G.solve(1.0)
    ok_r, x_r = R.solve(1.0)
    assert ok_g and ok_r and _rel(x_g, x_r) < 1e-5
    n_g, n_r = G.optimize(6), R.optimize(6)
    assert n_g == n_r
    assert np.array_equal(G.history()[2], R.history()[2]) and np.allclose(G.history()[0], R.history()[0], rtol=1e-5)
    scale = np.abs(R.state()[2]).max()
    for a, b in zip(G.state(), R.state()):
        assert np.abs(a - b).max() < 1e-5 * scale
    G.close(); G0.close(); R.close()


def test_robust_kernels_huber_width_is_single_precision():
    """RobustKernelHuber::dsqr is a float member of the vendored g2o (robust_kernel_impl.h:86): rho(e) = 2 sqrt(e) delta - dsqr of every
    outlier carries the single-precision square.  The device's and the oracle's robust chi2 follow the float formula to rounding; the
    double-precision square would be off by ~1e-7 per outlier -- far more than the agreement asked for."""
    pr = synth_ba.make_problem(n_cams=12, n_points=300, n_cuboids=0, seed=2)
    pr = dict(pr); pr["oe_i"] = pr["oe_i"][:0]; pr["oe_j"] = pr["oe_j"][:0]; pr["oe_meas"] = pr["oe_meas"][:0]; pr["oe_info"] = pr["oe_info"][:0]
    pr["e_huber"] = np.full(len(pr["e_pt"]), 0.9)       # (a width most edges exceed; 0.9^2 = 0.81 is not a float)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    chi_r, ep, _, _ = R.compute_errors()
    info = np.asarray(pr["e_info"]).reshape(-1, 2, 2)
    e = np.einsum("ei,eij,ej->e", ep, info, ep)
    d = pr["e_huber"]
    f32 = (d * d).astype(np.float32).astype(np.float64)
    want_float = np.where(e <= f32, e, 2 * np.sqrt(e) * d - f32).sum()
    want_double = np.where(e <= d * d, e, 2 * np.sqrt(e) * d - d * d).sum()
    assert (e > f32).sum() > 500 and abs(want_float - want_double) > 1e-11 * want_float
    assert abs(chi_r - want_float) < 1e-13 * want_float
    assert abs(G.compute_errors() - want_float) < 1e-13 * want_float
    G.close(); R.close()

def _odom_terms_on_the_host(pr, cams, cuboids, points):
    """The odometry edges' quadratic-form terms at the given estimates, evaluated on the host (the restated numeric linearizeOplus +
    constructQuadraticForm, base_binary_edge.hpp:54-205): what an application would obtain from the edges' own virtuals."""
    nc, no = len(pr["cams"]), len(pr["cuboids"])
    R = O.Problem(cams, pr["cam_fixed"], cuboids, pr["cub_fixed"], points[:0], np.zeros(0, np.int32))
    R.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
    chi = R.compute_errors()[0]
    Hpp, _, _, b = R.build_system()
    col = np.full(nc, -1); c = 0
    for i in range(nc):
        if not pr["cam_fixed"][i]:
            col[i] = c; c += 6
    cam36, cam6 = np.zeros((nc, 36)), np.zeros((nc, 6))
    for i in range(nc):
        if col[i] >= 0:
            cam36[i] = Hpp[col[i]:col[i] + 6, col[i]:col[i] + 6].ravel(); cam6[i] = b[col[i]:col[i] + 6]
    Hij = np.zeros((len(pr["oe_i"]), 81))
    for k, (i, j) in enumerate(zip(pr["oe_i"], pr["oe_j"])):
        if col[i] >= 0 and col[j] >= 0:
            Hij[k, :36] = Hpp[col[i]:col[i] + 6, col[j]:col[j] + 6].ravel()
    R.close()
    return cam36, cam6, Hij, chi


def test_external_edges_equal_the_same_edges_evaluated_on_the_device():
    """The CPU path for edge types the library does not evaluate (cs_ba_set_external_edges / _terms / _callback): the odometry edges
    of a graph are taken away from the device and handed in as external pose-pose edges whose terms the host evaluates -- chi2, the
    linear system, a damped solve and a whole LM run must equal the all-device graph's.  Plus unary terms on a point and a cuboid."""
    pr = synth_ba.make_problem(n_cams=40, n_points=2000, n_cuboids=6, seed=9)
    G = capi.ba_from_dict(pr)
    pr_x = dict(pr); pr_x["oe_i"] = pr["oe_i"][:0]; pr_x["oe_j"] = pr["oe_j"][:0]; pr_x["oe_meas"] = pr["oe_meas"][:0]; pr_x["oe_info"] = pr["oe_info"][:0]
    X = capi.ba_from_dict(pr_x)
    n_e = len(pr["oe_i"])
    X.set_external_edges(np.zeros(n_e), pr["oe_i"], np.zeros(n_e), pr["oe_j"])

    def refresh(want_system):
        cams, cubs, pts = X.state()
        cam36, cam6, Hij, chi = _odom_terms_on_the_host(pr, cams, cubs, pts)
        if want_system:
            X.set_external_terms(cam36=cam36, cam6=cam6, Hij81=Hij, chi2=chi)
        else:
            X.set_external_chi2(chi)
    refresh(1)
    chi_g = G.compute_errors()
    assert abs(X.compute_errors() - chi_g) < 1e-12 * chi_g
    sys_g, sys_x = G.build_system(), X.build_system()
    for a, b in zip(sys_g, sys_x):        # (the odometry blocks come from numeric delta = 1e-9 Jacobians evaluated by two different programs)
        assert _rel(b, a) < 1e-9
    assert G.solver_layout() == X.solver_layout()       # the external edges took part in the ordering: same band
    (ok_g, x_g), (ok_x, x_x) = G.solve(2.0), X.solve(2.0)
    assert ok_g and ok_x and _rel(x_x, x_g) < 1e-6
    # without a callback the library's own LM loop cannot re-evaluate them
    with pytest.raises(RuntimeError, match="callback"):
        X.optimize(2)
    X.set_external_callback(refresh)
    n_g, n_x = G.optimize(6), X.optimize(6)
    assert n_g == n_x and np.array_equal(G.history()[2], X.history()[2]) and np.allclose(G.history()[0], X.history()[0], rtol=1e-6)
    scale = np.abs(G.state()[2]).max()
    for a, b in zip(G.state(), X.state()):
        assert np.abs(a - b).max() < 1e-5 * scale
    # unary terms: a prior w |x - x0|^2 on every point and a 9 x 9 block on every cuboid land in A_ii / b_i of the free vertices only
    Y = capi.ba_from_dict(pr)
    base = Y.build_system()
    hc0, ho0, hp0 = [h.copy() for h in Y.vertex_hessians()]
    w = 3.0
    npt, no = len(pr["points"]), len(pr["cuboids"])
    pt9 = np.tile((w * np.eye(3)).ravel(), (npt, 1)); pt3 = np.full((npt, 3), -0.25)
    M = np.arange(81, dtype=float).reshape(9, 9); M = M + M.T
    cub81 = np.tile(M.ravel(), (no, 1)); cub9 = np.full((no, 9), 0.5)
    Y.set_external_terms(cub81=cub81, cub9=cub9, pt9=pt9, pt3=pt3, chi2=7.0)
    assert abs(Y.compute_errors() - (chi_g + 7.0)) < 1e-12 * chi_g
    ext = Y.build_system()
    hc1, ho1, hp1 = Y.vertex_hessians()
    free_p = np.asarray(pr["pt_fixed"]) == 0
    assert np.allclose(hp1[free_p] - hp0[free_p], w * np.eye(3), atol=1e-9) and np.array_equal(hp1[~free_p], hp0[~free_p])
    free_o = np.asarray(pr["cub_fixed"]) == 0
    assert np.allclose(ho1[free_o] - ho0[free_o], M, rtol=0, atol=1e-6 * np.abs(ho0).max()) and np.array_equal(hc1, hc0)
    assert np.allclose((ext[3] - base[3])[-3 * free_p.sum():], -0.25, atol=1e-9)
    G.close(); X.close(); Y.close()


def test_external_edges_on_one_vertex_pair_sum_in_a_fixed_order():
    """Several host-evaluated edges between the same two vertices -- one of them stored the other way round -- go into the reduced system
    one after the other inside one workgroup (ba_ext_offdiag_kernel: no atomics).  With dyadic block values the sum is exact in any
    order, so the damped solve must equal, bit for bit, a handle that carries the summed block as ONE edge; and repeat itself."""
    pr = synth_ba.make_problem(n_cams=30, n_points=1200, n_cuboids=3, seed=4)
    rng = np.random.default_rng(0)
    blk = rng.integers(-64, 64, (3, 6, 6)) / 16.0
    pairs = [(3, 7), (3, 7), (7, 3), (10, 11)]
    H = np.zeros((4, 81))
    H[0, :36] = blk[0].ravel(); H[1, :36] = blk[1].ravel(); H[2, :36] = blk[2].T.ravel(); H[3, :36] = blk[0].ravel()
    one = np.zeros((2, 81)); one[0, :36] = (blk[0] + blk[1] + blk[2]).ravel(); one[1, :36] = blk[0].ravel()
    nc = len(pr["cams"])
    xs = []
    for edges, Hij in ((pairs, H), ([(3, 7), (10, 11)], one), (pairs, H)):
        X = capi.ba_from_dict(pr)
        X.set_external_edges(np.zeros(len(edges)), [a for a, _ in edges], np.zeros(len(edges)), [b for _, b in edges])
        X.set_external_terms(cam36=np.zeros((nc, 36)), cam6=np.zeros((nc, 6)), Hij81=Hij, chi2=0.0)
        X.compute_errors(); X.build_system()
        ok, x = X.solve(1.5)
        assert ok
        xs.append(x)
        X.close()
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(xs[0], xs[2])
    G = capi.ba_from_dict(pr)
    G.compute_errors(); G.build_system()
    assert not np.array_equal(G.solve(1.5)[1], xs[0])         # (the blocks matter)
    G.close()


def test_external_edges_are_validated():
    pr = synth_ba.make_problem(n_cams=12, n_points=300, n_cuboids=2, seed=2)
    G = capi.ba_from_dict(pr)
    G.set_external_edges([0], [1], [2], [5])                 # camera - point: would change the Schur structure
    with pytest.raises(RuntimeError, match="cameras and cuboids only"):
        G.sizes()
    G.set_external_edges([0], [1], [0], [99])
    with pytest.raises(RuntimeError, match="out of range"):
        G.sizes()
    G.set_external_edges([0], [1], [1], [0])                 # camera - cuboid: the cuboids stay in the reduced system
    G.sizes()
    assert G.reduced_size()[1] is False
    G.set_shard(0, 2)
    with pytest.raises(RuntimeError, match="sharded"):
        G.sizes()
    G.close()


def test_check_finite_names_where_a_nan_was_born():
    """cs_ba_check_finite (the stand-in for the NaN checks of g2o's debug builds, sparse_optimizer.cpp:78-86 / block_solver.hpp:533-544):
    clean on a healthy problem; a poisoned point estimate is reported through the estimate itself, the squared error of an edge that sees
    it, and -- after the linearisation -- the blocks it contaminates, with the caller's indices."""
    pr = synth_ba.make_problem(n_cams=20, n_points=600, n_cuboids=3, seed=4)
    G = capi.ba_from_dict(pr)
    G.build_system()
    G.solve(1.0)
    n, rep = G.check_finite()
    assert n == 0 and rep == ""
    cams, cubs, pts = G.state()
    victim = int(np.asarray(pr["e_pt"])[137])
    pts[victim, 1] = np.nan
    G.set_estimates(points=pts)
    n, rep = G.check_finite()
    assert n > 0 and "point estimates" in rep and "first in point %d" % victim in rep
    line = [l for l in rep.splitlines() if l.startswith("squared error of a projection edge")][0]
    edge = int(line.rsplit(" ", 1)[1])
    assert np.asarray(pr["e_pt"])[edge] == victim
    G.build_system()
    n2, rep2 = G.check_finite()
    assert n2 > n and "A_jj of a point" in rep2 and "A_ii of a camera" in rep2 and "H_pl block" in rep2
    G.close()


def test_dump_and_load_round_trip(tmp_path):
    """cs_ba_dump / cs_ba_load: a problem in mid-optimisation (estimates that only exist on the device, all four edge classes, robust
    kernels) written to one flat file and read back into a fresh handle continues bit-identically."""
    pr = _with_kernels(synth_ba.make_problem(n_cams=30, n_points=1200, n_cuboids=5, seed=8, bbox_edges=True), 3, kinds=(0, 1, 3, 5))
    G = capi.ba_from_dict(pr)
    G.optimize(2)
    path = tmp_path / "problem.csba"
    G.dump(path)
    L = capi.BaProblem.load(path, (len(pr["cams"]), len(pr["cuboids"]), len(pr["points"]), len(pr["e_pt"])))
    for a, b in zip(G.state(), L.state()):
        assert np.array_equal(a, b)
    assert G.optimize(3) == L.optimize(3)
    for a, b in zip(G.history(), L.history()):
        assert np.array_equal(a, b)
    for a, b in zip(G.state(), L.state()):
        assert np.array_equal(a, b)
    # removing a class's kernels needs no edge count from the caller: a loaded handle (no setter ever ran on the Python object) can do it too
    c_with = L.compute_errors()
    for cls in (capi.EDGE_PROJ, capi.EDGE_CUBOID, capi.EDGE_CUBOID_PROJ, capi.EDGE_ODOM):
        L.set_robust_kernels(cls, None, None); G.set_robust_kernels(cls, None, None)
    assert L.compute_errors() == G.compute_errors() != c_with
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 100)
    with pytest.raises(RuntimeError, match="truncated"):
        capi.BaProblem.load(path, (1, 1, 1, 1))
    G.close(); L.close()


def test_robust_kernel_arguments_are_checked():
    pr = synth_ba.make_problem(n_cams=12, n_points=300, n_cuboids=2, seed=2)
    G = capi.ba_from_dict(pr)
    with pytest.raises(RuntimeError, match="number of edges"):
        G.set_robust_kernels(capi.EDGE_ODOM, [1, 1], [1.0, 1.0])
    with pytest.raises(RuntimeError, match="unknown kernel kind"):
        G.set_robust_kernels(capi.EDGE_ODOM, np.full(len(pr["oe_i"]), 9), np.ones(len(pr["oe_i"])))
    with pytest.raises(RuntimeError, match="delta > 0"):
        G.set_robust_kernels(capi.EDGE_ODOM, np.full(len(pr["oe_i"]), 3), np.zeros(len(pr["oe_i"])))
    G.set_robust_kernels(capi.EDGE_ODOM, np.full(len(pr["oe_i"]), 3), np.ones(len(pr["oe_i"])))
    c1 = G.compute_errors()
    G.set_robust_kernels(capi.EDGE_ODOM, None, None)
    assert G.compute_errors() != c1
    G.close()


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def test_system_parity_projection_only():
    pr = synth_ba.make_problem(n_cams=24, n_points=1200, n_cuboids=0, seed=7)
    pr["oe_i"] = pr["oe_i"][:0]; pr["oe_j"] = pr["oe_j"][:0]; pr["oe_meas"] = pr["oe_meas"][:0]; pr["oe_info"] = pr["oe_info"][:0]
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    chi_r = R.compute_errors()[0]
    assert abs(G.compute_errors() - chi_r) <= 1e-11 * chi_r
    Hpp_g, Hll_g, Hpl_g, b_g = G.build_system()
    Hpp_r, Hll_r, Hpl_r, b_r = R.build_system()
    assert _rel(Hpp_g, Hpp_r) < 1e-11 and _rel(Hll_g, Hll_r) < 1e-11 and _rel(Hpl_g, Hpl_r) < 1e-11 and _rel(b_g, b_r) < 1e-11
    ok_g, x_g = G.solve(10.0)
    ok_r, x_r = R.solve(10.0)
    assert ok_g and ok_r
    assert _rel(x_g, x_r) < 1e-8


def test_system_parity_all_edge_types():
    pr = synth_ba.make_problem(n_cams=30, n_points=1500, n_cuboids=6, seed=11)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    chi_r = R.compute_errors()[0]
    assert abs(G.compute_errors() - chi_r) <= 1e-10 * chi_r
    Hpp_g, Hll_g, Hpl_g, b_g = G.build_system()
    Hpp_r, Hll_r, Hpl_r, b_r = R.build_system()
    assert _rel(Hll_g, Hll_r) < 1e-11 and _rel(Hpl_g, Hpl_r) < 1e-11
    assert _rel(Hpp_g, Hpp_r) < 1e-5 and _rel(b_g, b_r) < 1e-5      # numeric (delta = 1e-9) Jacobians inside
    ok_g, x_g = G.solve(50.0)
    ok_r, x_r = R.solve(50.0)
    assert ok_g and ok_r
    assert _rel(x_g, x_r) < 1e-5


def test_pose_marginals_are_blocks_of_the_inverse_pose_hessian():
    """cs_ba_pose_marginals = Solver::computeMarginals (core/block_solver.hpp:488-499: LinearSolver::solvePattern on _Hpp, i.e. blocks of
    inv(H_pp) as buildSystem left it -- no lambda, no Schur complement): camera-camera, camera-cuboid, cuboid-cuboid and off-diagonal blocks
    against numpy's inverse of the device's own dense H_pp (1e-9) and of the ORACLE's H_pp (1e-5: numeric Jacobians inside); a fixed vertex
    is refused; before cs_ba_build_system the call says so."""
    pr = synth_ba.make_problem(n_cams=30, n_points=1500, n_cuboids=6, seed=11)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    with pytest.raises(RuntimeError):
        G.pose_marginals([((0, 1), (0, 1))])
    G.compute_errors(); R.compute_errors()
    Hpp_g = G.build_system()[0]
    Hpp_r = R.build_system()[0]
    cam_fixed, cub_fixed = np.asarray(pr["cam_fixed"]), np.asarray(pr["cub_fixed"])
    assert cam_fixed[0] and not cam_fixed[1:].any() and not cub_fixed.any()
    nfree = int((cam_fixed == 0).sum())
    cam_col = {i: 6 * (int((cam_fixed[:i] == 0).sum())) for i in range(len(cam_fixed)) if not cam_fixed[i]}
    cub_col = {j: 6 * nfree + 9 * j for j in range(len(cub_fixed))}
    pairs = [((0, 1), (0, 1)), ((0, 5), (0, 9)), ((0, 9), (0, 5)), ((0, 3), (1, 2)), ((1, 2), (0, 3)), ((1, 4), (1, 4)), ((1, 0), (1, 5)), ((0, 29), (0, 29))]
    blocks, pd = G.pose_marginals(pairs)
    assert pd
    inv_g, inv_r = np.linalg.inv(Hpp_g), np.linalg.inv(Hpp_r)
    for ((ca, ia), (cb, ib)), blk in zip(pairs, blocks):
        r0 = cam_col[ia] if ca == 0 else cub_col[ia]; c0 = cam_col[ib] if cb == 0 else cub_col[ib]
        da, db = (6, 9)[ca], (6, 9)[cb]
        assert blk.shape == (da, db)
        assert _rel(blk, inv_g[r0:r0 + da, c0:c0 + db]) < 1e-9, (ia, ib)
        assert _rel(blk, inv_r[r0:r0 + da, c0:c0 + db]) < 1e-4, (ia, ib)
    with pytest.raises(RuntimeError):
        G.pose_marginals([((0, 0), (0, 1))])       # camera 0 is fixed
    with pytest.raises(RuntimeError):
        G.pose_marginals([((2, 0), (0, 1))])       # a point
    G.close()


@pytest.mark.parametrize("huber", [True, False])
def test_optimize_parity_10_iterations(huber):
    pr = synth_ba.make_problem(n_cams=40, n_points=2500, n_cuboids=8, seed=3, huber=huber)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    n_g, n_r = G.optimize(10), R.optimize(10)
    assert n_g == n_r
    chi_g, lam_g, tr_g = G.history()
    chi_r, lam_r, tr_r = R.history()
    assert np.array_equal(tr_g, tr_r)                       # same accept / reject sequence
    assert np.allclose(chi_g, chi_r, rtol=1e-6)
    cg, og, pg = G.state()
    cr, orr, prr = R.state()
    # 1e-5 relative (BASELINE.json); translations relative to the scene scale
    scale = np.abs(prr).max()
    assert np.abs(pg - prr).max() < 1e-5 * scale
    assert np.abs(cg[:, :3] - cr[:, :3]).max() < 1e-5 * scale and np.abs(cg[:, 3:] - cr[:, 3:]).max() < 1e-5
    assert np.abs(og[:, :3] - orr[:, :3]).max() < 1e-5 * scale and np.abs(og[:, 3:] - orr[:, 3:]).max() < 1e-5


def test_fixed_points_and_cuboids_first_ordering():
    pr = synth_ba.make_problem(n_cams=16, n_points=500, n_cuboids=3, seed=5)
    pr["pt_fixed"] = pr["pt_fixed"].copy(); pr["pt_fixed"][::7] = 1
    G, R = capi.ba_from_dict(pr, cuboids_first=True), _oracle(pr, cuboids_first=True)
    Hpp_g, Hll_g, Hpl_g, b_g = G.build_system()
    Hpp_r, Hll_r, Hpl_r, b_r = R.build_system()
    assert Hpp_g.shape == Hpp_r.shape and Hll_g.shape == Hll_r.shape
    assert _rel(Hpp_g, Hpp_r) < 1e-5 and _rel(Hll_g, Hll_r) < 1e-11 and _rel(Hpl_g, Hpl_r) < 1e-11 and _rel(b_g, b_r) < 1e-5
    assert G.optimize(4) == R.optimize(4)
    assert np.abs(G.state()[2] - R.state()[2]).max() < 1e-5 * np.abs(R.state()[2]).max()


def test_reference_offline_sequence_on_gpu():
    """The reference's bundled 58-frame graph (main_obj.cpp:479-841, offline mode) through the HIP solver."""
    def mk(cams, cam_fixed, cuboid, cub_edges, odom_edges):
        P = capi.BaProblem(cams, cam_fixed, cuboids=cuboid[None, :], cub_fixed=[0], cuboids_first=True)
        if cub_edges:
            P.set_edges_cuboid([e[0] for e in cub_edges], [0] * len(cub_edges), np.array([e[1] for e in cub_edges]), np.array([e[2] for e in cub_edges]))
        if odom_edges:
            P.set_edges_odom([e[0] for e in odom_edges], [e[1] for e in odom_edges], np.array([e[2] for e in odom_edges]), np.tile(np.eye(6).ravel(), (len(odom_edges), 1)))
        return P
    cam_g, obj_g, it_g, fin_g = O.run_offline_sequence(DATA, make_problem=mk)
    cam_r, obj_r, it_r, fin_r = O.run_offline_sequence(DATA)
    # Every frame re-optimises a graph that is already at its minimum, so g2o's stop rules (rho == 0, three
    # iterations without 1e-3 relative progress) fire on rounding noise; the iteration count may differ on the
    # odd frame (measured: 2 of 58) while the states still agree to < 1e-6.
    assert (it_g == it_r).mean() >= 0.9
    assert np.abs(obj_g - obj_r).max() < 1e-5 * max(1.0, np.abs(obj_r).max())
    assert np.abs(fin_g - fin_r).max() < 1e-5 * max(1.0, np.abs(fin_r).max())


def test_graph_driver_reads_and_writes_the_reference_formats(tmp_path):
    """examples/object_slam_main.cpp = main_obj.cpp:479-841 (offline mode) + the result files of :305-336, in C++ on the C
    ABI: reads the reference's three text tables, writes output_cam_poses.txt / output_obj_poses.txt.  Checked against
    the restated driver at 12 digits and, at the reference's 6-digit format, against the envelope of the reference's own
    saved (online-mode) outputs (object within 0.1 m, cameras within 0.6 m)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "object_slam_main")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run([exe, DATA, str(tmp_path), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    cams = np.loadtxt(tmp_path / "output_cam_poses.txt")
    objs = np.loadtxt(tmp_path / "output_obj_poses.txt")
    cam_r, obj_r, it_r, fin_r = O.run_offline_sequence(DATA)
    assert cams.shape == (58, 8) and objs.shape == (58, 9)
    assert np.abs(objs - obj_r).max() < 1e-5 * max(1.0, np.abs(obj_r).max())
    assert np.abs(cams[:, 1:] - fin_r).max() < 1e-5 * max(1.0, np.abs(fin_r).max())
    truth = np.loadtxt(os.path.join(DATA, "truth_cam_poses.txt"))
    assert np.abs(cams[:, 0] - truth[:, 0]).max() < 1e-6
    # the reference's own format (6 significant digits) and its saved outputs as the sanity envelope
    out = subprocess.run([exe, DATA, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    with open(tmp_path / "output_cam_poses.txt") as f:
        assert f.readline() == "# timestamp tx ty tz qx qy qz qw\n"
    saved_obj = np.loadtxt(os.path.join(DATA, "output_obj_poses.txt"))
    saved_cam = np.loadtxt(os.path.join(DATA, "output_cam_poses.txt"))
    objs6 = np.loadtxt(tmp_path / "output_obj_poses.txt")
    cams6 = np.loadtxt(tmp_path / "output_cam_poses.txt")
    assert np.abs(objs6[:, :3] - saved_obj[:, :3]).max() < 0.1
    assert np.abs(cams6[:, 1:4] - saved_cam[:, 1:4]).max() < 0.6


def test_online_run_on_the_device_reproduces_the_references_saved_outputs():
    """main_obj.cpp's ONLINE mode over the reference's 58 bundled TUM frames with both paths on the device: every frame's
    cuboid from cs_detect_cuboids_gray (image in; roll/pitch sampling from the second frame on), the growing graph optimised by
    cs_ba_optimize.  Checked against the two result files the reference saved from its own run (object position to
    millimetres in every frame) and against the same pipeline on the two oracles."""
    pytest.importorskip("PIL")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_reference_frames as TR
    import tum_frames
    from oracle import edge_oracle_py as E
    dets = [capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=s, nominal_skew_ratio=2.0)) for s in (0, 1)]

    def detect(fr, gray, sample):
        got = dets[sample].detect_gray(fr, gray)
        return got[0][0] if got[0] else None

    def mk(cams, cam_fixed, cuboid, cub_edges, odom_edges):
        P = capi.BaProblem(cams, cam_fixed, cuboids=cuboid[None, :], cub_fixed=[0], cuboids_first=True)
        if cub_edges:
            P.set_edges_cuboid([e[0] for e in cub_edges], [0] * len(cub_edges), np.array([e[1] for e in cub_edges]), np.array([e[2] for e in cub_edges]))
        if odom_edges:
            P.set_edges_odom([e[0] for e in odom_edges], [e[1] for e in odom_edges], np.array([e[2] for e in odom_edges]), np.tile(np.eye(6).ravel(), (len(odom_edges), 1)))
        return P

    loader = lambda k: tum_frames.load_for_online_run(k, E.bgr_to_gray)
    obj_g, cam_g, n_g = O.run_online_sequence(tum_frames.DATA, loader, detect, make_problem=mk)
    assert n_g == 51
    TR.check_online_run_against_saved_outputs(obj_g, cam_g)
    obj_r, cam_r, _ = O.run_online_sequence(tum_frames.DATA, loader, TR._oracle_detect)
    assert np.abs(obj_g - obj_r).max() < 1e-5 * max(1.0, np.abs(obj_r).max())
    assert np.abs(cam_g - cam_r).max() < 1e-5 * max(1.0, np.abs(cam_r).max())
    for d in dets:
        d.close()


def _read_g2o(path):
    """g2o text -> {tag: list of float rows}; FIX ids in a set."""
    rows, fixed = {}, set()
    for line in open(path):
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        if tok[0] == "FIX":
            fixed.update(int(t) for t in tok[1:])
        else:
            rows.setdefault(tok[0], []).append([float(t) for t in tok[1:]])
    return {k: np.array(v) for k, v in rows.items()}, fixed


def test_graph_driver_g2o_text_round_trip_with_points_and_projection_edges(tmp_path):
    """Round 6: VERTEX_XYZ / EDGE_SE3_PROJECT_XYZ:EXPMAP lines (VertexSBAPointXYZ::write types/types_sba.cpp:47-55, EdgeSE3ProjectXYZ::write
    types/types_six_dof_expmap.cpp:134-146) with the CS_INTRINSICS / CS_ROBUST_HUBER state lines: a C3-shaped synthetic graph (200 cameras,
    4 000 points, 50 cuboids, Huber kernels) written as g2o text, `object_slam_main --g2o in out 5` loads it, runs five LM iterations on the
    device and saves; the saved estimates equal the same problem handed to the library through the Python binding (1e-5 relative, as the
    58-frame round trip below), and a reload of the saved file parses to the same numbers."""
    import subprocess
    from scipy.spatial.transform import Rotation
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "object_slam_main")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    pr = synth_ba.make_problem(n_cams=200, n_points=4000, n_cuboids=50, seed=5)
    nc, no, npt = len(pr["cams"]), len(pr["cuboids"]), len(pr["points"])
    cam_id = 1 + np.arange(nc); cub_id = 1000 + np.arange(no); pt_id = 5000 + np.arange(npt)      # cameras first, as the binding orders them
    def inv7(v):
        R = Rotation.from_quat(v[:, 3:7]); t = -R.inv().apply(v[:, :3]); q = R.inv().as_quat()
        q = np.where(q[:, 3:4] < 0, -q, q)
        return np.concatenate([t, q], axis=1)
    def minimal(c10):      # cuboid::toMinimalVector: x y z roll pitch yaw sx sy sz
        e = Rotation.from_quat(c10[:, 3:7]).as_euler("ZYX")
        return np.concatenate([c10[:, :3], e[:, [2, 1, 0]], c10[:, 7:10]], axis=1)
    f = lambda a: " ".join(repr(float(x)) for x in a)
    lines = []
    for i, v in enumerate(inv7(pr["cams"])):
        lines.append("VERTEX_SE3:EXPMAP %d %s" % (cam_id[i], f(v)))
        if pr["cam_fixed"][i]:
            lines.append("FIX %d" % cam_id[i])
    for i, v in enumerate(minimal(pr["cuboids"])):
        lines.append("VERTEX_CUBOID %d %s" % (cub_id[i], f(v)))
    for i, v in enumerate(pr["points"]):
        lines.append("VERTEX_XYZ %d %s" % (pt_id[i], f(v)))
    iu9, iu6 = np.triu_indices(9), np.triu_indices(6)
    for k in range(len(pr["ce_cam"])):
        lines.append("EDGE_SE3_CUBOID %d %d %s %s" % (cam_id[pr["ce_cam"][k]], cub_id[pr["ce_cub"][k]], f(minimal(pr["ce_meas"][k:k + 1])[0]), f(pr["ce_info"][k].reshape(9, 9)[iu9])))
    om = inv7(pr["oe_meas"])
    for k in range(len(pr["oe_i"])):
        lines.append("EDGE_SE3:EXPMAP %d %d %s %s" % (cam_id[pr["oe_i"][k]], cam_id[pr["oe_j"][k]], f(om[k]), f(pr["oe_info"][k].reshape(6, 6)[iu6])))
    lines.append("CS_INTRINSICS " + f(pr["e_intr"][0]))
    lines.append("CS_ROBUST_HUBER " + f(pr["e_huber"][:1]))
    assert (pr["e_intr"] == pr["e_intr"][0]).all() and (pr["e_huber"] == pr["e_huber"][0]).all()
    for k in range(len(pr["e_pt"])):
        I = pr["e_info"][k]
        lines.append("EDGE_SE3_PROJECT_XYZ:EXPMAP %d %d %s %s" % (pt_id[pr["e_pt"][k]], cam_id[pr["e_cam"][k]], f(pr["e_uv"][k]), f([I[0], I[1], I[3]])))
    (tmp_path / "in.g2o").write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, "--g2o", str(tmp_path / "in.g2o"), str(tmp_path / "opt.g2o"), "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stderr.strip() == "", out.stdout + out.stderr
    assert "%d points, %d camera-point edges" % (npt, len(pr["e_pt"])) in out.stdout and "LM iterations: 5" in out.stdout
    g, fixed = _read_g2o(tmp_path / "opt.g2o")
    assert g["VERTEX_XYZ"].shape == (npt, 4) and g["EDGE_SE3_PROJECT_XYZ:EXPMAP"].shape == (len(pr["e_pt"]), 7) and fixed == {int(cam_id[i]) for i in np.flatnonzero(pr["cam_fixed"])}
    # the same problem from the file's own numbers through the binding (the cuboid lines went through Euler angles, the poses through an inversion)
    g0, _ = _read_g2o(tmp_path / "in.g2o")
    def cub10(m):
        q = Rotation.from_euler("ZYX", m[:, [5, 4, 3]]).as_quat(); q = np.where(q[:, 3:4] < 0, -q, q)
        return np.concatenate([m[:, :3], q, m[:, 6:9]], axis=1)
    pr2 = dict(pr)
    pr2["cams"] = inv7(g0["VERTEX_SE3:EXPMAP"][:, 1:]); pr2["cuboids"] = cub10(g0["VERTEX_CUBOID"][:, 1:]); pr2["ce_meas"] = cub10(g0["EDGE_SE3_CUBOID"][:, 2:11])
    pr2["oe_meas"] = inv7(g0["EDGE_SE3:EXPMAP"][:, 2:9])
    G = capi.ba_from_dict(pr2)
    assert G.optimize(5) == 5
    cg, og, pg = G.state()
    G.close()
    # (1e-5: the cuboid / odometry edges' numeric delta = 1e-9 Jacobians amplify the last-digit differences of the two Euler <-> quaternion conversions)
    assert np.abs(g["VERTEX_XYZ"][:, 1:] - pg).max() < 1e-5 * np.abs(pg).max()
    assert np.abs(inv7(g["VERTEX_SE3:EXPMAP"][:, 1:]) - cg).max() < 1e-5 * np.abs(cg).max()
    assert np.abs(cub10(g["VERTEX_CUBOID"][:, 1:]) - og).max() < 1e-5 * np.abs(og).max()
    assert np.abs(pg - pr["points"]).max() > 1e-3                # (the run moved the estimates)
    # and a reload of the saved file parses to the same numbers
    out = subprocess.run([exe, "--g2o", str(tmp_path / "opt.g2o"), str(tmp_path / "again.g2o"), "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0
    g2, fixed2 = _read_g2o(tmp_path / "again.g2o")
    assert fixed2 == fixed and sorted(g2) == sorted(g)
    for k in g:
        assert g2[k].shape == g[k].shape and np.abs(g2[k] - g[k]).max() < 1e-11 * max(1.0, np.abs(g[k]).max()), k


def test_graph_driver_g2o_text_round_trip_on_the_58_frame_graph(tmp_path):
    """g2o's text format for the driver's graph (examples/g2o_text.h: OptimizableGraph::save / load, core/optimizable_graph.h:594-606, with the
    field order of VertexSE3Expmap / VertexCuboid / EdgeSE3Expmap::write).  The offline run leaves graph.g2o (58 cameras, the object, 57
    odometry edges, the camera-object edges); `--g2o in out 0` reads it back and writes it again: the same numbers; `--g2o in out 5`
    optimises the loaded graph on the device: the same five LM iterations as the graph built through the Python binding from the file's
    numbers, and -- the run having converged -- no visible change of the saved estimates."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "object_slam_main")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run([exe, DATA, str(tmp_path), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    g, fixed = _read_g2o(tmp_path / "graph.g2o")
    assert g["VERTEX_SE3:EXPMAP"].shape == (58, 8) and g["VERTEX_CUBOID"].shape == (1, 10) and g["EDGE_SE3:EXPMAP"].shape == (57, 2 + 7 + 21)
    assert g["EDGE_SE3_CUBOID"].shape[1] == 2 + 9 + 45 and 40 < len(g["EDGE_SE3_CUBOID"]) <= 58 and fixed == {1}
    # the camera lines are camera-to-world, as VertexSE3Expmap::write has them: the result file's poses
    cams = np.loadtxt(tmp_path / "output_cam_poses.txt")
    assert np.abs(g["VERTEX_SE3:EXPMAP"][:, 1:] - cams[:, 1:]).max() < 1e-11
    assert np.abs(g["VERTEX_CUBOID"][0, 1:] - np.loadtxt(tmp_path / "output_obj_poses.txt")[-1]).max() < 1e-11
    # read + write without optimising: the same file up to the Euler <-> quaternion round trip of the cuboid lines
    out = subprocess.run([exe, "--g2o", str(tmp_path / "graph.g2o"), str(tmp_path / "again.g2o"), "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "loaded 58 cameras, 1 cuboids" in out.stdout, out.stdout + out.stderr
    g2, fixed2 = _read_g2o(tmp_path / "again.g2o")
    assert fixed2 == fixed and sorted(g2) == sorted(g)
    for k in g:
        assert g2[k].shape == g[k].shape and np.abs(g2[k] - g[k]).max() < 1e-12, k
    # optimise the loaded graph: equal to the same graph handed to the library through the Python binding
    out = subprocess.run([exe, "--g2o", str(tmp_path / "graph.g2o"), str(tmp_path / "opt.g2o"), "5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    g3, _ = _read_g2o(tmp_path / "opt.g2o")
    assert np.abs(g3["VERTEX_SE3:EXPMAP"] - g["VERTEX_SE3:EXPMAP"]).max() < 1e-4 and np.abs(g3["VERTEX_CUBOID"] - g["VERTEX_CUBOID"]).max() < 1e-4
    from scipy.spatial.transform import Rotation
    def inv7(v):     # camera-to-world line -> world-to-camera estimate
        R = Rotation.from_quat(v[:, 3:7]); t = -R.inv().apply(v[:, :3]); q = R.inv().as_quat()
        q = np.where(q[:, 3:4] < 0, -q, q)
        return np.concatenate([t, q], axis=1)
    def cub10(m):    # minimal vector -> pose + half sizes
        q = Rotation.from_euler("ZYX", m[:, [5, 4, 3]]).as_quat(); q = np.where(q[:, 3:4] < 0, -q, q)
        return np.concatenate([m[:, :3], q, m[:, 6:9]], axis=1)
    def full(tri, n):
        M = np.zeros((len(tri), n, n)); iu = np.triu_indices(n)
        M[:, iu[0], iu[1]] = tri; M[:, iu[1], iu[0]] = tri
        return M.reshape(len(tri), n * n)
    ce, oe = g["EDGE_SE3_CUBOID"], g["EDGE_SE3:EXPMAP"]
    pr = dict(cams=inv7(g["VERTEX_SE3:EXPMAP"][:, 1:]), cam_fixed=np.array([1] + [0] * 57), cuboids=cub10(g["VERTEX_CUBOID"][:, 1:]), cub_fixed=np.array([0]),
              points=np.zeros((0, 3)), pt_fixed=np.zeros(0, int), e_pt=np.zeros(0, int), e_cam=np.zeros(0, int), e_uv=np.zeros((0, 2)), e_info=np.zeros((0, 4)),
              e_intr=np.zeros((0, 4)), e_huber=np.zeros(0),
              ce_cam=ce[:, 0].astype(int) - 1, ce_cub=ce[:, 1].astype(int), ce_meas=cub10(ce[:, 2:11]), ce_info=full(ce[:, 11:], 9),
              oe_i=oe[:, 0].astype(int) - 1, oe_j=oe[:, 1].astype(int) - 1, oe_meas=inv7(oe[:, 2:9]), oe_info=full(oe[:, 9:], 6))
    G = capi.ba_from_dict(pr, cuboids_first=True)
    G.optimize(5)
    cg, og, _ = G.state()
    # (1e-5: the edges' numeric delta = 1e-9 Jacobians amplify the last-digit differences of the two Euler <-> quaternion conversions to ~1e-7)
    assert np.abs(inv7(g3["VERTEX_SE3:EXPMAP"][:, 1:]) - cg).max() < 1e-5
    assert np.abs(cub10(g3["VERTEX_CUBOID"][:, 1:]) - og).max() < 1e-5
    G.close()


def test_graph_driver_online_mode_from_images(tmp_path):
    """examples/object_slam_main.cpp --online: the reference's online branch in C++ on the C ABI -- colour image in
    (PPM copies of the reference's JPEGs), cs_bgr_to_gray, cs_detect_cuboids_gray with the roll/pitch sampling schedule of
    main_obj.cpp:623, measurement conversion, graph, cs_ba_optimize, result files.  Against the restated driver on the two
    oracles and against the reference's saved output files."""
    pytest.importorskip("PIL")
    import subprocess
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_reference_frames as TR
    import tum_frames
    from oracle import edge_oracle_py as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "object_slam_main")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    ppm = tmp_path / "ppm"
    ppm.mkdir()
    for k in range(58):
        img = np.asarray(Image.open(os.path.join(tum_frames.DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
        with open(ppm / ("%04d.ppm" % k), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + np.ascontiguousarray(img).tobytes())
    # segments from the fixture files ...
    out = subprocess.run([exe, "--online", tum_frames.DATA, str(ppm), os.path.join(tum_frames.DATA, "segments"), str(tmp_path), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    cams_f = np.loadtxt(tmp_path / "output_cam_poses.txt")
    # ... and, the full image-in pipeline, from the library's own line-segment producer (cs_detect_lines_gray): identical files
    out = subprocess.run([exe, "--online", tum_frames.DATA, str(ppm), "detect", str(tmp_path), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    cams = np.loadtxt(tmp_path / "output_cam_poses.txt")
    objs = np.loadtxt(tmp_path / "output_obj_poses.txt")
    assert np.array_equal(cams, cams_f)
    TR.check_online_run_against_saved_outputs(objs, cams[:, 1:])
    obj_r, cam_r, _ = O.run_online_sequence(tum_frames.DATA, lambda k: tum_frames.load_for_online_run(k, E.bgr_to_gray), TR._oracle_detect)
    assert np.abs(objs - obj_r).max() < 1e-5 * max(1.0, np.abs(obj_r).max())
    assert np.abs(cams[:, 1:] - cam_r).max() < 1e-5 * max(1.0, np.abs(cam_r).max())


def _run_sharded_in_threads(pr, n_ranks, iters):
    """n_ranks cs_ba instances on one GPU, one thread each, with an in-process all-reduce (sum / max over threads)."""
    import ctypes as C
    import threading

    probs = []
    for r in range(n_ranks):
        P = capi.ba_from_dict(pr)
        P.set_shard(r, n_ranks)
        probs.append(P)
    assert len({P.reduced_size() for P in probs}) == 1        # every rank takes the same structural decisions (reduced system, elimination)
    barrier = threading.Barrier(n_ranks)
    slots = [None] * n_ranks
    import torch

    def make_cb(rank):
        def cb(ptr, n, on_device, op):
            if on_device:
                t = torch.as_tensor(capi.DeviceDoubles(ptr, n), device="cuda")
                slots[rank] = t.clone()
            else:
                a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,))
                slots[rank] = torch.from_numpy(a.copy())
            barrier.wait()
            tot = slots[0].clone()
            for s in slots[1:]:
                tot = tot + s if op == 0 else torch.maximum(tot, s)
            barrier.wait()
            if on_device:
                t.copy_(tot); torch.cuda.synchronize()
            else:
                a[:] = tot.cpu().numpy()
            return 0
        return cb

    done = [0] * n_ranks
    def work(r):
        done[r] = probs[r].optimize_sharded(iters, make_cb(r))
    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    [t.start() for t in th]; [t.join() for t in th]
    owners = probs[0].landmark_owners()      # the rule in force (separator mode: lowest column; fallback: first camera's subsequence)
    assert all(np.array_equal(owners, P.landmark_owners()) for P in probs)
    cams, cubs, _ = probs[0].state()
    pts = np.zeros_like(pr["points"])
    for r in range(n_ranks):
        pts[owners == r] = probs[r].state()[2][owners == r]
    hist = probs[0].history()
    _run_sharded_in_threads.info = [P.shard_info() for P in probs]
    _run_sharded_in_threads.owners = owners
    for r in range(1, n_ranks):     # every rank ends with the same cameras and cuboids
        cr, orr, _ = probs[r].state()
        assert np.array_equal(cr, cams) and np.array_equal(orr, cubs), (r, np.abs(cr - cams).max(), np.abs(orr - cubs).max())
    for P in probs:
        P.close()
    return done, hist, cams, cubs, pts


@pytest.mark.parametrize("n_ranks,n_cams,sep", [(2, 60, 0), (3, 60, 0), (2, 160, 1), (3, 240, 1), (4, 330, 1)])
def test_sharded_ba_equals_single_rank(n_ranks, n_cams, sep):
    """Sharded BA == the unsharded optimisation (same LM trajectory): small problems through the all-reduce of the partial reduced
    systems (shares narrower than the band), larger ones in separator mode -- interiors factorised per rank, separator complements
    exchanged, the separator system solved by every rank."""
    pr = synth_ba.make_problem(n_cams=n_cams, n_points=70 * n_cams, n_cuboids=max(8, n_cams // 8), seed=21)
    G = capi.ba_from_dict(pr)
    n1 = G.optimize(6)
    chi1, lam1, tr1 = G.history()
    c1, o1, p1 = G.state()
    done, (chiS, lamS, trS), cS, oS, pS = _run_sharded_in_threads(pr, n_ranks, 6)
    info = _run_sharded_in_threads.info
    assert [i["sep_mode"] for i in info] == [sep] * n_ranks, info
    if sep:    # what a rank sends per trial is the separator message + the solution vector, not the band
        assert all(i["bytes_per_trial"] < i["bytes_per_trial_allreduce"] for i in info), info
    assert done == [n1] * n_ranks
    assert np.array_equal(tr1, trS) and np.allclose(chi1, chiS, rtol=1e-9) and np.allclose(lam1, lamS, rtol=1e-9)
    scale = np.abs(p1).max()
    assert np.abs(pS - p1).max() < 1e-7 * scale and np.abs(cS - c1).max() < 1e-7 * scale and np.abs(oS - o1).max() < 1e-7 * scale


def test_sharded_ba_separator_mode_without_cuboid_elimination(monkeypatch):
    """g2o's reduced system (cuboids kept as unknowns, 9-column blocks in the ordering) through the separator mode."""
    monkeypatch.setenv("CS_BA_KEEP_CUBOIDS", "1")
    pr = synth_ba.make_problem(n_cams=260, n_points=12000, n_cuboids=30, seed=5)
    G = capi.ba_from_dict(pr)
    n1 = G.optimize(4)
    chi1, lam1, tr1 = G.history()
    c1, o1, p1 = G.state()
    assert G.reduced_size()[1] == 0
    G.close()
    done, (chiS, lamS, trS), cS, oS, pS = _run_sharded_in_threads(pr, 2, 4)
    assert [i["sep_mode"] for i in _run_sharded_in_threads.info] == [1, 1]
    assert done == [n1] * 2 and np.array_equal(tr1, trS) and np.allclose(chi1, chiS, rtol=1e-9)
    scale = np.abs(p1).max()
    assert np.abs(pS - p1).max() < 1e-7 * scale and np.abs(cS - c1).max() < 1e-7 * scale and np.abs(oS - o1).max() < 1e-7 * scale


def test_low_level_solve_on_a_separator_mode_shard_without_collective_is_refused():
    """cs_ba_solve on a handle sharded in separator mode has no way to obtain the other ranks' separator messages (no callback, no
    communicator): it must say so instead of solving a system assembled from zeros and reporting positive_definite = 1."""
    pr = synth_ba.make_problem(n_cams=160, n_points=70 * 160, n_cuboids=20, seed=21)
    P = capi.ba_from_dict(pr)
    P.set_shard(0, 2)
    assert P.shard_info()["sep_mode"] == 1
    P.build_system()
    with pytest.raises(RuntimeError, match="separator mode"):
        P.solve(1.0)
    P.close()


@pytest.mark.parametrize("shape", [(150, 6000, 30), (300, 9000, 0), (90, 3000, 12)])
def test_banded_solver_matches_dense_solver(shape, monkeypatch):
    """The persistent banded Cholesky (RCM ordering, one grid barrier per 32-column step) and rocSOLVER's dense
    potrf/potrs solve the same damped reduced system: increments equal to rounding, and so is a 5-iteration run."""
    nc, npt, no = shape
    pr = synth_ba.make_problem(n_cams=nc, n_points=npt, n_cuboids=no, seed=31)
    B = capi.ba_from_dict(pr)
    ld, team = B.solver_layout()
    assert ld > 0 and team >= 1, "trajectory-shaped graph must take the banded path"
    monkeypatch.setenv("CS_BA_FORCE_DENSE", "1")
    D = capi.ba_from_dict(pr)
    monkeypatch.delenv("CS_BA_FORCE_DENSE")
    assert D.solver_layout()[0] == 0
    B.build_system(); D.build_system()
    for lam in (1e-3, 50.0):
        ok_b, x_b = B.solve(lam)
        ok_d, x_d = D.solve(lam)
        assert ok_b and ok_d
        assert _rel(x_b, x_d) < 1e-9
    assert B.optimize(5) == D.optimize(5)
    assert np.array_equal(B.history()[2], D.history()[2]) and np.allclose(B.history()[0], D.history()[0], rtol=1e-9)
    for a, b in zip(B.state(), D.state()):
        assert a.shape == b.shape and (a.size == 0 or np.abs(a - b).max() < 1e-8 * max(1.0, np.abs(b).max()))
    B.close(); D.close()


def test_all_cameras_fixed_free_cuboids_and_points_still_optimise():
    """Every camera fixed (the graph driver's frame-0 shape): nothing is left of the cameras-only system, so the cuboids must stay in
    the reduced system (an elimination without a factorisation behind it would leave x = 0 and stop LM after one trial)."""
    pr = synth_ba.make_problem(n_cams=12, n_points=300, n_cuboids=3, seed=8)
    pr = dict(pr); pr["cam_fixed"] = np.ones_like(pr["cam_fixed"])
    G = capi.ba_from_dict(pr)
    n_red, elim = G.reduced_size()
    assert n_red == 9 * 3 and not elim
    chi0 = G.compute_errors()
    n = G.optimize(5)
    chi = G.history()[0]
    R = _oracle(pr)
    assert R.optimize(5) == n
    assert chi[-1] < 0.9 * chi0 and np.allclose(chi, R.history()[0], rtol=1e-6)
    scale = np.abs(R.state()[2]).max()
    for a, b in zip(G.state(), R.state()):
        assert np.abs(a - b).max() < 1e-5 * scale
    G.close(); R.close()


def test_banded_solver_reports_indefinite_system():
    """A non-positive pivot must surface as solve() == False (g2o's 'Cholesky failure', block_solver.hpp:580-584), not hang."""
    pr = synth_ba.make_problem(n_cams=150, n_points=6000, n_cuboids=30, seed=32)
    B = capi.ba_from_dict(pr)
    assert B.solver_layout()[0] > 0
    B.build_system()
    ok, _ = B.solve(-1e12)      # a hugely negative damping makes every diagonal negative
    assert not ok
    ok, x = B.solve(1.0)        # and the handle stays usable
    assert ok and np.isfinite(x).all()
    B.close()


def test_full_size_c4_properties(monkeypatch):
    """BASELINE.json's C4 (1 000 cameras, 200 000 landmarks, 500 cuboids, ~1 M projection edges) is too large for the
    CPU oracle in test time; checked through properties instead: chi2 falls monotonically over accepted LM steps and
    drops by more than 4x in four iterations, the estimate moves toward the generating truth, and the persistent banded solver and
    the dense rocSOLVER path give the same trajectory."""
    pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
    B = capi.ba_from_dict(pr)
    ld, team = B.solver_layout()
    assert 0 < ld < 400 and team >= 4
    p0 = B.state()[2].copy()
    n_b = B.optimize(4)
    chi, lam, tr = B.history()
    assert n_b == 4 and np.all(np.diff(chi[:n_b]) < 0) and chi[n_b - 1] < 0.25 * chi[0]
    truth = pr["truth"]
    free = np.asarray(pr["pt_fixed"]) == 0
    e0 = np.linalg.norm(p0[free] - truth["points"][free], axis=1).mean()
    e1 = np.linalg.norm(B.state()[2][free] - truth["points"][free], axis=1).mean()
    assert e1 < 0.7 * e0
    monkeypatch.setenv("CS_BA_FORCE_DENSE", "1")
    D = capi.ba_from_dict(pr)
    monkeypatch.delenv("CS_BA_FORCE_DENSE")
    assert D.solver_layout()[0] == 0 and D.optimize(4) == 4
    assert np.array_equal(D.history()[2], tr) and np.allclose(D.history()[0], chi, rtol=1e-9)
    for a, b in zip(B.state(), D.state()):
        assert np.abs(a - b).max() < 1e-7 * max(1.0, np.abs(b).max())
    B.close(); D.close()


@pytest.mark.parametrize("shape,keep", [((1000, 200000, 500), False), ((120, 6000, 16), True), ((200, 20000, 50), False)])
def test_one_linearisation_at_full_c4_size_against_the_oracle(shape, keep, monkeypatch):
    """BASELINE.json's C4 -- 1 000 cameras, 200 000 landmarks, 500 cuboids, ~1 M projection edges -- compared with oracle/ba_oracle.cpp
    itself, not only through properties: robust chi2, the gradient b, every H_ll and H_pl block, EVERY block of the damped reduced system
    (the oracle forms g2o's camera + cuboid system, block_solver.hpp:373-439; the device's extra cuboid elimination is applied to it with
    numpy) and one damped solve (the oracle's S factorised by LAPACK instead of its 24-minute textbook LDL^T; its own landmark
    back-substitution).  Bars as in test_system_parity_all_edge_types: 1e-9 for what is closed-form, 1e-5 for what passes through the
    numeric (delta = 1e-9) Jacobians of the cuboid and odometry edges.  The small shapes run the same comparison with and without the
    cuboid elimination."""
    from oracle import ba_parity
    if keep:
        monkeypatch.setenv("CS_BA_KEEP_CUBOIDS", "1")
    nc, npt, no = shape
    pr = synth_ba.make_problem(n_cams=nc, n_points=npt, n_cuboids=no, seed=42)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    assert G.reduced_size()[1] == (not keep) or (not keep and nc < 1000)      # (C4: the cuboids are eliminated; a small graph may keep them)
    G.build_system(dense_hpp=False)
    hc, ho, hp = G.vertex_hessians()
    lam = 1e-5 * max(np.abs(np.diagonal(hc, axis1=1, axis2=2)).max(), np.abs(np.diagonal(ho, axis1=1, axis2=2)).max(), np.abs(np.diagonal(hp, axis1=1, axis2=2)).max())
    d = ba_parity.compare_linearisation(G, R, pr, lam)
    print("C4 parity" if nc == 1000 else "parity", shape, "keep_cuboids" if keep else "cuboids eliminated", d)
    assert d["chi2"] < 1e-11 and d["H_ll"] < 1e-11 and d["H_pl"] < 1e-11
    # measured at C4 (MI355X): b 1e-12, S 1e-13 of its largest entry, 1.3e-7 in its worst block (numeric delta = 1e-9 Jacobians of the
    # cuboid / odometry edges inside), x 1e-10
    assert d["b"] < 1e-9 and d["S"] < 1e-9 and d["b_schur"] < 1e-9
    if "S_worst_block" in d:
        assert d["S_worst_block"] < 1e-5 and d["S_nonzero_where_oracle_is_zero"] == 0.0 and d["S_blocks_compared"] > nc
    assert d["solve_ok"] and d["x_pose"] < 1e-7 and d["x_landmarks"] < 1e-7
    G.close(); R.close()


def test_three_lm_iterations_at_full_c4_size_against_the_oracle():
    """BASELINE.json's C4 (1 000 cameras / 200 000 landmarks / 500 cuboids) as a TRAJECTORY, north_star's bar at the size it names: three LM
    iterations of the device beside three of oracle/ba_oracle.cpp -- its residuals, linearisation, Schur complement, back-substitution,
    update and LM control (optimization_algorithm_levenberg.cpp:61-189) in full; only the dense solve of its 10 494-unknown reduced system
    goes through LAPACK (Problem.use_lapack_solver: the restatement's textbook LDL^T would take 24 minutes per trial).  Same accept /
    reject sequence, chi2 / lambda histories to 1e-6, states to 1e-5 relative.  (~25 s of oracle time.)"""
    from oracle import ba_parity
    pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    assert G.sizes() == (6 * 999 + 9 * 500, 3 * 200000)
    assert G.solver_path() == "band" and G.band_order() == (True, 6)      # (the cameras-only system, 5 994 unknowns: 47 blocks of 128, six levels)
    R.use_lapack_solver()
    d = ba_parity.compare_trajectory(G, R, 3)
    print("C4 trajectory", d)
    assert d["iterations_device"] == d["iterations_oracle"] == 3 and d["same_trial_sequence"]
    assert d["chi2"] < 1e-6 and d["lambda"] < 1e-6
    assert d["chi2_first_last"][1] < 0.5 * d["chi2_first_last"][0]          # (the run moved: a trajectory, not three rejected trials)
    assert d["points"] < 1e-5 and d["camera_positions"] < 1e-5 and d["camera_quaternions"] < 1e-5
    assert d["cuboid_positions"] < 1e-5 and d["cuboid_quaternions_and_sizes"] < 1e-5
    G.close(); R.close()


def test_lm_properties_user_lambda_init_and_max_trials_against_the_oracle():
    """OptimizationAlgorithmLevenberg's two properties (optimization_algorithm_levenberg.cpp:50-51, setters :191-199) through
    cs_ba_set_lm_params: a user lambda far below tau * max |H_jj| makes the first steps overshoot -- iterations of up to seven trials, the
    nu-doubling branch (:143-148) taken again and again --, and the device has to walk the oracle's accept / reject sequence trial for trial;
    with maxTrialsAfterFailure = 2 the run ends at the iteration that uses both trials (:151), on both sides."""
    from oracle import ba_parity
    pr = synth_ba.make_problem(n_cams=60, n_points=3000, n_cuboids=8, seed=42)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    G.set_lm_params(1e-6, 10); R.set_lm_params(1e-6, 10)
    d = ba_parity.compare_trajectory(G, R, 5)
    print("user lambda 1e-6", d)
    assert d["iterations_device"] == d["iterations_oracle"] == 5 and d["same_trial_sequence"]
    assert sum(d["trials_oracle"]) >= 12 and max(d["trials_oracle"]) >= 4          # (rejected trials in a row)
    assert d["chi2"] < 1e-6 and d["lambda"] < 1e-6
    assert d["points"] < 1e-5 and d["camera_positions"] < 1e-5 and d["camera_quaternions"] < 1e-5
    G.close(); R.close()
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    G.set_lm_params(1e-8, 2); R.set_lm_params(1e-8, 2)
    n_g, n_r = G.optimize(6), R.optimize(6)
    assert n_g == n_r and n_r < 6                       # both gave up at the same iteration
    assert np.array_equal(G.history()[2], R.history()[2]) and int(R.history()[2][-1]) == 2
    assert np.allclose(G.history()[0], R.history()[0], rtol=1e-9)
    with pytest.raises(Exception):
        G.set_lm_params(0.0, 0)
    G.close(); R.close()


def test_rejected_trials_at_full_c4_size_against_the_oracle():
    """The branch the C4 trajectory test above never takes (its three iterations are accepted at the first trial): at C4's full size with a
    user lambda of 1e-4 (cs_ba_set_lm_params; at 1e-6 the first step is so badly conditioned that the chi2 it lands on is only reproducible to 3e-5,
    states to 1e-7) the second iteration rejects trial after trial -- estimates restored, lambda times nu, nu doubled
    (optimization_algorithm_levenberg.cpp:143-148), the speculated linearisation of the next iteration discarded every time -- and the device
    has to walk the oracle's sequence.  (The oracle's 10 494-unknown solves through LAPACK: ~8 s per trial.)"""
    from oracle import ba_parity
    pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    R.use_lapack_solver()
    G.set_lm_params(1e-4, 10); R.set_lm_params(1e-4, 10)
    d = ba_parity.compare_trajectory(G, R, 2)
    print("C4 with rejected trials", d)
    assert d["iterations_device"] == d["iterations_oracle"] == 2 and d["same_trial_sequence"]
    assert max(d["trials_oracle"]) >= 3                                             # (rejected trials in a row)
    assert d["chi2"] < 1e-6 and d["lambda"] < 1e-6
    assert d["points"] < 1e-5 and d["camera_positions"] < 1e-5 and d["camera_quaternions"] < 1e-5
    assert d["cuboid_positions"] < 1e-5 and d["cuboid_quaternions_and_sizes"] < 1e-5
    G.close(); R.close()


@pytest.mark.parametrize("views", [2, 4, 6, 9, 12])
def test_fused_linearise_schur_kernel_equals_the_classic_pair_bitwise(views, monkeypatch):
    """Round 5: from the second LM iteration on cs_ba_optimize linearises the landmark side of the projection edges INSIDE the Schur kernels
    (ba_lin_schur_kernel<MT>: one pass over the edges, H_pl kept in LDS as the matrix-core operands) instead of ba_lin_pt_kernel +
    ba_schur_fused_kernel<MT>.  Same proj_linearize, same sums in the same order, same products: the LM run must be BIT-identical to the
    classic pair (CS_BA_FUSE_LIN=0) -- chi2 / lambda histories, trial counts and every state -- for tracks of up to 2 / 4 / 6 / 9 / 12 views
    (the five tile-count instances), with Huber kernels on and fixed vertices in the graph."""
    pr = synth_ba.make_problem(n_cams=120, n_points=6000, n_cuboids=20, seed=7 + views, obs_per_point=views)
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CS_BA_FUSE_LIN", flag)
        G = capi.ba_from_dict(pr)
        assert G.schur_layout()[0]
        n = G.optimize(6)
        runs.append((n, G.history(), G.state()))
        G.close()
    (n1, h1, s1), (n0, h0, s0) = runs
    assert n1 == n0 >= 3
    for a, b in zip(h1, h0):
        assert np.array_equal(a, b)
    for a, b in zip(s1, s0):
        assert np.array_equal(a, b)
    assert h1[0][-1] < 0.5 * h1[0][0]


@pytest.mark.parametrize("vary", ["none", "info", "intr", "both"])
def test_uniform_information_and_intrinsics_constants_equal_the_per_edge_records(vary, monkeypatch):
    """Round 5: when every projection edge carries the same information matrix and / or the same intrinsics (one camera, one sigma) the
    kernels read 4 + 4 doubles (BaView::info_u / intr_u) instead of the 32-byte per-edge records.  Same values into the same arithmetic:
    the LM run must be BIT-identical to the per-edge records (CS_BA_UNIFORM=0), and with per-edge values that do differ (pyramid-level
    sigmas, a second camera) the handle must fall back to the records on its own and agree with the oracle."""
    pr = synth_ba.make_problem(n_cams=60, n_points=3000, n_cuboids=8, seed=91, obs_per_point=5)
    rng = np.random.default_rng(5)
    n_e = len(pr["e_pt"])
    if vary in ("info", "both"):
        pr["e_info"] = pr["e_info"] * (1.2 ** -rng.integers(0, 8, n_e))[:, None]
    if vary in ("intr", "both"):
        second = rng.random(n_e) < 0.3
        pr["e_intr"] = np.where(second[:, None], pr["e_intr"] * np.array([1.01, 0.99, 1.0, 1.0]), pr["e_intr"])
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("CS_BA_UNIFORM", flag)
        G = capi.ba_from_dict(pr)
        chi0 = G.compute_errors()
        n = G.optimize(5)
        runs.append((chi0, n, G.history(), G.state()))
        G.close()
    (c1, n1, h1, s1), (c0, n0, h0, s0) = runs
    assert c1 == c0 and n1 == n0 >= 3
    for a, b in zip(h1, h0):
        assert np.array_equal(a, b)
    for a, b in zip(s1, s0):
        assert np.array_equal(a, b)
    R = _oracle(pr)
    chi_r = R.compute_errors()[0]
    assert abs(c1 - chi_r) < 1e-9 * chi_r
    assert R.optimize(5) == n1 and np.allclose(h1[0], R.history()[0], rtol=1e-5)


def test_appended_edges_with_other_information_switch_the_handle_back_to_per_edge_records():
    """A handle whose first edges all share one information matrix reads the constant; edges appended later with another sigma must
    switch it to the per-edge records: chi2 and the LM run equal a handle that was given all edges at once."""
    pr = synth_ba.make_problem(n_cams=40, n_points=1500, n_cuboids=4, seed=17, obs_per_point=4)
    n_e = len(pr["e_pt"]); h = n_e // 2
    info = pr["e_info"].copy(); info[h:] *= 0.5
    args = lambda sl: (pr["e_pt"][sl], pr["e_cam"][sl], pr["e_uv"][sl], info[sl], pr["e_intr"][sl], pr["e_huber"][sl])
    mk = lambda: capi.BaProblem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"])
    A, B = mk(), mk()
    A.set_edges_proj(*args(slice(0, h))); A.append_edges_proj(*args(slice(h, n_e)))
    B.set_edges_proj(*args(slice(0, n_e)))
    for G in (A, B):
        G.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
        G.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
    assert A.compute_errors() == B.compute_errors()
    assert A.optimize(4) == B.optimize(4)
    for a, b in zip(A.history(), B.history()):
        assert np.array_equal(a, b)
    for a, b in zip(A.state(), B.state()):
        assert np.array_equal(a, b)
    pr2 = dict(pr); pr2["e_info"] = info
    R = _oracle(pr2); R.optimize(4)
    assert np.allclose(A.history()[0], R.history()[0], rtol=1e-5)
    A.close(); B.close()


def test_cuboid_projection_edges_system_and_optimize_parity():
    """EdgeSE3CuboidProj (4-dim bounding-box error of the projected cuboid, numeric Jacobians) next to the other three
    edge types: linear system and a 6-iteration LM run against the oracle."""
    pr = synth_ba.make_problem(n_cams=40, n_points=2000, n_cuboids=6, seed=13, bbox_edges=True)
    assert len(pr["pe_cam"]) == len(pr["ce_cam"]) > 50
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    chi_r = R.compute_errors()[0]
    assert abs(G.compute_errors() - chi_r) < 1e-9 * chi_r
    Hpp_g, Hll_g, Hpl_g, b_g = G.build_system()
    Hpp_r, Hll_r, Hpl_r, b_r = R.build_system()
    # numeric Jacobians of pixel-sized errors with a 1e-9 step: ~1e-6 relative agreement is what the step allows
    assert _rel(Hpp_g, Hpp_r) < 1e-5 and _rel(b_g, b_r) < 1e-5 and _rel(Hll_g, Hll_r) < 1e-11
    n_g, n_r = G.optimize(6), R.optimize(6)
    assert n_g == n_r
    # chi2 along the run: the 1e-9-step Jacobians carry ~1e-7 of noise that the iterations amplify -- against the oracle the
    # dense rocSOLVER path ends up at +2.4e-6 in iteration 5 and the banded one at -2.4e-6; the bar is 1e-5
    assert np.array_equal(G.history()[2], R.history()[2]) and np.allclose(G.history()[0], R.history()[0], rtol=1e-5)
    cg, og, pg = G.state()
    cr, orr, prr = R.state()
    scale = np.abs(prr).max()
    assert np.abs(pg - prr).max() < 1e-5 * scale and np.abs(cg - cr).max() < 1e-5 * scale and np.abs(og - orr).max() < 1e-5 * scale
    # the box edges must matter: without them the optimum differs
    pr2 = dict(pr); pr2["pe_cam"] = pr["pe_cam"][:0]
    G2 = capi.ba_from_dict(pr2); G2.optimize(6)
    assert np.abs(G2.state()[1] - og).max() > 1e-4
    G.close(); G2.close()


def test_band_solver_harness_shapes():
    """tools/microbench/band_bench: random SPD bands through the persistent Cholesky + substitution kernels, residual of
    A x = b computed on the host.  Shapes cover the nested four-front order (large n, bandwidth <= 256: separator rows,
    Schur accumulators, dense separator block; halves of unequal size; bandwidths at and next to multiples of 32), the
    two-front order (the same shapes forced with CS_BAND_TWO_FRONTS, and the ones too small or too wide for nesting), the
    one-sided order (fewer than 8 column blocks outside the middle), a middle block barely wider than the band,
    single-chunk and multi-chunk strips (bw > 192), tiny bandwidths, and sizes that are not multiples of the block."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "band_bench")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    shapes = [(10494, 183), (5000, 700), (3000, 40), (1000, 300), (777, 12), (500, 200), (439, 183), (440, 184), (471, 184), (300, 290), (2049, 33), (4097, 2),
              (5000, 257), (2001, 193), (2000, 192), (1500, 33), (1473, 161), (1217, 100), (1345, 129), (9999, 65), (138, 61), (150, 40), (129, 20), (260, 100)]
    for n, ld in shapes:
        for env in ({}, {"CS_BAND_TWO_FRONTS": "1"}):
            if env and (n, ld) not in [(10494, 183), (2001, 193), (1500, 33)]:
                continue
            out = subprocess.run([exe, str(n), str(ld), "2"], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
            assert out.returncode == 0, out.stderr
            res = [float(l.split("residual")[1].split()[0]) for l in out.stdout.splitlines() if "residual" in l]
            infos = [int(l.split("info")[1].split()[0]) for l in out.stdout.splitlines() if "info" in l]
            assert len(res) == 2 and max(res) < 1e-12 and infos == [0, 0], (n, ld, env, out.stdout)


def test_block_cyclic_reduction_harness_shapes():
    """tools/microbench/band_bench, order 2: the same random SPD bands through the block-cyclic-reduction solver (csrc/bcr_kernels.hip:
    bandwidth <= 128, blocks of 128 unknowns eliminated in odd-even order, six levels at C4's 5 994 unknowns).  Shapes: C4's and
    g2o's-system-sized bands, one / two / three / many blocks, a last block of one unknown, a full last block, the widest band the
    blocks admit (LD = 129), tiny bandwidths, block sizes below 128 (padding rows inside every block), and a different system every
    repetition (the kernels keep nothing between solves)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build_tmp", "band_bench")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    shapes = [(5994, 120, 128), (10494, 120, 128), (1194, 120, 128), (640, 120, 128), (3000, 40, 128), (777, 12, 128), (2049, 33, 128), (4097, 2, 128), (1217, 100, 128),
              (1345, 129, 128), (9999, 65, 128), (260, 100, 128), (129, 20, 128), (257, 129, 128), (256, 129, 128), (384, 90, 128), (1500, 101, 120), (1000, 60, 64), (2000, 97, 96),
              (130, 31, 31)]
    for n, ld, bv in shapes:
        out = subprocess.run([exe, str(n), str(ld), "3", "2", str(bv)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("rep")]
        assert len(lines) == 3 and all(float(l.split("residual")[1].split()[0]) < 1e-12 and int(l.split("info")[1].split()[0]) == 0 for l in lines), (n, ld, bv, out.stdout)


def test_band_write_through_handoffs_equal_the_fenced_build_bitwise():
    """The persistent band kernels hand values between workgroups (other XCDs) through write-through stores + drained counters
    (BAND_WT = 1) instead of agent-scope release / acquire fences.  The same kernels built with BAND_WT = 0 (plain stores, fences:
    the form the LLVM memory model orders) must give bit-identical factors and solutions on a multi-XCD device -- a hand-off that the
    write-through form fails to publish shows up here as a different hash (or a bad residual), not as a rare wrong factor in the field."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, exe_f = os.path.join(root, "build_tmp", "band_bench"), os.path.join(root, "build_tmp", "band_bench_fence")
    if not (os.path.exists(exe) and os.path.exists(exe_f)):
        import __graft_entry__
        __graft_entry__.build()
    for n, ld, one_sided in [(5994, 120, 0), (10494, 183, 0), (1902, 190, 0), (840, 241, 0), (2500, 97, 0), (630, 120, 1), (2000, 64, 1), (1345, 129, 0)]:
        outs = []
        for e in (exe, exe_f):
            out = subprocess.run([e, str(n), str(ld), "6", str(one_sided)], capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr
            lines = [l for l in out.stdout.splitlines() if l.startswith("rep")]
            assert len(lines) == 6 and all(float(l.split("residual")[1].split()[0]) < 1e-12 and int(l.split("info")[1].split()[0]) == 0 for l in lines), out.stdout
            outs.append([l.split("hash")[1].strip() for l in lines])
        assert outs[0] == outs[1], (n, ld, one_sided, outs)


def test_c3_at_its_named_size_against_the_oracle():
    """BASELINE.json config C3 at full size -- 200 cameras, 20 000 points, 50 cuboids, 10 LM iterations -- against the CPU
    oracle: same accept / reject sequence, chi2 history to 1e-6, states to 1e-5 relative (north_star's bar)."""
    pr = synth_ba.make_problem(n_cams=200, n_points=20000, n_cuboids=50, seed=42)
    assert len(pr["e_pt"]) > 90000 and len(pr["ce_cam"]) == 1000
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    assert G.sizes() == (6 * 199 + 9 * 50, 3 * 20000)
    n_g, n_r = G.optimize(10), R.optimize(10)
    assert n_g == n_r == 10
    chi_g, lam_g, tr_g = G.history()
    chi_r, lam_r, tr_r = R.history()
    assert np.array_equal(tr_g, tr_r)
    assert np.allclose(chi_g, chi_r, rtol=1e-6) and np.allclose(lam_g, lam_r, rtol=1e-6)
    assert chi_g[-1] < 0.2 * chi_g[0]
    cg, og, pg = G.state()
    cr, orr, prr = R.state()
    scale = np.abs(prr).max()
    assert np.abs(pg - prr).max() < 1e-5 * scale
    assert np.abs(cg[:, :3] - cr[:, :3]).max() < 1e-5 * scale and np.abs(cg[:, 3:] - cr[:, 3:]).max() < 1e-5
    assert np.abs(og[:, :3] - orr[:, :3]).max() < 1e-5 * scale and np.abs(og[:, 3:] - orr[:, 3:]).max() < 1e-5
    G.close(); R.close()


def test_loop_closure_stays_on_the_banded_solver_and_matches_the_oracle():
    """A closed trajectory (the last cameras re-observe the first landmarks, an odometry edge joins camera n-1 to camera 0): the
    block graph of the reduced system is a ring.  Reverse Cuthill-McKee walks a ring from one vertex in both directions, so the
    bandwidth about doubles against the open chain -- still a narrow band, still the persistent banded Cholesky, not the dense
    fallback -- and the LM trajectory is the oracle's."""
    open_pr = synth_ba.make_problem(n_cams=300, n_points=9000, n_cuboids=12, seed=11)
    pr = synth_ba.make_problem(n_cams=300, n_points=9000, n_cuboids=12, seed=11, loop=True)
    assert len(pr["oe_i"]) == 300 and pr["oe_j"][-1] == 0
    G0 = capi.ba_from_dict(open_pr)
    G0.optimize(1)
    ld_open, _ = G0.solver_layout()
    n_open, _ = G0.reduced_size()
    G0.close()
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    n_g, n_r = G.optimize(6), R.optimize(6)
    ld_loop, _ = G.solver_layout()
    n_loop, _ = G.reduced_size()
    assert ld_open > 0 and ld_loop > 0, (ld_open, ld_loop)                      # banded in both cases
    assert ld_loop <= 2.6 * ld_open and ld_loop < n_loop // 4, (ld_open, ld_loop, n_open, n_loop)
    assert n_g == n_r == 6
    chi_g, lam_g, tr_g = G.history()
    chi_r, lam_r, tr_r = R.history()
    assert np.array_equal(tr_g, tr_r)
    assert np.allclose(chi_g, chi_r, rtol=1e-6) and np.allclose(lam_g, lam_r, rtol=1e-6)
    cg, og, pg = G.state()
    cr, orr, prr = R.state()
    scale = np.abs(prr).max()
    assert np.abs(pg - prr).max() < 1e-5 * scale
    assert np.abs(cg[:, :3] - cr[:, :3]).max() < 1e-5 * scale and np.abs(cg[:, 3:] - cr[:, 3:]).max() < 1e-5
    assert np.abs(og[:, :3] - orr[:, :3]).max() < 1e-5 * scale and np.abs(og[:, 3:] - orr[:, 3:]).max() < 1e-5
    print("loop closure: band_ld %d (open chain %d), reduced unknowns %d" % (ld_loop, ld_open, n_loop))
    G.close(); R.close()


def test_c5_eight_shards_of_the_c4_problem_equal_the_single_rank_run():
    """BASELINE.json config C5: the C4 problem (1 000 cameras, 200 000 points, 500 cuboids) cut into 8 camera subsequences.
    Eight cs_ba handles share one GPU here (one thread each, in-process all-reduce); the sharded LM trajectory must be the
    single-rank one: same trials, chi2 / lambda histories to 1e-9, states to 1e-7 of the scene scale."""
    pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
    G = capi.ba_from_dict(pr)
    n1 = G.optimize(3)
    chi1, lam1, tr1 = G.history()
    c1, o1, p1 = G.state()
    G_n_red = G.reduced_size()[0]
    G.close()
    done, (chiS, lamS, trS), cS, oS, pS = _run_sharded_in_threads(pr, 8, 3)
    assert done == [n1] * 8
    assert np.array_equal(tr1, trS) and np.allclose(chi1, chiS, rtol=1e-9) and np.allclose(lam1, lamS, rtol=1e-9)
    scale = np.abs(p1).max()
    assert np.abs(pS - p1).max() < 1e-7 * scale and np.abs(cS - c1).max() < 1e-7 * scale and np.abs(oS - o1).max() < 1e-7 * scale
    # separator mode at this size: what a rank sends per LM trial is its separator message (3 w^2 + 2 w doubles) + the solution vector
    # + three scalars -- under half a megabyte, against the 5.8 MB band an all-reduce of [S | b] would move
    info = _run_sharded_in_threads.info
    assert all(i["sep_mode"] == 1 for i in info), info
    assert max(i["bytes_per_trial"] for i in info) < 500_000 < min(i["bytes_per_trial_allreduce"] for i in info), info
    assert max(i["interior_n"] for i in info) < 0.2 * G_n_red
    print("C5: separator system %d unknowns (w_max %d), interiors %s, bytes per trial and rank %d (all-reduce of the band: %d)"
          % (info[0]["n_sep"], info[0]["w_max"], [i["interior_n"] for i in info], info[0]["bytes_per_trial"], info[0]["bytes_per_trial_allreduce"]))
    # every shard owns a share of the landmarks and the shares partition them
    owners = _run_sharded_in_threads.owners
    cnt = np.bincount(owners, minlength=8)
    assert cnt.sum() == 200000 and cnt.min() > 0.5 * 200000 / 8


def test_stepwise_abi_driven_like_g2o_levenberg():
    """The step-wise C ABI driven exactly the way g2o's OptimizationAlgorithmLevenberg drives a g2o::Solver
    (optimization_algorithm_levenberg.cpp:61-189 through adapters/block_solver_hip.h): estimates uploaded
    (cs_ba_set_estimates), computeActiveErrors (cs_ba_compute_errors), buildSystem (cs_ba_build_system), lambda_0 from the
    vertices' mapped diagonal blocks (cs_ba_get_vertex_hessians = v->hessian(j, j), :166-180), then per trial push / setLambda +
    solve / update / chi2 / pop.  The host-side loop must land on cs_ba_optimize()'s trajectory, and lambda_0 on the oracle's."""
    pr = synth_ba.make_problem(n_cams=40, n_points=2500, n_cuboids=8, seed=3)
    A = capi.ba_from_dict(pr)       # driven step by step
    B = capi.ba_from_dict(pr)       # one-shot reference
    n_b = B.optimize(6)
    chi_b, lam_b, tr_b = B.history()
    cams, cubs, pts = (np.asarray(pr[k], float).copy() for k in ("cams", "cuboids", "points"))
    lam, ni, n_bad = -1.0, 2.0, 0
    chis, lams, trials = [], [], []
    for it in range(6):
        A.set_estimates(cams, cubs, pts)                 # adapter: upload_estimates() in buildSystem()
        cur = A.compute_errors()
        ini = tmp = cur
        A.build_system()
        if it == 0:
            hc, ho, hp = A.vertex_hessians()
            md = max(np.abs(np.einsum("kii->ki", hc)).max(), np.abs(np.einsum("kii->ki", ho)).max(), np.abs(np.einsum("kii->ki", hp)).max())
            lam = 1e-5 * md
            assert np.all(hc[np.asarray(pr["cam_fixed"]) != 0] == 0)      # fixed vertices carry no block
        rho, q = 0.0, 0
        while True:
            A.push()
            ok, _ = A.solve(lam)
            b, x = A.system_vectors()
            A.update()
            tmp = A.compute_errors()
            if not ok:
                tmp = np.finfo(float).max
            scale = float(np.dot(x, lam * x + b)) + 1e-3
            rho = (cur - tmp) / scale
            if rho > 0 and np.isfinite(tmp):
                lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                ni = 2.0
                cur = tmp
            else:
                lam *= ni
                ni *= 2
                A.pop()
            q += 1
            if not (rho < 0 and q < 10):
                break
        cams, cubs, pts = A.state()                      # g2o keeps the estimates on its side between iterations
        chis.append(cur); lams.append(lam); trials.append(q)
        if q == 10 or rho == 0:
            break
        n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
        if n_bad >= 3:
            break
    assert len(chis) == n_b and trials == list(tr_b)
    assert np.allclose(chis, chi_b, rtol=1e-9) and np.allclose(lams, lam_b, rtol=1e-9)
    # (the estimates make a host round trip per iteration -- re-normalised quaternions -- and the cuboid / odometry Jacobians are
    # 1e-9-step differences, which amplify that last-bit change: measured 3e-9 of the scene scale)
    for a, b in zip(A.state(), B.state()):
        assert np.abs(a - b).max() < 1e-7 * max(1.0, np.abs(b).max())
    # lambda_0 is the oracle's (max |H_jj| over all free vertices, landmarks included)
    C2 = capi.ba_from_dict(pr)
    C2.compute_errors(); C2.build_system()
    hc, ho, hp = C2.vertex_hessians()
    md = max(np.abs(np.einsum("kii->ki", hc)).max(), np.abs(np.einsum("kii->ki", ho)).max(), np.abs(np.einsum("kii->ki", hp)).max())
    Hpp_r, Hll_r, _, _ = _oracle(pr).build_system()
    md_r = max(np.abs(np.diag(Hpp_r)).max(), np.abs(Hll_r[:, [0, 4, 8]]).max())
    assert abs(md - md_r) < 1e-5 * md_r
    A.close(); B.close(); C2.close()


@pytest.mark.parametrize("obs", [2, 5, 7, 9, 10, 13, 14])
def test_fused_mfma_schur_matches_the_pair_major_path_and_the_oracle(obs, monkeypatch):
    """The Schur complement formed per segment of landmarks on the matrix cores (v_mfma_f64_16x16x4_f64; one to five
    16-row tiles for landmarks seen by <= 2 / 5 / 7 / 10 / 13 cameras) against the pair-major kernel (CS_BA_SCHUR_PAIRS=1, also what
    a problem made of longer tracks selects: 14 here) and against the oracle's block_solver.hpp:385-431 restatement."""
    pr = synth_ba.make_problem(n_cams=60, n_points=3000, n_cuboids=6, seed=17, obs_per_point=obs)
    kmax = np.bincount(np.asarray(pr["e_pt"])).max()
    assert kmax == obs
    F = capi.ba_from_dict(pr)
    fused, n_seg, n_part, n_blk = F.schur_layout()
    assert fused == (obs <= 13)
    monkeypatch.setenv("CS_BA_SCHUR_PAIRS", "1")
    Q = capi.ba_from_dict(pr)
    assert Q.schur_layout()[0] is False
    monkeypatch.delenv("CS_BA_SCHUR_PAIRS")
    if fused:
        assert n_seg > 0 and n_part < 0.5 * sum(k * (k + 1) // 2 for k in np.bincount(np.asarray(pr["e_pt"])))   # far fewer partial blocks than (landmark, pair) entries
        if not F.reduced_size()[1]:      # (with the cuboids eliminated too the fused path has the camera-camera blocks they fill in on top)
            assert n_blk == Q.schur_layout()[3]
    R = _oracle(pr)
    for P in (F, Q, R):
        P.compute_errors() if P is not R else None
        P.build_system()
    for lam in (1e-3, 30.0):
        ok_f, x_f = F.solve(lam)
        ok_q, x_q = Q.solve(lam)
        ok_r, x_r = R.solve(lam)
        assert ok_f and ok_q and ok_r
        assert _rel(x_f, x_q) < 1e-8          # same products, differently associated (2-view landmarks condition the system worst: 2e-10 measured)
        assert _rel(x_f, x_r) < 1e-5          # numeric cuboid / odometry Jacobians inside
    assert F.optimize(5) == Q.optimize(5)
    assert np.array_equal(F.history()[2], Q.history()[2]) and np.allclose(F.history()[0], Q.history()[0], rtol=1e-9)
    for a, b in zip(F.state(), Q.state()):
        assert a.size == 0 or np.abs(a - b).max() < 1e-8 * max(1.0, np.abs(b).max())
    F.close(); Q.close(); R.close()


def test_long_tracks_in_the_tail_keep_the_fused_build(monkeypatch):
    """A map with a TAIL of long tracks (7 % of the landmarks seen from up to 20 cameras, the rest from <= 5): the long ones go through the
    same segments and destination schedule with plain multiply-adds (ba_schur_long_kernel, tracks of up to 64 views), so the problem
    keeps the matrix-core Schur build and the cuboid elimination -- before, one such landmark sent all of it to the pair-major path.
    Same damped solves and LM run as that path (CS_BA_SCHUR_PAIRS=1) and as the oracle."""
    a = synth_ba.make_problem(n_cams=60, n_points=2800, n_cuboids=6, seed=17)
    b = synth_ba.make_problem(n_cams=60, n_points=200, n_cuboids=0, seed=18, obs_per_point=40)
    pr = dict(a)
    off = len(a["points"])
    pr["points"] = np.concatenate([a["points"], b["points"]]); pr["pt_fixed"] = np.concatenate([a["pt_fixed"], b["pt_fixed"]])
    for k in ("e_cam", "e_uv", "e_info", "e_intr", "e_huber"):
        pr[k] = np.concatenate([a[k], b[k]])
    pr["e_pt"] = np.concatenate([a["e_pt"], b["e_pt"] + off]).astype(np.int32)
    pr["truth"] = None
    assert np.bincount(pr["e_pt"]).max() == 40
    F = capi.ba_from_dict(pr)
    fused, n_seg, n_part, n_blk = F.schur_layout()
    assert fused and F.reduced_size()[1]
    monkeypatch.setenv("CS_BA_SCHUR_PAIRS", "1")
    Q = capi.ba_from_dict(pr)
    assert Q.schur_layout()[0] is False
    monkeypatch.delenv("CS_BA_SCHUR_PAIRS")
    R = _oracle(pr)
    for P in (F, Q, R):
        P.compute_errors() if P is not R else None
        P.build_system()
    for lam in (1e-3, 30.0):
        ok_f, x_f = F.solve(lam)
        ok_q, x_q = Q.solve(lam)
        ok_r, x_r = R.solve(lam)
        assert ok_f and ok_q and ok_r
        assert _rel(x_f, x_q) < 1e-8 and _rel(x_f, x_r) < 1e-5
    assert F.optimize(5) == Q.optimize(5) == R.optimize(5)
    assert np.array_equal(F.history()[2], Q.history()[2]) and np.allclose(F.history()[0], Q.history()[0], rtol=1e-9)
    assert np.array_equal(F.history()[2], R.history()[2]) and np.allclose(F.history()[0], R.history()[0], rtol=1e-6)
    for x, y in zip(F.state(), Q.state()):
        assert x.size == 0 or np.abs(x - y).max() < 1e-8 * max(1.0, np.abs(y).max())
    F.close(); Q.close(); R.close()


def test_fused_schur_with_fixed_cameras_and_fixed_points():
    """Fixed cameras (column -1: their W rows are zero and their blocks have no destination) and fixed points (no Schur term at
    all) inside the segments."""
    pr = synth_ba.make_problem(n_cams=30, n_points=1500, n_cuboids=0, seed=19)
    pr["cam_fixed"] = pr["cam_fixed"].copy(); pr["cam_fixed"][[0, 7, 8, 21]] = 1
    pr["pt_fixed"] = pr["pt_fixed"].copy(); pr["pt_fixed"][::5] = 1
    G, R = capi.ba_from_dict(pr), _oracle(pr)
    assert G.schur_layout()[0]
    G.compute_errors(); G.build_system(); R.build_system()
    ok_g, x_g = G.solve(5.0)
    ok_r, x_r = R.solve(5.0)
    assert ok_g and ok_r and _rel(x_g, x_r) < 1e-5
    assert G.optimize(4) == R.optimize(4)
    assert np.array_equal(G.history()[2], R.history()[2]) and np.allclose(G.history()[0], R.history()[0], rtol=1e-6)
    G.close(); R.close()


def test_rccl_path_inside_the_library_single_rank_communicator():
    """cs_ba_comm_unique_id / cs_ba_comm_init: the library creates its own RCCL communicator and issues ncclAllReduce on its stream
    (per trial: [S | b_schur], then [chi2, scale]).  One GPU here, so the communicator has one rank -- every collective call is
    executed for real and must leave the single-rank trajectory untouched.  (Two ranks cannot share a device under RCCL; the
    multi-rank arithmetic is covered by the thread / gloo tests through the callback entry.)"""
    pr = synth_ba.make_problem(n_cams=150, n_points=6000, n_cuboids=20, seed=23)
    A = capi.ba_from_dict(pr)
    n_a = A.optimize(5)
    B = capi.ba_from_dict(pr)
    B.comm_init(0, 1, capi.comm_unique_id())
    assert B.solver_layout()[0] > 0
    n_b = B.optimize_sharded(5)
    assert n_a == n_b
    assert np.array_equal(A.history()[2], B.history()[2]) and np.allclose(A.history()[0], B.history()[0], rtol=1e-12)
    for a, b in zip(A.state(), B.state()):
        assert np.array_equal(a, b)
    A.close(); B.close()


@pytest.mark.parametrize("case", ["plain", "bbox_edges", "fixed", "cuboids_first"])
def test_cuboid_elimination_equals_g2os_reduced_system(case, monkeypatch):
    """The cuboids' 9 x 9 blocks eliminated like landmarks (reduced system = cameras only) against g2o's reduced system (cameras and
    cuboids; CS_BA_KEEP_CUBOIDS=1) and against the oracle: same increments, same LM trajectory.  Cases: EdgeSE3Cuboid only; an
    EdgeSE3Cuboid and an EdgeSE3CuboidProj per (camera, cuboid) pair (two edges in one slot); fixed cuboids and fixed cameras among
    the observers; cuboids numbered before the cameras."""
    pr = synth_ba.make_problem(n_cams=150, n_points=6000, n_cuboids=24, seed=41, bbox_edges=(case == "bbox_edges"))
    cf = False
    if case == "fixed":
        pr["cub_fixed"] = pr["cub_fixed"].copy(); pr["cub_fixed"][[2, 9]] = 1
        pr["cam_fixed"] = pr["cam_fixed"].copy(); pr["cam_fixed"][[0, 40, 41, 100]] = 1
    if case == "cuboids_first":
        cf = True
    E = capi.ba_from_dict(pr, cuboids_first=cf)
    n_red, elim = E.reduced_size()
    assert elim and n_red == 6 * int((np.asarray(pr["cam_fixed"]) == 0).sum())
    monkeypatch.setenv("CS_BA_KEEP_CUBOIDS", "1")
    K = capi.ba_from_dict(pr, cuboids_first=cf)
    n_keep, elim_k = K.reduced_size()
    monkeypatch.delenv("CS_BA_KEEP_CUBOIDS")
    assert not elim_k and n_keep == E.sizes()[0] == K.sizes()[0] and n_keep > n_red
    assert n_red * E.solver_layout()[0] ** 2 < n_keep * K.solver_layout()[0] ** 2      # the cheaper banded factorisation (n bw^2) is what selects it
    R = _oracle(pr, cuboids_first=cf)
    E.build_system(); K.build_system(); R.build_system()
    for lam in (1e-2, 40.0):
        ok_e, x_e = E.solve(lam)
        ok_k, x_k = K.solve(lam)
        ok_r, x_r = R.solve(lam)
        assert ok_e and ok_k and ok_r
        assert _rel(x_e, x_k) < 1e-9 and _rel(x_e, x_r) < 1e-5
    ok_e, _ = E.solve(-1e12)                    # an indefinite damped cuboid block must surface as "not positive definite"
    assert not ok_e
    assert E.optimize(5) == K.optimize(5) == R.optimize(5)
    # (the two elimination orders agree to 1e-9 in one solve; over five iterations the 1e-9-step numeric Jacobians of the cuboid /
    # odometry edges amplify that to ~1e-7 in chi2 -- the bar stays north_star's 1e-5)
    assert np.array_equal(E.history()[2], K.history()[2]) and np.allclose(E.history()[0], K.history()[0], rtol=1e-6)
    assert np.array_equal(E.history()[2], R.history()[2]) and np.allclose(E.history()[0], R.history()[0], rtol=1e-5)
    for a, b in zip(E.state(), K.state()):
        assert a.size == 0 or np.abs(a - b).max() < 1e-5 * max(1.0, np.abs(b).max())
    E.close(); K.close(); R.close()


def test_hip_path_against_the_independent_projection_schur_fixture():
    """The same file the oracle is held to on the CPU (tests/test_ba_oracle.py, tools/make_ba_golden.py: a numpy restatement of the
    reference's projection edge, quadratic form, Schur complement and LM loop that shares nothing with oracle/ or csrc/), through
    the C ABI on the device: system blocks, one damped solve, five LM iterations, final states."""
    from test_ba_oracle import check_against_independent_fixture

    def make(g):
        P = capi.BaProblem(g["cams"], g["cam_fixed"], np.zeros((0, 10)), np.zeros(0, np.int32), g["points"], g["pt_fixed"])
        P.set_edges_proj(g["e_pt"], g["e_cam"], g["e_uv"], g["e_info"], g["e_intr"], g["e_huber"])
        return P
    check_against_independent_fixture(make, 1e-10, 1e-7)


def test_growing_graph_append_equals_rebuild_every_frame():
    """The reference's pattern (main_obj.cpp:802-803): a frame is added -- a camera, the landmarks and cuboids it sees first, its
    edges -- and optimize(5) runs on the grown graph.  cs_ba_append_* extends the handle (the estimates optimised on the device stay
    there); the result of every step must equal a handle built from scratch from the previous step's states and all edges."""
    pr = synth_ba.make_problem(n_cams=48, n_points=2400, n_cuboids=5, seed=3)
    nc = len(pr["cams"])
    first_pt = np.full(len(pr["points"]), nc); np.minimum.at(first_pt, pr["e_pt"], pr["e_cam"])
    first_cub = np.full(len(pr["cuboids"]), nc); np.minimum.at(first_cub, pr["ce_cub"], pr["ce_cam"])
    p_new = np.argsort(first_pt, kind="stable"); p_rank = np.empty_like(p_new); p_rank[p_new] = np.arange(len(p_new))
    o_new = np.argsort(first_cub, kind="stable"); o_rank = np.empty_like(o_new); o_rank[o_new] = np.arange(len(o_new))
    pts, pt_fixed, first_pt = pr["points"][p_new], pr["pt_fixed"][p_new], first_pt[p_new]
    cubs, cub_fixed, first_cub = pr["cuboids"][o_new], pr["cub_fixed"][o_new], first_cub[o_new]
    e_pt, ce_cub = p_rank[pr["e_pt"]], o_rank[pr["ce_cub"]]

    def edges_of(lo, hi):      # edges that arrive with the cameras lo .. hi - 1 (an odometry edge with its later camera)
        a = (pr["e_cam"] >= lo) & (pr["e_cam"] < hi)
        c = (pr["ce_cam"] >= lo) & (pr["ce_cam"] < hi)
        o = (np.maximum(pr["oe_i"], pr["oe_j"]) >= lo) & (np.maximum(pr["oe_i"], pr["oe_j"]) < hi)
        return a, c, o

    T0, steps = 24, 6
    a, c, o = edges_of(0, T0)
    n_p, n_o = int((first_pt < T0).sum()), int((first_cub < T0).sum())
    G = capi.BaProblem(pr["cams"][:T0], pr["cam_fixed"][:T0], cubs[:n_o], cub_fixed[:n_o], pts[:n_p], pt_fixed[:n_p])
    G.set_edges_proj(e_pt[a], pr["e_cam"][a], pr["e_uv"][a], pr["e_info"][a], pr["e_intr"][a], pr["e_huber"][a])
    G.set_edges_cuboid(pr["ce_cam"][c], ce_cub[c], pr["ce_meas"][c], pr["ce_info"][c])
    G.set_edges_odom(pr["oe_i"][o], pr["oe_j"][o], pr["oe_meas"][o], pr["oe_info"][o])
    acc = {k: [v] for k, v in dict(ep=e_pt[a], ec=pr["e_cam"][a], uv=pr["e_uv"][a], inf=pr["e_info"][a], intr=pr["e_intr"][a], hub=pr["e_huber"][a], cc=pr["ce_cam"][c], co=ce_cub[c],
                                   cm=pr["ce_meas"][c], ci=pr["ce_info"][c], oi=pr["oe_i"][o], oj=pr["oe_j"][o], om=pr["oe_meas"][o], oinf=pr["oe_info"][o]).items()}
    assert G.optimize(5) >= 1
    for t in range(T0, T0 + steps):
        cams_prev, cubs_prev, pts_prev = G.state()
        a, c, o = edges_of(t, t + 1)
        n_p2, n_o2 = int((first_pt < t + 1).sum()), int((first_cub < t + 1).sum())
        G.append_vertices(pr["cams"][t:t + 1], pr["cam_fixed"][t:t + 1], cubs[n_o:n_o2], cub_fixed[n_o:n_o2], pts[n_p:n_p2], pt_fixed[n_p:n_p2])
        G.append_edges_proj(e_pt[a], pr["e_cam"][a], pr["e_uv"][a], pr["e_info"][a], pr["e_intr"][a], pr["e_huber"][a])
        G.append_edges_cuboid(pr["ce_cam"][c], ce_cub[c], pr["ce_meas"][c], pr["ce_info"][c])
        G.append_edges_odom(pr["oe_i"][o], pr["oe_j"][o], pr["oe_meas"][o], pr["oe_info"][o])
        for k, v in dict(ep=e_pt[a], ec=pr["e_cam"][a], uv=pr["e_uv"][a], inf=pr["e_info"][a], intr=pr["e_intr"][a], hub=pr["e_huber"][a], cc=pr["ce_cam"][c], co=ce_cub[c],
                         cm=pr["ce_meas"][c], ci=pr["ce_info"][c], oi=pr["oe_i"][o], oj=pr["oe_j"][o], om=pr["oe_meas"][o], oinf=pr["oe_info"][o]).items():
            acc[k].append(v)
        n1 = G.optimize(5)
        cat = {k: np.concatenate(v) for k, v in acc.items()}
        R = capi.BaProblem(np.concatenate([cams_prev, pr["cams"][t:t + 1]]), pr["cam_fixed"][:t + 1], np.concatenate([cubs_prev, cubs[n_o:n_o2]]), cub_fixed[:n_o2],
                           np.concatenate([pts_prev, pts[n_p:n_p2]]), pt_fixed[:n_p2])
        R.set_edges_proj(cat["ep"], cat["ec"], cat["uv"], cat["inf"], cat["intr"], cat["hub"])
        R.set_edges_cuboid(cat["cc"], cat["co"], cat["cm"], cat["ci"])
        R.set_edges_odom(cat["oi"], cat["oj"], cat["om"], cat["oinf"])
        n2 = R.optimize(5)
        # (the rebuilt handle re-normalises the camera quaternions it is handed, and the cuboid / odometry Jacobians are central
        # differences with a 1e-9 step: last-bit differences of the states grow to ~1e-8 within five iterations)
        assert n1 == n2 and np.array_equal(G.history()[2], R.history()[2]) and np.allclose(G.history()[0], R.history()[0], rtol=1e-7)
        for x, y in zip(G.state(), R.state()):
            assert x.shape == y.shape and np.abs(x - y).max() <= 1e-6 * max(1.0, np.abs(y).max())
        R.close()
        n_p, n_o = n_p2, n_o2
    assert G.sizes()[0] > 0
    G.close()


def test_general_sparse_reduced_solve_on_a_covisibility_mesh():
    """A survey-flight graph (cameras on a 2-D grid looking down: synth_ba.make_mesh_problem): the cameras' covisibility graph is a mesh
    that reverse Cuthill-McKee cannot band narrowly.  The general sparse path (CS_BA_SPARSE=1: minimum-degree block ordering + symbolic
    factorisation on the host, ba_sparse.h; level-scheduled persistent Cholesky + substitution on the device, sparse_kernels.hip --
    the place of Eigen::SimplicialLDLT behind g2o's LinearSolverEigen, solvers/linear_solver_eigen.h:94-232) gives the same damped
    solve as the banded / dense paths (1e-9) and the same LM run as the oracle (north_star's 1e-5)."""
    import subprocess, sys, json, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, json, numpy as np
        sys.path.insert(0, %r)
        from cube_slam_wu_amd import capi, synth_ba
        pr = synth_ba.make_mesh_problem(16, 16, 20000)
        G = capi.ba_from_dict(pr)
        G.compute_errors(); G.build_system(dense_hpp=False)
        ok, x = G.solve(1e-3)
        n = G.optimize(6)
        chi, lam, tr = G.history()
        c, o, p = G.state()
        np.savez(sys.argv[1], x=x, chi=chi, lam=lam, tr=tr, cams=c, pts=p, ok=ok, n=n, path=np.array(G.solver_path()))
    """ % root)
    out = {}
    for mode in ("1", "0"):
        f = os.path.join(root, "build_tmp", "mesh_sparse_%s.npz" % mode)
        os.makedirs(os.path.dirname(f), exist_ok=True)
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=600, env={**os.environ, "CS_BA_SPARSE": mode})
        assert r.returncode == 0, r.stderr
        out[mode] = np.load(f)
    a, b = out["1"], out["0"]
    assert str(a["path"]) == "sparse" and str(b["path"]) in ("band", "dense")
    assert bool(a["ok"]) and bool(b["ok"]) and int(a["n"]) == int(b["n"]) == 6
    assert np.abs(a["x"] - b["x"]).max() <= 1e-9 * np.abs(b["x"]).max()
    assert np.array_equal(a["tr"], b["tr"]) and np.allclose(a["chi"], b["chi"], rtol=1e-9)
    pr = synth_ba.make_mesh_problem(16, 16, 20000)
    R = _oracle(pr)
    assert R.optimize(6) == 6
    chi_r, lam_r, tr_r = R.history()
    assert np.array_equal(a["tr"], tr_r) and np.allclose(a["chi"], chi_r, rtol=1e-6)
    cr, _, prr = R.state()
    assert np.abs(a["cams"] - cr).max() <= 1e-5 * np.abs(cr).max() and np.abs(a["pts"] - prr).max() <= 1e-5 * np.abs(prr).max()
    assert a["chi"][-1] < 0.5 * a["chi"][0]


def test_first_handle_of_a_fresh_process_on_a_small_graph():
    """The structure phase sends the edge tables to the device from a helper thread while the main thread builds the Schur schedule
    (csrc/ba_host.cpp); the device view must take the tables' addresses only after the helper is done.  On the FIRST handle of a process
    the helper's allocations are slow and a small graph's schedule is built in no time -- the order that once left null addresses in the
    view (found by __graft_entry__.smoke(), which the suite's warm process never reproduced).  A fresh interpreter per run, three runs."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from cube_slam_wu_amd import capi, synth_ba\n"
        "from oracle import ba_oracle_py as O\n"
        "pr = synth_ba.make_problem(n_cams=30, n_points=1500, n_cuboids=6, seed=11)\n"
        "G = capi.ba_from_dict(pr)\n"
        "R = O.Problem(pr['cams'], pr['cam_fixed'], pr['cuboids'], pr['cub_fixed'], pr['points'], pr['pt_fixed'])\n"
        "R.set_edges_proj(pr['e_pt'], pr['e_cam'], pr['e_uv'], pr['e_info'], pr['e_intr'], pr['e_huber'])\n"
        "R.set_edges_cuboid(pr['ce_cam'], pr['ce_cub'], pr['ce_meas'], pr['ce_info'])\n"
        "R.set_edges_odom(pr['oe_i'], pr['oe_j'], pr['oe_meas'], pr['oe_info'])\n"
        "assert G.optimize(4) == R.optimize(4)\n"
        "assert np.allclose(G.history()[0], R.history()[0], rtol=1e-6)\n"
        "print('fresh process ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _ in range(3):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "fresh process ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
