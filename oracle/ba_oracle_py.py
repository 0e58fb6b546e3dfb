"""ctypes binding of oracle/liboracle_ba.so + the reference's offline 58-frame graph driver.

TEST INFRASTRUCTURE (same rule as oracle_py.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def project_bbox(cub10, Tcw7, K9):
    """cuboid::projectOntoImageBbox of the oracle (known-answer tests)."""
    out = np.zeros(4)
    lib().ba_oracle_cuboid_project_bbox(_dp(_f(cub10, (10,))), _dp(_f(Tcw7, (7,))), _dp(_f(K9, (9,))), _dp(out))
    return out


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_ba.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(path)
        L.ba_oracle_create.restype = C.c_void_p
        L.ba_oracle_compute_errors.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _f(a, shape):
    return np.ascontiguousarray(np.asarray(a, np.float64).reshape(shape))


def _i(a):
    return np.ascontiguousarray(np.asarray(a, np.int32).ravel())


class Problem:
    """A g2o graph in flat arrays (see ba_oracle.cpp: ba_oracle_set_*)."""

    def __init__(self, cams, cam_fixed, cuboids=None, cub_fixed=None, points=None, pt_fixed=None, cuboids_first=False, marginalize_points=True):
        L = lib()
        self.h = C.c_void_p(L.ba_oracle_create())
        self.cams = _f(cams, (-1, 7)); self.nc = len(self.cams)
        self.cuboids = _f(cuboids if cuboids is not None else np.zeros((0, 10)), (-1, 10)); self.no = len(self.cuboids)
        self.points = _f(points if points is not None else np.zeros((0, 3)), (-1, 3)); self.np_ = len(self.points)
        cf = _i(cam_fixed); of = _i(cub_fixed if cub_fixed is not None else np.zeros(self.no)); pf = _i(pt_fixed if pt_fixed is not None else np.zeros(self.np_))
        L.ba_oracle_set_vertices(self.h, _dp(self.cams), _ip(cf), self.nc, _dp(self.cuboids), _ip(of), self.no, _dp(self.points), _ip(pf), self.np_,
                                 int(cuboids_first), int(marginalize_points))
        self._fixed, self._flags = (cf, of, pf), (int(cuboids_first), int(marginalize_points))
        self.n_proj = self.n_cub = self.n_odom = 0

    def set_estimates(self, cams, cuboids, points):
        """New estimates for the same graph (the edges stay)."""
        self.cams, self.cuboids, self.points = _f(cams, (-1, 7)), _f(cuboids, (-1, 10)), _f(points, (-1, 3))
        cf, of, pf = self._fixed
        lib().ba_oracle_set_vertices(self.h, _dp(self.cams), _ip(cf), self.nc, _dp(self.cuboids), _ip(of), self.no, _dp(self.points), _ip(pf), self.np_, *self._flags)

    def set_edges_proj(self, pt, cam, uv, info4, intr4, huber=None):
        pt, cam = _i(pt), _i(cam); self.n_proj = len(pt)
        uv, info4, intr4 = _f(uv, (-1, 2)), _f(info4, (-1, 4)), _f(intr4, (-1, 4))
        hb = _f(huber, (-1,)) if huber is not None else None
        lib().ba_oracle_set_edges_proj(self.h, self.n_proj, _ip(pt), _ip(cam), _dp(uv), _dp(info4), _dp(intr4), _dp(hb))

    def set_edges_cuboid(self, cam, cub, meas10, info81):
        cam, cub = _i(cam), _i(cub); self.n_cub = len(cam)
        lib().ba_oracle_set_edges_cuboid(self.h, self.n_cub, _ip(cam), _ip(cub), _dp(_f(meas10, (-1, 10))), _dp(_f(info81, (-1, 81))))

    def set_edges_cuboid_proj(self, cam, cub, meas4, info16, K9):
        """EdgeSE3CuboidProj (g2o_Object.h:264-293): bbox (cx, cy, w, h) of the projected cuboid against a measured one."""
        cam, cub = _i(cam), _i(cub); self.n_cproj = len(cam)
        lib().ba_oracle_set_edges_cuboid_proj(self.h, self.n_cproj, _ip(cam), _ip(cub), _dp(_f(meas4, (-1, 4))), _dp(_f(info16, (-1, 16))), _dp(_f(K9, (-1, 9))))

    def set_edges_odom(self, ci, cj, meas7, info36):
        ci, cj = _i(ci), _i(cj); self.n_odom = len(ci)
        lib().ba_oracle_set_edges_odom(self.h, self.n_odom, _ip(ci), _ip(cj), _dp(_f(meas7, (-1, 7))), _dp(_f(info36, (-1, 36))))

    def set_robust_kernels(self, edge_class, kind, delta):
        """Robust kernels of one edge class (0 projection, 1 EdgeSE3Cuboid, 2 EdgeSE3CuboidProj, 3 EdgeSE3Expmap); kinds as in
        core/robust_kernel_impl.cpp: 1 Huber, 2 PseudoHuber, 3 Cauchy, 4 Saturated, 5 DCS, 6 Tukey; 0 none."""
        kind, delta = _i(kind), _f(delta, (-1,))
        if lib().ba_oracle_set_robust_kernels(self.h, int(edge_class), len(kind), _ip(kind), _dp(delta)) != 0:
            raise ValueError("set_robust_kernels: edge class / count mismatch")

    def optimize(self, iters):
        return lib().ba_oracle_optimize(self.h, int(iters))

    def stage_ms(self, reset=False):
        """Wall milliseconds per stage since the last reset: residuals, linearise + quadratic form, Schur, linear solve, update."""
        out = np.zeros(5)
        lib().ba_oracle_stage_ms(self.h, _dp(out), int(bool(reset)))
        return dict(zip(("errors_ms", "linearize_ms", "schur_ms", "solve_ms", "update_ms"), out.tolist()))

    def set_ldlt_stride(self, stride):
        """Timing runs only: sample every stride-th column of the dense LDL^T (no increment is computed)."""
        lib().ba_oracle_set_ldlt_stride(self.h, int(stride))

    def set_lm_params(self, user_lambda_init=0.0, max_trials_after_failure=10):
        """setUserLambdaInit / setMaxTrialsAfterFailure (optimization_algorithm_levenberg.cpp:191-199)."""
        lib().ba_oracle_set_lm_params.argtypes = [C.c_void_p, C.c_double, C.c_int]
        lib().ba_oracle_set_lm_params(self.h, float(user_lambda_init), int(max_trials_after_failure))
        return self

    def use_lapack_solver(self, one_thread=False):
        """The reduced system's dense solve through LAPACK (scipy cho_factor / cho_solve on the matrix ba_oracle.cpp assembled) instead of
        the file's textbook LDL^T -- for sizes where that one is out of reach (C4: 10 494 unknowns, 24 minutes per solve).  Linearisation,
        Schur complement, back-substitution, update and the LM control stay the C++ restatement's.  Returns the list the solve times (s)
        are appended to."""
        import time
        import scipy.linalg
        times = []

        def solve(S_ptr, n, b_ptr, x_ptr):
            S = np.ctypeslib.as_array(S_ptr, shape=(n, n))
            b = np.ctypeslib.as_array(b_ptr, shape=(n,))
            x = np.ctypeslib.as_array(x_ptr, shape=(n,))
            t0 = time.perf_counter()
            try:
                if one_thread:
                    from threadpoolctl import threadpool_limits
                    with threadpool_limits(limits=1):
                        c, low = scipy.linalg.cho_factor(S.T, lower=False, overwrite_a=True, check_finite=False)
                        x[:] = scipy.linalg.cho_solve((c, low), b, check_finite=False)
                else:
                    c, low = scipy.linalg.cho_factor(S.T, lower=False, overwrite_a=True, check_finite=False)   # (S is symmetric and stored in full: its transpose view is Fortran-ordered)
                    x[:] = scipy.linalg.cho_solve((c, low), b, check_finite=False)
            except np.linalg.LinAlgError:
                return 1
            finally:
                times.append(time.perf_counter() - t0)
            return 0 if np.isfinite(x).all() else 1

        self._dense_cb = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double))(solve)
        lib().ba_oracle_set_dense_solver(self.h, self._dense_cb)
        return times

    def history(self, cap=64):
        chi, lam, tr = np.zeros(cap), np.zeros(cap), np.zeros(cap, np.int32)
        n = lib().ba_oracle_history(self.h, _dp(chi), _dp(lam), _ip(tr), cap)
        return chi[:n], lam[:n], tr[:n]

    def state(self):
        cams, cubs, pts = np.zeros((self.nc, 7)), np.zeros((self.no, 10)), np.zeros((self.np_, 3))
        lib().ba_oracle_get_state(self.h, _dp(cams), _dp(cubs), _dp(pts))
        return cams, cubs, pts

    def compute_errors(self):
        chi = lib().ba_oracle_compute_errors(self.h)
        ep, ec, eo = np.zeros((self.n_proj, 2)), np.zeros((self.n_cub, 9)), np.zeros((self.n_odom, 6))
        lib().ba_oracle_get_errors(self.h, _dp(ep), _dp(ec), _dp(eo))
        return chi, ep, ec, eo

    def errors_cuboid_proj(self):
        """EdgeSE3CuboidProj error vectors (n x 4) of the last compute_errors()."""
        e = np.zeros((getattr(self, "n_cproj", 0), 4))
        lib().ba_oracle_get_errors_cproj(self.h, _dp(e))
        return e

    def sizes(self):
        a, b = C.c_int(), C.c_int()
        lib().ba_oracle_sizes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def build_system(self):
        lib().ba_oracle_build_system(self.h)
        n, nl = self.sizes()
        Hpp, Hll, Hpl, b = np.zeros((n, n)), np.zeros((nl // 3, 9)), np.zeros((self.n_proj, 18)), np.zeros(n + nl)
        lib().ba_oracle_get_system(self.h, _dp(Hpp), _dp(Hll), _dp(Hpl), _dp(b))
        return Hpp, Hll, Hpl, b

    def build_system_blocks(self):
        """build_system without copying the dense H_pp out (C4: 881 MB): (Hll, Hpl, b)."""
        lib().ba_oracle_build_system(self.h)
        n, nl = self.sizes()
        Hll, Hpl, b = np.zeros((nl // 3, 9)), np.zeros((self.n_proj, 18)), np.zeros(n + nl)
        lib().ba_oracle_get_system(self.h, None, _dp(Hll), _dp(Hpl), _dp(b))
        return Hll, Hpl, b

    def schur(self, lam):
        """The damped reduced system of the current linearisation (block_solver.hpp:373-439): dense S (size_pose^2, g2o's order), b_schur."""
        n, _ = self.sizes()
        S, bs = np.zeros((n, n)), np.zeros(n)
        lib().ba_oracle_schur(self.h, C.c_double(lam), _dp(S), _dp(bs))
        return S, bs

    def backsub(self, lam, xp):
        """Landmark back-substitution (block_solver.hpp:457-482) for pose increments solved elsewhere: the full x."""
        n, nl = self.sizes()
        x = np.zeros(n + nl)
        lib().ba_oracle_backsub(self.h, C.c_double(lam), _dp(_f(xp, (n,))), _dp(x))
        return x

    def solve(self, lam):
        n, nl = self.sizes()
        x = np.zeros(n + nl)
        ok = lib().ba_oracle_solve(self.h, C.c_double(lam), _dp(x))
        return bool(ok), x

    def close(self):
        if self.h:
            lib().ba_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def robustify(kind, delta, e):
    """RobustKernel::robustify of the restatement: (rho, rho', rho'') of the squared error e."""
    out = np.zeros(3)
    lib().ba_oracle_robustify(int(kind), C.c_double(delta), C.c_double(e), _dp(out))
    return out


def se3_exp(u):
    o = np.zeros(7); lib().ba_oracle_se3_exp(_dp(_f(u, (6,))), _dp(o)); return o


def se3_log(v7):
    o = np.zeros(6); lib().ba_oracle_se3_log(_dp(_f(v7, (7,))), _dp(o)); return o


def se3_mul(a, b):
    o = np.zeros(7); lib().ba_oracle_se3_mul(_dp(_f(a, (7,))), _dp(_f(b, (7,))), _dp(o)); return o


def se3_inv(a):
    o = np.zeros(7); lib().ba_oracle_se3_inv(_dp(_f(a, (7,))), _dp(o)); return o


def cuboid_from_minimal(v9):
    o = np.zeros(10); lib().ba_oracle_cuboid_from_minimal(_dp(_f(v9, (9,))), _dp(o)); return o


def cuboid_transform(c10, Twc7, to_local):
    o = np.zeros(10); lib().ba_oracle_cuboid_transform(_dp(_f(c10, (10,))), _dp(_f(Twc7, (7,))), int(to_local), _dp(o)); return o


def cuboid_to_minimal(c10):
    o = np.zeros(9); lib().ba_oracle_cuboid_to_minimal(_dp(_f(c10, (10,))), _dp(o)); return o


IDENT7 = np.array([0, 0, 0, 0, 0, 0, 1.0])


def run_offline_sequence(data_dir, make_problem=None, n_frames=None):
    """incremental_build_graph() in offline mode (object_slam/src/main_obj.cpp:479-841, 682-722).

    make_problem(cams, cam_fixed, cuboid, edges...) -> object with optimize()/state(); defaults to the oracle.
    Returns per-frame optimised camera poses Twc (n x 7) and the cuboid after every frame (n x 9 minimal).
    """
    truth = np.loadtxt(os.path.join(data_dir, "truth_cam_poses.txt"))
    init = np.loadtxt(os.path.join(data_dir, "pop_cam_poses_saved.txt"))
    obs = np.loadtxt(os.path.join(data_dir, "detect_cuboids_saved.txt"))
    N = len(truth) if n_frames is None else n_frames
    fixed_init_Twc = se3_mul(IDENT7, truth[0, 1:8])  # SE3Quat(Vector7d) normalises
    cam_Tcw = []            # optimised world-to-camera per frame
    cube = None
    cub_edges = []          # (frame, meas10, info81)
    odom_edges = []         # (i, j, meas7)
    out_cam, out_obj, iters = [], [], []
    row = 0
    for k in range(N):
        odom_val = IDENT7.copy()
        if k == 0:
            Twc = fixed_init_Twc
        else:
            prev = cam_Tcw[k - 1]
            if k > 1:
                odom_val = se3_mul(prev, se3_inv(cam_Tcw[k - 2]))
            Twc = se3_inv(se3_mul(odom_val, prev))
        has = row < len(obs) and int(obs[row, 0]) == k
        if has:
            m = obs[row]
            ground = cuboid_from_minimal([m[1], m[2], m[3], 0, 0, m[4], m[5], m[6], m[7]])
            cam_val_Twc = se3_mul(IDENT7, init[k, 1:8])
            local = cuboid_transform(ground, cam_val_Twc, True)
            quality = (1 - m[8] + 0.5) / 2
            row += 1
        if k == 0:
            cube = cuboid_transform(local, Twc, False)
        cam_Tcw.append(se3_inv(Twc))
        if has:
            inv_sigma = np.ones(9) * 2.0 * quality
            cub_edges.append((k, local, np.diag(inv_sigma * inv_sigma).ravel()))
        if k > 0:
            odom_edges.append((k - 1, k, odom_val))
        cams = np.array(cam_Tcw)
        fixed = np.zeros(len(cams), np.int32); fixed[0] = 1
        mk = make_problem or (lambda **kw: _oracle_problem(**kw))
        P = mk(cams=cams, cam_fixed=fixed, cuboid=cube, cub_edges=cub_edges, odom_edges=odom_edges)
        iters.append(P.optimize(5))
        c, o, _ = P.state()
        cam_Tcw = [c[i].copy() for i in range(len(c))]
        cube = o[0].copy()
        P.close()
        out_cam.append(se3_inv(cam_Tcw[k]))
        out_obj.append(cuboid_to_minimal(cube))
    return np.array(out_cam), np.array(out_obj), np.array(iters), np.array([se3_inv(t) for t in cam_Tcw])


def _rot_to_quat(R):
    """Eigen Quaterniond(Matrix3d) (the constructor g2o::SE3Quat(R, t) goes through), x y z w."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def _quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def run_online_sequence(data_dir, load_frame, detect, make_problem=None):
    """incremental_build_graph() in ONLINE mode (object_slam/src/main_obj.cpp:479-841 with the branch :585-680): every frame
    is detected from its image, the detection becomes the cuboid measurement, the graph grows and is optimised.

    load_frame(k) -> (frame dict without pose, gray image) or None when the frame has no 2D box (an empty yolo file: no
    detection, :619-641); detect(frame, gray, sample_roll_pitch) -> cuboid dict or None.  The camera pose handed to the detector
    is the current estimate for frame 0 and the FIRST frame's pose with roll/pitch sampling for all later frames (:624-629);
    with sampling, the measurement is re-expressed in the sampled camera frame (:660-668).  Returns the cuboid after every
    frame (n x 9 minimal) and the final camera poses Twc (n x 7).
    """
    truth = np.loadtxt(os.path.join(data_dir, "truth_cam_poses.txt"))
    fixed_init_Twc = se3_mul(IDENT7, truth[0, 1:8])
    mk = make_problem or (lambda **kw: _oracle_problem(**kw))
    cam_Tcw, cub_edges, odom_edges, out_obj = [], [], [], []
    cube = None
    n_det = 0
    for k in range(len(truth)):
        odom_val = IDENT7.copy()
        if k == 0:
            Twc = fixed_init_Twc
        else:
            prev = cam_Tcw[k - 1]
            if k > 1:
                odom_val = se3_mul(prev, se3_inv(cam_Tcw[k - 2]))
            Twc = se3_inv(se3_mul(odom_val, prev))
        has = False
        loaded = load_frame(k)
        if loaded is not None:
            fr, gray = loaded
            sample = int(k != 0)
            pose = fixed_init_Twc if sample else Twc
            T = np.eye(4)
            T[:3, :3] = _quat_to_rot(pose[3:7])
            T[:3, 3] = pose[:3]
            fr = dict(fr, T_wc=T)
            c = detect(fr, gray, sample)
            if c is not None:
                has = True
                n_det += 1
                ground = cuboid_from_minimal([c["pos"][0], c["pos"][1], c["pos"][2], 0, 0, c["rotY"], c["scale"][0], c["scale"][1], c["scale"][2]])
                local = cuboid_transform(ground, Twc, True)
                if sample:   # the camera frame the detector actually used: sampled roll / pitch at the first frame's position
                    qx, qy, qz, qw = _rot_to_quat(T[:3, :3])
                    roll = np.arctan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy)) + c["camera_roll_delta"]
                    pitch = np.arcsin(2 * (qw * qy - qz * qx)) + c["camera_pitch_delta"]
                    yaw = np.arctan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz))
                    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
                    Rn = np.array([[cp * cy, sr * sp * cy - cr * sy, cr * sp * cy + sr * sy],
                                   [cp * sy, sr * sp * sy + cr * cy, cr * sp * sy - sr * cy],
                                   [-sp, sr * cp, cr * cp]])    # euler_zyx_to_rot (matrix_utils.cpp:84-99)
                    local = cuboid_transform(ground, se3_mul(IDENT7, np.concatenate([T[:3, 3], _rot_to_quat(Rn)])), True)
                quality = (1 - c["normalized_error"] + 0.5) / 2
        if k == 0:
            cube = cuboid_transform(local, Twc, False)
        cam_Tcw.append(se3_inv(Twc))
        if has:
            inv_sigma = np.ones(9) * 2.0 * quality
            cub_edges.append((k, local, np.diag(inv_sigma * inv_sigma).ravel()))
        if k > 0:
            odom_edges.append((k - 1, k, odom_val))
        fixed = np.zeros(len(cam_Tcw), np.int32)
        fixed[0] = 1
        P = mk(cams=np.array(cam_Tcw), cam_fixed=fixed, cuboid=cube, cub_edges=cub_edges, odom_edges=odom_edges)
        P.optimize(5)
        c_state, o_state, _ = P.state()
        cam_Tcw = [c_state[i].copy() for i in range(len(c_state))]
        cube = o_state[0].copy()
        P.close()
        out_obj.append(cuboid_to_minimal(cube))
    return np.array(out_obj), np.array([se3_inv(t) for t in cam_Tcw]), n_det


def _oracle_problem(cams, cam_fixed, cuboid, cub_edges, odom_edges):
    P = Problem(cams, cam_fixed, cuboids=cuboid[None, :], cub_fixed=[0], cuboids_first=True)
    if cub_edges:
        P.set_edges_cuboid([e[0] for e in cub_edges], [0] * len(cub_edges), np.array([e[1] for e in cub_edges]), np.array([e[2] for e in cub_edges]))
    if odom_edges:
        P.set_edges_odom([e[0] for e in odom_edges], [e[1] for e in odom_edges], np.array([e[2] for e in odom_edges]),
                         np.tile(np.eye(6).ravel(), (len(odom_edges), 1)))
    return P
