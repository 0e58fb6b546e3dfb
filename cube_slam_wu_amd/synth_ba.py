"""Synthetic KITTI-00-shaped bundle-adjustment problems.

Cameras every 0.8 m along a gently curving planar path (1.65 m above the ground, looking forward), points in
a corridor around the path each observed by up to 5 consecutive cameras (pixel noise N(0,1), information I2,
optional Huber delta = sqrt(5.991)), cuboids on the ground each observed from 20 consecutive cameras
(measurement = true camera-frame cuboid perturbed by N(0, 0.05), information diag((2q)^2), q in [0.5, 1],
object_slam/src/main_obj.cpp:732,775-780), odometry edges between consecutive cameras (information I6,
main_obj.cpp:794-798).  Initial estimates = truth perturbed by N(0, 0.02 rad / 0.1 m) (cameras, cuboids) and
N(0, 0.1 m) (points).  Camera 0 is fixed; points are marginalised.

Workload generation only; does not touch the oracle.
"""
from __future__ import annotations

import numpy as np

FX = FY = 718.856
CX, CY = 607.19, 185.22
IMG_W, IMG_H = 1241, 376


def quat_mul(a, b):
    """Hamilton product, quaternions as (..., 4) arrays in x y z w order."""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_rot(q, v):
    qv = q[..., :3]
    uv = 2 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def quat_from_R(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()
    q = np.where(q[..., 3:4] < 0, -q, q)
    return q


def pose_mul(a, b):
    """SE3 compose of 7-vectors x y z qx qy qz qw."""
    t = a[..., :3] + quat_rot(a[..., 3:], b[..., :3])
    q = quat_mul(a[..., 3:], b[..., 3:])
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    q = np.where(q[..., 3:4] < 0, -q, q)
    return np.concatenate([t, q], -1)


def pose_inv(a):
    q = a[..., 3:] * np.array([-1, -1, -1, 1.0])
    t = -quat_rot(q, a[..., :3])
    return np.concatenate([t, q], -1)


def small_pose(rng, n, rot_sigma, trans_sigma):
    w = rng.normal(0, rot_sigma, (n, 3))
    th = np.linalg.norm(w, axis=1, keepdims=True)
    q = np.concatenate([np.where(th > 0, np.sin(th / 2) / np.maximum(th, 1e-300), 0.5) * w, np.cos(th / 2)], 1)
    return np.concatenate([rng.normal(0, trans_sigma, (n, 3)), q], 1)


def quat_to_R(q):
    """(x, y, z, w) -> 3x3 rotation."""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


CORNERS_BODY = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], float)  # g2o_Object.h:169-171


def project_bbox(T_cw7, cub10, K):
    """cuboid::projectOntoImageBbox (g2o_Object.h:181-197) in numpy: (centre x, centre y, width, height)."""
    Xw = quat_to_R(cub10[3:7]) @ (CORNERS_BODY * cub10[7:10, None]) + cub10[:3, None]
    Xc = quat_to_R(T_cw7[3:7]) @ Xw + T_cw7[:3, None]
    p = K @ Xc
    u, v = p[0] / p[2], p[1] / p[2]
    return np.array([(u.max() + u.min()) / 2, (v.max() + v.min()) / 2, u.max() - u.min(), v.max() - v.min()])


def make_problem(n_cams=200, n_points=20000, n_cuboids=50, seed=42, huber=True, obs_per_point=5, obs_per_cuboid=20, bbox_edges=False, loop=False):
    """loop=True: the path is a closed circle -- the last cameras see the landmarks (and cuboids) of the first ones and an odometry
    edge joins camera n-1 to camera 0: a loop closure.  The block graph of the reduced system is then a ring, not a chain."""
    rng = np.random.default_rng(seed)
    # ---- trajectory: arc of radius 400 m (loop: the full circle the cameras fill), camera z forward along the tangent, y down, x right
    s = 0.8 * np.arange(n_cams)
    Rad = 0.8 * n_cams / (2 * np.pi) if loop else 400.0
    ang = s / Rad
    pos = np.stack([Rad * np.sin(ang), Rad * (1 - np.cos(ang)), np.full(n_cams, 1.65)], 1)
    fwd = np.stack([np.cos(ang), np.sin(ang), np.zeros(n_cams)], 1)
    down = np.tile(np.array([0, 0, -1.0]), (n_cams, 1))
    right = np.cross(down, fwd)
    R_wc = np.stack([right, down, fwd], 2)  # columns = camera axes in the world
    q_wc = quat_from_R(R_wc)
    T_wc = np.concatenate([pos, q_wc], 1)
    T_cw_true = pose_inv(T_wc)

    # ---- points: pick an anchor camera, sample in front of it, observe from the consecutive cameras behind it
    pts = np.zeros((n_points, 3))
    e_pt, e_cam, e_uv = [], [], []
    filled = 0
    while filled < n_points:
        m = int((n_points - filled) * 1.3) + 16
        anchor = rng.integers(0, n_cams, m)
        pc = np.stack([rng.uniform(-10, 10, m), rng.uniform(-2.3, 1.6, m), rng.uniform(8, 40, m)], 1)  # camera frame
        pw = quat_rot(q_wc[anchor], pc) + pos[anchor]
        ok_count = np.zeros(m, int)
        obs = []
        for d in range(obs_per_point):
            cam = anchor - d
            if loop:
                cam = cam % n_cams
            valid = cam >= 0
            camc = np.clip(cam, 0, n_cams - 1)
            p = quat_rot(T_cw_true[camc, 3:], pw) + T_cw_true[camc, :3]
            z = p[:, 2]
            u = FX * p[:, 0] / np.where(z > 0.1, z, 1) + CX
            v = FY * p[:, 1] / np.where(z > 0.1, z, 1) + CY
            valid &= (z > 0.5) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
            obs.append((camc, u, v, valid))
            ok_count += valid
        keep = np.nonzero(ok_count >= 2)[0][: n_points - filled]
        for camc, u, v, valid in obs:
            kk = keep[valid[keep]]
            e_pt.append(filled + np.searchsorted(keep, kk))
            e_cam.append(camc[kk])
            e_uv.append(np.stack([u[kk], v[kk]], 1))
        pts[filled:filled + len(keep)] = pw[keep]
        filled += len(keep)
    e_pt = np.concatenate(e_pt).astype(np.int32)
    e_cam = np.concatenate(e_cam).astype(np.int32)
    e_uv = np.concatenate(e_uv) + rng.normal(0, 1.0, (len(e_pt), 2))
    n_e = len(e_pt)
    perm = rng.permutation(n_e)  # graph insertion order is not sorted
    e_pt, e_cam, e_uv = e_pt[perm], e_cam[perm], e_uv[perm]
    info4 = np.tile(np.eye(2).ravel(), (n_e, 1))
    intr4 = np.tile(np.array([FX, FY, CX, CY]), (n_e, 1))
    hub = np.full(n_e, np.sqrt(5.991) if huber else 0.0)

    # ---- cuboids on the ground next to the path
    cub_true = np.zeros((n_cuboids, 10))
    ce_cam, ce_cub, ce_meas, ce_info = [], [], [], []
    for o in range(n_cuboids):
        c0 = int(rng.integers(0, n_cams if loop else max(1, n_cams - obs_per_cuboid)))
        mid = (c0 + obs_per_cuboid // 2) % n_cams if loop else min(n_cams - 1, c0 + obs_per_cuboid // 2)
        side = rng.choice([-1.0, 1.0]) * rng.uniform(3, 7)
        ahead = rng.uniform(10, 18)
        half = np.array([rng.uniform(1.5, 2.4), rng.uniform(0.7, 1.0), rng.uniform(0.5, 0.76)])
        center = pos[mid] + ahead * fwd[mid] + side * right[mid]
        center[2] = half[2]
        yaw = rng.uniform(-np.pi, np.pi)
        cub_true[o, :3] = center
        cub_true[o, 3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
        if cub_true[o, 6] < 0:
            cub_true[o, 3:7] *= -1
        cub_true[o, 7:] = half
        for c in ([cc % n_cams for cc in range(c0, c0 + obs_per_cuboid)] if loop else range(c0, min(n_cams, c0 + obs_per_cuboid))):
            local = pose_mul(T_cw_true[c], cub_true[o, :7])
            noisy = pose_mul(local, small_pose(rng, 1, 0.05, 0.05)[0])
            q = rng.uniform(0.5, 1.0)
            ce_cam.append(c); ce_cub.append(o)
            ce_meas.append(np.concatenate([noisy, half + rng.normal(0, 0.05, 3)]))
            ce_info.append(np.diag(np.full(9, (2 * q) ** 2)).ravel())
    # ---- odometry
    oe_i = np.arange(n_cams - 1, dtype=np.int32)
    oe_j = oe_i + 1
    if loop:
        oe_i = np.append(oe_i, np.int32(n_cams - 1)); oe_j = np.append(oe_j, np.int32(0))
    oe_meas = pose_mul(T_cw_true[oe_j], pose_inv(T_cw_true[oe_i]))   # error = log(C * T_i * T_j^-1) = 0 for C = T_j T_i^-1
    oe_meas = pose_mul(small_pose(rng, len(oe_i), 0.002, 0.01), oe_meas)
    oe_info = np.tile(np.eye(6).ravel(), (len(oe_i), 1))

    # ---- initial estimates
    cams0 = pose_mul(small_pose(rng, n_cams, 0.02, 0.1), T_cw_true)
    cams0[0] = T_cw_true[0]
    cub0 = cub_true.copy()
    if n_cuboids:
        cub0[:, :7] = pose_mul(cub_true[:, :7], small_pose(rng, n_cuboids, 0.02, 0.1))
        cub0[:, 7:] += rng.normal(0, 0.05, (n_cuboids, 3))
    pts0 = pts + rng.normal(0, 0.1, pts.shape)
    cam_fixed = np.zeros(n_cams, np.int32); cam_fixed[0] = 1
    # ---- optional EdgeSE3CuboidProj edges: the 2D detection box of every cuboid observation (own random stream, so that
    # the rest of the problem does not depend on the switch)
    pe_cam, pe_cub, pe_meas = [], [], []
    Kmat = np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1.0]])
    if bbox_edges:
        rng2 = np.random.default_rng(seed + 7919)
        for c, o in zip(ce_cam, ce_cub):
            pe_cam.append(c); pe_cub.append(o)
            pe_meas.append(project_bbox(T_cw_true[c], cub_true[o], Kmat) + rng2.normal(0, 2.0, 4))
    n_pe = len(pe_cam)
    return dict(
        pe_cam=np.array(pe_cam, np.int32), pe_cub=np.array(pe_cub, np.int32), pe_meas=np.array(pe_meas).reshape(-1, 4),
        pe_info=np.tile((np.eye(4) * 0.25).ravel(), (n_pe, 1)), pe_K=np.tile(Kmat.ravel(), (n_pe, 1)),
        cams=cams0, cam_fixed=cam_fixed, cuboids=cub0, cub_fixed=np.zeros(n_cuboids, np.int32), points=pts0,
        pt_fixed=np.zeros(n_points, np.int32),
        e_pt=e_pt, e_cam=e_cam, e_uv=e_uv, e_info=info4, e_intr=intr4, e_huber=hub,
        ce_cam=np.array(ce_cam, np.int32), ce_cub=np.array(ce_cub, np.int32),
        ce_meas=np.array(ce_meas).reshape(-1, 10), ce_info=np.array(ce_info).reshape(-1, 81),
        oe_i=oe_i, oe_j=oe_j, oe_meas=oe_meas, oe_info=oe_info,
        truth=dict(cams=T_cw_true, cuboids=cub_true, points=pts))


def make_mesh_problem(nx=16, ny=16, n_points=20000, seed=7, spacing=(7.0, 2.5), height=12.0, huber=True):
    """A survey flight instead of a street: nx x ny cameras on a grid, all looking straight down at the ground from `height`, every
    landmark seen by the cameras whose image it falls into -- the covisibility graph of the cameras is a 2-D MESH (each camera shares
    landmarks with its neighbours in both directions), which no ordering turns into a narrow band: the case of a general sparse
    reduced solve (the reference: Eigen::SimplicialLDLT behind g2o's LinearSolverEigen, solvers/linear_solver_eigen.h:94-232).
    Odometry edges follow the flight lines (boustrophedon).  Same dictionary as make_problem, without cuboids."""
    rng = np.random.default_rng(seed)
    n_cams = nx * ny
    gx, gy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    order = []
    for i in range(nx):                      # boustrophedon: the camera index follows the flight path
        col = [(i, j) for j in range(ny)]
        order += col if i % 2 == 0 else col[::-1]
    order = np.array(order)
    sx, sy = spacing
    pos = np.stack([order[:, 0] * sx, order[:, 1] * sy, np.full(n_cams, height)], 1).astype(float)
    pos[:, :2] += rng.normal(0, 0.3, (n_cams, 2))
    # camera axes in the world: x right = +x, y (image down) = -y, z forward = -z (down)
    R_wc = np.tile(np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]]), (n_cams, 1, 1))
    q_wc = quat_from_R(R_wc)
    T_wc = np.concatenate([pos, q_wc], 1)
    T_cw_true = pose_inv(T_wc)
    half_w, half_h = height * (IMG_W / 2) / FX, height * (IMG_H / 2) / FY
    pts = np.stack([rng.uniform(-half_w, (nx - 1) * sx + half_w, n_points), rng.uniform(-half_h, (ny - 1) * sy + half_h, n_points), rng.uniform(0.0, 1.0, n_points)], 1)
    e_pt, e_cam, e_uv = [], [], []
    seen = np.zeros(n_points, int)
    for c in range(n_cams):
        p = quat_rot(T_cw_true[c, 3:], pts) + T_cw_true[c, :3]
        z = p[:, 2]
        u = FX * p[:, 0] / np.where(z > 0.1, z, 1) + CX
        v = FY * p[:, 1] / np.where(z > 0.1, z, 1) + CY
        ok = np.nonzero((z > 0.5) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H))[0]
        e_pt.append(ok); e_cam.append(np.full(len(ok), c)); e_uv.append(np.stack([u[ok], v[ok]], 1))
        seen[ok] += 1
    e_pt = np.concatenate(e_pt); e_cam = np.concatenate(e_cam); e_uv = np.concatenate(e_uv)
    keep_pt = seen >= 2                       # a landmark needs two views
    remap = np.cumsum(keep_pt) - 1
    sel = keep_pt[e_pt]
    e_pt, e_cam, e_uv = remap[e_pt[sel]].astype(np.int32), e_cam[sel].astype(np.int32), e_uv[sel] + rng.normal(0, 1.0, (int(sel.sum()), 2))
    pts = pts[keep_pt]
    n_points = len(pts)
    n_e = len(e_pt)
    perm = rng.permutation(n_e)
    e_pt, e_cam, e_uv = e_pt[perm], e_cam[perm], e_uv[perm]
    oe_i = np.arange(n_cams - 1, dtype=np.int32); oe_j = oe_i + 1
    oe_meas = pose_mul(T_cw_true[oe_j], pose_inv(T_cw_true[oe_i]))
    oe_meas = pose_mul(small_pose(rng, len(oe_i), 0.002, 0.01), oe_meas)
    cams0 = pose_mul(small_pose(rng, n_cams, 0.01, 0.05), T_cw_true)
    cams0[0] = T_cw_true[0]
    cam_fixed = np.zeros(n_cams, np.int32); cam_fixed[0] = 1
    Kmat = np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1.0]])
    return dict(
        pe_cam=np.zeros(0, np.int32), pe_cub=np.zeros(0, np.int32), pe_meas=np.zeros((0, 4)), pe_info=np.zeros((0, 16)), pe_K=np.tile(Kmat.ravel(), (0, 1)),
        cams=cams0, cam_fixed=cam_fixed, cuboids=np.zeros((0, 10)), cub_fixed=np.zeros(0, np.int32), points=pts + rng.normal(0, 0.05, pts.shape),
        pt_fixed=np.zeros(n_points, np.int32),
        e_pt=e_pt, e_cam=e_cam, e_uv=e_uv, e_info=np.tile(np.eye(2).ravel(), (n_e, 1)), e_intr=np.tile(np.array([FX, FY, CX, CY]), (n_e, 1)),
        e_huber=np.full(n_e, np.sqrt(5.991) if huber else 0.0),
        ce_cam=np.zeros(0, np.int32), ce_cub=np.zeros(0, np.int32), ce_meas=np.zeros((0, 10)), ce_info=np.zeros((0, 81)),
        oe_i=oe_i, oe_j=oe_j, oe_meas=oe_meas, oe_info=np.tile(np.eye(6).ravel(), (len(oe_i), 1)),
        truth=dict(cams=T_cw_true, cuboids=np.zeros((0, 10)), points=pts))
