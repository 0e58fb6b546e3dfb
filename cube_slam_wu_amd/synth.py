"""Synthetic KITTI-shaped workloads for the detect_cuboid path.

Frames are 1241x376 with the KITTI-00 calibration, a camera 1.65 m above a z=0 ground plane, n 2D
boxes obtained by projecting random car-sized cuboids, M line segments (60 % noisy / broken copies of
the projected cuboid edges, 40 % uniform clutter) and, per box, the exact float32 L2 distance
transform of the rasterised segments inside the box's expanded ROI -- the map the reference obtains
from cv::Canny + cv::distanceTransform (detect_3d_cuboid/src/box_proposal_detail.cpp:320-327) and
which is an *input* of the scorer.

This module is workload generation for tests and bench.py; it does not touch the oracle.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

IMG_W, IMG_H = 1241, 376
KITTI_K = np.array([[718.856, 0.0, 607.19], [0.0, 718.856, 185.22], [0.0, 0.0, 1.0]])


def box_rois(box, img_w, img_h, sample_height=False):
    """ROI (left, top, width, height) and down-expansion of every height sample of one 2D box.

    Mirrors box_proposal_detail.cpp:143-172, 204-205, 242-248 (integer arithmetic).
    """
    left, top, w, h = int(box[0]), int(box[1]), int(box[2]), int(box[3])
    right = int(left + box[2])
    downs = [0]
    if sample_height:
        r = max(min(20, h - 90), 20)
        r = min(r, img_h - top - h - 1)
        if r > 10:
            downs.append(int(round(r // 2)))
        downs.append(r)
    out = []
    for d in downs:
        he = h + d
        dye = top + he
        e = min(max(min(20, w - 100), 10), max(min(20, he - 100), 10))
        l, r_ = max(0, left - e), min(img_w - 1, right + e)
        t, b = max(0, top - e), min(img_h - 1, dye + e)
        out.append(((l, t, r_ - l, b - t), d))
    return out


def _camera(rng):
    a = np.deg2rad(rng.uniform(-20, 20))          # heading about world z
    t = np.deg2rad(rng.uniform(1.5, 4.0))         # tilt down
    roll = np.deg2rad(rng.uniform(-0.6, 0.6))
    f = np.array([np.sin(a) * np.cos(t), np.cos(a) * np.cos(t), -np.sin(t)])
    r0 = np.array([np.cos(a), -np.sin(a), 0.0])
    d0 = np.cross(f, r0)
    r = np.cos(roll) * r0 + np.sin(roll) * d0
    d = np.cross(f, r)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2] = r, d, f
    T[:3, 3] = [rng.uniform(-1, 1), rng.uniform(-1, 1), 1.65]
    return T


def _cuboid_corners(center, yaw, half):
    body = np.array([[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [-1, -1, -1, -1, 1, 1, 1, 1]], float)
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    return (R @ (body * half[:, None])) + center[:, None]


_EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def _rasterise(segs, l, t, w, h):
    img = np.zeros((h, w), bool)
    for x1, y1, x2, y2 in segs:
        n = int(max(abs(x2 - x1), abs(y2 - y1))) + 2
        xs = np.rint(np.linspace(x1, x2, n)).astype(int) - l
        ys = np.rint(np.linspace(y1, y2, n)).astype(int) - t
        ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
        img[ys[ok], xs[ok]] = True
    return img


def make_frame(seed, n_boxes=8, n_lines=400, img_w=IMG_W, img_h=IMG_H, sample_height=False):
    """One synthetic frame.  Returns a dict with K (3x3), T_wc (4x4), boxes (n x 5), lines (M x 4),
    rois (per box list of ((l,t,w,h), down)), maps (per box list of float32 (h*w + w + 1,) arrays:
    the h x w distance map followed by w+1 zero floats, see include/cubeslam_hip.h)."""
    rng = np.random.default_rng(seed)
    K = KITTI_K.copy()
    T = _camera(rng)
    Tcw = np.linalg.inv(T)
    P = K @ Tcw[:3]
    boxes, edge_segs = [], []
    tries = 0
    while len(boxes) < n_boxes and tries < 20000:
        tries += 1
        dist, lat = rng.uniform(5, 32), rng.uniform(-9, 9)
        f_xy = T[:2, 2] / np.linalg.norm(T[:2, 2])
        r_xy = np.array([f_xy[1], -f_xy[0]])
        cxy = T[:2, 3] + dist * f_xy + lat * r_xy
        # full height < camera height (1.65 m): the top face must be visible, as the cuboid model assumes
        half = np.array([rng.uniform(1.5, 2.4), rng.uniform(0.7, 1.0), rng.uniform(0.5, 0.76)])
        c3 = _cuboid_corners(np.array([cxy[0], cxy[1], half[2]]), rng.uniform(-np.pi, np.pi), half)
        ph = P @ np.vstack([c3, np.ones((1, 8))])
        if (ph[2] <= 0.5).any():
            continue
        uv = ph[:2] / ph[2]
        x0, y0, x1, y1 = uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()
        bw, bh = x1 - x0, y1 - y0
        if not (80 <= bw <= 300 and 60 <= bh <= 200):
            continue
        if x0 < 25 or y0 < 25 or x1 > img_w - 26 or y1 > img_h - 26:
            continue
        boxes.append([np.floor(x0), np.floor(y0), np.floor(bw), np.floor(bh), rng.uniform(0.5, 1.0)])
        edge_segs.append([(uv[0, a], uv[1, a], uv[0, b], uv[1, b]) for a, b in _EDGES])
    if len(boxes) < n_boxes:
        raise RuntimeError("could not place boxes")
    boxes = np.array(boxes, float)

    n_obj = int(round(0.6 * n_lines))
    segs = []
    while len(segs) < n_obj:
        e = edge_segs[rng.integers(len(edge_segs))][rng.integers(12)]
        p1, p2 = np.array(e[:2]), np.array(e[2:])
        if np.linalg.norm(p2 - p1) < 12:
            continue
        if rng.random() < 0.45:  # broken into two pieces with a small gap (exercises merge_break_lines)
            s = rng.uniform(0.3, 0.7)
            g = rng.uniform(2, 9) / np.linalg.norm(p2 - p1)
            pieces = [(p1, p1 + (s - g / 2) * (p2 - p1)), (p1 + (s + g / 2) * (p2 - p1), p2)]
        else:
            a, b = sorted(rng.uniform(0, 1, 2))
            if b - a < 0.4:
                a, b = 0.0, 1.0
            pieces = [(p1 + a * (p2 - p1), p1 + b * (p2 - p1))]
        for q1, q2 in pieces:
            q1 = q1 + rng.normal(0, 1.0, 2)
            q2 = q2 + rng.normal(0, 1.0, 2)
            segs.append([q1[0], q1[1], q2[0], q2[1]])
    segs = segs[:n_obj]
    while len(segs) < n_lines:
        ln, ang = rng.uniform(15, 120), rng.uniform(0, np.pi)
        x, y = rng.uniform(5, img_w - 6), rng.uniform(5, img_h - 6)
        x2, y2 = x + ln * np.cos(ang), y + ln * np.sin(ang)
        if 0 <= x2 <= img_w - 1 and 0 <= y2 <= img_h - 1:
            segs.append([x, y, x2, y2])
    lines = np.array(segs, float)
    lines = np.clip(lines, 0, [img_w - 1, img_h - 1, img_w - 1, img_h - 1])
    lines = lines[rng.permutation(len(lines))]

    rois, maps = [], []
    for b in boxes:
        rr = box_rois(b, img_w, img_h, sample_height)
        rois.append(rr)
        mm = []
        for (l, t, w, h), _ in rr:
            edge = _rasterise(lines, l, t, w, h)
            if not edge.any():
                dt = np.full((h, w), 1e3, np.float32)
            else:
                dt = ndimage.distance_transform_edt(~edge).astype(np.float32)
            buf = np.zeros(h * w + w + 1, np.float32)
            buf[: h * w] = dt.ravel()
            mm.append(buf)
        maps.append(mm)
    return dict(K=K, T_wc=T, boxes=boxes, lines=lines, rois=rois, maps=maps, img_w=img_w, img_h=img_h, seed=seed)
