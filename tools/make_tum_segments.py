#!/usr/bin/env python3
"""Line segments for the reference's bundled TUM cabinet frames (object_slam/data/raw_imgs), written to
tests/golden/object_slam_data/segments/NNNN.txt (x1 y1 x2 y2 per row).

The reference gets its segments from line_lbd (EDLines through OpenCV), which is out of scope and
not installed here.  tests/test_reference_frames.py only needs *plausible* segments to run detect_cuboid on the reference's
own images and compare with the detections the reference saved (detect_cuboids_saved.txt), so this script uses a plain
edge-chain splitter: Canny (the repository's restatement, thresholds 40 / 100, whole image) -> 8-connected pixel chains ->
recursive split at the point of largest deviation (1.5 px) -> total-least-squares fit, segments of 15 px and more
(line_lbd's line_length_thres in main_obj.cpp:505).  Run in the build container:  python tools/make_tum_segments.py

How much the comparison depends on these choices (measured once, first choice kept): with Canny thresholds 30/90, 50/150 or
80/200 instead of 40/100 the online run's object poses stay within 0.6 mm of the reference's saved ones in all four cases
(camera positions: mean 2.8 / 3.0 / 3.9 / 6.1 cm); splitting chains at 1 px deviation instead of 1.5 px fragments the
segments enough to change the first frame's detection, and with it the object, by 9 cm.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edge_oracle_py as E  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "object_slam_data")


def segments_from_edges(edges, min_len=15.0, tol=1.5):
    H, W = edges.shape
    e = edges > 0
    visited = np.zeros_like(e)
    nbrs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    chains = []

    def walk(y, x):
        out = []
        while True:
            nxt = None
            for dy, dx in nbrs:
                yy, xx = y + dy, x + dx
                if 0 <= yy < H and 0 <= xx < W and e[yy, xx] and not visited[yy, xx]:
                    nxt = (yy, xx)
                    break
            if nxt is None:
                return out
            visited[nxt] = True
            out.append(nxt)
            y, x = nxt

    for y0, x0 in zip(*np.nonzero(e)):
        if visited[y0, x0]:
            continue
        visited[y0, x0] = True
        a = walk(y0, x0)
        b = walk(y0, x0)
        chain = a[::-1] + [(y0, x0)] + b
        if len(chain) >= min_len:
            chains.append(np.array(chain, float)[:, ::-1])  # (x, y)
    segs = []

    def split(P):
        if len(P) < 2:
            return
        a, b = P[0], P[-1]
        d = b - a
        L = np.hypot(*d)
        if L < 1e-9:
            return
        n = np.array([-d[1], d[0]]) / L
        dist = np.abs((P - a) @ n)
        i = int(np.argmax(dist))
        if dist[i] > tol and 2 < i < len(P) - 3:
            split(P[: i + 1])
            split(P[i:])
        elif L >= min_len:
            c = P.mean(0)
            dirv = np.linalg.svd(P - c)[2][0]
            t = (P - c) @ dirv
            segs.append(np.concatenate([c + dirv * t.min(), c + dirv * t.max()]))

    for P in chains:
        split(P)
    return np.array(segs) if segs else np.zeros((0, 4))


def main():
    os.makedirs(os.path.join(DATA, "segments"), exist_ok=True)
    for k in range(58):
        img = np.asarray(Image.open(os.path.join(DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
        gray = E.bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1]))
        H, W = gray.shape
        segs = segments_from_edges(E.canny_roi(gray, (0, 0, W, H), 40, 100))
        np.savetxt(os.path.join(DATA, "segments", "%04d.txt" % k), segs, fmt="%.3f")
        print(k, len(segs), flush=True)


if __name__ == "__main__":
    main()
