"""Differential run of the distance-map front end (Canny + 3 x 3 L2 distance transform) against its CPU restatement: random images, random
ROIs (tiny, thin, wide, whole-image), calls small and large (both hysteresis paths).  python tools/fuzz_edge.py [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi
from oracle import edge_oracle_py as E

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(77)
det = capi.Detector(capi.default_params())
bad = n_roi = 0
t0 = time.time()
for k in range(n_img):
    H, W = int(rng.integers(40, 400)), int(rng.integers(40, 1300))
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.full((H, W), float(rng.uniform(60, 160)))
    for _ in range(int(rng.integers(1, 30))):
        a = rng.uniform(0, np.pi)
        img += np.where((xx - rng.uniform(0, W)) * np.cos(a) + (yy - rng.uniform(0, H)) * np.sin(a) > 0, rng.uniform(-60, 60), 0)
    img += rng.uniform(0, 14) * np.sin(xx / rng.uniform(3, 9)) * np.cos(yy / rng.uniform(3, 9)) + rng.normal(0, rng.uniform(0, 8), (H, W))
    gray = np.clip(img, 0, 255).astype(np.uint8)
    rois = [(0, 0, W, H)]
    for _ in range(int(rng.integers(3, 14))):
        w, h = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
        rois.append((int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h))
    if k % 5 == 0:      # a large call (> 1024 ROIs): the fused hysteresis path
        rois = rois + [(int(rng.integers(0, max(1, W - 40))), int(rng.integers(0, max(1, H - 30))), min(40, W), min(30, H)) for _ in range(1100)]
    got = det.edge_distance_maps(gray, rois)
    for r, g in zip(rois[:20] + rois[-5:], got[:20] + got[-5:]):
        ref = E.edge_distance_map(gray, r)
        n_roi += 1
        if g.shape != ref.shape or not np.array_equal(g.view(np.uint32), ref.view(np.uint32)):
            bad += 1
            print("mismatch", k, (H, W), r)
print("%d images, %d ROIs compared, %d mismatches, %.0f s" % (n_img, n_roi, bad, time.time() - t0))
sys.exit(1 if bad else 0)
