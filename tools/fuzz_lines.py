"""Differential run of both branches of the segment producer (EDLines, LSD) against their CPU restatements on random images of random
sizes: same number of segments, same order, identical float coordinates.  python tools/fuzz_lines.py [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi
from oracle import edlines_oracle_py as ED
from oracle import lsd_oracle_py as LSD

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
det = capi.Detector(capi.default_params())
bad = seg = 0
t0 = time.time()
for k in range(n_img):
    h, w = int(rng.integers(20, 420)), int(rng.integers(20, 700))
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), float(rng.uniform(60, 160)))
    for _ in range(int(rng.integers(0, 25))):
        a = rng.uniform(0, np.pi)
        img += np.where((xx - rng.uniform(0, w)) * np.cos(a) + (yy - rng.uniform(0, h)) * np.sin(a) > 0, rng.uniform(-70, 70), 0)
    if rng.uniform() < 0.5:
        img += rng.uniform(2, 15) * np.sin(xx / rng.uniform(3, 12)) * np.cos(yy / rng.uniform(3, 12))
    img += rng.normal(0, rng.uniform(0, 8), (h, w))
    gray = np.clip(img, 0, 255).astype(np.uint8)
    thr = float(rng.choice([15.0, 30.0, 50.0]))
    for name, use_lsd, oracle in (("edlines", False, ED), ("lsd", True, LSD)):
        got = det.detect_lines(gray, thr, use_lsd=use_lsd)
        ref = oracle.detect_filter_lines(gray, thr)
        seg += len(ref)
        if got.shape != ref.shape or not np.array_equal(got, ref):
            bad += 1
            print("mismatch", name, k, (h, w), thr, got.shape, ref.shape)
print("%d images x 2 detectors, %d segments, %d mismatches, %.0f s" % (n_img, seg, bad, time.time() - t0))
sys.exit(1 if bad else 0)
