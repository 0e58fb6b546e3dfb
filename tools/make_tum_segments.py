#!/usr/bin/env python3
"""Line segments of the reference's bundled TUM cabinet frames (object_slam/data/raw_imgs), written to
tests/golden/object_slam_data/segments/NNNN.txt (x1 y1 x2 y2 per row, the float32 values as exact doubles).

They are what line_lbd_detect::detect_filter_lines returns for each JPEG in the reference's graph driver
(object_slam/src/main_obj.cpp:502-505,593: EDLines, one octave, line_length_thres 15), computed by the repository's restatement of
that detector (oracle/edlines_oracle.cpp).  Two things pin the restatement: the nine segments main_obj.cpp prints in a comment
(:601-611) are frame 17's segments 1-9 to all six printed digits (tests/test_reference_frames.py), and with these segments the
whole online pipeline reproduces the reference's saved output_obj_poses.txt / output_cam_poses.txt to the files' precision.
The files are fixtures for the C++ graph driver's --online mode and for the tests that do not run the line detector themselves.

Run in the build container:  python tools/make_tum_segments.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import edge_oracle_py as E  # noqa: E402
from oracle import edlines_oracle_py as L  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "object_slam_data")


def main():
    out_dir = os.path.join(DATA, "segments")
    os.makedirs(out_dir, exist_ok=True)
    for k in range(58):
        img = np.asarray(Image.open(os.path.join(DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
        gray = E.bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1]))
        seg = L.detect_filter_lines(gray, 15.0)
        with open(os.path.join(out_dir, "%04d.txt" % k), "w") as f:
            for r in seg:
                f.write(" ".join("%.17g" % float(v) for v in r) + "\n")     # the exact double of the float32, as the reference's float -> double copy (main_obj.cpp:596-599)
        print(k, len(seg))


if __name__ == "__main__":
    main()
