#!/usr/bin/env python3
"""tests/golden/detect_3d_cuboid_data/0000_gray.png from the reference's bundled frame: JPEG decode (PIL / libjpeg, what
cv::imread uses too) followed by OpenCV's BGR2GRAY fixed-point weights (B 1868, G 9617, R 4899, >> 14).  Run in the
build container only (it reads /root/reference); the PNG is the committed fixture."""
import os

import numpy as np
from PIL import Image

src = "/root/reference/detect_3d_cuboid/data/0000_rgb_raw.jpg"
dst = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "detect_3d_cuboid_data", "0000_gray.png")
a = np.asarray(Image.open(src).convert("RGB")).astype(np.int64)
gray = ((a[..., 2] * 1868 + a[..., 1] * 9617 + a[..., 0] * 4899 + 8192) >> 14).astype(np.uint8)
Image.fromarray(gray).save(dst, optimize=True)
print(dst, gray.shape)
