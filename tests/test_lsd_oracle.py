"""The LSD restatement (oracle/lsd_oracle.cpp) pinned against the reference's own saved detector output.

detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt is what the reference's LSD branch wrote for its bundled frame 0000
(detect_3d_cuboid/src/main.cpp reads it back; line_lbd/src/detect_lines.cpp writes x1 y1 x2 y2 with the stream's default six
significant digits).  The restatement, run on the gray image of the same JPEG (tools/make_c1_gray.py), has to reproduce that file
token for token: 271 segments, the same order, every coordinate equal once printed with six significant digits.  That covers the
three restated OpenCV calls (double-precision Gaussian blur, bilinear resize, fastAtan2) as well as the vendored detector."""
import os

import numpy as np
import pytest

from oracle import lsd_oracle_py as LSD

GDIR = os.path.join(os.path.dirname(__file__), "golden", "detect_3d_cuboid_data")


def _gray():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GDIR, "0000_gray.png")))


def test_lsd_restatement_reproduces_the_reference_s_saved_segments_token_for_token():
    got = LSD.detect_filter_lines(_gray(), 15.0)
    want_tokens = open(os.path.join(GDIR, "0000_edge.txt")).read().split()
    assert got.shape == (271, 4) and len(want_tokens) == 4 * 271
    mine = ["%g" % v for v in got.reshape(-1)]           # operator<<(float): %g with precision 6
    bad = [(i // 4, a, b) for i, (a, b) in enumerate(zip(mine, want_tokens)) if float(a) != float(b)]
    assert not bad, bad[:5]


def test_lsd_restatement_edge_cases():
    # a flat image has no defined gradient angle anywhere: no seeds, no segments
    assert LSD.detect_filter_lines(np.full((64, 80), 90, np.uint8)).shape == (0, 4)
    # one vertical step edge: segments along it, none hugging the border (LSDDetector's boundary filter, 10 pixels)
    img = np.zeros((120, 160), np.uint8)
    img[:, 80:] = 200
    seg = LSD.detect_filter_lines(img, 15.0)
    assert len(seg) >= 1
    assert np.all(np.abs(seg[:, [0, 2]] - 80.0) < 2.0)
    assert np.all(np.hypot(seg[:, 0] - seg[:, 2], seg[:, 1] - seg[:, 3]) > 15.0)
    # the length threshold is applied after detection: a larger one keeps a subset, in the same order
    g = _gray()
    a, b = LSD.detect_filter_lines(g, 15.0), LSD.detect_filter_lines(g, 60.0)
    keep = np.hypot(a[:, 0] - a[:, 2], a[:, 1] - a[:, 3]).astype(np.float32) > np.float32(60.0)
    assert len(b) < len(a) and np.array_equal(a[keep], b)
    with pytest.raises(RuntimeError):
        LSD.detect_filter_lines(g, 15.0, cap=10)
