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


def test_product_host_stage_equals_the_restatement_on_the_reference_frames_without_a_gpu(tmp_path):
    """The sequential half of the LSD producer (cube_slam_wu_amd/csrc/lsd_host.cpp) is plain host C++: compiled here against the planes the
    restatement computes (tools/hostonly/), it has to return the restatement's segments bit for bit on all 58 frames of the reference's
    sequence (object_slam/data/raw_imgs) -- region growing decided from the cos / sin sums, the interval form of rect_nfa's test and
    the word-wise seed scan included.  (With a GPU the same comparison runs through the C ABI in tests/test_lines_gpu.py.)"""
    import shutil, subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip_inc, hip_lib = "/opt/rocm/include", "/opt/rocm/lib"
    if shutil.which("g++") is None or not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("needs g++ and the HIP headers")
    raw = os.path.join(os.path.dirname(__file__), "golden", "object_slam_data", "raw_imgs")
    frames = sorted(f for f in os.listdir(raw) if f.endswith(".jpg"))
    assert len(frames) == 58
    blob = tmp_path / "frames.gray"
    with open(blob, "wb") as fo:
        for f in frames:
            img = np.asarray(Image.open(os.path.join(raw, f)).convert("L"))
            assert img.shape == (480, 640)
            fo.write(img.tobytes())
    exe = tmp_path / "lsd_host_check"
    o1, o2 = tmp_path / "planes.o", tmp_path / "check.o"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-c", os.path.join(root, "tools", "hostonly", "lsd_planes_from_oracle.cpp"), "-o", str(o1)])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc, "-c", os.path.join(root, "tools", "hostonly", "lsd_host_check.cpp"), "-o", str(o2)])
    subprocess.check_call(["g++", str(o2), str(o1), "-o", str(exe), "-L" + hip_lib, "-lamdhip64", "-Wl,-rpath," + hip_lib, "-pthread"])
    out = subprocess.run([str(exe), str(blob), "640", "480", "58", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "58 images" in out.stdout and " 0 differ" in out.stdout, out.stdout + out.stderr
