"""The sequential halves of the two segment producers are plain host C++ (cube_slam_wu_amd/csrc/lines_host.cpp, lsd_host.cpp): built
here WITHOUT a GPU against inputs the CPU restatements compute (tools/hostonly/), they have to return the restatements' segments bit
for bit.  With a GPU the same comparisons run through the C ABI, device stages included, in tests/test_lines_gpu.py; the LSD half of
this file lives beside its restatement's own pins in tests/test_lsd_oracle.py."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_INC, HIP_LIB = "/opt/rocm/include", "/opt/rocm/lib"


def _frames_blob(tmp_path):
    from PIL import Image
    raw = os.path.join(os.path.dirname(__file__), "golden", "object_slam_data", "raw_imgs")
    frames = sorted(f for f in os.listdir(raw) if f.endswith(".jpg"))
    assert len(frames) == 58
    blob = tmp_path / "frames.gray"
    with open(blob, "wb") as fo:
        for f in frames:
            img = np.asarray(Image.open(os.path.join(raw, f)).convert("L"))
            assert img.shape == (480, 640)
            fo.write(img.tobytes())
    return blob


def test_edlines_host_stage_equals_the_restatement_on_the_reference_frames_without_a_gpu(tmp_path):
    """EDLines (line_lbd/libs/binary_descriptor.cpp:1583-2905): smart routing, line fitting and validation of lines_host.cpp on the packed
    map (dx | (2 dy + anchor) << 16 per pixel; the gradient and direction maps evaluated from dx / dy where the routing reads them), the map
    itself from the restatement's blur / Sobel / anchor code: 58 frames of the reference's sequence, 945 segments, all equal."""
    if shutil.which("g++") is None or not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime.h")):
        pytest.skip("needs g++ and the HIP headers")
    blob = _frames_blob(tmp_path)
    o1, o2, exe = tmp_path / "map.o", tmp_path / "check.o", tmp_path / "lines_host_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-c", os.path.join(ROOT, "tools", "hostonly", "lines_map_from_oracle.cpp"), "-o", str(o1)])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC, "-c", os.path.join(ROOT, "tools", "hostonly", "lines_host_check.cpp"), "-o", str(o2)])
    subprocess.check_call(["g++", str(o2), str(o1), "-o", str(exe), "-L" + HIP_LIB, "-lamdhip64", "-Wl,-rpath," + HIP_LIB, "-pthread"])
    out = subprocess.run([str(exe), str(blob), "640", "480", "58"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "58 images" in out.stdout and " 0 differ" in out.stdout, out.stdout + out.stderr
