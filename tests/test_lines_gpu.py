"""The line-segment producer (cs_detect_lines_gray: EDLines with its per-pixel stages on the device, routing / fitting on the host)
against oracle/edlines_oracle.cpp, bit for bit: same number of segments, same order, identical float coordinates.  The LSD branch
(cs_detect_lsd_gray: use_LSD = true) the same way against oracle/lsd_oracle.cpp, and against the reference's own saved segments."""
import os

import numpy as np
import pytest

from cube_slam_wu_amd import capi
from oracle import edge_oracle_py as E
from oracle import edlines_oracle_py as L
from oracle import lsd_oracle_py as LSD

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "golden", "object_slam_data")


def _synthetic(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 100.0)
    for _ in range(12):
        a = rng.uniform(0, np.pi)
        img += np.where((xx - rng.uniform(0, w)) * np.cos(a) + (yy - rng.uniform(0, h)) * np.sin(a) > 0, rng.uniform(-60, 60), 0)
    img += rng.normal(0, 3, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_reference_tum_frames_bit_identical_to_the_oracle():
    from PIL import Image
    det = capi.Detector(capi.default_params())
    total = 0
    for k in range(58):
        img = np.asarray(Image.open(os.path.join(DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
        gray = E.bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1]))
        got = det.detect_lines(gray, 15.0)
        ref = L.detect_filter_lines(gray, 15.0)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.array_equal(got, ref), k
        total += len(ref)
    assert total > 800
    det.close()


@pytest.mark.parametrize("shape", [(480, 640), (376, 1241), (97, 131), (33, 64), (64, 33), (200, 7), (5, 300)])
def test_synthetic_images_and_odd_sizes(shape):
    """Sizes that are not multiples of the 32-pixel device tile, images narrower than the blur's support, the KITTI frame size."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    det = capi.Detector(capi.default_params())
    n = 0
    for _ in range(3):
        gray = _synthetic(rng, *shape)
        for thr in (15.0, 50.0):       # the graph driver's threshold and the class default (line_lbd_allclass.cpp:147)
            got = det.detect_lines(gray, thr)
            ref = L.detect_filter_lines(gray, thr)
            assert got.shape == ref.shape and np.array_equal(got, ref)
            n += len(ref)
    if min(shape) >= 64:
        assert n > 0
    det.close()


def test_flat_and_noise_images():
    det = capi.Detector(capi.default_params())
    flat = np.full((120, 160), 77, np.uint8)
    assert det.detect_lines(flat).shape == (0, 4) and L.detect_filter_lines(flat).shape == (0, 4)
    noise = np.random.default_rng(5).integers(0, 256, (240, 320), dtype=np.uint8)
    assert np.array_equal(det.detect_lines(noise), L.detect_filter_lines(noise))
    det.close()


def test_batch_entry_point_equals_the_single_image_calls():
    """cs_detect_lines_batch: the per-pixel stages of all images on the device, the sequential halves on the worker pool -- the same
    segments as one cs_detect_lines_gray call per image (which is the oracle's output), also on a second call that reuses the scratch
    and after a call with a different image size."""
    rng = np.random.default_rng(5)
    def image(h, w, seed):
        r = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.full((h, w), 100.0)
        for _ in range(14):
            a = r.uniform(0, np.pi)
            img += np.where((xx - r.uniform(0, w)) * np.cos(a) + (yy - r.uniform(0, h)) * np.sin(a) > 0, r.uniform(-60, 60), 0)
        return np.clip(img + r.normal(0, 3, img.shape), 0, 255).astype(np.uint8)
    det = capi.Detector(capi.default_params())
    grays = [image(200, 311, 40 + i) for i in range(9)]
    single = [det.detect_lines(g, 15.0) for g in grays]
    for _ in range(2):
        got = det.detect_lines_batch(grays, 15.0)
        assert len(got) == len(single)
        for a, b in zip(got, single):
            assert a.shape == b.shape and np.array_equal(a, b) and len(b) > 0
    small = image(97, 120, 77)
    assert np.array_equal(det.detect_lines_batch([small], 15.0)[0], det.detect_lines(small, 15.0))
    got = det.detect_lines_batch(grays[:3], 15.0)
    for a, b in zip(got, single[:3]):
        assert np.array_equal(a, b)
    t = det.lines_timing()
    assert t["device_ms"] > 0 and t["total_ms"] >= t["host_ms"] > 0
    det.close()


# ---- the LSD branch (use_LSD = true) ------------------------------------------------------------------------------------------
def test_lsd_reference_frame_reproduces_the_saved_segments_token_for_token():
    """The product on the gray image of the reference's bundled frame 0000 against detect_3d_cuboid/data/edge_detection/LSD/0000_edge.txt
    (271 rows the reference's LSD branch wrote, six significant digits): every token equal -- the GPU path needs no oracle for this one."""
    from PIL import Image
    gdir = os.path.join(os.path.dirname(__file__), "golden", "detect_3d_cuboid_data")
    gray = np.asarray(Image.open(os.path.join(gdir, "0000_gray.png")))
    det = capi.Detector(capi.default_params())
    got = det.detect_lines(gray, 15.0, use_lsd=True)
    want = open(os.path.join(gdir, "0000_edge.txt")).read().split()
    assert got.shape == (271, 4)
    bad = [(i // 4, a, b) for i, (a, b) in enumerate(zip(["%g" % v for v in got.reshape(-1)], want)) if float(a) != float(b)]
    assert not bad, bad[:5]
    assert np.array_equal(got, LSD.detect_filter_lines(gray, 15.0))
    det.close()


def test_lsd_tum_frames_bit_identical_to_the_oracle():
    from PIL import Image
    det = capi.Detector(capi.default_params())
    total = 0
    for k in range(0, 58, 3):
        img = np.asarray(Image.open(os.path.join(DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
        gray = E.bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1]))
        got = det.detect_lines(gray, 15.0, use_lsd=True)
        ref = LSD.detect_filter_lines(gray, 15.0)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert np.array_equal(got, ref), k
        total += len(ref)
    assert total > 400
    det.close()


@pytest.mark.parametrize("shape", [(480, 640), (376, 1241), (97, 131), (33, 64), (64, 33), (200, 9), (8, 300)])
def test_lsd_synthetic_images_and_odd_sizes(shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + 1)
    det = capi.Detector(capi.default_params())
    n = 0
    for _ in range(3):
        gray = _synthetic(rng, *shape)
        for thr in (15.0, 50.0):
            got = det.detect_lines(gray, thr, use_lsd=True)
            ref = LSD.detect_filter_lines(gray, thr)
            assert got.shape == ref.shape and np.array_equal(got, ref)
            n += len(ref)
    if min(shape) >= 64:
        assert n > 0
    flat = np.full(shape, 77, np.uint8)
    assert det.detect_lines(flat, 15.0, use_lsd=True).shape == (0, 4)
    det.close()


def test_lsd_batch_equals_single_calls_and_reports_errors():
    det = capi.Detector(capi.default_params())
    rng = np.random.default_rng(11)
    grays = [_synthetic(rng, 200, 311) for _ in range(9)]
    single = [det.detect_lines(g, 15.0, use_lsd=True) for g in grays]
    for _ in range(2):
        got = det.detect_lines_batch(grays, 15.0, use_lsd=True)
        for a, b, g in zip(got, single, grays):
            assert np.array_equal(a, b) and np.array_equal(b, LSD.detect_filter_lines(g, 15.0)) and len(b) > 0
    small = _synthetic(rng, 97, 120)
    assert np.array_equal(det.detect_lines_batch([small], 15.0, use_lsd=True)[0], LSD.detect_filter_lines(small, 15.0))
    # the EDLines producer of the same detector is untouched by the LSD calls in between
    assert np.array_equal(det.detect_lines(grays[0], 15.0), L.detect_filter_lines(grays[0], 15.0))
    t = det.lines_timing(use_lsd=True)
    assert t["device_ms"] > 0 and t["total_ms"] >= t["host_ms"] > 0
    with pytest.raises(RuntimeError):
        det.detect_lines(grays[0], 15.0, cap=3, use_lsd=True)
    with pytest.raises(RuntimeError):
        det.detect_lines(np.zeros((4, 4), np.uint8), 15.0, use_lsd=True)
    det.close()
