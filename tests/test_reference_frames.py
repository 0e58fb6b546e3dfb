"""The reference's own results on its own images: what pins the oracles.

object_slam/data holds 58 TUM frames, their 2D boxes, and three kinds of results of the reference itself:
* output_obj_poses.txt / output_cam_poses.txt -- what its ONLINE run wrote (main_obj.cpp:479-841): per frame EDLines segments
  (line_lbd), detect_cuboid (roll/pitch sampling from the second frame on), measurement conversion, the growing graph,
  optimize(5).  The restated pipeline -- oracle/edlines_oracle.cpp -> oracle/edge_oracle.cpp (Canny + distance transform) ->
  oracle/detect_oracle.cpp -> oracle/ba_oracle.cpp under the restated driver -- reproduces BOTH files to the six significant
  digits they are printed with, for every one of the 58 frames.  Any different ranking decision in any frame would move the
  object by centimetres, so this pins path A's integer decisions, path B's LM trajectory and the segment producer together.
* the nine segments printed in main_obj.cpp:603-611: frame 17's segments 1-9, to all printed digits.
* detect_cuboids_saved.txt / pop_cam_poses_saved.txt -- cuboids and camera poses of the author's earlier offline tool chain (other
  segments, other poses): the oracle lands on the same yaw sample in two thirds of the frames and within centimetres; kept as a
  sanity envelope, it cannot agree better than the different inputs allow.
"""
import numpy as np
import pytest

from oracle import ba_oracle_py as O
from oracle import edge_oracle_py as E
from oracle import edlines_oracle_py as L
from oracle import oracle_py

import tum_frames

pytest.importorskip("PIL")
STEP = 6.0 / 180.0 * np.pi      # the reference's yaw sweep (box_proposal_detail.cpp:184)


def _oracle_detection(k):
    fr, gray, row = tum_frames.load(k, E.bgr_to_gray)
    maps = []
    for (l, t, w, h), _ in fr["rois"][0]:
        buf = np.zeros(h * w + w + 1, np.float32)
        buf[: h * w] = E.edge_distance_map(gray, (l, t, w, h)).ravel()
        maps.append(buf)
    fr["maps"] = [maps]
    res, _ = oracle_py.detect_cuboid(fr, oracle_py.default_params(nominal_skew_ratio=2.0))
    return res[0], row


def test_oracle_reproduces_the_references_saved_detections():
    ids = tum_frames.frame_ids()
    assert len(ids) == 51
    dxy, dz, steps, exact = [], [], [], 0
    for k in ids:
        got, row = _oracle_detection(k)
        assert len(got) == 1, k
        c = got[0]
        d = np.array(c["pos"]) - row[:3]
        dyaw = (c["rotY"] - row[3] + np.pi / 4) % (np.pi / 2) - np.pi / 4      # a cuboid's yaw is defined modulo 90 degrees
        n = round(dyaw / STEP)
        assert abs(dyaw - n * STEP) < 6e-3, (k, dyaw)          # same sweep: yaw differs by whole samples (3-digit file)
        dxy.append(np.hypot(d[0], d[1])); dz.append(abs(d[2])); steps.append(abs(n))
        # same yaw sample: the cuboid is the reference's to within the file's precision and the segments' influence
        if n == 0 and np.hypot(d[0], d[1]) < 0.03 and abs(d[2]) < 0.01 and np.abs(np.sort(c["scale"][:2]) - np.sort(row[4:6])).max() < 0.05:
            exact += 1
    dxy, dz, steps = np.array(dxy), np.array(dz), np.array(steps)
    assert np.median(dxy) < 0.05 and np.median(dz) < 0.015
    assert (dxy < 0.2).mean() > 0.9 and dxy.max() < 0.5
    assert (steps == 0).mean() >= 0.6 and steps.max() <= 2
    assert exact >= 18


def _oracle_detect(fr, gray, sample_roll_pitch):
    maps = []
    for (l, t, w, h), _ in fr["rois"][0]:
        buf = np.zeros(h * w + w + 1, np.float32)
        buf[: h * w] = E.edge_distance_map(gray, (l, t, w, h)).ravel()
        maps.append(buf)
    res, _ = oracle_py.detect_cuboid(dict(fr, maps=[maps]), oracle_py.default_params(nominal_skew_ratio=2.0, whether_sample_cam_roll_pitch=sample_roll_pitch))
    return res[0][0] if res[0] else None


def check_online_run_against_saved_outputs(out_obj, final_cams):
    """output_obj_poses.txt / output_cam_poses.txt are what the reference wrote after its own online run over these images."""
    import os
    so = np.loadtxt(os.path.join(tum_frames.DATA, "output_obj_poses.txt"))
    sc = np.loadtxt(os.path.join(tum_frames.DATA, "output_cam_poses.txt"))
    assert out_obj.shape == so.shape == (58, 9)
    # both files are printed with six significant digits (main_obj.cpp:305-336): agreement to that precision, every frame, every field
    tol = lambda ref: 1.5e-6 * np.maximum(1.0, np.abs(ref)) + 1e-6
    assert np.all(np.abs(out_obj - so) <= 10 * tol(so)), np.abs(out_obj - so).max()
    assert np.all(np.abs(final_cams - sc[:, 1:8]) <= 10 * tol(sc[:, 1:8])), np.abs(final_cams - sc[:, 1:8]).max()
    assert np.abs(out_obj - so).max() < 2e-5 and np.abs(final_cams - sc[:, 1:8]).max() < 2e-5


def test_online_run_reproduces_the_references_saved_outputs():
    """Image in, trajectory and object out: the detector oracle (roll/pitch sampling as the reference's online branch uses it)
    feeding the bundle-adjustment oracle through main_obj.cpp's online graph construction, against the two result files the
    reference saved from its own run.  Both paths and the driver between them are pinned here -- up to the segment detector."""
    yaws = []

    def detect(fr, gray, sample):
        c = _oracle_detect(fr, gray, sample)
        if c is not None and sample:
            yaws.append(c["rotY"])
        return c

    def load(k):     # the segments come from the line-detector oracle here, not from the fixture files
        r = tum_frames.load_for_online_run(k, E.bgr_to_gray)
        if r is None:
            return None
        fr, gray = r
        return dict(fr, lines=L.detect_filter_lines(gray, 15.0).astype(np.float64)), gray

    out_obj, final_cams, n_det = O.run_online_sequence(tum_frames.DATA, load, detect)
    assert n_det == 51
    check_online_run_against_saved_outputs(out_obj, final_cams)
    # detect_3d_cuboid.h:44-56 prints a cuboid of this sequence (640 x 480 corners, the first camera's height in the plane
    # printed at object_3d_util.cpp:893-896): its rotY, -2.90009, is a value of the yaw sweep around the first camera's pose
    # (camera yaw from the quaternion, minus 90 degrees, +- k x 6 degrees accumulated by linespace) -- the oracle returns
    # exactly that number, to all six printed digits, for two of the frames
    assert sum(abs(y - (-2.90009)) < 5e-6 for y in yaws) >= 2


def test_segments_printed_in_the_references_source_are_frame_17s():
    """main_obj.cpp:603-611 prints (in a comment) the all_lines_raw matrix of one frame of this sequence -- nine segments, six
    significant digits.  They are segments 1-9 of frame 17 as the EDLines restatement detects them (JPEG decode, BGR2GRAY,
    5 x 5 Gaussian, Sobel, anchors, smart routing, least-squares fit, NFA validation, start / end ordering), in order, to every
    printed digit.  The fixture file of the frame holds the same numbers."""
    import os
    from PIL import Image
    printed = np.array([[518.164, 179.13, 533, 46], [453.637, 371.208, 516.2, 180.066], [285.62, 322.261, 451.626, 372.243],
                        [290.387, 120.984, 285.264, 319.981], [384.164, 22.1538, 291.708, 120.726], [514.869, 171.607, 290.926, 123.344],
                        [380.963, 167.12, 398.815, 172.601], [381.094, 202.619, 395.935, 206.264], [397.868, 176.387, 379.907, 170.274]])
    img = np.asarray(Image.open(os.path.join(tum_frames.DATA, "raw_imgs", "0017_rgb_raw.jpg")).convert("RGB"))
    seg = L.detect_filter_lines(E.bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1])), 15.0)
    assert seg.shape == (10, 4)
    got = seg[1:10].astype(np.float64)
    # six significant digits: half a unit of the last printed place
    ulp6 = 10.0 ** (np.floor(np.log10(np.abs(printed))) - 5)
    assert np.all(np.abs(got - printed) <= 0.5 * ulp6 + 1e-9), np.abs(got - printed).max()
    fixture = np.loadtxt(os.path.join(tum_frames.DATA, "segments", "0017.txt"))
    assert np.array_equal(fixture.astype(np.float32), seg)
