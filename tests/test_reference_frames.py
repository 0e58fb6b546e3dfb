"""detect_cuboid on the reference's own images against the detections the reference saved.

object_slam/data holds 58 TUM frames, their 2D boxes and detect_cuboids_saved.txt: one cuboid per frame (position, yaw, half
sizes to three digits) in the frame's ground coordinates -- the output of the reference's detector on exactly these images.
The oracle, fed with the image (its own Canny + distance transform restatement) and with segments from a plain edge-chain
splitter instead of the reference's EDLines, lands on the same cuboids: the yaw differs by whole 6-degree samples of the sweep
(0 in most frames), the position by centimetres.  This is the one place where results of the reference itself pin path A --
to the three digits the file carries, and up to the different segment detector.
"""
import numpy as np
import pytest

from oracle import ba_oracle_py as O
from oracle import edge_oracle_py as E
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
    assert (steps == 0).mean() >= 0.5 and steps.max() <= 2
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
    assert np.linalg.norm(out_obj[:, :3] - so[:, :3], axis=1).max() < 0.003            # object position: millimetres, every frame
    dyaw = (out_obj[:, 5] - so[:, 5] + np.pi / 4) % (np.pi / 2) - np.pi / 4
    assert np.abs(dyaw).max() < 0.003 and np.abs(out_obj[:, 3:5] - so[:, 3:5]).max() < 0.002
    assert np.abs(out_obj[:, 6:] - so[:, 6:]).max() < 0.012                            # half sizes (they average the detections)
    d = np.linalg.norm(final_cams[:, :3] - sc[:, 1:4], axis=1)
    assert d.mean() < 0.05 and d.max() < 0.2                                           # cameras follow the single detections


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

    out_obj, final_cams, n_det = O.run_online_sequence(tum_frames.DATA, lambda k: tum_frames.load_for_online_run(k, E.bgr_to_gray), detect)
    assert n_det == 51
    check_online_run_against_saved_outputs(out_obj, final_cams)
    # detect_3d_cuboid.h:44-56 prints a cuboid of this sequence (640 x 480 corners, the first camera's height in the plane
    # printed at object_3d_util.cpp:893-896): its rotY, -2.90009, is a value of the yaw sweep around the first camera's pose
    # (camera yaw from the quaternion, minus 90 degrees, +- k x 6 degrees accumulated by linespace) -- the oracle returns
    # exactly that number, to all six printed digits, for two of the frames
    assert sum(abs(y - (-2.90009)) < 5e-6 for y in yaws) >= 2
