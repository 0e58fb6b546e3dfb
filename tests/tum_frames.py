"""The reference's bundled TUM cabinet sequence as detect_cuboid inputs (test infrastructure).

Data under tests/golden/object_slam_data are copies of files the reference ships (object_slam/data: raw_imgs/*.jpg,
filter_2d_obj_txts/*.txt, pop_cam_poses_saved.txt, detect_cuboids_saved.txt); segments/*.txt are the output of the EDLines
restatement (oracle/edlines_oracle.cpp) on each JPEG, written by tools/make_tum_segments.py.  A frame is set up the way main_obj.cpp does it in
its online branch (:585-640): TUM calibration (:484-486), the first 2D box of the frame's file shifted to 0-based (:621),
no height sampling, nominal_skew_ratio 2 (:496-498) -- with one difference: the camera pose is the frame's row of
pop_cam_poses_saved.txt (camera above the origin of its own ground frame), because that is the frame
detect_cuboids_saved.txt is expressed in (main_obj.cpp:692-708), so the detector's output can be compared with it directly.
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "object_slam_data")
K_TUM = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1.0]])


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def frame_ids():
    """Frames that have a 2D box and a saved detection."""
    saved = np.loadtxt(os.path.join(DATA, "detect_cuboids_saved.txt"))
    ids = []
    for k in saved[:, 0].astype(int):
        p = os.path.join(DATA, "filter_2d_obj_txts", "%04d_yolo2_0.15.txt" % k)
        if os.path.getsize(p) > 0:
            ids.append(int(k))
    return ids


def load(k, bgr_to_gray):
    """(frame dict without maps, gray image, saved detection row [x y z yaw sx sy sz err]) of frame k."""
    from PIL import Image

    from cube_slam_wu_amd import synth
    img = np.asarray(Image.open(os.path.join(DATA, "raw_imgs", "%04d_rgb_raw.jpg" % k)).convert("RGB"))
    gray = bgr_to_gray(np.ascontiguousarray(img[:, :, ::-1]))
    H, W = gray.shape
    box = np.loadtxt(os.path.join(DATA, "filter_2d_obj_txts", "%04d_yolo2_0.15.txt" % k)).reshape(-1, 5)[:1].copy()
    box[:, :2] -= 1
    pop = np.loadtxt(os.path.join(DATA, "pop_cam_poses_saved.txt"))[k]
    T = np.eye(4)
    T[:3, :3] = _quat_to_R(pop[4:8])
    T[:3, 3] = pop[1:4]
    lines = np.loadtxt(os.path.join(DATA, "segments", "%04d.txt" % k)).reshape(-1, 4)
    saved = np.loadtxt(os.path.join(DATA, "detect_cuboids_saved.txt"))
    row = saved[saved[:, 0] == k][0][1:]
    fr = dict(K=K_TUM, T_wc=T, boxes=box, lines=lines, rois=[synth.box_rois(box[0], W, H, False)], img_w=W, img_h=H)
    return fr, gray, row


def load_for_online_run(k, bgr_to_gray):
    """(frame dict without pose and maps, gray) of frame k for oracle.ba_oracle_py.run_online_sequence, None without a 2D box."""
    if os.path.getsize(os.path.join(DATA, "filter_2d_obj_txts", "%04d_yolo2_0.15.txt" % k)) == 0:
        return None
    fr, gray, _ = load(k, bgr_to_gray)
    fr.pop("T_wc")
    return fr, gray
