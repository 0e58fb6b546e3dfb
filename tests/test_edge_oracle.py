"""Known-answer tests of the CPU restatement of cv::Canny / cv::distanceTransform(DIST_L2, 3) / BGR2GRAY
(oracle/edge_oracle.cpp).  OpenCV itself is not available (parity unpinned for this row): the cases below are
hand-computed from the published algorithms."""
import numpy as np

from oracle import edge_oracle_py as E


def test_bgr_to_gray_fixed_point_weights():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30]]], np.uint8)   # B G R
    # (B*1868 + G*9617 + R*4899 + 8192) >> 14
    assert E.bgr_to_gray(px).tolist() == [[29, 150, 76, 255, 22]]


def test_distance_transform_single_feature_is_chamfer_3x3():
    e = np.zeros((9, 11), np.uint8)
    e[4, 5] = 255
    d = E.dist_l2_3x3(e)
    a, b = 62587, 89738                      # round(0.955f * 2^16), round(1.3693f * 2^16)
    for i in range(9):
        for j in range(11):
            dy, dx = abs(i - 4), abs(j - 5)
            want = np.float32(min(dx, dy) * b + abs(dx - dy) * a) * np.float32(1 / 65536)
            assert d[i, j] == want, (i, j)
    assert d[4, 5] == 0 and abs(d[4, 6] - 0.955) < 1e-4 and abs(d[5, 6] - 1.3693) < 1e-4


def test_distance_transform_two_features_and_empty_image():
    e = np.zeros((5, 20), np.uint8)
    e[2, 2] = 255; e[2, 17] = 255
    d = E.dist_l2_3x3(e)
    assert d[2, 9] == np.float32(7 * 62587) * np.float32(1 / 65536) and d[2, 10] == d[2, 9]      # nearest feature on either side
    empty = E.dist_l2_3x3(np.zeros((4, 6), np.uint8))
    assert np.all(empty == np.float32(0xffffffff - 89738) * np.float32(1 / 65536))                # saturated DIST_MAX


def test_canny_vertical_step_edge():
    g = np.zeros((40, 60), np.uint8)
    g[:, 30:] = 200                                      # dx = 800 on columns 29 and 30, dy = 0
    c = E.canny_roi(g, (0, 0, 60, 40))
    cols = np.nonzero(c.any(axis=0))[0]
    # equal magnitudes on columns 29 / 30: 'm > left && m >= right' keeps the left one only
    assert cols.tolist() == [29] and np.all(c[:, 29] == 255)
    # thresholds are strict: a step of 50 gives magnitude 200, not > 200 -> nothing above 'high' -> no edges at all
    g2 = np.zeros((40, 60), np.uint8); g2[:, 30:] = 50
    assert not E.canny_roi(g2, (0, 0, 60, 40)).any()
    # ... but a weak edge is kept where it touches a strong one (hysteresis): rows 0..19 strong, rows 20..39 weak
    g3 = np.zeros((40, 60), np.uint8); g3[:20, 30:] = 200; g3[20:, 30:] = 40
    c3 = E.canny_roi(g3, (0, 0, 60, 40))
    assert np.all(c3[:17, 29] == 255) and np.all(c3[24:, 29] == 255)     # (the corner at rows 19-20 bends the gradient)
    g4 = np.zeros((40, 60), np.uint8); g4[:, 30:] = 40                    # the same weak edge alone: dropped
    assert not E.canny_roi(g4, (0, 0, 60, 40)).any()


def test_canny_roi_uses_parent_pixels_and_replicates_at_image_border():
    rng = np.random.default_rng(0)
    g = (rng.integers(0, 2, (30, 40)) * 255).astype(np.uint8)
    g = np.kron(g, np.ones((4, 4), np.uint8))            # 120 x 160 blocks: plenty of strong edges
    full = E.canny_roi(g, (0, 0, 160, 120))
    roi = (16, 12, 100, 80)
    sub = E.canny_roi(g, roi)
    # interior pixels of the ROI (>= 2 px from its border: NMS looks one pixel out, hysteresis components may be cut)
    # see the same gradients as in the full image
    gx_full = full[12 + 2:12 + 78, 16 + 2:16 + 98]
    assert gx_full.shape == sub[2:-2, 2:-2].shape
    strong_equal = (sub[2:-2, 2:-2] == gx_full).mean()
    assert strong_equal > 0.98
    # a constant image has no edges, also at the replicated border
    assert not E.canny_roi(np.full((20, 20), 77, np.uint8), (0, 0, 20, 20)).any()


def test_edge_distance_map_composition():
    g = np.zeros((50, 70), np.uint8); g[:, 35:] = 255
    roi = (5, 5, 60, 40)
    m = E.edge_distance_map(g, roi)
    c = E.canny_roi(g, roi)
    assert np.array_equal(m, E.dist_l2_3x3(c))
    assert m[:, 34 - 5].max() == 0 and abs(m[10, 34 - 5 + 3] - 3 * 0.955) < 1e-3
