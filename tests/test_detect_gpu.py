"""GPU parity: the HIP sweep (through the C ABI) against the CPU oracle, bit for bit.

Stages compared per (box, height sample): candidate rows [config, vp1 side, yaw, top id, dist error,
angle error, down expand, roll, pitch] and the 2x8 corners (box_proposal_detail.cpp:677-693), the kept
ids and normalised scores of fuse_normalize_scores_v2, and the final cuboid records.  Integer/ranking
outputs must be identical; doubles are compared with array_equal (bit-exact): both sides evaluate the
same IEEE operations in the same order with contraction off and share cs_atan2.
"""
import os

import numpy as np
import pytest

from cube_slam_wu_amd import capi, synth
from oracle import oracle_py

pytestmark = pytest.mark.gpu

CUBOID_KEYS = ["pos", "scale", "rotY", "box_config_type", "box_corners_2d", "box_corners_3d_world", "rect_detect_2d",
               "edge_distance_error", "edge_angle_error", "normalized_error", "skew_ratio", "down_expand_height",
               "camera_roll_delta", "camera_pitch_delta"]


def _oracle_params(p):
    return oracle_py.default_params(
        consider_config_1=p.consider_config_1, consider_config_2=p.consider_config_2,
        whether_sample_cam_roll_pitch=p.whether_sample_cam_roll_pitch, whether_sample_bbox_height=p.whether_sample_bbox_height,
        max_cuboid_num=p.max_cuboid_num, nominal_skew_ratio=p.nominal_skew_ratio, max_cut_skew=p.max_cut_skew,
        yaw_range_deg=p.yaw_range_deg, yaw_step_deg=p.yaw_step_deg)


def _same(a, b):
    """Bit-exact equality; NaN matches NaN (0/0 scores when every kept proposal has the same error)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


def _check(frames, params, cap=20000):
    det = capi.Detector(params)
    bat = capi.Batch(det, frames, debug=True)
    bat.run()
    n_cmp = 0
    for f, fr in enumerate(frames):
        ref, dbg = oracle_py.detect_cuboid(fr, _oracle_params(params), atan2_mode=1, debug_cap=cap)
        got = bat.cuboids(f)
        nb = len(fr["boxes"])
        for i in range(nb):
            nh = len(fr["maps"][i])
            for k in range(nh):
                slot = 3 * i + k
                V = int(dbg["n_valid"][slot])
                assert V <= cap
                rows, corners = bat.debug_candidates(f, i, k)
                assert rows.shape[0] == V, (f, i, k, rows.shape[0], V)
                assert _same(rows, dbg["cand_rows"][slot][:V]), (f, i, k)
                assert _same(corners, dbg["cand_corners"][slot][:V]), (f, i, k)
                ids, sc = bat.debug_kept(f, i, k)
                nk = int(dbg["n_keep"][slot])
                assert len(ids) == nk
                assert _same(ids, dbg["keep_ids"][slot][:nk])
                assert _same(sc, dbg["keep_scores"][slot][:nk])
                n_cmp += V
            assert len(got[i]) == len(ref[i]), (f, i, len(got[i]), len(ref[i]))
            for a, b in zip(got[i], ref[i]):
                for key in CUBOID_KEYS:
                    assert _same(a[key], b[key]), (f, i, key, a[key], b[key])
    bat.close()
    det.close()
    return n_cmp


def test_reference_sweep_6deg_no_sampling():
    frames = [synth.make_frame(1000 + s) for s in range(3)]
    n = _check(frames, capi.default_params(whether_sample_cam_roll_pitch=0))
    assert n > 500


def test_headline_sweep_half_degree():
    frames = [synth.make_frame(2000 + s) for s in range(2)]
    n = _check(frames, capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=0.5))
    assert n > 10000


def test_roll_pitch_sampling_carries_camera_yaw():
    frames = [synth.make_frame(3000 + s, n_boxes=4, n_lines=250) for s in range(2)]
    n = _check(frames, capi.default_params(whether_sample_cam_roll_pitch=1, yaw_step_deg=6.0))
    assert n > 5000


def test_height_sampling_and_topk():
    frames = [synth.make_frame(4000 + s, n_boxes=3, n_lines=200, sample_height=True) for s in range(2)]
    n = _check(frames, capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=1, max_cuboid_num=5, yaw_step_deg=3.0))
    assert n > 1000


def test_single_config_flags():
    frames = [synth.make_frame(5000)]
    _check(frames, capi.default_params(whether_sample_cam_roll_pitch=0, consider_config_2=0))
    _check(frames, capi.default_params(whether_sample_cam_roll_pitch=0, consider_config_1=0))


def test_edge_cases_empty_and_ragged():
    fr = synth.make_frame(6000, n_boxes=3, n_lines=120)
    # no line segments at all: every VP support is NaN -> constant angle penalty
    fr_nolines = dict(fr)
    fr_nolines["lines"] = np.zeros((0, 4))
    # a box too narrow to sample the top edge (w/10 < 1 -> the reference breaks, :215)
    fr_narrow = synth.make_frame(6001, n_boxes=2, n_lines=80)
    fr_narrow["boxes"] = fr_narrow["boxes"].copy()
    fr_narrow["boxes"][0, 2] = 9
    fr_narrow["rois"] = [synth.box_rois(b, fr_narrow["img_w"], fr_narrow["img_h"]) for b in fr_narrow["boxes"]]
    fr_narrow["maps"] = [[np.zeros(r[0][2] * r[0][3] + r[0][2] + 1, np.float32) for r in rr] for rr in fr_narrow["rois"]]
    # a frame without boxes, in the middle of a batch (ragged)
    fr_empty = dict(fr)
    fr_empty["boxes"] = np.zeros((0, 5)); fr_empty["maps"] = []; fr_empty["rois"] = []
    _check([fr, fr_empty, fr_nolines, fr_narrow], capi.default_params(whether_sample_cam_roll_pitch=0))


def test_single_frame_entry_point_matches_batch():
    import ctypes as C
    fr = synth.make_frame(7000, n_boxes=2, n_lines=100)
    p = capi.default_params(whether_sample_cam_roll_pitch=0)
    det = capi.Detector(p)
    bat = capi.Batch(det, [fr])
    bat.run()
    ref = bat.raw_out_bytes()
    # cs_detect_cuboids on the same frame
    K = np.ascontiguousarray(fr["K"]).reshape(9); T = np.ascontiguousarray(fr["T_wc"]).reshape(16)
    boxes = np.ascontiguousarray(fr["boxes"]); lines = np.ascontiguousarray(fr["lines"])
    arr = (C.POINTER(C.c_float) * 6)()
    keep = []
    for i in range(2):
        m = np.ascontiguousarray(fr["maps"][i][0]); keep.append(m)
        arr[3 * i] = m.ctypes.data_as(C.POINTER(C.c_float))
    d = capi.CsFrameDesc(capi._dp(K), capi._dp(T), fr["img_w"], fr["img_h"], capi._dp(boxes), 2, capi._dp(lines), len(lines), arr)
    out = (capi.CsCuboid * 2)()
    cnt = np.zeros(2, np.int32)
    rc = capi.lib().cs_detect_cuboids(det.h, C.byref(d), out, cnt.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, capi.last_error()
    assert bytes(out) == ref


def _check_final(frames, params):
    """Production path (ranking on the device, nothing but the winners comes back): final records vs the oracle."""
    det = capi.Detector(params)
    bat = capi.Batch(det, frames)
    bat.run()
    n = 0
    for f, fr in enumerate(frames):
        ref, _ = oracle_py.detect_cuboid(fr, _oracle_params(params), atan2_mode=1)
        got = bat.cuboids(f)
        for i in range(len(fr["boxes"])):
            assert len(got[i]) == len(ref[i]), (f, i, len(got[i]), len(ref[i]))
            for a, b in zip(got[i], ref[i]):
                for key in CUBOID_KEYS:
                    assert _same(a[key], b[key]), (f, i, key, a[key], b[key])
                n += 1
    tm = bat.timing()
    bat.close(); det.close()
    return n, tm


def test_device_ranking_matches_oracle_top1():
    frames = [synth.make_frame(8000 + s) for s in range(6)]
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=0.5))
    assert n >= 40 and tm["rank_kernel_ms"] > 0
    assert tm["n_fallback_boxes"] <= 4     # ties at a cut are rare on real-valued scores


def test_device_ranking_matches_oracle_topk_and_heights():
    frames = [synth.make_frame(8100 + s, n_boxes=4, n_lines=250, sample_height=True) for s in range(3)]
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=1, max_cuboid_num=5, yaw_step_deg=2.0))
    assert n >= 40


def test_device_ranking_falls_back_on_ties():
    """A constant distance map makes every distance error equal: every cut is a tie, so the kernel must hand the
    box to the exact host ranking -- and the result must still be the oracle's."""
    fr = synth.make_frame(8200, n_boxes=3, n_lines=200)
    fr["maps"] = [[np.zeros_like(m) for m in mm] for mm in fr["maps"]]
    n, tm = _check_final([fr], capi.default_params(whether_sample_cam_roll_pitch=0))
    assert tm["n_fallback_boxes"] == 3


def test_many_tie_boxes_take_the_separate_host_pass():
    """More than 64 boxes whose ties reach the output (constant maps again): the host ranks them in their own parallel pass
    after the records of the other boxes instead of inside it."""
    frames = []
    for s in range(10):
        fr = synth.make_frame(8400 + s, n_boxes=8, n_lines=150)
        fr["maps"] = [[np.zeros_like(m) for m in mm] for mm in fr["maps"]]
        frames.append(fr)
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=0))
    assert tm["n_fallback_boxes"] == 80


def test_boxes_too_large_for_the_wavefront_rankings_registers():
    """rank_wave_kernel keeps a height sample's upper key words in 28 registers per lane: 1 792 valid proposals.  A quarter-degree yaw sweep
    (361 samples) gives a box ~2 500 of them, so the wavefront runs rank_kernel's procedure (rank_block_body<64, true>) instead -- records
    still the oracle's, with top-3 and a height-sampled frame in the mix."""
    plain = [synth.make_frame(8700 + s) for s in range(2)]
    tall = [synth.make_frame(8710, n_boxes=3, n_lines=250, sample_height=True)]
    for kmax, hs in ((1, 0), (3, 1)):
        n, tm = _check_final(tall if hs else plain, capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=hs, yaw_step_deg=0.25, max_cuboid_num=kmax))
        assert n >= 3 and tm["rank_kernel_ms"] > 0
        assert tm["n_valid"] / max(1, tm["n_jobs"]) > 1792, (tm["n_valid"], tm["n_jobs"])      # (the average height sample is beyond the registers)


def test_workgroup_ranking_kernel_alone_gives_the_same_records():
    """The boxes of <= 1 792 valid proposals per height sample are ranked by rank_wave_kernel (one wavefront per box, the order
    statistics by a bit walk over candidate masks), the others by rank_kernel (a workgroup per box, radix selection): with
    CS_RANK_WAVE=0 every box goes to rank_kernel -- the ranking tests above must hold to the oracle either way, so the two kernels
    agree with each other record for record."""
    import subprocess, sys
    here = os.path.abspath(__file__)
    out = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-x", "-q", "-k", "device_ranking or tie_boxes or ties_decided or height_sampling"],
                         env={**os.environ, "CS_RANK_WAVE": "0"}, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(here)))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_device_ranking_with_roll_pitch_sampling_carries_camera_yaw():
    """The reference's default mode: 25 camera poses per box, and the camera yaw the next box of the frame starts from is the
    yaw of the pose of the LAST proposal the previous box's ranking kept.  The device ranking hands that proposal back
    (rank_kernel last_slot); the final records of every box -- whose yaw lists depend on it -- must be the oracle's."""
    frames = [synth.make_frame(8500 + s, n_boxes=4, n_lines=250) for s in range(4)]
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=1, yaw_step_deg=6.0))
    assert n >= 12 and tm["rank_kernel_ms"] > 0 and tm["rank_host_ms"] == 0
    # with height sampling and several winners per box
    frames = [synth.make_frame(8600 + s, n_boxes=3, n_lines=200, sample_height=True) for s in range(3)]
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=1, whether_sample_bbox_height=1, max_cuboid_num=3, yaw_step_deg=6.0))
    assert n >= 9 and tm["rank_host_ms"] == 0


def test_device_ranking_with_roll_pitch_sampling_ties_go_to_the_host():
    """Constant distance maps: every cut is a tie, the last kept proposal is the heap order's choice -- the box (and the yaw it
    leaves behind) comes from the exact host ranking."""
    frames = []
    for s in range(2):
        fr = synth.make_frame(8700 + s, n_boxes=3, n_lines=200)
        fr["maps"] = [[np.zeros_like(m) for m in mm] for mm in fr["maps"]]
        frames.append(fr)
    n, tm = _check_final(frames, capi.default_params(whether_sample_cam_roll_pitch=1, yaw_step_deg=6.0))
    assert tm["n_fallback_boxes"] == 6


def test_host_ranking_path_still_exact():
    frames = [synth.make_frame(8300)]
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0))
    a = capi.Batch(det, frames); a.run()
    b = capi.Batch(det, frames, force_host_rank=True); b.run()
    assert a.raw_out_bytes() == b.raw_out_bytes()
    assert b.timing()["rank_kernel_ms"] == 0


def test_ties_decided_on_the_device_agree_with_the_exact_host_ranking():
    """About 4 % of the C2 boxes have several proposals AT the distance cut.  rank_kernel keeps them on the device when it can
    prove that the reference's pick among the tied proposals cannot reach the output; the proof is checked
    here against the exact std::partial_sort ranking of every box (force_host_rank) on 960 boxes."""
    frames = [synth.make_frame(100000 + s) for s in range(120)]
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5))
    a = capi.Batch(det, frames); a.run()
    b = capi.Batch(det, frames, force_host_rank=True); b.run()
    assert a.raw_out_bytes() == b.raw_out_bytes()
    assert a.timing()["n_fallback_boxes"] <= 8 and b.timing()["rank_kernel_ms"] == 0
    a.close(); b.close(); det.close()


def test_host_and_device_line_setup_agree():
    """merge_break_lines on the device (line_setup_kernel) against the host implementation: identical records."""
    frames = [synth.make_frame(8400 + s, n_lines=600) for s in range(3)]
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=2.0, max_cuboid_num=3))
    a = capi.Batch(det, frames); a.run()
    b = capi.Batch(det, frames, force_host_setup=True); b.run()
    assert a.raw_out_bytes() == b.raw_out_bytes()
    assert a.timing()["line_setup_ms"] > 0 and b.timing()["line_setup_ms"] < 0.05


def test_line_setup_crowded_rois():
    """ROIs that hold nearly every segment of the frame (long merge chains, LDS tables close to capacity): the device
    line setup must still produce the host implementation's records."""
    frames = []
    for s in range(3):
        fr = synth.make_frame(8450 + s, n_boxes=3, n_lines=480)
        left, top, w, h = fr["boxes"][0][:4]
        L = np.asarray(fr["lines"], np.float64).reshape(-1, 4).copy()
        W, H = float(fr["img_w"]), float(fr["img_h"])
        L[:, [0, 2]] = left + 2 + L[:, [0, 2]] / W * (w - 4)      # squeeze every segment into box 0
        L[:, [1, 3]] = top + 2 + L[:, [1, 3]] / H * (h - 4)
        fr = dict(fr); fr["lines"] = L
        frames.append(fr)
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=3.0, max_cuboid_num=2))
    a = capi.Batch(det, frames); a.run()
    b = capi.Batch(det, frames, force_host_setup=True); b.run()
    assert a.raw_out_bytes() == b.raw_out_bytes() and a.counts_bytes() == b.counts_bytes()
    assert a.timing()["line_setup_ms"] > 0 and b.timing()["line_setup_ms"] < 0.05
    a.close(); b.close(); det.close()


def test_chunk_pipeline_matches_single_pass():
    """The lean production path cut into 4 chunks of a two-slot pipeline (host packs / finishes one chunk while the GPU sweeps the next);
    the records must be byte-identical to the one-pass path, also on a second run that reuses the slots, with ragged
    frames (no boxes, no lines) and tie boxes (host re-rank on the second stream) in the middle of the batch."""
    uniq = [synth.make_frame(8500 + s, n_boxes=1 + s % 4, n_lines=150 + 20 * s) for s in range(12)]
    empty = dict(uniq[0]); empty["boxes"] = np.zeros((0, 5)); empty["maps"] = []
    nolines = dict(uniq[1]); nolines["lines"] = np.zeros((0, 4))
    tie = synth.make_frame(8600, n_boxes=2, n_lines=150)
    tie["maps"] = [[np.zeros_like(m) for m in mm] for mm in tie["maps"]]
    frames = [uniq[i % 12] for i in range(70)]
    frames[17] = empty; frames[18] = nolines; frames[40] = tie; frames[69] = empty
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=3.0, max_cuboid_num=2))
    a = capi.Batch(det, frames, pipeline_chunks=4); a.run()
    b = capi.Batch(det, frames, force_no_pipeline=True); b.run()
    c = capi.Batch(det, frames); c.run()     # default: the lean path in one pass
    ref = b.raw_out_bytes()
    assert c.raw_out_bytes() == ref; c.close()
    assert a.raw_out_bytes() == ref
    assert a.counts_bytes() == b.counts_bytes()
    a.run()
    assert a.raw_out_bytes() == ref
    ta, tb = a.timing(), b.timing()
    assert ta["cand_kernel_launches"] == 4 and tb["cand_kernel_launches"] == 1
    assert ta["n_valid"] == tb["n_valid"] and ta["n_slots"] == tb["n_slots"] and ta["n_fallback_boxes"] == tb["n_fallback_boxes"] >= 2
    a.close(); b.close(); det.close()


def test_submit_collect_two_batches_of_one_detector_in_flight():
    """cs_batch_submit / cs_batch_collect: the next batch is packed and queued while the previous one is still on the device.  Two
    different batches of one detector, both submitted before either is collected, give the records of their synchronous runs (the
    device-written records of the lean path: byte-identical to the path that writes them on the host, tie boxes included); a second
    submit of an uncollected batch is refused."""
    fa = [synth.make_frame(8700 + s, n_boxes=1 + s % 3, n_lines=140 + 15 * s) for s in range(9)]
    fb = [synth.make_frame(8800 + s, n_boxes=2, n_lines=120) for s in range(5)]
    tie = synth.make_frame(8600, n_boxes=2, n_lines=150)
    tie["maps"] = [[np.zeros_like(m) for m in mm] for mm in tie["maps"]]
    fb[2] = tie
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=3.0, max_cuboid_num=2))
    ra = capi.Batch(det, fa, force_no_pipeline=True); ra.run()
    rb = capi.Batch(det, fb, force_no_pipeline=True); rb.run()
    a, b = capi.Batch(det, fa), capi.Batch(det, fb)
    for _ in range(2):          # the second round reuses the slots
        a.submit(); b.submit()
        with pytest.raises(RuntimeError):
            a.submit()
        a.collect(); b.collect()
        assert a.raw_out_bytes() == ra.raw_out_bytes() and a.counts_bytes() == ra.counts_bytes()
        assert b.raw_out_bytes() == rb.raw_out_bytes() and b.counts_bytes() == rb.counts_bytes()
    assert b.timing()["n_fallback_boxes"] >= 2
    a.collect()                 # nothing outstanding: a no-op
    for x in (a, b, ra, rb):
        x.close()
    det.close()


@pytest.mark.parametrize("heights,step", [(0, 0.5), (1, 3.0)])
def test_roll_pitch_rounds_queued_at_once_equal_the_round_by_round_path(heights, step):
    """The lean roll/pitch path (every yaw list of a frame laid down up front, the carried yaw picked on the device, all rounds queued
    at once, tie boxes ranked exactly from columns saved on the device) against the round-by-round path with a host decision after
    every box round: byte-identical records at the 0.5 degree sweep (25 camera poses per box: ~70 k valid proposals per frame), with
    ragged frames (no boxes, one box, no lines), a frame whose maps are all zero (every box a tie: the host ranks it, and its carried
    yaw has to match), two cuboids per box, with and without height sampling (three jobs per box), on a second run that reuses the
    slots, and through submit / collect."""
    uniq = [synth.make_frame(8900 + s, n_boxes=1 + s % 8, n_lines=200 + 25 * s, sample_height=bool(heights)) for s in range(10)]
    empty = dict(uniq[0]); empty["boxes"] = np.zeros((0, 5)); empty["maps"] = []
    nolines = dict(uniq[3]); nolines["lines"] = np.zeros((0, 4))
    tie = synth.make_frame(8950, n_boxes=3, n_lines=150, sample_height=bool(heights))
    tie["maps"] = [[np.zeros_like(m) for m in mm] for mm in tie["maps"]]
    frames = [uniq[i % 10] for i in range(24)]
    frames[5] = empty; frames[6] = nolines; frames[11] = tie; frames[23] = empty
    det = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=1, whether_sample_bbox_height=heights, yaw_step_deg=step, max_cuboid_num=2))
    ref = capi.Batch(det, frames, force_no_pipeline=True); ref.run()
    lean = capi.Batch(det, frames)
    for mode in ("run", "run", "submit"):
        if mode == "run":
            lean.run()
        else:
            lean.submit(); lean.collect()
        assert lean.raw_out_bytes() == ref.raw_out_bytes() and lean.counts_bytes() == ref.counts_bytes()
    tl, tr = lean.timing(), ref.timing()
    assert tl["n_valid"] == tr["n_valid"] and tl["n_fallback_boxes"] >= 3 and tl["rank_host_ms"] == 0
    assert tl["cand_kernel_launches"] == tr["cand_kernel_launches"] == 8      # eight box rounds either way (three height samples per box: one job each)
    print("lean %.2f ms (fallback boxes %d, frames redone %d), round by round %.2f ms" % (tl["total_ms"], tl["n_fallback_boxes"], tl["n_redo_frames"], tr["total_ms"]))
    lean.close(); ref.close(); det.close()


def test_full_size_c2_batch_against_oracle_sample_and_properties():
    """BASELINE.json's C2 at the bench's full batch size (1000 frames x 8 boxes x 181 yaw x ~400 segments, 31.5 M
    proposal slots per run): (a) 48 frames drawn from the batch are bit-identical to the oracle's records,
    (b) a second run is byte-identical, (c) the result of a frame does not depend on its position in the batch."""
    rng = np.random.default_rng(5)
    uniq = [synth.make_frame(100000 + s) for s in range(100)]
    order = rng.permutation(1000) % 100
    frames = [uniq[i] for i in order]
    params = capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5)
    det = capi.Detector(params)
    bat = capi.Batch(det, frames)
    bat.run()
    first = bat.raw_out_bytes()
    tm = bat.timing()
    assert tm["n_slots"] > 25e6 and tm["n_valid"] > 5e6
    op = _oracle_params(params)
    ref_of = {}
    n = 0
    for f in rng.choice(1000, 48, replace=False):
        u = int(order[f])
        if u not in ref_of:
            ref_of[u] = oracle_py.detect_cuboid(uniq[u], op, atan2_mode=1)[0]
        ref, got = ref_of[u], bat.cuboids(int(f))
        for i in range(len(uniq[u]["boxes"])):
            assert len(got[i]) == len(ref[i]), (f, i)
            for a, b in zip(got[i], ref[i]):
                for key in CUBOID_KEYS:
                    assert _same(a[key], b[key]), (f, i, key, a[key], b[key])
                n += 1
    assert n >= 300
    # (c) every copy of a unique frame carries the same records wherever it sits in the batch
    per = len(first) // 1000
    seen = {}
    for f in range(1000):
        rec = first[f * per:(f + 1) * per]
        assert seen.setdefault(int(order[f]), rec) == rec, f
    bat.run()
    assert bat.raw_out_bytes() == first
    bat.close(); det.close()


def _bundled_frame_c1(sample_height=False):
    """BASELINE.json's C1: the reference's bundled frame -- detect_3d_cuboid/src/main.cpp:37-60 constants (K, T_wc, the
    1-based box made 0-based) and data/edge_detection/LSD/0000_edge.txt (271 segments).  The distance map would come
    from cv::Canny + cv::distanceTransform of the bundled JPEG (OpenCV, absent here); the exact L2 transform of the
    rasterised segments stands in for it on both sides."""
    from scipy import ndimage
    K = np.array([[529.5, 0, 365.0], [0, 529.5, 265.0], [0, 0, 1.0]])
    T = np.array([[1, 0.0011, 0.0004, 0], [0, -0.3376, 0.9413, 0], [0.0011, -0.9413, -0.3376, 1.35], [0, 0, 0, 1.0]])
    box = np.array([[188 - 1, 189 - 1, 201, 311, 0.88]])
    lines = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "detect_3d_cuboid_data", "0000_edge.txt"))
    rois = [synth.box_rois(box[0], 730, 530, sample_height)]
    maps = []
    for (l, t, w, h), _ in rois[0]:
        edge = synth._rasterise(lines, l, t, w, h)
        buf = np.zeros(h * w + w + 1, np.float32)
        buf[: h * w] = ndimage.distance_transform_edt(~edge).astype(np.float32).ravel()
        maps.append(buf)
    return dict(K=K, T_wc=T, boxes=box, lines=lines, rois=rois, maps=[maps], img_w=730, img_h=530)


def test_c1_bundled_reference_frame():
    """C1 (the reference's own runnable case): every proposal row, kept set, score and the final cuboid bit-identical
    to the oracle, at the reference's settings (6 deg yaw step: 111 valid of 320 slots), at the headline 0.5 deg step
    (1251 valid), with roll/pitch sampling (RP = 20: 1799 valid) and with height sampling + top-5."""
    fr = _bundled_frame_c1()
    assert _check([fr], capi.default_params(whether_sample_cam_roll_pitch=0)) == 111
    assert _check([fr], capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=0.5)) == 1251
    assert _check([fr], capi.default_params(whether_sample_cam_roll_pitch=1)) == 1799
    frh = _bundled_frame_c1(sample_height=True)
    assert _check([frh], capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=1, max_cuboid_num=5)) > 111
    n, tm = _check_final([fr, fr], capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=0.5))
    assert n == 2


def test_integer_outputs_against_the_libm_only_oracle_build():
    """The oracle build that shares NO code with the product (oracle/liboracle_detect_libm.so: -DORACLE_LIBM_ONLY, std::atan2
    like a build of the reference): everything the reference decides in integers -- which proposals are valid, their
    configuration / vanishing-point side / yaw sample / top-edge sample, which ones fuse_normalize_scores_v2 keeps and in which
    order, how many cuboids come back, the winner's integer corners and configuration -- must be the device's, at the
    reference's 6 degree sweep and at the 0.5 degree headline sweep.  (Doubles that pass through atan2 may differ in the last
    place between libm and the correctly rounded cs_atan2; the float distance sums contain no atan2 and must be identical.)"""
    n_cmp = n_win = 0
    for params, seeds in ((capi.default_params(whether_sample_cam_roll_pitch=0), range(7000, 7004)),
                          (capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=0.5), range(7100, 7103)),
                          (capi.default_params(whether_sample_cam_roll_pitch=1, max_cuboid_num=3), range(7200, 7202))):
        frames = [synth.make_frame(s, n_boxes=4, n_lines=300) for s in seeds]
        det = capi.Detector(params)
        bat = capi.Batch(det, frames, debug=True)
        bat.run()
        for f, fr in enumerate(frames):
            ref, dbg = oracle_py.detect_cuboid(fr, _oracle_params(params), atan2_mode=0, debug_cap=20000, libm_only=True)
            got = bat.cuboids(f)
            for i in range(len(fr["boxes"])):
                for k in range(len(fr["maps"][i])):
                    slot = 3 * i + k
                    V, nk = int(dbg["n_valid"][slot]), int(dbg["n_keep"][slot])
                    rows, _ = bat.debug_candidates(f, i, k)
                    ids, _ = bat.debug_kept(f, i, k)
                    assert rows.shape[0] == V and len(ids) == nk
                    assert np.array_equal(rows[:, :5], dbg["cand_rows"][slot][:V, :5])      # config, vp side, yaw, top id, distance error
                    assert np.array_equal(rows[:, 6:], dbg["cand_rows"][slot][:V, 6:])
                    assert np.allclose(rows[:, 5], dbg["cand_rows"][slot][:V, 5], rtol=0, atol=1e-13)
                    assert np.array_equal(ids, dbg["keep_ids"][slot][:nk])
                    n_cmp += V
                assert len(got[i]) == len(ref[i])
                for a, b in zip(got[i], ref[i]):
                    assert np.array_equal(a["box_corners_2d"], b["box_corners_2d"]) and np.array_equal(a["box_config_type"], b["box_config_type"])
                    assert a["rotY"] == b["rotY"] and a["edge_distance_error"] == b["edge_distance_error"]
                    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["scale"], b["scale"])
                    n_win += 1
        bat.close(); det.close()
    assert n_cmp > 20000 and n_win > 20
