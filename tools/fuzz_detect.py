"""Differential run of the production path (no debug switches) against the CPU restatement on many random frames and parameter sets:
every field of every cuboid record must be equal, bit for bit.  python tools/fuzz_detect.py [frames per setting]"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth
from oracle import oracle_py

KEYS = ["pos", "scale", "rotY", "box_config_type", "box_corners_2d", "box_corners_3d_world", "rect_detect_2d", "edge_distance_error", "edge_angle_error",
        "normalized_error", "skew_ratio", "down_expand_height", "camera_roll_delta", "camera_pitch_delta"]
n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 60
settings = []
for rp, hs, step, kmax, cfgs in itertools.product((0, 1), (0, 1), (6.0, 2.0, 0.5), (1, 3), ((1, 1), (1, 0), (0, 1))):
    if step == 0.5 and (rp or kmax == 3 or cfgs != (1, 1)):
        continue                      # the dense sweep once per height setting: the oracle takes 6 ms per frame there
    if cfgs != (1, 1) and (rp or hs):
        continue
    settings.append((rp, hs, step, kmax, cfgs))
tot_frames = tot_cub = bad = 0
t0 = time.time()
for si, (rp, hs, step, kmax, cfgs) in enumerate(settings):
    p = capi.default_params(whether_sample_cam_roll_pitch=rp, whether_sample_bbox_height=hs, yaw_range_deg=45.0, yaw_step_deg=step, max_cuboid_num=kmax,
                            consider_config_1=cfgs[0], consider_config_2=cfgs[1])
    op = oracle_py.default_params(consider_config_1=cfgs[0], consider_config_2=cfgs[1], whether_sample_cam_roll_pitch=rp, whether_sample_bbox_height=hs,
                                  max_cuboid_num=kmax, nominal_skew_ratio=p.nominal_skew_ratio, max_cut_skew=p.max_cut_skew, yaw_range_deg=45.0, yaw_step_deg=step)
    n = n_per if step > 0.5 else max(4, n_per // 6)
    if rp:
        n = max(4, n // 3)
    frames = [synth.make_frame(7919 * (si + 1) + 31 * k, sample_height=bool(hs)) for k in range(n)]
    det = capi.Detector(p)
    bat = capi.Batch(det, frames)
    bat.run()
    for f, fr in enumerate(frames):
        ref, _ = oracle_py.detect_cuboid(fr, op, atan2_mode=1)
        got = bat.cuboids(f)
        for i in range(len(fr["boxes"])):
            if len(got[i]) != len(ref[i]):
                bad += 1; print("count mismatch", (rp, hs, step, kmax, cfgs), f, i, len(got[i]), len(ref[i])); continue
            for a, b in zip(got[i], ref[i]):
                tot_cub += 1
                for key in KEYS:
                    x, y = np.asarray(a[key]), np.asarray(b[key])
                    if not (np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)):
                        bad += 1; print("field mismatch", (rp, hs, step, kmax, cfgs), f, i, key, x, y)
    tot_frames += n
    bat.close(); det.close()
print("%d settings, %d frames, %d cuboid records compared, %d mismatches, %.0f s" % (len(settings), tot_frames, tot_cub, bad, time.time() - t0))
sys.exit(1 if bad else 0)
