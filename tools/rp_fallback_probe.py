"""How many boxes of the roll/pitch-sampling stress batch go to the host's exact ranking, and what the host stages cost: python tools/rp_fallback_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cube_slam_wu_amd import capi, synth
uniq = [synth.make_frame(100000 + s) for s in range(50)]
frames = [uniq[i % 50] for i in range(100)]
prp = capi.default_params(whether_sample_cam_roll_pitch=1, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5)
det = capi.Detector(prp)
bat = capi.Batch(det, frames)
bat.run()
t0 = time.perf_counter()
for _ in range(5):
    bat.run()
dt = (time.perf_counter() - t0) / 5
tm = bat.timing()
print("ms per batch of 100 frames: %.2f" % (dt * 1e3))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
