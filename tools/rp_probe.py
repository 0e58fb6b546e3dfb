import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from cube_slam_wu_amd import capi, synth
uniq = [synth.make_frame(100000 + s) for s in range(100)]
prp = capi.default_params(whether_sample_cam_roll_pitch=1, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5)
d = capi.Detector(prp); b = capi.Batch(d, uniq)
b.run(); b.run()
t = b.timing()
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in t.items()})
