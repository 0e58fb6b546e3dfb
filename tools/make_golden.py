#!/usr/bin/env python3
"""Record oracle outputs on seeded synthetic frames into tests/golden/detect_oracle_golden.json (hex floats)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cube_slam_wu_amd import synth  # noqa: E402
from oracle import oracle_py  # noqa: E402

cases = []
for seed, nb, nl, prm in [(9001, 4, 250, {}), (9002, 3, 200, {"yaw_step_deg": 0.5}), (9003, 2, 150, {"whether_sample_cam_roll_pitch": 1})]:
    fr = synth.make_frame(seed, n_boxes=nb, n_lines=nl)
    res, dbg = oracle_py.detect_cuboid(fr, oracle_py.default_params(**prm), atan2_mode=1, debug_cap=8000)
    winners = []
    for r in res:
        if not r:
            winners.append(None)
            continue
        c = r[0]
        winners.append(dict(box_corners_2d=c["box_corners_2d"].ravel().tolist(), rotY_hex=float(c["rotY"]).hex(),
                            normalized_error_hex=float(c["normalized_error"]).hex(), pos_hex=[float(v).hex() for v in c["pos"]]))
    cases.append(dict(seed=seed, n_boxes=nb, n_lines=nl, params=prm, n_valid=dbg["n_valid"][::3].tolist(), n_keep=dbg["n_keep"][::3].tolist(), winners=winners))
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "detect_oracle_golden.json")
with open(out, "w") as f:
    json.dump(dict(generator="tools/make_golden.py", cases=cases), f, indent=1)
print("wrote", out)
