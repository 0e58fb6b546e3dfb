import sys, numpy as np
sys.path.insert(0, '.')
from cube_slam_wu_amd import capi, synth
from oracle import oracle_py
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
fr = synth.make_frame(seed)
p = capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=step)
det = capi.Detector(p); bat = capi.Batch(det, [fr], debug=True); bat.run()
ref, dbg = oracle_py.detect_cuboid(fr, oracle_py.default_params(yaw_step_deg=step), atan2_mode=1, debug_cap=20000)
for i in range(len(fr['boxes'])):
    V = int(dbg['n_valid'][3*i]); rows, corners = bat.debug_candidates(0, i, 0)
    R = dbg['cand_rows'][3*i][:V]; Cc = dbg['cand_corners'][3*i][:V]
    print("box", i, "V", V, rows.shape)
    if rows.shape[0] != V: continue
    for c in range(9):
        bad = np.nonzero(rows[:, c] != R[:, c])[0]
        if len(bad): print("  col", c, "nbad", len(bad), "first", bad[:5], [(rows[b, c].hex(), R[b, c].hex()) for b in bad[:3]])
    badc = np.nonzero((corners != Cc).any(axis=1))[0]
    if len(badc):
        print("  corners nbad", len(badc), badc[:5])
        b = badc[0]; print("   ", [(corners[b, k].hex(), Cc[b, k].hex()) for k in range(16) if corners[b, k] != Cc[b, k]])
