"""The distance-map front end alone (bench.py's `edge_front_end` workload: the C2 batch's 8000 ROIs over 16 synthetic gray images):
device time per call, and -- with `check` -- bit equality of a sample of ROIs' maps with oracle/edge_oracle.cpp.
   python tools/edge_prof.py [reps] [check]
Under `rocprofv3 --kernel-trace --stats` the per-kernel split (tools/edge_prof.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
check = len(sys.argv) > 2 and sys.argv[2] == "check"
uniq = [synth.make_frame(100000 + s) for s in range(100)]
frames = [uniq[i % 100] for i in range(1000)]
H, W = int(uniq[0]["img_h"]), int(uniq[0]["img_w"])
rng = np.random.default_rng(7)
yy, xx = np.mgrid[0:H, 0:W]
grays = []
for _ in range(16):
    img = np.full((H, W), 90.0)
    for _ in range(25):
        a = rng.uniform(0, np.pi)
        img += np.where((xx - rng.uniform(0, W)) * np.cos(a) + (yy - rng.uniform(0, H)) * np.sin(a) > 0, rng.uniform(-40, 40), 0)
    img += 12 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 6, (H, W))
    grays.append(np.clip(img, 0, 255).astype(np.uint8))
rois, roi_img = [], []
for f, fr in enumerate(frames):
    for rr in fr["rois"]:
        for (l, t, w, h), _ in rr:
            rois.append((l, t, w, h)); roi_img.append(f % 16)
det = capi.Detector(capi.default_params())
det.edge_distance_maps_time(grays, rois[:64], roi_img[:64])
px = sum(w * h for _, _, w, h in rois)
for r in range(reps):
    ms = det.edge_distance_maps_time(grays, rois, roi_img)
    print("rep %d: %.3f ms for %d ROIs (%.1f M pixels): %.0f GB/s algorithmic (15 B / pixel)" % (r, ms, len(rois), px / 1e6, px * 15 / (ms * 1e-3) / 1e9))
if check:
    from oracle import edge_oracle_py as E
    E.lib()
    bad = 0
    sel = list(range(0, len(rois), 97))[:96] + [int(np.argmax([w * h for _, _, w, h in rois])), int(np.argmin([w * h for _, _, w, h in rois]))]
    for img_id in sorted(set(roi_img[k] for k in sel)):
        ks = [k for k in sel if roi_img[k] == img_id]
        maps = det.edge_distance_maps(grays[img_id], [rois[k] for k in ks])
        for k, m in zip(ks, maps):
            want = E.edge_distance_map(grays[img_id], rois[k])
            if m.shape != want.shape or m.tobytes() != want.tobytes():
                bad += 1
    print("check: %d ROIs against the oracle, %d differ" % (len(sel), bad))
    # the batch path (n_rois > 1024: fused hysteresis) against the single-image path on the same ROIs
    sys.exit(1 if bad else 0)
