"""Host-side timeline of the image-in loop (bench.py `image_in`): how long submit / refill / collect block, per step.
   python tools/image_in_probe.py [frames] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cube_slam_wu_amd import capi, synth

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
uniq = [synth.make_frame(100000 + s) for s in range(100)]
frames = [uniq[i % 100] for i in range(nf)]
H, W = int(uniq[0]["img_h"]), int(uniq[0]["img_w"])
rng = np.random.default_rng(17)
yy, xx = np.mgrid[0:H, 0:W]
imgs = []
for _ in range(8):      # (structured scenes, as bench.py's image_in entry: half planes + texture + noise)
    img = np.full((H, W), 90.0)
    for _ in range(25):
        a = rng.uniform(0, np.pi)
        img += np.where((xx - rng.uniform(0, W)) * np.cos(a) + (yy - rng.uniform(0, H)) * np.sin(a) > 0, rng.uniform(-40, 40), 0)
    img += 12 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 6, (H, W))
    imgs.append(np.clip(img, 0, 255).astype(np.uint8))
blocks = [torch.empty((nf, H, W), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
for q, b in enumerate(blocks):
    a = b.numpy()
    for f in range(nf):
        a[f] = imgs[(f + 3 * q) % 8]
prm = capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5)
det = capi.Detector(prm)
bat = capi.Batch(det, frames, grays=[blocks[0].numpy()[f] for f in range(nf)])
mode = sys.argv[3] if len(sys.argv) > 3 else "ahead"
bat.run(); bat.refill_gray(base_ptr=blocks[1].data_ptr()); bat.run(); bat.refill_gray(base_ptr=blocks[0].data_ptr()); bat.refill_wait(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    a = time.perf_counter()
    if mode == "ahead":      # the upload after next is queued behind the one the coming submit waits for: the copy stream never idles
        bat.refill_gray(base_ptr=blocks[(k + 1) % 2].data_ptr()); b = time.perf_counter(); bat.submit(); c = time.perf_counter()
    else:
        bat.submit(); b = time.perf_counter(); bat.refill_gray(base_ptr=blocks[(k + 1) % 2].data_ptr()); c = time.perf_counter()
    bat.collect(); d = time.perf_counter()
    tmg = bat.timing()
    print("step %d: first call %.2f ms  second call %.2f ms  collect %.2f ms   (step %.2f)   wait %.2f finalize %.2f | line_setup %.2f cand %.2f vp %.2f score %.2f rank %.2f" % (k, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (d - a) * 1e3,
          tmg["d2h_ms"], tmg["finalize_ms"], tmg["line_setup_ms"], tmg["cand_kernel_ms"], tmg["vp_kernel_ms"], tmg["score_kernel_ms"], tmg["rank_kernel_ms"]))
bat.run()        # (the last queued upload)
bat.refill_wait(); torch.cuda.synchronize()
print("%s: %.1f frames/s; timing %s" % (mode, nf * K / (time.perf_counter() - t0), {k: round(v, 2) for k, v in bat.timing().items() if k.endswith("_ms")}))
