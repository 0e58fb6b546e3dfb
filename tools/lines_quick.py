"""Batch throughput of the two segment producers alone (the line_producer entry of bench.py, without the rest of the bench):
   python tools/lines_quick.py [images per batch] [repeats] [host threads] [lsd|edlines]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ht = int(sys.argv[3]) if len(sys.argv) > 3 else 0
only = sys.argv[4] if len(sys.argv) > 4 else ""       # "lsd" / "edlines": that producer only
rng = np.random.default_rng(21)
H, W = 376, 1241
yy, xx = np.mgrid[0:H, 0:W]
imgs = []
for _ in range(min(8, n)):
    im = np.full((H, W), 95.0)
    for _ in range(30):
        a = rng.uniform(0, np.pi)
        im += np.where((xx - rng.uniform(0, W)) * np.cos(a) + (yy - rng.uniform(0, H)) * np.sin(a) > 0, rng.uniform(-45, 45), 0)
    imgs.append(np.clip(im + rng.normal(0, 4, im.shape), 0, 255).astype(np.uint8))
batch = [imgs[i % len(imgs)] for i in range(n)]
d = capi.Detector(capi.default_params(host_threads=ht), device=0)
for lsd in (False, True):
    if (only == "lsd" and not lsd) or (only == "edlines" and lsd):
        continue
    ref = d.detect_lines_batch(batch, 15.0, use_lsd=lsd)
    one = [d.detect_lines_batch([batch[i]], 15.0, use_lsd=lsd)[0] for i in range(min(8, n))]       # image by image = the unchunked path
    same = all(np.array_equal(ref[i], one[i]) for i in range(len(one))) and all(np.array_equal(ref[i], ref[i % len(imgs)]) for i in range(n))
    t0 = time.perf_counter(); dev = host = tot = 0.0
    for _ in range(reps):
        d.detect_lines_batch(batch, 15.0, use_lsd=lsd)
        t = d.lines_timing(use_lsd=lsd); dev += t["device_ms"]; host += t["host_ms"]; tot += t["total_ms"]
    dt = time.perf_counter() - t0
    print("%s: %d images per batch, %.0f images/s (wall incl. the python wrapper), library call %.2f ms, kernels %.3f ms, host stage (overlapped with the copies) %.2f ms, %.1f segments per image, batch == image-by-image: %s"
          % ("LSD" if lsd else "EDLines", n, n * reps / dt, tot / reps, dev / reps, host / reps, float(np.mean([len(x) for x in ref])), same))
d.close()
