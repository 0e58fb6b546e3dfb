"""Why is the LSD batch call slower inside bench.py's process than alone?  Times the same call
   (a) in a fresh process, (b) with four more detectors (parked pools, pinned buffers) alive, (c) after those detectors ran detect batches,
   (d) (c) with the producer's pool limited to 16 / 8 threads.    python tools/lsd_busy_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth

rng = np.random.default_rng(21)
H, W = 376, 1241
yy, xx = np.mgrid[0:H, 0:W]
imgs = []
for _ in range(8):
    im = np.full((H, W), 95.0)
    for _ in range(30):
        a = rng.uniform(0, np.pi)
        im += np.where((xx - rng.uniform(0, W)) * np.cos(a) + (yy - rng.uniform(0, H)) * np.sin(a) > 0, rng.uniform(-45, 45), 0)
    imgs.append(np.clip(im + rng.normal(0, 4, im.shape), 0, 255).astype(np.uint8))
batch = [imgs[i % 8] for i in range(64)]


def measure(d, tag, reps=12):
    d.detect_lines_batch(batch, 15.0, use_lsd=True)
    tot, host, dev = [], [], []
    for _ in range(reps):
        d.detect_lines_batch(batch, 15.0, use_lsd=True)
        t = d.lines_timing(use_lsd=True)
        tot.append(t["total_ms"]); host.append(t["host_ms"]); dev.append(t["device_ms"])
    print("%-70s call %.2f ms (min %.2f)  host stage %.2f (min %.2f)  kernels %.3f" % (tag, np.median(tot), min(tot), np.median(host), min(host), np.median(dev)), flush=True)


print("cpus: affinity %d, cgroup cpu.max %s" % (len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"))
d = capi.Detector(capi.default_params(), device=0)
measure(d, "(a) fresh process, default pool")
others = [capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, yaw_step_deg=1.0), device=0) for _ in range(4)]
measure(d, "(b) + four idle detectors")
frames = [synth.make_frame(5000 + i) for i in range(10)]
bats = [capi.Batch(o, [frames[i % 10] for i in range(200)]) for o in others]
for _ in range(3):
    for b in bats:
        b.run()
measure(d, "(c) + after they ran detect batches (resident batches, parked pools)")
time.sleep(0.5)
measure(d, "(c') the same after 0.5 s of idling")
for ht in (16, 8, 32):
    dd = capi.Detector(capi.default_params(host_threads=ht), device=0)
    measure(dd, "(d) a detector with host_threads=%d, same process" % ht)
    dd.close()
for b in bats:
    b.close()
for o in others:
    o.close()
measure(d, "(e) others closed")
d.close()
