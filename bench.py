#!/usr/bin/env python3
"""bench.py -- throughput of the detect_cuboid hot path on MI355X (BASELINE.json metric, config C2).

One "step" = one cs_batch_run() over a batch of synthetic KITTI-shaped frames whose inputs (distance maps,
line segments, boxes, cameras) are already resident in HBM: per frame 8 boxes x 181 yaw samples (0.5 deg
over +-45 deg) x ~10 top-edge samples x 2 configurations over ~400 line segments.
With --gpus N every rank runs its own batch (frames are independent units: no data-path collective,
weak scaling); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (candidate_kernel),
"cpu_baseline" = the CPU oracle timed on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1000, help="frames per batch (per GPU)")
    ap.add_argument("--unique", type=int, default=100, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0, help="worker threads of the host stages (0 = library default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch

    from cube_slam_wu_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- workload: distinct seeds per rank, tiled to the batch size (each copy gets its own HBM buffers)
    n_unique = max(1, min(args.unique, args.frames))
    uniq = [synth.make_frame(100000 * (rank + 1) + s) for s in range(n_unique)]
    frames = [uniq[i % n_unique] for i in range(args.frames)]
    params = capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5, host_threads=args.host_threads)
    det = capi.Detector(params, device=local_rank)
    bat = capi.Batch(det, frames)

    for _ in range(args.warmup):
        bat.run()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        bat.run()
        for k, v in bat.timing().items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = args.frames * args.steps * world
        value = total_frames / elapsed
        launches = max(1, int(acc["cand_kernel_launches"]))
        kern_ms = acc["cand_kernel_ms"] / launches
        alg_bytes = acc["cand_kernel_bytes"] / launches
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "frames/sec detect_cuboid", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: per-frame cuboid proposal sweep, 181 yaw x 8 boxes x ~400 line segments, 1241x376 KITTI-shaped",
                       "frames_per_batch_per_gpu": args.frames, "unique_frames": n_unique, "yaw_step_deg": 0.5,
                       "proposal_slots_per_frame": acc["n_slots"] / args.steps / args.frames,
                       "valid_proposals_per_frame": acc["n_valid"] / args.steps / args.frames, "parallelism": "frames sharded, no collective"},
            "roofline": {"kernel": "candidate_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": kern_ms},
            "stage_ms_per_step": {k: acc[k] / args.steps for k in acc if k.endswith("_ms")},
            "fallback_boxes_per_step": acc["n_fallback_boxes"] / args.steps,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py
            oracle_py.lib()
            op = oracle_py.default_params(yaw_step_deg=0.5)
            oracle_py.detect_cuboid(uniq[0], op, atan2_mode=0)  # warm
            n, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < args.cpu_seconds:
                oracle_py.detect_cuboid(uniq[n % n_unique], op, atan2_mode=0)
                n += 1
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same workload through oracle/detect_oracle.cpp (-O2, libm atan2, single thread) in %.1f s" % (n, dt)}
        print(json.dumps(out))
    bat.close()
    det.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
