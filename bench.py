#!/usr/bin/env python3
"""bench.py -- throughput of the detect_cuboid hot path on MI355X (BASELINE.json metric, config C2).

One "step" = one cs_batch_run() over a batch of synthetic KITTI-shaped frames whose inputs (distance maps,
line segments, boxes, cameras) are already resident in HBM: per frame 8 boxes x 181 yaw samples (0.5 deg
over +-45 deg) x ~10 top-edge samples x 2 configurations over ~400 line segments.
With --gpus N every rank runs its own batch (frames are independent units: no data-path collective,
weak scaling); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (score_kernel),
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
# The library runs five streams per detector (main, corner construction, crowded line setup, tie fetches, image upload) and the bench three detectors side
# by side; the ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default -- and streams that share a queue
# serialise.  Eight queues: 497-500 k -> 530 k frames/s at the contract run, steady 510-512 k -> 547-552 k (2: 453-460 k; 16: erratic) -- INTEGRATION.md.
# Read by the runtime when it initialises, so it is set before anything touches HIP; a value already in the environment wins and is reported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP64_FLOP_PER_PROPOSAL = 1700.0   # 88 map samples x 6, six cs_atan2 (light path ~90 each), the 3D lift; + ~300 for the corners the scorer now rebuilds (13 squared lengths, 6 line intersections, 3 ray hits)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak = half the FP32 vector peak of MI355X_MICROARCH.md (157.3)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _pmc_child_pass(args, n_unique, counters, kernels):
    """One rocprofv3 counter pass (--kernel-trace + --pmc only, as MI355X_MICROARCH.md prescribes) over a short child run of the same workload:
    {kernel: {counter: average per launch}} for the kernels named, None when the pass cannot run."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    # never a profiler inside a profiler: when this process itself runs under rocprofv3 / rocprofiler, report the tracked figure
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    d = tempfile.mkdtemp(prefix="cs_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "2", "--warmup", "1", "--inflight", "1", "--depth", "1", "--steady-steps", "0", "--latency-calls", "0", "--image-in-steps", "0", "--lines-images", "0", "--ba", "none", "--no-cpu-baseline", "--no-edge", "--rp-frames", "0", "--no-measure-traffic",
               "--frames", str(args.frames), "--unique", str(n_unique)]
        subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp", "CS_BENCH_CHILD": "1"}, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
        tot, cnt = {}, {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    kn = r.get("Kernel_Name", "")
                    for k in kernels:
                        if k in kn and r.get("Counter_Name") in counters:
                            key = (k, r["Counter_Name"])
                            tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"])
                            cnt[key] = cnt.get(key, 0) + 1
        if not tot:
            return None
        out = {}
        for (k, c), v in tot.items():
            out.setdefault(k, {})[c] = v / cnt[(k, c)]
        return out
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


SIMDS_PER_SE = 32   # MI355X: 8 XCDs x 4 shader engines, 8 CUs (32 SIMDs) per shader engine; SQ_BUSY_CYCLES is summed over the 32 engines


def measure_valu_issue_live(args, n_unique):
    """VALU issue utilisation of the two FP64-bound sweep kernels from one SQ counter pass over a child run of the same workload:
    frac = 4 x SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES x 32) -- SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD's arbiter spends issuing VALU
    instructions, summed over all SIMDs (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* are in quad-cycles); SQ_BUSY_CYCLES counts the cycles a
    shader engine's SQ is busy, summed over the 32 engines, each with 32 SIMDs.  None when the pass cannot run."""
    ctrs = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")
    res = _pmc_child_pass(args, n_unique, ctrs, ("score_kernel", "candidate_compact_kernel", "vp_support_kernel"))
    if not res:
        return None
    out = {"formula": "4 x SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES x %d SIMDs per shader engine): the share of the busy SIMD cycles in which a VALU instruction issues (counters per launch, one rocprofv3 --pmc pass, kernel-trace only, child run of this workload)" % SIMDS_PER_SE}
    for k, c in res.items():
        if c.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_ACTIVE_INST_VALU" in c:
            e = {"frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (c["SQ_BUSY_CYCLES"] * SIMDS_PER_SE)}
            if c.get("SQ_WAVES", 0) > 0 and "SQ_INSTS_VALU" in c:
                e["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            if c.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in c:
                e["wave_cycles_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            e["counters_per_launch"] = {n: c[n] for n in sorted(c)}
            out[k] = e
    return out if len(out) > 1 else None


def measure_traffic_live(args, n_unique):
    """HBM bytes per score_kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only, as
    MI355X_MICROARCH.md prescribes) over a short child run of the same workload; FETCH_SIZE x 2 + WRITE_SIZE x 1 (KiB -> bytes), the
    factors of profiles/r2_pmc_calibration.json.  None when rocprofv3 is not available or a pass fails."""
    avg = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        r = _pmc_child_pass(args, n_unique, (ctr,), ("score_kernel",))
        if not r or ctr not in r.get("score_kernel", {}):
            return None
        avg[ctr] = r["score_kernel"][ctr]
    return avg["FETCH_SIZE"] * 1024 * 2 + avg["WRITE_SIZE"] * 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--frames", type=int, default=1000, help="frames per batch (per GPU)")
    ap.add_argument("--unique", type=int, default=100, help="distinct synthetic frames generated (tiled to --frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0, help="worker threads of the host stages (0 = library default)")
    ap.add_argument("--ba", default="C4", choices=["C4", "C3", "none"], help="also time the g2o BA path (second half of the BASELINE metric)")
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--shard-probe", type=int, default=8, help="N = 1 only: time one middle rank of this many (separator-mode sharded BA, loop-back transport) for the multi-GPU projection; 0 = skip")
    ap.add_argument("--chunks", type=int, default=0, help="cut the batch into this many chunks of the two-slot host/GPU pipeline (0 = library default)")
    ap.add_argument("--no-edge", action="store_true", help="skip the distance-map front end (Canny + distance transform) timing")
    ap.add_argument("--rp-frames", type=int, default=100, help="frames of the roll/pitch-sampling stress variant (RP = 25 poses per box, the reference's class default); 0 = skip")
    ap.add_argument("--lines-images", type=int, default=64, help="images per batch of the line-producer entry (0 = skip)")
    ap.add_argument("--rp-inflight", type=int, default=4, help="batches in flight of the roll/pitch-sampling stress variant")
    ap.add_argument("--image-in-steps", type=int, default=12, help="steps of the image-in entry (upload + edge front end + sweep inside the clock); 0 = skip")
    ap.add_argument("--latency-calls", type=int, default=200, help="calls per entry point of the single-call latency report (0 = skip)")
    ap.add_argument("--depth", type=int, default=2, help="batches per pipeline: the next one is submitted (packed + queued) before the current one is collected")
    ap.add_argument("--steady-steps", type=int, default=200, help="steps of the steady-state measurement reported beside the contract run (0 = skip)")
    ap.add_argument("--inflight", type=int, default=4, help="(4 since eight hardware queues -- round 6, last: 4 x 2 550 k / 570 k steady against 3 x 2 521 k / 541 k; with the runtime's four queues 3 x 2 was the best) batches in flight per GPU: each has its own detector (streams, worker pool) and is driven by its own host thread, "
                    "so one batch's host stages (packing, record writing) overlap the other's sweep on the device")
    ap.add_argument("--no-measure-traffic", action="store_true", help="do not collect roofline.traffic in this run (two rocprofv3 --pmc passes -- FETCH_SIZE, WRITE_SIZE; "
                    "kernel-trace only -- over a short child run of the same workload, ~20 s at N = 1); report the figure of the last tools/profile_round.sh "
                    "(profiles/pmc_traffic.json) instead, which is also the fallback when rocprofv3 is missing or a pass fails")
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
    # The line proves what it ran with: every CS_* variable of the environment is printed with it (the library's diagnostic switches and this
    # script's own), and the one switch that changes RESULTS -- CS_DETECT_SKIP, honoured only by a diagnostic build (make DIAG=1) -- is refused.
    env_overrides = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("CS_") and k != "CS_BENCH_CHILD"}
    if "CS_DETECT_SKIP" in os.environ:
        raise SystemExit("bench.py: CS_DETECT_SKIP is set (kernels left out of the sweep: results meaningless) -- refusing to measure")
    if capi.lib().cs_diag_build():
        raise SystemExit("bench.py: libcubeslam_hip.so is a diagnostic build (make DIAG=1) -- rebuild without DIAG before measuring")
    # CS_BENCH_SHARE_GPU=1: functional check of the N > 1 code path on a box with fewer GPUs than ranks (every rank on
    # device 0, gloo instead of RCCL, which refuses two ranks on one device); never a performance number
    share_gpu = os.environ.get("CS_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
        # the banded solver's persistent kernels need all their workgroups co-resident: several processes on one device can
        # only promise that for the small two-front teams (26 workgroups each), not for the nested order (106 each)
        os.environ.setdefault("CS_BAND_TWO_FRONTS", "1")
    torch.cuda.set_device(local_rank)
    # Control plane (barriers, the max over ranks, the 128 bytes of the RCCL id): torch.distributed over gloo, on the CPU.  The data
    # path's collectives are issued by libcubeslam_hip on its own RCCL communicator (cs_ba_comm_init), so exactly one RCCL is ever
    # initialised in this process -- the one the library links -- and torch's bundled copy stays untouched.
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- workload: distinct seeds per rank, tiled to the batch size (each copy gets its own HBM buffers)
    n_unique = max(1, min(args.unique, args.frames))
    uniq = [synth.make_frame(100000 * (rank + 1) + s) for s in range(n_unique)]
    frames = [uniq[i % n_unique] for i in range(args.frames)]
    # host stages run on a worker pool per rank: with N ranks on one node the pools share the host's cores
    # Batches in flight: every one is a full copy of the workload with its own detector; the K steps are handed out first come, first served.
    import threading
    inflight = max(1, args.inflight)
    host_threads = args.host_threads
    granted = os.cpu_count() or 64
    quota = False
    try:   # a cgroup CPU quota is what the host stages of all ranks really share
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            granted, quota = max(1, min(granted, int(q) // int(per))), True
    except (OSError, ValueError):
        pass
    if world > 1:
        # N ranks on one node share the node's cores: a pipeline's host stages (packing, records, the tie boxes' exact ranking) want about
        # four of them, so a rank only keeps as many batches in flight as its share of the cores can feed
        inflight = max(1, min(inflight, granted // (4 * world)))
    if host_threads == 0 and world * inflight > 1:
        cpus = 3 * granted if quota else granted     # under a quota: three threads per granted CPU, as the library's default
        host_threads = max(8, min(64, cpus // (world * inflight)))
    params = capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5, host_threads=host_threads)
    # every pipeline (host thread + detector) owns `depth` batches: it submits the next one (cs_batch_submit: host packing + the sweep
    # queued on the detector's streams) before it collects the current one, so the device never waits for a pipeline's host stages
    depth = max(1, args.depth)
    dets = [capi.Detector(params, device=local_rank) for _ in range(inflight)]
    bats = [capi.Batch(dets[p // depth], frames, pipeline_chunks=(args.chunks if args.chunks > 0 else 1)) for p in range(inflight * depth)]
    det, bat = dets[0], bats[0]

    # warm-up: every pipeline alone (also the isolated kernel timings: nothing else runs on the device)
    iso = {}
    for p in range(inflight * depth):
        for _ in range(max(1, args.warmup)):     # (at least once: a batch's device and pinned buffers are sized by its first run -- set-up, whatever W says)
            bats[p].run()
            if p == 0:
                for k, v in bats[p].timing().items():
                    iso[k] = iso.get(k, 0) + v
    accs = [dict() for _ in range(inflight)]
    errs = []

    step_lock = threading.Lock()
    steps_taken = [0]
    last_collected = [None]
    warm_alone_runs = max(1, args.warmup) * inflight * depth
    warm_concurrent_steps = 0

    def take_step():   # the K steps are a shared queue: a pipeline that finishes early takes the next one (no idle tail at small K)
        with step_lock:
            if steps_taken[0] >= args.steps:
                return False
            steps_taken[0] += 1
            return True

    def drive(p):
        try:
            free, queued = [bats[p * depth + q] for q in range(depth)], []
            while True:
                while free and take_step():      # a step = one batch through the whole sweep; its collect below is inside the timed region too
                    bq = free.pop(0)
                    bq.submit()
                    queued.append(bq)
                if not queued:
                    break
                bq = queued.pop(0)
                bq.collect()
                last_collected[0] = bq      # (whichever pipeline collects last: the batch of the region's last finished step)
                free.append(bq)
                for k, v in bq.timing().items():
                    accs[p][k] = accs[p].get(k, 0) + v
        except Exception as e:   # surfaced after the join
            errs.append(e)
    def cgroup_cpu():      # (usage_usec, nr_throttled, throttled_usec) of this container's CPU quota, None without cgroup v2
        try:
            kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
            return int(kv.get("usage_usec", 0)), int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
        except (OSError, ValueError):
            return None
    # The warm-up steps once more the way the timed steps run: all pipelines side by side, four rounds of the batches in flight (untimed).  The loop
    # above runs every batch ALONE (it also gives the isolated kernel timings); a device that has only seen one batch at a time entered the timed region
    # cold -- a fixed ~16 ms per run whatever K (400 steps 2.74 ms per step, 40: 3.15, 20: 3.6 on one box) -- and the contract run read 8-20 % under the
    # steady rate for that reason, not for its pipe's fill and drain: with this warm-up 40 steps went 326-329 k -> 359-382 k frames/s on the same box,
    # 20 steps 313-338 k -> 347-390 k.  CS_BENCH_CONCURRENT_WARMUP=0: the old behaviour.
    if args.warmup > 0 and inflight > 1 and os.environ.get("CS_BENCH_CONCURRENT_WARMUP", "1") != "0":
        contract_steps = args.steps
        args.steps = max(args.warmup, inflight * depth) * int(os.environ.get("CS_BENCH_WARMUP_ROUNDS", "4"))
        warm_concurrent_steps = args.steps
        steps_taken[0] = 0
        accs_saved, accs[:] = list(accs), [dict() for _ in range(inflight)]
        th = [threading.Thread(target=drive, args=(p,)) for p in range(inflight)]
        [t.start() for t in th]
        [t.join() for t in th]
        accs[:] = accs_saved
        args.steps = contract_steps
        steps_taken[0] = 0
        if errs:
            raise errs[0]
    # (the interpreter's cyclic garbage collector stays out of the timed regions of this script, as in `timeit`: a generation-2 pass over the
    # script's own frame lists took 90 ms of one image_in step in a round-6 run -- Python housekeeping, not the library's work)
    import gc
    gc.collect(); gc.disable()
    barrier()
    cg0 = cgroup_cpu()
    t0 = time.perf_counter()
    if inflight == 1:
        drive(0)
    else:
        th = [threading.Thread(target=drive, args=(p,)) for p in range(inflight)]
        [t.start() for t in th]
        [t.join() for t in th]
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    cg1 = cgroup_cpu()
    # parity of the TIMED region's own output: the records the last timed step wrote, kept before any later leg reuses the batch
    N_PARITY = min(16, args.frames)
    timed_records = [last_collected[0].cuboids(f) for f in range(N_PARITY)] if (rank == 0 and last_collected[0] is not None) else None
    host_cpu = None
    if cg0 and cg1:      # how much CPU the timed region took and whether the container's quota throttled it (a throttled period stalls every thread)
        host_cpu = {"granted_cpus": granted if quota else None, "host_threads_per_pipeline": host_threads, "cpu_seconds_per_wall_second": (cg1[0] - cg0[0]) * 1e-6 / max(elapsed, 1e-9),
                    "throttled_periods": cg1[1] - cg0[1], "throttled_thread_ms": (cg1[2] - cg0[2]) * 1e-3}
    if errs:
        raise errs[0]
    acc = {}
    for a in accs:
        for k, v in a.items():
            acc[k] = acc.get(k, 0) + v
    # steady state: the contract run's K steps include the pipelines' fill and drain (4 batches in flight: 15-30 % at K = 10-20); the
    # same loop over >= 200 steps is the rate a long-running caller sees.  Reported beside `value`, never instead of it.
    steady = None
    if args.steady_steps > 0:
        contract_steps = args.steps
        args.steps = args.steady_steps
        steps_taken[0] = 0
        scratch = [dict() for _ in range(inflight)]
        accs_saved, accs[:] = list(accs), scratch
        gc.collect(); gc.disable()
        barrier()
        cg2 = cgroup_cpu()
        t1 = time.perf_counter()
        if inflight == 1:
            drive(0)
        else:
            th = [threading.Thread(target=drive, args=(p,)) for p in range(inflight)]
            [t.start() for t in th]
            [t.join() for t in th]
        barrier()
        el2 = max_over_ranks(time.perf_counter() - t1)
        gc.enable()
        cg3 = cgroup_cpu()
        accs[:] = accs_saved
        args.steps = contract_steps
        if errs:
            raise errs[0]
        steady = {"steps": args.steady_steps, "value": args.frames * args.steady_steps * world / el2, "unit": "frames/s", "ms_per_step": el2 / args.steady_steps * 1e3}
        if cg2 and cg3:
            steady["host_cpu"] = {"cpu_seconds_per_wall_second": (cg3[0] - cg2[0]) * 1e-6 / max(el2, 1e-9), "throttled_periods": cg3[1] - cg2[1], "throttled_thread_ms": (cg3[2] - cg2[2]) * 1e-3}

    # ---- stress variant of SURVEY 8(d): whether_sample_cam_roll_pitch = 1 (the reference class's default, detect_3d_cuboid.h:110;
    # main_obj.cpp:623 uses it from the second frame on): 5 x 5 roll / pitch samples around the camera pose, i.e. 25x the proposals
    # per box, boxes of a frame processed in rounds because the camera yaw carries over from box to box (box_proposal_detail.cpp:180).
    rp_out = None
    if rank == 0 and args.rp_frames > 0:
        prp = capi.default_params(whether_sample_cam_roll_pitch=1, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5, host_threads=host_threads)
        # like the headline: several (detector, batch) pairs in flight, one host thread each -- a batch's boxes go through the device in
        # rounds (the camera yaw carries from box to box) with a host decision between rounds, and other batches' rounds fill those gaps
        n_if = max(1, args.rp_inflight)
        drps = [capi.Detector(prp, device=local_rank) for _ in range(n_if)]
        brps = [capi.Batch(drps[q], frames[:args.rp_frames]) for q in range(n_if)]
        for bq in brps:
            bq.run()
        torch.cuda.synchronize()
        n_rp = 3 * n_if
        tacc = {}
        rp_lock = threading.Lock()
        rp_left = [n_rp]

        def rp_drive(q):
            while True:
                with rp_lock:
                    if rp_left[0] <= 0:
                        return
                    rp_left[0] -= 1
                brps[q].run()
                tq = brps[q].timing()
                with rp_lock:
                    for k, v in tq.items():
                        tacc[k] = tacc.get(k, 0) + v
        t1 = time.perf_counter()
        thr = [threading.Thread(target=rp_drive, args=(q,)) for q in range(n_if)]
        [t.start() for t in thr]
        [t.join() for t in thr]
        dt = time.perf_counter() - t1
        rp_out = {"what": "C2 with roll/pitch sampling (RP = 25 camera poses per box, 0.5 deg yaw step): ~25x the proposals of the headline sweep",
                  "frames_per_batch": args.rp_frames, "batches_in_flight": n_if, "value": args.rp_frames * n_rp / dt, "unit": "frames/s",
                  "valid_proposals_per_frame": tacc["n_valid"] / n_rp / args.rp_frames, "proposal_slots_per_frame": tacc["n_slots"] / n_rp / args.rp_frames,
                  "stage_ms_per_batch": {k: tacc[k] / n_rp for k in tacc if k.endswith("_ms")}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py
            oprp = oracle_py.default_params(yaw_step_deg=0.5, whether_sample_cam_roll_pitch=1)
            t1 = time.perf_counter()
            n = 0
            while time.perf_counter() - t1 < 4.0:
                oracle_py.detect_cuboid(uniq[n % n_unique], oprp, atan2_mode=0)
                n += 1
            rp_out["cpu_oracle_frames_per_s"] = n / (time.perf_counter() - t1)
        for bq in brps:
            bq.close()
        for dq in drps:
            dq.close()

    # ---- second half of the metric: LM iterations/s of the BA path (C4: 1k cams / 200k points / 500 cuboids).
    # N > 1: the landmarks are sharded by camera subsequence; one RCCL all-reduce of [S | b_schur] per damped solve.
    def ba_leg():
        ba_out = None
        if args.ba != "none":
            from cube_slam_wu_amd import synth_ba
            nc, npt, no = (1000, 200000, 500) if args.ba == "C4" else (200, 20000, 50)
            pr = synth_ba.make_problem(n_cams=nc, n_points=npt, n_cuboids=no, seed=42)
            # structure phase = what g2o does inside optimize() before the first iteration (index mapping sparse_optimizer.cpp:166-190,
            # BlockSolver::buildStructure block_solver.hpp:142-295): packing + upload of the problem, the two edge orderings, the Schur
            # pattern, the solver ordering.  Timed on its own and reported next to the steady-state rate.
            torch.cuda.synchronize()
            ts = time.perf_counter()
            P = capi.ba_from_dict(pr, device=local_rank)
            use_cb = share_gpu or os.environ.get("CS_BA_COMM") == "callback"   # RCCL refuses two ranks on one device
            if world > 1:
                if use_cb:
                    P.set_shard(rank, world)
                else:   # the library's own RCCL communicator: rank 0 draws the id, torch.distributed carries the 128 bytes
                    idt = torch.zeros(128, dtype=torch.uint8)
                    if rank == 0:
                        idt = torch.tensor(list(capi.comm_unique_id()), dtype=torch.uint8)
                    dist.broadcast(idt, 0)
                    P.comm_init(rank, world, bytes(idt.tolist()))
            P.sizes()                     # forces the structure phase
            torch.cuda.synchronize()
            structure_ms = (time.perf_counter() - ts) * 1e3
            # (the figure above is ONE sample on the process's first BA handle -- 18.5 to 30.7 ms over the round's boxes for the same code; two more
            # handles over the same arrays, built and dropped, say what a second and a third build cost in this process)
            structure_again = []
            if world == 1:
                import gc as _gc
                for _ in range(2):
                    _gc.collect()
                    torch.cuda.synchronize()
                    t_s = time.perf_counter()
                    P2 = capi.ba_from_dict(pr, device=local_rank)
                    P2.sizes()
                    torch.cuda.synchronize()
                    structure_again.append((time.perf_counter() - t_s) * 1e3)
                    P2.close()
            ar = capi.torch_allreduce(dist, torch.device("cuda", local_rank)) if (world > 1 and use_cb) else None
            run = (lambda n: P.optimize_sharded(n, ar)) if world > 1 else (lambda n: P.optimize(n))
            run(1)  # warm-up: first-launch costs (code object load, rocSOLVER handles)
            barrier()
            tb = time.perf_counter()
            n_it = run(args.ba_iters)
            barrier()
            ba_el = max_over_ranks(time.perf_counter() - tb)
            structure_ms = max_over_ranks(structure_ms)
            # `value` above is timed the way a caller runs the library: the stage split OFF (cs_ba_set_stage_timing; g2o's batch statistics are off
            # by default too, core/sparse_optimizer.cpp:379-397).  The split -- and with it the durations behind ba.roofline -- comes from the SAME
            # iterations run once more with the phase marks on the stream (HIP events; ~6 us of dispatch gap each, and no speculative linearisation
            # behind a trial), on a fresh handle of the same problem from the same initial estimates.
            if world == 1:
                P.close()
                P = capi.ba_from_dict(pr, device=local_rank)
                P.stage_timing(True)
                P.sizes()
                run = lambda n: P.optimize(n)
                run(1)
            else:
                P.stage_timing(True)
            t_before = P.timing()
            barrier()
            tb2 = time.perf_counter()
            n_it_split = run(args.ba_iters)
            barrier()
            ba_el_split = max_over_ranks(time.perf_counter() - tb2)
            tm = P.timing()
            d = {k: tm[k] - t_before[k] for k in tm if k.endswith("_ms")}
            nlin = max(1, tm["n_linearizations"] - t_before["n_linearizations"])
            nsol = max(1, tm["n_solves"] - t_before["n_solves"])
            build_ms = d["linearize_ms"] / nlin + d["reduce_ms"] / nsol
            sinfo = P.shard_info()
            sharding = "none"
            if world > 1:
                sharding = ("separator mode: every rank owns a column range of the banded reduced system (landmarks, cuboids and odometry edges by their lowest column), factorises its interior only; "
                            "per LM trial one all-gather of the separator messages (3 w^2 + 2 w doubles per rank), one all-reduce of the solution vector and one of [chi2, scale, failure flag]"
                            if sinfo["sep_mode"] else "landmarks by camera subsequence; per LM trial one all-reduce of [S | b_schur] + one of [chi2, scale, failure flag], replicated factorisation")
                sharding += ("; collectives issued by the library on its own RCCL communicator and stream" if not use_cb else "; collectives through a torch.distributed callback (host round trips)")
            ba_out = {"metric": "BA LM iterations/sec", "value": n_it / ba_el, "unit": "iters/s", "config": args.ba, "cams": nc, "points": npt, "cuboids": no,
                      "projection_edges": int(len(pr["e_pt"])), "iterations": int(n_it), "lm_trials": int(nsol), "sharding": sharding,
                      "ms_per_iteration": ba_el / max(1, n_it) * 1e3,
                      "with_stage_split": {"what": "the same run with cs_ba_set_stage_timing(1): the HIP-event phase marks behind stage_ms_per_iteration and ba.roofline (a caller who does not read the split leaves it off: `value`)",
                                           "value": n_it_split / ba_el_split, "iterations": int(n_it_split), "ms_per_iteration": ba_el_split / max(1, n_it_split) * 1e3},
                      "reduced_solve": (lambda pth, bo: {"path": pth[0], "unknowns": P.reduced_size()[0], "bandwidth": pth[1], "block_cyclic_reduction": bo[0], "levels": bo[1],
                                                         "what": "block cyclic reduction over blocks of 128 unknowns (bcr_kernels.hip)" if bo[0] else "see cs_ba_solver_path"})(P.solver_path(detail=True), P.band_order()),
                      "structure_ms": structure_ms, "structure_ms_second_and_third_handle": structure_again,
                      "value_including_structure": n_it / (ba_el + structure_ms * 1e-3),
                      "stage_ms_per_iteration": {k: v / max(1, n_it_split) for k, v in d.items()},
                      "build_only_ms_per_iteration": (d["linearize_ms"] + d["reduce_ms"] + d["errors_ms"]) / max(1, n_it_split),
                      "roofline": {"kernels": "linearise (ba_lin_cam, ba_*_edge) + Schur build with the landmark side linearised inside it (ba_lin_schur, ba_cam_rhs, ba_schur_gather, ba_cub_elim)", "bound": "hbm",
                                   "achieved": tm["linearize_bytes"] / (build_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": tm["linearize_bytes"] / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_linearisation": tm["linearize_bytes"],
                                   "ms_linearise_plus_schur": build_ms}}
            # (SURVEY 8(d)'s 136 B per projection edge include the edge's own 2 x 2 information matrix and intrinsics, 32 B each.  This synthetic
            # problem gives every edge the same ones -- one camera, one sigma -- and the library then reads them as 4-double constants
            # (BaView::info_u / intr_u): the bytes such a graph really needs are 64 B per edge fewer, and the fraction against THAT figure is the
            # stricter one.  Reported beside the survey's figure, which stays the line's `frac`.)
            uni = int((pr["e_info"] == pr["e_info"][0]).all()) + int((pr["e_intr"] == pr["e_intr"][0]).all())
            if uni and os.environ.get("CS_BA_UNIFORM", "1") != "0":
                need = tm["linearize_bytes"] - 32 * uni * len(pr["e_pt"])
                ba_out["roofline"]["uniform_edge_constants"] = {"classes": uni, "alg_bytes_per_linearisation": need, "frac": need / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if world > 1:
                # what the N ranks exchange per LM trial, this rank's stage times (rank 0's; the max over ranks is in ms_per_iteration), and the
                # same problem unsharded on rank 0's GPU right afterwards: the speed-ups of the build and of the whole iteration against N = 1
                st_sep = P.shard_timing()
                ba_out["multi_gpu"] = {"ranks": world, "separator_mode": bool(sinfo["sep_mode"]), "separator_system_unknowns": sinfo["n_sep"], "separator_width_max": sinfo["w_max"],
                                       "interior_unknowns_rank0": sinfo["interior_n"], "bytes_exchanged_per_trial": sinfo["bytes_per_trial"],
                                       "bytes_exchanged_per_trial_if_band_all_reduce": sinfo["bytes_per_trial_allreduce"],
                                       "separator_stage_ms_per_trial_rank0": {k: v / nsol for k, v in st_sep.items()}}
                if rank == 0:
                    P1 = capi.ba_from_dict(pr, device=local_rank)
                    P1.stage_timing(True)
                    P1.optimize(1)
                    tb1 = P1.timing()
                    t1 = time.perf_counter()
                    n1 = P1.optimize(args.ba_iters)
                    el1 = time.perf_counter() - t1
                    ta1 = P1.timing()
                    d1 = {k: ta1[k] - tb1[k] for k in ta1 if k.endswith("_ms")}
                    P1.close()
                    b1 = (d1["linearize_ms"] + d1["reduce_ms"] + d1["errors_ms"]) / max(1, n1)
                    ba_out["multi_gpu"].update({"single_gpu_ms_per_iteration": el1 / max(1, n1) * 1e3, "single_gpu_build_only_ms_per_iteration": b1,
                                                "speedup_iteration_vs_1gpu": (el1 / max(1, n1)) / (ba_el / max(1, n_it)),
                                                "speedup_build_only_vs_1gpu": b1 / ba_out["build_only_ms_per_iteration"]})
                barrier()
            elif args.ba == "C4" and os.environ.get("CS_BENCH_CHILD") is None and args.shard_probe > 1:
                # N = 1: what ONE rank of an R-rank job does per LM trial, measured alone on this GPU.  The handle is set up as a middle rank
                # of R; the transport is a loop-back (the gathered separator messages are R copies of this rank's own -- still a positive
                # definite separator system --, sums are the rank's own contribution), so every kernel of the sharded trial runs at its real
                # size while the numbers it produces are not a solution.  The stage times are GPU events; collectives are not timed here.
                R = args.shard_probe

                def shard_probe(prx, d1, nlin1, nsol1, it_1, what):
                    """one middle rank of R alone on this GPU (loop-back transport) against the single-GPU stage times d1 of the same problem"""
                    Pp = capi.ba_from_dict(prx, device=local_rank)
                    Pp.stage_timing(True)
                    Pp.set_shard(R // 2, R)
                    si = Pp.shard_info()
                    out = None
                    if si["sep_mode"]:
                        wmx = si["w_max"]
                        msg = 3 * wmx * wmx + 2 * wmx
                        mine = R // 2

                        def loopback(ptr, n, on_device, op):
                            if on_device and n == msg * R:
                                t = torch.as_tensor(capi.DeviceDoubles(ptr, n), device="cuda").view(R, msg)
                                t.copy_(t[mine].clone().expand(R, msg))
                                torch.cuda.synchronize()
                            return 0
                        Pp.optimize_sharded(1, loopback)
                        tb0, sb0 = Pp.timing(), Pp.shard_timing()
                        Pp.optimize_sharded(3, loopback)
                        ta0, sa0 = Pp.timing(), Pp.shard_timing()
                        ns_ = max(1, ta0["n_solves"] - tb0["n_solves"]); nl_ = max(1, ta0["n_linearizations"] - tb0["n_linearizations"])
                        stg = {k: (sa0[k] - sb0[k]) / ns_ for k in sa0}
                        lin_r, red_r = (ta0["linearize_ms"] - tb0["linearize_ms"]) / nl_, (ta0["reduce_ms"] - tb0["reduce_ms"]) / ns_
                        lin_1, red_1 = d1["linearize_ms"] / nlin1, d1["reduce_ms"] / nsol1
                        comm_us = 3 * 30.0       # three small collectives per trial over xGMI, latency-bound (assumed 30 us each: not measurable on one GPU)
                        solve_r = stg["interior_factor_ms"] + stg["separator_message_ms"] + stg["separator_solve_ms"] + stg["interior_backsolve_ms"]
                        rest_1 = (d1["backsub_ms"] + d1["errors_ms"] + d1["update_ms"]) / max(1, nsol1)
                        it_r = lin_r + red_r + solve_r + rest_1 / R + comm_us * 1e-3
                        out = {
                            "what": what, "cams": len(prx["cams"]), "points": len(prx["points"]),
                            "ranks": R, "interior_unknowns": si["interior_n"], "separator_system_unknowns": si["n_sep"], "bytes_exchanged_per_trial": si["bytes_per_trial"],
                            "bytes_exchanged_per_trial_if_band_all_reduce": si["bytes_per_trial_allreduce"],
                            "rank_ms": {"linearize": lin_r, "schur_reduce": red_r, **stg}, "single_gpu_ms": {"linearize": lin_1, "schur_reduce": red_1, "factor_and_substitution": d1["factor_ms"] / nsol1},
                            "projected_build_only_speedup": (lin_1 + red_1) / max(1e-9, lin_r + red_r),
                            "projected_solve_speedup": (d1["factor_ms"] / nsol1) / max(1e-9, solve_r),
                            "assumed_collective_latency_us": comm_us, "single_gpu_ms_per_iteration": it_1, "projected_ms_per_iteration": it_r, "projected_iteration_speedup": it_1 / max(1e-9, it_r)}
                    Pp.close()
                    return out
                sp = shard_probe(pr, d, nlin, nsol, ba_el_split / max(1, n_it_split) * 1e3,     # (stage split on, like the probe's own stage times)
                                 "one middle rank of %d, alone on this GPU, loop-back transport: GPU-event stage times of the separator-mode trial at C4/C5 size; the N-GPU figures below are "
                                 "arithmetic on these measured stage times, not measurements" % R)
                if sp:
                    ba_out["sharded_projection"] = sp
                # a second, larger point of the same probe (interiors ~25x the band instead of ~5x): where the latency chains of the separator
                # system stop dominating a rank's trial.  CS_BENCH_BA_WEAK=0 skips it (~25 s: generating 800 k points dominates).
                if os.environ.get("CS_BENCH_BA_WEAK", "1") != "0":
                    try:
                        prw = synth_ba.make_problem(n_cams=4000, n_points=800000, n_cuboids=2000, seed=42)
                        Pw = capi.ba_from_dict(prw, device=local_rank)
                        Pw.stage_timing(True)
                        Pw.optimize(1)
                        tbw = Pw.timing()
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                        nw = Pw.optimize(5)
                        torch.cuda.synchronize(); elw = time.perf_counter() - t1
                        taw = Pw.timing()
                        dw = {k: taw[k] - tbw[k] for k in taw if k.endswith("_ms")}
                        nlw = max(1, taw["n_linearizations"] - tbw["n_linearizations"]); nsw = max(1, taw["n_solves"] - tbw["n_solves"])
                        Pw.close()
                        spw = shard_probe(prw, dw, nlw, nsw, elw / max(1, nw) * 1e3,
                                          "the same probe on a 4x larger trajectory (4 000 cameras, 800 k points, 2 000 cuboids): one middle rank of %d alone on this GPU against the whole problem on "
                                          "this GPU; projected figures are arithmetic on measured stage times, not measurements" % R)
                        if spw:
                            spw["single_gpu_lm_iterations_per_s"] = nw / elw
                            ba_out["sharded_projection_large"] = spw
                        del prw
                    except Exception as ex:
                        ba_out["sharded_projection_large"] = {"error": repr(ex)}
            # the reference's usage pattern (main_obj.cpp:802-803): a frame is appended, then optimize(5) on the grown graph -- structure phase
            # included, because g2o pays it inside optimize() too (updateStructure)
            if world == 1 and os.environ.get("CS_BENCH_CHILD") is None:
                pr3 = synth_ba.make_problem(n_cams=200, n_points=20000, n_cuboids=50, seed=42) if args.ba == "C4" else pr
                nc3 = len(pr3["cams"])
                fp = np.full(len(pr3["points"]), nc3); np.minimum.at(fp, pr3["e_pt"], pr3["e_cam"])
                order3 = np.argsort(fp, kind="stable"); rank3 = np.empty_like(order3); rank3[order3] = np.arange(len(order3))
                fp = fp[order3]
                ep3 = rank3[pr3["e_pt"]]
                T0 = nc3 - 10
                keep_c = pr3["ce_cam"] < T0
                sel = pr3["e_cam"] < T0
                selo = np.maximum(pr3["oe_i"], pr3["oe_j"]) < T0
                n_p = int((fp < T0).sum())
                Pg = capi.BaProblem(pr3["cams"][:T0], pr3["cam_fixed"][:T0], pr3["cuboids"], pr3["cub_fixed"], pr3["points"][order3][:n_p], pr3["pt_fixed"][order3][:n_p], device=local_rank)
                Pg.set_edges_proj(ep3[sel], pr3["e_cam"][sel], pr3["e_uv"][sel], pr3["e_info"][sel], pr3["e_intr"][sel], pr3["e_huber"][sel])
                Pg.set_edges_cuboid(pr3["ce_cam"][keep_c], pr3["ce_cub"][keep_c], pr3["ce_meas"][keep_c], pr3["ce_info"][keep_c])
                Pg.set_edges_odom(pr3["oe_i"][selo], pr3["oe_j"][selo], pr3["oe_meas"][selo], pr3["oe_info"][selo])
                Pg.optimize(5)
                t_app, t_opt = [], []
                pts3, ptf3 = pr3["points"][order3], pr3["pt_fixed"][order3]
                for t in range(T0, nc3):
                    sel = pr3["e_cam"] == t; keep_c = pr3["ce_cam"] == t; selo = np.maximum(pr3["oe_i"], pr3["oe_j"]) == t
                    n_p2 = int((fp < t + 1).sum())
                    # the frame's rows are cut out of the synthetic problem BEFORE the clock starts (a tracker hands over arrays it already has;
                    # the boolean-mask slices of 100 k-edge arrays were 0.9 ms of this figure up to round 4)
                    a_v = (pr3["cams"][t:t + 1], pr3["cam_fixed"][t:t + 1], None, None, pts3[n_p:n_p2], ptf3[n_p:n_p2])
                    a_p = (ep3[sel], pr3["e_cam"][sel], pr3["e_uv"][sel], pr3["e_info"][sel], pr3["e_intr"][sel], pr3["e_huber"][sel])
                    a_c = (pr3["ce_cam"][keep_c], pr3["ce_cub"][keep_c], pr3["ce_meas"][keep_c], pr3["ce_info"][keep_c])
                    a_o = (pr3["oe_i"][selo], pr3["oe_j"][selo], pr3["oe_meas"][selo], pr3["oe_info"][selo])
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    Pg.append_vertices(*a_v)
                    Pg.append_edges_proj(*a_p)
                    Pg.append_edges_cuboid(*a_c)
                    Pg.append_edges_odom(*a_o)
                    Pg.sizes()          # the structure phase of the grown graph
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    Pg.optimize(5)
                    torch.cuda.synchronize()
                    t_app.append((t2 - t1) * 1e3); t_opt.append((time.perf_counter() - t2) * 1e3)
                    n_p = n_p2
                Pg.close()
                ba_out["growing_graph"] = {"what": "C3-sized graph (%d cameras, 20 k points, 50 cuboids) grown by one frame at a time through cs_ba_append_*, optimize(5) after every frame (main_obj.cpp:802-803); medians over 10 frames" % nc3,
                                           "append_and_structure_ms": float(np.median(t_app)), "optimize5_on_appended_graph_ms": float(np.median(t_opt)),
                                           "frame_ms": float(np.median(np.array(t_app) + np.array(t_opt)))}
            # the three factorisations of the reduced system (cs_ba_solver_path) on a graph that is NOT a trajectory: a survey flight, cameras on a
            # 2-D grid looking down (synth_ba.make_mesh_problem) -- a 2-D covisibility mesh; one damped solve (reduce + factor + substitute)
            # through the banded path, the general sparse path (minimum-degree ordering, level-scheduled Cholesky, dense tail) and rocSOLVER's
            # dense potrf.  CS_BENCH_BA_MESH=0 skips it (~25 s: generating the problem dominates).
            if world == 1 and os.environ.get("CS_BENCH_CHILD") is None and os.environ.get("CS_BENCH_BA_MESH", "1") != "0":
                try:
                    prm = synth_ba.make_mesh_problem(48, 40, 150000)
                    paths = {}
                    saved = {k: os.environ.get(k) for k in ("CS_BA_SPARSE", "CS_BA_FORCE_DENSE")}
                    xs = {}
                    for name, env in (("band", {"CS_BA_SPARSE": "0"}), ("sparse", {"CS_BA_SPARSE": "1"}), ("dense", {"CS_BA_SPARSE": "0", "CS_BA_FORCE_DENSE": "1"})):
                        for k in saved:
                            os.environ.pop(k, None)
                        os.environ.update(env)
                        Gm = capi.ba_from_dict(prm, device=local_rank)
                        n_red_m, _ = Gm.reduced_size()
                        path, bw_m, fill_m = Gm.solver_path(detail=True)
                        Gm.compute_errors(); Gm.build_system(dense_hpp=False)
                        okm, xm = Gm.solve(1e-3)
                        tsm = []
                        for _ in range(3):
                            torch.cuda.synchronize(); t1 = time.perf_counter(); okm, xm = Gm.solve(1e-3); torch.cuda.synchronize(); tsm.append((time.perf_counter() - t1) * 1e3)
                        paths[name] = {"path_taken": path, "solve_ms": min(tsm), "ok": bool(okm)}
                        if path == "band":
                            paths[name]["bandwidth"] = bw_m
                        if path == "sparse":
                            paths[name]["fill_of_dense_triangle"] = fill_m
                        xs[name] = xm
                        Gm.close()
                    for k, v in saved.items():
                        os.environ.pop(k, None)
                        if v is not None:
                            os.environ[k] = v
                    Gm = capi.ba_from_dict(prm, device=local_rank)       # ... and the one the library picks by itself
                    paths["chosen_by_default"] = Gm.solver_path()
                    Gm.close()
                    ref = xs["dense"]
                    ba_out["solver_paths"] = {"what": "one damped solve (reduce + factor + substitute) of a survey-flight graph -- %d cameras on a 48 x 40 grid, %d points, %d edges, %d unknowns in the reduced system: a 2-D covisibility mesh -- through each factorisation" % (len(prm["cams"]), len(prm["points"]), len(prm["e_pt"]), n_red_m),
                                              **paths,
                                              "max_rel_diff_to_dense": {k: float(np.abs(xs[k] - ref).max() / np.abs(ref).max()) for k in ("band", "sparse")}}
                    del prm
                except Exception as ex:
                    ba_out["solver_paths"] = {"error": repr(ex)}
            # CPU baseline of the BA half (rank 0, N = 1): the oracle (oracle/ba_oracle.cpp, -O2, one thread) on the SAME problem, wall time
            # per LM iteration split as g2o's G2OBatchStatistics does (core/batch_stats.h:48-62).  C3: the full run.  C4: one LM
            # iteration with residuals / linearisation / Schur complement / update in full and the dense LDL^T (the reference
            # constructs LinearSolverDense, main_obj.cpp:512) timed on every 256th column and scaled up (the whole factorisation of
            # the 10 494-unknown system is ~5e11 flop: minutes on one core).
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                from oracle import ba_oracle_py
                R = ba_oracle_py.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"])
                R.set_edges_proj(pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
                if len(pr["ce_cam"]):
                    R.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
                R.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
                sampled = args.ba == "C4"
                if sampled:
                    R.set_ldlt_stride(256)
                # parity at this size, printed instead of thrown away: one linearisation of a fresh handle at the initial estimates against
                # the oracle (oracle/ba_parity.py: chi2, b, H_ll, H_pl, every block of the damped reduced system, one damped solve with the
                # oracle's S factorised by LAPACK) -- the same comparison tests/test_ba_gpu.py asserts on
                try:
                    from oracle import ba_parity
                    G0 = capi.ba_from_dict(pr, device=local_rank)
                    t_par = time.perf_counter()
                    par = ba_parity.compare_linearisation(G0, R, pr, 50.0, one_thread=sampled)
                    par["seconds"] = time.perf_counter() - t_par
                    par["what"] = "max relative difference device vs oracle/ba_oracle.cpp, one linearisation + one damped solve (lambda = 50) of this problem at its initial estimates"
                    ba_out["parity_vs_oracle"] = par
                    G0.close()
                except Exception as ex:       # (a missing scipy or an out-of-memory host must not take the bench line with it)
                    ba_out["parity_vs_oracle"] = {"error": repr(ex)}
                # ... and as a trajectory: three LM iterations, device beside oracle (C4: the oracle's dense solve through LAPACK, everything else
                # of the iteration its own; oracle/ba_parity.py compare_trajectory -- tests/test_ba_gpu.py asserts on the same)
                if "error" not in ba_out["parity_vs_oracle"]:
                    try:
                        G1 = capi.ba_from_dict(pr, device=local_rank)
                        R1 = ba_oracle_py.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"])
                        R1.set_edges_proj(pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
                        if len(pr["ce_cam"]):
                            R1.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
                        R1.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
                        if sampled:
                            R1.use_lapack_solver()
                        tr = ba_parity.compare_trajectory(G1, R1, 3)
                        tr["what"] = ("3 LM iterations of this problem from its initial estimates, device vs oracle/ba_oracle.cpp%s: trial sequences, max relative differences of the chi2 / lambda histories and of the final states"
                                      % (" (its dense solve through LAPACK, the rest of the iteration its own)" if sampled else ""))
                        ba_out["parity_vs_oracle"]["trajectory"] = tr
                        G1.close(); R1.close()
                    except Exception as ex:
                        ba_out["parity_vs_oracle"]["trajectory"] = {"error": repr(ex)}
                tc = time.perf_counter()
                n_cpu = R.optimize(1 if sampled else args.ba_iters)
                cpu_wall = time.perf_counter() - tc
                st = R.stage_ms()
                note = ""
                if sampled:
                    # The oracle's LDL^T is the textbook unblocked loop: at n = 10 494 it streams the 881 MB matrix once per column and
                    # would take ~25 minutes, far slower than the blocked Eigen::LDLT the reference links.  The baseline therefore
                    # prices the dense solve with LAPACK's blocked dpotrf on ONE thread (at least as fast as Eigen's), measured at
                    # n = 4096 and scaled by n^3; the oracle's own extrapolated figure is kept beside it.
                    # Round 4: measured, not extrapolated -- the parity step above factorised the oracle's own damped 10 494-unknown S with
                    # LAPACK on one thread (oracle/ba_parity.py, one_thread); only if that step failed, the old n^3 scaling from n = 4096.
                    n_pose = P.sizes()[0]
                    pv = ba_out.get("parity_vs_oracle", {})
                    st["solve_unblocked_oracle_ms_extrapolated"] = st["solve_ms"]
                    if "dense_solve_seconds" in pv and pv.get("dense_solve_threads") == 1:
                        st["solve_ms"] = pv["dense_solve_seconds"] * 1e3
                        note = "; dense solve = LAPACK dpotrf + dpotrs of the oracle's own %d-unknown S on 1 thread, measured: %.2f s" % (pv["dense_solve_unknowns"], pv["dense_solve_seconds"])
                    else:
                        import scipy.linalg
                        from threadpoolctl import threadpool_limits
                        rng2 = np.random.default_rng(1)
                        m = 4096
                        M = rng2.standard_normal((m, 64))
                        A = M @ M.T + m * np.eye(m)
                        with threadpool_limits(limits=1):
                            scipy.linalg.cho_factor(A[:512, :512].copy(), lower=True)
                            t1 = time.perf_counter()
                            scipy.linalg.cho_factor(A, lower=True, overwrite_a=True, check_finite=False)
                            t_chol = time.perf_counter() - t1
                        st["solve_ms"] = t_chol * 1e3 * (n_pose / m) ** 3
                        note = "; dense solve = LAPACK dpotrf, 1 thread, %.2f s at n = %d scaled by (%d/%d)^3" % (t_chol, m, n_pose, m)
                tot_ms = sum(v for k, v in st.items() if not k.startswith("solve_unblocked"))
                ba_out["cpu_baseline"] = {"value": n_cpu / (tot_ms * 1e-3), "unit": "iters/s", "cores": 1, "kind": "port",
                                          "stage_ms_per_iteration": {k: v / n_cpu for k, v in st.items()},
                                          "sample": ("1 LM iteration (1 trial) of the same C4 problem through oracle/ba_oracle.cpp (-O2, single thread): residuals, linearisation, Schur complement "
                                                     "and update in full (%.1f s wall incl. the sampled unblocked LDL^T)%s" % (cpu_wall, note)) if sampled else
                                                    ("%d LM iterations of the same problem through oracle/ba_oracle.cpp (-O2, single thread), dense LDL^T in full, %.1f s" % (n_cpu, cpu_wall))}
                ba_out["speedup_vs_cpu"] = ba_out["value"] / ba_out["cpu_baseline"]["value"]
                build_cpu = (st["errors_ms"] + st["linearize_ms"] + st["schur_ms"]) / n_cpu
                ba_out["speedup_vs_cpu_build_only"] = build_cpu / ba_out["build_only_ms_per_iteration"]
                R.close()
            P.close()
        return ba_out

    # N = 1: here.  N > 1: after the headline has been assembled, under a watchdog (below) -- the multi-rank RCCL data path is the one
    # part of this script that cannot be exercised on a one-GPU box, and a fault in it must not take the path-A line with it.
    ba_out = ba_leg() if world == 1 else None

    # ---- next row: the distance-map front end (Canny + 3x3 distance transform of every ROI) on the device, timed on the
    # ROIs of this batch over synthetic gray images; not part of `value` (BASELINE.json's metric excludes Canny/DT)
    edge_out = None
    if rank == 0 and not args.no_edge:
        rng = np.random.default_rng(7)
        n_img = 16
        H, W = int(uniq[0]["img_h"]), int(uniq[0]["img_w"])
        yy, xx = np.mgrid[0:H, 0:W]
        grays = []
        for _ in range(n_img):
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
                    rois.append((l, t, w, h)); roi_img.append(f % n_img)
        det.edge_distance_maps_time(grays, rois[:64], roi_img[:64])     # warm-up
        ms = det.edge_distance_maps_time(grays, rois, roi_img)
        px = sum(w * h for _, _, w, h in rois)
        edge_out = {"what": "cv::Canny(80,200) + cv::distanceTransform(DIST_L2,3) of every ROI, on the device",
                    "rois": len(rois), "pixels": int(px), "device_ms": ms, "rois_per_s": len(rois) / (ms * 1e-3), "ms_per_1000_frames": ms * 1000.0 / max(1, args.frames),
                    "alg_bytes": int(px * (1 + 1 + 1 + 4 * 3)), "GB/s": px * 15 / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": px * 15 / (ms * 1e-3) / 1e9 / 8000.0,
                    "kernels": "edge_canny_bits_kernel (gray band in LDS, four Sobel responses per lane, hysteresis on bit planes in LDS) + edge_dt_kernel, workgroups largest ROI first"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import edge_oracle_py
            edge_oracle_py.lib()
            t1, n = time.perf_counter(), 0
            while time.perf_counter() - t1 < 3.0:
                edge_oracle_py.edge_distance_map(grays[roi_img[n % len(rois)]], rois[n % len(rois)])
                n += 1
            edge_out["cpu_oracle_rois_per_s"] = n / (time.perf_counter() - t1)

    # ---- image in, cuboids out, with the upload INSIDE the clock (the reference's caller hands detect_cuboid an image per call,
    # box_proposal_detail.cpp:84,320-327): per step the gray images of all frames go up from pinned host memory (cs_batch_refill_gray: one
    # contiguous block -> one DMA on a copy stream, beside the sweep that is still running), then Canny + distance transform of every ROI,
    # then the sweep.  Never `value` (BASELINE's metric has its inputs resident); reported beside the measured H2D rate, which bounds it.
    image_in = None
    if rank == 0 and world == 1 and args.image_in_steps > 0 and os.environ.get("CS_BENCH_CHILD") is None:
        try:
            H, W = int(uniq[0]["img_h"]), int(uniq[0]["img_w"])
            rngi = np.random.default_rng(17)
            yy, xx = np.mgrid[0:H, 0:W]
            base_imgs = []
            for _ in range(16):
                img = np.full((H, W), 90.0)
                for _ in range(25):
                    a = rngi.uniform(0, np.pi)
                    img += np.where((xx - rngi.uniform(0, W)) * np.cos(a) + (yy - rngi.uniform(0, H)) * np.sin(a) > 0, rngi.uniform(-40, 40), 0)
                img += 12 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rngi.normal(0, 6, (H, W))
                base_imgs.append(np.clip(img, 0, 255).astype(np.uint8))
            nfi = args.frames
            # two pinned blocks of nfi images each (different images at the same frame index), refilled alternately
            blocks = [torch.empty((nfi, H, W), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            for q, blk in enumerate(blocks):
                nb_ = blk.numpy()
                for f in range(nfi):
                    nb_[f] = base_imgs[(f + 5 * q) % len(base_imgs)]
            frames_img = [dict(fr) for fr in frames]
            # (one pipeline: the detector's own default worker count, not the main entry's per-pipeline share)
            det_i = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5, host_threads=args.host_threads), device=local_rank)
            bat_i = capi.Batch(det_i, frames_img, grays=[blocks[0].numpy()[f] for f in range(nfi)])
            bat_i.run()
            bat_i.refill_gray(base_ptr=blocks[1].data_ptr()); bat_i.run()          # warm: second buffer, copy stream
            bat_i.refill_gray(base_ptr=blocks[0].data_ptr()); bat_i.refill_wait()      # the first timed step's images: uploaded, front end not run yet
            torch.cuda.synchronize()
            K_i = args.image_in_steps
            step_log = []
            gc.collect(); gc.disable()
            t1 = time.perf_counter()
            for k in range(K_i):
                # the upload AFTER next goes behind the one this step's submit waits for (two image buffers, two uploads queued): the copy
                # engine never idles, and the step's whole device side (front end + sweep) runs beside an upload
                ts0 = time.perf_counter()
                bat_i.refill_gray(base_ptr=blocks[(k + 1) % 2].data_ptr())
                bat_i.submit()                                                       # waits for its images, queues front end + sweep
                ts1 = time.perf_counter()
                bat_i.collect()
                tmg_i = bat_i.timing()
                step_log.append([round((ts1 - ts0) * 1e3, 2), round((time.perf_counter() - ts1) * 1e3, 2), round(tmg_i["d2h_ms"], 2), round(tmg_i["finalize_ms"], 2)])
            bat_i.refill_wait()                      # (the upload queued by the last step: K_i whole uploads inside the clock, as K_i front ends and sweeps)
            torch.cuda.synchronize()
            dt_i = time.perf_counter() - t1
            gc.enable()
            bat_i.run()
            rec_i = bat_i.cuboids(0)
            # the H2D rate this box gives a pinned block of that size, alone
            dev_blk = torch.empty((nfi, H, W), dtype=torch.uint8, device="cuda")
            dev_blk.copy_(blocks[0], non_blocking=True); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for q in range(4):
                dev_blk.copy_(blocks[q % 2], non_blocking=True)
            torch.cuda.synchronize()
            h2d = 4 * nfi * H * W / (time.perf_counter() - t1)
            del dev_blk
            bytes_frame = H * W
            image_in = {"what": "image in, cuboids out: per step %d gray images (%d x %d, %.0f KB each) uploaded from one pinned block (cs_batch_refill_gray: one copy on a copy stream into the batch's second image buffer, beside the running front end + sweep), "
                                "Canny + distance transform of every ROI on the device, then the C2 sweep, records on the host; one batch, one detector; upload, front end, sweep and collection all inside the clock" % (nfi, W, H, bytes_frame / 1e3),
                        "steps": K_i, "value": nfi * K_i / dt_i, "unit": "frames/s", "ms_per_step": dt_i / K_i * 1e3,
                        "h2d_GBps_measured_alone": h2d / 1e9, "upload_bound_frames_per_s": h2d / bytes_frame, "fraction_of_upload_bound": (nfi * K_i / dt_i) / (h2d / bytes_frame),
                        "cuboids_in_frame_0": int(sum(len(c) for c in rec_i)),
                        "per_step_ms [refill + submit (waits for its upload), collect, of which: wait for the device, records on the host]": step_log}
            bat_i.close(); det_i.close()
            del blocks
        except Exception as ex:       # (pinning ~1 GB can fail on a constrained box: never fatal)
            gc.enable()
            image_in = {"error": repr(ex)}

    # ---- single-call latency: the drop-in call is per frame (detect_cuboid once per image, main_obj.cpp:633; detect_filter_lines :593)
    lat_out = None
    if rank == 0 and args.latency_calls > 0:
        def pct(call, n):
            call(); call()
            ts = []
            for _ in range(n):
                t1 = time.perf_counter(); call(); ts.append((time.perf_counter() - t1) * 1e3)
            ts.sort()
            return {"p50_ms": ts[len(ts) // 2], "p99_ms": ts[min(len(ts) - 1, int(len(ts) * 0.99))], "mean_ms": sum(ts) / len(ts), "calls": n}
        fr1 = uniq[0]
        H1, W1 = int(fr1["img_h"]), int(fr1["img_w"])
        rngl = np.random.default_rng(11)
        yyl, xxl = np.mgrid[0:H1, 0:W1]
        img1 = np.full((H1, W1), 95.0)
        for _ in range(30):
            a = rngl.uniform(0, np.pi)
            img1 += np.where((xxl - rngl.uniform(0, W1)) * np.cos(a) + (yyl - rngl.uniform(0, H1)) * np.sin(a) > 0, rngl.uniform(-45, 45), 0)
        gray1 = np.clip(img1 + rngl.normal(0, 4, img1.shape), 0, 255).astype(np.uint8)
        n_l = args.latency_calls
        d_rp = capi.Detector(capi.default_params(whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=6.0), device=local_rank)     # the reference class's defaults: 6 deg yaw step, roll/pitch sampling on
        d_c2 = capi.Detector(capi.default_params(whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0, yaw_range_deg=45.0, yaw_step_deg=0.5), device=local_rank)
        lat_out = {"what": "wall time of ONE call on one KITTI-shaped frame (1241 x 376, 8 boxes, ~400 segments), host buffers in, records out; p50 / p99 over %d calls" % n_l,
                   "cs_detect_cuboids (C2 sweep: 181 yaw, no roll/pitch sampling)": pct(d_c2.frame_call(fr1), n_l),
                   "cs_detect_cuboids_gray (the same, distance maps from the gray image on the device)": pct(d_c2.frame_call(fr1, gray1), n_l),
                   "cs_detect_cuboids (reference defaults: 6 deg yaw step, 25 roll/pitch samples)": pct(d_rp.frame_call(fr1), n_l),
                   "cs_detect_lines_gray (EDLines, one octave, length >= 15)": pct(lambda: d_c2.detect_lines(gray1, 15.0), n_l)}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py
            t1 = time.perf_counter()
            oracle_py.detect_cuboid(fr1, oracle_py.default_params(yaw_step_deg=0.5), atan2_mode=0)
            lat_out["cpu_oracle_detect_cuboid_C2_ms"] = (time.perf_counter() - t1) * 1e3
        d_rp.close(); d_c2.close()

    # ---- the segment producer in front of path A (line_lbd_detect::detect_filter_lines, EDLines): a batch of KITTI-sized images
    lines_out = None
    if rank == 0 and args.lines_images > 0:
        rngq = np.random.default_rng(21)
        Hq, Wq = int(uniq[0]["img_h"]), int(uniq[0]["img_w"])
        yyq, xxq = np.mgrid[0:Hq, 0:Wq]
        imgs = []
        for _ in range(min(8, args.lines_images)):
            im = np.full((Hq, Wq), 95.0)
            for _ in range(30):
                a = rngq.uniform(0, np.pi)
                im += np.where((xxq - rngq.uniform(0, Wq)) * np.cos(a) + (yyq - rngq.uniform(0, Hq)) * np.sin(a) > 0, rngq.uniform(-45, 45), 0)
            imgs.append(np.clip(im + rngq.normal(0, 4, im.shape), 0, 255).astype(np.uint8))
        batch_imgs = [imgs[i % len(imgs)] for i in range(args.lines_images)]
        # (the producer's own detector with the LIBRARY's default pool -- or the caller's --host-threads -- not the per-pipeline share of the
        # path-A run above: rounds 3-4 passed that share, 12 threads, and read 12-13 ms per LSD batch where the call alone takes 8.3 --
        # tools/lsd_busy_probe.py: the call does not care what else lives in the process, it scales with its pool up to ~32 threads)
        dl = capi.Detector(capi.default_params(host_threads=args.host_threads), device=local_rank)
        dl.detect_lines_batch(batch_imgs, 15.0)
        t1 = time.perf_counter()
        reps = 20      # (a burst of pool work can run into the cgroup's CPU quota and stall for the rest of a 100 ms period: enough batches to see the sustained rate, the median beside the mean)
        dev_ms = host_ms = call_ms = 0.0
        calls = []
        for _ in range(reps):
            segs = dl.detect_lines_batch(batch_imgs, 15.0)
            tq = dl.lines_timing(); dev_ms += tq["device_ms"]; host_ms += tq["host_ms"]; call_ms += tq["total_ms"]; calls.append(tq["total_ms"])
        dtq = time.perf_counter() - t1
        px = Hq * Wq * args.lines_images
        lines_out = {"what": "cs_detect_lines_batch: EDLines (one octave, length >= 15) of %d images of %d x %d; Gaussian / Sobel / gradient / anchors on the device (three packed bytes per pixel come back), routing + fitting + validation on the host pool, which starts on the first images while the later ones are still being computed and copied (chunks of 4)" % (args.lines_images, Wq, Hq),
                     "images_per_s": args.lines_images * reps / dtq, "segments_per_image": float(np.mean([len(x) for x in segs])),
                     "device_ms_per_batch": dev_ms / reps, "host_stage_ms_per_batch": host_ms / reps, "library_call_ms_per_batch": call_ms / reps, "library_call_ms_median": float(np.median(calls)),
                     "maps_kernel": {"alg_bytes_per_batch": 4 * px, "GB/s": 4 * px / (dev_ms / reps * 1e-3) / 1e9, "frac_of_hbm_peak": 4 * px / (dev_ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import edlines_oracle_py
            t1, nq = time.perf_counter(), 0
            while time.perf_counter() - t1 < 3.0:
                edlines_oracle_py.detect_filter_lines(batch_imgs[nq % len(batch_imgs)], 15.0)
                nq += 1
            lines_out["cpu_oracle_images_per_s"] = nq / (time.perf_counter() - t1)
        # the same batch through the LSD branch (use_LSD = true): blur / resize / gradient planes on the device (algorithmic bytes per
        # input pixel: 1 read + 0.64 x 12 written), region growing + rectangle validation on the host pool
        dl.detect_lines_batch(batch_imgs, 15.0, use_lsd=True)
        t1 = time.perf_counter()
        dev_ms = host_ms = call_ms = 0.0
        calls = []
        for _ in range(reps):
            segs = dl.detect_lines_batch(batch_imgs, 15.0, use_lsd=True)
            tq = dl.lines_timing(use_lsd=True); dev_ms += tq["device_ms"]; host_ms += tq["host_ms"]; call_ms += tq["total_ms"]; calls.append(tq["total_ms"])
        dtq = time.perf_counter() - t1
        lsd_bytes = px + 12 * int(round(Hq * 0.8)) * int(round(Wq * 0.8)) * args.lines_images
        lines_out["lsd"] = {"what": "cs_detect_lsd_batch: the reference's LSD branch on the same images",
                            "images_per_s": args.lines_images * reps / dtq, "segments_per_image": float(np.mean([len(x) for x in segs])),
                            "device_ms_per_batch": dev_ms / reps, "host_stage_ms_per_batch": host_ms / reps, "library_call_ms_per_batch": call_ms / reps, "library_call_ms_median": float(np.median(calls)),
                            "maps_kernels": {"alg_bytes_per_batch": lsd_bytes, "GB/s": lsd_bytes / (dev_ms / reps * 1e-3) / 1e9, "frac_of_hbm_peak": lsd_bytes / (dev_ms / reps * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import lsd_oracle_py
            t1, nq = time.perf_counter(), 0
            while time.perf_counter() - t1 < 3.0:
                lsd_oracle_py.detect_filter_lines(batch_imgs[nq % len(batch_imgs)], 15.0)
                nq += 1
            lines_out["lsd"]["cpu_oracle_images_per_s"] = nq / (time.perf_counter() - t1)
        dl.close()

    if rank == 0:
        total_frames = args.frames * args.steps * world
        value = total_frames / elapsed
        launches = max(1, int(acc["cand_kernel_launches"]))
        # dominant kernel of the sweep = the scorer (score_kernel); the geometry kernel is reported next to it
        alg_bytes = acc["score_kernel_bytes"] / launches
        # Duration of the kernel.  HIP events on the detector's stream bracket it in every launch.  In the timed region several
        # batches share the device, and an event pair then also times the kernel's wait for CUs that the other batches' kernels
        # hold (1.4-1.9 ms), which is not the kernel's duration (rocprofv3 of the same command: 0.9-1.1 ms).  The roofline
        # therefore uses the launches of this run in which nothing else was on the device -- the warm-up passes, one batch in
        # flight -- and reports the timed region's event figure beside it.
        timed_ms = acc["score_kernel_ms"] / launches
        iso_launches = max(1, int(iso.get("cand_kernel_launches", 0)))
        kern_ms = iso["score_kernel_ms"] / iso_launches if iso.get("score_kernel_ms") else timed_ms
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        geo_ms = acc["cand_kernel_ms"] / launches
        geo_bytes = acc["cand_kernel_bytes"] / launches
        # HBM traffic of the kernel as measured by the PMC passes of the same workload (profiles/pmc_traffic.json, written
        # by tools/profile_round.sh); null when this run's workload differs from the profiled one
        traffic = None
        traffic_source = None
        if not args.no_measure_traffic and world == 1 and not os.environ.get("CS_BENCH_CHILD"):
            traffic = measure_traffic_live(args, n_unique)
            traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only) over a child run of this workload, collected by this bench run" if traffic else None
        valu_issue = None
        if not args.no_measure_traffic and world == 1 and not os.environ.get("CS_BENCH_CHILD"):
            valu_issue = measure_valu_issue_live(args, n_unique)
        try:
            if traffic is not None:
                raise KeyError("measured live")
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pt = json.load(fh)
            wl = pt["workload"]
            if wl["frames_per_batch"] == args.frames and wl["unique"] == n_unique and world == 1:
                traffic = pt["kernels"]["score_kernel"]["hbm_bytes_per_launch"]
                traffic_source = "profiles/pmc_traffic.json: the same passes run by tools/profile_round.sh on this workload (not collected in this run: --no-measure-traffic, N > 1, or rocprofv3 unavailable)"
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "frames/sec detect_cuboid", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup,
            # what really ran before the clock started: every batch alone (sizes its buffers; the isolated kernel timings), then all pipelines side by side
            "warmup_untimed_steps": {"total": warm_alone_runs + warm_concurrent_steps, "each_batch_alone": warm_alone_runs, "all_pipelines_side_by_side": warm_concurrent_steps,
                                     "what": "--warmup W = %d: W runs of each of the %d batches alone, then max(W, batches) x %s steps driven exactly like the timed steps" % (args.warmup, inflight * depth, os.environ.get("CS_BENCH_WARMUP_ROUNDS", "4"))},
            "env_overrides": env_overrides, "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}, "diag_build": False,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not share_gpu else "synthetic (CS_BENCH_SHARE_GPU functional check: ranks share one device, not a performance number)",
            "config": {"workload": "C2: per-frame cuboid proposal sweep, 181 yaw x 8 boxes x ~400 line segments, 1241x376 KITTI-shaped",
                       "frames_per_batch_per_gpu": args.frames, "unique_frames": n_unique, "yaw_step_deg": 0.5, "batches_in_flight": inflight * depth, "pipelines": inflight, "batches_per_pipeline": depth,
                       "proposal_slots_per_frame": acc["n_slots"] / args.steps / args.frames,
                       "valid_proposals_per_frame": acc["n_valid"] / args.steps / args.frames, "parallelism": "frames sharded, no collective"},
            "roofline": {"kernel": "score_kernel", "bound": "hbm",
                         # what the SQ counters say the kernel waits for (profiles/r2u_detect_sq_counters.csv + the corner rebuild of round 3):
                         # ~3.4 k FP64 VALU instructions per proposal at ~55 % VALU-busy, the rest parked on the 77-99 scattered map samples;
                         # its HBM traffic is below its algorithmic bytes.  Staging the map in LDS (one workgroup per job) was measured and lost.
                         "limiter": "fp64 VALU issue + gather latency (HBM traffic = algorithmic bytes)",
                         "traffic_correction": "FETCH_SIZE x 2 + WRITE_SIZE x 1: measured on known byte counts for 4/8/16-byte coalesced, 8-byte strided and 4-byte gather patterns (profiles/r2_pmc_calibration.json)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "alg_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": kern_ms,
                         "measured": "HIP events around score_kernel on the detector's stream, this run's warm-up launches (one batch in flight, nothing else on the device)",
                         # the event pairs of the timed region: the kernel's duration plus its wait for CUs held by the other batches' kernels
                         "timed_region": {"event_ms_per_launch": timed_ms, "achieved": alg_bytes / (timed_ms * 1e-3) / 1e9 if timed_ms > 0 else 0.0,
                                          "frac": alg_bytes / (timed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if timed_ms > 0 else 0.0, "batches_in_flight": inflight},
                         # the per-proposal math is FP64 vector ALU -- both fractions, the larger one names the bound.
                         # ~1700 FP64 flop per valid proposal (FP64_FLOP_PER_PROPOSAL above)
                         "fp64_alu": {"flop_per_valid_proposal": FP64_FLOP_PER_PROPOSAL, "achieved": FP64_FLOP_PER_PROPOSAL * (acc["n_valid"] / launches) / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                                      "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": FP64_FLOP_PER_PROPOSAL * (acc["n_valid"] / launches) / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS if kern_ms > 0 else 0.0},
                         # the bound the kernel really runs against: the share of busy SIMD cycles that issue a VALU instruction (SQ counters, this run)
                         "valu_issue": valu_issue,
                         "other_kernels": {"candidate_compact_kernel (vanishing points + corners + ordered compaction)": {"ms": geo_ms, "alg_bytes": geo_bytes, "GB/s": geo_bytes / (geo_ms * 1e-3) / 1e9 if geo_ms > 0 else 0.0}}},
            "stage_ms_per_step": {k: acc[k] / args.steps for k in acc if k.endswith("_ms")},
            "fallback_boxes_per_step": acc["n_fallback_boxes"] / args.steps,
        }
        if host_cpu is not None:
            out["host_cpu"] = host_cpu
        if steady is not None:
            out["steady_state"] = steady
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
            # the records of the last TIMED step against the oracle (its cs_atan2 build: every field bit for bit, as tests/test_detect_gpu.py asserts)
            if timed_records is not None:
                opx = oracle_py.default_params(yaw_step_deg=0.5, whether_sample_cam_roll_pitch=0, whether_sample_bbox_height=0)
                n_cub = n_fld = n_bad = 0
                first_bad = None
                for f in range(len(timed_records)):
                    ref, _ = oracle_py.detect_cuboid(frames[f], opx, atan2_mode=1)
                    got = timed_records[f]
                    for bi in range(max(len(ref), len(got))):
                        g = got[bi] if bi < len(got) else []
                        r = ref[bi] if bi < len(ref) else []
                        if len(g) != len(r):
                            n_bad += 1
                            first_bad = first_bad or "frame %d box %d: %d cuboids against the oracle's %d" % (f, bi, len(g), len(r))
                            continue
                        for a, b in zip(g, r):
                            n_cub += 1
                            for k in b:
                                n_fld += 1
                                if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
                                    n_bad += 1
                                    first_bad = first_bad or "frame %d box %d field %s" % (f, bi, k)
                out["parity_vs_oracle"] = {"what": "the cs_cuboid records the last timed step wrote for its first %d frames against oracle/detect_oracle.cpp on the same frames (same parameters, shared cs_atan2): every field of every record compared bit for bit" % len(timed_records),
                                           "frames": len(timed_records), "cuboids_compared": n_cub, "fields_compared": n_fld, "mismatches": n_bad, "bit_identical": n_bad == 0}
                if first_bad:
                    out["parity_vs_oracle"]["first_mismatch"] = first_bad
            # for context only: the same restatement on every core this process may use, one independent process per core, the
            # frames dealt out among them -- the reference itself is single-threaded, so `value` above stays the baseline
            try:
                import subprocess
                ncore = len(os.sched_getaffinity(0))
                try:
                    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                    if q != "max":
                        ncore = max(1, min(ncore, int(q) // int(per)))
                except (OSError, ValueError):
                    pass
                code = ("import sys, time; sys.path.insert(0, %r)\n"
                        "from cube_slam_wu_amd import synth\n"
                        "from oracle import oracle_py\n"
                        "k = int(sys.argv[1]); uniq = [synth.make_frame(100000 + s) for s in range(k, %d, %d)][:6]\n"
                        "op = oracle_py.default_params(yaw_step_deg=0.5); oracle_py.detect_cuboid(uniq[0], op, atan2_mode=0)\n"
                        "n, t1 = 0, time.perf_counter()\n"
                        "while time.perf_counter() - t1 < 4.0:\n"
                        "    oracle_py.detect_cuboid(uniq[n %% len(uniq)], op, atan2_mode=0); n += 1\n"
                        "print(n / (time.perf_counter() - t1))\n") % (os.path.dirname(os.path.abspath(__file__)), n_unique, ncore)
                procs = [subprocess.Popen([sys.executable, "-c", code, str(k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(ncore)]
                rates = [float(pp.communicate(timeout=120)[0].strip().splitlines()[-1]) for pp in procs]
                out["cpu_baseline"]["all_cores"] = {"value": sum(rates), "unit": "frames/s", "cores": ncore, "kind": "port, one process per core (context only: the reference is single-threaded)"}
            except Exception as e:       # context only: never fatal
                out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
        if edge_out is not None:
            out["edge_front_end"] = edge_out
        if image_in is not None:
            out["image_in"] = image_in
        if rp_out is not None:
            out["roll_pitch_sampling_stress"] = rp_out
        if lat_out is not None:
            out["latency"] = lat_out
        if lines_out is not None:
            out["line_producer"] = lines_out
    else:
        out = None
    if world > 1 and args.ba != "none":
        # every rank arms the same watchdog: if the collective leg does not come back (or a rank dies in it and the others wait for
        # it), rank 0 prints the line it has -- `ba` saying so -- and every rank leaves with status 0
        limit = float(os.environ.get("CS_BENCH_BA_LIMIT_S", "240"))
        barrier()        # rank 0 comes from its single-GPU legs (front end, latency, line producer): the clocks below start together

        def bail(reason):
            if rank == 0:
                out["ba"] = {"error": reason, "note": "the multi-GPU BA leg was not measured in this run; the path-A fields above are complete"}
                print(json.dumps(out), flush=True)
            sys.stdout.flush()
            os._exit(0)
        tmr = threading.Timer(limit, bail, args=("multi-GPU BA leg did not finish within %.0f s" % limit,))
        tmr.daemon = True
        tmr.start()
        try:
            ba_out = ba_leg()
        except Exception as e:     # the other ranks may be waiting in a collective: no further rendezvous from here
            bail("multi-GPU BA leg failed on rank %d: %r" % (rank, e))
        tmr.cancel()
    parity_failed = False
    if rank == 0:
        if ba_out is not None:
            out["ba"] = ba_out
        print(json.dumps(out), flush=True)
        parity_failed = out.get("parity_vs_oracle", {}).get("bit_identical") is False
    for b_ in bats:
        b_.close()
    for d_ in dets:
        d_.close()
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the timed region's records differ from the oracle's (parity_vs_oracle in the line above)")


if __name__ == "__main__":
    main()
