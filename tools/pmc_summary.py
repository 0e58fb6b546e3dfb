#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, both reported in KiB per dispatch).
Usage: pmc_summary.py <fetch_dir> <write_dir>  -> CSV on stdout: kernel, dispatches, avg FETCH_SIZE KiB, avg WRITE_SIZE KiB,
and the HBM bytes per launch: reads = FETCH_SIZE KiB x 1024 x 2, writes = WRITE_SIZE KiB x 1024.  The factors are measured, not
assumed: tools/pmc_calib.sh (profiles/r2_pmc_calibration.json) streams / gathers / strides through a 1 GiB array with known byte
counts -- 16, 8 and 4 bytes per lane coalesced, 8 bytes at a 144-byte lane stride, 4-byte gathers priced at one 128-byte line each:
FETCH_SIZE reports exactly half of the bytes in every one of them (factor 1.98-2.00), WRITE_SIZE all of them (1.00).
With a third argument the same numbers are also written as JSON (profiles/pmc_traffic.json)."""
import csv, glob, os, sys, collections


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("kernel,dispatches,avg_FETCH_SIZE_KiB,avg_WRITE_SIZE_KiB,hbm_read_bytes_x2_corrected,hbm_write_bytes")
for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
    nf, sf = fe.get(k, [0, 0.0]); nw, sw = wr.get(k, [0, 0.0])
    af = sf / nf if nf else 0.0; aw = sw / nw if nw else 0.0
    print("%s,%d,%.1f,%.1f,%.0f,%.0f" % (k, max(nf, nw), af, aw, af * 1024 * 2, aw * 1024))

if len(sys.argv) > 3:
    import json
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh)",
           "correction": "HBM read bytes = FETCH_SIZE KiB x 1024 x 2, write bytes = WRITE_SIZE KiB x 1024; factors measured by tools/pmc_calib.sh on known byte counts (profiles/r2_pmc_calibration.json): 2.00 for 16 / 8 / 4-byte coalesced reads, 1.98 for 8-byte reads at a 144-byte stride, 2.00 for 4-byte gathers (one 128-byte line each), 1.00 for writes",
           "workload": {"frames_per_batch": 1000, "unique": 100, "yaw_step_deg": 0.5},   # bench.py's defaults, which tools/profile_round.sh runs
           "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(k, [0, 0.0]); nw, sw = wr.get(k, [0, 0.0])
        af = sf / nf if nf else 0.0; aw = sw / nw if nw else 0.0
        name = k.replace("void ", "").replace("cs::", "")
        out["kernels"][name] = {"fetch_kib": af, "write_kib": aw, "hbm_bytes_per_launch": af * 1024 * 2 + aw * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
