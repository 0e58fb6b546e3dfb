#!/bin/bash
# kernel timeline of ONE batch of the C2 sweep alone on the device (rocprofv3 --kernel-trace, --inflight 1): start offset, duration, gap to the end of
# whatever ran before, queue -- what a batch's ~3 ms consist of besides kernel time:  tools/detect_timeline.sh
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/det_tl -o tl -- python $R/bench.py --no-measure-traffic --steps 4 --warmup 2 --steady-steps 0 --no-cpu-baseline --ba none --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --image-in-steps 0 --inflight 1 --depth 1 > $R/gpurun_out/det_tl.log 2>&1
cd $R
f=$(find gpurun_out/det_tl -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "line_setup_small_kernel" in r["Kernel_Name"]]
a = idx[-2]
# the batch: from the table upload in front of the last-but-one line setup to the result copy behind its ranking
while a > 0 and "multi_copy_kernel" not in rows[a]["Kernel_Name"]: a -= 1
b = idx[-1]
while b > a and "multi_copy_kernel" not in rows[b]["Kernel_Name"]: b -= 1
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
print("one batch of 1 000 frames: %.1f us from its first kernel's start to the next batch's first kernel's start, %d kernels, sum of kernel time %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  dur %7.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?")[-3:], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs::", "")[:60]))
    prev_end = max(prev_end, e)
PY
find gpurun_out/det_tl -name '*kernel_trace.csv' -size +20M -delete
