#!/bin/bash
# kernel timeline of one LM iteration of the C4 bundle adjustment (rocprofv3 --kernel-trace): start offset, duration, gap to the previous
# kernel's end, queue -- shows what a trial's ~1.75 ms consist of besides kernel time
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ba_tl -o tl -- python $R/tools/ba_quick.py ${1:-C4} 6 5 ${2:-nosplit} > $R/gpurun_out/ba_tl.log 2>&1
cd $R
f=$(find gpurun_out/ba_tl -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete iteration: from the last-but-one ba_lin_cam_kernel to the last one
idx = [i for i, r in enumerate(rows) if "ba_lin_cam_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
print("one LM iteration: %.1f us wall, %d kernels, sum of kernel time %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  dur %7.1f  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?")[-3:], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs::", "")[:60]))
    prev_end = max(prev_end, e)
PY
find gpurun_out/ba_tl -name '*kernel_trace.csv' -size +5M -delete
