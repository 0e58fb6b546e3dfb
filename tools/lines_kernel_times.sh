#!/bin/bash
# kernel durations of the segment producer's device stages (EDLines maps, LSD blur / scale + gradient), 64 KITTI-sized images per batch:
#   tools/lines_kernel_times.sh <tag>  ->  gpurun_out/kt_lines_<tag>/kt_kernel_stats.csv
tag=${1:-rX}
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_lines_${tag} -o kt -- python $R/bench.py --steps 2 --warmup 1 --ba none --no-cpu-baseline --rp-frames 0 --latency-calls 0 --no-measure-traffic --steady-steps 0 --no-edge > $R/gpurun_out/kt_lines_${tag}.log 2>&1
cd $R
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/kt_lines_${tag}/kt_kernel_stats.csv")):
    if "lines_" in r["Name"] or "lsd_" in r["Name"]: print("%-60s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
