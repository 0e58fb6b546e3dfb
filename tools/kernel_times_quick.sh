#!/bin/bash
# per-kernel times of the detect path, one batch in flight, nothing else in the run:  tools/kernel_times_quick.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --ba none --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --image-in-steps 0 --steady-steps 0 --inflight ${INFLIGHT:-1}"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_${tag} -o kt -- python $R/bench.py --no-measure-traffic $ARGS > $R/gpurun_out/kt_${tag}.log 2>&1
cd $R
f=$(find gpurun_out/kt_${tag} -name '*kernel_stats.csv' | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print("%-60s calls %4s avg %9.1f us  min %9.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
PY
find gpurun_out/kt_${tag} -name '*kernel_trace.csv' -size +20M -delete
