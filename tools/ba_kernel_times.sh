#!/bin/bash
# per-kernel times of the BA path (C4) under rocprofv3:  tools/ba_kernel_times.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ba_${tag} -o prof -- python $R/bench.py --no-measure-traffic --frames 50 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_ba_${tag}.log 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_ba_${tag}/prof_kernel_stats.csv')))
for r in rows:
    if 'ba_' in r['Name'] or 'band_' in r['Name']:
        print("%-50s calls %5s avg %9.1f us total %8.2f ms" % (r['Name'].split('(')[0][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
find gpurun_out/prof_ba_${tag} -name '*kernel_trace.csv' -size +20M -delete
