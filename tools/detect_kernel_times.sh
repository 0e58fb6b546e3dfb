#!/bin/bash
# per-kernel times of the detect path under rocprofv3:  tools/detect_kernel_times.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_det_${tag} -o prof -- python $R/bench.py --no-measure-traffic --steps 5 --warmup 2 --no-cpu-baseline --ba none > $R/gpurun_out/prof_det_${tag}.log 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_det_${tag}/prof_kernel_stats.csv')))
for r in rows:
    if 'cs::' in r['Name']:
        print("%-70s calls %5s avg %9.1f us min %9.1f max %9.1f" % (r['Name'].split('(')[0][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
find gpurun_out/prof_det_${tag} -name '*kernel_trace.csv' -size +20M -delete
