#!/bin/bash
# per-kernel split of the distance-map front end: rocprofv3 --kernel-trace --stats over tools/edge_prof.py
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/edge_prof${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/edge_prof${TAG} -o ep -- python $R/tools/edge_prof.py 5 > $R/gpurun_out/edge_prof${TAG}.log 2>&1
cd $R
grep "^rep" gpurun_out/edge_prof${TAG}.log
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/edge_prof${TAG}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "edge" in r["Name"]: print("%-40s calls %5s  avg %10.1f us  total %10.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
find gpurun_out/edge_prof${TAG} -name '*.csv' -size +2M -delete
