#!/bin/bash
# kernel durations of one rank's LM trials (rank 4 of 8, loop-back) against the unsharded run:  tools/shard_probe_trace.sh
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for n in 8 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_$n -o sp -- python $R/tools/shard_probe_trace.py $n > $R/gpurun_out/sp_$n.log 2>&1
  f=$(find $R/gpurun_out/sp_$n -name '*kernel_stats.csv' | head -1)
  echo "== ranks $n"; python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$f")) if "ba_" in r["Name"] or "band" in r["Name"]]
tot = 0
for r in rows[:22]:
    print("%-52s calls %4s avg %8.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
