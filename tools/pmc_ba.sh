#!/bin/bash
# SQ counter pass over the bundle-adjustment kernels (one rocprofv3 run, kernel-trace + pmc only):  tools/pmc_ba.sh <tag>
# -> gpurun_out/pmc_<tag>_ba.csv: per kernel SQ_INSTS_VALU_MFMA_F64 (matrix-core issues), SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, waves ...
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_${tag}_ba -o pmc -- python $R/bench.py --no-measure-traffic --steps 1 --warmup 1 --frames 50 --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --ba-iters 4 > $R/gpurun_out/pmc_${tag}_ba.log 2>&1
cd $R
python - <<PY > gpurun_out/pmc_${tag}_ba.csv
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_${tag}_ba/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
names = sorted({c for k in acc for c in acc[k]})
print("kernel,dispatches," + ",".join(names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    if k.startswith("cs::ba_") or k.startswith("cs::band_"): print(k + "," + str(cnt[k]) + "," + ",".join("%.6g" % acc[k][c] for c in names))
PY
cat gpurun_out/pmc_${tag}_ba.csv
rm -rf gpurun_out/pmc_${tag}_ba
