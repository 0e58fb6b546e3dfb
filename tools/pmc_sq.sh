#!/bin/bash
# SQ counter pass over the detect path (one rocprofv3 run, kernel-trace + pmc only):  tools/pmc_sq.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -o pmc -- python $R/bench.py --no-measure-traffic --steps 3 --warmup 1 --no-cpu-baseline --ba none --steady-steps 0 --latency-calls 0 --image-in-steps 0 > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_${tag}_sq2 -o pmc -- python $R/bench.py --no-measure-traffic --steps 3 --warmup 1 --no-cpu-baseline --ba none --steady-steps 0 --latency-calls 0 --image-in-steps 0 > $R/gpurun_out/pmc_${tag}_sq2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("gpurun_out/pmc_${tag}_sq", "gpurun_out/pmc_${tag}_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(names))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        if k.startswith("cs::"): print(k + "," + ",".join("%.4g" % acc[k][c] for c in names))
PY
find gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_sq2 -name '*counter_collection.csv' -size +20M -delete
