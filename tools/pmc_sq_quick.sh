#!/bin/bash
# SQ counters of the detect kernels, one batch in flight, nothing else in the run:  tools/pmc_sq_quick.sh <tag>
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --ba none --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --image-in-steps 0 --steady-steps 0 --inflight 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcq_${tag}_a -o pmc -- python $R/bench.py --no-measure-traffic $ARGS > $R/gpurun_out/pmcq_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmcq_${tag}_b -o pmc -- python $R/bench.py --no-measure-traffic $ARGS > $R/gpurun_out/pmcq_${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $R/gpurun_out/pmcq_${tag}_c -o pmc -- python $R/bench.py --no-measure-traffic $ARGS > $R/gpurun_out/pmcq_${tag}_c.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
out = open("gpurun_out/pmcq_${tag}.csv", "w")
for d in ("gpurun_out/pmcq_${tag}_a", "gpurun_out/pmcq_${tag}_b", "gpurun_out/pmcq_${tag}_c"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel,launches," + ",".join(names), file=out)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        if k.startswith("cs::"): print(k + ",%d," % len(disp[k]) + ",".join("%.4g" % (acc[k][c] / len(disp[k])) for c in names), file=out)
PY
find gpurun_out/pmcq_${tag}_a gpurun_out/pmcq_${tag}_b gpurun_out/pmcq_${tag}_c -name '*counter_collection.csv' -size +20M -delete
cat gpurun_out/pmcq_${tag}.csv
