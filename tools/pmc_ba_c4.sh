#!/bin/bash
# Counters of the bundle-adjustment kernels over C4-ONLY launches (tools/ba_quick.py C4: nothing else in the process -- no C3 growing graph,
# no shard probe, no detect sweep), three separate rocprofv3 passes (kernel-trace + pmc only):  tools/pmc_ba_c4.sh <tag>
#   -> gpurun_out/<tag>_ba_pmc_summary.csv   FETCH_SIZE / WRITE_SIZE per kernel and launch, HBM bytes = FETCH x 2 + WRITE x 1 (profiles/r2_pmc_calibration.json)
#   -> gpurun_out/<tag>_ba_sq_counters.csv   SQ_INSTS_VALU, SQ_INSTS_VALU_MFMA_F64 (matrix-core issues), SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, waves per launch
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
CMD="python $R/tools/ba_quick.py C4 6"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_bapmc_fetch -o pmc -- $CMD > $R/gpurun_out/${tag}_bapmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_bapmc_write -o pmc -- $CMD > $R/gpurun_out/${tag}_bapmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/${tag}_bapmc_sq -o pmc -- $CMD > $R/gpurun_out/${tag}_bapmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_bapmc_kt -o kt -- $CMD > $R/gpurun_out/${tag}_bapmc_kt.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${tag}_bapmc_fetch gpurun_out/${tag}_bapmc_write > gpurun_out/${tag}_ba_pmc_summary.csv
python - <<PY > gpurun_out/${tag}_ba_sq_counters.csv
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/${tag}_bapmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("kernel,launches," + ",".join(n + "_per_launch" for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    if k.startswith(("cs::ba_", "cs::band_", "cs::bcr_")): print(k + ",%d," % len(disp[k]) + ",".join("%.6g" % (acc[k][c] / len(disp[k])) for c in names))
PY
cp $(find gpurun_out/${tag}_bapmc_kt -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_ba_kernel_stats.csv 2>/dev/null
head -30 gpurun_out/${tag}_ba_pmc_summary.csv; head -30 gpurun_out/${tag}_ba_sq_counters.csv; head -25 gpurun_out/${tag}_ba_kernel_stats.csv
find gpurun_out/${tag}_bapmc_fetch gpurun_out/${tag}_bapmc_write gpurun_out/${tag}_bapmc_sq gpurun_out/${tag}_bapmc_kt -name '*.csv' -size +5M -delete
