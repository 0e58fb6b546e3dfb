#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB per dispatch) on known byte counts: tools/pmc_calib.sh <tag>
# -> gpurun_out/pmc_calib_<tag>.json (copy to profiles/).  Two separate passes (kernel-trace + pmc only).
tag=${1:-x}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_calib_${tag}_f -o pmc -- $R/build_tmp/pmc_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_calib_${tag}_w -o pmc -- $R/build_tmp/pmc_calib > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections, json
def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = load("gpurun_out/pmc_calib_${tag}_f", "FETCH_SIZE"), load("gpurun_out/pmc_calib_${tag}_w", "WRITE_SIZE")
GiB = float(1 << 30)
known_read = {"calib_read16": GiB, "calib_read8": GiB, "calib_read4": GiB, "calib_read8_s144": (GiB // 144) * 144, "calib_gather4": (1 << 23) * 128.0}
known_write = {"calib_write16": GiB, "calib_write8_s144": (GiB // 144) * 144}
out = {"method": "1 GiB array (4x the Infinity Cache) touched once per kernel; factor = known HBM bytes / (counter KiB x 1024); calib_gather4 is priced at one 128-byte line per 4-byte gather",
       "read": {}, "write": {}}
for k, b in known_read.items():
    if k in fe: out["read"][k] = {"FETCH_SIZE_KiB": fe[k], "known_bytes": b, "factor": b / (fe[k] * 1024)}
for k, b in known_write.items():
    if k in wr: out["write"][k] = {"WRITE_SIZE_KiB": wr[k], "known_bytes": b, "factor": b / (wr[k] * 1024)}
json.dump(out, open("gpurun_out/pmc_calib_${tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_calib_${tag}_f gpurun_out/pmc_calib_${tag}_w
