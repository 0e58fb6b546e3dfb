# kernels of ONE cs_detect_cuboids_gray call (one KITTI-shaped frame, 8 ROIs): durations from a kernel trace of the bench's latency leg (run on the GPU box)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_gray -o kt -- python $R/bench.py --steps 1 --warmup 0 --ba none --no-cpu-baseline --rp-frames 0 --latency-calls 60 --lines-images 0 --no-measure-traffic > $R/gpurun_out/kt_gray.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/kt_gray/**/kt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last gray call: find the last edge_canny launch with a small grid and print the kernels from there to the next record_kernel
idx = [i for i, r in enumerate(rows) if "edge_canny" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 64 * 256]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 16]:
    print("%8.1f us  dur %7.1f  grid %6d x %4d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), r["Kernel_Name"][:60]))
PY
