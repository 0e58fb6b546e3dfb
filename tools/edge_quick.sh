# edge front end: parity tests + the bench's edge line + per-kernel times (run on the GPU box)
timeout 200 python -m pytest tests/test_edge_gpu.py -q -m gpu -x 2>&1 | tail -2
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_edge -o kt -- python $R/bench.py --steps 2 --warmup 1 --ba none --no-cpu-baseline --rp-frames 0 --latency-calls 0 --lines-images 0 --no-measure-traffic > $R/gpurun_out/kt_edge.log 2>&1
cd $R
python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/kt_edge/kt_kernel_stats.csv")))
for r in rows:
    if "edge_" in r["Name"]: print("%-40s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for l in open("gpurun_out/kt_edge.log"):
    if l.startswith("{"): print(json.loads(l)["edge_front_end"]["device_ms"])
PY
