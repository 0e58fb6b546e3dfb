#!/bin/bash
# kernel + copy timeline of the image-in loop (tools/image_in_probe.py) -- what runs beside what
R=$(pwd); export TMPDIR=/tmp; export TAG; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/imgin_tl${TAG} -o tl -- env ${IMGIN_ENV:-X=1} python $R/tools/image_in_probe.py 1000 4 ${MODE:-ahead} > $R/gpurun_out/imgin_tl${TAG}.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os
T = "gpurun_out/imgin_tl" + os.environ.get("TAG", "")
rows = []
for f in glob.glob(T + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?")[-2:], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cs::", "")[:48])))
for f in glob.glob(T + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
rows.sort()
# the last ~45 ms
t_end = rows[-1][1]
sel = [r for r in rows if r[0] > t_end - 46e6 and (r[1] - r[0] > 20000 or r[2].startswith("C"))]
t0 = sel[0][0]
for s, e, n in sel:
    print("%9.3f ms  dur %8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
PY
find gpurun_out/imgin_tl${TAG} -name '*.csv' -size +5M -delete
