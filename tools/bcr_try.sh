#!/bin/bash
# Block cyclic reduction against the banded kernels: residuals + times over a few shapes, then the kernel stats of the C4 shape.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { timeout 120 build_tmp/band_bench "$@" 2>&1; }
{
  for a in "5994 120" "1194 120" "3000 100" "640 120" "10494 120" "300 20" "129 40"; do
    echo "== $a  order 2 (BCR)"; run $a 3 2
    echo "== $a  order 0"; run $a 3 0
  done
  echo "== 1500 101 Bv 120"; run 1500 101 3 2 120
} > gpurun_out/bcr_try.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bcr_prof -o bcr -- $GRAFT_REPO_ROOT/build_tmp/band_bench 5994 120 20 2 > $GRAFT_REPO_ROOT/gpurun_out/bcr_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/bcr_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/bcr_kernel_stats.csv
grep -v "rep 0" gpurun_out/bcr_try.log
[ -f gpurun_out/bcr_kernel_stats.csv ] && head -6 gpurun_out/bcr_kernel_stats.csv
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/bcr_prof/bcr_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last solve: print kernel sequence with durations and gaps
seq=[r for r in rows if 'bcr' in r['Kernel_Name']]
last=seq[-24:]
t0=int(last[0]['Start_Timestamp'])
prev=None
for r in last:
    st=int(r['Start_Timestamp']);en=int(r['End_Timestamp'])
    print('%-28s grid %6s start %7.1f dur %6.1f gap %5.1f'%(r['Kernel_Name'][4:26],r.get('Grid_Size_X','?'),(st-t0)/1e3,(en-st)/1e3,(st-prev)/1e3 if prev else 0))
    prev=en
PY
