#!/bin/bash
# repeated solves through the window-resident fronts (CS_BAND_WIN=1), a different system every repetition: any residual above 1e-10 or a non-zero info is a failure
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { CS_BAND_WIN=1 timeout 300 build_tmp/band_bench "$@" 2>&1; }
{
  for round in 1 2 3; do
    for a in "5994 120" "630 120 20 1" "600 109" "2500 97" "300 60" "129 40" "4000 33" "3000 128" "3001 129" "2000 64 20 1" "1217 100" "2049 33" "4097 2" "1500 65"; do set -- $a; echo "## $1 $2 ${4:-0}"; run $1 $2 ${3:-20} ${4:-0}; done
  done
} > gpurun_out/band_win_stress.log
n=$(grep -c "^rep" gpurun_out/band_win_stress.log)
awk '/^##/ { shape = $0 } /^rep/ { if ($7 != 0 || $9 + 0 > 1e-10) { b++; print shape, $0 } } END { print "bad:", b + 0 }' gpurun_out/band_win_stress.log | tail -20
echo "window-front stress: $n solves"
