#!/bin/bash
# Stress of the banded solver's hand-offs: many shapes, a different system every repetition, every residual checked; the second half with two
# solver processes sharing the device (uneven load on the CUs that exchange data).  Any residual above 1e-10 or a non-zero info fails.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { timeout 300 build_tmp/band_bench "$@" 2>&1; }
{
  for a in "5994 120" "10494 183" "1902 190" "840 241" "3000 60" "600 109" "129 40" "4000 33" "2500 97" "7000 150"; do run $a 25; done
  for a in "630 120" "5994 120" "2000 64"; do run $a 15 1; done
  ( for a in "5994 120" "3000 60" "1902 190"; do run $a 20; done ) &
  p=$!
  for a in "10494 183" "2500 97" "840 241"; do run $a 20; done
  wait $p
} > gpurun_out/band_stress.log
n=$(grep -c "^rep" gpurun_out/band_stress.log)
bad=$(awk '/^rep/ { if ($7 != 0 || $9 + 0 > 1e-10) b++ } END { print b + 0 }' gpurun_out/band_stress.log)
echo "band stress: $n solves, $bad bad"
[ "$bad" = "0" ]
