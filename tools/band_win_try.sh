#!/bin/bash
# checks of the window-resident band fronts: residuals + time, against the cooperative kernels (CS_BAND_WIN=0)
#   SHAPES="n:LD[:reps[:one_sided]],..."  NOCOOP=1 skips the cooperative runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SHAPES=${SHAPES:-"300:60,630:120:3:1,600:109,2000:64,5994:120,129:40,2500:97,4000:33,3000:128"}
{
  for a in ${SHAPES//,/ }; do
    IFS=: read n ld reps os <<< "$a"
    reps=${reps:-3}; os=${os:-0}
    echo "== n=$n LD=$ld one_sided=$os  [window]"
    CS_BAND_WIN=1 timeout 60 build_tmp/band_bench $n $ld $reps $os 2>&1 | tail -8
    if [ -z "$NOCOOP" ]; then
      echo "== n=$n LD=$ld one_sided=$os  [cooperative]"
      timeout 60 build_tmp/band_bench $n $ld $reps $os 2>&1 | tail -4
    fi
  done
} > gpurun_out/band_win_try.log 2>&1
tail -120 gpurun_out/band_win_try.log
