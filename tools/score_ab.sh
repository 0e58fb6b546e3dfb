#!/bin/bash
# A / B of the scorer on one box: the current kernel against the round-3 form (CS_SCORE_R3=1).  Per form: the detect kernels alone
# (rocprofv3 kernel stats, one batch in flight) and the saturated pipeline's rate (bench.py, 4 pipelines, 60 steps, twice).
cd "$(dirname "$0")/.."
ARGS="--no-measure-traffic --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --steady-steps 0"
for form in r4 r3; do
  case $form in r4) unset CS_SCORE_R3;; r3) export CS_SCORE_R3=1;; esac
  echo "== $form"
  bash tools/kernel_times_quick.sh ab_$form 2>&1 | grep -E "score_kernel|rank_kernel|vp_support|candidate|line_setup" | head -8
  for rep in 1 2; do python bench.py $ARGS --steps 60 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   frames/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))"; done
done
