#!/bin/bash
# roll/pitch stress with four batches in flight: the LDS-resident ranking instance against the ordinary one, alternating
for rep in 1 2 3; do
  for m in big small; do
    if [ $m = small ]; then export CS_RANK_NO_BIG=1; else unset CS_RANK_NO_BIG; fi
    python bench.py --steps 2 --warmup 1 --ba none --no-cpu-baseline --latency-calls 0 --lines-images 0 --no-measure-traffic --steady-steps 0 --no-edge 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('rep $rep $m: %.0f frames/s' % o['roll_pitch_sampling_stress']['value'])"
  done
done
