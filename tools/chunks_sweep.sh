#!/bin/bash
# the driver's contract run (20 steps, 5 warm-up) for several host/GPU pipeline chunk counts per batch, alternating, to separate them from box noise
for rep in 1 2 3; do
  for c in ${1:-1 2 4}; do
    python bench.py --no-measure-traffic --steps 20 --warmup 5 --steady-steps 0 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --chunks $c 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('rep $rep  chunks $c: contract(20 steps) %.0f' % o['value'])"
  done
done
