#!/bin/bash
# contract (20 steps) and steady-state (200 steps) frames/s for several (pipelines x depth), alternating
for rep in 1 2 3; do
  for c in ${1:-4x1 4x2}; do
    n=${c%x*}; d=${c#*x}
    python bench.py --no-measure-traffic --steps 20 --warmup 5 --steady-steps 200 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --inflight $n --depth $d 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('rep $rep  $c: contract %.0f  steady %.0f' % (o['value'], o['steady_state']['value']))"
  done
done
