#!/bin/bash
# frames/s of the detect path against (pipelines, batches per pipeline):  tools/inflight_sweep.sh "4x1 2x2 3x2 4x2"
for c in ${1:-4x1 2x2 3x2 4x2}; do
  n=${c%x*}; d=${c#*x}
  python bench.py --no-measure-traffic --steps 20 --warmup 5 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --image-in-steps 0 --steps 40 --inflight $n --depth $d 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('pipelines $n x depth $d: contract(20 steps) %.0f  steady(200) %.0f  ' % (o['value'], o['steady_state']['value']), {k: round(v,2) for k,v in o['stage_ms_per_step'].items() if k in ('setup_host_ms','d2h_ms','finalize_ms','total_ms')})"
done
