#!/bin/bash
# steady-state frames/s against the workgroup size of rank_kernel (diagnostics: CS_RANK_THREADS)
for nt in 256 128 64 256 128 64; do
  CS_RANK_THREADS=$nt python bench.py --no-measure-traffic --steps 8 --warmup 3 --steady-steps 200 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('rank threads $nt: steady %.0f frames/s  rank_kernel %.3f ms per sweep (events, 4 in flight)' % (o['steady_state']['value'], o['stage_ms_per_step']['rank_kernel_ms']))"
done
