#!/bin/bash
# marginal cost of each detect kernel in the saturated pipeline: steady-state ms per sweep with the kernel left out (diagnostics)
for k in none line_setup vp_support candidate score rank records; do
  CS_DETECT_SKIP=$k python bench.py --no-measure-traffic --steps 8 --warmup 3 --steady-steps 160 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --depth 1 2>/dev/null | tail -1 | python -c "import json,sys; o=json.load(sys.stdin); print('skip %-12s steady %.3f ms per sweep' % ('$k', o['steady_state']['ms_per_step']))"
done
