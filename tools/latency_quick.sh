# single-call latencies alone (the bench's latency leg): p50 / p99 of cs_detect_cuboids, cs_detect_cuboids_gray, cs_detect_lines_gray (run on the GPU box)
python bench.py --steps 1 --warmup 0 --ba none --no-cpu-baseline --rp-frames 0 --latency-calls ${CALLS:-200} --lines-images 0 --no-measure-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['latency']
for k, v in d.items():
    if isinstance(v, dict): print('%-90s p50 %.3f  p99 %.3f ms' % (k[:90], v['p50_ms'], v['p99_ms']))
"
