import sys, json
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin   # a file argument, or a pipe
for line in src:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    s=d["stage_ms_per_step"]
    print("%.0f f/s  %.2f ms |"%(d["value"], d["ms_per_step"]), " ".join("%s=%.2f"%(k[:-3],v) for k,v in s.items()), "| fb", d["fallback_boxes_per_step"])
