import sys, json
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin   # a file argument, or a pipe
for l in src:
    if l.startswith("{"):
        d = json.loads(l)["ba"]
        print("%.1f it/s  %.2f ms/iter |" % (d["value"], d["ms_per_iteration"]), " ".join("%s=%.2f" % (k[:-3], v) for k, v in d["stage_ms_per_iteration"].items()), "| build GB/s %.0f" % d["roofline"]["achieved"])
    elif l.startswith("[band]"):
        print(l.rstrip())
