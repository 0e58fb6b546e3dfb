"""Summary of a rocprofv3 kernel trace: per kernel name total / mean duration, and over the busiest contiguous stretch (the timed
region of the bench run) the wall time, the time with at least one kernel resident, and the sum of kernel durations (= mean
concurrency x busy time)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the timed region: the last long run of launches with gaps < 5 ms
groups, cur = [], [rows[0]]
for a in rows[1:]:
    if a[0] - max(x[1] for x in cur[-50:]) > 5_000_000:
        groups.append(cur); cur = []
    cur.append(a)
groups.append(cur)
g = max(groups, key=len)
t0, t1 = g[0][0], max(x[1] for x in g)
ev = sorted([(s, 1) for s, e, _ in g] + [(e, -1) for s, e, _ in g])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d; last = t
tot = sum(e - s for s, e, _ in g)
per = defaultdict(lambda: [0, 0])
for s, e, n in g:
    per[n][0] += e - s; per[n][1] += 1
n_steps = per.get("cs::score_kernel", [0, 1])[1]
print("region: %d launches, %d sweeps, wall %.2f ms = %.3f ms per sweep; >= 1 kernel resident %.1f %%; sum of kernel durations %.2f ms = %.3f ms per sweep (mean concurrency %.2f)"
      % (len(g), n_steps, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / n_steps, 100.0 * busy / (t1 - t0), tot / 1e6, tot / 1e6 / n_steps, tot / max(1, busy)))
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-44s %5d launches  mean %8.1f us  per sweep %7.3f ms" % (n[:44], c, d / c / 1e3, d / 1e6 / n_steps))
