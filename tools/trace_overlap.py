"""Kernel overlap in a rocprofv3 --kernel-trace csv: per kernel name the count and mean duration, and for the busiest stretch of the
trace the union of the kernels' intervals against the sum of their durations (how many kernels run at once, how much of the wall
time has no kernel at all).   usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [last N kernels]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")))
rows.sort()
rows = rows[-last:]
by = defaultdict(list)
for s, e, n, q in rows:
    by[n].append(e - s)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s n=%5d mean %8.2f us  total %9.1f us" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
t0, t1 = rows[0][0], max(e for _, e, _, _ in rows)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in rows)
print("wall %.1f us, some kernel running %.1f us (%.1f %%), sum of durations %.1f us -> %.2f kernels at once while busy; queues: %s" %
      ((t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0), tot / 1e3, tot / busy, sorted(set(q for _, _, _, q in rows))))
