#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of the steady-state steps from a rocprofv3 kernel trace.
usage: tools/trace_gaps.py <rocprof_out_dir> [kernels_per_step]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ivf::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # steady state
dur = defaultdict(list); gap = defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    n = a["Kernel_Name"].split("(")[0][-40:]
    dur[n].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[n + " -> next"].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in dur.items():
    print("%-48s n=%-5d dur  mean %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3))
for k, v in gap.items():
    v2 = sorted(v)
    print("%-48s n=%-5d gap  mean %8.2f us  median %8.2f us" % (k, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3))
