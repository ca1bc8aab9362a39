#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into the rows that matter
(the ivf:: kernels of this library) and write them as CSV under profiles/.

usage: tools/summarize_prof.py <rocprof_out_dir> <profiles/name.csv> [note]
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_stats.csv under " + src)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "ivf::" in r["Name"]:
                    rows.append(r)
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w", newline="") as fh:
        if note:
            fh.write("# %s\n" % note)
        fh.write("# source: rocprofv3 --kernel-trace --stats (kernel_stats.csv), ivf:: kernels only\n")
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows:
        print("%-60s calls=%-6s avg=%10.1f us  min=%8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["MinNs"]) / 1e3))


if __name__ == "__main__":
    main()
