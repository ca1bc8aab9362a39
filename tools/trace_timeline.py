"""The last N kernels of a rocprofv3 --kernel-trace csv as a timeline: start (us, relative), duration, gap to the previous kernel's end
on the SAME queue, queue id, kernel name.   usage: python tools/trace_timeline.py <dir> [N]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ivf::", "")[:44], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
last_end = {}
for s, e, name, q, st in rows:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("%9.1f us  dur %7.1f  gap-on-queue %7.1f  q=%-3s s=%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, st, name))
    last_end[q] = e
