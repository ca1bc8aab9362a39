#!/bin/bash
# SQ / LDS counters of one kernel: tools/pmc_quick.sh <tag> <kernel substring> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; kern=$2; shift 2
tools/pmc_sq.sh $tag "$@" --no-sweep > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/sq_$tag/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$kern" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        v = v[len(v)//2:]
        print("   %-24s %16.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
find gpurun_out/sq_$tag -name "*.csv" -delete
