#!/bin/bash
# per-kernel time of one bench config: tools/ktrace.sh <tag> <bench args...>   (GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/kt_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py "$@" --no-cpu-baseline --no-sweep > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-86s calls=%-6s avg_us=%10.2f total_ms=%9.2f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cp $f gpurun_out/kt_${tag}_kernel_stats.csv
find $OUT -name "*.csv" -size +5M -delete
