#!/bin/bash
# instruction-cache counters of one kernel: tools/pmc_icache.sh <tag> <kernel substring> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; kern=$2; shift 2
OUT=gpurun_out/ic_$tag; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/p1 -- python3 bench.py "$@" --no-cpu-baseline > $OUT/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    v = v[len(v)//2:]
    print("   %-24s %16.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
tail -2 $OUT/p1.log | cut -c1-200
find $OUT -name "*.csv" -delete
