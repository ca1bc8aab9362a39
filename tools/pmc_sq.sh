#!/bin/bash
# SQ / LDS counters of the ivf:: kernels (own passes, kernel-trace only).  usage: tools/pmc_sq.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/sq_$tag
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/p1 -- python3 bench.py "$@" --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/p2 -- python3 bench.py "$@" --no-cpu-baseline > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            v = v[len(v)//2:]
            print("   %-24s %16.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
tail -3 $OUT/p1.log | cut -c1-300
