#!/bin/bash
# Derived utilisation metrics of the ivf:: kernels (one counter group per pass, kernel-trace only).
# usage: tools/pmc_derived.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/drv_$tag
mkdir -p $OUT
i=0
for grp in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "LDSBankConflict L2CacheHit" "TA_BUSY_avr TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/p$i -- python3 bench.py "$@" --no-cpu-baseline --no-sweep > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $OUT/p$i.log)"
done
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v)//2:]
        print("   %-32s %16.2f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
