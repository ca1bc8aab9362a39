#!/bin/bash
# tools/trace_one.sh <name> <bench args...>: ONE rocprofv3 kernel-trace pass of bench.py (single mode), the ivf:: rows of its stats to stdout
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
name=$1; shift
OUT=gpurun_out/trace_$name
rm -rf $OUT; mkdir -p $OUT
echo "== $name: $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py "$@" --single-mode --no-cpu-baseline --no-sweep > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
grep -E "^\"(void )?ivf::" $f | grep -v "tr_\|coarse_dist\|argmin\|encode" | cut -d, -f1-4 | head -12
find $OUT -name "*kernel_trace.csv" -delete
