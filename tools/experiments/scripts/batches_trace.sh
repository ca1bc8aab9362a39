#!/bin/bash
# kernel trace of ivfadc_search_batches on the bench's index: tools/batches_trace.sh <tag> [env assignments are inherited]   (GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/bt_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/batches_trace.py "$@" > $OUT/run.log 2>&1
python3 tools/trace_timeline.py $OUT 120 > gpurun_out/bt_${tag}_timeline.txt 2>&1
python3 tools/trace_overlap.py $OUT 1200 >> gpurun_out/bt_${tag}_timeline.txt 2>&1
tail -3 $OUT/run.log >> gpurun_out/bt_${tag}_timeline.txt
find $OUT -name "*.csv" -size +5M -delete
