#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and separate PMC passes for the bench configs.
# rocprofv3 runs the program itself after "--" (no env/bash hops: see the pool rules).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${1:-x}
mkdir -p $OUT
run() {  # name, extra bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${name}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${name}_write.log 2>&1
}
run sift1m --steps 50 --warmup 5
run sift1b --config sift1b --steps 3 --warmup 1
run deep1b --config deep1b --steps 3 --warmup 1
run hd --config hd --steps 3 --warmup 1
# keep only what is small enough to merge back (kernel traces of the training phase are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
du -sh $OUT
ls -R $OUT | head -60
