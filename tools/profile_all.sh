#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and separate PMC passes for the bench configs, ONE MODE PER PASS
# (bench.py --single-mode: no same-run comparison legs), so that every row of a kernel_stats.csv is one clean population.
# rocprofv3 runs the program itself after "--" (no env/bash hops: see the pool rules).
#   usage: tools/profile_all.sh <tag> [all | <name>...]     names: sift1m_hinted sift1m_plain sift1m_noprune sift1b_w8 sift1b_w1 sift1b_rank deep1b hd
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${1:-x}
mkdir -p $OUT
B="--single-mode --no-cpu-baseline --no-sweep"
run() {  # name, extra bench args...
  name=$1; shift
  echo "== $name: $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -- python3 bench.py "$@" $B > $OUT/${name}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/${name}_fetch -- python3 bench.py "$@" $B > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/${name}_write -- python3 bench.py "$@" $B > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/${name}_sq1 -- python3 bench.py "$@" $B > $OUT/${name}_sq1.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --kernel-include-regex "ivf::" --output-format csv -d $OUT/${name}_sq2 -- python3 bench.py "$@" $B > $OUT/${name}_sq2.log 2>&1
}
shift
want() { for x in "${SEL[@]}"; do [ "$x" = all -o "$x" = "$1" ] && return 0; done; return 1; }
SEL=("${@:-all}")
want sift1m_hinted  && run sift1m_hinted  --steps 50 --warmup 5 --windows 2
want sift1m_plain   && run sift1m_plain   --steps 50 --warmup 5 --windows 2 --no-next-hint
want sift1m_noprune && run sift1m_noprune --steps 50 --warmup 5 --windows 2 --no-next-hint --no-pruning
# (>= 20 steps behind 5 warm-up steps per pass: the rocprofv3 averages of round 4 carried the warm-up launches of 3-step passes)
want sift1b_w8      && run sift1b_w8 --config sift1b --steps 20 --warmup 5 --windows 1
want sift1b_w1      && run sift1b_w1 --config sift1b --w 1 --steps 30 --warmup 5 --windows 1
want sift1b_rank    && run sift1b_rank --config sift1b --nq 2048 --steps 40 --warmup 5 --windows 1   # one rank's share of the 8-GPU batch
want deep1b         && run deep1b --config deep1b --steps 25 --warmup 5 --windows 1
want hd             && run hd --config hd --steps 30 --warmup 5 --windows 1
# keep only what is small enough to merge back: the counter passes need their counter_collection.csv only, and the
# kernel traces of the training phase are large
find $OUT -path "*_fetch/*" -name "*kernel_trace.csv" -delete
find $OUT -path "*_write/*" -name "*kernel_trace.csv" -delete
find $OUT -path "*_sq[12]/*" -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
du -sh $OUT
