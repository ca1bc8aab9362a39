#!/bin/bash
# GPU box: tools/profile_all.sh for every configuration / mode, then the summaries (kernel stats, PMC, SQ, traffic.json) and ONLY the
# summaries are kept under gpurun_out/ (the raw rocprofv3 directories exceed what gpurun merges back).   usage: tools/profile_round.sh <tag> [names...]
tag=${1:-r04}; shift
cd "$GRAFT_REPO_ROOT"
bash tools/profile_all.sh $tag "$@"
OUT=gpurun_out/prof_$tag
DST=gpurun_out/profiles_$tag
mkdir -p $DST
for name in sift1m_hinted sift1m_plain sift1m_noprune sift1b_w8 sift1b_w1 sift1b_rank deep1b hd; do
  [ -d $OUT/${name}_trace ] || continue
  python3 tools/summarize_prof.py $OUT/${name}_trace profiles/${tag}_${name}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $(grep -m1 "== $name:" gpurun_out/*profile*.log 2>/dev/null | sed 's/.*: //') --single-mode --no-cpu-baseline --no-sweep" > $DST/${name}_kernels.txt 2>&1
  python3 tools/summarize_pmc.py $OUT $name $tag > $DST/${name}_pmc.txt 2>&1
  grep -h "^{" $OUT/${name}_trace.log | tail -1 >> $DST/bench_lines_single_mode.jsonl
done
cp profiles/${tag}_*_kernel_stats.csv profiles/${tag}_*_pmc.csv profiles/${tag}_*_sq.csv profiles/traffic.json $DST/ 2>/dev/null
rm -rf $OUT
du -sh gpurun_out
ls $DST
