#!/bin/bash
# Does the automatic plan pick the fastest scan plan?  For a grid of (batch, w) on one configuration: queries/s with
# the plan left to the library (qg 0), forced query-major (-1) and forced list-major (4).   usage: tools/plan_sweep.sh <config>
cd "$GRAFT_REPO_ROOT"
cfg=${1:-sift1m}
for nq in 64 1024 4096 16384; do
  for w in 1 8 32; do
    line="$cfg nq=$nq w=$w:"
    for qg in 0 -1 4; do
      v=$(python bench.py --config $cfg --no-cpu-baseline --no-sweep --steps 40 --warmup 5 --nq $nq --w $w --qg $qg 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2fM(%s)' % (j['value']/1e6, 'LM' if 'list-major' in j['roofline']['kernel'] else 'QM'))")
      line="$line  qg=$qg $v"
    done
    echo "$line"
  done
done
