#!/bin/bash
for spec in "8192 1" "1024 1" "1024 8" "512 8" "256 8"; do set -- $spec; nq=$1; w=$2; for qg in 0 1 2 4; do
  timeout -k 10 200 python bench.py --config sift1b --nq $nq --w $w --qg $qg --steps 10 --warmup 3 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('nq=$nq w=$w qg=$qg', 'ms/step=%.3f' % d['ms_per_step'], 'scan=%.3f' % r['scan_ms_per_launch'], r['kernel'], 'chunk', r['chunk_points'], 'grid', r['scan_grid'])" || echo "nq=$nq w=$w qg=$qg failed"
done; done
