#!/bin/bash
# plan sweep of the per-rank regime of the 8-GPU strong-scaling run (SIFT1B shape, 2048 / 4096 queries per rank)
for nq in 2048 4096; do for w in 8 1; do for qg in 0 1 2 4 -1; do
  timeout -k 10 200 python bench.py --config sift1b --nq $nq --w $w --qg $qg --steps 10 --warmup 3 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('nq=$nq w=$w qg=$qg', 'ms/step=%.3f' % d['ms_per_step'], 'scan=%.3f' % r['scan_ms_per_launch'], 'coarse=%.3f' % r['coarse_ms_per_launch'], r['kernel'], 'chunk', r['chunk_points'], 'grid', r['scan_grid'])" || echo "nq=$nq w=$w qg=$qg failed"
done; done; done
