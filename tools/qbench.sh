#!/bin/bash
# quick A/B lines: tools/qbench.sh "<config> <qg> <steps>" ...   (run on the GPU box)
for spec in "$@"; do
  set -- $spec
  python bench.py --config $1 --steps ${3:-10} --warmup 2 --no-cpu-baseline --no-sweep --qg ${2:-0} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 qg=${2:-0}', 'qps=%.0f' % d['value'], 'ms/step=%.4f' % d['ms_per_step'], r['kernel'], 'scan_ms=%.4f' % r['scan_ms_per_launch'], 'coarse_ms=%.4f' % r['coarse_ms_per_launch'], 'frac=%.3f' % r['frac'], 'lds=%.1f' % r['roofline_lds']['achieved'])"
done
