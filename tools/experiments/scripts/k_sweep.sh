#!/bin/bash
# on the GPU box: step time against K across the register-selector / LDS-selector boundary (K = 64 | 65) on one shape
# usage: tools/k_sweep.sh <config> K...
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], r["kernel"])'
cfg=$1; shift
for k in "$@"; do
  timeout -k 10 300 python bench.py --config $cfg --K $k --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host --single-mode 2>/dev/null | python -c "$fmt" "$cfg K=$k" || exit 1
done
