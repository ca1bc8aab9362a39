#!/bin/bash
# on the GPU box: K = 100 (LDS selectors) on the Deep1B, HD and SIFT1M shapes, with the probe-group LDS cap of the K > 64 plans at 40 KB
# (four workgroups per CU: one probe per round on m = 16) and at 56 KB (three workgroups, two probes per round)
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], r["kernel"])'
for cfg in deep1b hd sift1m; do
  for cap in 40 56; do
    IVFADC_PG_LDS_CAP_KB=$cap timeout -k 10 300 python bench.py --config $cfg --K 100 --steps 8 --warmup 2 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host --single-mode 2>/dev/null | python -c "$fmt" "$cfg K=100 cap=${cap}KB" || exit 1
  done
done
