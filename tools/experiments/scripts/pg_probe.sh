#!/bin/bash
# on the GPU box: probes per round of the query-major kernel (IVFADC_FORCE_PG) on the SIFT1M headline, both bench modes
cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], r["kernel"][:40], "tables", r["roofline_valu"]["lane_ops_per_launch"]["tables"])'
for mode in "" "--single-mode"; do
for pg in 2 1 4; do
IVFADC_FORCE_PG=$pg timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host $mode 2>gpurun_out/pg_err.txt | python -c "$fmt" "sift1m PG=$pg $mode" || tail -5 gpurun_out/pg_err.txt
done; done
