#!/bin/bash
# on the GPU box: batches in flight (bench.py --inflight 2 / 3 / 4) on the SIFT1M headline
cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], d.get("batches_in_flight", {}).get("lanes"), (r.get("roofline_valu") or {}).get("step", {}).get("frac"))'
for rep in 1 2; do
for n in 2 3 4; do
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host --inflight $n 2>gpurun_out/lanes_err.txt | python -c "$fmt" "sift1m inflight=$n" || tail -5 gpurun_out/lanes_err.txt
done; done
