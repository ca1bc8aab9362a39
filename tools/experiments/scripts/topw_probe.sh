#!/bin/bash
# on the GPU box: the top-w selection fused into the scan workgroups' prologue (default, kc <= 8192) against the stand-alone selection
# kernel (IVFADC_NO_FUSE_TOPW=1; IVFADC_TOPW_WPQ1=1: a wave per query) on the SIFT1M headline, 2 and 3 batches in flight
cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"])'
run() { timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host "$@" 2>gpurun_out/topw_err.txt | python -c "$fmt" "$LABEL" || tail -3 gpurun_out/topw_err.txt; }
for rep in 1 2; do
for n in 2 3; do
LABEL="fused inflight=$n" run --inflight $n
LABEL="stand-alone (4 waves/query) inflight=$n" IVFADC_NO_FUSE_TOPW=1 run --inflight $n
LABEL="stand-alone (1 wave/query) inflight=$n" IVFADC_NO_FUSE_TOPW=1 IVFADC_TOPW_WPQ1=1 run --inflight $n
done; done
