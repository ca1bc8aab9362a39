#!/bin/bash
# on the GPU box: the certified two-level coarse search forced on the SIFT1M shape (kc = 1024: 16 groups), both bench modes
cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], r["kernel"][:50], "parity", d["parity"]["ids_bit_exact"] if isinstance(d.get("parity"), dict) else d.get("parity"))'
for data in mixture lowrank; do
for mode in "" "--single-mode"; do
for cm in 0 6; do
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host --data $data --coarse-mode $cm $mode 2>gpurun_out/tl_err.txt | python -c "$fmt" "sift1m $data coarse_mode=$cm $mode" || tail -5 gpurun_out/tl_err.txt
done; done; done
