#!/bin/bash
# small-batch latency: device time per batch (ms_per_step with the queries resident) for tiny batches
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "us/step=%.1f" % (d["ms_per_step"]*1e3), "scan_us=%.1f" % (r["scan_ms_per_launch"]*1e3), "coarse_us=%.1f" % (r["coarse_ms_per_launch"]*1e3), r["kernel"], "grid", r["scan_grid"], "chunk", r["chunk_points"])'
for spec in "sift1m 1 8" "sift1m 1 1" "sift1m 16 8" "sift1b 16 8" "sift1b 1 8" "sift1b 16 1"; do set -- $spec
  timeout -k 10 280 python bench.py --config $1 --nq $2 --w $3 --steps 200 --warmup 20 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "$fmt" "$1 nq=$2 w=$3" || echo "$spec failed"
done
