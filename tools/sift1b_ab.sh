#!/bin/bash
# tools/sift1b_ab.sh: the SIFT1B-shape list-major scan, three regimes (16 384 x w = 8, 16 384 x w = 1, one rank's share 2048 x w = 8),
# one bench line each into gpurun_out/$1/ (single mode: one kernel population per run)
out=gpurun_out/${1:-sift1b_ab}; mkdir -p $out
for cfg in "w8:" "w1:--w 1" "r2048:--nq 2048"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  timeout -k 10 300 python3 bench.py --config sift1b --table-mode ${TM:-6} --single-mode --steps 20 --warmup 3 $extra > $out/$tag.json 2> $out/$tag.err || exit 1
  python3 - "$out/$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "scan_ms", r["scan_ms_per_launch"], r["kernel"], "parity", d.get("parity"))
PY
done
