#!/bin/bash
# eight-wave kernel, four (table mode 6) and eight (7) queries per code stream, against the four-wave (5) list-major kernel on the SIFT1B shape
# over w and batch size: scan ms per launch; last column: the plan's own choice (0)
for cfg in "--w 1" "--w 2" "--w 4" "" "--w 16" "--nq 2048" "--nq 4096 --w 2" "--nq 1024"; do
  line="$cfg:"
  for tm in 5 6 7 0; do
    r=$(timeout -k 10 300 python3 bench.py --config sift1b --table-mode $tm --single-mode --no-cpu-baseline --steps 8 --warmup 3 --windows 1 $cfg 2>/dev/null | grep -oE "scan_ms_per_launch\": ?[0-9.]+|\"kernel\": ?\"[a-z0-9_]*" | tr '\n' ' ' | sed 's/"kernel": \?"//; s/scan_ms_per_launch": \?//')
    line="$line | tm$tm $r"
  done
  echo "$line"
done
