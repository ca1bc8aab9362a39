#!/bin/bash
# every BASELINE.json shape through bench.py (CPU baseline + oracle parity included): tools/bench_lines.sh <out.jsonl>   (GPU box)
cd "$GRAFT_REPO_ROOT"
out=$1; : > $out
python bench.py --steps 1000 --warmup 50 2>/dev/null >> $out
python bench.py --config sift1b --steps 20 --warmup 3 2>/dev/null >> $out
python bench.py --config sift1b --w 1 --steps 50 --warmup 5 2>/dev/null >> $out
python bench.py --config deep1b --steps 50 --warmup 5 2>/dev/null >> $out
python bench.py --config hd --steps 100 --warmup 5 2>/dev/null >> $out
python bench.py --config toy --steps 1000 --warmup 50 --no-sweep 2>/dev/null >> $out
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 1000 --warmup 50 --no-sweep --no-cpu-baseline 2>/dev/null >> $out
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 1000 --warmup 50 --no-sweep --no-cpu-baseline --gather-every 8 2>/dev/null >> $out
python bench.py --single-process --gpus 1 --config sift1b --steps 10 --warmup 2 2>/dev/null >> $out
python - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    r = d.get("roofline") or {}
    print(d["config"]["workload"][:60], "| qps", d["value"], "| ms", d["ms_per_step"], "| scan", r.get("scan_ms_per_launch"), "coarse", r.get("coarse_ms_per_launch"),
          "| frac", r.get("frac"), r.get("bound"), "| lds", (r.get("roofline_lds") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| parity", d.get("parity"), d.get("distributed", {}).get("collectives_in_timed_region"))
PY
