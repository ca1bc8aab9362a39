#!/bin/bash
# on the GPU box: Deep1B-shape step with the matrix-core lower-bound tables (default) and with the exact f32 tables
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], "frac=%.3f" % r["frac"], r["kernel"], (r.get("table_build") or {}).get("survivors_per_query"))'
timeout -k 10 300 python bench.py --config deep1b --steps 8 --warmup 2 --no-cpu-baseline --no-sweep "$@" 2>/dev/null | python -c "$fmt" lb_tables
IVFADC_EXACT_TABLES=1 timeout -k 10 300 python bench.py --config deep1b --steps 8 --warmup 2 --no-cpu-baseline --no-sweep "$@" 2>/dev/null | python -c "$fmt" exact_tables
