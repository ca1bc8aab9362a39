cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"])'
run() { timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host "$@" 2>/dev/null | python -c "$fmt" "$LABEL"; }
for pg in 2 1; do
LABEL="mixture PG=$pg" IVFADC_FORCE_PG=$pg run
LABEL="lowrank PG=$pg" IVFADC_FORCE_PG=$pg run --data lowrank
LABEL="mixture no pruning PG=$pg" IVFADC_FORCE_PG=$pg run --no-pruning
LABEL="mixture w=32 PG=$pg" IVFADC_FORCE_PG=$pg run --w 32
LABEL="mixture w=2 PG=$pg" IVFADC_FORCE_PG=$pg run --w 2
LABEL="lowrank w=32 PG=$pg" IVFADC_FORCE_PG=$pg run --w 32 --data lowrank
LABEL="mixture single-mode PG=$pg" IVFADC_FORCE_PG=$pg run --single-mode
LABEL="lowrank single-mode PG=$pg" IVFADC_FORCE_PG=$pg run --single-mode --data lowrank
done
