cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; echo rc=$?; tail -2 gpurun_out/gpu_tests.txt | cut -c1-300
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"])'
run() { timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host "$@" 2>/dev/null | python -c "$fmt" "$LABEL"; }
LABEL="mixture adaptive" run
LABEL="mixture feedback off" IVFADC_NO_PG_FEEDBACK=1 run
LABEL="lowrank adaptive" run --data lowrank
LABEL="lowrank feedback off" IVFADC_NO_PG_FEEDBACK=1 run --data lowrank
LABEL="mixture w=32 adaptive" run --w 32
LABEL="mixture single-mode adaptive" run --single-mode
LABEL="mixture steps=20" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host 2>/dev/null | python -c "$fmt" "driver cmd steps=20"
LABEL="deep1b" run --config deep1b --steps 10 --warmup 2
