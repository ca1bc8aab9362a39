cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "coarse_ms=%.4f" % r["coarse_ms_per_launch"], r["kernel"][:50])'
for mode in "" "--single-mode"; do
for kcmin in 2048 1024; do
IVFADC_MFMA_MIN_KC=$kcmin timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host $mode 2>/dev/null | python -c "$fmt" "sift1m mfma_min_kc=$kcmin $mode"
done; done
