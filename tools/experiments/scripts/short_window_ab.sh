cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], d["windows"]["qps_min"], d["windows"]["qps_max"])'
for rep in 1 2; do
for st in 20 100; do
python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host 2>/dev/null | python -c "$fmt" "stand-alone steps=$st"
IVFADC_LANES_FUSE_TOPW=1 python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host 2>/dev/null | python -c "$fmt" "fused steps=$st"
done; done
