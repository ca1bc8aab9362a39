cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"])'
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-sweep --no-other-configs --no-host-to-host "$@" 2>/dev/null | python -c "$fmt" "$LABEL"; }
for n in 1 2; do
LABEL="deep1b inflight=$n" run --config deep1b --steps 20 --warmup 3 --inflight $n
LABEL="hd inflight=$n" run --config hd --steps 30 --warmup 3 --inflight $n
LABEL="sift1b w=1 inflight=$n" run --config sift1b --w 1 --steps 20 --warmup 3 --inflight $n
done
