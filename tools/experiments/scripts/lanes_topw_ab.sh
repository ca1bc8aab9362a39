cd $GRAFT_REPO_ROOT
fmt='import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]; h=d.get("host_to_host") or {}
print(sys.argv[1], "qps=%.0f" % d["value"], "ms/step=%.4f" % d["ms_per_step"], "scan_ms=%.4f" % r["scan_ms_per_launch"], "| h2h batches", (h.get("search_batches") or {}).get("library_pinned", {}).get("qps"), "pageable", (h.get("search_batches") or {}).get("pageable", {}).get("qps"), "| blocking pinned", (h.get("blocking_search") or {}).get("library_pinned", {}).get("qps"))'
run() { timeout -k 10 400 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-sweep --no-other-configs "$@" 2>gpurun_out/lanes_err.txt | python -c "$fmt" "$LABEL" || tail -3 gpurun_out/lanes_err.txt; }
for rep in 1 2; do
LABEL="lanes: stand-alone top-w" run
LABEL="lanes: fused (IVFADC_LANES_FUSE_TOPW)" IVFADC_LANES_FUSE_TOPW=1 run
done
LABEL="single-mode" run --single-mode
LABEL="hd two lanes? default" run --config hd --steps 20 --warmup 3
LABEL="hd fused" IVFADC_LANES_FUSE_TOPW=1 run --config hd --steps 20 --warmup 3
