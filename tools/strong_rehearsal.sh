#!/bin/bash
# One-GPU rehearsal of the 8-GPU strong-scaling regime on the SIFT1B shape (global batch 16 384): a rank of an N-GPU run
# searches 16384 / N queries against the full replica, so t(16384 / N) on ONE GPU is that rank's step time and
# t(16384) / (N t(16384 / N)) the predicted scaling efficiency (the all-gather of 84 B per query is noise at these step times).
# usage (GPU box): tools/strong_rehearsal.sh [out.jsonl] [extra bench args]
out=${1:-gpurun_out/strong_rehearsal.jsonl}; shift
: > $out
for w in 8 1; do for nq in 16384 8192 4096 2048; do
  timeout -k 10 280 python bench.py --config sift1b --nq $nq --w $w --steps 12 --warmup 3 --no-cpu-baseline --no-sweep "$@" 2>/dev/null >> $out || echo "{\"failed\": [$nq, $w]}" >> $out
done; done
python - $out <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
t = {}
for r in rows:
    if "failed" in r:
        print("FAILED", r); continue
    c = r["config"]; w = int(c["workload"].split("w=")[1].split(",")[0]); nq = c["global_batch"]
    ro = r["roofline"]
    t[(w, nq)] = r["ms_per_step"]
    print("w=%d nq=%5d  step %.3f ms  %.2f M q/s  scan %.3f ms  coarse %.3f ms  kernel %s chunk %s grid %s" % (
        w, nq, r["ms_per_step"], r["value"] / 1e6, ro["scan_ms_per_launch"], ro["coarse_ms_per_launch"], ro["kernel"], ro["chunk_points"], ro["scan_grid"]))
for w in (8, 1):
    if (w, 16384) in t:
        for n in (2, 4, 8):
            if (w, 16384 // n) in t:
                print("w=%d predicted efficiency at %d GPUs: t(16384) / (%d t(%d)) = %.3f  (speed-up %.2fx)" % (
                    w, n, n, 16384 // n, t[(w, 16384)] / (n * t[(w, 16384 // n)]), t[(w, 16384)] / t[(w, 16384 // n)]))
PY
