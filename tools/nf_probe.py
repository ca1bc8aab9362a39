"""Diagnostics of the narrow-field list-major kernel on the SIFT1B shape: scan time, exact evaluations per query, chunk scaling.
usage (GPU box): python tools/nf_probe.py [nq] [w]      env: NF_N (points), NF_QGS ("8,4"), NF_CHUNKS ("0,32768")"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import ivfadc_jl_amd as pkg

if os.environ.get("NF_DBG"):
    import ivfadc_jl_amd._native as nat
    nat.SO_PATH = nat.SO_PATH.replace("libivfadc_hip.so", "libivfadc_hip_dbg.so")
cfg = dict(bench.CONFIGS["sift1b"])
nq = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["nq"]
w = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["w"]
if os.environ.get("NF_N"):
    cfg["n"] = int(os.environ["NF_N"])
idx, _ = bench.build_synth(pkg, cfg, 0)
q = np.random.default_rng(11).standard_normal((nq, cfg["d"]), dtype=np.float32)
for qg in [int(x) for x in os.environ.get("NF_QGS", "8,4").split(",")]:
    for chunk in [int(x) for x in os.environ.get("NF_CHUNKS", "0").split(",")]:
        idx.set_tuning(qg, chunk)
        idx.search_raw(q, 10, w)
        idx.set_profiling(True)
        idx.reset_stats()
        for _ in range(3):
            idx.search_raw(q, 10, w)
        st = idx.get_stats()
        idx.set_profiling(False)
        n = st["scan_launches"]
        print("qg=%d chunk=%d (used %d) nf=%d scan_ms=%.4f coarse_ms=%.4f exact_evals/query=%.1f scanned/query=%.0f grid=%d lds=%d flags=%s" % (
            qg, chunk, st["last_chunk"], st["last_nf"], st["scan_ms"] / n, st["coarse_ms"] / n, st["lb_survivors"] / max(1, st["queries"]),
            st["scanned_points"] / max(1, st["queries"]), st["last_scan_grid"], st["last_scan_lds"], os.environ.get("IVFADC_DEBUG_FLAGS", "0")), flush=True)
