"""Diagnostics of the matrix-core lower-bound rounds on the HD shape: survivors per query, scan time, table mode A/B.
usage (GPU box): python tools/lb_probe.py [nq] [w] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import ivfadc_jl_amd as pkg

cfg = dict(bench.CONFIGS[os.environ.get("LB_CONFIG", "hd")])
nq = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["nq"]
w = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["w"]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if os.environ.get("LB_N"):
    cfg["n"] = int(os.environ["LB_N"])
idx, _ = bench.build_synth(pkg, cfg, 0)
idx.set_tuning(0, 0)   # reads IVFADC_FORCE_PG
q = np.random.default_rng(11).standard_normal((nq, cfg["d"]), dtype=np.float32)
for mode in (0, 1):
    idx.set_table_mode(mode)
    idx.search_raw(q, K, w)
    idx.set_profiling(True)
    idx.reset_stats()
    for _ in range(5):
        r = idx.search_raw(q, K, w)
    st = idx.get_stats()
    idx.set_profiling(False)
    n = st["scan_launches"]
    print("table_mode=%d last_lb=%d scan_ms=%.4f coarse_ms=%.4f survivors/query=%.1f scanned/query=%.0f pruned=%.3f lds=%d" % (
        mode, st["last_lb"], st["scan_ms"] / n, st["coarse_ms"] / n, st["lb_survivors"] / max(1, st["queries"]),
        st["scanned_points"] / max(1, st["queries"]), st["pruned_points"] / max(1, st["scanned_points"]), st["last_scan_lds"]), flush=True)
