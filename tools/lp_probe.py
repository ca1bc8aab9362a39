"""Where a rank's time goes in --partition lists (one GPU playing rank r of N): kernel times by HIP events.  usage: python tools/lp_probe.py [N] [nq] [w]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import ivfadc_jl_amd as pkg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = dict(bench.CONFIGS["sift1b"])
nq = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["nq"]
w = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["w"]
K = 10
idx, _ = bench.build_synth(pkg, cfg, 0)
dev = torch.device("cuda:0")
idx.set_stream(torch.cuda.current_stream().cuda_stream)
q = torch.as_tensor(np.random.default_rng(11).standard_normal((nq, cfg["d"]), dtype=np.float32)).to(dev)
keys = torch.zeros((nq, K), dtype=torch.int64, device=dev)
cnts = torch.zeros(nq, dtype=torch.int32, device=dev)
ids = torch.zeros(nq * K, dtype=torch.int32, device=dev); dd = torch.zeros(nq * K, dtype=torch.float32, device=dev)
for qg in [int(x) for x in os.environ.get("LP_QGS", "0").split(",")]:
  for chunk in [int(x) for x in os.environ.get("LP_CHUNKS", "0").split(",")]:
    for parts in (1, N):
        idx.set_tuning(qg, chunk)
        idx.set_list_partition(parts, 0)
        fn = (lambda: idx.search_device(nq, q.data_ptr(), K, w, ids.data_ptr(), dd.data_ptr(), cnts.data_ptr())) if parts == 1 else \
             (lambda: idx.search_device_partial(nq, q.data_ptr(), K, w, keys.data_ptr(), cnts.data_ptr()))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 10
        idx.set_profiling(True); idx.reset_stats()
        for _ in range(5): fn()
        st = idx.get_stats(); idx.set_profiling(False)
        n = max(1, st["scan_launches"])
        print("parts=%d qg=%d(%d) chunk=%d step=%.3f ms scan=%.3f coarse=%.3f other=%.3f scanned/query=%.0f grid=%d nf=%d" % (
            parts, qg, st["last_qg"], st["last_chunk"], el * 1e3, st["scan_ms"] / n, st["coarse_ms"] / n, el * 1e3 - st["scan_ms"] / n - st["coarse_ms"] / n,
            st["scanned_points"] / max(1, st["queries"]), st["last_scan_grid"], st["last_nf"]), flush=True)
