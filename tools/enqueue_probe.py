"""On the GPU box: is the SIFT1M headline loop bound by the HOST?  Per-step host time of the enqueue calls (no synchronisation inside the
loop) against the per-step time of the whole run, for 1 / 2 / 3 lanes, with and without the next-batch hint, and for the C entry alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import ivfadc_jl_amd as pkg

pkg.load_library()
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["sift1m"]
idx, x = bench.build_trained(pkg, cfg, dev, 0, None, "mixture")
nq, K, w = 1024, 10, 8
q = bench.global_queries(cfg, nq, dev, "mixture").contiguous()
idx.set_stream(torch.cuda.current_stream().cuda_stream)
for lanes in (1, 2, 3):
    ll = [idx] + [idx.clone_view() for _ in range(lanes - 1)]
    outs = [(torch.zeros(nq * K, dtype=torch.int32, device=dev), torch.zeros(nq * K, dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev)) for _ in ll]
    for hint in (True, False):
        def step(i):
            ln = ll[i % lanes]
            o = outs[i % lanes]
            if hint:
                ln.set_query_token(1)
                ln.set_next_queries(nq, q.data_ptr(), 1)
            ln.search_device(nq, q.data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        for i in range(300):
            step(i)
        torch.cuda.synchronize()
        N = 3000
        t0 = time.perf_counter()
        for i in range(N):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("lanes=%d hint=%d: host enqueue %.2f us/step, whole run %.2f us/step (%.1f M q/s); GPU still busy %.0f us after the last enqueue" % (
            lanes, hint, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, nq * N / (t2 - t0) / 1e6, (t2 - t1) * 1e6))
