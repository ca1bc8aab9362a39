"""How much of a SIFT1M-shape step is launch ramp and tail?  The index and views of it (ivfadc_clone_view: each with its own stream and
workspace), batches dealt to them in turn, against one handle with and without the next-batch hint.
usage (GPU box): python tools/two_stream_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import ivfadc_jl_amd as pkg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(bench.CONFIGS["sift1m"])
dev = torch.device("cuda:0")
K, w, nq = 10, cfg["w"], cfg["nq"]
idx0, x = bench.build_trained(pkg, cfg, dev, 0, None)
if os.environ.get("COARSE_MODE"):
    idx0.set_coarse_mode(int(os.environ["COARSE_MODE"]))   # (views copy the setting)
lanes = [idx0] + [idx0.clone_view() for _ in range(3)]   # views: the same device arrays, own stream and workspace
idx1 = lanes[1]
q = bench.global_queries(cfg, nq, dev)
outs = []
for _ in range(4):
    outs.append((torch.zeros(nq * K, dtype=torch.int32, device=dev), torch.zeros(nq * K, dtype=torch.float32, device=dev),
                 torch.zeros(nq, dtype=torch.int32, device=dev)))


def run(name, step):
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    best, issue = 1e9, 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
        issue = min(issue, (t1 - t0) / steps)
    print("%-34s %.2f us/step  %.2f M q/s   (host time to issue a step: %.2f us)" % (name, best * 1e6, nq / best / 1e6, issue * 1e6), flush=True)


def plain(i):
    o = outs[0]
    idx0.search_device(nq, q.data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())


def hinted(i):
    idx0.set_query_token(1)
    idx0.set_next_queries(nq, q.data_ptr(), 1)
    plain(i)


def two(i):
    h, o = (idx0, outs[0]) if (i & 1) == 0 else (idx1, outs[1])
    h.search_device(nq, q.data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())


def two_hinted(i):
    h, o = (idx0, outs[0]) if (i & 1) == 0 else (idx1, outs[1])
    h.set_query_token(1)
    h.set_next_queries(nq, q.data_ptr(), 1)
    h.search_device(nq, q.data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())


def lanes_n(n, hint):
    def f(i):
        h, o = lanes[i % n], outs[i % n]
        if hint:
            h.set_query_token(1)
            h.set_next_queries(nq, q.data_ptr(), 1)
        h.search_device(nq, q.data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    return f


only = os.environ.get("PROBE_ONLY")          # e.g. "2h": two lanes, hinted, nothing else (for a kernel trace of that regime)
if only:
    run("%s lanes, %s" % (only[0], "hinted" if only.endswith("h") else "plain"), lanes_n(int(only[0]), only.endswith("h")))
    sys.exit(0)
run("one handle, plain", plain)
run("one handle, next-batch hint", hinted)
for n in (2, 3, 4):
    run("%d lanes (index + views), plain" % n, lanes_n(n, False))
    run("%d lanes (index + views), hinted" % n, lanes_n(n, True))
a = [t.cpu().numpy() for t in outs[0]]
for k in range(1, 4):
    b = [t.cpu().numpy() for t in outs[k]]
    print("results of lane %d identical to lane 0:" % k, all(np.array_equal(u, v) for u, v in zip(a, b)))
