"""How much of a SIFT1M-shape step is launch ramp and tail?  Two handles on the same data (each with its own stream and scratch), batches
dealt to them in turn, against one handle with and without the next-batch hint.  No library change: a caller-level experiment.
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
# the second replica: same quantizers, same data, same encode path
cent, cbs, labels = idx0.quantizers() if hasattr(idx0, "quantizers") else (None, None, None)
if cent is None:
    idx1, _ = bench.build_trained(pkg, cfg, dev, 0, None)
else:
    idx1 = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=0)
    idx1._append(x.cpu().numpy(), np.arange(cfg["n"], dtype=np.uint32))
q = bench.global_queries(cfg, nq, dev)
outs = []
for _ in range(2):
    outs.append((torch.zeros(nq * K, dtype=torch.int32, device=dev), torch.zeros(nq * K, dtype=torch.float32, device=dev),
                 torch.zeros(nq, dtype=torch.int32, device=dev)))


def run(name, step):
    for i in range(20):
        step(i)
    idx0.sync(); idx1.sync(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        idx0.sync(); idx1.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print("%-34s %.2f us/step  %.2f M q/s" % (name, best * 1e6, nq / best / 1e6), flush=True)


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


run("one handle, plain", plain)
run("one handle, next-batch hint", hinted)
run("two handles in turn, plain", two)
run("two handles in turn, hinted", two_hinted)
a = [t.cpu().numpy() for t in outs[0]]
b = [t.cpu().numpy() for t in outs[1]]
print("results of the two handles identical:", all(np.array_equal(u, v) for u, v in zip(a, b)))
