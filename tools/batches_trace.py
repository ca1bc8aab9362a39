"""ivfadc_search_batches on the bench's SIFT1M-shape index (trained, Gaussian mixture), library-pinned arrays: a short run for
rocprofv3 --kernel-trace (tools/batches_trace.sh), or timing alone.  usage: batches_trace.py [calls=20] [nb=16]"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ivfadc_jl_amd as pkg
from ivfadc_jl_amd import _native as nat
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = dict(bench.CONFIGS["sift1m"])
dev = torch.device("cuda", 0)
idx, x = bench.build_trained(pkg, cfg, dev, 0, None, "mixture")
nq, d, K, w = cfg["nq"], cfg["d"], 10, cfg["w"]
qsrc = bench.global_queries(cfg, nb * nq, dev).cpu().numpy()
del x
L = nat.lib()
pa = (nat.PinnedArray((nb * nq, d), np.float32), nat.PinnedArray((nb * nq, K), np.uint32), nat.PinnedArray((nb * nq, K), np.float32), nat.PinnedArray(nb * nq, np.int32))
pa[0].a[...] = qsrc
bn = np.full(nb, nq, np.int64)
f = lambda: nat.check(L.ivfadc_search_batches(idx._h, nb, nat.ptr(bn, C.c_int64), nat.ptr(pa[0].a, C.c_float), K, w, nat.ptr(pa[1].a, C.c_uint32),
                                              nat.ptr(pa[2].a, C.c_float), nat.ptr(pa[3].a, C.c_int32)))
for _ in range(10): f()
t0 = time.perf_counter()
for _ in range(calls): f()
el = (time.perf_counter() - t0) / calls
print("ivfadc_search_batches, %d batches of %d: %.1f us per call, %.1f us per batch, %.2f M q/s" % (nb, nq, el * 1e6, el / nb * 1e6, nb * nq / el / 1e6))
