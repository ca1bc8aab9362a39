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
crowd = []
for _ in range(int(os.environ.get("CROWD", "0"))):      # a process that already owns streams (bench.py's, a serving framework's)
    st_ = torch.cuda.Stream()
    with torch.cuda.stream(st_):
        torch.zeros(16, device=dev).add_(1)
    crowd.append(st_)
torch.cuda.synchronize()
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
ts = []
t0 = time.perf_counter()
for _ in range(calls):
    ta = time.perf_counter(); f(); ts.append(time.perf_counter() - ta)
el = (time.perf_counter() - t0) / calls
print("ivfadc_search_batches, %d batches of %d: %.1f us per call, %.1f us per batch, %.2f M q/s" % (nb, nq, el * 1e6, el / nb * 1e6, nb * nq / el / 1e6))
tsu = np.array(ts) * 1e6
print("  per call us: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f | first 10 mean %.0f, last 10 mean %.0f | every 10th: %s" % (
    tsu.min(), np.percentile(tsu, 10), np.percentile(tsu, 50), np.percentile(tsu, 90), tsu.max(), tsu[:10].mean(), tsu[-10:].mean(),
    " ".join("%.0f" % v for v in tsu[::max(1, calls // 20)])))
if len(sys.argv) > 3 and sys.argv[3] == "split":
    # one batch of nq queries: blocking ivfadc_search against ONE ivfadc_search_batches call of s sub-batches (two lanes inside)
    one = lambda: nat.check(L.ivfadc_search(idx._h, nq, nat.ptr(pa[0].a, C.c_float), K, w, nat.ptr(pa[1].a, C.c_uint32), nat.ptr(pa[2].a, C.c_float), nat.ptr(pa[3].a, C.c_int32)))
    for _ in range(50): one()
    t0 = time.perf_counter()
    for _ in range(1000): one()
    print("blocking ivfadc_search(%d): %.1f us" % (nq, (time.perf_counter() - t0) / 1000 * 1e6))
    for split in (1, 2, 4):
        bs = np.full(split, nq // split, np.int64)
        g = lambda: nat.check(L.ivfadc_search_batches(idx._h, split, nat.ptr(bs, C.c_int64), nat.ptr(pa[0].a, C.c_float), K, w, nat.ptr(pa[1].a, C.c_uint32),
                                                      nat.ptr(pa[2].a, C.c_float), nat.ptr(pa[3].a, C.c_int32)))
        for _ in range(50): g()
        t0 = time.perf_counter()
        for _ in range(1000): g()
        print("one batch of %d as %d sub-batches in one ivfadc_search_batches call: %.1f us" % (nq, split, (time.perf_counter() - t0) / 1000 * 1e6))
