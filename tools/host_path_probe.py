import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivfadc_jl_amd as pkg
from ivfadc_jl_amd import _native as nat
n, d, kc, m, K = 1_000_000, 128, 1024, 8, 10
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024   # 1: the latency of a single knn_search(ivfadc, point, k) call
rng = np.random.default_rng(0)
cent = rng.random((kc, d), dtype=np.float32)
cbs = ((rng.random((m, 256, d // m), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
sizes = rng.multinomial(n, np.full(kc, 1.0 / kc)); off = np.zeros(kc + 1, np.int64); np.cumsum(sizes, out=off[1:])
idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, off, rng.integers(0, 256, (n, m), dtype=np.uint8), np.arange(n, dtype=np.uint32))
q = rng.random((nq, d), dtype=np.float32)
ids = np.zeros((nq, K), np.uint32); dists = np.zeros((nq, K), np.float32); counts = np.zeros(nq, np.int32)
L = nat.lib()
for w in (1, 8, 1, 8):
    for _ in range(20): L.ivfadc_search(idx._h, nq, nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32))
    t0 = time.perf_counter()
    for _ in range(200): L.ivfadc_search(idx._h, nq, nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32))
    raw = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): idx.search_raw(q, K, w)
    sr = (time.perf_counter() - t0) / 200
    print("w=%d raw ctypes call %.1f us, search_raw %.1f us" % (w, raw * 1e6, sr * 1e6))

# a run of batches from host memory (ivfadc_search_batches: one H2D, two batches in flight inside the library, one D2H) against the same
# batches as separate blocking calls -- the PCIe-inclusive rates of the reference-side calling patterns
if nq >= 64:
    nb = 16
    qs = rng.random((nb * nq, d), dtype=np.float32)
    bn = np.full(nb, nq, np.int64)
    oi = np.zeros((nb * nq, K), np.uint32); od = np.zeros((nb * nq, K), np.float32); oc = np.zeros(nb * nq, np.int32)
    for w in (1, 8):
        def run_batches():
            nat.check(L.ivfadc_search_batches(idx._h, nb, nat.ptr(bn, C.c_int64), nat.ptr(qs, C.c_float), K, w, nat.ptr(oi, C.c_uint32),
                                              nat.ptr(od, C.c_float), nat.ptr(oc, C.c_int32)))

        def run_loop():
            for b in range(nb):
                L.ivfadc_search(idx._h, nq, nat.ptr(qs[b * nq:(b + 1) * nq], C.c_float), K, w, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float),
                                nat.ptr(counts, C.c_int32))
        for f, name in ((run_loop, "16 blocking ivfadc_search calls"), (run_batches, "ivfadc_search_batches (16 batches)")):
            for _ in range(5): f()
            t0 = time.perf_counter()
            for _ in range(30): f()
            el = (time.perf_counter() - t0) / 30
            print("w=%d %-36s %.1f us per batch of %d, %.2f M q/s host to host" % (w, name, el / nb * 1e6, nq, nb * nq / el / 1e6))
