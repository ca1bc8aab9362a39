"""Host-to-host rates of the reference's calling patterns on the SIFT1M shape (knn_search takes and returns HOST arrays, index.jl:261-265):
blocking ivfadc_search per batch and ivfadc_search_batches over 16 batches, with pageable arrays, caller-registered arrays
(ivfadc_host_register) and library-allocated page-locked arrays (ivfadc_host_alloc), plus where the host time went (ivfadc_get_host_stats).
Run with IVFADC_HOST_LEGACY=1 for round 4's copy chain.  Usage: host_path_probe.py [nq=1024]"""
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
L = nat.lib()
nb = 16
qsrc = rng.random((nb * nq, d), dtype=np.float32)
print("mode: %s" % ("LEGACY copy chain (IVFADC_HOST_LEGACY)" if os.environ.get("IVFADC_HOST_LEGACY") else "kernel ingest + results written in place"))


def hstats():
    st = nat.HostStats()
    nat.check(L.ivfadc_get_host_stats(idx._h, C.byref(st)))
    return st


def arrays(kind):
    """(queries (nb*nq, d), ids, dists, counts, cleanup)"""
    if kind == "pageable":
        return qsrc.copy(), np.zeros((nb * nq, K), np.uint32), np.zeros((nb * nq, K), np.float32), np.zeros(nb * nq, np.int32), lambda: None
    if kind == "registered":
        a = (qsrc.copy(), np.zeros((nb * nq, K), np.uint32), np.zeros((nb * nq, K), np.float32), np.zeros(nb * nq, np.int32))
        for x in a:
            nat.host_register(x)
        return a + (lambda: [nat.host_unregister(x) for x in a],)
    p = (nat.PinnedArray((nb * nq, d), np.float32), nat.PinnedArray((nb * nq, K), np.uint32), nat.PinnedArray((nb * nq, K), np.float32),
         nat.PinnedArray(nb * nq, np.int32))
    p[0].a[...] = qsrc
    return p[0].a, p[1].a, p[2].a, p[3].a, lambda: [x.close() for x in p]


ref = {}
for kind in ("pageable", "registered", "library-pinned"):
    q, ids, dists, counts, cleanup = arrays(kind)
    bn = np.full(nb, nq, np.int64)
    for w in (1, 8):
        def run_loop():
            for b in range(nb):
                s = slice(b * nq, (b + 1) * nq)
                L.ivfadc_search(idx._h, nq, nat.ptr(q[s], C.c_float), K, w, nat.ptr(ids[s], C.c_uint32), nat.ptr(dists[s], C.c_float), nat.ptr(counts[s], C.c_int32))

        def run_batches():
            nat.check(L.ivfadc_search_batches(idx._h, nb, nat.ptr(bn, C.c_int64), nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                              nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        for f, name in ((run_loop, "16 blocking ivfadc_search calls"), (run_batches, "ivfadc_search_batches (16 batches)")):
            if nq < 64 and f is run_batches:
                continue
            for _ in range(5): f()
            L.ivfadc_reset_host_stats(idx._h)
            reps = 30
            t0 = time.perf_counter()
            for _ in range(reps): f()
            el = (time.perf_counter() - t0) / reps
            st = hstats()
            per = 1.0 / (reps * nb)
            print("%-14s w=%d %-36s %6.1f us per batch of %d, %6.2f M q/s | host us per batch: stage-in %.1f enqueue %.1f wait %.1f stage-out %.1f | "
                  "direct q/res %d/%d zero-copy %d of %d calls" % (kind, w, name, el / nb * 1e6, nq, nb * nq / el / 1e6, st.stage_in_us * per, st.enqueue_us * per,
                                                                     st.wait_us * per, st.stage_out_us * per, st.queries_direct, st.results_direct, st.zero_copy, st.calls))
            key = (w,)
            res = (ids.copy(), dists.copy(), counts.copy())
            if key in ref:
                assert all(np.array_equal(a, b) for a, b in zip(ref[key], res)), "results differ between memory kinds / entries"
            else:
                ref[key] = res
    cleanup()
print("results identical across memory kinds and entries: True")
# one batch of nq queries as ONE ivfadc_search_batches call of s sub-batches (two lanes inside): does splitting a blocking call pay?
if nq >= 256:
    q, ids, dists, counts, cleanup = arrays("library-pinned")
    for w in (1, 8):
        for split in (1, 2, 4):
            bn = np.full(split, nq // split, np.int64)
            f = lambda: nat.check(L.ivfadc_search_batches(idx._h, split, nat.ptr(bn, C.c_int64), nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                                          nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
            for _ in range(20): f()
            t0 = time.perf_counter()
            for _ in range(300): f()
            el = (time.perf_counter() - t0) / 300
            print("w=%d one batch of %d as %d sub-batches in one ivfadc_search_batches call: %.1f us" % (w, nq, split, el * 1e6))
    cleanup()
