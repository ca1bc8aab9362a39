"""PCIe-inclusive rate of the host-pointer entry point (ivfadc_search) vs the device-resident one, SIFT1M-shape
with random codes (the timing does not depend on the index being trained)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivfadc_jl_amd as pkg
n, d, kc, m, nq, K = 1_000_000, 128, 1024, 8, 1024, 10
rng = np.random.default_rng(0)
cent = rng.random((kc, d), dtype=np.float32)
cbs = ((rng.random((m, 256, d // m), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
sizes = rng.multinomial(n, np.full(kc, 1.0 / kc)); off = np.zeros(kc + 1, np.int64); np.cumsum(sizes, out=off[1:])
idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, off, rng.integers(0, 256, (n, m), dtype=np.uint8), np.arange(n, dtype=np.uint32))
q = rng.random((nq, d), dtype=np.float32)
for w in (1, 8):
    for _ in range(20): idx.search_raw(q, K, w)
    t0 = time.perf_counter()
    for _ in range(200): idx.search_raw(q, K, w)
    host = (time.perf_counter() - t0) / 200
    qd = torch.as_tensor(q).cuda(); out = torch.zeros(nq * (2 * K + 1), dtype=torch.int32, device="cuda")
    b = out.data_ptr()
    for _ in range(20): idx.search_device(nq, qd.data_ptr(), K, w, b, b + nq * K * 4, b + 2 * nq * K * 4)
    idx.sync(); t0 = time.perf_counter()
    for _ in range(200): idx.search_device(nq, qd.data_ptr(), K, w, b, b + nq * K * 4, b + 2 * nq * K * 4)
    idx.sync(); dev = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): idx.search_device(nq, qd.data_ptr(), K, w, b, b + nq * K * 4, b + 2 * nq * K * 4)
    enq = (time.perf_counter() - t0) / 200; idx.sync()
    print("w=%d: host-pointer ivfadc_search %.1f us/batch (%.2f M q/s) | device-resident %.1f us/batch (%.2f M q/s) | host enqueue cost %.1f us/batch"
          % (w, host * 1e6, nq / host / 1e6, dev * 1e6, nq / dev / 1e6, enq * 1e6))
