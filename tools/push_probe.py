"""push! latency: one point appended then one small search, repeated -- in place on the device vs the
re-layout path (IVFADC_NO_INPLACE_APPEND=1 in the environment).  SIFT1M-shape random lists.
usage: python tools/push_probe.py [n] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivfadc_jl_amd as ivf  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    d, kc, m = 128, 1024, 8
    rng = np.random.default_rng(0)
    cent = rng.random((kc, d), dtype=np.float32)
    cbs = (rng.random((m, 256, d // m), dtype=np.float32) - 0.5) * 0.2
    labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    sizes = np.full(kc, n // kc, np.int64)
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(sizes, out=offsets[1:])
    codes = rng.integers(0, 256, (int(offsets[-1]), m), dtype=np.uint8)
    ids = np.arange(int(offsets[-1]), dtype=np.uint32)
    idx = ivf.IVFADCIndex.from_arrays(cent, cbs, labels, offsets, codes, ids)
    q = rng.random((16, d), dtype=np.float32)
    idx.search_raw(q, 10, 8)
    pts = rng.random((reps + 8, d), dtype=np.float32)
    nid = int(offsets[-1])
    for i in range(8):   # warm-up
        idx._append(pts[i:i + 1], np.array([nid], np.uint32)); nid += 1
        idx.search_raw(q, 10, 8)
    t_push = t_search = 0.0
    for i in range(8, reps + 8):
        t0 = time.perf_counter()
        idx._append(pts[i:i + 1], np.array([nid], np.uint32)); nid += 1
        t1 = time.perf_counter()
        idx.search_raw(q, 10, 8)
        t2 = time.perf_counter()
        t_push += t1 - t0
        t_search += t2 - t1
    st = idx.get_stats()
    print({"n": n, "reps": reps, "inplace_appends": st["inplace_appends"],
           "mode": "relayout" if os.environ.get("IVFADC_NO_INPLACE_APPEND") else "inplace",
           "push_us": round(1e6 * t_push / reps, 1), "search_after_push_us": round(1e6 * t_search / reps, 1)})


if __name__ == "__main__":
    main()
