"""BASELINE.json configs [2], [3] and [4] at FULL size under `pytest -m gpu`.

Deep1B-shape (n=1e8, kc=65536, m=16, w=32, batch 10000), SIFT1B-shape (n=1e9, kc=8192, m=8, w in {1, 8},
batch 16384) and the HD text-embedding shape (n=1e7, d=768, kc=4096, m=48, w=8, batch 4096).

The index is synthesised on the device (`ivfadc_synth_lists`: counter-based code bytes the oracle can replay for
any probed list, ids = canonical position), exactly as bench.py builds these shapes.  Every test checks
  * the automatic plan against the CPU oracle on >= 64 sampled queries: ids bit-exact, Float32 distances within
    1e-4 relative (the north_star tolerance; reference semantics: src/index.jl:204-273);
  * the size-independent properties: full counts, ascending distances, distinct ids, ids inside [0, n);
  * the OTHER scan plan (forced) and the second, selection-free GPU implementation of the coarse stage (exact VALU
    kernel instead of the MFMA filter) agree with the automatic plan bit for bit;
  * idempotence: a repeated call returns the same bytes.
"""
import numpy as np
import pytest

import helpers
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

SEED_CODES = 20260101


def synth_index(native, d, n, kc, m, skew=False):
    rng = np.random.default_rng(7)
    cent = rng.standard_normal((kc, d), dtype=np.float32)
    cbs = rng.standard_normal((m, 256, d // m), dtype=np.float32)
    labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    p = np.full(kc, 1.0 / kc) if not skew else rng.dirichlet(np.full(kc, 0.5))
    sizes = rng.multinomial(n, p).astype(np.int64)
    off = np.zeros(kc + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    g = native.IVFADCIndex.from_arrays(cent, cbs, labels)
    g.synth_lists(off, SEED_CODES)
    o = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=SEED_CODES)
    return g, o


def properties(got, n, K, full=True):
    ids, dists, counts = got
    if full:
        assert (counts == K).all(), "every query of these shapes probes far more than K points"
    for r in np.nonzero(counts < K)[0]:
        assert np.isinf(dists[r, counts[r]:]).all()
    valid = np.arange(K)[None, :] < counts[:, None]
    dd = np.where(valid, dists, np.inf)
    assert (np.diff(dd, axis=1) >= 0).all(), "distances must ascend"
    assert (ids.astype(np.int64)[valid] < n).all()
    srt = np.sort(np.where(valid, ids.astype(np.int64), -1 - np.arange(K)[None, :]), axis=1)
    assert (np.diff(srt, axis=1) > 0).all(), "ids of one query must be distinct"


def same_bytes(a, b, what):
    for x, y, name in zip(a, b, ("ids", "dists", "counts")):
        assert np.array_equal(x, y), "%s: %s differ" % (what, name)


def oracle_sample(o, qs, got, K, w, nsample, seed, what):
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(qs.shape[0], nsample, replace=False))
    exp = o.knn_search(qs[pick], K, w, nthreads=ora.max_threads())
    helpers.assert_same_results(tuple(a[pick] for a in got), exp, rtol=1e-4, what=what)
    # the float order of the HIP path is the oracle's: distances are in fact identical bit for bit
    assert np.array_equal(got[1][pick], exp[1]), what + ": distances not bit-identical"


def test_deep1b_shape_full_size(native):
    """configs[2]: d=96 n=1e8 kc=65536 k=256 m=16, w=32, batch=10000 (query-major scan qscan_kernel<16,6,2>, MFMA coarse
    filter + tiled top-w at kc = 65536)."""
    d, n, kc, m, nq, w, K = 96, 100_000_000, 65536, 16, 10000, 32, 10
    g, o = synth_index(native, d, n, kc, m)
    assert len(g) == n
    qs = np.random.default_rng(11).standard_normal((nq, d), dtype=np.float32)
    auto = g.search_raw(qs, K, w)
    st = g.get_stats()
    assert st["last_qg"] == 0 and st["coarse_mfma"] == 1, st          # the plan this shape is benchmarked on
    properties(auto, n, K)
    oracle_sample(o, qs, auto, K, w, 64, 1, "deep1b auto")
    same_bytes(auto, g.search_raw(qs, K, w), "deep1b repeat")
    g.set_tuning(4, 0)                                                  # forced list-major, 4 queries per code stream
    other = g.search_raw(qs, K, w)
    assert g.get_stats()["last_qg"] == 4
    same_bytes(auto, other, "deep1b list-major vs query-major")
    g.set_tuning(0, 0)
    g.set_coarse_mode(1)                                                # exact VALU coarse kernel, no filter
    same_bytes(auto, g.search_raw(qs, K, w), "deep1b exact coarse vs MFMA filter")
    g.set_coarse_mode(0)
    # a different batch split (sub-batching must be invisible)
    part = g.search_raw(qs[:777], K, w)
    same_bytes(tuple(a[:777] for a in auto), part, "deep1b split")
    # K = 100: the LDS selectors, whose four waves share quarter keys (publish_bound) -- both plans, a slice of the batch against the oracle
    k100 = g.search_raw(qs[:2000], 100, w)
    properties(k100, n, 100)
    oracle_sample(o, qs[:2000], k100, 100, w, 32, 21, "deep1b K=100")
    g.set_tuning(4, 0)
    same_bytes(k100, g.search_raw(qs[:2000], 100, w), "deep1b K=100 list-major vs query-major")
    g.set_tuning(0, 0)


@pytest.mark.parametrize("w", [1, 8])
def test_sift1b_shape_full_size(native, w):
    """configs[3]: d=128 n=1e9 kc=8192 k=256 m=8, batch=16384 (list-major scan_kernel<8,16,QG> over 122 k-point lists)."""
    d, n, kc, m, nq, K = 128, 1_000_000_000, 8192, 8, 16384, 10
    g, o = synth_index(native, d, n, kc, m)
    assert len(g) == n
    qs = np.random.default_rng(11).standard_normal((nq, d), dtype=np.float32)
    auto = g.search_raw(qs, K, w)
    st = g.get_stats()
    assert st["last_qg"] >= 1, st                                       # list-major
    properties(auto, n, K)
    oracle_sample(o, qs, auto, K, w, 64, 2 + w, "sift1b auto w=%d" % w)
    same_bytes(auto, g.search_raw(qs, K, w), "sift1b repeat")
    # the other group widths of the list-major plan, and the query-major plan on a slice of the batch (one workgroup
    # per query walks 122 k-point lists: correct, slow)
    for qg in (1, 2, 4, 8):        # 8: the narrow-field kernel (nfscan.hip.h), eight queries per code stream
        if qg == st["last_qg"]:
            continue
        g.set_tuning(qg, 0)
        same_bytes(auto, g.search_raw(qs, K, w), "sift1b qg=%d vs auto" % qg)
        assert g.get_stats()["last_qg"] == qg
    g.set_tuning(-1, 0)
    sl = g.search_raw(qs[:2048], K, w)
    assert g.get_stats()["last_qg"] == 0
    same_bytes(tuple(a[:2048] for a in auto), sl, "sift1b query-major vs list-major")
    # K = 100 on 122 k-point lists: the LDS selectors' shared quarter keys do most of the pruning here (every group width, then the oracle)
    g.set_tuning(0, 0)
    k100 = g.search_raw(qs[:4096], 100, w)
    properties(k100, n, 100)
    oracle_sample(o, qs[:4096], k100, 100, w, 32, 30 + w, "sift1b K=100 w=%d" % w)
    for qg in (1, 2, 4):
        g.set_tuning(qg, 0)
        same_bytes(k100, g.search_raw(qs[:4096], 100, w), "sift1b K=100 qg=%d vs auto" % qg)


def test_hd_shape_full_size(native):
    """configs[4]: d=768 n=1e7 kc=4096 k=256 m=48, w=8, batch=4096 (qscan_kernel<48,16,1>, 48 KB tables)."""
    d, n, kc, m, nq, w, K = 768, 10_000_000, 4096, 48, 4096, 8, 10
    g, o = synth_index(native, d, n, kc, m)
    assert len(g) == n
    qs = np.random.default_rng(11).standard_normal((nq, d), dtype=np.float32)
    auto = g.search_raw(qs, K, w)
    st = g.get_stats()
    assert st["last_qg"] == 0 and st["coarse_mfma"] == 1, st
    properties(auto, n, K)
    oracle_sample(o, qs, auto, K, w, 64, 5, "hd auto")
    same_bytes(auto, g.search_raw(qs, K, w), "hd repeat")
    g.set_tuning(4, 0)                                                  # list-major (the width yields to the LDS limit)
    other = g.search_raw(qs, K, w)
    assert g.get_stats()["last_qg"] >= 1
    same_bytes(auto, other, "hd list-major vs query-major")
    g.set_tuning(0, 0)
    g.set_coarse_mode(1)
    same_bytes(auto, g.search_raw(qs, K, w), "hd exact coarse vs MFMA filter")
    g.set_coarse_mode(0)
    # K = 2000 at m = 48: the selector buffers do not fit LDS next to 48 KB of tables -- the library must route to the
    # dump-and-sort path instead of failing; 8 queries against the oracle
    big = g.search_raw(qs[:8], 2000, 2)
    helpers.assert_same_results(big, o.knn_search(qs[:8], 2000, 2, nthreads=ora.max_threads()), what="hd K=2000")


def test_skewed_lists_sift1b_slice(native):
    """Dirichlet(0.5)-skewed list sizes at n = 1e8 (lists from 0 to ~1e5 points): both plans against the oracle."""
    d, n, kc, m, nq, w, K = 128, 100_000_000, 8192, 8, 4096, 8, 10
    g, o = synth_index(native, d, n, kc, m, skew=True)
    qs = np.random.default_rng(12).standard_normal((nq, d), dtype=np.float32)
    auto = g.search_raw(qs, K, w)
    properties(auto, n, K, full=False)
    oracle_sample(o, qs, auto, K, w, 48, 9, "skewed auto")
    for mode in (-1, 1, 4, 8):
        g.set_tuning(mode, 0)
        same_bytes(auto, g.search_raw(qs, K, w), "skewed mode %d" % mode)
