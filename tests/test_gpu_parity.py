"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: neighbour ids bit-exact, Float32 distances within 1e-4 relative
(BASELINE.json north_star); in practice the float order is identical and distances match
bit for bit, which the tests also record."""
import numpy as np
import pytest

import helpers
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def gpu_index(native, oidx, index_type=np.uint32):
    return native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids,
                                          index_type=index_type)


def check(native, oidx, qs, K, w, gidx=None, what=""):
    gidx = gidx or gpu_index(native, oidx)
    got = gidx.search_raw(qs, K, w)
    exp = oidx.knn_search(qs, K, w)
    helpers.assert_same_results(got, exp, what=what)
    return got, exp


SHAPES = [
    # seed, n,    d,   kc,  m,  ksub, K,  w,  nq
    (1, 1000, 50, 100, 10, 256, 3, 1, 33),      # README toy shape (generic-m kernel, m=10)
    (2, 1000, 50, 100, 10, 256, 3, 7, 64),
    (3, 243, 10, 100, 2, 16, 3, 2, 10),         # test/index.jl helper shape
    (4, 5000, 128, 64, 8, 256, 10, 1, 100),     # SIFT-like m=8 kernel
    (5, 5000, 128, 64, 8, 256, 10, 8, 200),
    (6, 4000, 96, 50, 16, 256, 10, 5, 77),      # Deep-like m=16 kernel
    (7, 3000, 96, 40, 48, 64, 10, 4, 40),       # m=48 kernel, ksub < 256
    (8, 2000, 64, 16, 32, 256, 5, 16, 50),      # m=32, w == kc
    (9, 2000, 128, 30, 64, 256, 4, 3, 20),      # m=64
    (10, 1500, 12, 9, 4, 32, 6, 3, 25),         # m=4 (generic kernel), permuted labels
    (11, 800, 6, 7, 1, 256, 8, 2, 31),          # m=1 (reference default)
    (12, 600, 20, 33, 20, 8, 5, 40, 19),        # dsub=1, k=8 as in test/search.jl, w clamped to kc
]


@pytest.mark.parametrize("mode", [-1, 4])      # -1: query-major scan kernel, 4: list-major, 4 queries per code stream
@pytest.mark.parametrize("seed,n,d,kc,m,ksub,K,w,nq", SHAPES)
def test_search_matches_oracle(native, seed, n, d, kc, m, ksub, K, w, nq, mode):
    oidx, data = helpers.build_index(seed, n, d, kc, m, ksub, label_perm=(seed % 2 == 0))
    rng = np.random.default_rng(seed)
    qs = np.concatenate([rng.random((nq - 3, d), dtype=np.float32), data[:3]])
    gidx = gpu_index(native, oidx)
    gidx.set_tuning(mode, 0)
    got, exp = check(native, oidx, qs, K, w, gidx)
    assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf])   # distances are in fact bit-identical
    # K > 64 switches every selector from the register form to the LDS/bitonic form
    check(native, oidx, qs[:8], 100, w, gidx, what="K=100")


@pytest.mark.parametrize("qg", [1, 2, 4])
@pytest.mark.parametrize("m", [8, 16, 10])
def test_query_group_and_chunk_variants(native, qg, m):
    """Same answers whatever the work split: 1/2/4 queries per code stream, several chunks per list."""
    d = {8: 64, 16: 64, 10: 50}[m]
    oidx, _ = helpers.build_index(20 + m, 9000, d, 6, m, 256, mode="random")
    rng = np.random.default_rng(m)
    qs = rng.random((57, d), dtype=np.float32)
    gidx = gpu_index(native, oidx)
    gidx.set_tuning(qg, 1024)                     # ~1500-point lists -> 2 chunks each
    check(native, oidx, qs, 10, 4, gidx, what="qg=%d m=%d" % (qg, m))
    st = gidx.get_stats()
    assert st["last_qg"] == qg and st["last_chunk"] == 1024
    check(native, oidx, qs, 200, 4, gidx, what="K=200 qg=%d m=%d" % (qg, m))
    gidx.set_tuning(-1, 0)
    check(native, oidx, qs, 10, 4, gidx, what="query-major m=%d" % m)
    assert gidx.get_stats()["last_qg"] == 0


def test_ties_everywhere(native):
    """PQ makes exact ties common: only 3 distinct codes.  Order must be (distance, probe rank, list position)."""
    oidx, _ = helpers.build_index(31, 6000, 32, 5, 8, 256, mode="random", ndistinct=3)
    rng = np.random.default_rng(31)
    qs = rng.random((40, 32), dtype=np.float32)
    for qg in (1, 4, -1):
        gidx = gpu_index(native, oidx)
        gidx.set_tuning(qg, 1024)
        got, exp = check(native, oidx, qs, 25, 3, gidx, what="ties qg=%d" % qg)
        check(native, oidx, qs, 70, 3, gidx, what="ties K=70 qg=%d" % qg)
    assert len(np.unique(exp[1][0])) < 25


def test_duplicate_centroids_tie_to_lower_cluster(native):
    oidx, _ = helpers.build_index(32, 900, 16, 12, 8, 256, mode="random")
    oidx.centroids[5] = oidx.centroids[2]
    oidx.centroids[9] = oidx.centroids[2]
    rng = np.random.default_rng(32)
    check(native, oidx, rng.random((30, 16), dtype=np.float32), 10, 2)


def test_fewer_than_k_and_empty_lists(native):
    oidx, _ = helpers.build_index(33, 40, 8, 30, 8, 256, mode="random")     # most lists hold 0-3 points
    rng = np.random.default_rng(33)
    qs = rng.random((50, 8), dtype=np.float32)
    got, exp = check(native, oidx, qs, 10, 1)
    assert (got[2] < 10).any()
    check(native, oidx, qs, 10, 30)
    g4 = gpu_index(native, oidx)
    g4.set_tuning(4, 0)
    check(native, oidx, qs, 10, 1, g4)
    check(native, oidx, qs, 10, 30, g4)
    # completely empty index: zero neighbours for every query
    e = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, np.zeros(31, np.int64),
                        np.zeros((0, 8), np.uint8), np.zeros(0, np.uint32))
    got, _ = check(native, e, qs, 5, 3)
    assert (got[2] == 0).all()


def test_large_k_and_w(native):
    oidx, _ = helpers.build_index(34, 20000, 32, 300, 8, 256, mode="random")
    rng = np.random.default_rng(34)
    qs = rng.random((9, 32), dtype=np.float32)
    for mode in (-1, 2):
        gidx = gpu_index(native, oidx)
        gidx.set_tuning(mode, 0)
        check(native, oidx, qs, 1000, 200, gidx, what="K=1000 w=200")
        check(native, oidx, qs, 2048, 300, gidx, what="K=2048 w=kc")
        check(native, oidx, qs, 1, 1, gidx, what="K=1")
        check(native, oidx, qs, 64, 64, gidx, what="K=64 w=64")
        check(native, oidx, qs, 65, 65, gidx, what="K=65 w=65")


def test_assertions_and_limits(native):
    """test/search.jl:14-15 and the library limits."""
    oidx, _ = helpers.build_index(35, 100, 8, 4, 2, 16)
    gidx = gpu_index(native, oidx)
    q = np.zeros(8, np.float32)
    with pytest.raises(AssertionError):
        native.knn_search(gidx, q, 0)
    with pytest.raises(AssertionError):
        native.knn_search(gidx, q, 1, w=0)
    with pytest.raises(AssertionError):
        gidx.search_raw(q[None], 0, 1)           # enforced inside the C ABI too
    with pytest.raises(AssertionError):
        gidx.search_raw(q[None], 1, 0)
    ids, dists, cnt = gidx.search_raw(q[None], 5000, 1)     # K beyond the selection kernels: generic path, "at most k"
    exp = oidx.knn_search(q[None], 5000, 1)
    helpers.assert_same_results((ids, dists, cnt), exp, what="K=5000 on 100 points")


def test_api_types(native):
    """test/search.jl:11-13,19-21: single query -> (Vector{I}, Vector{T}); batch -> vectors of vectors."""
    oidx, _ = helpers.build_index(36, 243, 10, 100, 2, 16)
    gidx = gpu_index(native, oidx, index_type=np.uint16)
    rng = np.random.default_rng(36)
    ids, dists = native.knn_search(gidx, rng.random(10, dtype=np.float32), 3, w=2)
    assert ids.dtype == np.uint16 and dists.dtype == np.float32 and ids.ndim == 1 and len(ids) <= 3
    idl, dl = native.knn_search(gidx, [rng.random(10, dtype=np.float32) for _ in range(10)], 3, w=2)
    assert isinstance(idl, list) and len(idl) == 10 and all(a.dtype == np.uint16 for a in idl)
    assert isinstance(dl, list) and all(a.dtype == np.float32 for a in dl)
    assert repr(gidx) == "IVFADCIndex, naive coarse quantizer, 4-byte encoding (2 + 1×2), 243 Float32 vectors"
    assert gidx.size == (10, 243) and len(gidx) == 243


def test_encode_and_push_match_oracle(native):
    oidx, data = helpers.build_index(37, 3000, 64, 40, 8, 256, label_perm=True)
    gidx = native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels)
    glist, gcodes = gidx.encode(data)
    olist, ocodes = oidx.encode(data)
    assert np.array_equal(glist, olist) and np.array_equal(gcodes, ocodes)
    # building by push!-style appends reproduces the oracle-built lists exactly
    gidx._append(data, np.arange(3000, dtype=np.uint32))
    offsets, codes, ids = gidx._lists()
    assert np.array_equal(offsets, oidx.offsets) and np.array_equal(codes, oidx.codes) and np.array_equal(ids, oidx.ids)
    rng = np.random.default_rng(37)
    check(native, oidx, rng.random((20, 64), dtype=np.float32), 10, 3, gidx)
    # single push!: id = length(ivfadc), appended at the end of its list (utils.jl:139-143)
    p = rng.random(64, dtype=np.float32)
    native.push(gidx, p)
    assert len(gidx) == 3001
    pl, pc = oidx.encode(p)
    lst = gidx.inverse_index[int(pl[0])]
    assert lst.idxs[-1] == 3000 and np.array_equal(lst.codes[-1], pc[0])
    ids1, _ = native.knn_search(gidx, p, 1)
    assert len(ids1) == 1


def _oracle_of(gidx, oidx):
    offsets, codes, ids = gidx._lists()
    return ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, offsets, codes, ids)


@pytest.mark.parametrize("shape", [(64, 8, 256), (96, 16, 256), (50, 10, 256), (12, 4, 32)])
def test_interleaved_push_and_search(native, shape):
    """push! between searches: appends are written into the spare capacity behind each list on the device
    (utils.jl:139-145 semantics: end of the list, in call order); a list that outgrows its capacity forces
    a re-layout.  Every state is searched and compared with the oracle over the same lists."""
    d, m, ksub = shape
    oidx, data = helpers.build_index(300 + d, 1500, d, 24, m, ksub, label_perm=(ksub < 256))
    gidx = gpu_index(native, oidx)
    rng = np.random.default_rng(300 + d)
    qs = rng.random((40, d), dtype=np.float32)
    check(native, oidx, qs, 10, 5, gidx)                      # device layout is current from here on
    assert gidx.get_stats()["inplace_appends"] == 0
    nid = 1500
    inplace_seen = relayout_seen = 0
    for step, batch in enumerate([1, 1, 3, 17, 1, 64, 2, 500, 1, 5, 2000, 1, 1]):
        pts = data[rng.integers(0, 1500, batch)] + 0.01 * rng.standard_normal((batch, d)).astype(np.float32)
        before = gidx.get_stats()["inplace_appends"]
        gidx._append(pts, np.arange(nid, nid + batch, dtype=np.uint32))
        nid += batch
        if gidx.get_stats()["inplace_appends"] == before + 1:
            inplace_seen += 1
        else:
            relayout_seen += 1
        assert len(gidx) == nid
        onow = _oracle_of(gidx, oidx)
        for K, w in ((10, 5), (3, 24), (100, 2)):
            helpers.assert_same_results(gidx.search_raw(qs, K, w), onow.knn_search(qs, K, w), what="step %d K=%d w=%d" % (step, K, w))
        # the appended points are findable: each new point's nearest stored code is (at worst) its own
        got_ids, _, cnt = gidx.search_raw(pts[:4], 1, 24)
        assert (cnt == 1).all()
    assert inplace_seen >= 6 and relayout_seen >= 2, (inplace_seen, relayout_seen)
    # the host mirror equals lists built by the oracle's encoder in the same order
    offsets, codes, ids = gidx._lists()
    assert offsets[-1] == nid and sorted(ids.tolist()) == list(range(nid))


def test_push_into_empty_and_single_lists(native):
    """Empty lists have capacity too: a fresh index takes single push! calls in place after the first search."""
    cent, cbs, labels = helpers.make_quantizers(77, 32, 50, 8, 256)
    gidx = native.IVFADCIndex.from_arrays(cent, cbs, labels)
    rng = np.random.default_rng(77)
    pts = rng.random((120, 32), dtype=np.float32)
    assert gidx.search_raw(pts[:3], 5, 4)[2].tolist() == [0, 0, 0]       # kc empty lists
    for i in range(120):
        gidx._append(pts[i:i + 1], np.array([i], np.uint32))
        if i % 17 == 0 or i == 119:
            offsets, codes, ids = gidx._lists()
            onow = ora.OracleIndex(cent, cbs, labels, offsets, codes, ids)
            helpers.assert_same_results(gidx.search_raw(pts[:30], 7, 50), onow.knn_search(pts[:30], 7, 50), what="i=%d" % i)
    assert gidx.get_stats()["inplace_appends"] >= 100


def test_push_capacity_and_dimension_asserts(native):
    """test/utils.jl:1-29 with index_type UInt8: 256 points fit, the 257th push! asserts."""
    oidx, data = helpers.build_index(38, 243, 10, 100, 2, 16)
    gidx = gpu_index(native, oidx, index_type=np.uint8)
    rng = np.random.default_rng(38)
    for _ in range(256 - 243):
        native.push(gidx, rng.random(10, dtype=np.float32))
    assert len(gidx) == 256
    with pytest.raises(AssertionError):
        native.push(gidx, rng.random(10, dtype=np.float32))
    native.delete_from_index(gidx, [1])
    with pytest.raises(AssertionError):
        native.push(gidx, rng.random(11, dtype=np.float32))
    native.pushfirst(gidx, rng.random(10, dtype=np.float32))
    assert len(gidx) == 256
    with pytest.raises(AssertionError):
        native.pushfirst(gidx, rng.random(10, dtype=np.float32))
    v = native.pop(gidx)
    assert v.shape == (10,) and len(gidx) == 255
    v = native.popfirst(gidx)
    assert v.shape == (10,) and len(gidx) == 254


def test_synthetic_lists_match_oracle_generator(native):
    """Device-synthesised codes (counter-based RNG) == the oracle's generator, so the oracle can
    regenerate any probed list of a billion-scale synthetic index on demand."""
    for m, d in ((8, 32), (16, 32), (10, 20)):
        cent, cbs, labels = helpers.make_quantizers(40 + m, d, 50, m, 256)
        rng = np.random.default_rng(40 + m)
        sizes = rng.integers(0, 3000, 50)
        offsets = np.zeros(51, np.int64)
        np.cumsum(sizes, out=offsets[1:])
        gidx = native.IVFADCIndex.from_arrays(cent, cbs, labels)
        gidx.synth_lists(offsets, 1234)
        osyn = ora.OracleIndex(cent, cbs, labels, offsets, None, None, synth_seed=1234)
        qs = rng.random((30, d), dtype=np.float32)
        helpers.assert_same_results(gidx.search_raw(qs, 10, 4), osyn.knn_search(qs, 10, 4), what="synth m=%d" % m)
        # and equal to the materialised arrays
        omat = ora.OracleIndex(cent, cbs, labels, offsets, ora.synth_fill(1234, 0, int(offsets[-1]), m),
                               np.arange(int(offsets[-1]), dtype=np.uint32))
        helpers.assert_same_results(osyn.knn_search(qs, 10, 4), omat.knn_search(qs, 10, 4))


def test_batches_are_independent_and_repeatable(native):
    """Idempotence: repeated calls and different batch splits give identical results."""
    oidx, _ = helpers.build_index(50, 20000, 128, 128, 8, 256, mode="random")
    gidx = gpu_index(native, oidx)
    rng = np.random.default_rng(50)
    qs = rng.random((300, 128), dtype=np.float32)
    a = gidx.search_raw(qs, 10, 8)
    b = gidx.search_raw(qs, 10, 8)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    c = [gidx.search_raw(qs[s:s + 77], 10, 8) for s in range(0, 300, 77)]
    for i in range(3):
        assert np.array_equal(a[i], np.concatenate([x[i] for x in c]))
    helpers.assert_same_results(a, oidx.knn_search(qs, 10, 8))


@pytest.mark.parametrize("mode", [-1, 1, 4])
@pytest.mark.parametrize("K", [10, 70])
def test_topk_concentrated_in_one_wave(native, mode, K):
    """Regression: all K best points sit in ONE wave's block of the list (positions 256..), so the workgroup-shared
    pruning bound equals an entry that another wave must still accept when the four waves' results are merged."""
    oidx, _ = helpers.build_index(60, 4096, 32, 2, 8, 256, mode="random")
    rng = np.random.default_rng(60)
    qs = rng.random((6, 32), dtype=np.float32)
    for r in range(qs.shape[0]):
        cl, _ = oidx.coarse_search(qs[r], 1)
        lo, hi = int(oidx.offsets[cl[0]]), int(oidx.offsets[cl[0] + 1])
        assert hi - lo > 600
        _, best = oidx.encode(qs[r][None])                    # the code closest to this query's residual
        oidx.codes[lo + 256 + 7 * r: lo + 256 + 7 * r + K] = best[0]   # K copies (exact ties) inside positions 256..511
    gidx = gpu_index(native, oidx)
    gidx.set_tuning(mode, 0)
    got, exp = check(native, oidx, qs, K, 1, gidx, what="concentrated mode=%d K=%d" % (mode, K))
    check(native, oidx, qs, K, 2, gidx, what="concentrated w=2 mode=%d K=%d" % (mode, K))


@pytest.mark.parametrize("mode", [-1, 1, 2, 4])
def test_selector_limits_with_ties(native, mode):
    """K <= 64 is selected in registers, K > 64 in LDS buffers of pow2(K + 64) keys whose waves share quarter keys (publish_bound): every K
    around 64, 128 and 192 (buffer sizes 256 | 512), in every plan, on lists with many exact ties (duplicate codes: keys differ in the
    visit order only), with lists shorter than K, with w = 1 (a single list, fewer than K points for some queries) and with the K best
    concentrated in one wave's block."""
    oidx, data = helpers.build_index(62, 6000, 32, 12, 8, 256, mode="random", ndistinct=40)
    rng = np.random.default_rng(62)
    qs = np.concatenate([rng.random((37, 32), dtype=np.float32), data[:3]])
    cl, _ = oidx.coarse_search(qs[0], 1)
    lo = int(oidx.offsets[cl[0]])
    _, best = oidx.encode(qs[0][None])
    oidx.codes[lo + 256: lo + 256 + 128] = best[0]     # 128 exact ties inside one wave's block of the first query's closest list
    gidx = gpu_index(native, oidx)
    gidx.set_tuning(mode, 0)
    for K in (64, 65, 66, 96, 100, 127, 128, 129, 192, 193):
        for w in (1, 3, 12):
            got, exp = check(native, oidx, qs, K, w, gidx, what="selector limits mode=%d K=%d w=%d" % (mode, K, w))
            assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf])
    # a tiny index: fewer than K points in all
    oidx2, _ = helpers.build_index(63, 90, 16, 4, 8, 256, mode="random")
    g2 = gpu_index(native, oidx2)
    g2.set_tuning(mode, 0)
    for K in (65, 100, 128):
        check(native, oidx2, qs[:9, :16].copy(), K, 4, g2, what="selector limits, 90 points, mode=%d K=%d" % (mode, K))


@pytest.mark.parametrize("w", [1, 2, 3, 8])
def test_closest_cell_alone_first_round(native, w):
    """The query-major kernel scans the closest cell alone in its first round when the second cell lies far behind it (dc[1] > 2 dc[0]),
    and builds ONE table in any round whose second probe is pruned, deferred or past the end (odd w).  Queries on a centroid (dc[0] = 0:
    the rule fires), queries halfway between two centroids (ratio 1: it does not), random queries; pruning on and off (off: every round
    is a pair, the first included); results are the oracle's in every case and identical between the two."""
    oidx, data = helpers.build_index(64, 9000, 128, 24, 8, 256, mode="random")
    rng = np.random.default_rng(64)
    cent = oidx.centroids
    on = cent[rng.integers(0, 24, 40)] + np.float32(1e-3) * rng.standard_normal((40, 128)).astype(np.float32)
    mid = (0.5 * (cent[rng.integers(0, 24, 40)] + cent[rng.integers(0, 24, 40)])).astype(np.float32)
    qs = np.concatenate([on.astype(np.float32), mid, rng.random((40, 128), dtype=np.float32), cent[:4]])
    g = gpu_index(native, oidx)
    g.set_tuning(-1, 0)
    exp = oidx.knn_search(qs, 10, w)
    res = {}
    for prune in (1, 0):
        g.set_pruning(prune)
        res[prune] = g.search_raw(qs, 10, w)
        helpers.assert_same_results(res[prune], exp, what="first round alone: w=%d pruning=%d" % (w, prune))
        assert g.get_stats()["last_qg"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    g.set_pruning(1)
    helpers.assert_same_results(g.search_raw(qs, 100, w), oidx.knn_search(qs, 100, w), what="first round alone: K=100 w=%d" % w)


def test_sub_batching_is_invisible(native):
    """A tiny workspace limit forces the batch through many sub-batches: identical results."""
    oidx, _ = helpers.build_index(61, 6000, 32, 200, 8, 256, mode="random")
    rng = np.random.default_rng(61)
    qs = rng.random((700, 32), dtype=np.float32)
    for mode in (-1, 4):
        g = gpu_index(native, oidx)
        g.set_tuning(mode, 0)
        ref = g.search_raw(qs, 10, 6)
        g.set_workspace_limit(1 << 20)            # 1 MiB: ~64-query sub-batches
        sub = g.search_raw(qs, 10, 6)
        assert all(np.array_equal(a, b) for a, b in zip(ref, sub))
        helpers.assert_same_results(sub, oidx.knn_search(qs, 10, 6), what="sub-batched mode=%d" % mode)
    g.search_raw(qs[:0], 10, 6)                   # empty batch is a no-op


def test_full_size_sift1m_shape_properties(native):
    """BASELINE configs[1] at full size (n = 1e6, kc = 1024, m = 8, batch 1024): size-independent properties
    (ascending distances, counts, the two scan plans and every group width agree bit for bit) plus the oracle on a
    sample.  Random codes / quantizers: the properties do not need a trained index."""
    n, d, kc, m = 1_000_000, 128, 1024, 8
    rng = np.random.default_rng(62)
    cent = rng.random((kc, d), dtype=np.float32)
    cbs = ((rng.random((m, 256, d // m), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    sizes = rng.multinomial(n, np.full(kc, 1.0 / kc))
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(sizes, out=offsets[1:])
    codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
    ids = rng.permutation(n).astype(np.uint32)
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, codes, ids)
    g = gpu_index(native, oidx)
    qs = rng.random((1024, d), dtype=np.float32)
    base = None
    for mode in (-1, 1, 2, 4):
        g.set_tuning(mode, 0)
        for w in (1, 8):
            got = g.search_raw(qs, 10, w)
            assert (got[2] == 10).all()
            assert (np.diff(got[1], axis=1) >= 0).all()                       # ascending
            assert all(len(set(r.tolist())) == 10 for r in got[0][:64])        # ids distinct
            if mode == -1:
                base = base or {}
                base[w] = got
            else:
                assert all(np.array_equal(a, b) for a, b in zip(base[w], got)), "plan %d differs at w=%d" % (mode, w)
    sample = rng.choice(1024, 48, replace=False)
    helpers.assert_same_results(tuple(a[sample] for a in base[8]), oidx.knn_search(qs[sample], 10, 8), what="full-size sample")
    # a checksum of the whole result block, stable across repeated calls
    g.set_tuning(0, 0)
    h1 = hash(g.search_raw(qs, 10, 8)[0].tobytes())
    h2 = hash(g.search_raw(qs, 10, 8)[0].tobytes())
    assert h1 == h2 == hash(base[8][0].tobytes())


# ---- coarse search on the matrix cores: filter (f32 MFMA scores) + certificate + exact refine -------------------
def _coarse_case(native, oidx, qs, K, w, what, expect_mfma=True, mode_list=(-1, 4)):
    for mode in mode_list:
        g = gpu_index(native, oidx)
        g.set_tuning(mode, 0)
        g.set_coarse_mode(2)                       # filter from kc >= 128 (the automatic threshold is kc >= 2048)
        g.reset_stats()
        got = g.search_raw(qs, K, w)
        st = g.get_stats()
        assert st["coarse_mfma"] == (1 if expect_mfma else 0), (what, st)
        helpers.assert_same_results(got, oidx.knn_search(qs, K, w), what="%s mode=%d" % (what, mode))
        g.set_coarse_mode(1)                       # the exact VALU kernel gives the same bits
        ex = g.search_raw(qs, K, w)
        assert g.get_stats()["coarse_mfma"] == 0
        assert all(np.array_equal(a, b) for a, b in zip(got, ex)), what
    return st


def test_mfma_coarse_matches_exact_random(native):
    for seed, d, kc, w in ((70, 32, 300, 8), (71, 128, 1024, 32), (72, 96, 2000, 48), (73, 64, 128, 1)):
        oidx, data = helpers.build_index(seed, 20000, d, kc, 8, 256, mode="random")
        rng = np.random.default_rng(seed)
        qs = np.concatenate([rng.random((150, d), dtype=np.float32), data[:20], oidx.centroids[:10]])   # incl. exact hits
        st = _coarse_case(native, oidx, qs, 10, w, "mfma random d=%d kc=%d w=%d" % (d, kc, w))
        assert st["coarse_fallbacks"] == 0


def test_mfma_coarse_not_used_outside_its_domain(native):
    oidx, _ = helpers.build_index(74, 5000, 50, 200, 10, 256, mode="random")        # d % 4 != 0
    qs = np.random.default_rng(74).random((40, 50), dtype=np.float32)
    _coarse_case(native, oidx, qs, 5, 4, "d=50", expect_mfma=False)
    oidx, _ = helpers.build_index(75, 5000, 32, 200, 8, 256, mode="random")
    qs = np.random.default_rng(75).random((40, 32), dtype=np.float32)
    _coarse_case(native, oidx, qs, 5, 64, "w=64", expect_mfma=False)               # w > 48
    oidx, _ = helpers.build_index(76, 2000, 32, 100, 8, 256, mode="random")
    _coarse_case(native, oidx, qs, 5, 4, "kc=100", expect_mfma=False)             # kc < 128


def test_mfma_coarse_duplicate_and_near_duplicate_centroids(native):
    """More than 64 - w exact / 1-ulp ties at the boundary: the certificate must fail and the exact fallback must
    produce the oracle's order (ties -> lower cluster id)."""
    oidx, _ = helpers.build_index(77, 30000, 64, 512, 8, 256, mode="random")
    rng = np.random.default_rng(77)
    base = oidx.centroids[5].copy()
    dup = rng.choice(np.arange(6, 512), 150, replace=False)
    oidx.centroids[dup] = base                                       # 151 identical centroids
    near = rng.choice(np.setdiff1d(np.arange(6, 512), dup), 100, replace=False)
    oidx.centroids[near] = np.nextafter(base, np.float32(2.0))       # 100 one-ulp neighbours
    qs = np.concatenate([base[None] + 0.01 * rng.standard_normal((40, 64)).astype(np.float32),
                         rng.random((40, 64), dtype=np.float32)])
    st = _coarse_case(native, oidx, qs, 10, 16, "duplicates")
    assert st["coarse_fallbacks"] >= 40                               # the 40 queries next to the duplicate cluster


def test_mfma_coarse_automatic_threshold(native):
    oidx, _ = helpers.build_index(80, 30000, 32, 2048, 8, 256, mode="random")
    g = gpu_index(native, oidx)
    qs = np.random.default_rng(80).random((128, 32), dtype=np.float32)       # (more than the small-batch path takes: that one is exact)
    helpers.assert_same_results(g.search_raw(qs, 10, 8), oidx.knn_search(qs, 10, 8))
    assert g.get_stats()["coarse_mfma"] == 1                          # kc = 2048: automatic
    oidx2, _ = helpers.build_index(81, 30000, 32, 1024, 8, 256, mode="random")
    g2 = gpu_index(native, oidx2)
    g2.search_raw(qs, 10, 8)
    assert g2.get_stats()["coarse_mfma"] == 0                         # kc = 1024: VALU kernel


def test_mfma_coarse_large_offsets(native):
    """Large common offset: ||c||, ||q|| >> distances, the cancellation in ||c||^2 - 2 q.c + ||q||^2 is severe, the
    error bound grows with the norms and must still hold (or the fallback must take over)."""
    for off in (10.0, 300.0, 5000.0):
        oidx, _ = helpers.build_index(78, 20000, 32, 400, 8, 256, mode="random")
        oidx.centroids += np.float32(off)
        rng = np.random.default_rng(int(off))
        qs = (rng.random((120, 32), dtype=np.float32) + np.float32(off)).astype(np.float32)
        _coarse_case(native, oidx, qs, 10, 8, "offset %g" % off)


def test_mfma_coarse_clustered_centroids(native):
    """Centroids in tight clumps (spacing ~1e-4 relative): many near-ties around the w-th distance."""
    rng = np.random.default_rng(79)
    oidx, _ = helpers.build_index(79, 20000, 48, 1000, 8, 256, mode="random")
    centres = rng.random((20, 48), dtype=np.float32)
    oidx.centroids[:] = centres[rng.integers(0, 20, 1000)] + (1e-4 * rng.standard_normal((1000, 48))).astype(np.float32)
    qs = centres[rng.integers(0, 20, 100)] + (1e-3 * rng.standard_normal((100, 48))).astype(np.float32)
    _coarse_case(native, oidx, qs.astype(np.float32), 10, 24, "clumps")


def test_fuzz_shapes_plans_and_modes(native):
    """Randomised differential test: 60 random (shape, K, w, batch, scan plan, coarse mode, table mode) draws against the oracle
    (IVFADC_FUZZ_DRAWS / IVFADC_FUZZ_SEED widen it for soak runs; every draw also mutates the index -- a few pushes and
    a delete in place on the device -- and searches again)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("IVFADC_FUZZ_SEED", "2026")))
    ms = [1, 2, 3, 4, 5, 8, 10, 12, 16, 24, 32, 48]
    for it in range(int(os.environ.get("IVFADC_FUZZ_DRAWS", "60"))):
        m = int(rng.choice(ms))
        dsub = int(rng.choice([1, 2, 3, 4, 6, 8, 16]))
        d = m * dsub
        if d > 512:
            dsub = max(1, 512 // m)
            d = m * dsub
        kc = int(rng.choice([2, 3, 17, 64, 128, 130, 257, 600]))
        ksub = int(rng.choice([1, 2, 16, 255, 256]))
        n = int(rng.choice([0, 1, 50, 700, 5000]))
        K = int(rng.choice([1, 2, 10, 63, 64, 65, 200, 2500]))
        w = int(rng.choice([1, 2, 7, 16, 47, 48, 49, 64, 100]))
        nq = int(rng.choice([1, 3, 64, 130]))
        mode = int(rng.choice([-1, 1, 2, 4, 0, -2, -3]))      # -2: the generic dump-and-sort path, -3: query-major behind the stand-alone top-w
        cmode = int(rng.choice([0, 1, 2, 6, 8]))              # ... 6: the certified two-level search (where it can be built), 8: bf16-split filter
        tmode = int(rng.choice([0, 0, 1, 2, 3, 4]))           # ADC tables: automatic, exact f32 everywhere, matrix-core rounds wherever built (3 / 4: from the bf16 split)
        if rng.random() < 0.2:
            # the shapes the matrix-core table rounds are instantiated for, in the regime they take (register selectors, w <= 32)
            m, dsub = ((48, 16), (16, 6))[int(rng.integers(0, 2))]
            d = m * dsub
            ksub = 256
            K = int(rng.choice([1, 10, 64]))
            w = int(rng.choice([1, 2, 7, 16, 32]))
            mode = int(rng.choice([-1, -3, 0]))
            tmode = int(rng.choice([0, 2, 4]))
        build_mode = "encode" if (n and n <= 700 and rng.random() < 0.5) else "random"
        oidx, data = helpers.build_index(1000 + it, n, d, kc, m, ksub, label_perm=bool(rng.random() < 0.5), mode=build_mode,
                                         ndistinct=(3 if rng.random() < 0.2 else None))
        qs = rng.random((nq, d), dtype=np.float32)
        if n:
            qs[: min(nq, 2)] = data[: min(nq, 2)]
        g = gpu_index(native, oidx)
        g.set_tuning(mode, int(rng.choice([0, 1024])))
        g.set_coarse_mode(cmode)
        g.set_table_mode(tmode)
        what = "fuzz %d: m=%d dsub=%d kc=%d ksub=%d n=%d K=%d w=%d nq=%d plan=%d coarse=%d tables=%d %s" % (
            it, m, dsub, kc, ksub, n, K, w, nq, mode, cmode, tmode, build_mode)
        helpers.assert_same_results(g.search_raw(qs, K, w), oidx.knn_search(qs, K, w), what=what)
        if it % 3 == 0 and ksub > 1:
            # mutate in place: pushes, then a delete, then search the edited device copy
            npush = int(rng.choice([1, 5, 40]))
            pts = rng.random((npush, d), dtype=np.float32)
            g._append(pts, np.arange(n, n + npush, dtype=np.uint32))
            if n + npush > 2:
                g._delete_ids(rng.integers(0, n + npush, int(rng.choice([1, 3, 30]))).astype(np.uint32))
            helpers.assert_same_results(g.search_raw(qs, K, w), _oracle_of(g, oidx).knn_search(qs, K, w), what=what + " after edits")


def test_multi_device_front_end(native):
    """ivfadc_mg_*: replicas + contiguous query blocks, exercised on one GPU by listing device 0 three times."""
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    oidx, data = helpers.build_index(95, 8000, 32, 50, 8, 256)
    L = nat.lib()
    g = C.c_void_p()
    devs = np.array([0, 0, 0], np.int32)
    nat.check(L.ivfadc_mg_create(C.byref(g), 3, nat.ptr(devs, C.c_int32), 32, 50, 8, 256, nat.ptr(oidx.centroids, C.c_float),
                                 nat.ptr(oidx.codebooks, C.c_float), nat.ptr(oidx.labels, C.c_uint8)))
    try:
        nat.check(L.ivfadc_mg_set_lists(g, nat.ptr(oidx.offsets, C.c_int64), nat.ptr(oidx.codes, C.c_uint8),
                                        nat.ptr(oidx.ids, C.c_uint32)))
        rng = np.random.default_rng(95)
        for nq in (100, 2, 1):                                     # 34+33+33, 1+1+0, 1+0+0
            qs = rng.random((nq, 32), dtype=np.float32)
            ids = np.zeros((nq, 10), np.uint32); dists = np.zeros((nq, 10), np.float32); counts = np.zeros(nq, np.int32)
            nat.check(L.ivfadc_mg_search(g, nq, nat.ptr(qs, C.c_float), 10, 4, nat.ptr(ids, C.c_uint32),
                                         nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
            helpers.assert_same_results((ids, dists, counts), oidx.knn_search(qs, 10, 4), what="mg nq=%d" % nq)
        # append goes to every replica
        new = rng.random((5, 32), dtype=np.float32)
        nid = np.arange(8000, 8005, dtype=np.uint32)
        lst = np.zeros(5, np.int32); cod = np.zeros((5, 8), np.uint8)
        nat.check(L.ivfadc_mg_append(g, 5, nat.ptr(new, C.c_float), nat.ptr(nid, C.c_uint32), nat.ptr(lst, C.c_int32),
                                     nat.ptr(cod, C.c_uint8)))
        ol, oc = oidx.encode(new)
        assert np.array_equal(lst, ol) and np.array_equal(cod, oc)
        ids = np.zeros((5, 1), np.uint32); dists = np.zeros((5, 1), np.float32); counts = np.zeros(5, np.int32)
        nat.check(L.ivfadc_mg_search(g, 5, nat.ptr(new, C.c_float), 1, 1, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float),
                                     nat.ptr(counts, C.c_int32)))
        assert (counts == 1).all()
        with pytest.raises(AssertionError):
            nat.check(L.ivfadc_mg_search(g, 5, nat.ptr(new, C.c_float), 0, 1, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float),
                                         nat.ptr(counts, C.c_int32)))
        # delete and shift reach every replica too: whichever replica serves a query block, the answer is the same
        dele = np.array([0, 17, 8001, 8004, 123456], np.uint32)
        removed = C.c_int64(0)
        nat.check(L.ivfadc_mg_delete_ids(g, dele.shape[0], nat.ptr(dele, C.c_uint32), C.byref(removed)))
        assert removed.value == 4
        nat.check(L.ivfadc_mg_shift_ids(g, 1))
        single = gpu_index(native, oidx)
        single._append(new, nid)
        single._delete_ids(dele)
        single._shift_ids(1)
        onow = _oracle_of(single, oidx)
        qs = rng.random((90, 32), dtype=np.float32)
        ids = np.zeros((90, 10), np.uint32); dists = np.zeros((90, 10), np.float32); counts = np.zeros(90, np.int32)
        nat.check(L.ivfadc_mg_search(g, 90, nat.ptr(qs, C.c_float), 10, 4, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float),
                                     nat.ptr(counts, C.c_int32)))
        helpers.assert_same_results((ids, dists, counts), onow.knn_search(qs, 10, 4), what="mg after delete + shift")
    finally:
        L.ivfadc_mg_destroy(g)


class _RefModel:
    """Literal restatement of the reference's list maintenance (utils.jl:1-105) on Python lists."""

    def __init__(self, offsets, codes, ids):
        self.lists = [[(int(ids[p]), codes[p].copy()) for p in range(offsets[l], offsets[l + 1])] for l in range(len(offsets) - 1)]

    def n(self):
        return sum(len(l) for l in self.lists)

    def delete(self, points_1based):                       # utils.jl:90-105
        for point in sorted(set(int(p) - 1 for p in points_1based), reverse=True):
            for l in self.lists:
                pos = [i for i, (pid, _) in enumerate(l) if pid == point]
                if pos:
                    del l[pos[0]]
                    for ll in self.lists:                  # _shift_inverse_index!
                        for i, (pid, c) in enumerate(ll):
                            if pid > point:
                                ll[i] = (pid - 1, c)
                    break

    def pop(self, first):                                  # utils.jl:41-68
        vecid, shift = (0, 1) if first else (self.n() - 1, 0)
        for l in self.lists:
            pos = [i for i, (pid, _) in enumerate(l) if pid == vecid]
            if pos:
                del l[pos[0]]
        for l in self.lists:
            for i, (pid, c) in enumerate(l):
                l[i] = (pid - shift, c)

    def push(self, cluster, code, first):                  # utils.jl:127-145
        if first:
            for l in self.lists:
                for i, (pid, c) in enumerate(l):
                    l[i] = (pid + 1, c)
            self.lists[cluster].append((0, code))
        else:
            self.lists[cluster].append((self.n(), code))

    def arrays(self, m):
        offsets = np.zeros(len(self.lists) + 1, np.int64)
        np.cumsum([len(l) for l in self.lists], out=offsets[1:])
        ids = np.array([pid for l in self.lists for pid, _ in l], np.uint32)
        codes = np.array([c for l in self.lists for _, c in l], np.uint8).reshape(-1, m)
        return offsets, codes, ids


@pytest.mark.gpu
def test_delete_pop_pushfirst_in_place_on_device(native):
    """delete_from_index! / pop! / popfirst! / pushfirst! / push! between searches: the device copy is edited in place
    (ivfadc_delete_ids, ivfadc_shift_ids, ivfadc_append); lists must equal a literal restatement of utils.jl and the
    searches (which read the DEVICE copy) must equal the oracle over those lists."""
    oidx, data = helpers.build_index(610, 900, 24, 13, 8, 256)
    gidx = gpu_index(native, oidx)
    model = _RefModel(oidx.offsets, oidx.codes, oidx.ids)
    rng = np.random.default_rng(610)
    qs = rng.random((48, 24), dtype=np.float32)
    check(native, oidx, qs, 10, 5, gidx)                     # device layout current from here on

    def verify(what):
        off, codes, ids = gidx._lists()
        moff, mcodes, mids = model.arrays(8)
        assert np.array_equal(off, moff) and np.array_equal(ids, mids) and np.array_equal(codes, mcodes), what
        onow = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, off, codes, ids)
        for K, w in ((10, 5), (4, 13)):
            helpers.assert_same_results(gidx.search_raw(qs, K, w), onow.knn_search(qs, K, w), what="%s K=%d w=%d" % (what, K, w))
        assert len(gidx) == model.n()

    pts = rng.integers(1, 901, 57).tolist() + [1, 900, 900, 5000]          # duplicates and an unknown id
    native.delete_from_index(gidx, pts)
    model.delete(pts)
    verify("delete 57")
    for step in range(6):
        first = step % 2 == 0
        rec = native.popfirst(gidx) if first else native.pop(gidx)
        assert rec.shape == (24,)
        model.pop(first)
        verify("pop %d" % step)
    for step in range(5):
        p = (data[rng.integers(0, 900)] + 0.01 * rng.standard_normal(24)).astype(np.float32)
        cl, code = oidx.encode(p)
        first = step % 2 == 1
        (native.pushfirst if first else native.push)(gidx, p)
        model.push(int(cl[0]), code[0], first)
        verify("push %d" % step)
    # a whole list goes, then everything
    off, _, ids = gidx._lists()
    big = int(np.argmax(np.diff(off)))
    native.delete_from_index(gidx, (ids[off[big]:off[big + 1]].astype(np.int64) + 1).tolist())
    model.delete((ids[off[big]:off[big + 1]].astype(np.int64) + 1).tolist())
    verify("delete list %d" % big)
    native.delete_from_index(gidx, list(range(1, len(gidx) + 1)))
    assert len(gidx) == 0 and gidx.search_raw(qs[:3], 5, 13)[2].tolist() == [0, 0, 0]
    with pytest.raises(AssertionError):
        native.pop(gidx)


class _JuliaShimReplay:
    """julia/IVFADCHip.jl call by call: the same C symbols in the same order with the same arguments, through ctypes.  `model` plays the
    Julia-side lists (a literal restatement of utils.jl); the handle is what `_handles[ivfadc]` holds.  julia is not in the image, so this
    is how the shim's sequence -- hip_sync!, knn_search (both forms), push!, pushfirst!, pop!, popfirst!, delete_from_index! -- is
    exercised against the device."""

    def __init__(self, native, oidx, model, index_bits=32):
        import ctypes as C
        self.C, self.L, self.nat = C, native.load_library(), native
        self.o, self.model, self.bits = oidx, model, index_bits
        self.h = None

    def _check(self, rc):
        if rc == 0:
            return
        msg = self.L.ivfadc_last_error().decode()
        if rc == 1:
            raise AssertionError(msg)
        raise RuntimeError("ivfadc_hip status %d: %s" % (rc, msg))

    def _p(self, a, ty):
        return a.ctypes.data_as(self.C.POINTER(ty))

    def hip_sync(self):
        C, o = self.C, self.o
        if self.h is None:
            h = C.c_void_p()
            self._check(self.L.ivfadc_create(C.byref(h), 0, o.d, o.kc, o.m, o.ksub, self._p(o.centroids, C.c_float),
                                             self._p(o.codebooks, C.c_float), self._p(o.labels, C.c_uint8)))
            self.h = h
        off, codes, ids = self.model.arrays(o.m)
        self._check(self.L.ivfadc_set_lists(self.h, self._p(off, C.c_int64), self._p(codes, C.c_uint8), self._p(ids, C.c_uint32)))
        return self.h

    def hip_release(self):
        if self.h is not None:
            self.L.ivfadc_destroy(self.h)
            self.h = None

    def _handle(self):
        return self.h if self.h is not None else self.hip_sync()

    def _on_device(self, f):
        try:
            return f()
        except Exception:
            self.hip_release()
            raise

    def knn_search(self, points, k, w=1):
        C = self.C
        assert k >= 1 and w >= 1
        h = self._handle()
        q = np.ascontiguousarray(points, np.float32)
        nq = q.shape[0]
        ids = np.zeros((nq, k), np.uint32); dists = np.zeros((nq, k), np.float32); counts = np.zeros(nq, np.int32)
        self._check(self.L.ivfadc_search(h, nq, self._p(q, C.c_float), k, min(w, self.o.kc), self._p(ids, C.c_uint32),
                                         self._p(dists, C.c_float), self._p(counts, C.c_int32)))
        return ids, dists, counts

    def knn_search_batches(self, batches, k, w=1):
        C = self.C
        assert k >= 1 and w >= 1
        h = self._handle()
        sizes = np.array([len(b) for b in batches], np.int64)
        total = int(sizes.sum())
        q = np.ascontiguousarray(np.concatenate([np.asarray(b, np.float32).reshape(-1, self.o.d) for b in batches]), np.float32)
        ids = np.zeros((total, k), np.uint32); dists = np.zeros((total, k), np.float32); counts = np.zeros(total, np.int32)
        self._check(self.L.ivfadc_search_batches(h, len(sizes), self._p(sizes, C.c_int64), self._p(q, C.c_float), k, min(w, self.o.kc),
                                                 self._p(ids, C.c_uint32), self._p(dists, C.c_float), self._p(counts, C.c_int32)))
        ends = np.cumsum(sizes)
        return [(ids[e - s:e], dists[e - s:e], counts[e - s:e]) for s, e in zip(sizes.tolist(), ends.tolist())]

    def _gpu_push(self, point, first):
        C = self.C
        nvectors = self.model.n()
        assert point.shape[0] == self.o.d, "Adding to index requires %d-element vectors" % self.o.d
        assert self.bits >= np.log2(nvectors + 1), "Cannot index, exceeding index capacity"
        h = self._handle()
        vecid, shift = (0, 1) if first else (nvectors, 0)
        lst = C.c_int32(0); code = np.zeros(self.o.m, np.uint8); idv = np.array([vecid], np.uint32)
        pt = np.ascontiguousarray(point, np.float32)

        def dev():
            if shift:
                self._check(self.L.ivfadc_shift_ids(h, shift))
            self._check(self.L.ivfadc_append(h, 1, self._p(pt, C.c_float), self._p(idv, C.c_uint32), C.byref(lst), self._p(code, C.c_uint8)))
        self._on_device(dev)
        self.model.push(int(lst.value), code.copy(), first)      # l.idxs .+= 1 (pushfirst!), then push!(idxs, vecid), push!(codes, code)

    def push(self, point):
        self._gpu_push(point, False)

    def pushfirst(self, point):
        self._gpu_push(point, True)

    def _gpu_delete(self, ids):
        C = self.C
        if self.h is None:
            return
        ids = np.ascontiguousarray(ids, np.uint32)
        self._on_device(lambda: self._check(self.L.ivfadc_delete_ids(self.h, ids.shape[0], self._p(ids, C.c_uint32), None)))

    def pop(self):
        n = self.model.n()
        assert n > 0, "Cannot pop element from empty index"
        self.model.pop(False)                       # invoke(pop!, ...): the reference's own method on the Julia lists
        self._gpu_delete(np.array([n - 1], np.uint32))

    def popfirst(self):
        assert self.model.n() > 0, "Cannot pop element from empty index"
        self.model.pop(True)
        self._gpu_delete(np.array([0], np.uint32))

    def delete_from_index(self, points):
        self.model.delete(points)
        self._gpu_delete(np.unique(np.asarray(points, np.int64)) - 1)


@pytest.mark.gpu
def test_julia_shim_call_sequence_replayed(native):
    """The exact C call sequence of julia/IVFADCHip.jl, interleaving push! / delete_from_index! / pushfirst! / pop! / popfirst! with
    knn_search (single batch and the run-of-batches form): after every mutation the DEVICE copy is searched and must equal the oracle
    over the Julia-side lists -- a stale device copy would show here.  A mutation before the first search (no handle yet) and a
    release + re-sync in the middle are part of the sequence."""
    d, kc, m = 32, 17, 8
    oidx, data = helpers.build_index(811, 700, d, kc, m, 256)
    model = _RefModel(oidx.offsets, oidx.codes, oidx.ids)
    shim = _JuliaShimReplay(native, oidx, model, index_bits=16)
    rng = np.random.default_rng(811)
    qs = rng.random((40, d), dtype=np.float32)

    def verify(what):
        off, codes, ids = model.arrays(m)
        onow = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, off, codes, ids)
        helpers.assert_same_results(shim.knn_search(qs, 10, 4), onow.knn_search(qs, 10, 4), what=what)
        one = shim.knn_search(qs[:1], 3, 1)
        helpers.assert_same_results(one, onow.knn_search(qs[:1], 3, 1), what=what + " (single query)")
        return onow

    shim.delete_from_index([3, 4, 4, 700])          # before any GPU call: only the Julia lists change; the first search uploads them
    assert shim.h is None
    verify("delete before the first search")
    for step in range(12):
        op = ["push", "pushfirst", "delete", "pop", "popfirst", "push"][step % 6]
        if op in ("push", "pushfirst"):
            p = (data[rng.integers(0, 700)] + 0.01 * rng.standard_normal(d)).astype(np.float32)
            getattr(shim, op)(p)
        elif op == "delete":
            shim.delete_from_index(rng.integers(1, model.n() + 1, 9).tolist() + [10 ** 6 % 60000])
        else:
            getattr(shim, op)()
        onow = verify("%s (step %d)" % (op, step))
        if step == 5:
            shim.hip_release()                       # hip_release!: the next call re-syncs from the Julia lists
    # the run-of-batches form: one ccall, every batch equal to its own knn_search
    batches = [qs[:17], qs[17:17], qs[17:30], qs[30:]]
    got = shim.knn_search_batches(batches, 10, 4)
    for b, g in zip(batches, got):
        assert g[0].shape[0] == b.shape[0]
        if b.shape[0]:
            helpers.assert_same_results(g, onow.knn_search(b, 10, 4), what="knn_search over a run of batches")
    with pytest.raises(AssertionError):
        shim.push(np.zeros(d + 1, np.float32))       # utils.jl:133, raised before the ccall
    with pytest.raises(AssertionError):
        shim.knn_search(qs, 0, 1)
    shim.hip_release()


@pytest.mark.gpu
def test_search_batches_runs_one_launch_per_batch(native):
    """ivfadc_search_batches: a run of batches from host memory, each batch naming its successor inside the library.  Results per batch
    are ivfadc_search's (the oracle's); on a plan with the rider form every batch after the first starts from rows that rode behind its
    predecessor's scan."""
    oidx, data = helpers.build_index(821, 30000, 128, 130, 8, 256, mode="random")
    rng = np.random.default_rng(821)
    g = gpu_index(native, oidx)
    g.set_tuning(-1, 0)
    batches = [rng.random((n, 128), dtype=np.float32) for n in (100, 100, 0, 257, 100, 3)]
    got = g.search_batches_raw(batches, 10, 8)
    st = g.get_stats()
    assert st["coarse_prefetched"] == 1 and st["last_rider"] == 0, st      # the last batch found its rows standing and named no successor
    for b, r in zip(batches, got):
        if b.shape[0]:
            helpers.assert_same_results(r, oidx.knn_search(b, 10, 8), what="search_batches")
    # the same through the automatic plan (small batches take the latency path, which ignores hints) and the Python surface
    g.set_tuning(0, 0)
    res = native.knn_search_batches(g, [b for b in batches], 5, w=3)
    for b, (idl, dl) in zip(batches, res):
        assert len(idl) == b.shape[0]
        if b.shape[0]:
            ei, ed, ec = oidx.knn_search(b, 5, 3)
            for r in range(b.shape[0]):
                assert np.array_equal(idl[r], ei[r, :ec[r]]) and np.allclose(dl[r], ed[r, :ec[r]], rtol=1e-4, atol=0)
    assert g.search_batches_raw([], 10, 8) == [] and g.search_batches_raw([batches[2]], 10, 8)[0][0].shape[0] == 0


@pytest.mark.gpu
def test_views_share_the_index_and_go_stale_when_it_changes(native):
    """ivfadc_clone_view: a second handle on the same device arrays with its own stream and workspace.  Searches on the index and on
    the view -- also interleaved, both in flight -- return the oracle's results; a view cannot change anything; any change to the index
    makes it refuse to search; a fresh view sees the new state.  Device-synthesised lists and the run-of-batches call (whose odd batches
    run on an internal view) are covered too."""
    import torch
    oidx, data = helpers.build_index(4410, 20000, 128, 96, 8, 256, mode="random")
    rng = np.random.default_rng(4410)
    g = gpu_index(native, oidx)
    v = g.clone_view()
    qs = rng.random((300, 128), dtype=np.float32)
    exp = oidx.knn_search(qs, 10, 8)
    helpers.assert_same_results(g.search_raw(qs, 10, 8), exp, what="index")
    helpers.assert_same_results(v.search_raw(qs, 10, 8), exp, what="view")
    # both in flight: alternate device-pointer searches on the two streams, different query blocks and output buffers
    dev = torch.device("cuda:0")
    blocks = [torch.as_tensor(rng.random((257, 128), dtype=np.float32)).to(dev) for _ in range(6)]
    outs = [(torch.zeros(257 * 10, dtype=torch.int32, device=dev), torch.zeros(257 * 10, dtype=torch.float32, device=dev),
             torch.zeros(257, dtype=torch.int32, device=dev)) for _ in range(6)]
    torch.cuda.synchronize()
    for i, (b, o) in enumerate(zip(blocks, outs)):
        h = g if i % 2 == 0 else v
        h.search_device(257, b.data_ptr(), 10, 8, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    g.sync()
    v.sync()
    for b, o in zip(blocks, outs):
        got = (o[0].cpu().numpy().view(np.uint32).reshape(257, 10), o[1].cpu().numpy().reshape(257, 10), o[2].cpu().numpy())
        helpers.assert_same_results(got, oidx.knn_search(b.cpu().numpy(), 10, 8), what="two streams")
    # a view is read-only and keeps no mirror
    for call in (lambda: v._append(data[:3], np.arange(3, dtype=np.uint32) + 900000), lambda: v._delete_ids(np.array([1], np.uint32)),
                 lambda: v._lists(), lambda: g.clone_view().clone_view()):
        with pytest.raises(Exception, match="view"):
            call()
    # the index changes: the old view refuses, a new one sees the change
    newp = rng.random((5, 128), dtype=np.float32)
    g._append(newp, np.arange(5, dtype=np.uint32) + 500000)
    oidx = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, *g._lists())
    with pytest.raises(Exception, match="changed since this view"):
        v.search_raw(qs[:4], 10, 8)
    v2 = g.clone_view()
    qn = np.concatenate([newp, qs[:20]])
    exp2 = oidx.knn_search(qn, 10, 8)
    helpers.assert_same_results(v2.search_raw(qn, 10, 8), exp2, what="fresh view")
    helpers.assert_same_results(g.search_raw(qn, 10, 8), exp2, what="index after push")
    # run of batches: odd batches on the internal view
    batches = [rng.random((n, 128), dtype=np.float32) for n in (64, 300, 0, 129, 70, 5, 257)]
    g.reset_stats()
    for b, r in zip(batches, g.search_batches_raw(batches, 10, 8)):
        if b.shape[0]:
            helpers.assert_same_results(r, oidx.knn_search(b, 10, 8), what="pipelined run of batches")
    assert g.get_stats()["queries"] == sum(b.shape[0] for b in batches)      # the internal view's share is counted
    g._append(newp[:1], np.array([777777], np.uint32))            # stale internal view: replaced on the next run
    oidx = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, *g._lists())
    for b, r in zip(batches, g.search_batches_raw(batches, 10, 8)):
        if b.shape[0]:
            helpers.assert_same_results(r, oidx.knn_search(b, 10, 8), what="pipelined run of batches after push")
    # a mutator waits for the searches still in flight on the views: results issued before a push are those of the index before it
    v4 = g.clone_view()
    big = torch.as_tensor(rng.random((4000, 128), dtype=np.float32)).to(dev)
    o4 = (torch.zeros(4000 * 10, dtype=torch.int32, device=dev), torch.zeros(4000 * 10, dtype=torch.float32, device=dev),
          torch.zeros(4000, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    before = oidx.knn_search(big.cpu().numpy(), 10, 8)
    v4.search_device(4000, big.data_ptr(), 10, 8, o4[0].data_ptr(), o4[1].data_ptr(), o4[2].data_ptr())
    g._append(big[:64].cpu().numpy(), np.arange(64, dtype=np.uint32) + 3_000_000)     # exact hits for the first 64 queries -- afterwards
    torch.cuda.synchronize()
    helpers.assert_same_results((o4[0].cpu().numpy().view(np.uint32).reshape(4000, 10), o4[1].cpu().numpy().reshape(4000, 10), o4[2].cpu().numpy()),
                                before, what="view search in flight across a push")
    oidx = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, *g._lists())
    helpers.assert_same_results(g.search_raw(big[:64].cpu().numpy(), 10, 8), oidx.knn_search(big[:64].cpu().numpy(), 10, 8), what="after the push")
    # a view that outlives its index refuses politely
    v3 = g.clone_view()
    g.__del__()                                                   # ivfadc_destroy of the index while a view of it is alive
    with pytest.raises(Exception, match="destroyed"):
        v3.search_raw(qs[:4], 10, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["random", "ragged_last_tile", "all_equal_centroids", "sorted_centroids", "pairs_one_ulp_apart"])
def test_tiled_topw_large_batch(native, case):
    """Stand-alone top-w with one wave per query (batches >= 8192) behind the matrix-core filter.  Automatic mode (0): the
    split-bf16 kernel writes per-tile records (the four smallest keys of every (query, 64-centroid tile)) and NO score
    matrix, and select_listed enumerates every centroid under the certified bound -- whole tiles where a record's last key is
    still under it ("sorted_centroids": a query's nearest centroids are neighbours in one tile).  Mode 4: the score matrix +
    tile minima + candidate pool + certificate of round 1.  Both must equal the oracle, including a last tile that is cut
    off by kc and a row where every tile ties (more qualifying tiles than the list holds -> exact recompute)."""
    d, nq = 16, 8192
    kc = {"random": 4096, "ragged_last_tile": 2500, "all_equal_centroids": 8330, "sorted_centroids": 4096, "pairs_one_ulp_apart": 4096}[case]
    oidx, data = helpers.build_index(700 + kc, 30000, d, kc, 4, 256, mode="random")
    rng = np.random.default_rng(kc)
    if case == "all_equal_centroids":
        oidx.centroids[:] = oidx.centroids[0]
    if case == "sorted_centroids":
        t = np.arange(kc, dtype=np.float32)[:, None] / np.float32(kc)
        oidx.centroids[:] = t * np.ones((1, d), np.float32) + rng.random((kc, d), dtype=np.float32) * np.float32(1e-3)
    if case == "pairs_one_ulp_apart":
        oidx.centroids[1::2] = np.nextafter(oidx.centroids[0::2], np.float32(2.0))
    qs = np.concatenate([rng.random((nq - 64, d), dtype=np.float32), data[:32], oidx.centroids[:32]]).astype(np.float32)
    if case == "sorted_centroids":
        qs[: nq - 64] = rng.random((nq - 64, 1), dtype=np.float32) * np.ones((1, d), np.float32) + \
            rng.random((nq - 64, d), dtype=np.float32) * np.float32(1e-3)
    exp = oidx.knn_search(qs[::8], 5, 6)
    res = {}
    for mode in (0, 4):
        g = gpu_index(native, oidx)
        g.set_coarse_mode(mode)
        g.set_tuning(4, 0)                                   # list-major: the stand-alone top-w kernel
        got = g.search_raw(qs, 5, 6)
        st = g.get_stats()
        assert st["coarse_mfma"] == 1 and st["last_qg"] == 4 and st["coarse_listed"] == (1 if mode == 0 else 0), (mode, st)
        helpers.assert_same_results(tuple(a[::8] for a in got), exp, what="tiled top-w %s mode %d" % (case, mode))
        if case == "all_equal_centroids":
            assert st["coarse_fallbacks"] >= nq - 64          # every tie row: exact fallback
        res[mode] = got
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[4])), "listed mode differs from the score-matrix mode"
    if case != "all_equal_centroids":                         # wider probes, all queries vs mode 4: w = 12 still takes the records
        for w2 in (12, 24):                                   # where there are >= 4 w tiles (kc = 4096), w = 24 the score matrix
            g = gpu_index(native, oidx)
            g.set_tuning(4, 0)
            g4 = gpu_index(native, oidx)
            g4.set_coarse_mode(4)
            g4.set_tuning(4, 0)
            a, b = g.search_raw(qs, 10, w2), g4.search_raw(qs, 10, w2)
            assert g.get_stats()["coarse_listed"] == (1 if (kc + 63) // 64 >= 4 * w2 else 0)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
            helpers.assert_same_results(tuple(x[::64] for x in a), oidx.knn_search(qs[::64], 10, w2), what="tiled top-w w=%d %s" % (w2, case))


@pytest.mark.gpu
def test_generic_path_any_K_and_w(native):
    """K > 2048 or w > 2048 (the reference accepts any k, and w up to the number of clusters): every key is written
    out and sorted (generic.hip.h).  Also forced (qg = -2) at ordinary K / w as a second, independent implementation
    of the same semantics: it must agree with the oracle AND with the selection kernels."""
    rng = np.random.default_rng(88)
    # (a) forced, ordinary shapes incl. permuted labels, ksub < 256, odd m, duplicates / ties, empty lists
    for seed, n, d, kc, m, ksub, K, w, nq in ((1, 3000, 24, 50, 6, 256, 10, 7, 70), (2, 900, 10, 100, 2, 16, 3, 100, 33),
                                              (3, 2000, 15, 9, 5, 200, 64, 4, 19), (4, 40, 8, 30, 8, 256, 5, 30, 11)):
        oidx, data = helpers.build_index(880 + seed, n, d, kc, m, ksub, label_perm=(seed % 2 == 0),
                                         ndistinct=(3 if seed == 3 else None))
        qs = np.concatenate([rng.random((nq - 2, d), dtype=np.float32), data[:2]])
        exp = oidx.knn_search(qs, K, w)
        fast = gpu_index(native, oidx)
        gen = gpu_index(native, oidx)
        gen.set_tuning(-2, 0)
        got = gen.search_raw(qs, K, w)
        assert gen.get_stats()["last_qg"] == -2
        helpers.assert_same_results(got, exp, what="generic seed %d" % seed)
        helpers.assert_same_results(got, fast.search_raw(qs, K, w), what="generic vs selection kernels, seed %d" % seed)
    # (b) beyond the selection kernels
    oidx, data = helpers.build_index(889, 30000, 16, 3000, 4, 256, mode="random")
    qs = np.concatenate([rng.random((20, 16), dtype=np.float32), data[:4]])
    g = gpu_index(native, oidx)
    for K, w in ((5000, 40), (10, 2500), (3000, 3000), (40000, 3000)):
        got = g.search_raw(qs, K, w)
        assert g.get_stats()["last_qg"] == -2
        helpers.assert_same_results(got, oidx.knn_search(qs, K, w), what="K=%d w=%d" % (K, w))
    assert (got[2] == 30000).all()            # w == kc, K > n: every stored point comes back, in order
    # sub-batching of both stages under a small workspace budget
    g.set_workspace_limit(2 << 20)
    helpers.assert_same_results(g.search_raw(qs, 2100, 100), oidx.knn_search(qs, 2100, 100), what="tiny workspace")


@pytest.mark.gpu
def test_mg_rccl_gather_and_synth_lists(native):
    """ivfadc_mg_set_gather(g, 1): the final merge of a batch is ONE ncclAllGather of the packed per-device blocks, issued
    by the library itself (RCCL bound with dlopen, ncclCommInitAll, one stream per device).  Every device visible to the
    test takes part (one on the single-GPU box: the collective degenerates to a copy but runs the same code; the
    driver's multi-GPU node runs it over xGMI).  Lists come from ivfadc_mg_synth_lists (every replica synthesises its own
    copy), results are checked against the oracle's replay of the same generator."""
    import ctypes as C
    import torch
    from ivfadc_jl_amd import _native as nat
    ndev = max(1, min(8, torch.cuda.device_count()))
    d, kc, m = 32, 64, 8
    cent, cbs, labels = helpers.make_quantizers(96, d, kc, m, 256)
    rng = np.random.default_rng(96)
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(rng.integers(0, 2000, kc), out=offsets[1:])
    osyn = ora.OracleIndex(cent, cbs, labels, offsets, None, None, synth_seed=4242)
    L = nat.lib()
    g = C.c_void_p()
    devs = np.arange(ndev, dtype=np.int32)
    nat.check(L.ivfadc_mg_create(C.byref(g), ndev, nat.ptr(devs, C.c_int32), d, kc, m, 256, nat.ptr(cent, C.c_float),
                                 nat.ptr(cbs, C.c_float), nat.ptr(labels, C.c_uint8)))
    try:
        assert L.ivfadc_mg_num_devices(g) == ndev
        nat.check(L.ivfadc_mg_synth_lists(g, nat.ptr(offsets, C.c_int64), C.c_uint64(4242)))
        for mode in (0, 1):
            nat.check(L.ivfadc_mg_set_gather(g, mode))
            for nq in (257, 3, 1):
                qs = rng.random((nq, d), dtype=np.float32)
                ids = np.zeros((nq, 10), np.uint32); dists = np.zeros((nq, 10), np.float32); counts = np.zeros(nq, np.int32)
                nat.check(L.ivfadc_mg_search(g, nq, nat.ptr(qs, C.c_float), 10, 5, nat.ptr(ids, C.c_uint32),
                                             nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
                helpers.assert_same_results((ids, dists, counts), osyn.knn_search(qs, 10, 5), what="mg gather=%d nq=%d" % (mode, nq))
        ncoll = C.c_int64(0)
        nat.check(L.ivfadc_mg_collectives(g, C.byref(ncoll)))
        assert ncoll.value == 3                                   # one collective per batch, RCCL mode only
    finally:
        L.ivfadc_mg_destroy(g)
    # duplicate devices cannot form a communicator: refused with a message, host gather still works
    g = C.c_void_p()
    devs = np.array([0, 0], np.int32)
    nat.check(L.ivfadc_mg_create(C.byref(g), 2, nat.ptr(devs, C.c_int32), d, kc, m, 256, nat.ptr(cent, C.c_float),
                                 nat.ptr(cbs, C.c_float), nat.ptr(labels, C.c_uint8)))
    try:
        assert L.ivfadc_mg_set_gather(g, 1) == 2
    finally:
        L.ivfadc_mg_destroy(g)


@pytest.mark.gpu
def test_large_K_with_wide_codes_routes_to_generic_path(native):
    """m = 48 with K in 1985..2048 needs 4 * 4096 * 8 B of selector buffers next to 48 KB of tables -- more
    than a CU's 160 KB of LDS.  The library must answer through the dump-and-sort path, not fail; also w in 961..2048
    (top-w selector buffers above the default dynamic-LDS limit)."""
    oidx, _ = helpers.build_index(97, 6000, 96, 12, 48, 256, mode="random")
    g = gpu_index(native, oidx)
    qs = np.random.default_rng(97).random((6, 96), dtype=np.float32)
    for K in (1984, 1985, 2048):
        for mode in (0, -1, 4):
            g.set_tuning(mode, 0)
            check(native, oidx, qs, K, 4, g, what="m=48 K=%d mode=%d" % (K, mode))
    oidx2, _ = helpers.build_index(98, 30000, 16, 1600, 8, 256, mode="random")
    g2 = gpu_index(native, oidx2)
    qs2 = np.random.default_rng(98).random((5, 16), dtype=np.float32)
    for w in (960, 961, 1500, 1600):
        for mode in (-1, 4):
            g2.set_tuning(mode, 0)
            check(native, oidx2, qs2, 20, w, g2, what="w=%d mode=%d" % (w, mode))


@pytest.mark.gpu
def test_two_handles_share_kernels_with_different_lds(native):
    """hipFuncAttributeMaxDynamicSharedMemorySize is per (function, device).  Two live handles on one device that
    share a kernel instantiation with different LDS sizes (different K through the LDS selectors) must both keep working
    when their searches interleave."""
    oa, _ = helpers.build_index(99, 5000, 64, 20, 8, 256, mode="random")
    ga, gb = gpu_index(native, oa), gpu_index(native, oa)
    qs = np.random.default_rng(99).random((12, 64), dtype=np.float32)
    for mode in (-1, 4):
        ga.set_tuning(mode, 0)
        gb.set_tuning(mode, 0)
        for _ in range(2):
            check(native, oa, qs, 2000, 6, ga, what="handle A K=2000")     # large LDS request
            check(native, oa, qs, 100, 6, gb, what="handle B K=100")       # smaller request on the same kernel
            check(native, oa, qs, 2000, 6, ga, what="handle A again")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["random", "offset_300", "near_duplicates", "d_not_multiple_of_32"])
def test_bf16_split_coarse_filter(native, case):
    """Large coarse problems score the centroids with three bf16 MFMAs per product (split operands, x = hi + lo) instead
    of one f32 MFMA.  Scores only rank; the certificate + exact refine must still return the oracle's probes -- also when
    a common offset makes the scores cancel (the certificate fails and the exact fallback takes over), when centroids
    are one ulp apart, and when d is padded to the 32-wide k-step.  Modes: 0 = automatic (bf16 here), 3 = f32 MFMA
    filter, 1 = exact VALU kernel: all identical, bit for bit, and equal to the oracle."""
    d = 40 if case == "d_not_multiple_of_32" else 64
    kc, nq, m = 2048, 4096, 8
    oidx, data = helpers.build_index(900 + len(case), 40000, d, kc, m, 256, mode="random")
    rng = np.random.default_rng(len(case))
    if case == "offset_300":
        oidx.centroids += np.float32(300.0)
    if case == "near_duplicates":
        oidx.centroids[1::2] = oidx.centroids[0::2]
        oidx.centroids[1::4] = np.nextafter(oidx.centroids[1::4], np.float32(2.0))       # one ulp apart
    qs = rng.random((nq, d), dtype=np.float32)
    if case == "offset_300":
        qs += np.float32(300.0)
    qs[:16] = oidx.centroids[:16]
    res = {}
    for mode in (0, 8, 3, 1):                      # 0: the one-product f16 form (round 5), 8: the three-product bf16 split
        g = gpu_index(native, oidx)
        g.set_coarse_mode(mode)
        res[mode] = g.search_raw(qs, 10, 16)
        st = g.get_stats()
        assert st["coarse_mfma"] == (0 if mode == 1 else 1)
        assert st["coarse_f16"] == (1 if mode == 0 else 0)
        if mode in (0, 8) and case == "offset_300":
            assert st["coarse_fallbacks"] > 0          # the bound cannot separate the candidates: exact recompute
    for mode in (0, 8, 3):
        assert all(np.array_equal(a, b) for a, b in zip(res[mode], res[1])), "coarse mode %d differs from the exact kernel" % mode
    pick = np.sort(rng.choice(nq, 96, replace=False))
    helpers.assert_same_results(tuple(a[pick] for a in res[0]), oidx.knn_search(qs[pick], 10, 16), what="bf16 coarse " + case)


@pytest.mark.gpu
def test_in_library_allgather_single_rank(native):
    """ivfadc_comm_*: the one-process-per-GPU merge inside the library.  One rank here (the communicator of a single
    process: the collective degenerates to a copy but takes the same path -- search on the handle's stream, ncclAllGather on
    its side stream, per-slot completion events); the gathered block must equal the local block and the oracle, over more
    batches than there are slots (slot reuse waits for the previous collective on the device).  The calls also carry the next-batch
    hint (ivfadc_set_next_queries): the collective entry takes it like the plain one."""
    import torch
    oidx, _ = helpers.build_index(120, 20000, 32, 64, 8, 256, mode="random")
    g = gpu_index(native, oidx)
    g.comm_init(1, 0, native.comm_unique_id())
    nq, K, w = 300, 10, 5
    width = 2 * K + 1
    rng = np.random.default_rng(120)
    blocks = [torch.zeros(nq * width, dtype=torch.int32, device="cuda") for _ in range(3)]
    gath = [torch.zeros(nq * width, dtype=torch.int32, device="cuda") for _ in range(3)]
    qsets = [rng.random((nq, 32), dtype=np.float32) for _ in range(7)]
    qdev = [torch.as_tensor(q).cuda() for q in qsets]
    torch.cuda.synchronize()
    got = []
    for i, qd in enumerate(qdev):
        s = i % 3
        if i >= 3:                                     # the slot is about to be overwritten: read batch i - 3 first
            g.comm_wait(); g.sync(); torch.cuda.synchronize()
            got.append((i - 3, gath[s].cpu().numpy().copy(), blocks[s].cpu().numpy().copy()))
        if i + 1 < len(qdev) and i != 3:               # the next batch is known (not after batch 3: that search computes its own rows)
            g.set_next_queries(nq, qdev[i + 1].data_ptr(), 1000 + i + 1)
        g.set_query_token(1000 + i)
        g.search_device_allgather(nq, qd.data_ptr(), K, w, blocks[s].data_ptr(), gath[s].data_ptr(), s)
    assert g.comm_wait() == 7
    g.sync(); torch.cuda.synchronize()
    for i in range(4, 7):
        got.append((i, gath[i % 3].cpu().numpy().copy(), blocks[i % 3].cpu().numpy().copy()))
    assert len(got) == 7
    for i, ga, bl in got:
        assert np.array_equal(ga, bl), "gathered block differs from the local block (batch %d)" % i
        ids = ga[:nq * K].view(np.uint32).reshape(nq, K)
        dists = ga[nq * K:2 * nq * K].view(np.float32).reshape(nq, K)
        counts = ga[2 * nq * K:]
        helpers.assert_same_results((ids, dists, counts), oidx.knn_search(qsets[i], K, w), what="allgather batch %d" % i)
    # the query-major plan carries riders: hinted batches through the collective entry
    g.set_tuning(-1, 0)
    for i in range(3):
        g.set_next_queries(nq, qdev[i + 1].data_ptr(), 2000 + i + 1)
        g.set_query_token(2000 + i)
        g.search_device_allgather(nq, qdev[i].data_ptr(), K, w, blocks[i].data_ptr(), gath[i].data_ptr(), i)
        st = g.get_stats()                             # (synchronises the search stream)
        assert st["last_rider"] == 1 and st["coarse_prefetched"] == (1 if i else 0), st
    g.comm_wait(); g.sync(); torch.cuda.synchronize()
    for i in range(3):
        ga = gath[i].cpu().numpy()
        assert np.array_equal(ga, blocks[i].cpu().numpy())
        helpers.assert_same_results((ga[:nq * K].view(np.uint32).reshape(nq, K), ga[nq * K:2 * nq * K].view(np.float32).reshape(nq, K),
                                     ga[2 * nq * K:]), oidx.knn_search(qsets[i], K, w), what="hinted allgather batch %d" % i)
    g.set_tuning(0, 0)
    # two batches in flight per rank: searches alternate between the index and a view of it, every collective on the index's communicator
    v = g.clone_view()
    blocks2 = [torch.zeros(nq * width, dtype=torch.int32, device="cuda") for _ in range(6)]
    gath2 = [torch.zeros(nq * width, dtype=torch.int32, device="cuda") for _ in range(6)]
    base = g.comm_wait()
    for i in range(6):
        g.search_device_allgather_on(g if i % 2 == 0 else v, nq, qdev[i].data_ptr(), K, w, blocks2[i].data_ptr(), gath2[i].data_ptr(), i)
    assert g.comm_wait() == base + 6
    g.sync(); v.sync(); torch.cuda.synchronize()
    for i in range(6):
        ga = gath2[i].cpu().numpy()
        assert np.array_equal(ga, blocks2[i].cpu().numpy())
        helpers.assert_same_results((ga[:nq * K].view(np.uint32).reshape(nq, K), ga[nq * K:2 * nq * K].view(np.float32).reshape(nq, K),
                                     ga[2 * nq * K:]), oidx.knn_search(qsets[i], K, w), what="two-lane allgather batch %d" % i)
    with pytest.raises(native.IVFADCError):           # a view of ANOTHER index cannot search for this communicator
        g.search_device_allgather_on(gpu_index(native, oidx).clone_view(), nq, qdev[0].data_ptr(), K, w, blocks[0].data_ptr(), gath[0].data_ptr(), 0)
    with pytest.raises(native.IVFADCError):
        g.search_device_allgather(nq, qdev[0].data_ptr(), K, w, blocks[0].data_ptr(), gath[0].data_ptr(), 99)
    g2 = gpu_index(native, oidx)
    with pytest.raises(native.IVFADCError):           # no communicator yet
        g2.search_device_allgather(nq, qdev[0].data_ptr(), K, w, blocks[0].data_ptr(), gath[0].data_ptr(), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("m,d", [(8, 128), (16, 96)])
def test_striped_filter_and_reference_order_kernels_agree(native, m, d):
    """The list-major scan with four queries per code stream has two forms: bank-striped tables with rotated-order sums as
    a filter (survivors recomputed in the reference's order) and the round-1 kernel that keeps the reference's
    order in every lane (ivfadc_set_table_mode(h, 1)).  Same ids, same distance bits, both equal to the oracle -- on random
    codes, on a list made of a handful of distinct codes (exact ties in every step: every copy of the best code passes
    the filter together) and with K above the register selectors' reach."""
    for ndistinct in (None, 5):
        oidx, _ = helpers.build_index(130 + m, 60000, d, 24, m, 256, mode="random", ndistinct=ndistinct)
        rng = np.random.default_rng(m)
        qs = rng.random((96, d), dtype=np.float32)
        for K in (10, 100):
            res = {}
            for mode in (0, 1):
                g = gpu_index(native, oidx)
                g.set_tuning(4, 4096)                       # list-major, 4 queries per stream, several chunks per list
                g.set_table_mode(mode)
                res[mode] = g.search_raw(qs, K, 6)
                assert g.get_stats()["last_striped"] == (1 if mode == 0 else 0)
            assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1])), "striped vs reference-order kernel (ndistinct=%s, K=%d)" % (ndistinct, K)
            helpers.assert_same_results(res[0], oidx.knn_search(qs, K, 6), what="striped m=%d ndistinct=%s K=%d" % (m, ndistinct, K))
            if m == 8 and K <= 64:      # the narrow-field kernel: eight queries per code stream (K beyond the register selectors falls back to 4)
                g = gpu_index(native, oidx)
                g.set_tuning(8, 4096)
                r8 = g.search_raw(qs, K, 6)
                assert g.get_stats()["last_nf"] == 1
                assert all(np.array_equal(a, b) for a, b in zip(r8, res[1])), "narrow-field vs reference-order kernel (ndistinct=%s, K=%d)" % (ndistinct, K)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 8, 40, 256, 60000), (96, 16, 31, 256, 30000), (50, 10, 17, 64, 8000)])
def test_list_partitioned_mode_partial_keys_and_merge(native, shape):
    """ivfadc_set_list_partition / ivfadc_search_device_partial / ivfadc_merge_partials_device: the strong-scaling mode of the multi-GPU
    path rehearsed on ONE GPU -- the handle plays every rank in turn (ranks hold identical replicas and run the identical coarse search),
    the partial keys of all "ranks" are stacked as the all-gather would leave them, and the merge must give the oracle's full search:
    ids exact, distance bits exact.  Each rank's keys must also equal a numpy restatement of the partial scan (global visit orders).
    nparts = 2, 3, 8 (more parts than some queries have probes: empty partials), K in and beyond the register selectors, ties across
    ranks (few distinct codes), group widths 1 / 4 / 8."""
    import torch
    d, m, kc, ksub, n = shape
    oidx, _ = helpers.build_index(900 + d, n, d, kc, m, ksub, mode="random", ndistinct=(30 if d == 96 else None), label_perm=(d == 50))
    rng = np.random.default_rng(d)
    nq = 75
    qs = rng.random((nq, d), dtype=np.float32)
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(qs).to(dev)
    g = gpu_index(native, oidx)
    for nparts, K, w, qg in ((2, 10, 6, 0), (3, 100, 5, 4), (8, 10, 3, 1), (8, 10, min(kc, 16), 8 if (m == 8 and d == 128) else 2)):
        exp = oidx.knn_search(qs, K, w)
        g.set_tuning(qg, 0)
        keys_all = torch.zeros((nparts, nq, K), dtype=torch.int64, device=dev)
        cnts_all = torch.zeros((nparts, nq), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
        for part in range(nparts):
            g.set_list_partition(nparts, part)
            g.search_device_partial(nq, qd.data_ptr(), K, w, keys_all[part].data_ptr(), cnts_all[part].data_ptr())
            torch.cuda.synchronize()
            if K <= 10 and n <= 30000:
                rk, rc, _ = helpers.numpy_partial_keys(oidx, qs[:12], K, w, nparts, part)
                gk = keys_all[part].cpu().numpy().view(np.uint64)[:12]
                gc = cnts_all[part].cpu().numpy()[:12]
                assert np.array_equal(gc, rc) and all(np.array_equal(gk[r, :rc[r]], rk[r, :rc[r]]) for r in range(12)), \
                    "partial keys of part %d / %d" % (part, nparts)
        ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
        dist = torch.zeros(nq * K, dtype=torch.float32, device=dev)
        cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
        g.merge_partials_device(nq, K, nparts, keys_all.data_ptr(), cnts_all.data_ptr(), ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        got = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dist.cpu().numpy().reshape(nq, K), cnt.cpu().numpy())
        helpers.assert_same_results(got, exp, what="list-partitioned nparts=%d K=%d w=%d qg=%d" % (nparts, K, w, qg))
        assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf])
    # switched off again: an ordinary search
    g.set_list_partition(1, 0)
    g.set_tuning(0, 0)
    check(native, oidx, qs, 10, 4, g, what="partition off")
    # a merge that does not belong to the handle's last partial search is refused
    with pytest.raises(native.IVFADCError):
        g.merge_partials_device(nq, 10, 2, keys_all.data_ptr(), cnts_all.data_ptr(), ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["random", "permuted_labels", "few_codes", "one_list", "exact_hits", "far_queries", "clustered"])
def test_narrow_field_list_major_kernel(native, case):
    """nfscan.hip.h: list-major scan with eight queries per code stream, 4-bit table fields (one byte per query and entry, bias-started
    accumulators: a candidate is a field below 128), four private table copies (conflict-free gathers) and NO resident f32 tables --
    whatever passes is recomputed in the reference's order from the f32 codebook.  Ids and distance bits must be those of the oracle
    and of the reference-order kernel: groups that are full, partial (a list probed by 1 .. 7 queries) and several per list, one and
    several chunks per list, K = 1 / 10 / 64, permuted labels, lists of a handful of distinct codes (exact ties), queries that hit
    codewords exactly (entries of 0), queries far from every centroid (entries nearly constant), and a trained-like clustered set
    where bounds tighten early and whole work items are pruned."""
    d, m = 128, 8
    kc = 1 if case == "one_list" else 14
    n = 30000
    oidx, data = helpers.build_index(1500 + len(case), n, d, kc, m, 256, label_perm=(case == "permuted_labels"),
                                     mode="encode" if case == "clustered" else "random", ndistinct=(4 if case == "few_codes" else None))
    rng = np.random.default_rng(77 + len(case))
    qs = rng.random((61, d), dtype=np.float32)
    if case == "exact_hits":
        for i in range(16):
            cl = i % kc
            code = rng.integers(0, 256, m)
            qs[i] = oidx.centroids[cl] + np.concatenate([oidx.codebooks[ii, code[ii]] for ii in range(m)])
    elif case == "far_queries":
        qs += np.float32(50.0)
    elif case == "clustered":
        qs = (data[rng.integers(0, n, 61)] + 0.01 * rng.standard_normal((61, d))).astype(np.float32)
    gref = gpu_index(native, oidx)
    gref.set_tuning(4, 0)
    gref.set_table_mode(1)                                  # reference-order f32 tables in every lane
    for K, w, chunk in ((10, 3, 0), (1, 1, 1024), (64, min(kc, 5), 4096), (10, kc, 0)):
        exp = oidx.knn_search(qs, K, w)
        g = gpu_index(native, oidx)
        g.set_tuning(8, chunk)
        g.reset_stats()
        got = g.search_raw(qs, K, w)
        st = g.get_stats()
        assert st["last_nf"] == 1 and st["last_qg"] == 8 and st["last_scan_lds"] <= 80 * 1024, st
        helpers.assert_same_results(got, exp, what="narrow-field %s K=%d w=%d chunk=%d" % (case, K, w, chunk))
        ref = gref.search_raw(qs, K, w)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), "narrow-field vs reference-order kernel: %s K=%d w=%d" % (case, K, w)
        # a second search on the same handle (re-armed queues, bounds and counters), fewer queries (partial groups everywhere)
        got2 = g.search_raw(qs[:9], K, w)
        helpers.assert_same_results(got2, tuple(a[:9] for a in exp), what="narrow-field %s, second call" % case)
    # pruning on / off agree
    g = gpu_index(native, oidx)
    g.set_tuning(8, 0)
    g.set_pruning(0)
    helpers.assert_same_results(g.search_raw(qs, 10, min(kc, 6)), oidx.knn_search(qs, 10, min(kc, 6)), what="narrow-field %s, pruning off" % case)


@pytest.mark.parametrize("d,m,kc,nq", [(8, 8, 1, 1), (96, 16, 65, 17), (136, 8, 130, 70), (264, 8, 64, 33), (768, 48, 37, 16),
                                       (128, 8, 1024, 1024)])
def test_small_problem_exact_coarse_kernel(native, d, m, kc, nq):
    """coarse_sgpr_kernel (centroid per lane, queries in SGPRs; taken by small exact coarse searches with d % 8 == 0):
    w = kc probes every cell, so each centroid's distance seeds some returned sum and orders the probes -- one d-chunk,
    a partial last chunk (136, 264), six chunks (768), partial centroid and query tiles, and the benchmark shape."""
    n = 3000 if kc < 1024 else 20000
    oidx, _ = helpers.build_index(300 + d, n, d, kc, m, 256, mode="random")
    rng = np.random.default_rng(d + kc)
    qs = rng.random((nq, d), dtype=np.float32)
    g = gpu_index(native, oidx)
    g.set_coarse_mode(1)                           # exact VALU kernels only
    w = kc if kc < 1024 else 8
    got, exp = check(native, oidx, qs if kc < 1024 else qs[:64], 50, w, g, what="sgpr coarse d=%d kc=%d" % (d, kc))
    assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf])
    if kc == 1024:                                 # the whole batch in one launch (grid of 16 x 64 tiles), sampled check
        got = g.search_raw(qs, 10, 8)
        exp = oidx.knn_search(qs[::16], 10, 8)
        helpers.assert_same_results((got[0][::16], got[1][::16], got[2][::16]), exp, what="sgpr coarse full batch")


@pytest.mark.parametrize("m,d", [(8, 128), (16, 96), (48, 768), (10, 50)])
def test_probe_pruning_is_exact(native, m, d):
    """Query-major scan: a probe whose coarse distance lies above the K-th best key found so far ends the query (every ADC
    sum starts from its list's coarse distance and only grows).  Well separated cells make that fire on most probes; results
    must be the oracle's with pruning on and off, for K below and above the register-selector limit, ties included."""
    kc, n = 40, 6000
    rng = np.random.default_rng(m * d)
    cent, cbs, labels = helpers.make_quantizers(m + d, d, kc, m, 256, scale=0.05)
    cent = (cent * np.float32(4.0)).astype(np.float32)
    assign = rng.integers(0, kc, n)
    data = (cent[assign] + (rng.random((n, d), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)).astype(np.float32)
    tmp = ora.OracleIndex(cent, cbs, labels, np.zeros(kc + 1, np.int64), np.zeros((0, m), np.uint8), np.zeros(0, np.uint32))
    lst, codes = tmp.encode(data)
    order = np.argsort(lst, kind="stable")
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(np.bincount(lst, minlength=kc), out=offsets[1:])
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, np.ascontiguousarray(codes[order]), order.astype(np.uint32))
    qs = np.concatenate([data[:40] + np.float32(0.01), rng.random((8, d), dtype=np.float32) * np.float32(4.0)]).astype(np.float32)
    for K, w in ((10, 8), (3, 2), (100, 8), (10, 40)):
        exp = oidx.knn_search(qs, K, w)
        for plan in (-1, 4, 1):                              # query-major; list-major with 4 / 1 queries per code stream (item-level pruning)
            res = {}
            for on in (1, 0):
                g = gpu_index(native, oidx)
                g.set_pruning(on)
                g.set_tuning(plan, 1024 if plan > 0 else 0)
                g.reset_stats()
                res[on] = g.search_raw(qs, K, w)
                st = g.get_stats()
                helpers.assert_same_results(res[on], exp, what="pruning=%d plan=%d m=%d K=%d w=%d" % (on, plan, m, K, w))
                if on == 0:
                    assert st["pruned_points"] == 0
                elif plan == -1 and w >= 8 and K == 10:
                    assert 0 < st["pruned_points"] < st["scanned_points"], st
            assert all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))


@pytest.mark.parametrize("case", ["outlier_codewords", "zero_codebooks", "tiny_scale", "huge_scale", "dc_dominates_300", "dc_dominates_5000",
                                  "dc_zero_huge_entries"])
def test_integer_filter_extremes(native, case):
    """m = 8 list-major scan, four queries per code stream: the candidate filter runs on 16-bit integer tables scaled by each
    query's largest table entry (quantize_tables_m8).  Whatever the scale does -- one far codeword per sub-quantizer that
    flattens every other entry to 0, tables that are all zero, entries in the denormal range (the scale overflows), entries near
    the top of the float range -- the filter may only let MORE points through; ids and distances stay those of the oracle and of
    the reference-order kernel (table mode 1).  dc_dominates_*: a common offset of the centroids (300, 5000) with near-degenerate
    codebooks (scale 1e-3): every sum is dominated by the coarse distance, the tables' maxima are far below it and thr * inv is large
    (the regime in which the target's float evaluation leans on its 2^-18 slack); dc_zero_huge_entries: queries ON centroids
    (dc = +0) with codebooks scaled by 1e3."""
    d, m, kc = 128, 8, 12
    oidx, _ = helpers.build_index(1400 + len(case), 40000, d, kc, m, 256, mode="random")
    rng = np.random.default_rng(len(case))
    if case == "outlier_codewords":
        oidx.codebooks[:, 7, :] *= np.float32(1000.0)
    elif case == "zero_codebooks":
        oidx.codebooks[:] = 0
    elif case == "tiny_scale":
        oidx.codebooks *= np.float32(1e-21)
        oidx.centroids *= np.float32(1e-21)
    elif case == "huge_scale":
        oidx.codebooks *= np.float32(1e15)
        oidx.centroids *= np.float32(1e15)
    qs = rng.random((64, d), dtype=np.float32)
    if case.startswith("dc_dominates"):
        oidx.centroids += np.float32(300.0 if case.endswith("300") else 5000.0)
        oidx.codebooks *= np.float32(1e-3)
        qs[32:] += np.float32(300.0 if case.endswith("300") else 5000.0)      # half of the queries near the centroids, half far away
    elif case == "dc_zero_huge_entries":
        oidx.codebooks *= np.float32(1e3)
        qs[:12] = oidx.centroids[:12]
    if case == "tiny_scale":
        qs *= np.float32(1e-21)
    elif case == "huge_scale":
        qs *= np.float32(1e15)
    elif case == "zero_codebooks":
        qs[:8] = oidx.centroids[:8]                      # every sum equals dc = 0: ties across whole lists
    exp = oidx.knn_search(qs, 10, 4)
    res = {}
    for mode in (0, 1):
        g = gpu_index(native, oidx)
        g.set_tuning(4, 8192)
        g.set_table_mode(mode)
        res[mode] = g.search_raw(qs, 10, 4)
        assert g.get_stats()["last_striped"] == (1 if mode == 0 else 0)
        helpers.assert_same_results(res[mode], exp, what="integer filter %s mode %d" % (case, mode))
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    # the narrow-field kernel (eight queries per code stream, 4-bit fields, no resident f32 tables) under the same extremes
    g = gpu_index(native, oidx)
    g.set_tuning(8, 8192)
    r8 = g.search_raw(qs, 10, 4)
    st = g.get_stats()
    assert st["last_nf"] == 1 and st["last_qg"] == 8, st
    helpers.assert_same_results(r8, exp, what="narrow-field filter %s" % case)
    assert all(np.array_equal(a, b) for a, b in zip(r8, res[1]))


def _lb_index(seed, n, kc, case, label_perm=False, ndistinct=None, d=768, m=48):
    oidx, data = helpers.build_index(seed, n, d, kc, m, 256, label_perm=label_perm, mode="random", ndistinct=ndistinct)
    rng = np.random.default_rng(seed)
    qs = rng.random((40, d), dtype=np.float32)
    if case == "outlier_codewords":            # one far codeword per sub-quantizer: the a-priori range flattens every other entry
        oidx.codebooks[:, 7, :] *= np.float32(1000.0)
    elif case == "zero_codebooks":             # every entry equals ||r_ii||^2: all sums of a list tie
        oidx.codebooks[:] = 0
        qs[:8] = oidx.centroids[:8]
    elif case == "tiny_scale":
        oidx.codebooks *= np.float32(1e-21)
        oidx.centroids *= np.float32(1e-21)
        qs *= np.float32(1e-21)
    elif case == "huge_scale":
        oidx.codebooks *= np.float32(1e15)
        oidx.centroids *= np.float32(1e15)
        qs *= np.float32(1e15)
    elif case == "offset":                     # common offset: residuals are differences of large numbers
        oidx.centroids += np.float32(300.0)
        qs += np.float32(300.0)
    elif case == "far_queries":                # ||r|| >> ||codeword||: base > 0, entries nearly constant per sub-quantizer
        qs += np.float32(50.0)
    elif case == "tiny_codebooks":             # ||codeword|| << ||r||
        oidx.codebooks *= np.float32(1e-3)
    elif case == "big_codebooks":              # ||codeword|| >> ||r||
        oidx.codebooks *= np.float32(100.0)
    elif case == "exact_hits":                 # residual == a codeword in every sub-space: entries of 0 next to large ones
        for i in range(8):
            cl = i % kc
            code = rng.integers(0, 256, m)
            cw = np.concatenate([oidx.codebooks[ii, code[ii]] for ii in range(m)])
            qs[i] = oidx.centroids[cl] + cw
    return oidx, qs


@pytest.mark.parametrize("d,m", [(768, 48), (96, 16)])
@pytest.mark.parametrize("case", ["random", "ties", "labels", "outlier_codewords", "zero_codebooks", "tiny_scale", "huge_scale", "offset",
                                  "far_queries", "tiny_codebooks", "big_codebooks", "exact_hits"])
def test_matrix_core_lower_bound_tables(native, case, d, m):
    """m = 48 (dsub = 16) and m = 16 (dsub = 6: sub-space rows padded to the matrix instruction's k-step) query-major rounds with 8-bit LOWER-BOUND tables from the matrix cores (lbscan.hip.h): the integer sums only filter,
    every survivor gets its reference-order sum from the f32 codebook, so ids and distances must be the oracle's -- and those of
    the exact-table kernel (table mode 1) -- whatever the scale and the cancellation do to the bound."""
    oidx, qs = _lb_index(4800 + len(case) + m, 9000, 24, case, label_perm=(case == "labels"), ndistinct=(5 if case == "ties" else None), d=d, m=m)
    for K, w in ((10, 8), (1, 1), (3, 2), (64, 5), (10, 24)):
        exp = oidx.knn_search(qs, K, w)
        res = {}
        for mode in (0, 1, 2):
            g = gpu_index(native, oidx)
            # -3: behind the stand-alone top-w selection, the way large batches run
            g.set_tuning(-3 if mode == 2 else -1, 0)
            g.set_table_mode(1 if mode == 1 else 2)      # 2: the matrix-core rounds also where they do not pay (m = 16)
            g.reset_stats()
            res[mode] = g.search_raw(qs, K, w)
            st = g.get_stats()
            assert st["last_lb"] == (0 if mode == 1 else 1), st
            if mode != 1:
                assert st["lb_survivors"] >= min(K, 1)
            helpers.assert_same_results(res[mode], exp, what="lb tables %s mode %d K=%d w=%d" % (case, mode, K, w))
        assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
        assert all(np.array_equal(a, b) for a, b in zip(res[0], res[2]))


@pytest.mark.parametrize("d,m", [(768, 48), (96, 16)])
@pytest.mark.parametrize("case", ["random", "labels", "outlier_codewords", "offset", "far_queries", "tiny_codebooks", "big_codebooks", "exact_hits",
                                  "tiny_scale", "huge_scale"])
@pytest.mark.parametrize("table_mode", [0, 3])
def test_matrix_core_tables_bound_the_reference_entries(native, case, d, m, table_mode):
    """The contract of the matrix-core table build (lbscan.hip.h), entry by entry: with E = ||cb - r||^2 in exact arithmetic on the
    f32 operands the reference uses (r = fl(q - c), src/coarsequantizers.jl:40-45; src/index.jl:232-236), every byte q of the table satisfies
    base + q / inv <= E (a LOWER bound: what makes the filter exact) and E <= base + (q + 1) / inv + mu (||cb||^2 + ||r||^2) with
    q <= 254 (no saturation: what makes the upper-bound selector valid)."""
    oidx, qs = _lb_index(5200 + len(case) + m, 600, 6, case, label_perm=(case == "labels"), d=d, m=m)
    g = gpu_index(native, oidx)
    g.set_table_mode(table_mode)
    MU = 1.48e-3 if table_mode == 0 else 9.2e-5        # f16 codewords (round 5: 2^-9.4) / the three-product bf16 split (2^-13.4)
    dsub = d // m
    worst_lo, worst_hi, steps = 0.0, 0.0, []
    for qi in range(6):
        for cell in (qi % 6, (qi + 3) % 6):
            tab, inv, sbase, nn, base, r2 = g.debug_lb_table(qs[qi], cell)
            r = (qs[qi] - oidx.centroids[cell]).astype(np.float32)                      # one IEEE subtraction per element
            for ii in range(m):
                cb = oidx.codebooks[ii].astype(np.float64)                              # [ksub, dsub]
                rr = r[ii * dsub:(ii + 1) * dsub].astype(np.float64)
                E = ((cb - rr) ** 2).sum(1)                                             # exact to double rounding
                N = (cb ** 2).sum(1) + (rr ** 2).sum()
                q = tab[ii][oidx.labels[ii]].astype(np.float64)                         # slot = label
                assert abs(r2[ii] - (rr ** 2).sum()) <= 1e-5 * (rr ** 2).sum() + 1e-30
                assert q.max() <= 254, (case, qi, cell, ii, q.max())
                if inv <= 0.0:
                    assert (q == 0).all() and base[ii] <= E.min() * (1 + 1e-12)
                    continue
                lo = float(base[ii]) + q / inv
                hi = float(base[ii]) + (q + 1.0) / inv + MU * N
                assert (lo <= E * (1 + 1e-12) + 1e-300).all(), (case, qi, cell, ii, float((lo - E).max()), inv)
                assert (E <= hi * (1 + 1e-12)).all(), (case, qi, cell, ii, float((E - hi).max()), inv)
                steps.append(float(((E - lo) * inv).mean()))
            assert abs(sbase - float(base.astype(np.float64).sum())) <= 1e-5 * max(1e-30, float(base.sum()))
    if steps:
        assert 0.0 <= np.mean(steps) <= 2.5, np.mean(steps)    # the bound is tight: about one quantisation step below the entry


def test_pruning_on_a_trained_million_point_index(native):
    """Probe pruning where it matters: a TRAINED index (native k-means + PQ) over n = 1e6 clustered points, the BASELINE mixture's
    shape (d = 128, kc = 1024, m = 8).  Query-major (probe-level pruning) and list-major with four and one queries per code stream
    (work-item pruning): pruning on and off give the same bits, the oracle agrees on a sample, and on this data pruning fires."""
    import torch
    n, d, kc, m = 1_000_000, 128, 1024, 8
    gen = torch.Generator().manual_seed(99)
    cent0 = torch.rand((kc, d), generator=gen)
    gen.manual_seed(1234)
    which = torch.randint(0, kc, (n,), generator=gen)
    x = (cent0[which] + 0.1 * torch.randn((n, d), generator=gen)).numpy()
    gen.manual_seed(4321)
    qs = (cent0[torch.randint(0, kc, (256,), generator=gen)] + 0.1 * torch.randn((256, d), generator=gen)).numpy()
    cent, cbs, labels = native.trainer.train_ivfadc_hip(x[::5].copy(), kc, 256, m, 10, 10, seed=7)
    g = native.IVFADCIndex.from_arrays(cent, cbs, labels)
    g._append(x, np.arange(n, dtype=np.uint32))
    offsets, codes, ids = g._lists()
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, codes, ids)
    exp = oidx.knn_search(qs[:32], 10, 8)
    for plan, chunk in ((-1, 0), (4, 1024), (1, 1024)):
        res = {}
        for on in (1, 0):
            g.set_tuning(plan, chunk)
            g.set_pruning(on)
            g.reset_stats()
            res[on] = g.search_raw(qs, 10, 8)
            st = g.get_stats()
            if on:
                assert 0 < st["pruned_points"] < st["scanned_points"], (plan, st)
            else:
                assert st["pruned_points"] == 0
        assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1])), plan
        helpers.assert_same_results(tuple(a[:32] for a in res[1]), exp, what="trained 1e6 plan %d" % plan)


@pytest.mark.parametrize("seed,n,d,kc,m,ksub", [
    (21, 6000, 128, 64, 8, 256),        # m = 8 kernel, coarse search inside the launch
    (22, 5000, 96, 50, 16, 256),        # m = 16 / dsub = 6
    (23, 3000, 96, 40, 48, 64),         # m = 48, ksub < 256
    (24, 1500, 12, 9, 4, 32),           # generic kernel, kc % 4 != 0 (streaming row selection)
    (25, 800, 8, 7, 1, 256),            # m = 1
    (26, 9000, 32, 2304, 8, 256),       # kc > 2048: the exact coarse kernel runs first (two launches)
])
def test_small_batch_single_launch_path(native, seed, n, d, kc, m, ksub):
    """The latency path (smallq.hip.h): nq in {1, 2, 3, 16} x w in {1, 8, 32}, one launch, (query, probe, chunk)-parallel with a
    last-arriver merge -- the reference's primary entry is ONE query with w = 1 (src/index.jl:204-208).  Ids and distances are the
    oracle's, repeated calls reuse the arrival counters, and the throughput plans agree bit for bit."""
    oidx, data = helpers.build_index(seed, n, d, kc, m, ksub, label_perm=(seed % 2 == 0))
    rng = np.random.default_rng(seed)
    g = gpu_index(native, oidx)
    for nq in (1, 2, 3, 16):
        qs = np.concatenate([rng.random((nq - 1, d), dtype=np.float32), data[:1]]) if nq > 1 else data[:1].copy()
        for w in (1, 8, 32):
            for K in (1, 10, 64):
                g.set_tuning(0, 0)
                g.reset_stats()
                got = g.search_raw(qs, K, w)
                st = g.get_stats()
                assert st["last_qg"] == (-3 if nq * min(w, kc) <= 512 else st["last_qg"]), st
                exp = oidx.knn_search(qs, K, w)
                helpers.assert_same_results(got, exp, what="small batch nq=%d w=%d K=%d" % (nq, w, K))
                assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf])
                if st["last_qg"] == -3:
                    # B_alg of the latency path: the lengths of the probed lists (the total used to be read from an inactive lane)
                    sizes = np.diff(oidx.offsets)
                    probed = sum(int(sizes[oidx.coarse_search(qs[r], min(w, kc))[0]].sum()) for r in range(nq))
                    assert st["scanned_points"] == probed, (st["scanned_points"], probed)
                again = g.search_raw(qs, K, w)                     # the counters were re-armed by the last arriver
                helpers.assert_same_results(again, exp, what="small batch, second call nq=%d w=%d K=%d" % (nq, w, K))
        g.set_tuning(-1, 0)                                        # the query-major batch kernel on the same queries
        other = g.search_raw(qs, 10, 8)
        g.set_tuning(0, 0)
        helpers.assert_same_results(g.search_raw(qs, 10, 8), other, what="small batch vs query-major")
        g.set_coarse_mode(5)                                       # the coarse search inside the launch (kc <= 2048)
        helpers.assert_same_results(g.search_raw(qs, 10, 8), other, what="small batch, coarse inside")
        assert g.get_stats()["last_qg"] == -3
        g.set_coarse_mode(0)


@pytest.mark.parametrize("seed,n,d,kc,m,ksub,expect_rider", [
    (41, 20000, 128, 130, 8, 256, True),        # the headline kernel family (m = 8, dsub = 16), partial centroid / query tiles
    (42, 9000, 96, 257, 16, 256, True),         # m = 16 / dsub = 6
    (43, 6000, 40, 64, 5, 64, True),            # generic kernel, ksub < 256
    (44, 6000, 50, 100, 10, 256, False),        # d % 8 != 0: no rider form of the exact coarse kernel -> the hint is ignored
])
def test_next_batch_coarse_rides_behind_the_scan(native, seed, n, d, kc, m, ksub, expect_rider):
    """ivfadc_set_next_queries: the exact coarse tiles of the hinted batch ride behind a query-major scan launch and the hinted search
    starts from the finished rows.  Results are the oracle's in every order of hints, batches, K / w and in-place edits; a hint is
    good for one search, and rows computed for other queries are never used."""
    import torch
    oidx, data = helpers.build_index(seed, n, d, kc, m, ksub, label_perm=(seed % 2 == 1))
    rng = np.random.default_rng(seed)
    g = gpu_index(native, oidx)
    g.set_tuning(-1, 0)                                             # query-major at every batch size
    dev = torch.device("cuda:0")
    sets = [np.concatenate([rng.random((nq - 1, d), dtype=np.float32), data[:1]]) for nq in (100, 100, 37, 130)]
    qdev = [torch.from_numpy(x).to(dev) for x in sets]

    def search(i, K, w, hint=None, oracle=None):
        nq = sets[i].shape[0]
        ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
        dist = torch.zeros(nq * K, dtype=torch.float32, device=dev)
        cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
        if hint is not None:
            g.set_next_queries(sets[hint].shape[0], qdev[hint].data_ptr(), 100 + hint)   # the sets never change: one token each
        g.set_query_token(100 + i)
        g.search_device(nq, qdev[i].data_ptr(), K, w, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        st = g.get_stats()
        got = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dist.cpu().numpy().reshape(nq, K), cnt.cpu().numpy())
        helpers.assert_same_results(got, (oracle or oidx).knn_search(sets[i], K, w), what="riders: set %d K=%d w=%d hint=%s" % (i, K, w, hint))
        return st

    st = search(0, 10, 8, hint=1)
    assert st["last_rider"] == (1 if expect_rider else 0) and st["coarse_prefetched"] == 0, st
    st = search(1, 10, 8, hint=0)                                  # the hinted batch: its rows stand; it carries riders itself
    assert st["coarse_prefetched"] == (1 if expect_rider else 0) and st["last_rider"] == (1 if expect_rider else 0), st
    st = search(0, 3, 20)                                          # hinted by the search before, other K and w, no hint of its own
    assert st["coarse_prefetched"] == (1 if expect_rider else 0) and st["last_rider"] == 0, st
    st = search(0, 10, 8)                                          # nothing stands any more
    assert st["coarse_prefetched"] == 0 and st["last_rider"] == 0, st
    search(0, 10, 8, hint=2)                                       # rows for set 2 (37 queries) ...
    st = search(3, 10, 8, hint=2)                                  # ... but another batch comes first: its rows are computed as usual
    assert st["coarse_prefetched"] == 0, st
    st = search(2, 64, 3)
    assert st["coarse_prefetched"] == (1 if expect_rider else 0), st
    search(1, 10, 8, hint=3)
    if ksub > 1:                                                   # in-place edits between the hint and the hinted search: centroids do not move
        pts = rng.random((25, d), dtype=np.float32)
        g._append(pts, np.arange(n, n + 25, dtype=np.uint32))
        g._delete_ids(np.array([0, 5, n + 3], np.uint32))
        st = search(3, 10, 8, oracle=_oracle_of(g, oidx))
        assert st["coarse_prefetched"] == (1 if expect_rider else 0), st
    g.set_next_queries(sets[1].shape[0], qdev[1].data_ptr(), 101)
    g.set_next_queries(0, 0, 0)                                    # a withdrawn hint
    st = search(0, 10, 8, oracle=_oracle_of(g, oidx))
    assert st["last_rider"] == 0, st
    if not expect_rider:
        return
    cur = _oracle_of(g, oidx)

    def raw(i, K, w, token):
        nq = sets[i].shape[0]
        ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
        dist = torch.zeros(nq * K, dtype=torch.float32, device=dev)
        cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
        if token is not None:
            g.set_query_token(token)
        g.search_device(nq, qdev[i].data_ptr(), K, w, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        st = g.get_stats()
        got = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dist.cpu().numpy().reshape(nq, K), cnt.cpu().numpy())
        helpers.assert_same_results(got, cur.knn_search(sets[i], K, w), what="riders / tokens: set %d" % i)
        return st

    # a search that declares no token never picks rows up (the safe default) ...
    g.set_next_queries(sets[1].shape[0], qdev[1].data_ptr(), 7)
    assert raw(0, 10, 8, None)["last_rider"] == 1
    assert raw(1, 10, 8, None)["coarse_prefetched"] == 0
    # ... nor does one that declares another generation: the staging buffer was REFILLED between the hint and its search
    g.set_next_queries(sets[1].shape[0], qdev[1].data_ptr(), 8)
    assert raw(0, 10, 8, None)["last_rider"] == 1
    sets[1] = np.ascontiguousarray(sets[1][::-1] * np.float32(0.5))
    qdev[1].copy_(torch.from_numpy(sets[1]))
    torch.cuda.synchronize()
    assert raw(1, 10, 8, 9)["coarse_prefetched"] == 0
    # rows serve the very next search on EVERY path: a generic-path search in between, the buffer refilled, then a search that even
    # declares the hint's token -- nothing stands any more (the rows used to survive searches that did not look at them)
    g.set_next_queries(sets[1].shape[0], qdev[1].data_ptr(), 10)
    assert raw(0, 10, 8, None)["last_rider"] == 1
    g.set_tuning(-2, 0)
    raw(2, 10, 8, None)
    g.set_tuning(-1, 0)
    sets[1] = np.ascontiguousarray(sets[1][::-1] + np.float32(0.25))
    qdev[1].copy_(torch.from_numpy(sets[1]))
    torch.cuda.synchronize()
    assert raw(1, 10, 8, 10)["coarse_prefetched"] == 0
    # the declared generation matches: picked up
    g.set_next_queries(sets[1].shape[0], qdev[1].data_ptr(), 11)
    assert raw(0, 10, 8, None)["last_rider"] == 1
    assert raw(1, 10, 8, 11)["coarse_prefetched"] == 1


def test_fuzz_next_batch_hints(native):
    """Randomised: chains of device-pointer searches with next-batch hints that are right, wrong (another batch follows), stale (two
    searches later) or absent, over random shapes, batch sizes, K, w, plans and table / coarse modes -- every result against the oracle
    (IVFADC_FUZZ_DRAWS / IVFADC_FUZZ_SEED widen it)."""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("IVFADC_FUZZ_SEED", "2027")))
    dev = torch.device("cuda:0")
    riders = 0
    for it in range(max(8, int(os.environ.get("IVFADC_FUZZ_DRAWS", "60")) // 4)):
        m = int(rng.choice([1, 2, 4, 5, 8, 16]))
        dsub = int(rng.choice([2, 4, 6, 8, 16]))
        d = m * dsub
        kc = int(rng.choice([3, 64, 130, 600, 1024]))
        ksub = int(rng.choice([16, 255, 256]))
        n = int(rng.choice([50, 700, 5000]))
        oidx, data = helpers.build_index(7000 + it, n, d, kc, m, ksub, label_perm=bool(rng.random() < 0.5))
        g = gpu_index(native, oidx)
        g.set_tuning(int(rng.choice([-1, -1, 0, 4, -3])), 0)
        g.set_coarse_mode(int(rng.choice([0, 1])))
        sets = [rng.random((int(rng.choice([1, 17, 70, 130, 300])), d), dtype=np.float32) for _ in range(3)]
        qdev = [torch.from_numpy(x).to(dev) for x in sets]
        for step in range(6):
            i = int(rng.integers(0, 3))
            K, w = int(rng.choice([1, 10, 64, 100])), int(rng.choice([1, 3, 8, 40]))
            nq = sets[i].shape[0]
            ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
            dist = torch.zeros(nq * K, dtype=torch.float32, device=dev)
            cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
            if rng.random() < 0.7:
                h = int(rng.integers(0, 3))
                g.set_next_queries(sets[h].shape[0], qdev[h].data_ptr(), 1 + h)
            if rng.random() < 0.85:
                g.set_query_token(1 + i)
            g.search_device(nq, qdev[i].data_ptr(), K, w, ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
            torch.cuda.synchronize()
            riders += g.get_stats()["coarse_prefetched"]
            got = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dist.cpu().numpy().reshape(nq, K), cnt.cpu().numpy())
            helpers.assert_same_results(got, oidx.knn_search(sets[i], K, w),
                                        what="hint fuzz %d.%d: m=%d dsub=%d kc=%d ksub=%d n=%d nq=%d K=%d w=%d" % (it, step, m, dsub, kc, ksub, n, nq, K, w))
    assert riders > 0          # some searches did start from rows that rode behind their predecessor


def test_fuzz_views_lanes_and_mutations(native):
    """Randomised: searches dealt to the index and to views of it -- several in flight, with hints and tokens per lane --, interleaved with
    push! / delete on the index (after which the old views must refuse and fresh ones see the new state) and with runs of host batches
    (whose odd batches use the internal view).  Every result against an oracle rebuilt from the index's own lists."""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("IVFADC_FUZZ_SEED", "3031")))
    dev = torch.device("cuda:0")
    for it in range(max(4, int(os.environ.get("IVFADC_FUZZ_DRAWS", "60")) // 10)):
        m = int(rng.choice([2, 4, 8, 16]))
        dsub = int(rng.choice([4, 6, 16]))
        d = m * dsub
        kc = int(rng.choice([8, 100, 1024]))
        n = int(rng.choice([600, 6000]))
        oidx, data = helpers.build_index(8100 + it, n, d, kc, m, 256)
        g = gpu_index(native, oidx)
        g.set_tuning(int(rng.choice([-1, 0, 4])), 0)
        views = [g.clone_view() for _ in range(2)]
        next_id = 10_000_000
        sets = [rng.random((int(rng.choice([1, 33, 130, 257])), d), dtype=np.float32) for _ in range(4)]
        qdev = [torch.from_numpy(x).to(dev) for x in sets]
        for step in range(10):
            op = rng.random()
            if op < 0.2:                                   # the index changes under its views
                if rng.random() < 0.6:
                    pts = rng.random((int(rng.integers(1, 6)), d), dtype=np.float32)
                    g._append(pts, np.arange(next_id, next_id + pts.shape[0], dtype=np.uint32))
                    next_id += pts.shape[0]
                else:
                    g._delete_ids(rng.integers(0, n, 3).astype(np.uint32))
                oidx = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, *g._lists())
                with pytest.raises(Exception, match="changed since this view"):
                    views[0].search_raw(sets[0][:1], 3, 1)
                views = [g.clone_view() for _ in range(2)]
                continue
            if op < 0.35:                                  # a run of host batches through the library's own two lanes
                K, w = int(rng.choice([1, 10])), int(rng.choice([1, 4]))
                bs = [sets[int(rng.integers(0, 4))] for _ in range(int(rng.integers(1, 6)))]
                for b, r in zip(bs, g.search_batches_raw(bs, K, w)):
                    helpers.assert_same_results(r, oidx.knn_search(b, K, w), what="fuzz %d.%d run of batches" % (it, step))
                continue
            # three searches in flight on three lanes, each with its own outputs
            K, w = int(rng.choice([1, 10, 70])), int(rng.choice([1, 3, 8]))
            lanes = [g] + views
            picks = [int(rng.integers(0, 4)) for _ in lanes]
            outs = [(torch.zeros(sets[i].shape[0] * K, dtype=torch.int32, device=dev), torch.zeros(sets[i].shape[0] * K, dtype=torch.float32, device=dev),
                     torch.zeros(sets[i].shape[0], dtype=torch.int32, device=dev)) for i in picks]
            torch.cuda.synchronize()       # (torch fills them on ITS stream: done before a lane's stream writes results into them)
            for ln, i, o in zip(lanes, picks, outs):
                nq = sets[i].shape[0]
                if rng.random() < 0.6:
                    hq = int(rng.integers(0, 4))
                    ln.set_next_queries(sets[hq].shape[0], qdev[hq].data_ptr(), 1 + hq)
                if rng.random() < 0.8:
                    ln.set_query_token(1 + i)
                ln.search_device(nq, qdev[i].data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
                if os.environ.get("IVFADC_FUZZ_SERIAL"):      # (debugging aid: one search at a time)
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            for ln, i, o in zip(lanes, picks, outs):
                nq = sets[i].shape[0]
                got = (o[0].cpu().numpy().view(np.uint32).reshape(nq, K), o[1].cpu().numpy().reshape(nq, K), o[2].cpu().numpy())
                helpers.assert_same_results(got, oidx.knn_search(sets[i], K, w),
                                            what="fuzz %d.%d lanes: m=%d dsub=%d kc=%d n=%d nq=%d K=%d w=%d" % (it, step, m, dsub, kc, n, nq, K, w))


@pytest.mark.parametrize("kc,coarse", [(600, 0), (2500, 0), (2500, 1)])
def test_device_entries_of_an_index_with_views_select_top_w_stand_alone(native, kc, coarse):
    """Several batches in flight (the index has views) through the device-pointer entry: batches of >= 4 x CUs queries take the top-w
    selection out of the scan workgroups' prologue (make_plan: pl.lanes; a wave per query) -- with the exact coarse kernel and behind the
    matrix-core filter, with and without next-batch hints, on the index and on the view, results as without views and as the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    oidx, data = helpers.build_index(70 + coarse, 30000, 64, kc, 8, 256, mode="random")
    rng = np.random.default_rng(70)
    sets = [np.concatenate([rng.random((1100, 64), dtype=np.float32), data[:5]]) for _ in range(2)]
    qdev = [torch.as_tensor(x, device=dev) for x in sets]
    g = gpu_index(native, oidx)
    g.set_coarse_mode(coarse)
    K, w = 10, 6
    alone = [g.search_raw(x, K, w) for x in sets]                # no views yet: the fused selection
    exp = [oidx.knn_search(x, K, w) for x in sets]
    for a, e in zip(alone, exp):
        helpers.assert_same_results(a, e, what="no views kc=%d" % kc)
    v = g.clone_view()
    for hint in (False, True):
        outs = []
        for ln, i in ((g, 0), (v, 1), (g, 1), (v, 0)):
            nq = sets[i].shape[0]
            o = (torch.zeros(nq * K, dtype=torch.int32, device=dev), torch.zeros(nq * K, dtype=torch.float32, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            if hint:
                ln.set_query_token(1 + i)
                ln.set_next_queries(sets[1 - i].shape[0], qdev[1 - i].data_ptr(), 2 - i)
            ln.search_device(nq, qdev[i].data_ptr(), K, w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            outs.append((i, nq, o))
        torch.cuda.synchronize()
        for i, nq, o in outs:
            got = (o[0].cpu().numpy().view(np.uint32).reshape(nq, K), o[1].cpu().numpy().reshape(nq, K), o[2].cpu().numpy())
            helpers.assert_same_results(got, exp[i], what="lanes kc=%d coarse=%d hint=%d" % (kc, coarse, hint))
            assert all(np.array_equal(a, b) for a, b in zip(got, alone[i]))


def test_probes_per_round_follow_the_pruned_fraction(native):
    """make_plan switches the query-major kernel from two probes per round to one once the scans report that most probed points are pruned
    (asynchronous counter snapshots: fb_snapshot / fb_poll).  Thirty searches in a row on clustered queries (everything but the closest
    cell pruned), then thirty on queries between the centroids (little pruned), on an index and a view of it: every result is the oracle's,
    whatever the plan was at the time."""
    oidx, data = helpers.build_index(81, 20000, 128, 40, 8, 256, mode="random")
    rng = np.random.default_rng(81)
    cent = oidx.centroids
    near = (cent[rng.integers(0, 40, 300)] + np.float32(1e-3) * rng.standard_normal((300, 128))).astype(np.float32)
    far = rng.random((300, 128), dtype=np.float32)
    g = gpu_index(native, oidx)
    g.set_tuning(-1, 0)
    v = g.clone_view()
    for qs, what in ((near, "clustered"), (far, "spread"), (near, "clustered again")):
        exp = oidx.knn_search(qs, 10, 8)
        for it in range(30):
            ln = g if it % 2 == 0 else v
            helpers.assert_same_results(ln.search_raw(qs, 10, 8), exp, what="probes per round: %s, search %d" % (what, it))


def test_small_batch_path_chunks_long_lists_and_ties(native):
    """Few queries on long lists: every probe is cut into chunks, one workgroup each; ties across chunks resolve by visit order."""
    oidx, _ = helpers.build_index(31, 60000, 128, 4, 8, 256, mode="random", ndistinct=7)
    rng = np.random.default_rng(31)
    qs = rng.random((2, 128), dtype=np.float32)
    g = gpu_index(native, oidx)
    g.reset_stats()
    got = g.search_raw(qs, 64, 4)
    st = g.get_stats()
    assert st["last_qg"] == -3 and st["last_scan_grid"] > 2 * 4, st       # more workgroups than (query, probe) pairs: chunks
    helpers.assert_same_results(got, oidx.knn_search(qs, 64, 4), what="chunked small batch with ties")


# ---- round 5: host-pointer entries with memory the library knows (ivfadc_host_alloc / ivfadc_host_register) ------------------------------
def _host_stats(native, gidx):
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    st = nat.HostStats()
    nat.check(nat.lib().ivfadc_get_host_stats(gidx._h, C.byref(st)))
    return {f: getattr(st, f) for f, _ in nat.HostStats._fields_}


def _raw_search(gidx, q, K, w, ids, dists, counts):
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    nat.check(nat.lib().ivfadc_search(gidx._h, q.shape[0], nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float),
                                      nat.ptr(counts, C.c_int32)))


@pytest.mark.parametrize("nq", [1, 7, 64, 65, 700])
def test_host_entries_known_memory_same_results(native, nq):
    """ivfadc_search with pageable arrays, with caller-registered arrays and with library-allocated page-locked arrays (queries ingested
    from / read in place in the caller's array, results written into the caller's arrays by the final kernel): one answer, the oracle's."""
    from ivfadc_jl_amd import _native as nat
    oidx, data = helpers.build_index(41, 6000, 128, 64, 8, 256)
    rng = np.random.default_rng(nq)
    qs = np.concatenate([rng.random((nq - 1, 128), dtype=np.float32), data[:1]])
    K, w = 10, 5
    gidx = gpu_index(native, oidx)
    exp = oidx.knn_search(qs, K, w)
    # pageable
    got = gidx.search_raw(qs, K, w)
    helpers.assert_same_results(got, exp, what="pageable")
    st0 = _host_stats(native, gidx)
    assert st0["queries_direct"] == 0 and st0["results_direct"] == 0
    # caller-registered, deliberately only 4-byte aligned (a slice one float into a larger registered block)
    blk_q = np.zeros(nq * 128 + 1, np.float32)
    blk_i = np.zeros(nq * K + 1, np.uint32); blk_d = np.zeros(nq * K + 1, np.float32); blk_c = np.zeros(nq + 1, np.int32)
    for a in (blk_q, blk_i, blk_d, blk_c):
        nat.host_register(a)
    try:
        q = blk_q[1:].reshape(nq, 128); q[...] = qs
        ids = blk_i[1:].reshape(nq, K); dists = blk_d[1:].reshape(nq, K); counts = blk_c[1:]
        _raw_search(gidx, q, K, w, ids, dists, counts)
        helpers.assert_same_results((ids, dists, counts), exp, what="registered, 4-byte aligned")
        st1 = _host_stats(native, gidx)
        assert st1["queries_direct"] == st0["queries_direct"] + 1 and st1["results_direct"] == st0["results_direct"] + 1
        # queries known, results not (mixed): the library's pinned block carries the results
        ids2 = np.zeros((nq, K), np.uint32); d2 = np.zeros((nq, K), np.float32); c2 = np.zeros(nq, np.int32)
        _raw_search(gidx, q, K, w, ids2, d2, c2)
        helpers.assert_same_results((ids2, d2, c2), exp, what="registered queries, pageable results")
        assert _host_stats(native, gidx)["results_direct"] == st1["results_direct"]
    finally:
        for a in (blk_q, blk_i, blk_d, blk_c):
            nat.host_unregister(a)
    # once unregistered the same arrays are staged again
    _raw_search(gidx, q, K, w, ids, dists, counts)
    helpers.assert_same_results((ids, dists, counts), exp, what="after unregister")
    # library-allocated blocks through the reference-shaped entry (knn_search packs into them)
    oi, od = native.knn_search(gidx, [qs[i] for i in range(nq)], K, w)
    for r in range(nq):
        c = int(exp[2][r])
        assert np.array_equal(oi[r], exp[0][r, :c]) and np.array_equal(od[r], exp[1][r, :c])
    st2 = _host_stats(native, gidx)
    assert st2["queries_direct"] >= st1["queries_direct"] + 1
    if nq <= 64:
        assert st2["zero_copy"] >= 1          # the latency path read its rows in place


def test_host_register_contract(native):
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    a = np.zeros(4096, np.float32)
    nat.host_register(a)
    with pytest.raises(nat.IVFADCError):          # overlapping registration
        nat.host_register(a[10:100])
    with pytest.raises(nat.IVFADCError):          # not the start of a registered range
        nat.check(nat.lib().ivfadc_host_unregister(C.c_void_p(a.ctypes.data + 64)))
    nat.host_unregister(a)
    with pytest.raises(nat.IVFADCError):
        nat.host_unregister(a)
    p = nat.PinnedArray((16, 8), np.float32)
    with pytest.raises(nat.IVFADCError):          # a library block is freed with ivfadc_host_free, not unregistered
        nat.check(nat.lib().ivfadc_host_unregister(p._p))
    p.close()
    assert nat.lib().ivfadc_abi_version() == nat.ABI_VERSION


@pytest.mark.parametrize("known", ["none", "queries", "all"])
def test_search_batches_known_memory_and_ragged_batches(native, known):
    """ivfadc_search_batches: ragged batches (empty ones, a single query, the latency path, large ones), the queries / the results in
    memory the library knows or not -- every batch's results are ivfadc_search's, i.e. the oracle's."""
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    oidx, data = helpers.build_index(43, 9000, 128, 128, 8, 256)
    rng = np.random.default_rng(7)
    sizes = np.array([300, 0, 1, 257, 64, 0, 1024, 5, 333, 300, 300, 300, 300, 300, 300, 300, 300, 300, 300, 2], np.int64)
    total = int(sizes.sum())
    K, w = 10, 8
    qs = rng.random((total, 128), dtype=np.float32)
    exp = oidx.knn_search(qs, K, w)
    gidx = gpu_index(native, oidx)
    pins = []
    if known == "none":
        q = qs; ids = np.zeros((total, K), np.uint32); dists = np.zeros((total, K), np.float32); counts = np.zeros(total, np.int32)
    else:
        pq = nat.PinnedArray((total, 128), np.float32); pins.append(pq)
        q = pq.a; q[...] = qs
        if known == "all":
            pi, pd, pc = nat.PinnedArray((total, K), np.uint32), nat.PinnedArray((total, K), np.float32), nat.PinnedArray(total, np.int32)
            pins += [pi, pd, pc]
            ids, dists, counts = pi.a, pd.a, pc.a
        else:
            ids = np.zeros((total, K), np.uint32); dists = np.zeros((total, K), np.float32); counts = np.zeros(total, np.int32)
    for rep in range(3):
        ids[...] = 0; dists[...] = 0; counts[...] = -1
        nat.check(nat.lib().ivfadc_search_batches(gidx._h, len(sizes), nat.ptr(sizes, C.c_int64), nat.ptr(q, C.c_float), K, w,
                                                  nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        helpers.assert_same_results((ids, dists, counts), exp, what="batches, known=%s, rep %d" % (known, rep))
    st = _host_stats(native, gidx)
    assert st["batches"] == 3 * int((sizes > 0).sum())
    assert st["queries_direct"] == (3 if known != "none" else 0) and st["results_direct"] == (3 if known == "all" else 0)
    # many small batches: more batches than upload groups (256), one lane's worth each
    nb = 700
    sz = np.full(nb, 3, np.int64)
    q2 = np.ascontiguousarray(qs[:nb * 3])
    i2 = np.zeros((nb * 3, K), np.uint32); d2 = np.zeros((nb * 3, K), np.float32); c2 = np.zeros(nb * 3, np.int32)
    nat.check(nat.lib().ivfadc_search_batches(gidx._h, nb, nat.ptr(sz, C.c_int64), nat.ptr(q2, C.c_float), K, w,
                                              nat.ptr(i2, C.c_uint32), nat.ptr(d2, C.c_float), nat.ptr(c2, C.c_int32)))
    helpers.assert_same_results((i2, d2, c2), (exp[0][:nb * 3], exp[1][:nb * 3], exp[2][:nb * 3]), what="700 batches of 3")
    for p in pins:
        p.close()


def test_search_batches_error_leaves_nothing_running_and_stats_survive_a_push(native):
    """A failing batch run returns with both lanes drained, and the counters of a second lane that a push! made stale
    stay in the totals (they used to go backwards)."""
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    oidx, data = helpers.build_index(44, 5000, 128, 64, 8, 256)
    gidx = gpu_index(native, oidx)
    rng = np.random.default_rng(3)
    qs = rng.random((4 * 300, 128), dtype=np.float32)
    K, w = 10, 4
    sizes = np.full(4, 300, np.int64)
    out = lambda: (np.zeros((1200, K), np.uint32), np.zeros((1200, K), np.float32), np.zeros(1200, np.int32))
    ids, dists, counts = out()
    L = nat.lib()
    nat.check(L.ivfadc_search_batches(gidx._h, 4, nat.ptr(sizes, C.c_int64), nat.ptr(qs, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                      nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
    before = gidx.get_stats()
    assert before["queries"] == 1200
    # a workspace too small for any plan makes a later batch fail after earlier ones were enqueued
    gidx.set_workspace_limit(1)
    rc = L.ivfadc_search_batches(gidx._h, 4, nat.ptr(sizes, C.c_int64), nat.ptr(qs, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                 nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32))
    gidx.set_workspace_limit(8 << 30)
    # (whether or not the tiny budget fails on this shape, the next run must be whole and right)
    ids, dists, counts = out()
    nat.check(L.ivfadc_search_batches(gidx._h, 4, nat.ptr(sizes, C.c_int64), nat.ptr(qs, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                      nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
    helpers.assert_same_results((ids, dists, counts), oidx.knn_search(qs, K, w), what="after a failed run (rc %d)" % rc)
    mid = gidx.get_stats()["queries"]
    assert mid >= before["queries"] + 1200
    # push! invalidates the second lane; its share of the counters stays
    native.push(gidx, data[0] + 0.5)
    ids, dists, counts = out()
    nat.check(L.ivfadc_search_batches(gidx._h, 4, nat.ptr(sizes, C.c_int64), nat.ptr(qs, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                      nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
    after = gidx.get_stats()
    assert after["queries"] == mid + 1200, (before["queries"], mid, after["queries"])
    assert after["scanned_points"] > gidx.get_stats()["scanned_points"] - 1 and after["scanned_points"] >= before["scanned_points"]


# ---- round 5: robustness ---------------------------------------------------------------------------------------
@pytest.mark.timeout(300)
@pytest.mark.parametrize("plan", ["query_major", "list_major", "latency", "generic", "mfma_coarse", "lb_m48", "topw_standalone"])
def test_non_finite_queries_are_contained(native, plan):
    """Non-finite input.  Quantizers: refused at ivfadc_create (IVFADC_ERR_INVALID).  Queries are not scanned on the hot path; the
    contract is containment: a query with a NaN / infinite / overflowing component gets unspecified neighbours but valid counts and
    stored ids, no plan hangs or writes out of bounds, and every finite query of the same batch gets exactly the oracle's answer."""
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    if plan == "lb_m48":
        oidx, data = helpers.build_index(51, 3000, 768, 40, 48, 256)
    elif plan == "mfma_coarse":
        oidx, data = helpers.build_index(52, 6000, 128, 256, 8, 256)
    else:
        oidx, data = helpers.build_index(53, 6000, 128, 64, 8, 256)
    d = oidx.d
    gidx = gpu_index(native, oidx)
    if plan == "query_major":
        gidx.set_tuning(-1, 0)
    elif plan == "list_major":
        gidx.set_tuning(4, 0)
    elif plan == "generic":
        gidx.set_tuning(-2, 0)
    elif plan == "mfma_coarse":
        gidx.set_coarse_mode(2)
    elif plan == "topw_standalone":
        gidx.set_tuning(-3, 0)
    nq = 12 if plan == "latency" else 96
    rng = np.random.default_rng(5)
    qs = rng.random((nq, d), dtype=np.float32)
    bad = {1: np.nan, 3: np.inf, 4: -np.inf, 6: 3e38, 7: -3e38}
    qs[0, :] = np.nan                      # a whole row of NaN
    for r, v in bad.items():
        qs[r, (7 * r) % d] = v
    qs[9, :] = np.inf
    bad_rows = sorted([0, 9] + list(bad))
    good = np.array([r for r in range(nq) if r not in bad_rows])
    K, w = 10, 4
    valid_ids = set(oidx.ids.tolist())
    for rep in range(2):
        ids, dists, counts = gidx.search_raw(qs, K, w)
        assert ((counts >= 0) & (counts <= K)).all()
        for r in bad_rows:
            assert set(ids[r, :counts[r]].tolist()) <= valid_ids, (plan, r, ids[r])
        exp = oidx.knn_search(qs[good], K, w)
        helpers.assert_same_results((ids[good], dists[good], counts[good]), exp, what="finite queries beside non-finite ones (%s)" % plan)
    # the handle is as good as before
    q2 = rng.random((33, d), dtype=np.float32)
    helpers.assert_same_results(gidx.search_raw(q2, K, w), oidx.knn_search(q2, K, w), what="after non-finite batches (%s)" % plan)
    # quantizers with a non-finite value are refused before any device work
    cent = oidx.centroids.copy(); cent[3, 5] = np.nan
    with pytest.raises(nat.IVFADCError, match="non-finite"):
        native.IVFADCIndex.from_arrays(cent, oidx.codebooks, oidx.labels)
    cbs = oidx.codebooks.copy(); cbs[1, 7, 0] = np.inf
    with pytest.raises(nat.IVFADCError, match="non-finite"):
        native.IVFADCIndex.from_arrays(oidx.centroids, cbs, oidx.labels)


def test_failing_and_noop_mutators_leave_views_valid(native):
    """begin_mutation runs after validation -- a mutator that fails or changes nothing must not invalidate the views (or the
    internal second lane of ivfadc_search_batches), and the cumulative counters never go backwards."""
    oidx, data = helpers.build_index(54, 4000, 128, 32, 8, 256)
    g = gpu_index(native, oidx)
    v = g.clone_view()
    qs = np.random.default_rng(1).random((50, 128), dtype=np.float32)
    exp = oidx.knn_search(qs, 10, 4)
    g._shift_ids(0)                                                   # no-op
    assert g._delete_ids(np.array([4_000_000, 4_000_001], np.uint32)) == 0      # ids that are not stored: nothing changes
    with pytest.raises(Exception):
        g.set_lists(np.array([1] + [0] * 32, np.int64), oidx.codes, oidx.ids)     # bad offsets: refused
    with pytest.raises(AssertionError):
        g._append(np.zeros((1, 128), np.float32), None) if False else native.push(g, np.zeros(5, np.float32))   # wrong dimension
    helpers.assert_same_results(v.search_raw(qs, 10, 4), exp, what="view after failing / no-op mutators")
    g._append(data[:1] + 0.5, np.array([4000], np.uint32))            # a real change: now the view is stale
    with pytest.raises(Exception, match="changed since this view"):
        v.search_raw(qs[:2], 10, 4)


# ---- round 5: certified two-level coarse search (twolevel.hip.h) ----------------------------------------------------------
def _clustered_index(seed, n, d, kc, m, ncl, sigma, dup=0):
    """An index whose centroids have the structure a trained quantizer has: ncl true centres, kc centroids scattered around them."""
    rng = np.random.default_rng(seed)
    centres = rng.random((ncl, d), dtype=np.float32)
    cent = (centres[rng.integers(0, ncl, kc)] + sigma * rng.standard_normal((kc, d))).astype(np.float32)
    for i in range(dup):                       # exact duplicates: ties that only the cluster id breaks
        cent[kc - 1 - i] = cent[i]
    _, cbs, labels = helpers.make_quantizers(seed, d, kc, m, 256)
    lst = rng.integers(0, kc, n).astype(np.int32)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    order = np.argsort(lst, kind="stable")
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(np.bincount(lst, minlength=kc), out=offsets[1:])
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, np.ascontiguousarray(codes[order]), rng.permutation(n).astype(np.uint32))
    queries = (centres[rng.integers(0, ncl, 300)] + sigma * rng.standard_normal((300, d))).astype(np.float32)
    return oidx, queries


@pytest.mark.parametrize("plan", [0, -1, 4])
@pytest.mark.parametrize("shape", ["clustered", "uniform", "duplicates"])
def test_two_level_coarse_search_is_exact(native, shape, plan):
    """ivfadc_set_coarse_mode(h, 6): groups of centroids, triangle-inequality bounds, exact distances of what the bounds let through.
    Exact by construction -- the oracle's probes, ties to the lower cluster id, for structured centroids (most groups skipped), for
    unstructured ones (nothing skipped: the every-group fallback) and with exact duplicate centroids -- under every scan plan."""
    if shape == "uniform":
        oidx, _ = helpers.build_index(61, 20000, 64, 2048, 8, 256, mode="random")
        qs = np.random.default_rng(61).random((300, 64), dtype=np.float32)
    else:
        oidx, qs = _clustered_index(62, 20000, 64, 2048, 8, 24, 0.02, dup=40 if shape == "duplicates" else 0)
        if shape == "duplicates":
            qs[:40] = oidx.centroids[:40]            # on the duplicated centroids themselves: distance 0 twice
    g = gpu_index(native, oidx)
    g.set_tuning(plan, 0)
    g.set_coarse_mode(6)
    for K, w in ((10, 1), (10, 8), (5, 32), (10, 64), (70, 16)):
        g.reset_stats()
        got = g.search_raw(qs, K, w)
        helpers.assert_same_results(got, oidx.knn_search(qs, K, w), what="two-level %s plan %d K=%d w=%d" % (shape, plan, K, w))
        st = g.get_stats()
        assert st["last_twolevel"] == 1 and st["twolevel_groups"] >= 32
        frac = st["coarse_visited"] / (qs.shape[0] * 2048.0)
        if shape == "clustered" and w <= 8:
            assert frac < 0.2, frac                 # the bounds cut: a few groups per query
        if shape == "uniform" and w >= 8:
            assert frac > 0.5, frac                 # no structure: most groups are visited (the every-group fallback), the result is still exact
    # w > 64 is outside the two-level search's reach: the exhaustive kernels answer, same results
    got = g.search_raw(qs[:50], 10, 100)
    helpers.assert_same_results(got, oidx.knn_search(qs[:50], 10, 100), what="w = 100 falls back")
    assert g.get_stats()["last_twolevel"] == 0
    # mode 7 / 0 on a quantizer below the automatic threshold: exhaustive again; identical results
    g.set_coarse_mode(7)
    helpers.assert_same_results(g.search_raw(qs, 10, 8), oidx.knn_search(qs, 10, 8), what="two-level off")
    assert g.get_stats()["last_twolevel"] == 0


def test_two_level_automatic_mode_follows_the_self_probe(native):
    """kc >= 4096: the grouping is built on the first search; a structured quantizer keeps it (self-probe: a small fraction of the kc
    distances), an unstructured one stays with the exhaustive kernels.  Views and the run-of-batches call inherit the decision."""
    oidx, qs = _clustered_index(63, 30000, 32, 8192, 4, 256, 0.02)
    g = gpu_index(native, oidx)
    exp = oidx.knn_search(qs, 10, 8)
    helpers.assert_same_results(g.search_raw(qs, 10, 8), exp, what="automatic, structured")
    st = g.get_stats()
    assert st["last_twolevel"] == 1 and 0.0 <= st["twolevel_probe_fraction"] <= 0.02, st
    v = g.clone_view()
    helpers.assert_same_results(v.search_raw(qs, 10, 8), exp, what="view inherits the grouping")
    assert v.get_stats()["last_twolevel"] == 1
    for b, r in zip([qs[:100], qs[100:300]], g.search_batches_raw([qs[:100], qs[100:300]], 10, 8)):
        helpers.assert_same_results(r, oidx.knn_search(b, 10, 8), what="batches, two-level")
    # in-place push keeps the grouping (it depends on the quantizer alone)
    native.push(g, qs[0])
    oidx2 = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, *g._lists())
    helpers.assert_same_results(g.search_raw(qs, 10, 8), oidx2.knn_search(qs, 10, 8), what="after push")
    # unstructured quantizer of the same size: the probe says no
    oidx3, _ = helpers.build_index(64, 30000, 64, 4096, 8, 256, mode="random")
    g3 = gpu_index(native, oidx3)
    q3 = np.random.default_rng(64).random((200, 64), dtype=np.float32)
    helpers.assert_same_results(g3.search_raw(q3, 10, 8), oidx3.knn_search(q3, 10, 8), what="automatic, unstructured")
    st3 = g3.get_stats()
    assert st3["last_twolevel"] == 0 and st3["twolevel_probe_fraction"] > 0.02, st3


@pytest.mark.timeout(600)
def test_threads_index_view_and_a_mutator(native):
    """Host threads: two threads search the index itself (calls on one handle are serialised by its mutex), one
    searches a view of it (concurrently: another handle), and a fourth pushes and deletes points all the while.  The pushed points lie
    far from every query, so the oracle's answer never changes: every search that returns must return exactly it; a view search may
    instead refuse because the index changed since the view was taken -- then the thread takes a new view and goes on."""
    import threading
    oidx, data = helpers.build_index(71, 20000, 64, 64, 8, 256)
    g = gpu_index(native, oidx)
    rng = np.random.default_rng(71)
    qs = [rng.random((64 + 17 * i, 64), dtype=np.float32) for i in range(4)]
    exp = [oidx.knn_search(q, 10, 6) for q in qs]
    errors, stale, done = [], [0], [0, 0, 0]
    stop = threading.Event()

    def same(got, e):
        return all(np.array_equal(a, b) for a, b in zip(got, e))

    def on_index(slot):
        try:
            for i in range(400):
                j = (i + slot) % 4
                if not same(g.search_raw(qs[j], 10, 6), exp[j]):
                    errors.append("index thread %d: wrong result at iteration %d" % (slot, i))
                    return
                done[slot] += 1
        except Exception as e:           # noqa: BLE001
            errors.append("index thread %d: %r" % (slot, e))

    def on_view():
        try:
            v = g.clone_view()
            for i in range(400):
                j = i % 4
                try:
                    got = v.search_raw(qs[j], 10, 6)
                except Exception as e:   # noqa: BLE001
                    if "changed since this view" not in str(e):
                        raise
                    stale[0] += 1
                    v = g.clone_view()
                    continue
                if not same(got, exp[j]):
                    errors.append("view thread: wrong result at iteration %d" % i)
                    return
                done[2] += 1
        except Exception as e:           # noqa: BLE001
            errors.append("view thread: %r" % (e,))

    def mutator():
        try:
            far = np.full((3, 64), 100.0, np.float32)
            k = 0
            while not stop.is_set() and k < 200:
                ids = np.arange(3, dtype=np.uint32) + 20000
                g._append(far + k, ids)
                assert g._delete_ids(ids) == 3
                k += 1
        except Exception as e:           # noqa: BLE001
            errors.append("mutator: %r" % (e,))

    ts = [threading.Thread(target=on_index, args=(0,)), threading.Thread(target=on_index, args=(1,)), threading.Thread(target=on_view)]
    tm = threading.Thread(target=mutator)
    for t in ts:
        t.start()
    tm.start()
    for t in ts:
        t.join()
    stop.set()
    tm.join()
    assert not errors, errors
    assert done[0] == 400 and done[1] == 400 and done[2] + stale[0] == 400
    assert stale[0] > 0                      # the view did meet a changed index, and said so
    # afterwards everything is as before
    helpers.assert_same_results(g.search_raw(qs[0], 10, 6), exp[0], what="after the threads")
    assert len(g) == 20000


@pytest.mark.parametrize("listed", [True, False])
def test_f16_coarse_filter_flags_queries_that_leave_its_range(native, listed):
    """Round 5: the large matrix-core coarse filter computes ONE f16 product per score on operands scaled by a power of two.  A query
    component that would leave the f16 range saturates and flags its query, and a flagged query is recomputed exactly -- so queries with
    huge components (1e6 against centroids in [0, 1)), tiny ones, and the ordinary ones beside them all get the oracle's probes; the
    scale follows the centroids (the same index scaled by 2^-20 and by 2^+20 behaves alike)."""
    for scale in (1.0, 2.0 ** -20, 2.0 ** 20):
        oidx, _ = helpers.build_index(81, 30000, 64, 4096 if listed else 2048, 8, 256, mode="random")
        oidx.centroids *= np.float32(scale)
        oidx.codebooks *= np.float32(scale)
        rng = np.random.default_rng(81)
        nq = 8192 if listed else 4096               # (per-tile records are written for the stand-alone top-w of a large batch)
        qs = (rng.random((nq, 64), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        qs[5, 3] = np.float32(1e6 * scale)          # far outside the f16 range after scaling
        qs[9, :] = np.float32(-3e5 * scale)
        qs[11, 7] = np.float32(1e-30 * scale)       # underflows: harmless
        g = gpu_index(native, oidx)
        if listed:
            g.set_tuning(-3, 0)
        w = 16 if listed else 8
        got = g.search_raw(qs, 10, w)
        st = g.get_stats()
        assert st["coarse_f16"] == 1 and st["coarse_fallbacks"] >= 2, st      # the two flagged queries (at least) took the exact path
        assert st["coarse_listed"] == (1 if listed else 0), st
        pick = np.concatenate([[5, 9, 11], np.sort(rng.choice(nq, 90, replace=False))])
        helpers.assert_same_results(tuple(a[pick] for a in got), oidx.knn_search(qs[pick], 10, w), what="f16 filter, scale %g" % scale)
        g.set_coarse_mode(8)
        got8 = g.search_raw(qs, 10, w)
        assert all(np.array_equal(a, b) for a, b in zip(got, got8)), "f16 form and bf16 split disagree"


def test_fuzz_large_coarse_stage(native):
    """Randomised, large coarse problems (where the 128 x 128 matrix-core filter, its per-tile records and the two-level search run): kc, d
    (also not a multiple of 32), data scale from 2^-12 to 2^+12, clustered or uniform centroids, batch, w, plan and coarse mode drawn at
    random; all modes must agree with each other bit for bit and a sample of queries with the oracle.  IVFADC_FUZZ_DRAWS widens it."""
    import os
    rng = np.random.default_rng(int(os.environ.get("IVFADC_FUZZ_SEED", "2027")))
    for it in range(max(6, int(os.environ.get("IVFADC_FUZZ_DRAWS", "60")) // 6)):
        kc = int(rng.choice([2048, 4096, 5000, 9000]))
        d = int(rng.choice([32, 40, 64, 96, 128]))
        m = int(rng.choice([4, 8]))
        nq = int(rng.choice([4096, 6000, 8192]))
        w = int(rng.choice([1, 8, 16, 32, 48]))
        K = int(rng.choice([1, 10, 64]))
        scale = np.float32(2.0 ** int(rng.integers(-12, 13)))
        clustered = bool(rng.random() < 0.5)
        if clustered:
            oidx, qs0 = _clustered_index(3000 + it, 20000, d, kc, m, int(rng.choice([16, 64, 300])), 0.02)
            qs = np.concatenate([qs0, np.random.default_rng(it).random((nq - qs0.shape[0], d), dtype=np.float32)])
        else:
            oidx, _ = helpers.build_index(3000 + it, 20000, d, kc, m, 256, mode="random")
            qs = rng.random((nq, d), dtype=np.float32)
        oidx.centroids *= scale
        oidx.codebooks *= scale
        qs = (qs * scale).astype(np.float32)
        qs[:8] = oidx.centroids[:8]
        plan = int(rng.choice([0, -3, -1]))
        res = {}
        for cmode in (0, 8, 6, 1):
            g = gpu_index(native, oidx)
            g.set_tuning(plan, 0)
            g.set_coarse_mode(cmode)
            res[cmode] = g.search_raw(qs, K, w)
        what = "large coarse fuzz %d: kc=%d d=%d m=%d nq=%d w=%d K=%d scale=%g clustered=%s plan=%d" % (it, kc, d, m, nq, w, K, scale, clustered, plan)
        for cmode in (0, 8, 6):
            assert all(np.array_equal(a, b) for a, b in zip(res[cmode], res[1])), what + ": coarse mode %d differs from the exact kernel" % cmode
        pick = np.sort(rng.choice(nq, 48, replace=False))
        helpers.assert_same_results(tuple(a[pick] for a in res[1]), oidx.knn_search(qs[pick], K, w), what=what)


def test_two_level_coarse_search_under_list_partition(native):
    """The two-level coarse stage feeds the list-partitioned multi-GPU mode like the exhaustive one: every rank's partial keys and the
    merged result are those of the oracle (visit-order bases global, only this rank's lists in the probe histogram)."""
    import torch
    oidx, qs = _clustered_index(65, 20000, 64, 2048, 8, 24, 0.02)
    qs = qs[:80]
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(qs).to(dev)
    g = gpu_index(native, oidx)
    g.set_coarse_mode(6)
    nq, K, w, nparts = qs.shape[0], 10, 6, 3
    exp = oidx.knn_search(qs, K, w)
    keys_all = torch.zeros((nparts, nq, K), dtype=torch.int64, device=dev)
    cnts_all = torch.zeros((nparts, nq), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
    for part in range(nparts):
        g.set_list_partition(nparts, part)
        g.search_device_partial(nq, qd.data_ptr(), K, w, keys_all[part].data_ptr(), cnts_all[part].data_ptr())
        torch.cuda.synchronize()
        assert g.get_stats()["last_twolevel"] == 1
        rk, rc, _ = helpers.numpy_partial_keys(oidx, qs[:10], K, w, nparts, part)
        gk = keys_all[part].cpu().numpy().view(np.uint64)[:10]
        gc = cnts_all[part].cpu().numpy()[:10]
        assert np.array_equal(gc, rc) and all(np.array_equal(gk[r, :rc[r]], rk[r, :rc[r]]) for r in range(10)), "partial keys of part %d" % part
    ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
    dist = torch.zeros(nq * K, dtype=torch.float32, device=dev)
    cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()       # (torch fills the outputs on ITS stream: finished before the library's stream writes into them)
    g.merge_partials_device(nq, K, nparts, keys_all.data_ptr(), cnts_all.data_ptr(), ids.data_ptr(), dist.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    got = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dist.cpu().numpy().reshape(nq, K), cnt.cpu().numpy())
    helpers.assert_same_results(got, exp, what="two-level + list partition")


def test_two_python_threads_knn_search_on_one_index(native):
    """knn_search packs into, and reads out of, page-locked blocks that belong to the index, and ctypes releases the GIL
    during the native call.  Two Python threads calling knn_search on the SAME index -- different batch sizes, so the blocks are
    regrown while the other thread's call may be running -- must each get the oracle's answers (the binding serialises the packing,
    the call and the copy-out per index)."""
    import threading
    oidx, _ = helpers.build_index(4711, 6000, 64, 24, 8, 256, mode="random")
    g = gpu_index(native, oidx)
    rng = np.random.default_rng(3)
    sets = [rng.random((n, 64), dtype=np.float32) for n in (7, 300, 41, 1200, 3, 650)]
    exp = [oidx.knn_search(q, 5, 3) for q in sets]
    errs = []

    def worker(order):
        try:
            for rep in range(6):
                for i in order:
                    ids, dists = native.knn_search(g, sets[i], 5, w=3)
                    ei, ed, ec = exp[i]
                    for r in range(sets[i].shape[0]):
                        c = int(ec[r])
                        assert np.array_equal(ids[r], ei[r, :c].astype(ids[r].dtype)) and np.array_equal(dists[r], ed[r, :c]), (i, r)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3, 4, 5], [5, 3, 1, 4, 2, 0], [3, 0, 5, 2])]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[0]
