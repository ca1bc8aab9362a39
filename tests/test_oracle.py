"""CPU tests of the oracle (the checker itself): against an independent numpy restatement,
against the reference's only result-level test (test/search.jl:26-49) and its assertion
conventions (test/search.jl:11-21)."""
import numpy as np
import pytest

import helpers
import torch_kmeans
from oracle import oracle as ora


@pytest.mark.parametrize("seed,n,d,kc,m,ksub,K,w", [
    (1, 300, 10, 20, 2, 16, 3, 2),        # test/index.jl helper shape (10-dim, k=16, m=2)
    (2, 500, 16, 12, 4, 256, 10, 3),
    (3, 400, 50, 30, 10, 64, 5, 30),      # m=10 as in the README toy; w == kc
    (4, 200, 8, 5, 8, 8, 7, 99),          # dsub = 1, w clamped to kc
])
def test_oracle_matches_numpy_restatement(seed, n, d, kc, m, ksub, K, w):
    oidx, data = helpers.build_index(seed, n, d, kc, m, ksub, label_perm=(seed % 2 == 0))
    rng = np.random.default_rng(seed)
    qs = rng.random((6, d), dtype=np.float32)
    ids, dists, counts = oidx.knn_search(qs, K, w)
    for r in range(qs.shape[0]):
        ei, ed = helpers.numpy_knn(oidx, qs[r], K, w)
        assert counts[r] == len(ei)
        assert np.array_equal(ids[r, :counts[r]], ei)
        assert np.array_equal(dists[r, :counts[r]], ed)      # bit-exact: same float order


def test_oracle_ties_follow_visit_order():
    # only 3 distinct codes -> masses of exact ties; SortedMultiDict semantics == (dist, visit order)
    oidx, _ = helpers.build_index(7, 600, 8, 6, 4, 16, mode="random", ndistinct=3)
    rng = np.random.default_rng(7)
    qs = rng.random((8, 8), dtype=np.float32)
    ids, dists, counts = oidx.knn_search(qs, 20, 4)
    for r in range(qs.shape[0]):
        ei, ed = helpers.numpy_knn(oidx, qs[r], 20, 4)
        assert np.array_equal(ids[r, :counts[r]], ei) and np.array_equal(dists[r, :counts[r]], ed)
        assert len(np.unique(ed)) < len(ed)                  # ties really occurred


def test_oracle_fewer_than_k_and_empty_lists():
    oidx, _ = helpers.build_index(9, 12, 4, 8, 2, 16, mode="random")
    rng = np.random.default_rng(9)
    qs = rng.random((5, 4), dtype=np.float32)
    ids, dists, counts = oidx.knn_search(qs, 10, 1)
    assert (counts <= 10).all() and (counts < 10).any()
    for r in range(5):
        assert np.all(np.diff(dists[r, :counts[r]]) >= 0)


def test_oracle_assertions():
    # test/search.jl:14-15: k = 0 and w = 0 raise AssertionError
    oidx, _ = helpers.build_index(11, 50, 4, 4, 2, 16)
    q = np.zeros((1, 4), np.float32)
    with pytest.raises(AssertionError):
        oidx.knn_search(q, 0, 1)
    with pytest.raises(AssertionError):
        oidx.knn_search(q, 1, 0)


def test_coarse_search_is_stable():
    cent = np.zeros((6, 3), np.float32)
    cent[[1, 4]] = 1.0                                      # clusters {0,2,3,5} tie at distance 0
    oidx = ora.OracleIndex(cent, np.zeros((1, 2, 3), np.float32), np.array([[0, 1]], np.uint8),
                           np.zeros(7, np.int64), np.zeros((0, 1), np.uint8), np.zeros(0, np.uint32))
    cl, dist = oidx.coarse_search(np.zeros(3, np.float32), 5)
    assert cl.tolist() == [0, 2, 3, 5, 1] and dist.tolist() == [0, 0, 0, 0, 3]


def _search_jl_index():
    """test/search.jl:27-30: 2 x 13 hand-made data in three obvious clusters, kc=3, k=8, m=2,
    trained with the build's own trainer (the reference's kmeans is unseeded)."""
    import ivfadc_jl_amd as pkg
    data = np.array([[0, 0, 0, 1, 1, 1, 1, 1, 20, 20, 20, 20, 20],
                     [0.1, 0.11, 0.12, 8, 10, 15, 14, 16, 5, 5.1, 5.2, 5.4, 5.5]], np.float32).T.copy()
    for seed in range(20):                                   # kmeans++ may merge clusters; the reference test
        cent, cbs, labels = torch_kmeans.train_ivfadc(data, 3, 8, 2, seed=seed, device="cpu")   # tolerates that too
        if len({tuple(np.round(c, 3)) for c in cent}) == 3 and np.ptp(cent[:, 0]) > 15:
            break
    tmp = ora.OracleIndex(cent, cbs, labels, np.zeros(4, np.int64), np.zeros((0, 2), np.uint8), np.zeros(0, np.uint32))
    lst, codes = tmp.encode(data)
    order = np.argsort(lst, kind="stable")
    offsets = np.zeros(4, np.int64)
    np.cumsum(np.bincount(lst, minlength=3), out=offsets[1:])
    return ora.OracleIndex(cent, cbs, labels, offsets, codes[order], order.astype(np.uint32)), data


def test_reference_known_answers():
    """test/search.jl:26-49: returned 1-based ids are a subset of the expected sets for w=1 and w=2."""
    oidx, _ = _search_jl_index()
    points = np.array([[1.0, 10.0], [0.0, 0.0], [20.0, 5.0]], np.float32)
    exp_w1 = [{5, 4, 7, 6, 8}, {1, 2, 3}, {9, 10, 11, 12, 13}]
    exp_w2 = [{5, 4, 7, 6, 8}, {1, 2, 3, 4, 5}, {9, 10, 11, 12, 13}]
    for w, exp in ((1, exp_w1), (2, exp_w2)):
        ids, dists, counts = oidx.knn_search(points, 5, w)
        for r in range(3):
            got = set((ids[r, :counts[r]].astype(int) + 1).tolist())
            assert got and got <= exp[r], (w, r, got, exp[r])
    ids, _, counts = oidx.knn_search(points[1:2], 5, 1)
    assert counts[0] == 3                                    # "at most k": the cell holds 3 points


def test_encode_matches_numpy_argmin():
    oidx, data = helpers.build_index(13, 100, 12, 7, 3, 32, label_perm=True)
    lst, codes = oidx.encode(data[:20])
    f32 = np.float32
    for p in range(20):
        acc = np.zeros(oidx.kc, f32)
        for i in range(oidx.d):
            t = oidx.centroids[:, i] - data[p, i]
            acc = acc + t * t
        cl = int(np.argmin(acc))
        assert lst[p] == cl
        r = data[p] - oidx.centroids[cl]
        for i in range(oidx.m):
            s = np.zeros(oidx.ksub, f32)
            for t_ in range(oidx.dsub):
                df = oidx.codebooks[i, :, t_] - r[i * oidx.dsub + t_]
                s = s + df * df
            assert codes[p, i] == oidx.labels[i, int(np.argmin(s))]


def test_synth_codes_are_reproducible():
    a = ora.synth_fill(42, 1000, 64, 8)
    b = ora.synth_fill(42, 1000 + 16, 16, 8)
    assert np.array_equal(a[16:32], b)
    assert len(np.unique(a)) > 100
    c = ora.synth_fill(42, 5, 40, 10)                        # m not a multiple of 8: bytes straddle hash words
    d = ora.synth_fill(42, 6, 10, 10)
    assert np.array_equal(c[1:11], d)
