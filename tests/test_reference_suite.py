"""The reference's own test files, replayed line by line against the HIP path, for both values of the reference's
`for coarse_quantizer in [:naive, :hnsw]` loops (an :hnsw request is answered by the exhaustive GPU search of the
centroids -- the thing the HNSW graph approximates).  /root/reference/test/{index,utils,search,persistency}.jl are the model for
what is asserted; nothing is read from /root/reference at run time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NVECTORS, NROWS = 243, 10        # test/index.jl:1-2


QUANTIZERS = ["naive", "hnsw"]


def build_index_random_data(native, index_type=np.uint32, seed=0, coarse_quantizer="naive"):
    """test/index.jl:5-29: rand(10, 243), kc=100, k=16, m=2, 25 iterations (here: the library's own trainer)."""
    data = np.random.default_rng(seed).random((NVECTORS, NROWS), dtype=np.float32)
    return native.IVFADCIndex(data, kc=100, k=16, m=2, coarse_quantizer=coarse_quantizer, coarse_maxiter=25,
                              quantization_maxiter=25, index_type=index_type, seed=seed), data


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_index_constructor(native, cq):
    """test/index.jl:32-42."""
    idx, _ = build_index_random_data(native, coarse_quantizer=cq)
    assert isinstance(idx, native.IVFADCIndex)
    data = np.random.default_rng(1).random((300, 2), dtype=np.float32)
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=1, k=2, m=1)          # kc fail
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=2, k=301, m=1)        # k fail
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=2, k=300, m=3)        # m fail
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, index_type=np.uint8)     # index_type fail: 300 points do not fit UInt8


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_utils_push_pushfirst(native, cq):
    """test/utils.jl:1-29."""
    idx, _ = build_index_random_data(native, index_type=np.uint8, coarse_quantizer=cq)
    rng = np.random.default_rng(2)
    ol = len(idx)
    nnv = 256 - NVECTORS
    for _ in range(nnv):
        native.push(idx, rng.random(NROWS, dtype=np.float32))
    assert len(idx) == ol + nnv
    with pytest.raises(AssertionError):
        native.push(idx, rng.random(NROWS, dtype=np.float32))          # index is full
    native.delete_from_index(idx, [1])
    with pytest.raises(AssertionError):
        native.push(idx, rng.random(NROWS + 1, dtype=np.float32))      # wrong dimension
    for i in range(1, nnv):                                             # pushfirst!
        native.delete_from_index(idx, [i])
    for _ in range(nnv):
        native.pushfirst(idx, rng.random(NROWS, dtype=np.float32))
    assert len(idx) == ol + nnv
    with pytest.raises(AssertionError):
        native.pushfirst(idx, rng.random(NROWS, dtype=np.float32))     # index is full
    native.delete_from_index(idx, [1])
    with pytest.raises(AssertionError):
        native.pushfirst(idx, rng.random(NROWS + 1, dtype=np.float32))  # wrong dimension
    # and the edited index still answers (every id once, 0..n-1)
    _, _, ids = idx._lists()
    assert sorted(ids.tolist()) == list(range(len(idx)))
    got, _ = native.knn_search(idx, rng.random(NROWS, dtype=np.float32), 3, w=100)
    assert len(got) == 3


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_utils_pop_popfirst(native, cq):
    """test/utils.jl:32-56."""
    idx, _ = build_index_random_data(native, index_type=np.uint8, coarse_quantizer=cq)
    ol = len(idx)
    v = native.pop(idx)
    assert isinstance(v, np.ndarray) and v.shape == (idx.size[0],)
    assert len(idx) == ol - 1
    ol = len(idx)
    v = native.popfirst(idx)
    assert isinstance(v, np.ndarray) and v.shape == (idx.size[0],)
    assert len(idx) == ol - 1


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_utils_delete_from_index(native, cq):
    """test/utils.jl:59-106: the deleted points are gone, every other point keeps its code and gets its shifted id."""
    idx, _ = build_index_random_data(native, coarse_quantizer=cq)
    before = [(lst.idxs.copy(), [c.copy() for c in lst.codes]) for lst in idx.inverse_index]
    n = len(idx)
    L1s, L1e, L2s, L2e, L3s, L3e = 1, 5, 10, 30, n - 5, n
    to_delete = list(range(L1s, L1e + 1)) + list(range(L2s, L2e + 1)) + list(range(L3s, L3e + 1))
    native.delete_from_index(idx, to_delete)
    assert len(idx) == n - len(to_delete)
    after = idx.inverse_index
    mismatches = 0
    for cl, (cluster_indexes, codes) in enumerate(before):
        cluster_indexes_del = after[cl].idxs
        found = np.intersect1d(cluster_indexes, np.array(to_delete) - 1)
        assert len(cluster_indexes) == len(cluster_indexes_del) + len(found)
        for i, idx1 in enumerate(cluster_indexes.astype(np.int64) + 1):
            if L1e < idx1 < L2s:
                shift = L1e - L1s + 1
            elif L2e < idx1 < L3s:
                shift = (L1e - L1s + 1) + (L2e - L2s + 1)
            else:
                shift = None
            if shift is not None:
                newval = idx1 - shift - 1
                newpos = int(np.nonzero(cluster_indexes_del == newval)[0][0])
                if not np.array_equal(codes[i], after[cl].codes[newpos]):
                    mismatches += 1
    assert mismatches == 0


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_search_types_methods(native, cq):
    """test/search.jl:1-24."""
    idx, _ = build_index_random_data(native, index_type=np.uint32, coarse_quantizer=cq)
    rng = np.random.default_rng(3)
    K = 3
    query = rng.random(NROWS, dtype=np.float32)
    idxs, dists = native.knn_search(idx, query, K, w=2)
    assert idxs.dtype == np.uint32 and idxs.ndim == 1 and dists.dtype == query.dtype and dists.ndim == 1
    with pytest.raises(AssertionError):
        native.knn_search(idx, query, 0)
    with pytest.raises(AssertionError):
        native.knn_search(idx, query, 1, w=0)
    queries = [rng.random(NROWS, dtype=np.float32) for _ in range(10)]
    idxs, dists = native.knn_search(idx, queries, K, w=2)
    assert isinstance(idxs, list) and all(a.dtype == np.uint32 and a.ndim == 1 for a in idxs)
    assert isinstance(dists, list) and all(a.dtype == np.float32 and a.ndim == 1 for a in dists)


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_search_results(native, cq):
    """test/search.jl:27-49 (set-level known answers; tolerant of the trainer's randomness, as in the reference)."""
    data = np.array([[0, 0, 0, 1, 1, 1, 1, 1, 20, 20, 20, 20, 20],
                     [0.1, 0.11, 0.12, 8, 10, 15, 14, 16, 5, 5.1, 5.2, 5.4, 5.5]], np.float32).T
    idx = native.IVFADCIndex(data, kc=3, k=8, m=2, coarse_quantizer=cq)
    points = [np.array(p, np.float32) for p in ([1.0, 10.0], [0.0, 0.0], [20.0, 5.0])]
    neighbors_w1 = [[5, 4, 7, 6, 8], [1, 2, 3], [9, 10, 11, 12, 13]]
    for point, result in zip(points, neighbors_w1):
        neighbors = native.knn_search(idx, point, 5, w=1)[0].astype(np.int64) + 1
        assert set(neighbors.tolist()) <= set(result)
    neighbors_w2 = [[5, 4, 7, 6, 8], [1, 2, 3, 4, 5], [9, 10, 11, 12, 13]]
    for point, result in zip(points, neighbors_w2):
        neighbors = native.knn_search(idx, point, 5, w=2)[0].astype(np.int64) + 1
        assert set(neighbors.tolist()) <= set(result)


@pytest.mark.parametrize("cq", QUANTIZERS)
def test_persistency_roundtrip(tmp_path, native, cq):
    """test/persistency.jl: save, load, every field equal."""
    idx, _ = build_index_random_data(native, index_type=np.uint16, coarse_quantizer=cq)
    path = os.path.join(str(tmp_path), "ivfadc.bin")
    native.save_ivfadc_index(path, idx)
    idx2 = native.load_ivfadc_index(path)
    assert idx2.index_type == idx.index_type and len(idx2) == len(idx) and idx2.size == idx.size
    assert np.array_equal(idx2._centroids, idx._centroids) and np.array_equal(idx2._codebooks, idx._codebooks)
    assert np.array_equal(idx2._labels, idx._labels)
    for a, b in zip(idx.inverse_index, idx2.inverse_index):
        assert np.array_equal(a.idxs, b.idxs) and all(np.array_equal(x, y) for x, y in zip(a.codes, b.codes))
