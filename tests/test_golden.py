"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must match them (GPU)."""
import glob
import os

import numpy as np
import pytest

import helpers
from oracle import oracle as ora

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    oidx = ora.OracleIndex(z["centroids"], z["codebooks"], z["labels"], z["offsets"], z["codes"], z["ids"])
    return z, oidx


def test_fixtures_exist():
    assert {os.path.basename(p) for p in GOLDEN} >= {"readme_toy.npz", "search_jl.npz", "ties.npz", "few.npz"}


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z, oidx = load(path)
    for K, w in z["cases"]:
        ids, dists, counts = oidx.knn_search(z["queries"], int(K), int(w))
        assert np.array_equal(counts, z["counts_K%d_w%d" % (K, w)])
        assert np.array_equal(ids, z["ids_K%d_w%d" % (K, w)])
        assert np.array_equal(dists, z["dists_K%d_w%d" % (K, w)])


def test_search_jl_golden_satisfies_reference_sets():
    """the committed expected ids of the test/search.jl case lie in the reference's expected sets (:34-47)"""
    z, _ = load([p for p in GOLDEN if p.endswith("search_jl.npz")][0])
    exp = {1: [{5, 4, 7, 6, 8}, {1, 2, 3}, {9, 10, 11, 12, 13}], 2: [{5, 4, 7, 6, 8}, {1, 2, 3, 4, 5}, {9, 10, 11, 12, 13}]}
    for w in (1, 2):
        ids, counts = z["ids_K5_w%d" % w], z["counts_K5_w%d" % w]
        for r in range(3):
            assert set((ids[r, :counts[r]].astype(int) + 1).tolist()) <= exp[w][r]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [-1, 4])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_matches_golden(native, path, mode):
    z, oidx = load(path)
    gidx = native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids)
    gidx.set_tuning(mode, 0)
    for K, w in z["cases"]:
        got = gidx.search_raw(z["queries"], int(K), int(w))
        exp = (z["ids_K%d_w%d" % (K, w)], z["dists_K%d_w%d" % (K, w)], z["counts_K%d_w%d" % (K, w)])
        helpers.assert_same_results(got, exp, what="%s K=%d w=%d" % (os.path.basename(path), K, w))
