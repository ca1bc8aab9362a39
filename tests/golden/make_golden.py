#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference holds no numeric golden vectors (its tests draw unseeded rand()) and cannot run in this
image, so these fixtures are produced by the CPU oracle (oracle/ivfadc_oracle.c) on seeded inputs; they
pin the oracle against regressions and give the HIP path fixed expected outputs.  Each .npz holds the
index arrays, the queries and the expected (ids, dists, counts).
  readme_toy   README.md:33-47 shape: 50 x 1000 Float32, kc=100, k=256, m=10, UInt16 ids, K=3, trained index
  search_jl    test/search.jl:27-32: the 2 x 13 hand-made data, kc=3, k=8, m=2, K=5, w=1 and w=2
  ties         3 distinct codes only: exact distance ties everywhere (order = distance, probe rank, position)
  few          fewer than K candidates: sparse lists, K=10
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
import torch_kmeans  # noqa: E402
import ivfadc_jl_amd as pkg  # noqa: E402
from oracle import oracle as ora  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, oidx, queries, cases):
    d = dict(centroids=oidx.centroids, codebooks=oidx.codebooks, labels=oidx.labels, offsets=oidx.offsets,
             codes=oidx.codes, ids=oidx.ids, queries=queries, cases=np.array([[K, w] for K, w in cases], np.int32))
    for K, w in cases:
        ids, dists, counts = oidx.knn_search(queries, K, w)
        d["ids_K%d_w%d" % (K, w)] = ids
        d["dists_K%d_w%d" % (K, w)] = dists
        d["counts_K%d_w%d" % (K, w)] = counts
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if k in ("codes", "queries")})


def trained(data, kc, k, m, seed):
    cent, cbs, labels = torch_kmeans.train_ivfadc(data, kc, k, m, seed=seed, device="cpu")
    tmp = ora.OracleIndex(cent, cbs, labels, np.zeros(kc + 1, np.int64), np.zeros((0, m), np.uint8), np.zeros(0, np.uint32))
    lst, codes = tmp.encode(data)
    order = np.argsort(lst, kind="stable")
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(np.bincount(lst, minlength=kc), out=offsets[1:])
    return ora.OracleIndex(cent, cbs, labels, offsets, codes[order], order.astype(np.uint32))


def main():
    rng = np.random.default_rng(1)
    data = rng.random((1000, 50), dtype=np.float32)
    oidx = trained(data, 100, 256, 10, seed=1)
    q = np.concatenate([data[[122, 5, 77]], rng.random((13, 50), dtype=np.float32)])
    save("readme_toy", oidx, q, [(3, 1), (3, 8), (10, 100)])

    sj = np.array([[0, 0, 0, 1, 1, 1, 1, 1, 20, 20, 20, 20, 20],
                   [0.1, 0.11, 0.12, 8, 10, 15, 14, 16, 5, 5.1, 5.2, 5.4, 5.5]], np.float32).T.copy()
    for seed in range(20):
        o2 = trained(sj, 3, 8, 2, seed=seed)
        if len({tuple(np.round(c, 3)) for c in o2.centroids}) == 3 and np.ptp(o2.centroids[:, 0]) > 15:
            break
    save("search_jl", o2, np.array([[1.0, 10.0], [0.0, 0.0], [20.0, 5.0]], np.float32), [(5, 1), (5, 2)])

    o3, _ = helpers.build_index(7, 3000, 16, 6, 8, 256, mode="random", ndistinct=3)
    save("ties", o3, np.random.default_rng(7).random((12, 16), dtype=np.float32), [(25, 3), (70, 6)])

    o4, _ = helpers.build_index(9, 60, 8, 40, 4, 16, mode="random")
    save("few", o4, np.random.default_rng(9).random((20, 8), dtype=np.float32), [(10, 1), (10, 5)])


if __name__ == "__main__":
    main()
