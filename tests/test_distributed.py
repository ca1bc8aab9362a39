"""World-size-2 gloo test of the multi-GPU path on CPU: queries sharded across ranks, one gather of
the packed top-k.  The per-rank search is the oracle here (tests may use it); on the GPU box the same
sharded_knn_search wraps IVFADCIndex.search_raw."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, K, w, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import helpers
    import ivfadc_jl_amd as pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx, _ = helpers.build_index(77, 2000, 16, 20, 4, 64)          # identical replica on every rank
    qs = np.random.default_rng(5).random((nq, 16), dtype=np.float32)
    ids, dists, counts = pkg.distributed.sharded_knn_search(lambda q, k, ww: oidx.knn_search(q, k, ww), qs, K, w)
    ei, ed, ec = oidx.knn_search(qs, K, w)
    ok = np.array_equal(ids, ei) and np.array_equal(dists, ed) and np.array_equal(counts, ec)
    np.save(os.path.join(tmpdir, "ok%d.npy" % rank), np.array([int(ok)]))
    dist.destroy_process_group()


def test_sharded_search_world2(tmp_path):
    for nq in (11, 8):          # ragged (6 + 5) and even splits
        port = _free_port()
        mp.spawn(_worker, args=(2, port, nq, 5, 3, str(tmp_path)), nprocs=2, join=True)
        for r in range(2):
            assert np.load(os.path.join(str(tmp_path), "ok%d.npy" % r))[0] == 1


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    import ivfadc_jl_amd as pkg
    for nq in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [pkg.distributed.shard_bounds(nq, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nq
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
