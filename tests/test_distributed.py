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


def _worker_lists(rank, world, port, nq, K, w, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import helpers
    import ivfadc_jl_amd as pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx, _ = helpers.build_index(78, 3000, 16, 23, 4, 64, mode="random", ndistinct=40)   # identical replica on every rank; ties across lists
    qs = np.random.default_rng(6).random((nq, 16), dtype=np.float32)
    ids, dists, counts = pkg.distributed.list_partitioned_knn_search(
        lambda q, k, ww, nparts, part: helpers.numpy_partial_keys(oidx, q, k, ww, nparts, part), qs, K, w)
    ei, ed, ec = oidx.knn_search(qs, K, w)
    ok = np.array_equal(counts, ec) and all(np.array_equal(ids[r, :ec[r]], ei[r, :ec[r]]) and np.array_equal(dists[r, :ec[r]], ed[r, :ec[r]])
                                            for r in range(nq))
    np.save(os.path.join(tmpdir, "okl%d.npy" % rank), np.array([int(ok)]))
    dist.destroy_process_group()


def test_list_partitioned_search_world2(tmp_path):
    """Strong-scaling mode: every rank gets ALL queries and scans the probed lists l with l % world == rank; one all-gather of the
    partial (distance, visit order) keys, K-way merge.  The merged result must be the oracle's full search (ids and distance bits),
    including ties that straddle the ranks' lists and queries whose probes all fall to one rank (w = 1)."""
    for K, w in ((5, 6), (12, 1), (3, 23)):
        port = _free_port()
        mp.spawn(_worker_lists, args=(2, port, 9, K, w, str(tmp_path)), nprocs=2, join=True)
        for r in range(2):
            assert np.load(os.path.join(str(tmp_path), "okl%d.npy" % r))[0] == 1


def test_merge_partial_topk_is_a_k_smallest_of_unique_keys():
    sys.path.insert(0, ROOT)
    import ivfadc_jl_amd as pkg
    rng = np.random.default_rng(3)
    nparts, nq, K = 5, 7, 6
    pool = rng.permutation(10 ** 6)[:nparts * nq * K].astype(np.uint64).reshape(nparts, nq, K)
    pool.sort(axis=2)
    counts = rng.integers(0, K + 1, (nparts, nq))
    mk, mc = pkg.distributed.merge_partial_topk(pool, counts, K)
    for q in range(nq):
        allv = np.sort(np.concatenate([pool[p, q, :counts[p, q]] for p in range(nparts)]))
        assert mc[q] == min(K, len(allv)) and np.array_equal(mk[q, :mc[q]], allv[:mc[q]])


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    import ivfadc_jl_amd as pkg
    for nq in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [pkg.distributed.shard_bounds(nq, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nq
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _run_bench(args, timeout=600):
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line: %r" % out.stdout[-500:]
    # the harness keeps a bounded tail of stdout: the line must be the LAST thing on it and stay well under that bound
    assert out.stdout.rstrip("\n").endswith(lines[0])
    assert len(lines[0]) < 4096, "bench.py's stdout line is %d bytes (limit 4096): detail belongs in gpurun_out/bench_full.json" % len(lines[0])
    return json.loads(lines[0])


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")


def test_bench_launcher_partition_and_gather_world2():
    """`python3 bench.py --gpus 2` with no torchrun around it must launch 2 ranks itself (a torch.distributed.run child,
    spawned before anything touches a GPU), partition ONE global batch into contiguous per-rank blocks and gather the packed
    results with ONE collective per batch.  On this GPU-less host the ranks run over gloo with the stub searcher
    (--selftest-cpu): bench.py's own launcher, shard_bounds, Rings and collective code, end to end."""
    for gather_every, steps in ((1, 6), (4, 6)):
        line = _run_bench(["--gpus", "2", "--selftest-cpu", "--steps", str(steps), "--warmup", "3", "--nq", "5",
                           "--gather-every", str(gather_every)])
        assert line["n_gpus"] == 2 and line["steps"] == steps
        assert all(k in line for k in CONTRACT_KEYS), sorted(line)
        assert "workload" in line["config"] and "model" not in line["config"]
        d = line["distributed"]
        assert d["ranks_seen_by_rccl"] == 2 and d["gather_check"] is True and d["partition_check"] is True
        assert d["batches_per_collective"] == gather_every
        # one collective per batch in the headline mode; ceil(steps / G) with batching
        assert d["collectives_in_timed_region"] == (steps + gather_every - 1) // gather_every
        assert line["config"]["global_batch"] == 10 and line["scaling"] == "weak"
    # strong scaling: the global batch is fixed (here 10 queries) and every rank takes half of it
    line = _run_bench(["--gpus", "2", "--selftest-cpu", "--steps", "5", "--warmup", "2", "--nq", "10", "--scaling", "strong"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 10 and line["config"]["queries_per_rank"] == 5
    d = line["distributed"]
    assert d["ranks_seen_by_rccl"] == 2 and d["gather_check"] is True and d["partition_check"] is True
    assert d["collectives_in_timed_region"] == 5


def test_bench_eight_ranks_strong_partition_of_the_sift1b_batch():
    """The first real 8-GPU run must not fail on launcher arithmetic: eight ranks (gloo, stub searcher), the SIFT1B configuration's global
    batch of 16 384 queries cut into eight contiguous blocks (--scaling strong), one collective per batch, the compact line."""
    line = _run_bench(["--gpus", "8", "--selftest-cpu", "--scaling", "strong", "--nq", "16384", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 16384 and line["config"]["queries_per_rank"] == 2048
    assert line["ranks_seen_by_rccl"] == 8 and line["gather_check"] is True
    d = line["distributed"]
    assert d["ranks_seen_by_rccl"] == 8 and d["gather_check"] is True and d["partition_check"] is True
    assert d["collectives_in_timed_region"] == 3 and d["batches_per_collective"] == 1
    # weak scaling at eight ranks: every rank gets the per-GPU batch
    line = _run_bench(["--gpus", "8", "--selftest-cpu", "--nq", "64", "--steps", "2", "--warmup", "1"])
    assert line["config"]["global_batch"] == 512 and line["distributed"]["partition_check"] is True


def test_bench_compact_line_of_a_full_record():
    """compact_line() on the largest record the repo holds (round 5's 22 KB line, profiles/r05_bench_line.json): under the limit, with the
    contract's keys, the roofline / cpu_baseline objects and the contract rates as scalars."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    c = bench.compact_line(full)
    text = json.dumps(c, separators=(",", ":"))
    assert len(text) <= bench.LINE_LIMIT < 4096, len(text)
    assert all(k in c for k in CONTRACT_KEYS)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "alg_bytes_per_launch", "scan_ms_per_launch"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    for k in ("plain_one_in_flight_qps", "host_blocking_qps", "host_batches_qps", "scaling_base_qps"):
        assert isinstance(c["rates"][k], float), k
    assert all(set(v) >= {"ms_per_step", "frac", "physical_hbm_frac", "parity"} for v in c["other_configs"].values())


def test_bench_strong_scaling_refuses_uneven_blocks():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu", "--steps", "2", "--warmup", "1",
                        "--nq", "7", "--scaling", "strong"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "multiple of the number of GPUs" in (r.stderr + r.stdout)


def test_bench_shard_bounds_match_library_rule():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import ivfadc_jl_amd as pkg
    for nq in (0, 1, 7, 1024, 16385):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert bench.shard_bounds(nq, world, r) == pkg.distributed.shard_bounds(nq, world, r)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_single_rank_rccl_and_single_process_front_end():
    """On the GPU box: (i) bench.py under a real RCCL process group (one rank, BENCH_FORCE_DIST=1) -- one all-gather per
    batch, gather_check, oracle parity; (ii) `--single-process`: ivfadc_mg_search with the library's own ncclAllGather."""
    import json
    import subprocess
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-sweep"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert len(out.stdout.strip()) < 4096
    assert line["distributed"]["ranks_seen_by_rccl"] == 1 and line["gather_check"] is True
    assert line["distributed"]["collectives_in_timed_region"] == 12
    assert line["parity"]["ids_bit_exact"] and line["parity"]["dists_rtol_1e-4"]
    line = _run_bench(["--single-process", "--gpus", "1", "--config", "sift1b", "--n", "20000000", "--nq", "2048", "--steps", "3",
                       "--warmup", "1"], timeout=900)
    assert line["n_gpus"] == 1 and line["collectives_in_timed_region"] == 3
    assert line["parity"]["ids_bit_exact"] and line["parity"]["dists_rtol_1e-4"]


@pytest.mark.gpu
def test_bench_driver_command_prints_a_compact_parsable_line():
    """The driver's own command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`): ONE stdout line under 4 KB carrying the contract's
    keys, a live roofline and CPU baseline, the contract rates and one tuple per other BASELINE shape; the full record in the side file."""
    import json
    line = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], timeout=900)
    assert all(k in line for k in CONTRACT_KEYS), sorted(line)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["value"] > 0
    rl, cb = line["roofline"], line["cpu_baseline"]
    assert rl["bound"] and rl["frac"] > 0 and rl["peak"] == 8000.0 and rl["scan_ms_per_launch"] > 0 and rl["alg_bytes_per_launch"] > 0
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port"
    assert line["parity"]["ids_bit_exact"] and line["parity"]["dists_rtol_1e-4"]
    for k in ("plain_one_in_flight_qps", "host_blocking_qps", "host_batches_qps", "scaling_base_qps"):
        assert line["rates"][k] and line["rates"][k] > 0, (k, line["rates"])
    oc = line["other_configs"]
    assert {"deep1b w=32", "hd w=8", "sift1b w=8", "sift1b w=1", "sift1b w=8 batch=2048"} <= set(oc), sorted(oc)
    assert all(v.get("parity") is True for v in oc.values()), oc
    full = json.load(open(os.path.join(ROOT, line["full_record"])))
    assert full["value"] == line["value"] and "host_to_host" in full
