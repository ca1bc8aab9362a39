#!/usr/bin/env python3
"""bench.py -- queries/sec of batch knn_search (K=10) on the BASELINE.json shapes.

One "step" = one pass of the hot path (coarse search -> residuals -> ADC tables -> list scan ->
top-k) over one batch of queries already resident in HBM.  Default workload = BASELINE.json
configs[1] (SIFT1M-shape: d=128, n=1e6, kc=1024, k=256, m=8, batch=1024).  With --gpus N the
driver launches one rank per GPU (torch.distributed / RCCL); every rank holds a full index replica
and its own batch (weak scaling), and the only collective is the gather of the packed top-k.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290

CONFIGS = {
    # name: d, n, kc, m, nq, w, kind
    "toy": dict(d=50, n=1000, kc=100, m=10, nq=64, w=1, kind="trained"),
    "sift1m": dict(d=128, n=1_000_000, kc=1024, m=8, nq=1024, w=8, kind="trained"),
    "deep1b": dict(d=96, n=100_000_000, kc=65536, m=16, nq=10000, w=32, kind="synth"),
    "sift1b": dict(d=128, n=1_000_000_000, kc=8192, m=8, nq=16384, w=8, kind="synth"),
    "hd": dict(d=768, n=10_000_000, kc=4096, m=48, nq=4096, w=8, kind="synth"),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def mixture(n, d, ncent, sigma, seed_c, seed_x, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed_c)
    cent = torch.rand((ncent, d), generator=g, device=dev)
    g.manual_seed(seed_x)
    which = torch.randint(0, ncent, (n,), generator=g, device=dev)
    x = cent[which] + sigma * torch.randn((n, d), generator=g, device=dev)
    return x.contiguous()


def build_trained(pkg, cfg, dev, device_index, rank=0):
    """SIFT1M-shape: Gaussian-mixture data (seed 1234), index trained by the build's own trainer,
    data encoded through the HIP push!/encode path."""
    d, n, kc, m = cfg["d"], cfg["n"], cfg["kc"], cfg["m"]
    t0 = time.time()
    x = mixture(n, d, 1024, 0.1, 99, 1234, dev)
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
        # rank 0 trains, every rank gets the same quantizers (the trainer is deterministic per seed anyway), and the
        # deterministic HIP encode then builds identical replicas
        ct = torch.empty((kc, d), dtype=torch.float32, device=dev)
        bt = torch.empty((m, 256, d // m), dtype=torch.float32, device=dev)
        if tdist.get_rank() == 0:
            cent, cbs, labels = _train(pkg, x, kc, m)
            ct.copy_(torch.as_tensor(cent))
            bt.copy_(torch.as_tensor(cbs))
        tdist.broadcast(ct, 0)
        tdist.broadcast(bt, 0)
        cent, cbs = ct.cpu().numpy(), bt.cpu().numpy()
        labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    else:
        cent, cbs, labels = _train(pkg, x, kc, m)
    idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=device_index)
    xh = x.cpu().numpy()
    idx._append(xh, np.arange(n, dtype=np.uint32))
    q = mixture(cfg["nq"], d, 1024, 0.1, 99, 4321 + rank, dev)   # each rank owns a different batch
    log("[bench] index built in %.1fs: %r" % (time.time() - t0, idx))
    return idx, x, q


def _train(pkg, x, kc, m):
    """the library's own trainer (ivfadc_train: HIP k-means++ / Lloyd, 25 iterations, deterministic per seed)"""
    return pkg.trainer.train_ivfadc_hip(x.cpu().numpy(), kc, 256, m, 25, 25, seed=7, device=x.device.index or 0)


def synth_sizes(n, kc, seed, skew=False):
    rng = np.random.default_rng(seed)
    p = np.full(kc, 1.0 / kc)
    if skew:
        p = rng.dirichlet(np.full(kc, 0.5))
    sizes = rng.multinomial(n, p).astype(np.int64)
    off = np.zeros(kc + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    return off


def build_synth(pkg, cfg, dev, device_index, skew=False, rank=0):
    """Billion-scale shapes: quantizers ~N(0,1) (seed 7), code bytes synthesised on the device by the
    counter-based RNG the oracle can replay, ids = position, queries ~N(0,1) (seed 11)."""
    d, n, kc, m = cfg["d"], cfg["n"], cfg["kc"], cfg["m"]
    rng = np.random.default_rng(7)
    cent = rng.standard_normal((kc, d), dtype=np.float32)
    cbs = rng.standard_normal((m, 256, d // m), dtype=np.float32)
    labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=device_index)
    off = synth_sizes(n, kc, 7, skew)
    t0 = time.time()
    idx.synth_lists(off, 20260101)
    log("[bench] synthetic lists on device in %.1fs: %r" % (time.time() - t0, idx))
    q = torch.as_tensor(np.random.default_rng(11 + rank).standard_normal((cfg["nq"], d), dtype=np.float32)).to(dev)
    return idx, None, q, (cent, cbs, labels, off)


def recall_at_1(x, q, ids, counts):
    """fraction of queries whose exact L2 nearest neighbour is among the returned ids."""
    best = torch.empty(q.shape[0], dtype=torch.int64, device=q.device)
    bd = torch.full((q.shape[0],), float("inf"), device=q.device)
    qn = (q * q).sum(1, keepdim=True)
    for s in range(0, x.shape[0], 262144):
        xb = x[s:s + 262144]
        dist = qn - 2.0 * (q @ xb.t()) + (xb * xb).sum(1)[None, :]
        md, a = dist.min(1)
        upd = md < bd
        bd = torch.where(upd, md, bd)
        best = torch.where(upd, a + s, best)
    ids = ids.to(torch.int64)
    K = ids.shape[1]
    valid = torch.arange(K, device=ids.device)[None, :] < counts[:, None]
    hit = ((ids == best[:, None]) & valid).any(1)
    return float(hit.float().mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--gather-every", type=int, default=8, help="multi-GPU: batches per all-gather")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="sift1m", choices=sorted(CONFIGS))
    ap.add_argument("--nq", type=int, default=0)
    ap.add_argument("--n", type=int, default=0, help="override the number of indexed vectors (synthetic configs)")
    ap.add_argument("--kc", type=int, default=0, help="override the number of coarse cells (synthetic configs)")
    ap.add_argument("--w", type=int, default=0)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--qg", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--skew", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--check", type=int, default=0, help="verify this many sampled queries against the oracle")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ivfadc_jl_amd as pkg
    if pkg.needs_build():
        pkg.build_library()
    pkg.load_library()

    cfg = dict(CONFIGS[args.config])
    if args.nq:
        cfg["nq"] = args.nq
    if args.n:
        cfg["n"] = args.n
    if args.kc:
        cfg["kc"] = args.kc
    if args.w:
        cfg["w"] = args.w
    K, w, nq = args.K, cfg["w"], cfg["nq"]
    synth_arrays = None
    if cfg["kind"] == "trained":
        idx, x, q = build_trained(pkg, cfg, dev, local_rank, rank)
    else:
        idx, x, q, synth_arrays = build_synth(pkg, cfg, dev, local_rank, args.skew, rank)
    idx.set_tuning(args.qg, args.chunk)
    stream = torch.cuda.current_stream()
    idx.set_stream(stream.cuda_stream)

    width = 2 * K + 1
    use_dist = dist is not None
    # Results land in rings of G batch slots.  Multi-GPU: ONE all-gather per G batches (two rings, double-buffered):
    # the collective of ring r runs on a side stream while ring 1-r is being filled, and its fixed host + launch cost
    # (~20-30 us, comparable to a whole SIFT1M-shape batch) is paid once per G batches -- fewer, larger collectives,
    # as xGMI wants them.  Single GPU: G = 2 plain double buffering, no collective.
    G = max(1, args.gather_every) if use_dist else 2
    NR = 2 if use_dist else 1
    ring = [torch.zeros(G * nq * width, dtype=torch.int32, device=dev) for _ in range(NR)]
    gath = [torch.zeros(world * G * nq * width, dtype=torch.int32, device=dev) for _ in range(NR)] if use_dist else None

    def slot_of(i):
        return (i // G) % NR, i % G

    def slot_view(i):
        r, sl = slot_of(i)
        return ring[r][sl * nq * width:(sl + 1) * nq * width]

    def ptrs(buf):
        base = buf.data_ptr()
        return base, base + nq * K * 4, base + 2 * nq * K * 4

    # Plain (non-async_op) collectives: their host cost is 12 us vs 28 us for the Work-object form (tools/ag_micro.py).
    side = [torch.cuda.Stream(device=dev) for _ in range(2)] if use_dist else None
    busy = [False, False]
    filled = [0, 0]

    def flush(r):
        side[r].wait_stream(stream)
        with torch.cuda.stream(side[r]):
            dist.all_gather_into_tensor(gath[r], ring[r])
        busy[r] = True
        filled[r] = 0

    def step(i):
        r, sl = slot_of(i)
        if use_dist and sl == 0 and busy[r]:
            stream.wait_stream(side[r])      # the ring's previous gather must have read it
            busy[r] = False
        p_ids, p_d, p_c = ptrs(slot_view(i))
        idx.search_device(nq, q.data_ptr(), K, w, p_ids, p_d, p_c)
        if use_dist:
            filled[r] = sl + 1
            if sl == G - 1:
                flush(r)

    def drain():
        if use_dist:
            for r in (0, 1):
                if filled[r]:
                    flush(r)                 # a partly filled ring is gathered whole
            for r in (0, 1):
                if busy[r]:
                    stream.wait_stream(side[r])
                    busy[r] = False

    def timed(nsteps):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    # Untimed settling: the W warm-up steps of a fast configuration last well under a millisecond, far too short for the
    # GPU to reach its sustained clock (SIFT1M-shape: 72 us per step in a cold 100-step run, 64 us once warm).  Keep
    # issuing untimed steps until ~0.1 s of them have run; the timed region below is still exactly K steps.
    t_settle = time.perf_counter()
    i = args.warmup
    while True:
        for _ in range(16):
            step(i)
            i += 1
        drain()
        torch.cuda.synchronize()
        done = time.perf_counter() - t_settle >= 0.1
        if use_dist:
            # every rank must issue the same number of steps (they carry collectives): all stop together
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break

    elapsed = timed(args.steps)
    qps = world * nq * args.steps / elapsed

    # ---- roofline of the dominant kernel (list scan): HIP events on the launch stream, live
    idx.set_profiling(True)
    idx.reset_stats()
    prof_steps = max(1, min(args.steps, 50))
    timed(prof_steps)
    st = idx.get_stats()
    idx.set_profiling(False)
    launches = max(1, st["scan_launches"])
    balg_per_launch = st["scanned_points"] * cfg["m"] / launches
    scan_ms = st["scan_ms"] / launches
    achieved = balg_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    kname = ("scan_kernel<M=%d,QG=%d> (list-major)" % (cfg["m"], st["last_qg"])) if st["last_qg"] > 0 else \
        ("qscan_kernel<M=%d> (query-major)" % cfg["m"])
    roofline = {"bound": "hbm", "kernel": kname,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_launch": int(balg_per_launch), "scan_ms_per_launch": round(scan_ms, 5),
                "coarse_ms_per_launch": round(st["coarse_ms"] / launches, 5),
                "chunk_points": st["last_chunk"], "scan_grid": st["last_scan_grid"], "scan_lds_bytes": st["last_scan_lds"]}

    # ---- results of the last step: recall (trained configs) and oracle spot-check
    last_i = prof_steps - 1
    res = slot_view(last_i)
    ids = res[:nq * K].view(nq, K)
    dists = res[nq * K:2 * nq * K].view(torch.float32).view(nq, K)
    counts = res[2 * nq * K:]
    recall = recall_at_1(x, q, ids, counts) if x is not None else None

    sweep = None
    if rank == 0 and world == 1 and not use_dist and args.config == "sift1m" and not args.w and not args.no_sweep:
        # the reference default is w=1; BASELINE.md asks for w in {1, 8, 32}
        sweep = {}
        for ws in (1, 8, 32):
            def step_w(i, ws=ws):
                p_ids, p_d, p_c = ptrs(slot_view(i & 1))
                idx.search_device(nq, q.data_ptr(), K, ws, p_ids, p_d, p_c)
            nsw = max(100, min(args.steps, 1000))
            for i in range(20):
                step_w(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(nsw):
                step_w(i)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            r = slot_view(1)
            rec = recall_at_1(x, q, r[:nq * K].view(nq, K), r[2 * nq * K:]) if x is not None else None
            sweep["w=%d" % ws] = {"qps": round(nq * nsw / el, 1), "recall_at_1_in_top%d" % K: rec}
        # For information only (never `value`): two replicas of the index on two caller streams, batches alternating
        # between them, so the tail of one batch's scan overlaps the next batch's coarse search.
        if synth_arrays is None:
            off_, codes_, ids_ = idx._lists()
            idx2 = pkg.IVFADCIndex.from_arrays(idx._centroids, idx._codebooks, idx._labels, off_, codes_, ids_, device=local_rank)
            s2 = torch.cuda.Stream(device=dev)
            idx2.set_stream(s2.cuda_stream)

            def step2(i):
                p_ids, p_d, p_c = ptrs(slot_view(i & 1))
                (idx2 if (i & 1) else idx).search_device(nq, q.data_ptr(), K, w, p_ids, p_d, p_c)
            for i in range(40):
                step2(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(nsw):
                step2(i)
            torch.cuda.synchronize()
            sweep["w=%d, two replicas on two streams" % w] = {"qps": round(nq * nsw / (time.perf_counter() - t0), 1)}
            del idx2
        # leave the buffers holding the headline-w results for the checks below
        for i in range(prof_steps):
            step(i)
        drain()
        torch.cuda.synchronize()

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as ora
        if synth_arrays is None:
            offsets, codes, lids = idx._lists()
            oidx = ora.OracleIndex(idx._centroids, idx._codebooks, idx._labels, offsets, codes, lids)
        else:
            cent, cbs, labels, off = synth_arrays
            oidx = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=20260101)
        qh = q.cpu().numpy()
        cores = ora.max_threads()
        # bounded sample: grow it until the all-cores run takes a few seconds
        ns = min(nq, 64)
        t0 = time.perf_counter()
        oi, od, oc = oidx.knn_search(qh[:ns], K, w, nthreads=cores)
        t_mt = time.perf_counter() - t0
        if t_mt < 2.0 and ns < nq:
            ns = int(min(nq, max(ns, ns * 4.0 / max(t_mt, 1e-3))))
            t0 = time.perf_counter()
            oi, od, oc = oidx.knn_search(qh[:ns], K, w, nthreads=cores)
            t_mt = time.perf_counter() - t0
        n1 = max(1, min(ns, int(ns * 3.0 / max(t_mt * cores, 1e-3))))
        t0 = time.perf_counter()
        oidx.knn_search(qh[:n1], K, w, nthreads=1)
        t_1 = time.perf_counter() - t0
        cpu_baseline = {"value": round(ns / t_mt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                        "sample": "first %d queries of the batch, same index arrays, oracle/ivfadc_oracle.c "
                                  "(gcc -O2 -ffp-contract=off), OpenMP over queries" % ns,
                        "single_thread_qps": round(n1 / t_1, 2), "single_thread_sample": n1}
        gi = ids[:ns].cpu().numpy().view(np.uint32)
        gd = dists[:ns].cpu().numpy()
        gc = counts[:ns].cpu().numpy()
        ok_ids = bool(np.array_equal(gc, oc) and all(np.array_equal(gi[r, :gc[r]], oi[r, :oc[r]]) for r in range(ns)))
        ok_d = bool(all(np.allclose(gd[r, :gc[r]], od[r, :oc[r]], rtol=1e-4, atol=0) for r in range(ns)))
        parity = {"queries_checked": ns, "ids_bit_exact": ok_ids, "dists_rtol_1e-4": ok_d}

    if rank == 0:
        line = {
            "metric": "queries/sec at recall@1 (k=10), SIFT1M-shape d=128 m=8 k=256, 1/2/4/8 GPU"
                      if args.config == "sift1m" else "queries/sec, %s-shape, K=%d" % (args.config, K),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape: d=%d n=%d kc=%d k=256 m=%d UInt8 codes, batch=%d queries/GPU, K=%d, w=%d%s"
                                   % (args.config, cfg["d"], cfg["n"], cfg["kc"], cfg["m"], nq, K, w,
                                      ", skewed lists" if args.skew else ""),
                       "index": "trained (k-means + PQ, 25 iters), Gaussian-mixture data" if cfg["kind"] == "trained"
                                else "device-synthesised codes, N(0,1) quantizers",
                       "parallelism": "queries sharded over %d GPU(s), index replicated, 1 all-gather of packed top-k per batch" % world
                                      if world > 1 else "1 GPU",
                       "recall_at_1_in_top%d" % K: recall},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity, "sweep": sweep,
        }
        if use_dist:
            # the gathered block of this rank must equal its local results (ring of the last timed step)
            rr, _ = slot_of(prof_steps - 1)
            blk = G * nq * width
            line["gather_check"] = bool(torch.equal(gath[rr][rank * blk:(rank + 1) * blk], ring[rr]))
            line["config"]["collective"] = "1 all_gather_into_tensor per %d batches (%d B per rank), side stream" % (G, blk * 4)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
