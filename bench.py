#!/usr/bin/env python3
"""bench.py -- queries/sec of batch knn_search (K=10) on the BASELINE.json shapes.

One "step" = one pass of the hot path (coarse search -> residuals -> ADC tables -> list scan ->
top-k) over one batch of queries already resident in HBM.  Default workload = BASELINE.json
configs[1] (SIFT1M-shape: d=128, n=1e6, kc=1024, k=256, m=8, batch=1024 per GPU).

Multi-GPU (`--gpus N`, N > 1): one process per GPU.  Under the driver's torchrun the ranks come from
RANK / LOCAL_RANK / WORLD_SIZE; started plainly (`python3 bench.py --gpus N`) this process spawns the N
ranks itself -- as a `torch.distributed.run` child, BEFORE anything touches the GPU -- and exits with the
child's code.  Every rank holds a full index replica; a step is ONE global batch of
N x nq queries partitioned into contiguous blocks, one block per rank (queries are independent,
/root/reference/src/index.jl:269-271), and ONE collective per batch: the all-gather of the packed
[ids | dists | counts] top-k block over RCCL/xGMI (`--gather-every G` batches up G batches per
collective: a labelled option, not the headline).  Default: per-GPU work is fixed as N grows ("scaling": "weak").
`--scaling strong` fixes the GLOBAL batch at the configuration's batch (sift1b: 16 384 queries) and gives every rank
1/N of it -- the north-star's "QPS at 8 GPUs vs 1 GPU on the SIFT1B shape" as posed; `--nq` then overrides the global batch.
`--single-process` drives the C ABI's own multi-device front end instead (ivfadc_mg_search, host
pointers, optional in-library ncclAllGather).

Prints ONE JSON line on rank 0.  `value` is the MEDIAN of `--windows` (default 5) timed windows of exactly `--steps` steps each (every
window bracketed by barrier + synchronize on both sides, MAX over ranks); min / max of the windows are in `windows`.  The default
single-GPU run also measures the other BASELINE.json shapes (Deep1B, HD, SIFT1B w = 8 and w = 1) briefly -- step time, scan-kernel
time, roofline fractions and a 64-query oracle parity bit each -- and reports them under `other_configs`.

Batches in flight (`--inflight`, default: by shape): on the shapes whose launch is only a few workgroups per CU (sift1m, hd, toy) the steps
alternate between the index and a read-only VIEW of it (ivfadc_clone_view: the same device arrays, a second stream and workspace), so that a
launch's ramp and tail overlap the neighbouring batches' kernels; every batch is still searched whole, results bit-identical.  `value` is
that rate; `batches_in_flight` carries the one-at-a-time rates of the same run (with the next-batch hint, and as plain knn_search-per-batch
calls), and everything profiled (roofline, pruning off, sweep, other_configs' primary figures) runs one batch at a time.

`--single-mode` runs ONE kernel population only (no same-run comparison legs, no sweep, no other configs): the form the rocprofv3
passes of tools/profile_all.sh are taken on, once per mode (hinted / `--no-next-hint` / `--no-next-hint --no-pruning`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12     # plain f32 vector operations (one per lane and cycle of a SIMD16): 39.3 T/s; FMA would count two flops each
NOMINAL_CLOCK_HZ = 2.4e9       # max shader clock (MI355X_MICROARCH.md chip table); the sustained clock is lower
NUM_CU = 256

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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_bounds(nq_total, world, rank):
    """Contiguous query block of `rank` in a global batch of nq_total queries (sizes differ by at most one): the same
    rule as ivfadc.jl_amd/distributed.py and ivfadc_mg_search."""
    base, rem = divmod(int(nq_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python3 bench.py --gpus N` with no torchrun around it
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, cpu_selftest):
    """Spawn N ranks as a torch.distributed.run child and exit with its status.  Nothing in this process has touched the
    GPU (torch.cuda.device_count() does not initialise it on this image), and the child is a fresh process tree."""
    if not cpu_selftest:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (n, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("[bench] spawning %d ranks: %s" % (n, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd, env=env))


# ---------------------------------------------------------------------------------------------------------------
# data / index builders
# ---------------------------------------------------------------------------------------------------------------
def mixture(n, d, ncent, sigma, seed_c, seed_x, dev):
    """Isotropic Gaussian mixture (1024 centres ~U[0,1]^d, sigma 0.1)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed_c)
    cent = torch.rand((ncent, d), generator=g, device=dev)
    g.manual_seed(seed_x)
    which = torch.randint(0, ncent, (n,), generator=g, device=dev)
    x = cent[which] + sigma * torch.randn((n, d), generator=g, device=dev)
    return x.contiguous()


def lowrank_mixture(n, d, ncent, seed_c, seed_x, dev, rank=16, sigma_in=0.5, sigma_out=0.01):
    """A dataset with structure a product quantizer can use AND cells that overlap: 64 mixture centres in U[0,1]^d (each
    is cut into ~16 of the kc = 1024 Voronoi cells, so a query's true neighbour often sits in a neighbouring cell),
    within-cluster spread confined to a random rank-16 subspace (sigma 0.5 along it) plus a little isotropic noise
    (sigma 0.01).  The isotropic mixture above puts sigma = 0.1 on all 128 axes and one centre per cell: the PQ residual
    is white noise, recall@1 sits at the PQ ceiling (~0.2) whatever w is, and the number says nothing about the search.
    Here recall moves with w and stops at the ceiling the quantizer allows."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed_c)
    cent = torch.rand((ncent, d), generator=g, device=dev)
    basis = torch.linalg.qr(torch.randn((d, rank), generator=g, device=dev))[0]          # d x rank, orthonormal columns
    g.manual_seed(seed_x)
    which = torch.randint(0, ncent, (n,), generator=g, device=dev)
    z = sigma_in * torch.randn((n, rank), generator=g, device=dev)
    x = cent[which] + z @ basis.t() + sigma_out * torch.randn((n, d), generator=g, device=dev)
    return x.contiguous()


def make_data(kind, n, d, seed_x, dev):
    return mixture(n, d, 1024, 0.1, 99, seed_x, dev) if kind == "mixture" else lowrank_mixture(n, d, 64, 99, seed_x, dev)


def _train(pkg, x, kc, m):
    """the library's own trainer (ivfadc_train: HIP k-means++ / Lloyd, 25 iterations, deterministic per seed)"""
    return pkg.trainer.train_ivfadc_hip(x.cpu().numpy(), kc, 256, m, 25, 25, seed=7, device=x.device.index or 0)


def build_trained(pkg, cfg, dev, device_index, dist, data_kind="mixture"):
    """SIFT1M-shape: synthetic data (seed 1234), index trained by the build's own trainer, data encoded through the HIP
    push!/encode path.  Returns (index, data on the device)."""
    import torch
    d, n, kc, m = cfg["d"], cfg["n"], cfg["kc"], cfg["m"]
    t0 = time.time()
    x = make_data(data_kind, n, d, 1234, dev)
    if dist is not None and dist.get_world_size() > 1:
        # rank 0 trains, every rank gets the same quantizers, and the deterministic HIP encode builds identical replicas
        ct = torch.empty((kc, d), dtype=torch.float32, device=dev)
        bt = torch.empty((m, 256, d // m), dtype=torch.float32, device=dev)
        if dist.get_rank() == 0:
            cent, cbs, labels = _train(pkg, x, kc, m)
            ct.copy_(torch.as_tensor(cent))
            bt.copy_(torch.as_tensor(cbs))
        dist.broadcast(ct, 0)
        dist.broadcast(bt, 0)
        cent, cbs = ct.cpu().numpy(), bt.cpu().numpy()
        labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    else:
        cent, cbs, labels = _train(pkg, x, kc, m)
    idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=device_index)
    idx._append(x.cpu().numpy(), np.arange(n, dtype=np.uint32))
    log("[bench] %s index built in %.1fs: %r" % (data_kind, time.time() - t0, idx))
    return idx, x


def synth_sizes(n, kc, seed, skew=False):
    rng = np.random.default_rng(seed)
    p = np.full(kc, 1.0 / kc)
    if skew:
        p = rng.dirichlet(np.full(kc, 0.5))
    sizes = rng.multinomial(n, p).astype(np.int64)
    off = np.zeros(kc + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    return off


def synth_quantizers(cfg):
    d, kc, m = cfg["d"], cfg["kc"], cfg["m"]
    rng = np.random.default_rng(7)
    cent = rng.standard_normal((kc, d), dtype=np.float32)
    cbs = rng.standard_normal((m, 256, d // m), dtype=np.float32)
    labels = np.tile(np.arange(256, dtype=np.uint8), (m, 1))
    return cent, cbs, labels


def build_synth(pkg, cfg, device_index, skew=False):
    """Billion-scale shapes: quantizers ~N(0,1) (seed 7), code bytes synthesised on the device by the
    counter-based RNG the oracle can replay, ids = position."""
    cent, cbs, labels = synth_quantizers(cfg)
    idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=device_index)
    off = synth_sizes(cfg["n"], cfg["kc"], 7, skew)
    t0 = time.time()
    idx.synth_lists(off, 20260101)
    log("[bench] synthetic lists on device in %.1fs: %r" % (time.time() - t0, idx))
    return idx, (cent, cbs, labels, off)


def global_queries(cfg, nq_total, dev, data_kind="mixture"):
    """The global batch of a step: every rank generates the same nq_total queries and keeps its own block."""
    import torch
    if cfg["kind"] == "trained":
        return make_data(data_kind, nq_total, cfg["d"], 4321, dev)
    return torch.as_tensor(np.random.default_rng(11).standard_normal((nq_total, cfg["d"]), dtype=np.float32)).to(dev)


def recall_at_1(x, q, ids, counts):
    """fraction of queries whose exact L2 nearest neighbour is among the returned ids."""
    import torch
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


# ---------------------------------------------------------------------------------------------------------------
# the per-step machinery: result rings + ONE collective per G batches
# ---------------------------------------------------------------------------------------------------------------
class StubIndex:
    """`--selftest-cpu` only: stands in for the HIP index so the launcher, the query partition and the result gather can
    be exercised end to end on a host without a GPU (gloo).  It computes nothing: slot q of a rank's block is filled with
    a pattern of the GLOBAL query number, so the gathered batch can be verified exactly.  Never used when a GPU is present."""

    def __init__(self, lo):
        self.lo = lo

    def fill(self, view, nq, K):
        import torch
        gq = torch.arange(self.lo, self.lo + nq, dtype=torch.int32)
        ids = (gq[:, None] * K + torch.arange(K, dtype=torch.int32)[None, :]).reshape(-1)
        dists = (gq[:, None].float() + 0.5 * torch.arange(K)[None, :]).reshape(-1).view(torch.int32)
        view[:nq * K] = ids
        view[nq * K:2 * nq * K] = dists
        view[2 * nq * K:] = K


class Rings:
    """Results of batch i land in slot (i % G) of ring ((i // G) % NR).  With a process group, a full ring is all-gathered
    by ONE collective on a side stream while the next ring fills (NR rings: the main stream only waits for a ring's
    previous gather when it comes round again)."""

    def __init__(self, torch, dist, dev, world, nq, K, G, NR, gpu):
        self.torch, self.dist, self.dev, self.world, self.nq, self.K, self.G, self.gpu = torch, dist, dev, world, nq, K, G, gpu
        self.width = 2 * K + 1
        self.NR = NR if dist is not None else 1
        blk = G * nq * self.width
        self.ring = [torch.zeros(blk, dtype=torch.int32, device=dev) for _ in range(self.NR)]
        self.gath = [torch.zeros(world * blk, dtype=torch.int32, device=dev) for _ in range(self.NR)] if dist is not None else None
        self.main = torch.cuda.current_stream() if gpu else None
        self.side = [torch.cuda.Stream(device=dev) for _ in range(self.NR)] if (gpu and dist is not None) else None
        self.busy = [False] * self.NR
        self.filled = [0] * self.NR
        self.collectives = 0

    def slot_of(self, i):
        return (i // self.G) % self.NR, i % self.G

    def slot_view(self, i):
        r, sl = self.slot_of(i)
        per = self.nq * self.width
        return self.ring[r][sl * per:(sl + 1) * per]

    def ptrs(self, view):
        base = view.data_ptr()
        return base, base + self.nq * self.K * 4, base + 2 * self.nq * self.K * 4

    def before_step(self, i):
        r, sl = self.slot_of(i)
        if self.dist is not None and sl == 0 and self.busy[r]:
            if self.gpu:
                self.main.wait_stream(self.side[r])      # the ring's previous gather must have read it
            self.busy[r] = False

    def after_step(self, i):
        if self.dist is None:
            return
        r, sl = self.slot_of(i)
        self.filled[r] = sl + 1
        if sl == self.G - 1:
            self.flush(r)

    def flush(self, r):
        # Plain (non-async_op) collective: its host cost is 12 us vs 28 us for the Work-object form (tools/ag_micro.py).
        if self.gpu:
            self.side[r].wait_stream(self.main)
            with self.torch.cuda.stream(self.side[r]):
                self.dist.all_gather_into_tensor(self.gath[r], self.ring[r])
        else:
            self.dist.all_gather_into_tensor(self.gath[r], self.ring[r])
        self.collectives += 1
        self.busy[r] = True
        self.filled[r] = 0

    def drain(self):
        if self.dist is None:
            return
        for r in range(self.NR):
            if self.filled[r]:
                self.flush(r)                 # a partly filled ring is gathered whole
        for r in range(self.NR):
            if self.busy[r]:
                if self.gpu:
                    self.main.wait_stream(self.side[r])
                self.busy[r] = False



# ---------------------------------------------------------------------------------------------------------------
# roofline accounting of one scan launch (shared by the headline configuration and `other_configs`)
# ---------------------------------------------------------------------------------------------------------------
def kernel_label(m, st):
    if st["last_qg"] == -3:
        return "sq_kernel<M=%d> (small batch: one launch, (query, probe, chunk)-parallel, last-arriver merge)" % m
    if st["last_qg"] > 0 and st.get("last_striped", 0) == 2:
        return "wg8_scan_kernel<M=%d,QG=%d> (list-major, eight waves per workgroup, four conflict-free table copies)" % (m, st["last_qg"])
    if st["last_qg"] > 0 and st.get("last_striped", 0) == 3:
        return "wg8q8_scan_kernel<M=%d,QG=%d> (list-major, eight waves per workgroup, 16-byte entries in two conflict-free copies)" % (m, st["last_qg"])
    if st["last_qg"] > 0:
        return "scan_kernel<M=%d,QG=%d%s> (list-major)" % (m, st["last_qg"], ",NF" if st.get("last_nf", 0) else "")
    if st.get("last_lb", 0):
        return "qscan_kernel<M=%d, LB> (query-major, 8-bit lower-bound tables from the matrix cores)" % m
    return "qscan_kernel<M=%d> (query-major)" % m


def lds_form(m, st):
    if st.get("last_nf", 0):
        return "nf5x%d" % st["last_qg"]
    if st.get("last_striped", 0) == 2:
        return "q16x4cf"
    if st.get("last_striped", 0) == 3:
        return "q16x8cf"
    if st.get("last_striped", 0):
        return "q16x4" if m == 8 else "striped"
    if st.get("last_lb", 0):
        return "u8"
    return "b128x4" if st["last_qg"] == 4 else ("b64x2" if st["last_qg"] == 2 else "b32")


def traffic_replay(key):
    """HBM bytes per scan launch from the committed PMC passes (profiles/traffic.json, written by tools/summarize_pmc.py: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2*FETCH + WRITE)*1024 per the gfx950 correction of MI355X_MICROARCH.md), replayed
    ONLY next to a run with the same workload / plan / kernel key.  Returns (bytes or None, source string)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
    except Exception as e:           # noqa: BLE001
        return None, "none: %s" % e
    ent = tj.get(key)
    if ent is None:
        return None, "none: no committed PMC pass for %r" % key
    return ent.get("hbm_bytes_per_launch"), "REPLAYED from profiles/traffic.json (%s), not measured in this run" % ent.get("source", "?")


def traffic_entry(key):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:            # noqa: BLE001
        return None


def lds_roofs():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "lds_roof.json")))
    except Exception:            # noqa: BLE001
        return {}


def roofline_of(config_name, cfg, nq, w, K, st, pruning_on=True, riders=False):
    """The roofline block of the dominant kernel from the library's own HIP-event timings (`st` = ivfadc_get_stats after a profiled region).
    Query-major kernels read a list once per (query, probe): `frac` = SURVEY 8(d)'s algorithmic bytes actually scanned / time / 8 TB/s.
    List-major kernels share ONE code stream among the queries that probe a list, so 8(d)'s per-(query, point) bytes exceed what any
    memory system must move: there that figure is `alg_frac_shared_stream` and `frac` is the BINDING one -- the larger of the physical
    HBM fraction (replayed PMC traffic / time / 8 TB/s) and the LDS-gather fraction (query-lookups per clock and CU / measured roof)."""
    m = cfg["m"]
    launches = max(1, st["scan_launches"])
    scan_ms = st["scan_ms"] / launches
    pruned_frac = st.get("pruned_points", 0) / max(1, st["scanned_points"])
    balg_sec8d = st["scanned_points"] / launches * m
    balg = balg_sec8d * (1.0 - pruned_frac)
    t = scan_ms * 1e-3
    alg_gbs = balg / t / 1e9 if t > 0 else 0.0
    kname = kernel_label(m, st)
    if riders:
        kname = "qscan_coarse_kernel = " + kname.split(" (")[0] + "+riders (the next batch's exact coarse tiles in the same launch)"
    key = "%s|n=%d|kc=%d|nq=%d|w=%d|K=%d|pruning=%d|%s" % (config_name, cfg["n"], cfg["kc"], nq, w, K, 1 if pruning_on else 0, kname.split(" (")[0])
    traffic, traffic_source = traffic_replay(key)
    phys = (traffic / t / 1e9 / HBM_PEAK_GBS) if (traffic and t > 0) else None
    lookups = balg / t / NOMINAL_CLOCK_HZ / NUM_CU if t > 0 else 0.0
    form = lds_form(m, st)
    roof = lds_roofs().get(form, {})
    lds_peak = roof.get("lookups_per_clk_cu")
    rl_lds = {"achieved": round(lookups, 2), "unit": "query-lookups/clk/CU at %.1f GHz nominal" % (NOMINAL_CLOCK_HZ / 1e9), "form": form,
              "peak": lds_peak, "frac": round(lookups / lds_peak, 4) if lds_peak else None}
    if roof.get("conflict_free"):
        rl_lds["peak_conflict_free"] = roof["conflict_free"]
        rl_lds["frac_conflict_free"] = round(lookups / roof["conflict_free"], 4)
    list_major = st["last_qg"] > 0
    alg_frac = alg_gbs / HBM_PEAK_GBS
    r = {"kernel": kname, "traffic_key": key, "scan_ms_per_launch": round(scan_ms, 5), "alg_bytes_per_launch": int(balg),
         "traffic": traffic, "traffic_source": traffic_source, "physical_hbm_frac": round(phys, 4) if phys is not None else None,
         "coarse_ms_per_launch": round(st["coarse_ms"] / launches, 5), "roofline_lds": rl_lds}
    if list_major and st["last_qg"] > 1:
        # the HBM roofline of a shared code stream is the PHYSICAL one (PMC bytes of the committed pass of this very kernel and workload / this
        # run's time / 8 TB/s); without a committed pass for the key the LDS-gather fraction is all there is, and it is labelled as such
        if phys is not None:
            # `frac` stays the physical HBM fraction; the label names the pipe that is nearer its roof (per-XCD work queues took the eight-wave
            # kernel's traffic from 31 to 16 GB per launch: its gathers are then nearer the measured conflict-free LDS rate than HBM its peak)
            nearer_lds = rl_lds["frac"] is not None and rl_lds["frac"] > phys
            r.update({"bound": "lds (gathers nearer their measured roof than HBM its peak; frac = physical HBM fraction)" if nearer_lds else "hbm",
                      "achieved": round(traffic / t / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s (physical: replayed PMC bytes)", "frac": round(phys, 4)})
        else:
            r.update({"bound": "lds", "achieved": round(lookups, 2), "peak": lds_peak, "unit": "query-lookups/clk/CU", "frac": rl_lds["frac"]})
        r["lds_gather_frac_random_bank"] = rl_lds["frac"]
        r["alg_frac_shared_stream"] = round(alg_frac, 4)
        r["alg_GBps_shared_stream"] = round(alg_gbs, 2)
        r["frac_note"] = "list-major: %d queries share one code stream, so SURVEY 8(d)'s m bytes per (query, point) pair are not bytes any memory " \
                         "must move; frac = physical_hbm_frac (replayed PMC traffic / this run's scan time / 8 TB/s)" % st["last_qg"]
    else:
        bound = "hbm"
        if rl_lds["frac"] is not None and phys is not None and rl_lds["frac"] > max(phys, alg_frac):
            bound = "lds"
        if max(alg_frac, phys or 0.0, rl_lds["frac"] or 0.0) < 0.3:
            bound = "latency/issue (no pipe near its roof: fixed per-query costs and the launch tail decide; the HBM fraction is not the yardstick here)"
        r.update({"bound": bound, "achieved": round(alg_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_frac, 4)})
    if not list_major and not st.get("last_lb", 0) and t > 0:
        # The query-major kernel rebuilds an exact ADC table per (query, probe): m x 256 entries of dsub dimensions, three f32 vector
        # operations per dimension (sub, mul, add: the reference's sums -- index.jl:230-233 -- round the product and the sum separately, so
        # neither an FMA nor the matrix cores may stand in), plus m adds per scanned point, plus -- when the next batch's exact coarse tiles ride
        # in the same launch -- nq x kc x d x 3 for those.  Against the plain-f32 vector rate (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: packed
        # adds / multiplies are double-pass on gfx950 and gain nothing).  A LOWER bound of the work: tables are counted for the probes that were
        # scanned (a round of two probes builds both tables even when the second is pruned), address arithmetic and selection are not counted.
        dsub = cfg["d"] // m
        pts = (st["scanned_points"] - st.get("pruned_points", 0)) / launches
        probes = pts / max(1.0, cfg["n"] / cfg["kc"])
        ops_tab = probes * m * 256 * dsub * 3
        ops_scan = pts * m
        ops_rider = float(nq) * cfg["kc"] * cfg["d"] * 3 if riders else 0.0
        ops = ops_tab + ops_scan + ops_rider
        r["roofline_valu"] = {"achieved": round(ops / t / 1e12, 2), "peak": VALU_PEAK_TOPS, "unit": "T f32 lane-operations/s (sub, mul, add)",
                              "frac": round(ops / t / 1e12 / VALU_PEAK_TOPS, 4),
                              "lane_ops_per_launch": {"tables": int(ops_tab), "scan_adds": int(ops_scan), "rider_coarse_tiles": int(ops_rider)},
                              "note": "lower bound of the exact arithmetic the launch executes / its duration / the plain-f32 vector peak; the reference's "
                                      "rounding (product and sum rounded separately, sums in index order) admits neither FMA nor MFMA here"}
    if r.get("bound") == "lds":
        # the measured roof prices the bank conflicts of random codes as unavoidable; against the conflict-free rate of the same ds_read form:
        r["frac_lds_conflict_free"] = rl_lds.get("frac_conflict_free")
    # the same kernel's average in the committed rocprofv3 --kernel-trace --stats pass of this very workload (replayed, like `traffic`)
    tr = traffic_entry(key)
    if tr and tr.get("trace_avg_ms"):
        tms = tr["trace_avg_ms"]
        r["trace"] = {"scan_ms_per_launch": round(tms, 5), "calls": tr.get("trace_calls"), "source": "REPLAYED from profiles/traffic.json (%s)" % tr.get("trace_source", "?"),
                      "alg_frac": round(balg / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "physical_hbm_frac": round(traffic / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                      "events_over_trace": round(scan_ms / tms, 4)}
    r["pruned_fraction_of_sec8d_bytes"] = round(pruned_frac, 4)
    r["sec8d_alg_bytes_per_launch"] = int(balg_sec8d)
    r.update({"chunk_points": st["last_chunk"], "scan_grid": st["last_scan_grid"], "scan_lds_bytes": st["last_scan_lds"]})
    return r


# ---------------------------------------------------------------------------------------------------------------
# the ONE stdout line: compact (the harness keeps a bounded tail of stdout); everything else goes to a side file and stderr
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4000          # bytes of the stdout JSON line (asserted by tests/test_distributed.py and tests/test_gpu_parity.py)
FULL_RECORD = os.path.join("gpurun_out", "bench_full.json")


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short_bound(b):
    b = str(b or "")
    for tag in ("hbm", "lds", "valu", "latency", "mfma"):
        if b.startswith(tag):
            return tag
    return b[:24]


def _rl_compact(rl):
    """The roofline object of the compact line: the contract's keys + the figures a reader needs to recompute them."""
    if not isinstance(rl, dict):
        return None
    out = _pick(rl, ("kernel", "traffic_key", "bound", "achieved", "peak", "unit", "frac", "traffic", "physical_hbm_frac", "alg_bytes_per_launch",
                     "scan_ms_per_launch", "coarse_ms_per_launch", "alg_frac_shared_stream", "frac_lds_conflict_free", "scan_grid"))
    out["kernel"] = str(rl.get("kernel", "")).split(" (")[0][:80]
    out["bound"] = _short_bound(rl.get("bound"))
    out["unit"] = str(rl.get("unit", "GB/s")).split(" (")[0]
    out.setdefault("traffic", None)
    if rl.get("traffic") is not None:
        out["traffic_source"] = "replayed: profiles/traffic.json (committed rocprofv3 --pmc passes of this workload)"
    pr = rl.get("pruning") or {}
    out["sec8d_bytes_per_launch"] = rl.get("sec8d_alg_bytes_per_launch")
    out["pruned_fraction"] = rl.get("pruned_fraction_of_sec8d_bytes")
    off = pr.get("pruning_off_same_run") or {}
    if off:
        out["pruning_off"] = _pick(off, ("scan_ms_per_launch", "frac", "qps"))
    tb = rl.get("table_build")
    if tb:
        out["table_build"] = _pick(tb, ("ms_per_launch", "achieved", "peak", "frac"))
    return out


def compact_line(full):
    """The stdout line from the full record: the contract's keys, the roofline and CPU-baseline objects, the contract rates as scalars and
    one small tuple per other shape.  Every explanation, window list, sweep and host-path breakdown stays in the full record."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "selftest_cpu", "gather_check", "ranks_seen_by_rccl", "efficiency_vs_scaling_base", "collectives_in_timed_region")
    c = {k: full[k] for k in keep if k in full}
    cfg = full.get("config") or {}
    c["config"] = _pick(cfg, ("workload", "global_batch", "queries_per_rank", "index", "partition", "pruning", "single_mode", "batches_in_flight",
                              "recall_at_1_in_top10", "recall_ceiling_w=kc"))
    if isinstance(cfg.get("parallelism"), str):
        c["config"]["parallelism"] = cfg["parallelism"][:160]
    hint = full.get("next_batch_hint") or {}
    if hint:
        c["config"]["next_batch_hint"] = bool(hint.get("used_by_this_plan"))
    w_ = full.get("windows") or {}
    if w_:
        c["windows"] = _pick(w_, ("n", "steps_each", "qps_min", "qps_max"))
    c["roofline"] = _rl_compact(full.get("roofline"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_qps", "host_logical_cpus"))
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:150]
    else:
        c["cpu_baseline"] = None
    c["parity"] = full.get("parity")
    # the contract rates, one scalar each (queries/s)
    bif = full.get("batches_in_flight") or {}
    rates = {}
    one_h = (bif.get("one_in_flight_same_run") or {}).get("qps")
    one_p = (bif.get("one_in_flight_without_hint_same_run") or {}).get("qps")
    if bif.get("inflight") == 1 and full.get("value") is not None:
        # the headline itself ran one batch at a time: it IS the one-in-flight rate (plain when no hint was in use)
        if hint.get("used_by_this_plan"):
            one_h = full["value"]
        else:
            one_p = full["value"]
    rates["plain_one_in_flight_qps"] = one_p
    rates["hinted_one_in_flight_qps"] = one_h
    hh = full.get("host_to_host") or {}
    for name, key in (("host_blocking_qps", "blocking_search"), ("host_batches_qps", "search_batches")):
        kinds = hh.get(key) or {}
        ent = kinds.get("registered") or kinds.get("library_pinned") or kinds.get("pageable") or {}
        rates[name] = ent.get("qps")
        if name == "host_blocking_qps" and ent:
            rates["host_blocking_us_per_batch"] = ent.get("us_per_batch")
            rates["host_blocking_wait_us"] = (ent.get("host_us_per_batch") or {}).get("wait")
    if hh.get("error"):
        rates["host_error"] = str(hh["error"])[:120]
    if "results_identical_across_kinds_and_entries" in hh:
        rates["host_results_identical"] = hh["results_identical_across_kinds_and_entries"]
        rates["host_parity_ids"] = (hh.get("parity_64") or {}).get("ids_bit_exact")
    sb = full.get("scaling_base") or {}
    rates["scaling_base_qps"] = sb.get("qps_per_rank")
    if sb.get("error"):
        rates["scaling_base_error"] = str(sb["error"])[:120]
    c["rates"] = rates
    oc = full.get("other_configs")
    if isinstance(oc, dict):
        o = {}
        for tag, ent in oc.items():
            if not isinstance(ent, dict):
                continue
            if "error" in ent:
                o[tag[:40]] = {"error": str(ent["error"])[:100]}
                continue
            if "ms_per_step" not in ent:
                continue
            short = tag.replace(" (one rank's share of the 8-GPU batch)", "")
            par = ent.get("parity_64") or {}
            o[short] = {"ms_per_step": ent.get("ms_per_step"), "scan_ms": ent.get("scan_ms"), "coarse_ms": ent.get("coarse_ms"), "frac": ent.get("frac"),
                        "bound": _short_bound(ent.get("bound")), "physical_hbm_frac": ent.get("physical_hbm_frac"),
                        "lds_frac": (ent.get("roofline_lds") or {}).get("frac"),
                        "alg_GB": round(ent["alg_bytes"] / 1e9, 3) if ent.get("alg_bytes") else None,
                        "parity": bool(par.get("ids_bit_exact") and par.get("dists_rtol_1e-4")) if par else None}
        c["other_configs"] = o
    d = full.get("distributed")
    if isinstance(d, dict):
        c["distributed"] = _pick(d, ("ranks_seen_by_rccl", "gather_check", "partition_check", "collectives_in_timed_region", "batches_per_collective",
                                     "bytes_per_rank_per_collective"))
        c["distributed"]["backend"] = str(d.get("backend", ""))[:60]
    c["full_record"] = FULL_RECORD
    return c


def emit(full, json_out):
    """Full record -> gpurun_out/bench_full.json and stderr; compact line (<= LINE_LIMIT bytes, enforced) -> stdout."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, FULL_RECORD), "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        log("[bench] could not write %s: %s" % (FULL_RECORD, e))
    log("[bench] full record: " + json.dumps(full))
    c = compact_line(full)
    text = json.dumps(c, separators=(",", ":"))
    for victim in ("other_configs", "windows", "rates", "distributed"):      # never reached with today's fields; the limit is a promise
        if len(text) <= LINE_LIMIT:
            break
        c.pop(victim, None)
        c["dropped_to_fit"] = c.get("dropped_to_fit", []) + [victim]
        text = json.dumps(c, separators=(",", ":"))
    print(text, file=json_out, flush=True)


def median_of(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def oracle_parity(ora, oidx, qh, K, w, ids, dists, counts, pick):
    """ids bit-exact / distances within 1e-4 relative against the oracle on the queries `pick` of the batch."""
    oi, od, oc = oidx.knn_search(qh[pick], K, w, nthreads=ora.max_threads())
    gi, gd, gc = ids[pick], dists[pick], counts[pick]
    ok_ids = bool(np.array_equal(gc, oc) and all(np.array_equal(gi[r, :gc[r]], oi[r, :oc[r]]) for r in range(len(pick))))
    ok_d = bool(all(np.allclose(gd[r, :gc[r]], od[r, :oc[r]], rtol=1e-4, atol=0) for r in range(len(pick))))
    return {"queries_checked": int(len(pick)), "ids_bit_exact": ok_ids, "dists_rtol_1e-4": ok_d}


TWO_LANE_CONFIGS = ("sift1m", "hd", "deep1b", "toy")   # shapes on which a second batch in flight pays (measured; see --inflight: round 5 HD +6 %, Deep1B +2 %, SIFT1B -2 %)


def measure_other_config(torch, pkg, name, cases, dev, device_index, budget_s=20.0, skew=False, two_lanes=True):
    """One of the other BASELINE.json shapes, briefly: a few windows of steps, a profiled region for the scan kernel's own time, the
    roofline fractions and a 64-query oracle parity bit.  Device-synthesised index (seconds), queries resident in HBM."""
    from oracle import oracle as ora
    cfg = dict(CONFIGS[name])
    out = []
    t_build = time.perf_counter()
    idx, (cent, cbs, labels, off) = build_synth(pkg, cfg, device_index, skew)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    q_all = global_queries(cfg, cfg["nq"], dev).contiguous()
    oidx = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=20260101)
    t_build = time.perf_counter() - t_build
    for case in cases:
        w, K = case[0], case[1]
        nq = case[2] if len(case) > 2 else cfg["nq"]        # (a third entry: another batch size, e.g. one rank's share of the 8-GPU batch)
        q = q_all[:nq].contiguous()
        qh = q.cpu().numpy()
        t_cfg = time.perf_counter()
        res = torch.zeros(nq * (2 * K + 1), dtype=torch.int32, device=dev)
        p_ids, p_d, p_c = res.data_ptr(), res.data_ptr() + nq * K * 4, res.data_ptr() + 2 * nq * K * 4

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                idx.search_device(nq, q.data_ptr(), K, w, p_ids, p_d, p_c)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        t1 = run(3) / 3.0                                       # warm-up, and a first estimate of the step
        nsteps = int(max(3, min(50, 0.2 * budget_s / 5.0 / max(t1, 1e-5))))
        wins = [run(nsteps) for _ in range(5)]
        idx.set_profiling(True)
        idx.reset_stats()
        run(min(nsteps, 10))
        st = idx.get_stats()
        idx.set_profiling(False)
        rl = roofline_of(name + ("-skewed" if skew else ""), cfg, nq, w, K, st)     # (the traffic replay is keyed: no skewed run next to uniform bytes)
        torch.cuda.synchronize()
        h = res.cpu().numpy()
        ids = h[:nq * K].view(np.uint32).reshape(nq, K)
        dists = h[nq * K:2 * nq * K].view(np.float32).reshape(nq, K)
        counts = h[2 * nq * K:]
        pick = np.sort(np.random.default_rng(5).choice(nq, 64, replace=False))
        par = oracle_parity(ora, oidx, qh, K, w, ids, dists, counts, pick)
        med = median_of(wins)
        two = None
        if name in TWO_LANE_CONFIGS and two_lanes:
            # two batches in flight (index + view in turn, results into separate buffers); the roofline above stays that of one at a time
            view = idx.clone_view()
            res2 = torch.zeros_like(res)
            lanes = [(idx, (p_ids, p_d, p_c)), (view, (res2.data_ptr(), res2.data_ptr() + nq * K * 4, res2.data_ptr() + 2 * nq * K * 4))]

            def run2(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    h_, o_ = lanes[i & 1]
                    h_.search_device(nq, q.data_ptr(), K, w, o_[0], o_[1], o_[2])
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            run2(4)
            wins2 = [run2(nsteps) for _ in range(5)]
            med2 = median_of(wins2)
            two = {"qps": round(nq * nsteps / med2, 1), "ms_per_step": round(med2 / nsteps * 1e3, 4),
                   "results_identical_to_one_in_flight": bool(torch.equal(res, res2))}
            del view
        out.append({"workload": "%s-shape: d=%d n=%d kc=%d m=%d, batch=%d, K=%d, w=%d (device-synthesised codes, N(0,1) quantizers%s)"
                                % (name, cfg["d"], cfg["n"], cfg["kc"], cfg["m"], nq, K, w,
                                   "; SKEWED list sizes: synth_sizes(skew=True), longest list %d points against a mean of %d" % (int(np.max(np.diff(off))), cfg["n"] // cfg["kc"]) if skew else ""),
                    "w": w, "K": K, "skew": bool(skew), "batch": nq,
                    "qps": round(nq * nsteps / med, 1), "ms_per_step": round(med / nsteps * 1e3, 4),
                    "windows": {"n": len(wins), "steps_each": nsteps, "ms_per_step_min": round(min(wins) / nsteps * 1e3, 4),
                                "ms_per_step_max": round(max(wins) / nsteps * 1e3, 4)},
                    "scan_ms": rl["scan_ms_per_launch"], "coarse_ms": rl["coarse_ms_per_launch"], "alg_bytes": rl["alg_bytes_per_launch"],
                    "frac": rl["frac"], "bound": rl["bound"], "physical_hbm_frac": rl["physical_hbm_frac"],
                    "frac_lds_conflict_free": rl.get("frac_lds_conflict_free"), "trace": rl.get("trace"),
                    "alg_frac_shared_stream": rl.get("alg_frac_shared_stream"),
                    "roofline_lds": {k: rl["roofline_lds"].get(k) for k in ("achieved", "form", "peak", "frac", "peak_conflict_free", "frac_conflict_free")},
                    "roofline_valu": ({k: rl["roofline_valu"].get(k) for k in ("achieved", "peak", "unit", "frac")} if rl.get("roofline_valu") else None),
                    "kernel": rl["kernel"].split(" (")[0], "traffic_key": rl["traffic_key"], "traffic": rl["traffic"],
                    "parity_64": par, "two_batches_in_flight": two, "seconds": round(time.perf_counter() - t_cfg, 1),
                    "twolevel_probe_fraction": round(float(st.get("twolevel_probe_fraction", -1.0)), 4), "coarse_two_level": bool(st.get("last_twolevel", 0))})
        log("[bench] other config %s w=%d: %.4f ms/step, scan %.4f ms, frac %s (%s), parity %s" %
            (name, w, med / nsteps * 1e3, rl["scan_ms_per_launch"], rl["frac"], rl["bound"].split(" ")[0], par["ids_bit_exact"]))
    out[0]["index_build_seconds"] = round(t_build, 1)
    del idx
    torch.cuda.empty_cache()
    return out


def measure_two_level(torch, pkg, K, dev, device_index, n_train=200_000, n_index=20_000_000, nq=10_000, w=32):
    """the certified two-level coarse search on a TRAINED kc = 65 536 quantizer (Deep1B shape d = 96, m = 16; the library's
    own trainer on the Gaussian-mixture data of BASELINE.md: 1024 centres, sigma 0.1 -- a few Lloyd iterations, the structure is what
    matters here), device-synthesised lists, fresh mixture queries.  The same batch with the exhaustive coarse stage (matrix-core
    filter + certified refine + top-w: ivfadc_set_coarse_mode(h, 7)) and with the two-level search (automatic mode's own decision is
    reported, then mode 6): whole-step times, the coarse stage as step - scan, the fraction of the kc distances still computed,
    identical results, and 64 queries against the oracle."""
    from oracle import oracle as ora
    d, kc, m = 96, 65536, 16
    t0 = time.perf_counter()
    x = mixture(n_train, d, 1024, 0.1, 99, 777, dev)
    cent, cbs, labels = pkg.trainer.train_ivfadc_hip(x.cpu().numpy(), kc, 256, m, 2, 2, seed=7, device=device_index)
    del x
    t_train = time.perf_counter() - t0
    idx = pkg.IVFADCIndex.from_arrays(cent, cbs, labels, device=device_index)
    off = synth_sizes(n_index, kc, 7, False)
    idx.synth_lists(off, 20260101)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    q = mixture(nq, d, 1024, 0.1, 99, 4321, dev)
    res = [torch.zeros(nq * (2 * K + 1), dtype=torch.int32, device=dev) for _ in range(2)]

    def run(mode, r, n):
        idx.set_coarse_mode(mode)
        p = res[r].data_ptr()
        for _ in range(3):
            idx.search_device(nq, q.data_ptr(), K, w, p, p + nq * K * 4, p + 2 * nq * K * 4)
        torch.cuda.synchronize()
        wins = []
        for _ in range(5):
            t1 = time.perf_counter()
            for _ in range(n):
                idx.search_device(nq, q.data_ptr(), K, w, p, p + nq * K * 4, p + 2 * nq * K * 4)
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t1) / n)
        idx.set_profiling(True)
        idx.reset_stats()
        for _ in range(5):
            idx.search_device(nq, q.data_ptr(), K, w, p, p + nq * K * 4, p + 2 * nq * K * 4)
        st = idx.get_stats()
        idx.set_profiling(False)
        step = median_of(wins)
        scan = st["scan_ms"] / max(1, st["scan_launches"])
        return {"ms_per_step": round(step * 1e3, 4), "scan_ms": round(scan, 4), "coarse_stage_ms (step - scan)": round(step * 1e3 - scan, 4),
                "qps": round(nq / step, 1), "two_level": bool(st["last_twolevel"]),
                "fraction_of_kc_distances_computed": round(st["coarse_visited"] / (5.0 * nq * kc), 5) if st["last_twolevel"] else 1.0}, st
    auto, st_auto = run(0, 0, 10)              # the first search builds the grouping and runs the self-probe
    exh, _ = run(7, 0, 10)
    two, st_two = run(6, 1, 10)
    same = bool(torch.equal(res[0], res[1]))
    oidx = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=20260101)
    qh = q.cpu().numpy()
    h_ = res[1].cpu().numpy()
    pick = np.sort(np.random.default_rng(5).choice(nq, 64, replace=False))
    par = oracle_parity(ora, oidx, qh, K, w, h_[:nq * K].view(np.uint32).reshape(nq, K), h_[nq * K:2 * nq * K].view(np.float32).reshape(nq, K),
                        h_[2 * nq * K:], pick)
    del idx
    torch.cuda.empty_cache()
    return {"workload": "trained quantizer (ivfadc_train, 2 Lloyd iterations on %d mixture points: 1024 centres, sigma 0.1), d=%d kc=%d m=%d, "
                        "%d device-synthesised points, batch=%d mixture queries, K=%d, w=%d" % (n_train, d, kc, m, n_index, nq, K, w),
            "train_seconds": round(t_train, 1), "groups": int(st_two["twolevel_groups"]),
            "self_probe_fraction": round(float(st_two["twolevel_probe_fraction"]), 5), "automatic_mode_chose_two_level": bool(auto["two_level"]),
            "exhaustive": exh, "two_level": two, "automatic": auto,
            "coarse_stage_speedup": round(exh["coarse_stage_ms (step - scan)"] / max(1e-9, two["coarse_stage_ms (step - scan)"]), 2),
            "step_speedup": round(exh["ms_per_step"] / two["ms_per_step"], 3),
            "results_identical_exhaustive_vs_two_level": same, "parity_64": par,
            "note": "exact in both modes (the bounds only skip groups that cannot hold a top-w centroid, ties included).  The BASELINE bench "
                    "quantizers are N(0,1) (no structure): their self-probe fraction is ~1 and automatic mode keeps the exhaustive kernels "
                    "there -- see other_configs[*].twolevel_probe_fraction"}


def measure_scaling_base(torch, pkg, idx, nq, q, K, w, dev, steps, nwin, hinted):
    """The per-rank step of an N > 1 run, on ONE rank: one batch at a time on the index itself (no second lane), every batch followed by
    the library's own ncclAllGather of the packed top-k on the handle's side stream (ivfadc_search_device_allgather, eight result slots in
    rotation) -- here on a communicator of ONE rank, so the collective degenerates to a copy.  N x this rate is what an N-GPU run of the
    same per-GPU batch would reach with a free all-gather; value(N) / (N x scaling_base) is the like-for-like efficiency (the N = 1
    headline runs two batches in flight and no collective: not the base of a scaling curve)."""
    width = 2 * K + 1
    ring = [torch.zeros(nq * width, dtype=torch.int32, device=dev) for _ in range(8)]
    gath = [torch.zeros(nq * width, dtype=torch.int32, device=dev) for _ in range(8)]
    idx.comm_init(1, 0, pkg.comm_unique_id())
    try:
        def step(i):
            if hinted:
                idx.set_query_token(1)
                idx.set_next_queries(nq, q.data_ptr(), 1)
            r = i % 8
            idx.search_device_allgather(nq, q.data_ptr(), K, w, ring[r].data_ptr(), gath[r].data_ptr(), r)
        for i in range(64):
            step(i)
        idx.comm_wait()
        torch.cuda.synchronize()
        wins = []
        for _ in range(max(1, nwin)):
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            idx.comm_wait()
            torch.cuda.synchronize()
            wins.append(time.perf_counter() - t0)
        el = median_of(wins)
        same = bool(torch.equal(ring[(steps - 1) % 8], gath[(steps - 1) % 8]))
    finally:
        idx.comm_destroy()
    return {"qps_per_rank": round(nq * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4), "windows": len(wins),
            "qps_min": round(nq * steps / max(wins), 1), "qps_max": round(nq * steps / min(wins), 1), "gathered_equals_block": same,
            "mode": "one batch at a time per rank, next-batch hint %s, library collective (ncclAllGather on a side stream, 8 slots) on a ONE-rank "
                    "communicator: the mode N > 1 runs take, with a collective that moves nothing between GPUs" % ("on" if hinted else "off")}


def measure_host_to_host(torch, pkg, idx, cfg, K, w, dev, device_index, data_kind, budget_s=0.4, kinds=("pageable", "registered", "library_pinned")):
    """The reference's own call contract on the headline shape -- knn_search(ivfadc, points, k) takes HOST vectors and returns HOST vectors
    (index.jl:261-265) -- through the C ABI's host-pointer entries: blocking ivfadc_search per batch, and ivfadc_search_batches over 16
    consecutive batches (what the Julia shim's run-of-batches knn_search is ONE ccall of).  Three kinds of caller memory: pageable arrays (the
    library stages them through its own pinned blocks), arrays the caller registered (ivfadc_host_register) and blocks the library handed
    out (ivfadc_host_alloc: what the shim and the Python mirror pack the caller's vectors into) -- the last two are ingested / written in place.
    A handle of its own (stream of its own, as a Julia caller has), sixteen DIFFERENT batches, every rate a wall-clock time around the calls,
    results compared across kinds bit for bit and with the oracle on 64 queries."""
    import ctypes as C
    from ivfadc_jl_amd import _native as nat
    from oracle import oracle as ora
    L = nat.lib()
    nb, nq, d = 16, cfg["nq"], cfg["d"]
    offsets, codes, lids = idx._lists()
    h = pkg.IVFADCIndex.from_arrays(idx._centroids, idx._codebooks, idx._labels, offsets, codes, lids, device=device_index)
    qsrc = global_queries(cfg, nb * nq, dev, data_kind).cpu().numpy()
    bn = np.full(nb, nq, np.int64)

    def hstats():
        st = nat.HostStats()
        nat.check(L.ivfadc_get_host_stats(h._h, C.byref(st)))
        return st

    def arrays(kind):
        if kind == "pageable":
            return qsrc.copy(), np.zeros((nb * nq, K), np.uint32), np.zeros((nb * nq, K), np.float32), np.zeros(nb * nq, np.int32), (lambda: None)
        if kind == "registered":
            a = (qsrc.copy(), np.zeros((nb * nq, K), np.uint32), np.zeros((nb * nq, K), np.float32), np.zeros(nb * nq, np.int32))
            for x_ in a:
                nat.host_register(x_)
            return a + ((lambda: [nat.host_unregister(x_) for x_ in a]),)
        pa = (nat.PinnedArray((nb * nq, d), np.float32), nat.PinnedArray((nb * nq, K), np.uint32), nat.PinnedArray((nb * nq, K), np.float32),
              nat.PinnedArray(nb * nq, np.int32))
        pa[0].a[...] = qsrc
        return pa[0].a, pa[1].a, pa[2].a, pa[3].a, (lambda: [x_.close() for x_ in pa])

    out = {"what": "knn_search's own contract (index.jl:261-265): host vectors in, host vectors out, wall clock around the C calls; a handle and "
                   "stream of its own, 16 different batches of %d queries, K=%d, w=%d.  pageable: plain arrays, staged through the library's pinned "
                   "blocks; registered: the caller's arrays page-locked once with ivfadc_host_register; library_pinned: blocks from ivfadc_host_alloc "
                   "(what julia/IVFADCHip.jl and the Python mirror pack the caller's vectors into).  Known memory is ingested by the first launch "
                   "and written by the last one: no staging copy, no D2H copy" % (nq, K, w),
           "batch": nq, "batches_per_call": nb, "blocking_search": {}, "search_batches": {}}
    ref = None
    same = True
    for kind in kinds:
        q, ids, dists, counts, cleanup = arrays(kind)

        def run_loop():
            for b in range(nb):
                s_ = slice(b * nq, (b + 1) * nq)
                nat.check(L.ivfadc_search(h._h, nq, nat.ptr(q[s_], C.c_float), K, w, nat.ptr(ids[s_], C.c_uint32), nat.ptr(dists[s_], C.c_float),
                                          nat.ptr(counts[s_], C.c_int32)))

        def run_batches():
            nat.check(L.ivfadc_search_batches(h._h, nb, nat.ptr(bn, C.c_int64), nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                              nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        for f, name in ((run_loop, "blocking_search"), (run_batches, "search_batches")):
            for _ in range(5):
                f()
            t0 = time.perf_counter()
            f()
            one = max(1e-6, time.perf_counter() - t0)
            reps = int(max(5, min(2000, budget_s / 5 / one)))
            wins, calls_us = [], []
            for _w in range(5):
                L.ivfadc_reset_host_stats(h._h)
                t0 = time.perf_counter()
                for _ in range(reps):
                    ta = time.perf_counter()
                    f()
                    calls_us.append((time.perf_counter() - ta) * 1e6)
                wins.append((time.perf_counter() - t0) / reps)
            el = median_of(wins)
            st = hstats()
            per = 1.0 / (reps * nb)
            cu = np.sort(np.array(calls_us))
            out[name][kind] = {"qps": round(nb * nq / el, 1), "us_per_batch": round(el / nb * 1e6, 2), "calls_timed": 5 * reps,
                               "value_is": "median of 5 windows of %d calls" % reps,
                               "qps_min": round(nb * nq / max(wins), 1), "qps_max": round(nb * nq / min(wins), 1),
                               "us_per_call_p10_p50_p90_max": [round(float(cu[int(0.1 * (len(cu) - 1))]), 1), round(float(cu[len(cu) // 2]), 1),
                                                               round(float(cu[int(0.9 * (len(cu) - 1))]), 1), round(float(cu[-1]), 1)],
                               "host_us_per_batch": {"stage_in": round(st.stage_in_us * per, 2), "enqueue": round(st.enqueue_us * per, 2),
                                                     "wait": round(st.wait_us * per, 2), "stage_out": round(st.stage_out_us * per, 2)},
                               "queries_read_in_place": bool(st.queries_direct > 0), "results_written_in_place": bool(st.results_direct > 0),
                               "streams_replaced_by_probe": int(st.streams_replaced)}
            res = (ids.copy(), dists.copy(), counts.copy())
            if ref is None:
                ref = res
            else:
                same = same and all(np.array_equal(a_, b_) for a_, b_ in zip(ref, res))
        cleanup()
    out["results_identical_across_kinds_and_entries"] = bool(same)
    try:     # a CPU quota on the box shows up as rare multi-millisecond calls (the polling thread is throttled): recorded, not hidden
        out["cgroup"] = {"cpu.max": open("/sys/fs/cgroup/cpu.max").read().split(),
                         "cpu.stat": {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
                                      if k in ("nr_periods", "nr_throttled", "throttled_usec")}}
    except (OSError, ValueError):
        pass
    # the oracle on 64 queries of the LAST batch (the one farthest from anything a warm-up could have left behind)
    o0 = (nb - 1) * nq
    oidx = ora.OracleIndex(idx._centroids, idx._codebooks, idx._labels, offsets, codes, lids)
    oi, od, oc = oidx.knn_search(qsrc[o0:o0 + 64], K, w, nthreads=ora.max_threads())
    gi, gd, gc = ref[0][o0:o0 + 64], ref[1][o0:o0 + 64], ref[2][o0:o0 + 64]
    out["parity_64"] = {"ids_bit_exact": bool(np.array_equal(gc, oc) and all(np.array_equal(gi[r, :gc[r]], oi[r, :oc[r]]) for r in range(64))),
                        "dists_rtol_1e-4": bool(all(np.allclose(gd[r, :gc[r]], od[r, :oc[r]], rtol=1e-4, atol=0) for r in range(64)))}
    del h
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--gather-every", type=int, default=1,
                    help="multi-GPU: batches per all-gather (1 = one collective per batch, the headline mode)")
    ap.add_argument("--config", default="sift1m", choices=sorted(CONFIGS))
    ap.add_argument("--nq", type=int, default=0, help="queries per GPU and batch (--scaling strong: of the GLOBAL batch)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU gets the configuration's batch (global batch = N x batch); strong: the global batch is "
                         "the configuration's batch, every GPU gets 1/N of it")
    ap.add_argument("--partition", default="queries", choices=["queries", "lists"],
                    help="multi-GPU partition of a step.  queries (default): contiguous query blocks per rank, one all-gather of the packed "
                         "top-k.  lists: the strong-scaling mode for a FIXED global batch -- every rank gets ALL queries and scans the probed "
                         "lists l with l %% N == rank (ivfadc_search_device_listpart: partial keys, one all-gather, K-way merge on every rank)")
    ap.add_argument("--lists-rehearsal", type=int, default=0, metavar="N",
                    help="ONE GPU: rehearse --partition lists for N ranks -- the step of the full batch, every rank's slice of it (lists l %% N == r "
                         "for r = 0 .. N-1, one after the other) and the N-way merge are timed; prints the predicted speed-up t(full) / (max_r "
                         "t(slice r) + t(merge) [+ an assumed all-gather time])")
    ap.add_argument("--n", type=int, default=0, help="override the number of indexed vectors (synthetic configs)")
    ap.add_argument("--kc", type=int, default=0, help="override the number of coarse cells (synthetic configs)")
    ap.add_argument("--w", type=int, default=0)
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--qg", type=int, default=0)
    ap.add_argument("--coarse-mode", type=int, default=0, help="ivfadc_set_coarse_mode (A/B runs: 6 = certified two-level coarse search, 1 = exact kernel, 2 = MFMA filter from kc = 128)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--table-mode", type=int, default=0, help="ivfadc_set_table_mode (A/B runs: 5 = never the eight-wave list-major kernel, 6 = wherever it exists, 7 = its eight-query form wherever that exists)")
    ap.add_argument("--skew", action="store_true")
    ap.add_argument("--data", default="mixture", choices=["mixture", "lowrank"], help="trained configs: dataset")
    ap.add_argument("--no-next-hint", action="store_true",
                    help="do not tell the library which queries the next step searches (ivfadc_set_next_queries): every step then runs its "
                         "coarse search as a launch of its own instead of behind the previous step's scan")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; `value` is their median")
    ap.add_argument("--no-pruning", action="store_true", help="scan every probed list (ivfadc_set_pruning(h, 0)): the reference's own byte count")
    ap.add_argument("--single-mode", action="store_true",
                    help="one kernel population only: no same-run comparison legs (un-hinted, pruning off, table build alone), no sweep, no "
                         "other configs -- what the rocprofv3 passes are taken on")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief measurement of the other BASELINE.json shapes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-base", action="store_true", help="skip the per-rank rate in the N > 1 mode (one lane + the library's collective on a one-rank communicator)")
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the host-pointer entries' block (host vectors in, host vectors out)")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--single-process", action="store_true",
                    help="drive ivfadc_mg_search (the C ABI's own multi-device front end, host pointers) over --gpus devices")
    ap.add_argument("--mg-gather", default="rccl", choices=["host", "rccl"], help="--single-process: result merge")
    ap.add_argument("--inflight", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="batches in flight per GPU: 2 = steps alternate between the index and a read-only view of it (ivfadc_clone_view: same "
                         "device arrays, second stream and workspace), so one launch's ramp and tail overlap the next; 1 = one at a time; "
                         "0 (default) = 2 where it was measured to pay (query-major shapes whose launch is a few workgroups per CU: sift1m, hd, "
                         "toy), 1 on the billion-scale shapes (deep1b: -1 %%, sift1b: -11 %%: two batches' work items share the L2 badly)")
    ap.add_argument("--collective", default="native", choices=["native", "torch"],
                    help="multi-GPU result merge: the library's own ncclAllGather on a side stream of the handle "
                         "(ivfadc_search_device_allgather; a few us of host time per batch), or torch.distributed's")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="no GPU: run the launcher, the query partition and the gather over gloo with a stub searcher")
    ap.add_argument("--full", action="store_true",
                    help="also run the long legs: recall sweep over w (and the low-rank dataset), pruning-off / un-hinted windows, K = 1 / K = 100 / "
                         "skewed variants of the other shapes, the trained kc = 65 536 two-level block, every kind of caller memory on the host "
                         "path.  The default run measures what the compact line carries and nothing else")
    args = ap.parse_args()

    if args.single_process:
        return single_process(args)
    if args.lists_rehearsal:
        return lists_rehearsal(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:], args.selftest_cpu)          # does not return

    # The ONE JSON line must be the only thing on stdout: RCCL prints a version banner with printf (flushed at exit, after the
    # line), so file descriptor 1 is pointed at stderr for everything native and the line goes out through a private copy.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = not args.selftest_cpu
    if gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

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
    by_lists = args.partition == "lists"
    if by_lists:
        args.scaling = "strong"             # the global batch is the configuration's batch; every rank searches ALL of it over its own lists
        nq_total = nq
    elif args.scaling == "strong":
        nq_total = nq                         # the global batch is fixed; every rank gets an equal contiguous share of it
        if nq_total % world != 0:
            raise SystemExit("--scaling strong: the global batch (%d) must be a multiple of the number of GPUs (%d): the all-gather "
                             "moves equal blocks" % (nq_total, world))
        nq = nq_total // world
    else:
        nq_total = world * nq                 # ONE global batch per step, partitioned over the ranks
    lo, hi = (0, nq_total) if by_lists else shard_bounds(nq_total, world, rank)
    assert hi - lo == nq

    G = max(1, args.gather_every) if dist is not None else max(2, args.inflight)   # (no group: one result slot per lane)
    NR = 8 if dist is not None else 1     # (the library's collective entry waits for a slot's previous all-gather once per several steps when >= 8 slots rotate)
    rings = Rings(torch, dist, dev, world, nq, K, G, NR, gpu)

    synth_arrays = None
    x = None
    if gpu:
        import ivfadc_jl_amd as pkg
        if pkg.needs_build():
            pkg.build_library()
        pkg.load_library()
        if cfg["kind"] == "trained":
            idx, x = build_trained(pkg, cfg, dev, local_rank, dist, args.data)
        else:
            idx, synth_arrays = build_synth(pkg, cfg, local_rank, args.skew)
        q = global_queries(cfg, nq_total, dev, args.data)[lo:hi].contiguous()
        idx.set_tuning(args.qg, args.chunk)
        if args.coarse_mode:
            idx.set_coarse_mode(args.coarse_mode)
        if args.table_mode:
            idx.set_table_mode(args.table_mode)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
    else:
        idx = StubIndex(lo)
        q = None

    # one collective per batch, issued by the library itself: the ranks join an RCCL communicator of their own (the id
    # travels over torch.distributed), and a step is ONE C call -- search + ncclAllGather on the handle's side stream
    native_coll = gpu and dist is not None and (args.collective == "native" or by_lists) and G == 1
    if by_lists and not native_coll:
        raise SystemExit("--partition lists runs under a process group with the library's own collective (one batch per collective)")
    if native_coll:
        # every rank must end up on the same path: a rank that cannot set the communicator up (no librccl for dlopen, ...)
        # takes all of them back to torch.distributed's collective
        ok = 1
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        try:
            if rank == 0:
                idt.copy_(torch.as_tensor(pkg.comm_unique_id()))
        except Exception as e:           # noqa: BLE001
            log("[bench] native collective unavailable: %s" % e)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.broadcast(flag, 0)
        if int(flag.item()) == 1:
            dist.broadcast(idt, 0)
            try:
                idx.comm_init(world, rank, idt.cpu().numpy())
            except Exception as e:       # noqa: BLE001
                print("[bench] rank %d: ivfadc_comm_init failed: %s" % (rank, e), file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native_coll = int(flag.item()) == 1
        if by_lists and not native_coll:
            raise SystemExit("--partition lists: the library's RCCL communicator could not be set up on every rank")
    lp = None
    if by_lists:
        idx.set_list_partition(world, rank)
        words = int(pkg.load_library().ivfadc_listpart_block_words(nq, K))
        lp = {"block": torch.zeros(words, dtype=torch.int32, device=dev), "gath": torch.zeros(world * words, dtype=torch.int32, device=dev),
              "ids": torch.zeros(nq * K, dtype=torch.int32, device=dev), "dists": torch.zeros(nq * K, dtype=torch.float32, device=dev),
              "counts": torch.zeros(nq, dtype=torch.int32, device=dev)}

    hint_next = gpu and not args.no_next_hint
    single_mode = args.single_mode
    if gpu and args.no_pruning:
        idx.set_pruning(0)
    pruning_on = not (args.no_pruning or os.environ.get("IVFADC_NO_PRUNE"))
    # Two batches in flight: even steps on the index, odd steps on a view of it (taken after the settings above: a view copies them).
    # Only where a step's results have a buffer of their own per lane: one process without a group, or the library's own collective
    # (whose ring slots alternate); the profiled passes (--single-mode) and everything measured after the headline run one at a time.
    lane_list = [idx]
    # (under a process group the default is one batch at a time: with RCCL's and torch's streams on the device the lanes' streams end up sharing
    # hardware queues -- measured with a single-rank communicator: no overlap left, and twice the slot waits; --inflight 2 asks for it anyway)
    want_lanes = args.inflight if args.inflight else (2 if (args.config in TWO_LANE_CONFIGS and dist is None) else 1)
    if gpu and want_lanes >= 2 and not single_mode and not by_lists and (dist is None or (native_coll and G == 1)):
        for _ in range(want_lanes - 1 if dist is None else 1):
            lane_list.append(idx.clone_view())     # (without a group the result ring has one slot per lane: step i writes slot i % lanes)
    inflight_used = len(lane_list)

    def step(i):
        ln = lane_list[i % len(lane_list)]
        if hint_next:
            # a serving loop knows its next batch: its exact coarse tiles ride behind this step's scan launch (computed every step, by
            # the same kernel code; never cached) -- only plans with the rider form use it.  The batch buffer's contents never change
            # in this loop, so its generation token is a constant.  (Per lane: a lane's next batch is the one after the next.)
            ln.set_query_token(1)
            ln.set_next_queries(nq, q.data_ptr(), 1)
        if lp is not None:
            idx.search_device_listpart(nq, q.data_ptr(), K, w, lp["block"].data_ptr(), lp["gath"].data_ptr(), lp["ids"].data_ptr(),
                                       lp["dists"].data_ptr(), lp["counts"].data_ptr())
            rings.collectives += 1
            return
        if native_coll:
            r, _ = rings.slot_of(i)
            idx.search_device_allgather_on(ln, nq, q.data_ptr(), K, w, rings.ring[r].data_ptr(), rings.gath[r].data_ptr(), r)
            rings.collectives += 1
            return
        rings.before_step(i)
        view = rings.slot_view(i)
        if gpu:
            p_ids, p_d, p_c = rings.ptrs(view)
            ln.search_device(nq, q.data_ptr(), K, w, p_ids, p_d, p_c)
        else:
            idx.fill(view, nq, K)
        rings.after_step(i)

    def drain_all():
        if native_coll:
            idx.comm_wait()          # the search stream waits (on the device) for every collective in flight
        else:
            rings.drain()

    def sync():
        if gpu:
            torch.cuda.synchronize()

    def timed(nsteps):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        drain_all()
        sync()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def windows(nsteps, nwin):
        """nwin timed windows of exactly nsteps steps; (median, all)"""
        ws_ = [timed(nsteps) for _ in range(max(1, nwin))]
        return median_of(ws_), ws_

    for i in range(args.warmup):
        step(i)
    drain_all()
    sync()
    # Untimed settling: the W warm-up steps of a fast configuration last well under a millisecond, far too short for the
    # GPU to reach its sustained clock (SIFT1M-shape: 72 us per step in a cold 100-step run, 64 us once warm).  Keep
    # issuing untimed steps until ~0.1 s of them have run; every timed window below is still exactly K steps.
    t_settle = time.perf_counter()
    i = args.warmup
    while gpu:
        for _ in range(16):
            step(i)
            i += 1
        drain_all()
        sync()
        done = time.perf_counter() - t_settle >= 0.1
        if dist is not None:
            # every rank must issue the same number of steps (they carry collectives): all stop together
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break

    coll0 = rings.collectives
    elapsed, wins = windows(args.steps, args.windows)
    coll_timed = (rings.collectives - coll0) // max(1, len(wins))
    # both lanes searched the same batch into their own result slots: the blocks must be identical, bit for bit
    lanes_agree = None
    if gpu and len(lane_list) == 2 and dist is None and rings.G >= 2:
        lanes_agree = bool(torch.equal(rings.slot_view(0), rings.slot_view(1)))
    qps = nq_total * args.steps / elapsed
    win_info = {"n": len(wins), "steps_each": args.steps, "value_is": "median window",
                "ms_per_step_min": round(min(wins) / args.steps * 1e3, 4), "ms_per_step_max": round(max(wins) / args.steps * 1e3, 4),
                "qps_min": round(nq_total * args.steps / max(wins), 1), "qps_max": round(nq_total * args.steps / min(wins), 1)}
    # the same steps without the next-batch hint (every step's coarse search as a launch of its own), same run: reported beside `value`
    hint_info = None
    if gpu:
        st_h = idx.get_stats()
        hint_used = bool(hint_next and st_h.get("coarse_prefetched", 0))
        hint_info = {"hinted": bool(hint_next), "used_by_this_plan": hint_used,
                     "what": "ivfadc_set_next_queries (+ content token) before every step: the next step's exact coarse tiles ride behind this "
                             "step's scan launch (recomputed every step, never cached; results bit-identical). The Julia shim reaches the same "
                             "path through knn_search(ivfadc, batches, k) -> ivfadc_search_batches; `without_hint_same_run` is the plain "
                             "knn_search-per-batch contract"}
        def variant(hinted):
            nonlocal hint_next
            keep = hint_next
            hint_next = hinted
            for i in range(20):
                step(i)
            el_v, w_v = windows(args.steps, args.windows)
            hint_next = keep
            return {"qps": round(nq_total * args.steps / el_v, 1), "ms_per_step": round(el_v / args.steps * 1e3, 4), "windows": len(w_v),
                    "qps_min": round(nq_total * args.steps / max(w_v), 1), "qps_max": round(nq_total * args.steps / min(w_v), 1)}

        if hint_used and dist is None and not single_mode and args.full:
            hint_info["without_hint_same_run"] = variant(False)
    # one batch in flight, same run: the index alone, a step waits for nothing but the stream order (with the hint, and as plain
    # knn_search-per-batch calls -- the reference's own calling pattern)
    inflight_info = None
    if gpu:
        inflight_info = {"inflight": inflight_used, "results_identical_across_lanes": lanes_agree,
                         "what": "steps alternate between the index and a read-only view of it (ivfadc_clone_view: the same device arrays, a "
                                 "second stream and workspace): a launch's ramp and tail overlap the neighbouring batches' kernels; every batch is "
                                 "searched whole by the same kernels, results bit-identical.  ivfadc_search_batches does the same inside the "
                                 "library, which is how knn_search(ivfadc, batches, k) of the Julia shim reaches it" if inflight_used == 2 else
                                 "one batch at a time"}
        if inflight_used == 2 and dist is None and not single_mode:
            lanes_keep = lane_list
            lane_list = lane_list[:1]
            inflight_info["one_in_flight_same_run"] = variant(hint_next)
            if hint_info is not None and hint_info.get("used_by_this_plan"):
                inflight_info["one_in_flight_without_hint_same_run"] = variant(False)
            lane_list = lanes_keep
    # everything below (profiled launches, pruning off, sweeps, parity) runs one batch at a time on the index itself
    lane_list = lane_list[:1]
    if gpu:
        for i in range(2):           # the steps below (profiling) start from the steady state of the index's own lane again
            step(i)
        sync()

    # ---- multi-rank checks: who RCCL saw, and that every rank's gathered copy of the last batch is what the owners hold
    dist_info = None
    if dist is not None:
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        last = args.steps - 1
        rr, sl = rings.slot_of(last)
        # (the timed loop drained: the last ring has been gathered whole)
        per = nq * rings.width
        mine = rings.ring[rr][sl * per:(sl + 1) * per]
        blk = G * per
        ok = True
        if lp is not None:
            # every rank merged the same gathered keys: all ranks must hold the same ids / distances / counts
            torch.cuda.synchronize()
            for name in ("ids", "dists", "counts"):
                exp = lp[name].clone()
                dist.broadcast(exp, 0)
                ok = ok and bool(torch.equal(lp[name].view(torch.int32), exp.view(torch.int32)))
            blk = lp["block"].numel()
        for r in range(world if lp is None else 0):
            got = rings.gath[rr][r * blk + sl * per: r * blk + (sl + 1) * per]
            exp = mine.clone()
            dist.broadcast(exp, r)                      # rank r's own block of the batch
            ok = ok and bool(torch.equal(got, exp))
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        dist_info = {"ranks_seen_by_rccl": int(ones.item()), "gather_check": bool(okt.item()),
                     "collectives_in_timed_region": coll_timed, "timed_region": "one window of --steps steps", "batches_per_collective": G,
                     "bytes_per_rank_per_collective": blk * 4,
                     "backend": ("RCCL, ncclAllGather issued by libivfadc_hip.so (ivfadc_search_device_allgather)" if native_coll
                                 else "RCCL through torch.distributed") if gpu else "gloo (CPU self-test)"}
        if not gpu:
            # stub pattern: slot q of the gathered batch must carry GLOBAL query number q
            full = torch.cat([rings.gath[rr][r * blk + sl * per: r * blk + sl * per + nq * K] for r in range(world)])
            dist_info["partition_check"] = bool(torch.equal(full, torch.arange(nq_total * K, dtype=torch.int32)))

    if not gpu:
        if rank == 0:
            emit({"metric": "launcher/partition/gather self-test (no GPU, no search)", "value": round(qps, 1),
                  "unit": "stub batches x queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "vs_baseline": None, "dtype": "none", "data": "stub",
                  "selftest_cpu": True, "distributed": dist_info, "scaling": args.scaling, "windows": win_info,
                  "gather_check": dist_info["gather_check"] if dist_info else None,
                  "ranks_seen_by_rccl": dist_info["ranks_seen_by_rccl"] if dist_info else None,
                  "config": {"workload": "stub", "global_batch": nq_total, "queries_per_rank": nq, "partition": args.partition}}, json_out)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (list scan): HIP events on the launch stream, live.  With the next-batch hint the scan launch
    # also carries the next step's coarse tiles; its duration is recorded, then (unless --single-mode) the hint is dropped so that the
    # roofline below (and the pruning-off / table-build measurements) describe the scan kernel alone
    prof_steps = max(1, min(args.steps, 50))

    def profiled(nst, level=True):
        idx.set_profiling(level)
        idx.reset_stats()
        el = timed(nst)
        st_ = idx.get_stats()
        idx.set_profiling(False)
        return el, st_

    if hint_info is not None and hint_info.get("used_by_this_plan") and not single_mode:
        _, st_r = profiled(prof_steps)
        hint_info["scan_launch_with_riders_ms"] = round(st_r["scan_ms"] / max(1, st_r["scan_launches"]), 5)
        hint_next = False
        step(0)          # uses up the rows the last hinted step left
        drain_all()
        sync()
    el_prof, st = profiled(prof_steps)
    roofline = roofline_of(args.config, cfg, nq, w, K, st, pruning_on, riders=bool(hint_next and st.get("last_rider", 0)))
    roofline["profiled_ms_per_step"] = round(el_prof / prof_steps * 1e3, 4)
    pruned_frac = roofline["pruned_fraction_of_sec8d_bytes"]
    balg_sec8d = roofline["sec8d_alg_bytes_per_launch"]
    scan_ms = roofline["scan_ms_per_launch"]
    # Exact probe pruning (ivfadc_set_pruning, on by default): lists whose coarse distance already exceeds the K-th best key are
    # not read.  The roofline prices the bytes ACTUALLY scanned in the timed configuration; SURVEY 8(d)'s B_alg (every point
    # of every probed list, which is what the reference algorithm reads) and the same measurement with pruning off are reported
    # next to it, so that nothing is counted that the kernel did not do.
    no_prune = None
    if world == 1 and pruned_frac > 0 and not single_mode:
        idx.set_pruning(0)
        _, st0 = profiled(prof_steps)
        el0, w0 = windows(args.steps, args.windows if args.full else 1)
        scan_ms0 = st0["scan_ms"] / max(1, st0["scan_launches"])
        no_prune = {"qps": round(nq_total * args.steps / el0, 1), "ms_per_step": round(el0 / args.steps * 1e3, 4), "windows": len(w0),
                    "scan_ms_per_launch": round(scan_ms0, 5), "alg_bytes_per_launch": int(balg_sec8d),
                    "frac": round(balg_sec8d / (scan_ms0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if scan_ms0 > 0 else None}
        idx.set_pruning(1)
    roofline["pruning"] = {"pruned_fraction_of_sec8d_bytes": round(pruned_frac, 4), "sec8d_alg_bytes_per_launch": int(balg_sec8d),
                           "frac_if_sec8d_bytes_were_counted": round(balg_sec8d / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if scan_ms > 0 else None,
                           "pruning_off_same_run": no_prune,
                           "note": "exact (a sum starts from its list's coarse distance and only grows, index.jl:242-244; probes ascend): "
                                   "bit-identical on/off; achieved / frac count only the bytes actually scanned"}
    # ADC tables on the matrix cores (lower-bound tables, lbscan.hip.h): the build timed ALONE -- an extra launch of the same build
    # code over the same probes (ivfadc_set_profiling(h, 2)) -- against the MFMA peak of the form it uses; flops = SURVEY 8(d)'s
    # 2 k d per (query, probe), and what the split-bf16 form executes (three bf16 products per f32 product)
    if st.get("last_lb", 0) and world == 1 and not single_mode:
        _, stb = profiled(max(1, min(args.steps, 10)), 2)
        if stb.get("lb_build_launches", 0) > 0:
            tb_ms = stb["lb_build_ms"] / stb["lb_build_launches"]
            flops = 2.0 * 256 * cfg["d"] * nq * cfg["w"]
            roofline["table_build"] = {
                "kernel": "lb_build_only_kernel (the table build of qscan_kernel<..., LB> run alone: same code, same LDS footprint)",
                "ms_per_launch": round(tb_ms, 5), "alg_gflop_per_launch": round(flops / 1e9, 3),
                "achieved": round(flops / (tb_ms * 1e-3) / 1e12, 3), "executed": round(3 * flops / (tb_ms * 1e-3) / 1e12, 3),
                "peak": 2500.0, "unit": "TFLOP/s (bf16 MFMA, dense)", "frac": round(3 * flops / (tb_ms * 1e-3) / 1e12 / 2500.0, 5),
                "bound": "load latency of the codewords, not the matrix pipe (DESIGN.md 4.4b)",
                "codebook_bytes_per_launch": int(cfg["d"] * 256 * 4 * nq * ((cfg["w"] + 3) // 4)),
                "survivors_per_query": round(st.get("lb_survivors", 0) / max(1, st["queries"]), 2)}

    # ---- results of the last step: recall (trained configs) and oracle spot-check
    last_i = prof_steps - 1

    def results_of(i):
        if lp is not None:
            return lp["ids"].view(nq, K), lp["dists"].view(nq, K), lp["counts"]
        res = rings.slot_view(i)
        return (res[:nq * K].view(nq, K), res[nq * K:2 * nq * K].view(torch.float32).view(nq, K), res[2 * nq * K:])

    for i in range(prof_steps):          # leave the buffers holding results of the headline configuration
        step(i)
    drain_all()
    sync()
    ids, dists, counts = results_of(last_i)
    recall = recall_at_1(x, q, ids, counts) if x is not None else None

    def run_w(ws, nsw):
        def step_w(i):
            p_ids, p_d, p_c = rings.ptrs(rings.slot_view(i & 1))
            idx.search_device(nq, q.data_ptr(), K, ws, p_ids, p_d, p_c)
        for i in range(min(20, nsw)):
            step_w(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsw):
            step_w(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        r_ids, _, r_counts = results_of(1 if nsw > 1 else 0)
        return nq * nsw / el, (recall_at_1(x, q, r_ids, r_counts) if x is not None else None)

    sweep = None
    recall_ceiling = None
    if rank == 0 and world == 1 and dist is None and cfg["kind"] == "trained" and not args.w and not args.no_sweep and not single_mode and args.full:
        # the reference default is w=1; BASELINE.md asks for w in {1, 8, 32}; w = kc scans every list: the recall the
        # product quantizer itself allows (the ceiling no choice of w can beat).  Un-hinted plain searches.
        sweep = {}
        nsw = max(100, min(args.steps, 1000))
        for ws in (1, 8, 32):
            qps_w, rec = run_w(ws, nsw)
            sweep["w=%d" % ws] = {"qps": round(qps_w, 1), "recall_at_1_in_top%d" % K: rec}
        qps_c, recall_ceiling = run_w(cfg["kc"], 3)
        sweep["w=kc=%d (PQ ceiling)" % cfg["kc"]] = {"qps": round(qps_c, 1), "recall_at_1_in_top%d" % K: recall_ceiling}
        if args.config == "sift1m" and args.data == "mixture":
            # second dataset, same shape, with structure PQ can use: recall moves with w there
            cfg2 = dict(cfg)
            idx2, x2 = build_trained(pkg, cfg2, dev, local_rank, None, "lowrank")
            q2 = global_queries(cfg2, nq, dev, "lowrank")
            idx2.set_stream(torch.cuda.current_stream().cuda_stream)
            low = {}
            for ws in (1, 8, 32, cfg["kc"]):
                nrep = 50 if ws < cfg["kc"] else 2
                p_ids, p_d, p_c = rings.ptrs(rings.slot_view(0))
                for _ in range(3):
                    idx2.search_device(nq, q2.data_ptr(), K, ws, p_ids, p_d, p_c)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nrep):
                    idx2.search_device(nq, q2.data_ptr(), K, ws, p_ids, p_d, p_c)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                r_ids, _, r_counts = results_of(0)
                low["w=%d" % ws if ws < cfg["kc"] else "w=kc=%d (PQ ceiling)" % ws] = {
                    "qps": round(nq * nrep / el, 1), "recall_at_1_in_top%d" % K: recall_at_1(x2, q2, r_ids, r_counts)}
            sweep["lowrank dataset (64 centres, rank-16 within-cluster spread, same shape)"] = low
            del idx2, x2, q2
        # leave the buffers holding the headline-w results for the checks below
        for i in range(prof_steps):
            step(i)
        drain_all()
        torch.cuda.synchronize()
        ids, dists, counts = results_of(last_i)

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
        # threads: the host's logical CPUs, or the share of them this process is allowed to use for longer than a burst (cgroup quota:
        # 128 threads finish one 1024-query batch in milliseconds, unthrottled, and then run at the quota's rate -- measured 280 k against
        # 38 k queries/s on a 16-CPU share)
        cores = ora.max_threads()
        host_cpus = cores
        try:
            qt, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if qt != "max":
                cores = max(1, min(cores, int(-(-int(qt) // int(per)))))
        except (OSError, ValueError):
            pass
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
        # about 20 CPU-seconds in all: when the whole batch takes the host's cores only milliseconds, the sample is the batch again and again
        reps = int(max(1, min(200, round(20.0 / max(t_mt * cores, 1e-6)))))
        if reps > 1:
            t0 = time.perf_counter()
            for _ in range(reps):
                oidx.knn_search(qh[:ns], K, w, nthreads=cores)
            t_mt = (time.perf_counter() - t0) / reps
        n1 = max(1, min(ns, int(ns * 3.0 / max(t_mt * cores, 1e-3))))
        t0 = time.perf_counter()
        oidx.knn_search(qh[:n1], K, w, nthreads=1)
        t_1 = time.perf_counter() - t0
        cpu_baseline = {"value": round(ns / t_mt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                        "sample": "%s first %d queries of the batch (%.1f s of wall time on %d threads), same index arrays, oracle/ivfadc_oracle.c "
                                  "(gcc -O2 -ffp-contract=off), OpenMP over queries" % (("%d x the" % reps) if reps > 1 else "the", ns, t_mt * reps, cores),
                        "single_thread_qps": round(n1 / t_1, 2), "single_thread_sample": n1, "host_logical_cpus": host_cpus}
        gi = ids[:ns].cpu().numpy().view(np.uint32)
        gd = dists[:ns].cpu().numpy()
        gc = counts[:ns].cpu().numpy()
        ok_ids = bool(np.array_equal(gc, oc) and all(np.array_equal(gi[r, :gc[r]], oi[r, :oc[r]]) for r in range(ns)))
        ok_d = bool(all(np.allclose(gd[r, :gc[r]], od[r, :oc[r]], rtol=1e-4, atol=0) for r in range(ns)))
        parity = {"queries_checked": ns, "ids_bit_exact": ok_ids, "dists_rtol_1e-4": ok_d}

    # ---- the base of a like-for-like scaling curve: this rank's rate in the mode N > 1 runs take (measured on every rank of every run;
    # a rank's own one-rank communicator).  N = 1: reported as `scaling_base`; N > 1: the line also carries value / (N x base)
    scaling_base = None
    if gpu and not single_mode and not by_lists and not args.no_scaling_base and (dist is None or native_coll):
        try:
            if native_coll:
                idx.comm_wait()
                torch.cuda.synchronize()
                idx.comm_destroy()          # (the run's own communicator has done its work: checks above, timing long before)
            scaling_base = measure_scaling_base(torch, pkg, idx, nq, q, K, w, dev, args.steps, args.windows, bool(gpu and not args.no_next_hint))
        except Exception as e:           # noqa: BLE001
            scaling_base = {"error": "%s: %s" % (type(e).__name__, e)}
        if dist is not None:
            # (collective: EVERY rank gets here, whether its own measurement succeeded or not -- a rank that failed contributes 0 and the
            # efficiency is then left out, instead of the other ranks waiting for it for ever)
            mine = float(scaling_base.get("qps_per_rank", 0.0))
            t = torch.tensor([mine], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if float(t.item()) > 0.0 and "error" not in scaling_base:
                scaling_base["qps_per_rank_min_over_ranks"] = round(float(t.item()), 1)
                scaling_base["efficiency_vs_scaling_base"] = round(qps / (world * float(t.item())), 4)

    # ---- the reference's own contract: host vectors in, host vectors out (trained single-GPU configurations)
    host_to_host = None
    if rank == 0 and world == 1 and dist is None and cfg["kind"] == "trained" and not single_mode and not args.no_host_to_host:
        try:
            host_to_host = measure_host_to_host(torch, pkg, idx, cfg, K, w, dev, local_rank, args.data,
                                                kinds=("pageable", "registered", "library_pinned") if args.full else ("registered",))
        except Exception as e:           # noqa: BLE001  (a failure here must not cost the headline line)
            host_to_host = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the other BASELINE.json shapes, briefly (single GPU, default workload only): driver-witnessed step times and roofline fractions
    other = None
    if rank == 0 and world == 1 and dist is None and args.config == "sift1m" and not single_mode and not args.no_other_configs \
            and not (args.nq or args.n or args.kc or args.w or args.qg or args.chunk):
        other = {}
        t_o = time.perf_counter()
        # (the reference is generic in k, index.jl:204-208: K = 1 and K = 100 on the Deep1B shape -- K > 64 leaves the register selectors --
        # and its lists are as uneven as its data: the SIFT1B shape once more with skewed list sizes)
        # (... and one rank's share of the SIFT1B configuration's 8-GPU batch: 16 384 / 8 = 2048 queries against the full replica)
        shapes = (("deep1b", ((32, K), (32, 1), (32, 100)), False), ("hd", ((8, K),), False),
                  ("sift1b", ((8, K), (1, K), (8, K, 2048)), False), ("sift1b", ((8, K),), True)) if args.full else \
                 (("deep1b", ((32, K),), False), ("hd", ((8, K),), False), ("sift1b", ((8, K), (1, K), (8, K, 2048)), False))
        for name, cases, skew in shapes:
            try:
                for ent in measure_other_config(torch, pkg, name, cases, dev, local_rank, skew=skew, two_lanes=args.full):
                    tag = "%s w=%d" % (name, ent["w"]) + ("" if ent["K"] == K else " K=%d" % ent["K"]) + (" skewed" if skew else "") + \
                          ("" if ent["batch"] == CONFIGS[name]["nq"] else " batch=%d (one rank's share of the 8-GPU batch)" % ent["batch"])
                    other[tag] = ent
            except Exception as e:           # noqa: BLE001  (a failure here must not cost the headline line)
                other["%s%s (failed)" % (name, " skewed" if skew else "")] = {"error": "%s: %s" % (type(e).__name__, e)}
        other["seconds_total"] = round(time.perf_counter() - t_o, 1)
        try:
            if args.full:
                t_tl = time.perf_counter()
                other["two_level_coarse (trained kc=65536)"] = measure_two_level(torch, pkg, K, dev, local_rank)
                other["two_level_coarse (trained kc=65536)"]["seconds"] = round(time.perf_counter() - t_tl, 1)
        except Exception as e:           # noqa: BLE001
            other["two_level_coarse (failed)"] = {"error": "%s: %s" % (type(e).__name__, e)}

    rv = roofline.get("roofline_valu")
    if rv and not st.get("coarse_mfma", 0) and not st.get("last_twolevel", 0) and elapsed > 0:
        # the whole step against the same peak: the exact coarse search (one launch per batch, or riding in the scan launch: then it is in
        # lane_ops_per_launch already) is vector arithmetic of the same kind, nq x kc x d x (sub, mul, add)
        lo = rv["lane_ops_per_launch"]
        ops_step = lo["tables"] + lo["scan_adds"] + float(nq) * cfg["kc"] * cfg["d"] * 3
        t_step = elapsed / args.steps
        rv["step"] = {"lane_ops": int(ops_step), "ms_per_step": round(t_step * 1e3, 4), "achieved": round(ops_step / t_step / 1e12, 2),
                      "frac": round(ops_step / t_step / 1e12 / VALU_PEAK_TOPS, 4),
                      "what": "exact coarse distances + ADC tables of the scanned probes + scan adds of one batch / the step time of the timed mode "
                              "(two batches in flight overlap one batch's coarse launch with the other's scan)"}
    if rank == 0:
        line = {
            "metric": "queries/sec at recall@1 (k=10), SIFT1M-shape d=128 m=8 k=256, 1/2/4/8 GPU"
                      if args.config == "sift1m" else "queries/sec, %s-shape, K=%d" % (args.config, K),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"global_batch": nq_total, "queries_per_rank": nq, "workload": "%s-shape: d=%d n=%d kc=%d k=256 m=%d UInt8 codes, batch=%d queries/GPU (global batch %d), K=%d, w=%d%s"
                                   % (args.config, cfg["d"], cfg["n"], cfg["kc"], cfg["m"], nq, nq_total, K, w,
                                      ", skewed lists" if args.skew else ""),
                       "index": ("trained (k-means + PQ, 25 iters), %s data" % ("Gaussian-mixture" if args.data == "mixture" else "low-rank mixture"))
                                if cfg["kind"] == "trained" else "device-synthesised codes, N(0,1) quantizers",
                       "parallelism": (("one global batch of %d queries per step, every one of the %d GPUs searches ALL of it over its own lists (l %% %d == rank), "
                                        "index replicated, 1 all-gather of the partial top-k keys per batch + K-way merge on every rank" % (nq_total, world, world))
                                       if by_lists else
                                       ("one global batch of %d queries per step partitioned over %d GPUs (contiguous blocks), index "
                                        "replicated, %s" % (nq_total, world, "1 all-gather of the packed top-k per batch" if G == 1 else
                                                            "1 all-gather per %d batches (--gather-every)" % G)))
                                      if world > 1 else "1 GPU",
                       "partition": args.partition,
                       "pruning": pruning_on, "single_mode": single_mode, "batches_in_flight": inflight_used,
                       "recall_at_1_in_top%d" % K: recall, "recall_ceiling_w=kc": recall_ceiling,
                       "recall_note": "the BASELINE dataset (isotropic Gaussian mixture, sigma = 0.1 in 128 dimensions) leaves an 8-byte product "
                                      "quantizer nothing to use: recall sits at the PQ ceiling (w = kc) for every w, so 'at recall@1' carries no "
                                      "information here; the recall-vs-w curve is the one of sweep['lowrank dataset ...'] (same shape, structured residuals)"
                                      if (cfg["kind"] == "trained" and args.data == "mixture") else None},
            "windows": win_info,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity, "next_batch_hint": hint_info, "batches_in_flight": inflight_info, "host_to_host": host_to_host, "scaling_base": scaling_base, "other_configs": other,
            "sweep": sweep,
        }
        if isinstance(scaling_base, dict) and "efficiency_vs_scaling_base" in scaling_base:
            line["efficiency_vs_scaling_base"] = scaling_base["efficiency_vs_scaling_base"]
        if dist_info is not None:
            line["distributed"] = dist_info
            line["gather_check"] = dist_info["gather_check"]
            line["ranks_seen_by_rccl"] = dist_info["ranks_seen_by_rccl"]
        emit(line, json_out)
    if dist is not None:
        dist.destroy_process_group()


def lists_rehearsal(args):
    """`--lists-rehearsal N` on ONE GPU: what `--partition lists` does on N GPUs, piece by piece.  A rank of the N-GPU run does exactly
    what this GPU does with ivfadc_set_list_partition(h, N, r): the coarse search of ALL queries, the scan of the probed lists l with
    l % N == r, then (after the all-gather, which one GPU cannot time) the N-way merge.  Prints one JSON line."""
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import ivfadc_jl_amd as pkg
    if pkg.needs_build():
        pkg.build_library()
    pkg.load_library()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N = args.lists_rehearsal
    cfg = dict(CONFIGS[args.config])
    for k_, v in (("nq", args.nq), ("n", args.n), ("kc", args.kc), ("w", args.w)):
        if v:
            cfg[k_] = v
    if cfg["kind"] != "synth":
        raise SystemExit("--lists-rehearsal runs the device-synthesised shapes: --config sift1b")
    K, w, nq = args.K, cfg["w"], cfg["nq"]
    idx, (cent, cbs, labels, off) = build_synth(pkg, cfg, 0, args.skew)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_tuning(args.qg, args.chunk)
    q = global_queries(cfg, nq, dev).contiguous()
    ids = torch.zeros(nq * K, dtype=torch.int32, device=dev)
    dists = torch.zeros(nq * K, dtype=torch.float32, device=dev)
    counts = torch.zeros(nq, dtype=torch.int32, device=dev)
    keys_all = torch.zeros((N, nq, K), dtype=torch.int64, device=dev)
    cnts_all = torch.zeros((N, nq), dtype=torch.int32, device=dev)

    def timed(fn, nsteps, nwin):
        for _ in range(max(2, args.warmup)):
            fn()
        wins = []
        for _ in range(nwin):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                fn()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / nsteps)
        return median_of(wins), wins

    nsteps = max(3, min(args.steps, 20))
    t_full, _ = timed(lambda: idx.search_device(nq, q.data_ptr(), K, w, ids.data_ptr(), dists.data_ptr(), counts.data_ptr()), nsteps, args.windows)
    full = (ids.cpu().numpy().view(np.uint32).reshape(nq, K).copy(), dists.cpu().numpy().reshape(nq, K).copy(), counts.cpu().numpy().copy())
    t_slice = []
    for r in range(N):
        idx.set_list_partition(N, r)
        t, _ = timed(lambda: idx.search_device_partial(nq, q.data_ptr(), K, w, keys_all[r].data_ptr(), cnts_all[r].data_ptr()), nsteps, args.windows)
        t_slice.append(t)
    # (the handle's probe arrays are those of the last partial search -- the same batch, as on every rank)
    t_merge, _ = timed(lambda: idx.merge_partials_device(nq, K, N, keys_all.data_ptr(), cnts_all.data_ptr(), ids.data_ptr(), dists.data_ptr(),
                                                       counts.data_ptr()), nsteps, args.windows)
    torch.cuda.synchronize()
    merged = (ids.cpu().numpy().view(np.uint32).reshape(nq, K), dists.cpu().numpy().reshape(nq, K), counts.cpu().numpy())
    same = bool(np.array_equal(merged[2], full[2]) and np.array_equal(merged[0], full[0]) and np.array_equal(merged[1], full[1]))
    parity = None
    if not args.no_cpu_baseline:
        from oracle import oracle as ora
        oidx = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=20260101)
        pick = np.sort(np.random.default_rng(5).choice(nq, min(64, nq), replace=False))
        parity = oracle_parity(ora, oidx, q.cpu().numpy(), K, w, merged[0], merged[1], merged[2], pick)
    ag_bytes = int(pkg.load_library().ivfadc_listpart_block_words(nq, K)) * 4
    ag_assumed_ms = 0.02 + (N - 1) * ag_bytes / 100e9 * 1e3      # assumption: 20 us of latency + ring all-gather at 100 GB/s bus bandwidth
    worst = max(t_slice)
    line = {"metric": "one-GPU rehearsal of --partition lists on %d GPUs, %s-shape, K=%d" % (N, args.config, K), "n_gpus": 1,
            "config": {"workload": "%s-shape: d=%d n=%d kc=%d m=%d, global batch %d, K=%d, w=%d" % (args.config, cfg["d"], cfg["n"], cfg["kc"], cfg["m"], nq, K, w)},
            "t_full_ms": round(t_full * 1e3, 4), "t_slice_ms": [round(t * 1e3, 4) for t in t_slice], "t_merge_ms": round(t_merge * 1e3, 4),
            "allgather": {"bytes_per_rank": ag_bytes, "assumed_ms": round(ag_assumed_ms, 4),
                          "assumption": "not measurable on one GPU: 20 us + (N-1) x block / 100 GB/s"},
            "predicted_speedup_at_%d_gpus" % N: round(t_full / (worst + t_merge + ag_assumed_ms * 1e-3), 3),
            "predicted_speedup_without_allgather": round(t_full / (worst + t_merge), 3),
            "predicted_qps": round(nq / (worst + t_merge + ag_assumed_ms * 1e-3), 1),
            "merged_equals_full_search": same, "parity": parity,
            "note": "a rank of the N-GPU run does what this GPU does per slice: coarse search of all queries, scan of its lists, merge; the curve itself is the driver's to measure"}
    print(json.dumps(line), file=json_out, flush=True)


def single_process(args):
    """`--single-process`: the C ABI's own multi-device front end (ivfadc_mg_*): one host process, one index replica per
    device, ivfadc_mg_search splits every batch into contiguous blocks and (with --mg-gather rccl) merges the packed
    results with ONE ncclAllGather issued by the library.  Host pointers in and out: the rate INCLUDES the PCIe copies,
    so this line is never the headline `value` of the HBM-resident contract; it shows the in-library RCCL path working."""
    import ctypes as C
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")   # see main(): native code (RCCL's banner) must not share stdout with the JSON line
    os.dup2(2, 1)
    import torch
    ndev = args.gpus
    have = torch.cuda.device_count()
    if have < ndev:
        raise SystemExit("bench.py --single-process --gpus %d: only %d GPU(s) visible" % (ndev, have))
    import ivfadc_jl_amd as pkg
    from ivfadc_jl_amd import _native as nat
    if pkg.needs_build():
        pkg.build_library()
    L = pkg.load_library()
    cfg = dict(CONFIGS[args.config if args.config != "sift1m" or args.n else "sift1b"]) if False else dict(CONFIGS[args.config])
    if cfg["kind"] != "synth":
        raise SystemExit("--single-process runs the device-synthesised shapes (deep1b, sift1b, hd): --config sift1b")
    for k_, v in (("nq", args.nq), ("n", args.n), ("kc", args.kc), ("w", args.w)):
        if v:
            cfg[k_] = v
    K, w = args.K, cfg["w"]
    nq_total = cfg["nq"] * ndev
    cent, cbs, labels = synth_quantizers(cfg)
    off = synth_sizes(cfg["n"], cfg["kc"], 7, args.skew)
    g = C.c_void_p()
    devs = np.arange(ndev, dtype=np.int32)
    nat.check(L.ivfadc_mg_create(C.byref(g), ndev, nat.ptr(devs, C.c_int32), cfg["d"], cfg["kc"], cfg["m"], 256,
                                 nat.ptr(cent, C.c_float), nat.ptr(cbs, C.c_float), nat.ptr(labels, C.c_uint8)))
    try:
        t0 = time.time()
        nat.check(L.ivfadc_mg_synth_lists(g, nat.ptr(off, C.c_int64), C.c_uint64(20260101)))
        log("[bench] %d replicas synthesised in %.1fs" % (ndev, time.time() - t0))
        nat.check(L.ivfadc_mg_set_gather(g, 1 if args.mg_gather == "rccl" else 0))
        q = np.random.default_rng(11).standard_normal((nq_total, cfg["d"]), dtype=np.float32)
        ids = np.zeros((nq_total, K), np.uint32)
        dists = np.zeros((nq_total, K), np.float32)
        counts = np.zeros(nq_total, np.int32)

        def step():
            nat.check(L.ivfadc_mg_search(g, nq_total, nat.ptr(q, C.c_float), K, w, nat.ptr(ids, C.c_uint32),
                                         nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        for _ in range(args.warmup):
            step()
        c0 = C.c_int64(0)
        nat.check(L.ivfadc_mg_collectives(g, C.byref(c0)))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        el = time.perf_counter() - t0
        c1 = C.c_int64(0)
        nat.check(L.ivfadc_mg_collectives(g, C.byref(c1)))
        parity = None
        if not args.no_cpu_baseline:
            from oracle import oracle as ora
            oidx = ora.OracleIndex(cent, cbs, labels, off, None, None, synth_seed=20260101)
            pick = np.sort(np.random.default_rng(3).choice(nq_total, min(64, nq_total), replace=False))
            oi, od, oc = oidx.knn_search(q[pick], K, w, nthreads=ora.max_threads())
            parity = {"queries_checked": int(pick.shape[0]),
                      "ids_bit_exact": bool(np.array_equal(counts[pick], oc) and np.array_equal(ids[pick], oi)),
                      "dists_rtol_1e-4": bool(np.allclose(dists[pick], od, rtol=1e-4, atol=0))}
        print(json.dumps({
            "metric": "queries/sec, %s-shape, K=%d, single-process multi-device front end (host pointers: PCIe-inclusive)" % (args.config, K),
            "value": round(nq_total * args.steps / el, 1), "unit": "queries/s", "n_gpus": ndev, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape: d=%d n=%d kc=%d k=256 m=%d, global batch %d (%d per GPU), K=%d, w=%d"
                                   % (args.config, cfg["d"], cfg["n"], cfg["kc"], cfg["m"], nq_total, cfg["nq"], K, w),
                       "parallelism": "ivfadc_mg_search over %d device(s), index replicated, merge = %s"
                                      % (ndev, "1 ncclAllGather per batch inside the library" if args.mg_gather == "rccl" else "host gather")},
            "collectives_in_timed_region": int(c1.value - c0.value), "parity": parity}), file=json_out, flush=True)
    finally:
        L.ivfadc_mg_destroy(g)


if __name__ == "__main__":
    main()
