"""Multi-GPU knn_search: queries partitioned across ranks, index replicated per GPU.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
on CPU for tests).  Queries are independent (/root/reference/src/index.jl:269-271), so the
only exchange is the final gather of each rank's top-k block -- one collective per batch of
packed [nq_local, K] (id, dist) + counts; no reduction is needed because ranks own disjoint
queries.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(nq, world_size, rank):
    """Contiguous query block of `rank`: sizes differ by at most one."""
    base, rem = divmod(int(nq), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(ids, dists, counts):
    """(nq, K) uint32, (nq, K) float32, (nq,) int32 -> one int32 tensor (nq, 2K+1) for a single collective."""
    t_ids = torch.as_tensor(np.ascontiguousarray(ids).view(np.int32)) if isinstance(ids, np.ndarray) else ids.view(torch.int32)
    t_d = torch.as_tensor(np.ascontiguousarray(dists).view(np.int32)) if isinstance(dists, np.ndarray) else dists.view(torch.int32)
    t_c = torch.as_tensor(np.ascontiguousarray(counts)) if isinstance(counts, np.ndarray) else counts
    return torch.cat([t_ids, t_d, t_c.view(-1, 1).to(torch.int32)], dim=1).contiguous()


def unpack_results(packed, K):
    ids = packed[:, :K].contiguous()
    dists = packed[:, K:2 * K].contiguous().view(torch.float32)
    counts = packed[:, 2 * K].contiguous()
    return ids, dists, counts


def gather_results(packed_local, nq_total, group=None):
    """All-gather the per-rank packed blocks (ragged by at most one row) into (nq_total, 2K+1)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = packed_local.shape[1]
    base, rem = divmod(int(nq_total), world)
    maxrows = base + (1 if rem else 0)
    pad = torch.zeros((maxrows, width), dtype=packed_local.dtype, device=packed_local.device)
    pad[:packed_local.shape[0]] = packed_local
    out = torch.empty((world * maxrows, width), dtype=packed_local.dtype, device=packed_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(nq_total, world, r)
        rows.append(out[r * maxrows: r * maxrows + (hi - lo)])
    del rank
    return torch.cat(rows, dim=0)


def sharded_knn_search(search_fn, queries, K, w, group=None):
    """search_fn(q_local, K, w) -> (ids, dists, counts) numpy; returns the gathered results of ALL queries
    on every rank, in query order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = queries.shape[0]
    lo, hi = shard_bounds(nq, world, rank)
    ids, dists, counts = search_fn(queries[lo:hi], K, w)
    packed = gather_results(pack_results(ids, dists, counts), nq, group)
    gi, gd, gc = unpack_results(packed, K)
    return gi.numpy().view(np.uint32), gd.numpy(), gc.numpy()
