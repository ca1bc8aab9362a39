"""Multi-GPU knn_search: queries partitioned across ranks, index replicated per GPU.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
on CPU for tests).  Queries are independent (/root/reference/src/index.jl:269-271), so the
only exchange is the final gather of each rank's top-k block -- one collective per batch of
packed [nq_local, K] (id, dist) + counts; no reduction is needed because ranks own disjoint
queries.

Strong scaling of a FIXED global batch takes the other partition (`list_partitioned_knn_search`): every rank keeps the replica and
ALL queries, scans only the probed lists l with l % world == rank, and the exchange is one all-gather of each rank's K smallest
(distance, visit order) keys per query followed by a K-way merge -- probes are independent given the bound
(/root/reference/src/index.jl:228-255), and visit orders are global, so keys of different ranks compare.
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


def merge_partial_topk(keys_all, counts_all, K):
    """K-way merge of per-rank partial results.  keys_all (nparts, nq, K) uint64 = float32 bits of the distance << 32 | visit order,
    ascending per (rank, query) with counts_all (nparts, nq) valid entries each.  Keys are unique per query (a visit order names one
    stored point), so the K smallest of the union are the K smallest of the whole scan (index.jl:247-254).  Returns (keys (nq, K), counts)."""
    keys_all = np.asarray(keys_all, np.uint64)
    counts_all = np.asarray(counts_all)
    nparts, nq, kk = keys_all.shape
    valid = np.arange(kk)[None, None, :] < counts_all[:, :, None]
    flat = np.where(valid, keys_all, np.uint64(0xFFFFFFFFFFFFFFFF)).transpose(1, 0, 2).reshape(nq, nparts * kk)
    flat.sort(axis=1)
    counts = np.minimum(valid.sum(axis=(0, 2)), K).astype(np.int32)
    return np.ascontiguousarray(flat[:, :K]), counts


def list_partitioned_knn_search(partial_fn, queries, K, w, group=None):
    """partial_fn(queries, K, w, nparts, part) -> (keys (nq, K) uint64, counts (nq,) int32, payload (nq, K) uint32: the stored ids of the
    keys) for the probed lists l with l % nparts == part.  Every rank passes ALL queries.  One all-gather of the packed
    [keys | payload | counts] block; returns ids, dists, counts of the whole scan on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    keys, counts, payload = partial_fn(queries, K, w, world, rank)
    nq = keys.shape[0]
    blk = torch.cat([torch.as_tensor(np.ascontiguousarray(keys).view(np.int64)),
                     torch.as_tensor(np.ascontiguousarray(payload).astype(np.int64)),
                     torch.as_tensor(np.ascontiguousarray(counts).astype(np.int64)).view(-1, 1)], dim=1).contiguous()
    out = torch.empty((world * nq, 2 * K + 1), dtype=torch.int64)
    dist.all_gather_into_tensor(out, blk, group=group)
    g = out.numpy().reshape(world, nq, 2 * K + 1)
    keys_all = g[:, :, :K].view(np.uint64)
    ids_all = g[:, :, K:2 * K]
    counts_all = g[:, :, 2 * K]
    mk, mc = merge_partial_topk(keys_all, counts_all, K)
    # payload of the winners: a key occurs once in the union
    ids = np.zeros((nq, K), np.uint32)
    flat_k = keys_all.transpose(1, 0, 2).reshape(nq, world * K)
    flat_i = ids_all.transpose(1, 0, 2).reshape(nq, world * K)
    flat_v = (np.arange(K)[None, None, :] < counts_all[:, :, None]).transpose(1, 0, 2).reshape(nq, world * K)
    for q in range(nq):
        lut = {int(k): int(i) for k, i, v in zip(flat_k[q], flat_i[q], flat_v[q]) if v}
        for j in range(int(mc[q])):
            ids[q, j] = lut[int(mk[q, j])]
    dists = (mk >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return ids, dists, mc
