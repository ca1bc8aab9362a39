"""Index training: coarse k-means and per-sub-space residual k-means (product quantizer).

Counterpart of the training half of the reference constructor
(/root/reference/src/index.jl:127-147: Clustering.kmeans(init=:kmpp) and
QuantizedArrays.build_quantizer(method=:pq)).  Both are third-party and unseeded in
the reference, so only statistical equivalence is possible; this module is host-side
plumbing (torch tensors on the CPU, or on the GPU when one is present) that produces
*an* index for the hot path.  The encoding of the data (list assignment + PQ codes) is
NOT done here: it goes through the HIP push!/encode path (ivfadc_append).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat


def _sqdist_argmin(x, c, chunk=65536):
    """argmin_j ||x_i - c_j||^2 for every row, chunked; returns (assign, mindist)."""
    n = x.shape[0]
    cn = (c * c).sum(1)
    assign = torch.empty(n, dtype=torch.int64, device=x.device)
    mind = torch.empty(n, dtype=x.dtype, device=x.device)
    for s in range(0, n, chunk):
        xb = x[s:s + chunk]
        dist = (xb * xb).sum(1, keepdim=True) - 2.0 * (xb @ c.t()) + cn[None, :]
        md, a = dist.min(1)
        assign[s:s + chunk] = a
        mind[s:s + chunk] = md.clamp_min(0)
    return assign, mind


def kmeans(x, k, maxiter=25, seed=0, init_sample=None):
    """Lloyd k-means with k-means++ seeding.  x: (n, d) float32 tensor.  Returns (k, d)."""
    n, d = x.shape
    assert 1 <= k <= n, "k must be in 1..n"
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    # k-means++ on a subsample keeps seeding O(sample * k)
    ns = n if init_sample is None else min(n, int(init_sample))
    sub = x if ns == n else x[torch.randperm(n, generator=g)[:ns].to(x.device)]
    cent = torch.empty(k, d, dtype=x.dtype, device=x.device)
    first = int(torch.randint(0, ns, (1,), generator=g))
    cent[0] = sub[first]
    mind = ((sub - cent[0]) ** 2).sum(1)
    for j in range(1, k):
        tot = float(mind.sum())
        if tot <= 0.0:
            idx = int(torch.randint(0, ns, (1,), generator=g))
        else:
            r = float(torch.rand(1, generator=g)) * tot
            idx = int(torch.searchsorted(torch.cumsum(mind, 0), torch.tensor(r, dtype=mind.dtype, device=x.device)))
            idx = min(idx, ns - 1)
        cent[j] = sub[idx]
        mind = torch.minimum(mind, ((sub - cent[j]) ** 2).sum(1))
    for _ in range(int(maxiter)):
        assign, _ = _sqdist_argmin(x, cent)
        sums = torch.zeros_like(cent)
        sums.index_add_(0, assign, x)
        cnt = torch.bincount(assign, minlength=k).to(x.dtype)
        new = sums / cnt.clamp_min(1)[:, None]
        empty = cnt == 0
        if bool(empty.any()):
            ne = int(empty.sum())
            new[empty] = x[torch.randint(0, n, (ne,), generator=g).to(x.device)]
        shift = float(((new - cent) ** 2).sum())
        cent = new
        if shift == 0.0:
            break
    return cent


def train_ivfadc(data, kc, k, m, coarse_maxiter=25, quantization_maxiter=25, seed=0, device=None):
    """data (n, d) float32 -> centroids (kc, d), codebooks (m, k, dsub), labels (m, k) uint8."""
    x = torch.as_tensor(np.ascontiguousarray(data, np.float32))
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    x = x.to(device)
    n, d = x.shape
    assert d % m == 0, "d must be a multiple of m"
    dsub = d // m
    init_sample = None if n <= 65536 else max(65536, 64 * kc)
    cent = kmeans(x, kc, coarse_maxiter, seed, init_sample)
    assign, _ = _sqdist_argmin(x, cent)
    resid = x - cent[assign]
    cbs = []
    for i in range(m):
        sub = resid[:, i * dsub:(i + 1) * dsub].contiguous()
        cbs.append(kmeans(sub, k, quantization_maxiter, seed + 1 + i,
                          None if n <= 65536 else max(65536, 64 * k)))
    codebooks = torch.stack(cbs, 0)
    labels = np.tile(np.arange(k, dtype=np.uint8), (m, 1))
    return (cent.cpu().numpy().astype(np.float32), codebooks.cpu().numpy().astype(np.float32), labels)


def train_ivfadc_hip(data, kc, k, m, coarse_maxiter=25, quantization_maxiter=25, seed=0, device=0):
    """The native trainer (ivfadc_train: k-means++ + Lloyd on the GPU, deterministic per seed).
    data (n, d) float32 -> centroids (kc, d), codebooks (m, k, dsub), labels (m, k) uint8."""
    x = np.ascontiguousarray(data, np.float32)
    n, d = x.shape
    cent = np.zeros((kc, d), np.float32)
    cbs = np.zeros((m, k, d // max(m, 1)), np.float32)
    nat.check(nat.lib().ivfadc_train(int(device), d, n, nat.ptr(x, C.c_float), int(kc), int(k), int(m), int(coarse_maxiter),
                                     int(quantization_maxiter), C.c_uint64(int(seed)), nat.ptr(cent, C.c_float),
                                     nat.ptr(cbs, C.c_float)))
    labels = np.tile(np.arange(k, dtype=np.uint8), (m, 1))
    return cent, cbs, labels
