"""CPU k-means / PQ trainer in plain torch (Lloyd + k-means++): TEST FIXTURE ONLY.

Produces *an* index for golden fixtures and is the yardstick the native trainer (ivfadc_train, HIP) is compared with in
tests/test_trainer.py.  It is not part of the product package (moved out of ivfadc.jl_amd/ in round 2)."""
import numpy as np
import torch


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
