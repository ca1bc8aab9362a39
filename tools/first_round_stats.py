"""On the GPU box: for the SIFT1M-shape bench indexes (mixture / lowrank data) -- how often is the SECOND probe prunable once the closest
cell has been scanned (dc[1] > K-th best ADC distance of cell 0), and how does that relate to the ratio dc[1] / dc[0] the query-major
kernel's first-round rule looks at?  Usage: python tools/first_round_stats.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import ivfadc_jl_amd as pkg

pkg.load_library()
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["sift1m"]
K = 10
for kind in ("mixture", "lowrank"):
    idx, x = bench.build_trained(pkg, cfg, dev, 0, None, kind)
    q = bench.global_queries(cfg, 4096, dev, kind).cpu().numpy()
    cent = idx._centroids
    d2 = ((q[:, None, :] - cent[None, :, :]) ** 2).sum(-1) if False else None
    # coarse distances of the two closest cells (float64 is fine for statistics)
    qq = (q.astype(np.float64) ** 2).sum(1)[:, None]
    cc = (cent.astype(np.float64) ** 2).sum(1)[None, :]
    D = qq + cc - 2.0 * q.astype(np.float64) @ cent.astype(np.float64).T
    part = np.partition(D, 1, axis=1)[:, :2]
    part.sort(axis=1)
    dc0, dc1 = part[:, 0], part[:, 1]
    ids, dists, counts = idx.search_raw(q, K, 1)
    thr1 = np.where(counts >= K, dists[:, K - 1], np.inf)
    prunable = dc1 > thr1
    ratio = dc1 / np.maximum(dc0, 1e-30)
    print("%s: second probe prunable after the closest cell for %.1f %% of %d queries" % (kind, 100.0 * prunable.mean(), q.shape[0]))
    for thr in (1.1, 1.25, 1.5, 2.0, 3.0, 4.0, 8.0):
        sel = ratio > thr
        print("   rule dc1 > %.2f dc0: fires for %.1f %%; of those %.1f %% prunable; of the prunable ones it catches %.1f %%" % (
            thr, 100.0 * sel.mean(), 100.0 * (prunable[sel].mean() if sel.any() else 0.0), 100.0 * (sel[prunable].mean() if prunable.any() else 0.0)))
    print("   ratio quantiles (10/50/90 %%): prunable %s | not prunable %s" % (
        np.round(np.quantile(ratio[prunable], [0.1, 0.5, 0.9]), 2) if prunable.any() else "-",
        np.round(np.quantile(ratio[~prunable], [0.1, 0.5, 0.9]), 2) if (~prunable).any() else "-"))
    # an alternative rule on absolute scale: thr1 is unknown beforehand, but dc1 against dc0 + typical within-cell ADC spread?
    idx.close() if hasattr(idx, "close") else None
