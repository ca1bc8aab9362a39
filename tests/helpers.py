"""Shared test helpers: seeded index builders and an independent numpy restatement used to
cross-check the C oracle (tests only)."""
import numpy as np

from oracle import oracle as ora


def make_quantizers(seed, d, kc, m, ksub, label_perm=False, scale=0.25):
    rng = np.random.default_rng(seed)
    dsub = d // m
    cent = rng.random((kc, d), dtype=np.float32)
    cbs = ((rng.random((m, ksub, dsub), dtype=np.float32) - 0.5) * 2 * scale).astype(np.float32)
    if label_perm:
        labels = np.stack([rng.permutation(256)[:ksub].astype(np.uint8) for _ in range(m)])
    else:
        labels = np.tile(np.arange(ksub, dtype=np.uint8), (m, 1))
    return cent, cbs, labels


def build_index(seed, n, d, kc, m, ksub=256, label_perm=False, mode="encode", ndistinct=None, shuffle_ids=True):
    """Returns (OracleIndex, data).  mode: 'encode' = lists/codes from the oracle's _encode_point;
    'random' = random list assignment and random codes (optionally only `ndistinct` different
    codes -> many exact distance ties)."""
    rng = np.random.default_rng(seed + 1000)
    cent, cbs, labels = make_quantizers(seed, d, kc, m, ksub, label_perm)
    data = rng.random((n, d), dtype=np.float32)
    tmp = ora.OracleIndex(cent, cbs, labels, np.zeros(kc + 1, np.int64), np.zeros((0, m), np.uint8), np.zeros(0, np.uint32))
    if mode == "encode":
        lst, codes = tmp.encode(data) if n else (np.zeros(0, np.int32), np.zeros((0, m), np.uint8))
    else:
        lst = rng.integers(0, kc, n).astype(np.int32)
        if ndistinct:
            pool = np.stack([labels[i][rng.integers(0, ksub, ndistinct)] for i in range(m)], 1)   # (ndistinct, m)
            codes = pool[rng.integers(0, ndistinct, n)]
        else:
            codes = np.stack([labels[i][rng.integers(0, ksub, n)] for i in range(m)], 1).astype(np.uint8)
    order = np.argsort(lst, kind="stable")
    ids = order.astype(np.uint32)            # id = original position, ascending within a list
    if shuffle_ids and mode != "encode":
        ids = rng.permutation(n).astype(np.uint32)
    offsets = np.zeros(kc + 1, np.int64)
    np.cumsum(np.bincount(lst, minlength=kc), out=offsets[1:])
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, np.ascontiguousarray(codes[order]), ids)
    return oidx, data


def numpy_knn(oidx, q, K, w):
    """Independent restatement: exhaustive (dist, visit order) list, then a lexicographic sort.
    float32 throughout, sums sequential in ascending index (elementwise numpy ops round once)."""
    f32 = np.float32
    q = np.asarray(q, f32)
    w = min(w, oidx.kc)
    acc = np.zeros(oidx.kc, f32)
    for i in range(oidx.d):
        t = oidx.centroids[:, i] - q[i]
        acc = acc + t * t
    order = np.lexsort((np.arange(oidx.kc), acc))[:w]
    cand_d, cand_id = [], []
    for j, cl in enumerate(order):
        dc = acc[cl]
        r = q - oidx.centroids[cl]
        tab = np.zeros((oidx.m, 256), f32)
        for i in range(oidx.m):
            s = np.zeros(oidx.ksub, f32)
            for t_ in range(oidx.dsub):
                df = oidx.codebooks[i, :, t_] - r[i * oidx.dsub + t_]
                s = s + df * df
            tab[i, oidx.labels[i]] = s
        lo, hi = int(oidx.offsets[cl]), int(oidx.offsets[cl + 1])
        dd = np.full(hi - lo, dc, f32)
        for ii in range(oidx.m):
            dd = dd + tab[ii, oidx.codes[lo:hi, ii]]
        cand_d.append(dd)
        cand_id.append(oidx.ids[lo:hi])
    cd = np.concatenate(cand_d) if cand_d else np.zeros(0, f32)
    ci = np.concatenate(cand_id) if cand_id else np.zeros(0, np.uint32)
    sel = np.lexsort((np.arange(cd.shape[0]), cd))[:K]
    return ci[sel], cd[sel]


def assert_same_results(got, exp, rtol=1e-4, what=""):
    """ids bit-exact, Float32 distances within 1e-4 relative (the north_star tolerance)."""
    gi, gd, gc = got
    ei, ed, ec = exp
    assert np.array_equal(gc, ec), "%s counts differ: %s vs %s" % (what, gc[:16], ec[:16])
    for r in range(gc.shape[0]):
        c = int(gc[r])
        assert np.array_equal(gi[r, :c], ei[r, :c]), "%s ids differ at query %d: %s vs %s (d %s vs %s)" % (
            what, r, gi[r, :c], ei[r, :c], gd[r, :c], ed[r, :c])
        assert np.allclose(gd[r, :c], ed[r, :c], rtol=rtol, atol=0.0), "%s dists differ at query %d" % (what, r)


def numpy_partial_keys(oidx, qs, K, w, nparts, part):
    """List-partitioned restatement: for every query the K smallest (distance, visit order) keys over the probed lists l with
    l % nparts == part -- visit orders counted over ALL w probes, as one rank of ivfadc_search_device_partial leaves them --
    plus the stored ids of those keys.  float32 throughout, sums sequential in ascending index."""
    f32 = np.float32
    qs = np.asarray(qs, f32)
    w = min(w, oidx.kc)
    nq = qs.shape[0]
    keys = np.full((nq, K), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
    ids = np.zeros((nq, K), np.uint32)
    counts = np.zeros(nq, np.int32)
    for qi in range(nq):
        q = qs[qi]
        acc = np.zeros(oidx.kc, f32)
        for i in range(oidx.d):
            t = oidx.centroids[:, i] - q[i]
            acc = acc + t * t
        order = np.lexsort((np.arange(oidx.kc), acc))[:w]
        base = 0
        ck, ci = [], []
        for cl in order:
            lo, hi = int(oidx.offsets[cl]), int(oidx.offsets[cl + 1])
            if cl % nparts == part and hi > lo:
                r = q - oidx.centroids[cl]
                tab = np.zeros((oidx.m, 256), f32)
                for i in range(oidx.m):
                    s = np.zeros(oidx.ksub, f32)
                    for t_ in range(oidx.dsub):
                        df = oidx.codebooks[i, :, t_] - r[i * oidx.dsub + t_]
                        s = s + df * df
                    tab[i, oidx.labels[i]] = s
                dd = np.full(hi - lo, acc[cl], f32)
                for ii in range(oidx.m):
                    dd = dd + tab[ii, oidx.codes[lo:hi, ii]]
                ck.append((dd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (base + np.arange(hi - lo)).astype(np.uint64))
                ci.append(oidx.ids[lo:hi])
            base += hi - lo
        if ck:
            allk, alli = np.concatenate(ck), np.concatenate(ci)
            sel = np.argsort(allk, kind="stable")[:K]
            keys[qi, :len(sel)] = allk[sel]
            ids[qi, :len(sel)] = alli[sel]
            counts[qi] = len(sel)
    return keys, counts, ids
