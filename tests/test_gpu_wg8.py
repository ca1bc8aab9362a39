"""GPU parity tests of the eight-wave list-major kernel (csrc/wg8scan.hip.h: m = 8, dsub = 16, K <= 64; four conflict-free copies of the
16-bit integer filter table in LDS, f32 tables in device memory, bounds from the integer sums in a crowd).  Through the C ABI against the
CPU oracle and against the reference-order kernel (table mode 1) on the same seeded inputs: ids bit-exact, distance bits identical."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def gpu_index(native, oidx):
    return native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids)


# the two forms of the kernel: (table mode, last_striped, queries per code stream).  Mode 6: four queries per 8-byte entry, four table copies
# (wg8scan.hip.h); mode 7: eight queries per 16-byte entry, two copies (wg8q8scan.hip.h) -- wherever they are instantiated (default: lists
# of >= 8192 points; the eight-query form from ten probes per list)
FORMS = {"q4": (6, 2, 4), "q8": (7, 3, 8)}
FORM = ["q4"]     # the form the helpers below build (set by the `form` fixture)


@pytest.fixture(params=["q4", "q8"], autouse=True)
def form(request):
    FORM[0] = request.param
    return request.param


def wg8_index(native, oidx, chunk=0):
    g = gpu_index(native, oidx)
    g.set_tuning(4, chunk)
    g.set_table_mode(FORMS[FORM[0]][0])
    return g


def striped():
    return FORMS[FORM[0]][1]


@pytest.mark.parametrize("case", ["random", "permuted_labels", "few_codes", "one_list", "exact_hits", "far_queries", "clustered", "short_lists"])
def test_eight_wave_list_major_kernel(native, case):
    """Groups that are full, partial (a list probed by 1 .. 3 queries) and several per list; one and several chunks per list; K = 1 / 10 / 64
    (ONE pool of K keys per query slot in LDS, filled by the eight waves with compare-and-swap: offers that tie on the pool's maximum, pools
    that never fill -- fewer than K points in a chunk --, K = 64 = one entry per lane); permuted labels;
    lists of a handful of distinct codes (exact ties across whole steps: the crowd bound must not cut a tie); queries that hit codewords
    exactly (entries of 0); queries far from every centroid; a clustered set where bounds tighten early and whole work items are pruned;
    lists shorter than one step of eight waves (idle waves, empty lists)."""
    d, m = 128, 8
    kc = 1 if case == "one_list" else (300 if case == "short_lists" else 14)
    n = 30000
    oidx, data = helpers.build_index(2500 + len(case), n, d, kc, m, 256, label_perm=(case == "permuted_labels"),
                                     mode="encode" if case == "clustered" else "random", ndistinct=(4 if case == "few_codes" else None))
    rng = np.random.default_rng(177 + len(case))
    qs = rng.random((61, d), dtype=np.float32)
    if case == "exact_hits":
        for i in range(16):
            cl = i % kc
            code = rng.integers(0, 256, m)
            qs[i] = oidx.centroids[cl] + np.concatenate([oidx.codebooks[ii, code[ii]] for ii in range(m)])
    elif case == "far_queries":
        qs += np.float32(50.0)
    elif case == "clustered":
        qs = (data[rng.integers(0, n, 61)] + 0.01 * rng.standard_normal((61, d))).astype(np.float32)
    gref = gpu_index(native, oidx)
    gref.set_tuning(4, 0)
    gref.set_table_mode(1)                                  # reference-order f32 tables in every lane
    for K, w, chunk in ((10, 3, 0), (1, 1, 1024), (64, min(kc, 5), 4096), (10, min(kc, 14), 0), (16, 2, 0), (17, 2, 2048)):
        exp = oidx.knn_search(qs, K, w)
        g = wg8_index(native, oidx, chunk)
        g.reset_stats()
        got = g.search_raw(qs, K, w)
        st = g.get_stats()
        assert st["last_striped"] == striped() and st["last_qg"] == FORMS[FORM[0]][2] and st["last_scan_lds"] <= 80 * 1024, st
        helpers.assert_same_results(got, exp, what="wg8 %s K=%d w=%d chunk=%d" % (case, K, w, chunk))
        ref = gref.search_raw(qs, K, w)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), "wg8 vs reference-order kernel: %s K=%d w=%d" % (case, K, w)
        # a second search on the same handle (re-armed queue, bounds and counters), fewer queries (partial groups everywhere)
        got2 = g.search_raw(qs[:9], K, w)
        helpers.assert_same_results(got2, tuple(a[:9] for a in exp), what="wg8 %s, second call" % case)
    # pruning on / off agree; K > 64 leaves the kernel (LDS selectors: the four-wave kernel)
    g = wg8_index(native, oidx)
    g.set_pruning(0)
    helpers.assert_same_results(g.search_raw(qs, 10, min(kc, 6)), oidx.knn_search(qs, 10, min(kc, 6)), what="wg8 %s, pruning off" % case)
    g = wg8_index(native, oidx)
    helpers.assert_same_results(g.search_raw(qs[:8], 100, min(kc, 3)), oidx.knn_search(qs[:8], 100, min(kc, 3)), what="wg8 %s, K=100" % case)
    assert g.get_stats()["last_striped"] not in (2, 3)


@pytest.mark.parametrize("case", ["outlier_codewords", "zero_codebooks", "tiny_scale", "huge_scale", "dc_dominates_300", "dc_dominates_5000",
                                  "dc_zero_huge_entries"])
def test_eight_wave_kernel_filter_extremes(native, case):
    """The cases of test_integer_filter_extremes through the eight-wave kernel: whatever the scale does -- one far codeword per
    sub-quantizer that flattens every other entry to 0, tables that are all zero (every point of every list ties: the crowd bound is
    not used when the scale is not a normal number, and where it is used it may never cut a point that belongs to the result), entries in
    the denormal range, entries near the top of the float range, sums dominated by the coarse distance -- the filter and the bounds taken
    from the integer sums may only let MORE points through."""
    d, m, kc = 128, 8, 12
    oidx, _ = helpers.build_index(1400 + len(case), 40000, d, kc, m, 256, mode="random")
    rng = np.random.default_rng(len(case))
    if case == "outlier_codewords":
        oidx.codebooks[:, 7, :] *= np.float32(1000.0)
    elif case == "zero_codebooks":
        oidx.codebooks[:] = 0
    elif case == "tiny_scale":
        oidx.codebooks *= np.float32(1e-21)
        oidx.centroids *= np.float32(1e-21)
    elif case == "huge_scale":
        oidx.codebooks *= np.float32(1e15)
        oidx.centroids *= np.float32(1e15)
    qs = rng.random((64, d), dtype=np.float32)
    if case.startswith("dc_dominates"):
        oidx.centroids += np.float32(300.0 if case.endswith("300") else 5000.0)
        oidx.codebooks *= np.float32(1e-3)
        qs[32:] += np.float32(300.0 if case.endswith("300") else 5000.0)
    elif case == "dc_zero_huge_entries":
        oidx.codebooks *= np.float32(1e3)
        qs[:12] = oidx.centroids[:12]
    if case == "tiny_scale":
        qs *= np.float32(1e-21)
    elif case == "huge_scale":
        qs *= np.float32(1e15)
    elif case == "zero_codebooks":
        qs[:8] = oidx.centroids[:8]
    for K, w in ((10, 4), (64, 2)):
        exp = oidx.knn_search(qs, K, w)
        g = wg8_index(native, oidx, 8192)
        got = g.search_raw(qs, K, w)
        assert g.get_stats()["last_striped"] == striped()
        helpers.assert_same_results(got, exp, what="wg8 filter %s K=%d" % (case, K))
        gref = gpu_index(native, oidx)
        gref.set_tuning(4, 8192)
        gref.set_table_mode(1)
        assert all(np.array_equal(a, b) for a, b in zip(got, gref.search_raw(qs, K, w)))


def test_eight_wave_kernel_on_long_lists(native):
    """Lists of ~10 000 points (whole steps of eight waves, several per wave), four queries per code stream: the eight-wave kernel (table
    mode 6) and the four-wave kernel (table mode 5) agree bit for bit (40 lists, 256 queries, w = 4: ~25 probes per list)."""
    d, m, kc, n = 128, 8, 40, 400000
    oidx, _ = helpers.build_index(4242, n, d, kc, m, 256, mode="random")
    qs = np.random.default_rng(9).random((256, d), dtype=np.float32)
    g = wg8_index(native, oidx, 0)   # (the list-major plan forced: this batch is small enough for the query-major one)
    got = g.search_raw(qs, 10, 4)
    st = g.get_stats()
    assert st["last_striped"] == striped() and st["last_qg"] == FORMS[FORM[0]][2], st
    g4 = gpu_index(native, oidx)
    g4.set_tuning(4, 0)
    g4.set_table_mode(5)
    ref = g4.search_raw(qs, 10, 4)
    assert g4.get_stats()["last_striped"] == 1
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    pick = np.arange(0, 256, 8)
    exp = oidx.knn_search(qs[pick], 10, 4)
    helpers.assert_same_results(tuple(a[pick] for a in got), exp, what="wg8 default plan")


def test_fuzz_eight_wave_kernel(native):
    """Randomised differential test in the eight-wave kernel's own domain (m = 8, d = 128, ksub = 256, K <= 64, four queries per stream,
    table mode 6): list counts and sizes from empty lists to a few thousand points, chunk sizes that give partial last steps and several
    chunks per list, K from 1 to 64, w up to kc, batches that leave partial groups, permuted labels, few distinct codes (ties across whole
    steps), pruning on and off, in-place pushes and deletes between searches.  Against the oracle, ids exact and distance bits equal.
    IVFADC_FUZZ_DRAWS / IVFADC_FUZZ_SEED widen it for soak runs."""
    import os
    rng = np.random.default_rng(int(os.environ.get("IVFADC_FUZZ_SEED", "8086")))
    d, m = 128, 8
    for it in range(int(os.environ.get("IVFADC_FUZZ_DRAWS", "24"))):
        kc = int(rng.choice([1, 2, 5, 14, 33, 120]))
        n = int(rng.choice([0, 7, 300, 3000, 20000, 45000]))
        K = int(rng.choice([1, 2, 8, 9, 10, 16, 17, 33, 64]))
        w = int(rng.choice([1, 2, 3, 8, 14, 200]))
        nq = int(rng.choice([1, 4, 5, 37, 130]))
        chunk = int(rng.choice([0, 0, 1024, 2048, 8192]))
        oidx, data = helpers.build_index(7000 + it, n, d, kc, m, 256, label_perm=bool(rng.random() < 0.4),
                                         mode="encode" if (n and n <= 3000 and rng.random() < 0.4) else "random",
                                         ndistinct=(3 if rng.random() < 0.25 else None))
        qs = rng.random((nq, d), dtype=np.float32)
        if n:
            qs[: min(nq, 3)] = data[: min(nq, 3)]
        if rng.random() < 0.2:
            qs += np.float32(20.0)
        g = wg8_index(native, oidx, chunk)
        if rng.random() < 0.3:
            g.set_pruning(0)
        what = "wg8 fuzz %d: kc=%d n=%d K=%d w=%d nq=%d chunk=%d" % (it, kc, n, K, w, nq, chunk)
        got = g.search_raw(qs, K, w)
        assert g.get_stats()["last_striped"] == striped(), what
        exp = oidx.knn_search(qs, K, w)
        helpers.assert_same_results(got, exp, what=what)
        assert np.array_equal(got[1][exp[1] < np.inf], exp[1][exp[1] < np.inf]), what
        if it % 3 == 0:
            npush = int(rng.choice([1, 9, 200]))
            pts = rng.random((npush, d), dtype=np.float32)
            g._append(pts, np.arange(n, n + npush, dtype=np.uint32))
            if n + npush > 2:
                g._delete_ids(rng.integers(0, n + npush, int(rng.choice([1, 5, 60]))).astype(np.uint32))
            offsets, codes, ids = g._lists()
            from oracle import oracle as ora
            o2 = ora.OracleIndex(oidx.centroids, oidx.codebooks, oidx.labels, offsets, codes, ids)
            helpers.assert_same_results(g.search_raw(qs, K, w), o2.knn_search(qs, K, w), what=what + " after edits")
