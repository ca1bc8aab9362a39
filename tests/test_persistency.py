"""Reader/writer of the reference's on-disk format (src/persistency.jl:1-78, 82-134)."""
import io
import os

import numpy as np
import pytest

import helpers


def _write_reference_style(path, cent, cbs, labels, offsets, codes, ids, itype="UInt32"):
    """Byte-for-byte what save_ivfadc_index (persistency.jl:1-78) writes for a NaiveQuantizer index."""
    kc, d = cent.shape
    m, k, dsub = cbs.shape
    idt = {"UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32}[itype]
    with open(path, "wb") as f:
        f.write(("%d %d\n%d %d %d %d\nNaiveQuantizer\nQuantizedArrays.OrthogonalQuantization\nUInt8\n%s\n"
                 "Distances.SqEuclidean\nDistances.SqEuclidean\nFloat32\n" % (d, kc, len(ids), m, k, dsub, itype)).encode())
        for c in range(kc):                                   # :44-49 column by column
            f.write(cent[c].astype("<f4").tobytes())
        for i in range(m):                                    # :56-61
            f.write(labels[i].tobytes())
            for j in range(dsub):
                f.write(cbs[i][:, j].astype("<f4").tobytes())  # vectors[j, :]
        for i in range(d):                                    # :62-64 rotation
            f.write(np.eye(d, dtype="<f4")[:, i].tobytes())
        for l in range(kc):                                   # :68-78
            lo, hi = int(offsets[l]), int(offsets[l + 1])
            f.write(np.int64(hi - lo).tobytes())
            f.write(ids[lo:hi].astype(idt).tobytes())
            for p in range(lo, hi):
                f.write(codes[p].tobytes())


def test_read_reference_layout(tmp_path, native):
    oidx, _ = helpers.build_index(3, 200, 12, 9, 3, 32, label_perm=True)
    path = os.path.join(str(tmp_path), "ref.bin")
    _write_reference_style(path, oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids, "UInt16")
    import ivfadc_file_format as fmt
    a = fmt.read_ivfadc_file(path)
    assert np.array_equal(a["centroids"], oidx.centroids) and np.array_equal(a["codebooks"], oidx.codebooks)
    assert np.array_equal(a["labels"], oidx.labels) and np.array_equal(a["offsets"], oidx.offsets)
    assert np.array_equal(a["codes"], oidx.codes) and np.array_equal(a["ids"], oidx.ids)
    assert a["index_type"] == np.dtype(np.uint16) and a["n"] == 200


@pytest.mark.gpu
def test_save_load_roundtrip(tmp_path, native):
    """test/persistency.jl: save -> load equality of every field, and the file equals the reference layout."""
    oidx, _ = helpers.build_index(4, 500, 16, 11, 4, 64)
    g = native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids,
                                       index_type=np.uint16)
    path = os.path.join(str(tmp_path), "idx.bin")
    native.save_ivfadc_index(path, g)
    ref = os.path.join(str(tmp_path), "ref.bin")
    _write_reference_style(ref, oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids, "UInt16")
    assert open(path, "rb").read() == open(ref, "rb").read()
    g2 = native.load_ivfadc_index(path)
    assert g2.index_type == np.dtype(np.uint16) and len(g2) == 500
    o1, c1, i1 = g._lists()
    o2, c2, i2 = g2._lists()
    assert np.array_equal(o1, o2) and np.array_equal(c1, c2) and np.array_equal(i1, i2)
    qs = np.random.default_rng(4).random((10, 16), dtype=np.float32)
    a, b = g.search_raw(qs, 5, 3), g2.search_raw(qs, 5, 3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("itype", ["UInt8", "UInt16", "UInt32"])
def test_native_reader_and_writer(tmp_path, native, itype):
    """ivfadc_load_index reads a file laid out exactly as persistency.jl:1-78 writes it; ivfadc_save_index writes
    those bytes back; the numpy writer/reader agree with both."""
    n = 200 if itype == "UInt8" else 700
    oidx, _ = helpers.build_index(6, n, 12, 9, 3, 32, label_perm=True)
    ref = os.path.join(str(tmp_path), "ref.bin")
    _write_reference_style(ref, oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids, itype)
    g = native.load_ivfadc_index(ref)
    assert g.index_type == np.dtype({"UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32}[itype]) and len(g) == n
    o, c, i = g._lists()
    assert np.array_equal(o, oidx.offsets) and np.array_equal(c, oidx.codes) and np.array_equal(i, oidx.ids)
    assert np.array_equal(g._centroids, oidx.centroids) and np.array_equal(g._codebooks, oidx.codebooks)
    qs = np.random.default_rng(6).random((16, 12), dtype=np.float32)
    helpers.assert_same_results(g.search_raw(qs, 5, 4), oidx.knn_search(qs, 5, 4), what="loaded " + itype)
    out = os.path.join(str(tmp_path), "out.bin")
    native.save_ivfadc_index(out, g)
    assert open(out, "rb").read() == open(ref, "rb").read()
    import ivfadc_file_format as fmt
    out2 = os.path.join(str(tmp_path), "out2.bin")
    fmt.write_ivfadc_file(out2, g)
    assert open(out2, "rb").read() == open(ref, "rb").read()


@pytest.mark.gpu
def test_native_reader_rejects_what_it_cannot_search(tmp_path, native):
    oidx, _ = helpers.build_index(7, 100, 8, 5, 2, 16)
    path = os.path.join(str(tmp_path), "bad.bin")
    _write_reference_style(path, oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids, "UInt16")
    data = open(path, "rb").read()
    hnsw = data.replace(b"NaiveQuantizer", b"HNSWQuantizer", 1)
    open(path, "wb").write(hnsw)
    with pytest.raises(native.IVFADCError):
        native.load_ivfadc_index(path)
    open(path, "wb").write(data[:len(data) - 7])          # truncated last list
    with pytest.raises(native.IVFADCError):
        native.load_ivfadc_index(path)
    with pytest.raises(native.IVFADCError):
        native.load_ivfadc_index(os.path.join(str(tmp_path), "missing.bin"))


def _native_load_rc(native, path):
    import ctypes as C
    lib = native.load_library()
    h = C.c_void_p()
    bits = C.c_int(0)
    rc = lib.ivfadc_load_index(C.byref(h), 0, str(path).encode(), C.byref(bits))
    msg = lib.ivfadc_last_error().decode()
    if rc == 0:
        lib.ivfadc_destroy(h)
    return rc, msg


def _small_file(tmp_path, name="f.bin", itype="UInt16"):
    oidx, _ = helpers.build_index(8, 120, 8, 5, 2, 16)
    path = os.path.join(str(tmp_path), name)
    _write_reference_style(path, oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids, itype)
    return path, oidx


def test_loader_gate_runs_before_any_device_call(tmp_path, native):
    """Every file the HIP path cannot search with the reference's semantics -- another quantization (header line 4), another
    coarse / residual distance (lines 7, 8; persistency.jl:14-19), a rotated quantizer -- and every corrupt or hostile
    header (sizes beyond the file: ADVICE r1, SIGABRT on n = 1e15) comes back as IVFADC_ERR_INVALID.  No GPU needed:
    the gate sits in front of ivfadc_create."""
    path, oidx = _small_file(tmp_path)
    good = open(path, "rb").read()
    bad = os.path.join(str(tmp_path), "bad.bin")

    def rc_of(data):
        open(bad, "wb").write(data)
        return _native_load_rc(native, bad)

    cases = {
        "quantization": good.replace(b"QuantizedArrays.OrthogonalQuantization", b"QuantizedArrays.AdditiveQuantization", 1),
        "coarse distance": good.replace(b"Distances.SqEuclidean", b"Distances.Euclidean", 1),
        "residual distance": good.replace(b"Distances.SqEuclidean\nFloat32", b"Distances.CosineDist\nFloat32", 1),
        "coarse quantizer": good.replace(b"NaiveQuantizer", b"HNSWQuantizer", 1),
        "U": good.replace(b"\nUInt8\n", b"\nUInt16\n", 1),
        "I": good.replace(b"\nUInt16\n", b"\nUInt64\n", 1),
        "T": good.replace(b"\nFloat32\n", b"\nFloat16\n", 1),
        "huge n": good.replace(b"\n120 2 16 4\n", b"\n1000000000000000 2 16 4\n", 1),
        "huge nclusters": good.replace(b"8 5\n", b"8 2000000000\n", 1),
        "huge nrows": good.replace(b"8 5\n", b"80000000 5\n", 1),
        "m * dsub != nrows": good.replace(b"\n120 2 16 4\n", b"\n120 3 16 4\n", 1),
        "n smaller than the lists": good.replace(b"\n120 2 16 4\n", b"\n100 2 16 4\n", 1),
        "truncated header": good[:40],
        "truncated centroids": good[:160],
        "truncated last list": good[:-7],
        "empty": b"",
    }
    for what, data in cases.items():
        rc, msg = rc_of(data)
        assert rc == 2, "%s: rc=%d (%s)" % (what, rc, msg)
    # a list length beyond what the header's n leaves
    hdr_len = len(good) - len(good.split(b"Float32\n", 1)[1])
    d, kc, m, k, dsub = 8, 5, 2, 16, 4
    first_list = hdr_len + 4 * d * kc + m * (k + 4 * dsub * k) + 4 * d * d
    evil = bytearray(good)
    evil[first_list:first_list + 8] = np.int64(1 << 40).tobytes()
    rc, msg = rc_of(bytes(evil))
    assert rc == 2, msg
    evil[first_list:first_list + 8] = np.int64(-5).tobytes()
    assert rc_of(bytes(evil))[0] == 2
    # a rotated residual quantizer (:opq; one off-diagonal entry of the rotation matrix) is NOT refused: knn_search never reads rot
    # (index.jl:204-258), the file passes the gate (on this GPU-less host the call then ends at ivfadc_create: no device); a non-finite
    # rotation entry is refused like any other non-finite quantizer value
    rot0 = hdr_len + 4 * d * kc + m * (k + 4 * dsub * k)
    rotated = bytearray(good)
    rotated[rot0 + 4:rot0 + 8] = np.float32(0.25).tobytes()
    rc, msg = rc_of(bytes(rotated))
    assert rc != 2, msg
    nanrot = bytearray(good)
    nanrot[rot0 + 4:rot0 + 8] = np.float32(np.nan).tobytes()
    rc, msg = rc_of(bytes(nanrot))
    assert rc == 2 and "rotation" in msg
    # the numpy reader applies the same gate
    import ivfadc_file_format as fmt
    for what in ("quantization", "coarse distance", "residual distance"):
        open(bad, "wb").write(cases[what])
        with pytest.raises(NotImplementedError):
            fmt.read_ivfadc_file(bad)
    open(bad, "wb").write(bytes(rotated))
    assert fmt.read_ivfadc_file(bad)["rot"][0, 1] == np.float32(0.25)        # (row i of the returned array = column i of rot)
    assert np.array_equal(fmt.read_ivfadc_file(path)["rot"], np.eye(d, dtype=np.float32))
    assert fmt.read_ivfadc_file(path)["n"] == 120


@pytest.mark.gpu
def test_loader_accepts_both_spellings_of_type_names(tmp_path, native):
    """string(Dc) is `Distances.SqEuclidean` or `SqEuclidean` depending on what the writing session imported
    (persistency.jl:137-144 reads both): both load and search identically."""
    path, oidx = _small_file(tmp_path)
    good = open(path, "rb").read()
    short = good.replace(b"QuantizedArrays.OrthogonalQuantization", b"OrthogonalQuantization", 1).replace(b"Distances.SqEuclidean", b"SqEuclidean")
    p2 = os.path.join(str(tmp_path), "short.bin")
    open(p2, "wb").write(short)
    qs = np.random.default_rng(8).random((9, 8), dtype=np.float32)
    a = native.load_ivfadc_index(path).search_raw(qs, 4, 3)
    b = native.load_ivfadc_index(p2).search_raw(qs, 4, 3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    helpers.assert_same_results(a, oidx.knn_search(qs, 4, 3), what="loaded")


@pytest.mark.gpu
def test_save_refuses_to_truncate_ids(tmp_path, native):
    """ids >= 2^bits must not be narrowed silently (the reference asserts the capacity of I, index.jl:124-125)."""
    oidx, _ = helpers.build_index(9, 700, 8, 5, 2, 16)
    g = native.IVFADCIndex.from_arrays(oidx.centroids, oidx.codebooks, oidx.labels, oidx.offsets, oidx.codes, oidx.ids)
    import ctypes as C
    lib = native.load_library()
    out = os.path.join(str(tmp_path), "narrow.bin")
    assert lib.ivfadc_save_index(g._h, out.encode(), 8) == 1           # IVFADC_ERR_ASSERT: 700 ids do not fit UInt8
    assert lib.ivfadc_save_index(g._h, out.encode(), 16) == 0


# ---- round 5: byte fixtures assembled by hand from the reference writer's `write` calls --------------------------------
def _hand_fixture_expected():
    """The arrays the hand-assembled files hold, in the C ABI's layout, from the generator's closed-form value formulas (Julia indexing,
    1-based).  Nothing here goes through a reader."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_persistency_fixture", os.path.join(here, "make_persistency_fixture.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    d = g.NROWS
    cent = np.array([[g.centroid(r, c) for r in range(1, d + 1)] for c in range(1, g.NCLUSTERS + 1)], np.float32)            # [kc][d]
    cbs = np.array([[[g.codeword(i, j, c) for j in range(1, g.DSUB + 1)] for c in range(1, g.K + 1)] for i in range(1, g.M + 1)],
                   np.float32)                                                                                              # [m][k][dsub]
    labels = np.array([[g.label(i, c) for c in range(1, g.K + 1)] for i in range(1, g.M + 1)], np.uint8)
    offsets = np.concatenate([[0], np.cumsum(g.LIST_SIZES)]).astype(np.int64)
    ids = np.array([g.list_id(i, j) for i in range(1, g.NCLUSTERS + 1) for j in range(1, g.LIST_SIZES[i - 1] + 1)], np.uint32)
    codes = np.array([[g.list_code(i, j, ii) for ii in range(1, g.M + 1)] for i in range(1, g.NCLUSTERS + 1)
                      for j in range(1, g.LIST_SIZES[i - 1] + 1)], np.uint8).reshape(-1, g.M)
    return g, here, dict(centroids=cent, codebooks=cbs, labels=labels, offsets=offsets, ids=ids, codes=codes)


def test_hand_assembled_fixture_is_what_the_generator_makes_and_the_numpy_reader_reads_it():
    """The committed files equal what the generator assembles call by call (persistency.jl:22-78), and the numpy restatement of the
    loader reads the expected arrays out of them -- arrays that come from the value formulas, not from another reader."""
    import ivfadc_file_format as fmt
    g, here, exp = _hand_fixture_expected()
    for name, (T, I, rotated) in g.FILES.items():
        path = os.path.join(here, name)
        assert open(path, "rb").read() == g.build(T, I, rotated), name
        a = fmt.read_ivfadc_file(path)
        exp_rot = np.array([[g.rotation(r, c) if rotated else float(r == c) for r in range(1, g.NROWS + 1)] for c in range(1, g.NROWS + 1)], np.float32)
        assert np.array_equal(a["rot"], exp_rot), name
        for key in ("centroids", "codebooks", "labels", "offsets", "ids", "codes"):
            assert np.array_equal(a[key], exp[key]), (name, key)
        assert a["index_type"] == np.dtype({"UInt16": np.uint16, "UInt32": np.uint32}[I]) and a["T"] == T and a["n"] == len(exp["ids"])
    # the first centroid column and a codeword row, spelled out: 100*col + row + 0.5 and 1000*i + 10*c + j + 0.125
    assert exp["centroids"][0].tolist() == [101.5, 102.5, 103.5, 104.5] and exp["centroids"][2][3] == 304.5
    assert exp["codebooks"][1][3].tolist() == [2041.125, 2042.125]      # codebook 2, codeword 4: vectors[1:2, 4]
    assert exp["labels"][0].tolist() == [3, 0, 2, 1]


@pytest.mark.gpu
def test_native_reader_reads_the_hand_assembled_fixture(tmp_path, native):
    """ivfadc_load_index against bytes nobody's reader produced: every field lands where the reference's writer put it (both element
    widths; Float64 narrowed), the handle searches like the oracle on the expected arrays, and ivfadc_save_index writes the Float32
    file back byte for byte."""
    g, here, exp = _hand_fixture_expected()
    from oracle import oracle as ora
    for name, (T, I, rotated) in g.FILES.items():
        idx = native.load_ivfadc_index(os.path.join(here, name))
        # an :opq file (non-identity rotation) is searched like any other -- knn_search never reads rot (index.jl:204-258) --, keeps its
        # matrix (written back below, byte for byte) and refuses push! / encode (quantize_data would need the rotation)
        assert idx.rotated == rotated
        exp_rot = np.array([[g.rotation(r, c) if rotated else float(r == c) for r in range(1, g.NROWS + 1)] for c in range(1, g.NROWS + 1)], np.float32)
        assert np.array_equal(idx.rotation(), exp_rot)
        if rotated:
            with pytest.raises(native.IVFADCError):
                idx._append(exp["centroids"][:1].copy(), np.array([999], np.uint32))
            with pytest.raises(native.IVFADCError):
                idx.encode(exp["centroids"][:1].copy())
        assert idx.index_type == np.dtype({"UInt16": np.uint16, "UInt32": np.uint32}[I]) and len(idx) == len(exp["ids"])
        assert np.array_equal(idx._centroids, exp["centroids"]) and np.array_equal(idx._codebooks, exp["codebooks"])
        assert np.array_equal(idx._labels, exp["labels"])
        o, c, i = idx._lists()
        assert np.array_equal(o, exp["offsets"]) and np.array_equal(c, exp["codes"]) and np.array_equal(i, exp["ids"])
        oidx = ora.OracleIndex(exp["centroids"], exp["codebooks"], exp["labels"], exp["offsets"], exp["codes"], exp["ids"])
        qs = (exp["centroids"] + np.float32(0.25)).astype(np.float32)
        helpers.assert_same_results(idx.search_raw(qs, 3, 2), oidx.knn_search(qs, 3, 2), what=name)
        if T == "Float32":
            out = os.path.join(str(tmp_path), "back.bin")
            native.save_ivfadc_index(out, idx)
            assert open(out, "rb").read() == open(os.path.join(here, name), "rb").read()
