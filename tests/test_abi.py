"""CPU checks of the drop-in boundary: the library builds for gfx950, loads, exports every
symbol include/ivfadc_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest


def _declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "ivfadc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ivfadc_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(native):
    lib = native.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), "libivfadc_hip.so does not export %s" % s


def test_argument_validation_needs_no_gpu(native):
    lib = native.load_library()
    h = C.c_void_p()
    z = np.zeros(16, np.float32)
    lab = np.zeros(16, np.uint8)
    fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    # d % m != 0 and ksub > 256 are rejected before any device call
    rc = lib.ivfadc_create(C.byref(h), 0, 5, 2, 2, 2, z.ctypes.data_as(fp), z.ctypes.data_as(fp), lab.ctypes.data_as(u8p))
    assert rc == 2 and b"d % m" in lib.ivfadc_last_error()
    rc = lib.ivfadc_create(C.byref(h), 0, 4, 2, 2, 300, z.ctypes.data_as(fp), z.ctypes.data_as(fp), lab.ctypes.data_as(u8p))
    assert rc == 2
    # m > d is one of the reference's constructor assertions (index.jl:120)
    rc = lib.ivfadc_create(C.byref(h), 0, 2, 2, 3, 2, z.ctypes.data_as(fp), z.ctypes.data_as(fp), lab.ctypes.data_as(u8p))
    assert rc == 1


def test_no_cpu_fallback(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the loud-failure path is exercised on CPU-only hosts")
    cent = np.zeros((2, 4), np.float32)
    cbs = np.zeros((2, 4, 2), np.float32)
    labels = np.tile(np.arange(4, dtype=np.uint8), (2, 1))
    with pytest.raises(native.IVFADCError):
        native.IVFADCIndex.from_arrays(cent, cbs, labels)


def test_constructor_assertions_match_reference(native):
    """test/index.jl:37-40: kc<2, k>n, m>d, index_type too small -> AssertionError, before any device work."""
    data = np.random.default_rng(0).random((300, 2), dtype=np.float32)
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=1, k=2, m=1)
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=2, k=301, m=1)
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, kc=2, k=300, m=3)
    with pytest.raises(AssertionError):
        native.IVFADCIndex(data, index_type=np.uint8)


def test_header_is_plain_c(tmp_path):
    """include/ivfadc_hip.h must compile as C99 (the Julia ccall / cgo / ctypes side sees a C ABI, not C++), and the C
    smoke program must compile against it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(str(tmp_path), "abi_smoke.o")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", obj])
    assert os.path.getsize(obj) > 0


@pytest.mark.gpu
def test_c_program_through_the_abi(tmp_path, native):
    """A plain C program (tests/c/abi_smoke.c) builds an index by ivfadc_append, searches it and prints the result; the
    same arrays go through the oracle here."""
    import subprocess
    import numpy as np
    import oracle.oracle as ora
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "ivfadc.jl_amd", "csrc")
    exe = os.path.join(str(tmp_path), "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
                           "-o", exe, "-L", csrc, "-livfadc_hip", "-Wl,-rpath," + csrc])
    out = subprocess.check_output([exe, "0", "7"], text=True)
    lines = {l.split()[0]: l.split()[1:] for l in out.splitlines() if not l.startswith("q ")}
    D, KC, M, KSUB, N, NQ, K, W = 12, 7, 3, 16, 90, 5, 4, 3
    fl = lambda key: np.array([float.fromhex(x) for x in lines[key]], np.float32)
    cent = fl("cent").reshape(KC, D)
    cbs = fl("cbs").reshape(M, KSUB, D // M)
    qs = fl("qs").reshape(NQ, D)
    offsets = np.array(lines["offsets"], np.int64)
    codes = np.array(lines["codes"], np.uint8).reshape(N, M)
    ids = np.array(lines["ids"], np.uint32)
    assert int(lines["n"][0]) == N and offsets[-1] == N and sorted(ids.tolist()) == list(range(N))
    labels = np.tile(np.arange(KSUB, dtype=np.uint8), (M, 1))
    oidx = ora.OracleIndex(cent, cbs, labels, offsets, codes, ids)
    eid, edist, ecnt = oidx.knn_search(qs, K, W)
    qlines = [l for l in out.splitlines() if l.startswith("q ")]
    assert len(qlines) == NQ
    for i, l in enumerate(qlines):
        tok = l.split()
        cnt = int(tok[3].rstrip(":"))
        assert cnt == ecnt[i]
        got_ids = [int(x) for x in tok[4::2]]
        got_d = [float.fromhex(x) for x in tok[5::2]]
        assert got_ids == eid[i, :cnt].tolist()
        assert np.array_equal(np.array(got_d, np.float32), edist[i, :cnt])


def test_julia_shim_file_matches_integration_doc_and_library(native):
    """julia/IVFADCHip.jl is the shim of INTEGRATION.md section 3 as a file: the two texts must not drift, and
    every symbol it ccall's must be declared in the header and exported by the built library."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jl = open(os.path.join(root, "julia", "IVFADCHip.jl")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```julia\n(.*?)```", doc, re.S).group(1)
    assert block.strip() in jl, "julia/IVFADCHip.jl and INTEGRATION.md section 3 differ"
    syms = sorted(set(re.findall(r"ccall\(\(:(ivfadc_[a-z_]+),", jl)))
    assert len(syms) >= 6, syms
    declared = set(_declared_symbols())
    lib = native.load_library()
    for sname in syms:
        assert sname in declared, "%s is not declared in include/ivfadc_hip.h" % sname
        assert hasattr(lib, sname), "%s is not exported" % sname
    assert os.path.exists(os.path.join(root, "tools", "julia", "make_fixture.jl"))
