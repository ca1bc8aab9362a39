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


# ---- static check of the Julia shim's ccall signatures against the C prototypes (a wrong argument tuple is silent UB) -----------
def _c_prototypes():
    """{name: (ret, [arg types])} from include/ivfadc_hip.h, every type canonicalised (const and parameter names dropped)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "ivfadc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)

    def canon(t):
        t = re.sub(r"\bconst\b", " ", t)
        t = t.replace("*", " * ")
        toks = t.split()
        # drop a trailing parameter name (an identifier that is not a type word and not '*')
        types = {"void", "int", "char", "float", "double", "size_t", "int32_t", "int64_t", "uint8_t", "uint32_t", "uint64_t",
                 "ivfadc_t", "ivfadc_mg_t", "ivfadc_stats", "ivfadc_host_stats", "unsigned"}
        if len(toks) > 1 and toks[-1] != "*" and toks[-1] not in types:
            toks = toks[:-1]
        return " ".join(toks)

    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ivfadc_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        a = [canon(x) for x in args.split(",")] if args.strip() and args.strip() != "void" else []
        protos[name] = (canon(ret), a)
    return protos


_JL_TO_C = {
    "Cint": "int", "Cvoid": "void", "Cstring": "char *", "Csize_t": "size_t", "Int32": "int32_t", "Int64": "int64_t", "UInt64": "uint64_t",
    "Ptr{Float32}": "float *", "Ptr{UInt32}": "uint32_t *", "Ptr{UInt8}": "uint8_t *", "Ptr{Int64}": "int64_t *", "Ptr{Int32}": "int32_t *",
    "Ref{Int32}": "int32_t *", "Ptr{Cint}": "int *", "Ref{Cint}": "int *",
}
# a Julia Ptr{Cvoid} is an opaque handle or untyped memory; Ref{Ptr{Cvoid}} is a pointer to one
_JL_OPAQUE = {"Ptr{Cvoid}": {"ivfadc_t *", "ivfadc_mg_t *", "void *"}, "Ref{Ptr{Cvoid}}": {"ivfadc_t * *", "ivfadc_mg_t * *", "void * *"}}


def _split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _julia_ccalls(jl):
    """[(symbol, ret, [arg types], n_values_passed, line)] for every ccall((:sym, LIBIVFADC), ...) in the text"""
    calls = []
    for m in re.finditer(r"ccall\(", jl):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[j], 0)
            j += 1
        parts = _split_top(jl[i:j - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*LIBIVFADC\s*\)", parts[0])
        assert sym, "ccall without (:symbol, LIBIVFADC): %s" % parts[0]
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), parts[2]
        types = _split_top(argt[1:-1])
        calls.append((sym.group(1), parts[1].strip(), types, len(parts) - 3, jl.count("\n", 0, m.start()) + 1))
    return calls


def test_julia_ccall_signatures_match_the_header():
    """Every ccall of julia/IVFADCHip.jl and of tools/julia/make_fixture.jl: return type, arity (types AND values passed) and each
    argument's C type equal the prototype in include/ivfadc_hip.h.  The shim has never executed (no julia in the image): this is the
    check that a drifting prototype -- e.g. the token argument ivfadc_set_next_queries gained in round 4 -- cannot go unnoticed."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    protos = _c_prototypes()
    assert protos["ivfadc_search"] == ("int", ["ivfadc_t *", "int64_t", "float *", "int", "int", "uint32_t *", "float *", "int32_t *"])
    assert protos["ivfadc_host_alloc"] == ("int", ["size_t", "void * *"])
    assert protos["ivfadc_last_error"] == ("char *", []) and protos["ivfadc_destroy"] == ("void", ["ivfadc_t *"])
    seen = set()
    for rel in (os.path.join("julia", "IVFADCHip.jl"), os.path.join("tools", "julia", "make_fixture.jl")):
        jl = open(os.path.join(root, rel)).read()
        for sym, ret, types, nvals, line in _julia_ccalls(jl):
            where = "%s:%d ccall(:%s)" % (rel, line, sym)
            assert sym in protos, "%s: not declared in include/ivfadc_hip.h" % where
            cret, cargs = protos[sym]
            assert _JL_TO_C.get(ret) == cret, "%s returns %s, the header says %s" % (where, ret, cret)
            assert len(types) == len(cargs), "%s passes %d argument types, the header declares %d" % (where, len(types), len(cargs))
            assert nvals == len(types), "%s: %d values for %d argument types" % (where, nvals, len(types))
            for k, (jt, ct) in enumerate(zip(types, cargs)):
                ok = (_JL_TO_C.get(jt) == ct) or (ct in _JL_OPAQUE.get(jt, ()))
                assert ok, "%s: argument %d is %s, the header says %s" % (where, k + 1, jt, ct)
            seen.add(sym)
    # the calls the shim's surface rests on are all there (and were all checked)
    for need in ("ivfadc_abi_version", "ivfadc_create", "ivfadc_set_lists", "ivfadc_search", "ivfadc_search_batches", "ivfadc_append",
                 "ivfadc_shift_ids", "ivfadc_delete_ids", "ivfadc_destroy", "ivfadc_last_error", "ivfadc_host_alloc", "ivfadc_host_free"):
        assert need in seen, need
    # the version the shim checks for is the header's
    hdr = open(os.path.join(root, "include", "ivfadc_hip.h")).read()
    ver = int(re.search(r"#define\s+IVFADC_ABI_VERSION\s+(\d+)", hdr).group(1))
    jl = open(os.path.join(root, "julia", "IVFADCHip.jl")).read()
    assert int(re.search(r"const ABI_VERSION = (\d+)", jl).group(1)) == ver
    from ivfadc_jl_amd import _native as nat
    assert nat.ABI_VERSION == ver


def test_ccall_checker_catches_drift():
    """the checker itself: a wrong arity, a wrong pointer type and a wrong integer width are all seen"""
    protos = _c_prototypes()
    good = 'ccall((:ivfadc_shift_ids, LIBIVFADC), Cint, (Ptr{Cvoid}, Int32), h.ptr, shift)'
    (sym, ret, types, nvals, _), = _julia_ccalls(good)
    assert (sym, ret, types, nvals) == ("ivfadc_shift_ids", "Cint", ["Ptr{Cvoid}", "Int32"], 2)
    assert [_JL_TO_C.get(t, None) or t for t in types][1] == protos[sym][1][1]
    bad_width = _julia_ccalls('ccall((:ivfadc_shift_ids, LIBIVFADC), Cint, (Ptr{Cvoid}, Int64), h.ptr, shift)')[0]
    assert _JL_TO_C[bad_width[2][1]] != protos["ivfadc_shift_ids"][1][1]
    old_hint = _julia_ccalls('ccall((:ivfadc_set_next_queries, LIBIVFADC), Cint, (Ptr{Cvoid}, Int64, Ptr{Float32}), h.ptr, n, q)')[0]
    assert len(old_hint[2]) != len(protos["ivfadc_set_next_queries"][1])          # the round-4 in-place change would have been caught
    nested = _julia_ccalls('_check(ccall((:ivfadc_host_alloc, LIBIVFADC), Cint, (Csize_t, Ref{Ptr{Cvoid}}), max(a, (b + 1)), out))')[0]
    assert nested[3] == 2 and nested[2] == ["Csize_t", "Ref{Ptr{Cvoid}}"]
