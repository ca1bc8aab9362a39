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
