"""ctypes binding of libivfadc_hip.so (include/ivfadc_hip.h).

There is no CPU fallback: if the library is missing or no HIP device is present
every compute entry point raises.  ``build()`` compiles the library in-tree with
hipcc for gfx950 (cross-compiles without a GPU)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libivfadc_hip.so")
SOURCES = [os.path.join(CSRC, "ivfadc_hip.hip"), os.path.join(CSRC, "kernels.hip.h"), os.path.join(CSRC, "train.hip.h"), os.path.join(CSRC, "wave_sort.hip.h"),
           os.path.join(CSRC, "generic.hip.h"), os.path.join(CSRC, "lbscan.hip.h"), os.path.join(CSRC, "nfscan.hip.h"), os.path.join(CSRC, "smallq.hip.h"), os.path.join(CSRC, "twolevel.hip.h"), os.path.join(CSRC, "wg8scan.hip.h"), os.path.join(CSRC, "wg8q8scan.hip.h"),
           os.path.join(os.path.dirname(_HERE), "include", "ivfadc_hip.h")]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"]

OK, ERR_ASSERT, ERR_INVALID, ERR_HIP, ERR_STATE = 0, 1, 2, 3, 4


class IVFADCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ivfadc_hip status %d: %s" % (code, msg))
        self.code = code


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(s) > t for s in SOURCES)


def build(force=False, verbose=False):
    """Compile csrc/ivfadc_hip.hip -> csrc/libivfadc_hip.so for gfx950."""
    if not force and not needs_build():
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO_PATH + ".tmp", SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


class Stats(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("coarse_ms", C.c_double), ("scan_launches", C.c_int64),
                ("scanned_points", C.c_int64), ("queries", C.c_int64), ("last_qg", C.c_int32),
                ("last_chunk", C.c_int32), ("last_scan_grid", C.c_int32), ("last_scan_lds", C.c_int32),
                ("coarse_fallbacks", C.c_int64), ("coarse_mfma", C.c_int32), ("inplace_appends", C.c_int32),
                ("last_striped", C.c_int32), ("coarse_listed", C.c_int32), ("pruned_points", C.c_int64),
                ("lb_survivors", C.c_int64), ("last_lb", C.c_int32), ("last_rider", C.c_int32),
                ("lb_build_ms", C.c_double), ("lb_build_launches", C.c_int64),
                ("coarse_prefetched", C.c_int32), ("last_nf", C.c_int32), ("last_twolevel", C.c_int32), ("twolevel_groups", C.c_int32),
                ("coarse_visited", C.c_int64), ("twolevel_probe_fraction", C.c_float), ("coarse_f16", C.c_int32)]


class HostStats(C.Structure):
    _fields_ = [("stage_in_us", C.c_double), ("enqueue_us", C.c_double), ("wait_us", C.c_double), ("stage_out_us", C.c_double),
                ("calls", C.c_int64), ("batches", C.c_int64), ("queries_direct", C.c_int64), ("results_direct", C.c_int64),
                ("zero_copy", C.c_int64), ("streams_replaced", C.c_int64)]


ABI_VERSION = 4     # include/ivfadc_hip.h: IVFADC_ABI_VERSION this binding was written for

_lib = None


def lib():
    """Load the shared library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise IVFADCError(ERR_STATE, "libivfadc_hip.so is not built (run __graft_entry__.build()); "
                                     "there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    vp, fp, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    i32p, u32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)
    L.ivfadc_last_error.restype = C.c_char_p
    L.ivfadc_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, u8p]
    L.ivfadc_set_lists.argtypes = [vp, i64p, u8p, u32p]
    L.ivfadc_train.argtypes = [C.c_int, C.c_int, C.c_int64, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, fp, fp]
    L.ivfadc_train.restype = C.c_int
    L.ivfadc_synth_lists.argtypes = [vp, i64p, C.c_uint64]
    L.ivfadc_encode.argtypes = [vp, C.c_int64, fp, i32p, u8p]
    L.ivfadc_append.argtypes = [vp, C.c_int64, fp, u32p, i32p, u8p]
    L.ivfadc_search.argtypes = [vp, C.c_int64, fp, C.c_int, C.c_int, u32p, fp, i32p]
    L.ivfadc_debug_lb_table.argtypes = [vp, fp, C.c_int, u8p, fp]
    L.ivfadc_get_dims.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.ivfadc_get_quantizers.argtypes = [vp, fp, fp, u8p]
    L.ivfadc_search_device.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp]
    L.ivfadc_sync.argtypes = [vp]
    L.ivfadc_set_stream.argtypes = [vp, vp]
    L.ivfadc_ntotal.argtypes = [vp, i64p, i64p]
    L.ivfadc_get_lists.argtypes = [vp, i64p, u8p, u32p]
    L.ivfadc_set_profiling.argtypes = [vp, C.c_int]
    L.ivfadc_reset_stats.argtypes = [vp]
    L.ivfadc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ivfadc_set_tuning.argtypes = [vp, C.c_int, C.c_int]
    L.ivfadc_set_workspace_limit.argtypes = [vp, C.c_uint64]
    L.ivfadc_set_coarse_mode.argtypes = [vp, C.c_int]
    L.ivfadc_set_pruning.argtypes = [vp, C.c_int]
    L.ivfadc_set_next_queries.argtypes = [vp, C.c_int64, C.c_void_p, C.c_uint64]
    L.ivfadc_set_query_token.argtypes = [vp, C.c_uint64]
    L.ivfadc_search_batches.argtypes = [vp, C.c_int, i64p, fp, C.c_int, C.c_int, u32p, fp, i32p]
    L.ivfadc_set_table_mode.argtypes = [vp, C.c_int]
    L.ivfadc_delete_ids.argtypes = [vp, C.c_int64, u32p, i64p]
    L.ivfadc_shift_ids.argtypes = [vp, C.c_int32]
    L.ivfadc_save_index.argtypes = [vp, C.c_char_p, C.c_int]
    L.ivfadc_load_index.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    L.ivfadc_get_rotation.argtypes = [vp, C.POINTER(C.c_int), fp]
    L.ivfadc_destroy.argtypes = [vp]
    L.ivfadc_destroy.restype = None
    L.ivfadc_abi_version.argtypes = []
    L.ivfadc_abi_version.restype = C.c_int
    if L.ivfadc_abi_version() != ABI_VERSION:
        raise IVFADCError(ERR_STATE, "libivfadc_hip.so has ABI version %d, this binding was written for %d: rebuild "
                                     "(__graft_entry__.build())" % (L.ivfadc_abi_version(), ABI_VERSION))
    L.ivfadc_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.ivfadc_host_free.argtypes = [vp]
    L.ivfadc_host_register.argtypes = [vp, C.c_size_t]
    L.ivfadc_host_unregister.argtypes = [vp]
    L.ivfadc_get_host_stats.argtypes = [vp, C.POINTER(HostStats)]
    L.ivfadc_reset_host_stats.argtypes = [vp]
    for name in ("host_alloc", "host_free", "host_register", "host_unregister", "get_host_stats", "reset_host_stats"):
        getattr(L, "ivfadc_" + name).restype = C.c_int
    L.ivfadc_mg_create.argtypes = [C.POINTER(vp), C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, u8p]
    L.ivfadc_mg_set_lists.argtypes = [vp, i64p, u8p, u32p]
    L.ivfadc_mg_append.argtypes = [vp, C.c_int64, fp, u32p, i32p, u8p]
    L.ivfadc_mg_delete_ids.argtypes = [vp, C.c_int64, u32p, i64p]
    L.ivfadc_mg_shift_ids.argtypes = [vp, C.c_int32]
    L.ivfadc_mg_search.argtypes = [vp, C.c_int64, fp, C.c_int, C.c_int, u32p, fp, i32p]
    L.ivfadc_mg_synth_lists.argtypes = [vp, i64p, C.c_uint64]
    L.ivfadc_mg_num_devices.argtypes = [vp]
    L.ivfadc_mg_set_gather.argtypes = [vp, C.c_int]
    L.ivfadc_mg_collectives.argtypes = [vp, i64p]
    L.ivfadc_comm_unique_id.argtypes = [u8p]
    L.ivfadc_comm_init.argtypes = [vp, C.c_int, C.c_int, u8p]
    L.ivfadc_search_device_allgather.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.ivfadc_search_device_allgather_on.argtypes = [vp, vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.ivfadc_clone_view.argtypes = [vp, C.POINTER(vp)]
    L.ivfadc_set_list_partition.argtypes = [vp, C.c_int, C.c_int]
    L.ivfadc_search_device_partial.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp]
    L.ivfadc_merge_partials_device.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.ivfadc_listpart_block_words.argtypes = [C.c_int64, C.c_int]
    L.ivfadc_listpart_block_words.restype = C.c_int64
    L.ivfadc_search_device_listpart.argtypes = [vp, C.c_int64, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.ivfadc_comm_wait.argtypes = [vp, i64p]
    L.ivfadc_comm_destroy.argtypes = [vp]
    L.ivfadc_mg_destroy.argtypes = [vp]
    L.ivfadc_mg_destroy.restype = None
    for name in ("mg_create", "mg_set_lists", "mg_append", "mg_search", "mg_delete_ids", "mg_shift_ids", "mg_synth_lists",
                 "mg_num_devices", "mg_set_gather", "mg_collectives", "set_list_partition", "search_device_partial", "merge_partials_device", "search_device_listpart", "comm_unique_id", "comm_init", "search_device_allgather", "search_device_allgather_on", "clone_view", "comm_wait", "comm_destroy"):
        getattr(L, "ivfadc_" + name).restype = C.c_int
    for name in ("create", "set_lists", "synth_lists", "encode", "append", "search", "search_device", "sync", "set_stream",
                 "ntotal", "get_lists", "set_profiling", "reset_stats", "get_stats", "set_tuning", "set_workspace_limit", "set_coarse_mode", "set_pruning", "set_table_mode", "set_next_queries", "set_query_token", "search_batches", "save_index", "load_index", "delete_ids", "shift_ids"):
        getattr(L, "ivfadc_" + name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    if rc == OK:
        return
    msg = lib().ivfadc_last_error().decode("utf-8", "replace")
    if rc == ERR_ASSERT:
        raise AssertionError(msg)      # the reference raises AssertionError for these
    raise IVFADCError(rc, msg)


def ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class PinnedArray:
    """A numpy array in page-locked host memory owned by the library (ivfadc_host_alloc): a host-pointer search that is given
    such arrays reads its queries from them and writes its results into them with no staging copy.  Keep the object alive as
    long as `.a` (or any view of it) is in use; the memory goes back with the object."""

    def __init__(self, shape, dtype):
        import numpy as np
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        n = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = max(1, n * self.dtype.itemsize)
        p = C.c_void_p()
        check(lib().ivfadc_host_alloc(self.nbytes, C.byref(p)))
        self._p = p
        buf = (C.c_char * self.nbytes).from_address(p.value)
        self.a = np.frombuffer(buf, dtype=self.dtype, count=n).reshape(self.shape)

    def close(self):
        if self._p is not None and self._p.value:
            self.a = None
            lib().ivfadc_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_register(a):
    """Page-locks a C-contiguous numpy array of the caller's (ivfadc_host_register); undo with host_unregister(a) BEFORE the array
    is freed or resized."""
    assert a.flags["C_CONTIGUOUS"]
    check(lib().ivfadc_host_register(C.c_void_p(a.ctypes.data), a.nbytes))


def host_unregister(a):
    check(lib().ivfadc_host_unregister(C.c_void_p(a.ctypes.data)))
