"""Host-side mirror of the reference's public surface for the knn_search hot path.

Same names, argument meaning and error behaviour as /root/reference/src:
  IVFADCIndex(data; kc, k, m, ...)        index.jl:103-165
  knn_search(ivfadc, point|points, k; w)  index.jl:204-273
  push!/pushfirst!                        utils.jl:114-145   (push / pushfirst here)
  pop!/popfirst!/delete_from_index!       utils.jl:29-105    (host bookkeeping only)
  length / size / show                    index.jl:56-77

Layout note: Julia matrices are d x n column-major; the same memory is an (n, d)
row-major numpy array, which is what every function here takes (one vector per row).
All arithmetic of the search and of the push! encoding runs in the HIP library behind
the C ABI (include/ivfadc_hip.h); nothing here computes a distance.
"""
import ctypes as C
import math
import threading

import numpy as np

from . import _native as nat
from . import trainer

DEFAULT_COARSE_K = 2                     # defaults.jl:2-10
DEFAULT_QUANTIZATION_K = 256
DEFAULT_QUANTIZATION_M = 1
DEFAULT_QUANTIZATION_METHOD = "pq"
DEFAULT_COARSE_DISTANCE = "SqEuclidean"
DEFAULT_COARSE_QUANTIZER = "naive"
DEFAULT_QUANTIZATION_DISTANCE = "SqEuclidean"
DEFAULT_COARSE_MAXITER = 25
DEFAULT_QUANTIZATION_MAXITER = 25

_TYPE_TO_BITS = {np.dtype(np.uint8): 8, np.dtype(np.uint16): 16, np.dtype(np.uint32): 32}
_JULIA_NAMES = {np.dtype(np.uint8): "UInt8", np.dtype(np.uint16): "UInt16", np.dtype(np.uint32): "UInt32"}


class NaiveQuantizer:
    """coarsequantizers.jl:18-20: brute-force coarse quantizer; vectors is (kc, d)."""

    def __init__(self, vectors):
        self.vectors = vectors

    def __repr__(self):
        kc, d = self.vectors.shape
        return "NaiveQuantizer{SqEuclidean,Float32}, %d×%d cluster centres" % (d, kc)


class CodeBook:
    """QuantizedArrays.CodeBook(codes, vectors): vectors is (ksub, dsub) (== dsub x ksub column-major)."""

    def __init__(self, codes, vectors):
        self.codes = codes
        self.vectors = vectors


class ResidualQuantizer:
    def __init__(self, codebooks, k, dims):
        self.codebooks = codebooks
        self.k = k
        self.dims = dims


class InvertedList:
    """index.jl:8-11: idxs (0-based ids) and codes ((len, m) bytes) of one Voronoi cell."""

    def __init__(self, idxs, codes):
        self.idxs = idxs
        self.codes = codes

    def __repr__(self):
        return "InvertedList{%s,UInt8}, %d vectors" % (_JULIA_NAMES[self.idxs.dtype], len(self.idxs))


class IVFADCIndex:
    def __init__(self, data, kc=DEFAULT_COARSE_K, k=DEFAULT_QUANTIZATION_K, m=DEFAULT_QUANTIZATION_M,
                 coarse_quantizer=DEFAULT_COARSE_QUANTIZER, coarse_distance=DEFAULT_COARSE_DISTANCE,
                 quantization_distance=DEFAULT_QUANTIZATION_DISTANCE, quantization_method=DEFAULT_QUANTIZATION_METHOD,
                 coarse_maxiter=DEFAULT_COARSE_MAXITER, quantization_maxiter=DEFAULT_QUANTIZATION_MAXITER,
                 index_type=np.uint32, seed=0, device=0):
        data = np.ascontiguousarray(data, np.float32)
        assert data.ndim == 2, "data must be a matrix (one vector per row)"
        nvectors, nrows = data.shape
        index_type = np.dtype(index_type)
        assert index_type in _TYPE_TO_BITS, "index_type must be uint8, uint16 or uint32"
        bits_required = int(math.ceil(math.log2(nvectors))) if nvectors > 1 else 0
        # index.jl:118-125
        assert kc >= 2, "Number of coarse clusters has to be >= 2"
        assert k <= nvectors, "Number of quantization levels  has to be <= %d" % nvectors
        assert 1 <= m <= nrows, "Number of codebooks has to be between 1 and %d" % nrows
        assert coarse_quantizer in ("naive", "hnsw"), "Coarse quantizer can be :naive or :hnsw only"
        assert coarse_maxiter > 0, "Number of clustering iterations has to be > 0"
        assert quantization_maxiter > 0, "Number of clustering iterations has to be > 0"
        assert _TYPE_TO_BITS[index_type] >= bits_required, \
            "%d vectors require at least %d index bits" % (nvectors, bits_required)
        # :hnsw (coarsequantizers.jl:58-92) asks for an APPROXIMATE search of the coarse centroids through a graph; here the
        # centroids are searched exhaustively on the GPU, which is what that graph approximates -- the request is
        # accepted and answered with the naive quantizer's (exact) cells
        self.requested_coarse_quantizer = coarse_quantizer
        if coarse_distance != "SqEuclidean" or quantization_distance != "SqEuclidean" or quantization_method != "pq":
            raise NotImplementedError("the HIP path implements SqEuclidean / :pq only (the reference defaults)")
        if nrows % m != 0:
            raise NotImplementedError("d % m != 0: QuantizedArrays.rowrange for ragged sub-spaces is unverifiable")
        if k > 256:
            raise NotImplementedError("k > 256 does not fit UInt8 codes")
        cent, cbs, labels = trainer.train_ivfadc_hip(data, kc, k, m, coarse_maxiter, quantization_maxiter, seed, device)
        self._init_native(cent, cbs, labels, index_type, device)
        if nvectors:
            self._append(data, np.arange(nvectors, dtype=np.uint32))

    # ---- construction from existing arrays (loaded index, tests) ------------------------------
    @classmethod
    def from_arrays(cls, centroids, codebooks, labels, offsets=None, codes=None, ids=None,
                    index_type=np.uint32, device=0):
        self = cls.__new__(cls)
        self._init_native(centroids, codebooks, labels, np.dtype(index_type), device)
        if offsets is not None:
            self.set_lists(offsets, codes, ids)
        return self

    @classmethod
    def from_file(cls, filename, device=0):
        """An index saved by IVFADC.jl (or by save_ivfadc_index) -- native reader, ivfadc_load_index."""
        self = cls.__new__(cls)
        h = C.c_void_p()
        bits = C.c_int(0)
        nat.check(nat.lib().ivfadc_load_index(C.byref(h), int(device), str(filename).encode(), C.byref(bits)))
        self._h = h
        self.index_type = np.dtype({8: np.uint8, 16: np.uint16, 32: np.uint32}[bits.value])
        self.device = device
        # quantizer arrays for the reference-shaped views (host copies of what the native reader uploaded)
        dims = [C.c_int32(0) for _ in range(4)]
        nat.check(nat.lib().ivfadc_get_dims(h, *[C.byref(x) for x in dims]))
        d, kc, m, ksub = (int(x.value) for x in dims)
        self._centroids = np.zeros((kc, d), np.float32)
        self._codebooks = np.zeros((m, ksub, d // m), np.float32)
        self._labels = np.zeros((m, ksub), np.uint8)
        nat.check(nat.lib().ivfadc_get_quantizers(h, self._centroids.ctypes.data_as(C.POINTER(C.c_float)),
                                                  self._codebooks.ctypes.data_as(C.POINTER(C.c_float)),
                                                  self._labels.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.kc, self.d = self._centroids.shape
        self.m, self.ksub, self.dsub = self._codebooks.shape
        self._mirror = None
        # an :opq index carries a rotation: searched as it is (knn_search never reads rot, index.jl:204-258); push! would need it
        rotated = C.c_int(0)
        nat.check(nat.lib().ivfadc_get_rotation(h, C.byref(rotated), None))
        self.rotated = bool(rotated.value)
        return self

    def rotation(self):
        """The residual quantizer's rotation matrix (d, d), column by column as persistency.jl:62-64 stores it (identity for :pq)."""
        out = np.zeros((self.d, self.d), np.float32)
        nat.check(nat.lib().ivfadc_get_rotation(self._h, None, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def _init_native(self, centroids, codebooks, labels, index_type, device):
        self._centroids = np.ascontiguousarray(centroids, np.float32)
        self._codebooks = np.ascontiguousarray(codebooks, np.float32)
        self._labels = np.ascontiguousarray(labels, np.uint8)
        self.kc, self.d = self._centroids.shape
        self.m, self.ksub, self.dsub = self._codebooks.shape
        assert self.m * self.dsub == self.d and self._labels.shape == (self.m, self.ksub)
        self.index_type = np.dtype(index_type)
        self._h = C.c_void_p()
        nat.check(nat.lib().ivfadc_create(C.byref(self._h), int(device), self.d, self.kc, self.m, self.ksub,
                                          nat.ptr(self._centroids, C.c_float), nat.ptr(self._codebooks, C.c_float),
                                          nat.ptr(self._labels, C.c_uint8)))
        self._mirror = None

    def clone_view(self):
        """A read-only view of this index (ivfadc_clone_view): the same device arrays, a stream and a workspace of its own, so that
        two batches can be in flight on one replica.  Any change to this index afterwards makes the view refuse to search."""
        v = IVFADCIndex.__new__(IVFADCIndex)
        v._h = C.c_void_p()
        nat.check(nat.lib().ivfadc_clone_view(self._h, C.byref(v._h)))
        for name in ("_centroids", "_codebooks", "_labels", "kc", "d", "m", "ksub", "dsub", "index_type"):
            setattr(v, name, getattr(self, name))
        v.device = getattr(self, "device", 0)
        v.requested_coarse_quantizer = getattr(self, "requested_coarse_quantizer", "naive")
        v._mirror = None
        v._view_of = self          # keeps the index alive as long as the view
        return v

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                nat.lib().ivfadc_destroy(h)
            except Exception:
                pass
            self._h = C.c_void_p()

    # ---- lists -------------------------------------------------------------------------------
    def set_lists(self, offsets, codes, ids):
        offsets = np.ascontiguousarray(offsets, np.int64)
        n = int(offsets[-1])
        codes = np.ascontiguousarray(codes, np.uint8).reshape(n, self.m)
        ids = np.ascontiguousarray(ids, np.uint32)
        nat.check(nat.lib().ivfadc_set_lists(self._h, nat.ptr(offsets, C.c_int64), nat.ptr(codes, C.c_uint8),
                                             nat.ptr(ids, C.c_uint32)))
        self._mirror = None

    def synth_lists(self, offsets, seed):
        offsets = np.ascontiguousarray(offsets, np.int64)
        nat.check(nat.lib().ivfadc_synth_lists(self._h, nat.ptr(offsets, C.c_int64), C.c_uint64(int(seed))))
        self._mirror = None

    def _lists(self):
        if self._mirror is None:
            n = len(self)
            offsets = np.zeros(self.kc + 1, np.int64)
            codes = np.zeros((n, self.m), np.uint8)
            ids = np.zeros(n, np.uint32)
            nat.check(nat.lib().ivfadc_get_lists(self._h, nat.ptr(offsets, C.c_int64), nat.ptr(codes, C.c_uint8),
                                                 nat.ptr(ids, C.c_uint32)))
            self._mirror = (offsets, codes, ids)
        return self._mirror

    def _append(self, pts, ids):
        pts = np.ascontiguousarray(pts, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        nat.check(nat.lib().ivfadc_append(self._h, pts.shape[0], nat.ptr(pts, C.c_float), nat.ptr(ids, C.c_uint32),
                                          None, None))
        self._mirror = None

    def _delete_ids(self, ids):
        """delete_from_index! on 0-based ids (ivfadc_delete_ids: in place on the device); returns how many went."""
        ids = np.ascontiguousarray(ids, np.uint32)
        removed = C.c_int64(0)
        nat.check(nat.lib().ivfadc_delete_ids(self._h, ids.shape[0], nat.ptr(ids, C.c_uint32), C.byref(removed)))
        self._mirror = None
        return int(removed.value)

    def _shift_ids(self, delta):
        nat.check(nat.lib().ivfadc_shift_ids(self._h, int(delta)))
        self._mirror = None

    def encode(self, pts):
        """_encode_point for a batch: (list (n,) int32 0-based, codes (n, m) uint8)."""
        pts = np.ascontiguousarray(pts, np.float32)
        if pts.ndim == 1:
            pts = pts[None, :]
        assert pts.shape[1] == self.d
        lst = np.zeros(pts.shape[0], np.int32)
        codes = np.zeros((pts.shape[0], self.m), np.uint8)
        nat.check(nat.lib().ivfadc_encode(self._h, pts.shape[0], nat.ptr(pts, C.c_float), nat.ptr(lst, C.c_int32),
                                          nat.ptr(codes, C.c_uint8)))
        return lst, codes

    # ---- reference-shaped views ----------------------------------------------------------------
    @property
    def coarse_quantizer(self):
        return NaiveQuantizer(self._centroids)

    @property
    def residual_quantizer(self):
        cbs = [CodeBook(self._labels[i], self._codebooks[i]) for i in range(self.m)]
        return ResidualQuantizer(cbs, self.ksub, (self.d, len(self)))

    @property
    def inverse_index(self):
        offsets, codes, ids = self._lists()
        return [InvertedList(ids[offsets[l]:offsets[l + 1]].astype(self.index_type),
                             codes[offsets[l]:offsets[l + 1]]) for l in range(self.kc)]

    def __len__(self):                                   # index.jl:56
        n = C.c_int64(0)
        nat.check(nat.lib().ivfadc_ntotal(self._h, C.byref(n), None))
        return int(n.value)

    @property
    def size(self):                                      # index.jl:65
        return (self.d, len(self))

    def list_sizes(self):
        n = C.c_int64(0)
        sizes = np.zeros(self.kc, np.int64)
        nat.check(nat.lib().ivfadc_ntotal(self._h, C.byref(n), nat.ptr(sizes, C.c_int64)))
        return sizes

    def __repr__(self):                                  # index.jl:69-77
        idxsize = self.index_type.itemsize
        # an index built with coarse_quantizer=:hnsw says so: the request is served by the exhaustive (exact) search of the
        # centroids, which is what the graph of coarsequantizers.jl:58-92 approximates
        req = getattr(self, "requested_coarse_quantizer", "naive")
        cq = "naive" if req == "naive" else "%s (requested; served by the exact naive search)" % req
        return ("IVFADCIndex, %s coarse quantizer, %d-byte encoding (%d + 1×%d), %d Float32 vectors"
                % (cq, self.m + idxsize, idxsize, self.m, len(self)))

    # ---- search --------------------------------------------------------------------------------
    def _io_lock(self):
        """The lock that goes with the page-locked blocks of _io: they belong to the index, ctypes releases the GIL during the native
        call, and a second Python thread calling knn_search on the same index would repack (or free and regrow) the query block the
        first call's kernels are reading.  Held from the packing to the last read of the result blocks."""
        lk = self.__dict__.get("_pin_lock")
        if lk is None:
            lk = self.__dict__.setdefault("_pin_lock", threading.Lock())
        return lk

    def _io(self, nq, ka):
        """Page-locked query / result arrays of this index (ivfadc_host_alloc; grown on demand) -- what the Julia shim keeps per
        index: knn_search packs the caller's vectors straight into the query block and reads ids / distances / counts out of the
        result blocks, so the library stages nothing (include/ivfadc_hip.h: ivfadc_host_alloc).  Call with _io_lock() held."""
        pin = getattr(self, "_pin", None)
        if pin is None or pin[0].a.size < nq * self.d or pin[1].a.size < nq * ka or pin[3].a.size < nq:
            grow = lambda old, need: max(need + need // 2, old)
            o = [0, 0, 0] if pin is None else [pin[0].a.size, pin[1].a.size, pin[3].a.size]
            self._pin = None                      # (frees the old blocks first)
            pin = (nat.PinnedArray(grow(o[0], nq * self.d), np.float32), nat.PinnedArray(grow(o[1], nq * ka), np.uint32),
                   nat.PinnedArray(grow(o[1], nq * ka), np.float32), nat.PinnedArray(grow(o[2], nq), np.int32))
            self._pin = pin
        return (pin[0].a[:nq * self.d].reshape(nq, self.d), pin[1].a[:nq * ka].reshape(nq, ka), pin[2].a[:nq * ka].reshape(nq, ka),
                pin[3].a[:nq])

    def search_raw(self, queries, k, w=1):
        """(nq, d) -> ids (nq, k) uint32, dists (nq, k) float32, counts (nq,) int32 via ivfadc_search."""
        q = np.ascontiguousarray(queries, np.float32)
        assert q.ndim == 2 and q.shape[1] == self.d, "queries must be (nq, %d)" % self.d
        nq = q.shape[0]
        ka = max(int(k), 1)
        ids = np.zeros((nq, ka), np.uint32)
        dists = np.full((nq, ka), np.inf, np.float32)
        counts = np.zeros(nq, np.int32)
        nat.check(nat.lib().ivfadc_search(self._h, nq, nat.ptr(q, C.c_float), int(k), int(w), nat.ptr(ids, C.c_uint32),
                                          nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        return ids, dists, counts

    # ---- measurement / tuning -------------------------------------------------------------------
    def set_profiling(self, on):
        """0 / False: off; 1 / True: HIP events around the coarse and scan kernels; 2: also the matrix-core table build alone."""
        nat.check(nat.lib().ivfadc_set_profiling(self._h, int(on)))

    def reset_stats(self):
        nat.check(nat.lib().ivfadc_reset_stats(self._h))

    def get_stats(self):
        st = nat.Stats()
        nat.check(nat.lib().ivfadc_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in nat.Stats._fields_}

    def set_tuning(self, qg=0, chunk_points=0):
        nat.check(nat.lib().ivfadc_set_tuning(self._h, int(qg), int(chunk_points)))

    def set_coarse_mode(self, mode):
        """0: automatic (matrix-core score filter + certified exact refine for kc >= 2048; per-tile records instead of a score
        matrix where the stand-alone top-w reads them), 1: always the exact VALU kernel, 2: the filter from kc >= 128 on,
        3: f32 MFMA filter only, 4: as 0 but always with the full score matrix.  Results are identical in every mode."""
        nat.check(nat.lib().ivfadc_set_coarse_mode(self._h, int(mode)))

    def set_pruning(self, on):
        """Exact probe pruning of the query-major scan (default on): a list whose coarse distance exceeds the K-th best key
        so far cannot contribute (every ADC sum starts from it and only grows)."""
        nat.check(nat.lib().ivfadc_set_pruning(self._h, int(bool(on))))

    def set_next_queries(self, nq, d_queries_ptr, token):
        """Hint for the search made right after this call: the search after THAT one will be on the `nq` queries at device pointer
        `d_queries_ptr` (already there, unchanged until searched) -- their exact coarse distances are then computed behind the
        first search's scan launch (ivfadc_set_next_queries).  `token` (!= 0) is the caller's generation number of the buffer's
        contents: the hinted search picks the rows up only if it declares the same token (set_query_token).  Results are unchanged."""
        nat.check(nat.lib().ivfadc_set_next_queries(self._h, int(nq), C.c_void_p(d_queries_ptr) if d_queries_ptr else None,
                                                    C.c_uint64(int(token))))

    def set_query_token(self, token):
        """Declares the generation of the queries the NEXT search reads (ivfadc_set_query_token); consumed by that search."""
        nat.check(nat.lib().ivfadc_set_query_token(self._h, C.c_uint64(int(token))))

    def search_batches_raw(self, batches, k, w=1):
        """A run of consecutive batches (list of (nq_b, d) arrays) in ONE call (ivfadc_search_batches): batch b is searched with
        batch b + 1 named as its successor.  Returns per batch (ids, dists, counts) as search_raw does."""
        qs = [np.asarray(b, np.float32).reshape(-1, self.d) for b in batches]
        sizes = np.array([q.shape[0] for q in qs], np.int64)
        total = int(sizes.sum())
        ka = max(int(k), 1)
        # packed once, into the index's page-locked blocks (see _io); the results are copied out of them before they are reused
        with self._io_lock():
            allq, ids, dists, counts = self._io(total, ka)
            s = 0
            for q in qs:
                allq[s:s + q.shape[0]] = q
                s += q.shape[0]
            nat.check(nat.lib().ivfadc_search_batches(self._h, len(qs), nat.ptr(sizes, C.c_int64), nat.ptr(allq, C.c_float), int(k), int(w),
                                                      nat.ptr(ids, C.c_uint32), nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
            out, s = [], 0
            for n in sizes.tolist():
                out.append((ids[s:s + n].copy(), dists[s:s + n].copy(), counts[s:s + n].copy()))
                s += n
        return out

    def set_table_mode(self, mode):
        """0: automatic (filter tables where they exist and pay), 1: the reference's f32 tables in every lane, 2: as 0 plus the
        matrix-core lower-bound rounds for every shape they are instantiated for; 3 / 4: as 0 / 2 with those tables built from the
        three-product bf16 split instead of one f16 product per entry."""
        nat.check(nat.lib().ivfadc_set_table_mode(self._h, int(mode)))

    def debug_lb_table(self, query, cell):
        """Test hook: the 8-bit lower-bound ADC table of one (query, cell) pair as the matrix-core build makes it.
        Returns (table uint8 [m, 256], inv, sbase, nn, base [m], r2 [m])."""
        q = np.ascontiguousarray(query, np.float32)
        tab = np.zeros((self.m, 256), np.uint8)
        cf = np.zeros(3 + 2 * self.m, np.float32)
        nat.check(nat.lib().ivfadc_debug_lb_table(self._h, q.ctypes.data_as(C.POINTER(C.c_float)), int(cell),
                                                  tab.ctypes.data_as(C.POINTER(C.c_uint8)), cf.ctypes.data_as(C.POINTER(C.c_float))))
        return tab, float(cf[0]), float(cf[1]), float(cf[2]), cf[3:3 + self.m].copy(), cf[3 + self.m:].copy()

    def set_workspace_limit(self, nbytes):
        nat.check(nat.lib().ivfadc_set_workspace_limit(self._h, C.c_uint64(int(nbytes))))

    def sync(self):
        nat.check(nat.lib().ivfadc_sync(self._h))

    def set_stream(self, hip_stream):
        """Use a caller-owned hipStream_t (int handle), e.g. torch.cuda.current_stream().cuda_stream."""
        nat.check(nat.lib().ivfadc_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def search_device(self, nq, q_ptr, k, w, ids_ptr, dists_ptr, counts_ptr):
        """Asynchronous search on raw device pointers (ints); see ivfadc_search_device."""
        nat.check(nat.lib().ivfadc_search_device(self._h, int(nq), C.c_void_p(q_ptr), int(k), int(w), C.c_void_p(ids_ptr),
                                                 C.c_void_p(dists_ptr), C.c_void_p(counts_ptr)))


def _comm_methods():
    """one-process-per-GPU merge inside the library (ivfadc_comm_*): methods of IVFADCIndex"""
    def comm_init(self, nranks, rank, id128):
        buf = np.ascontiguousarray(id128, np.uint8)
        assert buf.shape == (128,)
        nat.check(nat.lib().ivfadc_comm_init(self._h, int(nranks), int(rank), nat.ptr(buf, C.c_uint8)))

    def search_device_allgather_on(self, searcher, nq, q_ptr, k, w, block_ptr, gathered_ptr, slot):
        """search on `searcher` (this index or a view of it), the collective on this index's communicator"""
        nat.check(nat.lib().ivfadc_search_device_allgather_on(self._h, searcher._h, int(nq), C.c_void_p(q_ptr), int(k), int(w),
                                                             C.c_void_p(block_ptr), C.c_void_p(gathered_ptr), int(slot)))

    def search_device_allgather(self, nq, q_ptr, k, w, block_ptr, gathered_ptr, slot):
        nat.check(nat.lib().ivfadc_search_device_allgather(self._h, int(nq), C.c_void_p(q_ptr), int(k), int(w), C.c_void_p(block_ptr),
                                                          C.c_void_p(gathered_ptr), int(slot)))

    def comm_wait(self):
        n = C.c_int64(0)
        nat.check(nat.lib().ivfadc_comm_wait(self._h, C.byref(n)))
        return int(n.value)

    def set_list_partition(self, nparts, part):
        """List-partitioned multi-GPU mode (ivfadc_set_list_partition): this handle scans the probed lists l with l % nparts == part."""
        nat.check(nat.lib().ivfadc_set_list_partition(self._h, int(nparts), int(part)))

    def search_device_partial(self, nq, q_ptr, k, w, keys_ptr, counts_ptr):
        nat.check(nat.lib().ivfadc_search_device_partial(self._h, int(nq), C.c_void_p(q_ptr), int(k), int(w), C.c_void_p(keys_ptr), C.c_void_p(counts_ptr)))

    def merge_partials_device(self, nq, k, nparts, keys_all_ptr, counts_all_ptr, ids_ptr, dists_ptr, counts_ptr):
        nat.check(nat.lib().ivfadc_merge_partials_device(self._h, int(nq), int(k), int(nparts), C.c_void_p(keys_all_ptr), C.c_void_p(counts_all_ptr),
                                                         C.c_void_p(ids_ptr), C.c_void_p(dists_ptr), C.c_void_p(counts_ptr)))

    def search_device_listpart(self, nq, q_ptr, k, w, block_ptr, gathered_ptr, ids_ptr, dists_ptr, counts_ptr):
        nat.check(nat.lib().ivfadc_search_device_listpart(self._h, int(nq), C.c_void_p(q_ptr), int(k), int(w), C.c_void_p(block_ptr),
                                                          C.c_void_p(gathered_ptr), C.c_void_p(ids_ptr), C.c_void_p(dists_ptr), C.c_void_p(counts_ptr)))

    def comm_destroy(self):
        nat.check(nat.lib().ivfadc_comm_destroy(self._h))

    IVFADCIndex.comm_destroy = comm_destroy
    IVFADCIndex.set_list_partition = set_list_partition
    IVFADCIndex.search_device_partial = search_device_partial
    IVFADCIndex.merge_partials_device = merge_partials_device
    IVFADCIndex.search_device_listpart = search_device_listpart
    IVFADCIndex.comm_init = comm_init
    IVFADCIndex.search_device_allgather = search_device_allgather
    IVFADCIndex.search_device_allgather_on = search_device_allgather_on
    IVFADCIndex.comm_wait = comm_wait


_comm_methods()


def comm_unique_id():
    """128-byte RCCL id for IVFADCIndex.comm_init (call on rank 0, hand to every rank)."""
    buf = np.zeros(128, np.uint8)
    nat.check(nat.lib().ivfadc_comm_unique_id(nat.ptr(buf, C.c_uint8)))
    return buf


def knn_search(ivfadc, points, k, w=1):
    """knn_search(ivfadc, point, k; w=1) / knn_search(ivfadc, points, k; w=1)  (index.jl:204-273).

    A 1-D `points` is one query and returns (ids[I], dists[float32]), at most k long, ascending.
    A 2-D array or a list of vectors returns (list of ids arrays, list of dists arrays)."""
    assert k >= 1, "Number of neighbors must be k >= 1"                          # index.jl:210
    assert w >= 1, "Number of clusters to search in must be w >= 1"             # index.jl:211
    single = isinstance(points, np.ndarray) and points.ndim == 1
    if not single and not isinstance(points, np.ndarray):
        points = np.stack([np.asarray(p, np.float32) for p in points]) if len(points) else np.zeros((0, ivfadc.d), np.float32)
        single = points.ndim == 1
    pts = np.asarray(points)
    if single:
        pts = pts[None, :]
    assert pts.ndim == 2 and pts.shape[1] == ivfadc.d, "queries must be (nq, %d)" % ivfadc.d
    nq = pts.shape[0]
    # the vectors are packed ONCE, straight into page-locked memory the kernels read, and the results are read out of page-locked
    # memory the final kernel wrote: the reference's host-arrays-in / host-arrays-out contract with no staging copy in the library
    # (the blocks are the index's: one Python thread at a time packs, searches and reads them out -- see _io_lock)
    with ivfadc._io_lock():
        q, ids, dists, counts = ivfadc._io(nq, int(k))
        q[...] = pts
        if nq:
            nat.check(nat.lib().ivfadc_search(ivfadc._h, nq, nat.ptr(q, C.c_float), int(k), int(w), nat.ptr(ids, C.c_uint32),
                                              nat.ptr(dists, C.c_float), nat.ptr(counts, C.c_int32)))
        out_i = [ids[i, :counts[i]].astype(ivfadc.index_type) for i in range(nq)]
        out_d = [dists[i, :counts[i]].copy() for i in range(nq)]
    if single:
        return out_i[0], out_d[0]
    return out_i, out_d


def knn_search_batches(ivfadc, batches, k, w=1):
    """[knn_search(ivfadc, points, k; w) for points in batches] (index.jl:261-273 once per batch) as one native call
    (ivfadc_search_batches): the same results, batch by batch."""
    assert k >= 1, "Number of neighbors must be k >= 1"                          # index.jl:210
    assert w >= 1, "Number of clusters to search in must be w >= 1"             # index.jl:211
    out = []
    for ids, dists, counts in ivfadc.search_batches_raw(batches, k, w):
        out.append(([ids[i, :counts[i]].astype(ivfadc.index_type) for i in range(ids.shape[0])],
                    [dists[i, :counts[i]].copy() for i in range(ids.shape[0])]))
    return out


def _push(ivfadc, point, position):
    """utils.jl:127-145."""
    point = np.asarray(point, np.float32)
    nrows, nvectors = ivfadc.size
    assert point.ndim == 1 and nrows == point.shape[0], "Adding to index requires %d-element vectors" % nrows
    assert _TYPE_TO_BITS[ivfadc.index_type] >= math.log2(nvectors + 1), \
        "Cannot index, exceeding index capacity of %d points" % (2 ** _TYPE_TO_BITS[ivfadc.index_type])
    if position == "first":
        ivfadc._shift_ids(1)                                        # _shift_up_inverse_index! (utils.jl:1-6), on the device
        vecid = 0
    else:
        vecid = nvectors
    ivfadc._append(point[None, :], np.array([vecid], np.uint32))
    return None


def push(ivfadc, point):
    """push!(ivfadc, point) (utils.jl:114)."""
    return _push(ivfadc, point, "last")


def pushfirst(ivfadc, point):
    """pushfirst!(ivfadc, point) (utils.jl:123)."""
    return _push(ivfadc, point, "first")


def _decode_point(ivfadc, codes):
    """utils.jl:71-81."""
    out = np.empty(ivfadc.d, np.float32)
    for i in range(ivfadc.m):
        c = int(np.nonzero(ivfadc._labels[i] == codes[i])[0][0])
        out[i * ivfadc.dsub:(i + 1) * ivfadc.dsub] = ivfadc._codebooks[i, c]
    return out


def _pop(ivfadc, position):
    """utils.jl:41-68."""
    n = len(ivfadc)
    assert n > 0, "Cannot pop element from empty index"
    offsets, codes, ids = ivfadc._lists()
    vecid, shift = (n - 1, 0) if position == "last" else (0, 1)
    pos = int(np.nonzero(ids == vecid)[0][-1])
    cluster = int(np.searchsorted(offsets, pos, side="right") - 1)
    rec = ivfadc._centroids[cluster] + _decode_point(ivfadc, codes[pos])
    # deleteat! + _shift_down_inverse_index!(shift): removing id 0 lowers every other id by one, removing the
    # highest id lowers none -- exactly ivfadc_delete_ids' rule, applied in place on the device
    ivfadc._delete_ids(np.array([vecid], np.uint32))
    return rec


def pop(ivfadc):
    """pop!(ivfadc) (utils.jl:29)."""
    return _pop(ivfadc, "last")


def popfirst(ivfadc):
    """popfirst!(ivfadc) (utils.jl:37)."""
    return _pop(ivfadc, "first")


def delete_from_index(ivfadc, points):
    """delete_from_index!(ivfadc, points) (utils.jl:90-105): `points` are 1-based positions.  In place on the
    device (ivfadc_delete_ids); ids that are not stored are ignored, as in the reference."""
    shifted = np.unique(np.asarray(points, np.int64) - 1)
    shifted = shifted[(shifted >= 0) & (shifted < 2 ** 32)]
    ivfadc._delete_ids(shifted.astype(np.uint32))
    return None
