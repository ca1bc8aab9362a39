"""ctypes front-end of the CPU oracle (oracle/ivfadc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Numeric parity with the Julia implementation is unpinned (see the C header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "ivfadc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, u8p, i32p, u32p, i64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_int64))
        L.ora_coarse_search.argtypes = [C.c_int, C.c_int, fp, fp, C.c_int, i32p, fp]
        L.ora_coarse_search.restype = C.c_int
        L.ora_knn_search.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, u8p, i64p, u8p, u32p, C.c_uint64,
                                     C.c_int, fp, C.c_int, C.c_int, u32p, fp, i32p, C.c_int]
        L.ora_knn_search.restype = C.c_int
        L.ora_encode_points.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, u8p, C.c_int, fp, i32p, u8p]
        L.ora_encode_points.restype = C.c_int
        L.ora_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u8p]
        L.ora_synth_fill.restype = None
        L.ora_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class OracleIndex:
    """Flat arrays of an IVFADC index in the layout the C ABI also takes.

    centroids  (kc, d) float32  -- row c is centroid c (== d x kc column-major)
    codebooks  (m, ksub, dsub) float32 -- codebooks[i, c] is codeword c of sub-space i
    labels     (m, ksub) uint8
    offsets    (kc+1,) int64; codes (n, m) uint8 or None (synthetic); ids (n,) uint32 or None
    """

    def __init__(self, centroids, codebooks, labels, offsets, codes=None, ids=None, synth_seed=0):
        self.centroids = np.ascontiguousarray(centroids, np.float32)
        self.codebooks = np.ascontiguousarray(codebooks, np.float32)
        self.labels = np.ascontiguousarray(labels, np.uint8)
        self.offsets = np.ascontiguousarray(offsets, np.int64)
        self.codes = None if codes is None else np.ascontiguousarray(codes, np.uint8)
        self.ids = None if ids is None else np.ascontiguousarray(ids, np.uint32)
        self.synth_seed = int(synth_seed)
        self.kc, self.d = self.centroids.shape
        self.m, self.ksub, self.dsub = self.codebooks.shape
        assert self.m * self.dsub == self.d
        assert self.labels.shape == (self.m, self.ksub)
        assert self.offsets.shape == (self.kc + 1,)

    def knn_search(self, queries, K, w=1, nthreads=1):
        """queries (nq, d) float32 -> ids (nq, K) uint32, dists (nq, K) float32, counts (nq,) int32."""
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        assert q.shape[1] == self.d
        Ka = max(int(K), 1)
        ids = np.zeros((nq, Ka), np.uint32)
        dists = np.full((nq, Ka), np.inf, np.float32)
        counts = np.zeros(nq, np.int32)
        rc = lib().ora_knn_search(self.d, self.kc, self.m, self.ksub,
                                  _p(self.centroids, C.c_float), _p(self.codebooks, C.c_float),
                                  _p(self.labels, C.c_uint8), _p(self.offsets, C.c_int64),
                                  _p(self.codes, C.c_uint8), _p(self.ids, C.c_uint32),
                                  C.c_uint64(self.synth_seed), nq, _p(q, C.c_float), int(K), int(w),
                                  _p(ids, C.c_uint32), _p(dists, C.c_float), _p(counts, C.c_int32), int(nthreads))
        if rc == 1:
            raise AssertionError("oracle: k >= 1 and w >= 1 required (index.jl:210-211)")
        if rc != 0:
            raise MemoryError("oracle: rc=%d" % rc)
        return ids, dists, counts

    def coarse_search(self, point, w):
        p = np.ascontiguousarray(point, np.float32)
        cl = np.zeros(w, np.int32)
        dist = np.zeros(w, np.float32)
        rc = lib().ora_coarse_search(self.d, self.kc, _p(self.centroids, C.c_float), _p(p, C.c_float), int(w),
                                     _p(cl, C.c_int32), _p(dist, C.c_float))
        if rc != 0:
            raise AssertionError("oracle coarse_search rc=%d" % rc)
        return cl, dist

    def encode(self, points):
        """points (npts, d) -> (list (npts,) int32 0-based, codes (npts, m) uint8)."""
        p = np.ascontiguousarray(points, np.float32)
        if p.ndim == 1:
            p = p[None, :]
        n = p.shape[0]
        lst = np.zeros(n, np.int32)
        codes = np.zeros((n, self.m), np.uint8)
        rc = lib().ora_encode_points(self.d, self.kc, self.m, self.ksub, _p(self.centroids, C.c_float),
                                     _p(self.codebooks, C.c_float), _p(self.labels, C.c_uint8), n,
                                     _p(p, C.c_float), _p(lst, C.c_int32), _p(codes, C.c_uint8))
        if rc != 0:
            raise AssertionError("oracle encode rc=%d" % rc)
        return lst, codes


def synth_fill(seed, g0, n, m):
    out = np.zeros((n, m), np.uint8)
    lib().ora_synth_fill(C.c_uint64(seed), C.c_uint64(g0), C.c_uint64(n), int(m), _p(out, C.c_uint8))
    return out


def max_threads():
    return int(lib().ora_max_threads())
