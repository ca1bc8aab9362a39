"""Index training through the native trainer (ivfadc_train: k-means++ + Lloyd on the GPU).

Counterpart of the training half of the reference constructor
(/root/reference/src/index.jl:127-147: Clustering.kmeans(init=:kmpp) and
QuantizedArrays.build_quantizer(method=:pq)).  Both are third-party and unseeded in
the reference, so only statistical equivalence is possible.  The encoding of the data
(list assignment + PQ codes) is NOT done here: it goes through the HIP push!/encode path
(ivfadc_append).  There is no CPU trainer in the product (a torch Lloyd used by test
fixtures lives in tests/torch_kmeans.py).
"""
import ctypes as C

import numpy as np

from . import _native as nat


def train_ivfadc_hip(data, kc, k, m, coarse_maxiter=25, quantization_maxiter=25, seed=0, device=0):
    """The native trainer (ivfadc_train: k-means++ + Lloyd on the GPU, deterministic per seed).
    data (n, d) float32 -> centroids (kc, d), codebooks (m, k, dsub), labels (m, k) uint8."""
    x = np.ascontiguousarray(data, np.float32)
    n, d = x.shape
    cent = np.zeros((kc, d), np.float32)
    cbs = np.zeros((m, k, d // max(m, 1)), np.float32)
    nat.check(nat.lib().ivfadc_train(int(device), d, n, nat.ptr(x, C.c_float), int(kc), int(k), int(m), int(coarse_maxiter),
                                     int(quantization_maxiter), C.c_uint64(int(seed)), nat.ptr(cent, C.c_float),
                                     nat.ptr(cbs, C.c_float)))
    labels = np.tile(np.arange(k, dtype=np.uint8), (m, 1))
    return cent, cbs, labels
