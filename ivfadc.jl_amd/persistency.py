"""Reader / writer of the reference's on-disk format for NaiveQuantizer indexes
(/root/reference/src/persistency.jl:1-78 writer, :82-134 loader).

Layout: 9 text lines
    "<nrows> <nclusters>" / "<n> <m> <k> <d>" / coarse-quantizer name / quantization type /
    U / I / Dc / Dr / T
then raw little-endian binary: centroids column by column (:44-49); per codebook its `codes`
then `vectors` ROW j across all k codewords (:56-61); the nrows x nrows rotation matrix
(:62-64, never read by knn_search); per list clsize::Int64, idxs, then each vector's m codes
(:68-78).  Only T=Float32, U=UInt8 files can be searched by the HIP path.  Both directions are native (C++, ivfadc_hip.hip); the
numpy restatement they are tested against lives in tests/ivfadc_file_format.py.
"""


def save_ivfadc_index(filename, ivfadc):
    """save_ivfadc_index(filename, ivfadc) (persistency.jl:1-78) through the native writer (ivfadc_save_index)."""
    from . import _native as nat
    nat.check(nat.lib().ivfadc_save_index(ivfadc._h, str(filename).encode(), int(ivfadc.index_type.itemsize) * 8))


def load_ivfadc_index(filename, device=0):
    """load_ivfadc_index(filename) (persistency.jl:82-134) through the native reader (ivfadc_load_index)."""
    from .index import IVFADCIndex
    return IVFADCIndex.from_file(filename, device=device)
