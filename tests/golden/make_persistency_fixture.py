"""Hand-assembled index files in IVFADC.jl's on-disk format, one `write` call of /root/reference/src/persistency.jl:22-78 at a time.

This generator does NOT share code with the readers under test (ivfadc_load_index in ivfadc.jl_amd/csrc/ivfadc_hip.hip, and the numpy
restatement tests/ivfadc_file_format.py): every field is produced by a scalar struct.pack in the order the reference's writer emits
it, from closed-form VALUE FORMULAS in Julia's own indexing (vectors[row, column], 1-based) -- so a field-order or transposition
mistake shared by the two readers shows up as a wrong number, not as agreement.  The expected arrays in tests/test_persistency.py
are written from the same formulas in the C ABI's layout.

    python tests/golden/make_persistency_fixture.py      # rewrites tests/golden/persistency_hand_{f32_u16,f64_u32,f32_u16_opq}.bin
"""
import os
import struct

NROWS, NCLUSTERS, M, K, DSUB = 4, 3, 2, 4, 2          # d = nrows = M * DSUB
LIST_SIZES = (2, 0, 3)                                # an empty list in the middle
N = sum(LIST_SIZES)


def centroid(row, col):          # coarse_quantizer.vectors[row, col], 1-based
    return 100.0 * col + row + 0.5


def codeword(i, j, c):           # residual_quantizer.codebooks[i].vectors[j, c], 1-based
    return 1000.0 * i + 10.0 * c + j + 0.125


def label(i, c):                 # residual_quantizer.codebooks[i].codes[c], 1-based; a permutation for i = 1, identity for i = 2
    return (3, 0, 2, 1)[c - 1] if i == 1 else c - 1


def list_id(i, j):               # inverse_index[i].idxs[j] (0-based ids as the reference stores them, index.jl:189)
    return 40 * i + j


def list_code(i, j, ii):         # inverse_index[i].codes[j][ii]: a byte that is a label of codebook ii
    return label(ii, ((i + j + ii) % K) + 1)


def rotation(row, col):          # residual_quantizer.rot[row, col], 1-based: a plane rotation of rows / columns 1, 3 plus a sign flip of 4
    c, s_ = 0.8, 0.6             # (what an :opq quantizer carries; exactly representable products are not needed: the bytes are copied)
    if (row, col) == (1, 1) or (row, col) == (3, 3):
        return c
    if (row, col) == (1, 3):
        return -s_
    if (row, col) == (3, 1):
        return s_
    if (row, col) == (2, 2):
        return 1.0
    if (row, col) == (4, 4):
        return -1.0
    return 0.0


def build(T, I, rotated=False):
    fmt_t = {"Float32": "<f", "Float64": "<d"}[T]
    fmt_i = {"UInt8": "<B", "UInt16": "<H", "UInt32": "<I"}[I]
    out = bytearray()
    # persistency.jl:22-30 -- nine println lines
    for line in ("%d %d" % (NROWS, NCLUSTERS), "%d %d %d %d" % (N, M, K, DSUB), "NaiveQuantizer", "QuantizedArrays.OrthogonalQuantization",
                 "UInt8", I, "Distances.SqEuclidean", "Distances.SqEuclidean", T):
        out += (line + "\n").encode()
    # :44-49 _write_naive_coarse_quantizer: for i in 1:nclusters  write(fid, quantizer.vectors[:, i])
    for i in range(1, NCLUSTERS + 1):
        for row in range(1, NROWS + 1):
            out += struct.pack(fmt_t, centroid(row, i))
    # :52-65 _write_residual_quantizer
    for i in range(1, M + 1):
        for c in range(1, K + 1):                        # write(fid, quantizer.codebooks[i].codes)
            out += struct.pack("<B", label(i, c))
        for j in range(1, DSUB + 1):                     # for j in 1:d  write(fid, quantizer.codebooks[i].vectors[j, :])
            for c in range(1, K + 1):
                out += struct.pack(fmt_t, codeword(i, j, c))
    for i in range(1, NROWS + 1):                        # for i in 1:nrows  write(fid, quantizer.rot[:, i])  (identity for :pq)
        for row in range(1, NROWS + 1):
            out += struct.pack(fmt_t, rotation(row, i) if rotated else (1.0 if row == i else 0.0))
    # :68-78 _write_inverse_index
    for i in range(1, NCLUSTERS + 1):
        clsize = LIST_SIZES[i - 1]
        out += struct.pack("<q", clsize)                 # write(fid, clsize)   (Int = Int64)
        for j in range(1, clsize + 1):                   # write(fid, inverse_index[i].idxs)
            out += struct.pack(fmt_i, list_id(i, j))
        for j in range(1, clsize + 1):                   # for j in 1:clsize  write(fid, inverse_index[i].codes[j])
            for ii in range(1, M + 1):
                out += struct.pack("<B", list_code(i, j, ii))
    return bytes(out)


FILES = {"persistency_hand_f32_u16.bin": ("Float32", "UInt16", False), "persistency_hand_f64_u32.bin": ("Float64", "UInt32", False),
         "persistency_hand_f32_u16_opq.bin": ("Float32", "UInt16", True)}      # the same index with a non-identity rotation (:opq)

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, (T, I, rotated) in FILES.items():
        with open(os.path.join(here, name), "wb") as f:
            f.write(build(T, I, rotated))
        print(name, len(build(T, I, rotated)), "bytes")
