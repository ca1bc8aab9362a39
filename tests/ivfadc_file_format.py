"""numpy reader / writer of the reference's on-disk format for NaiveQuantizer indexes (/root/reference/src/persistency.jl:1-78
writer, :82-134 loader): TEST infrastructure -- the independent restatement the native reader / writer (ivfadc_load_index /
ivfadc_save_index) are checked against.  Layout: see ivfadc.jl_amd/persistency.py."""
import numpy as np

_I_TYPES = {"UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32, "UInt64": np.uint64}
_T_TYPES = {"Float32": np.float32, "Float64": np.float64}


def write_ivfadc_file(filename, ivfadc):
    """The same file from Python (numpy): cross-check of the native writer, no part of the product path."""
    offsets, codes, ids = ivfadc._lists()
    d, kc, m, k, dsub = ivfadc.d, ivfadc.kc, ivfadc.m, ivfadc.ksub, ivfadc.dsub
    iname = {1: "UInt8", 2: "UInt16", 4: "UInt32"}[ivfadc.index_type.itemsize]
    with open(filename, "wb") as f:
        hdr = ["%d %d" % (d, kc), "%d %d %d %d" % (len(ivfadc), m, k, dsub), "NaiveQuantizer",
               "QuantizedArrays.OrthogonalQuantization", "UInt8", iname, "Distances.SqEuclidean",
               "Distances.SqEuclidean", "Float32"]
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(np.ascontiguousarray(ivfadc._centroids, "<f4").tobytes())
        for i in range(m):
            f.write(np.ascontiguousarray(ivfadc._labels[i], np.uint8).tobytes())
            # vectors[j, :] for j in 1:d  == the transpose of our (k, dsub) block, row-major
            f.write(np.ascontiguousarray(ivfadc._codebooks[i].T, "<f4").tobytes())
        f.write(np.eye(d, dtype="<f4").tobytes())
        for l in range(kc):
            lo, hi = int(offsets[l]), int(offsets[l + 1])
            f.write(np.int64(hi - lo).tobytes())
            f.write(np.ascontiguousarray(ids[lo:hi].astype(ivfadc.index_type)).tobytes())
            f.write(np.ascontiguousarray(codes[lo:hi], np.uint8).tobytes())


def read_ivfadc_file(filename, quantizers_only=False):
    """Parse the file into flat arrays (no GPU needed).  quantizers_only: stop after the codebooks."""
    with open(filename, "rb") as f:
        lines = [f.readline().decode().strip() for _ in range(9)]
        nrows, nclusters = (int(x) for x in lines[0].split())
        n, m, k, dsub = (int(x) for x in lines[1].split())
        if lines[2].split(".")[-1] != "NaiveQuantizer":
            raise NotImplementedError("only NaiveQuantizer files are supported, got %r" % lines[2])
        # the same gate as the native loader: what the HIP path cannot search with the reference's semantics is rejected
        # (type names are written as string(T): `X` or `Module.X`, persistency.jl:14-19, 137-144)
        if lines[3].split(".")[-1] != "OrthogonalQuantization":
            raise NotImplementedError("quantization %r is not supported (only OrthogonalQuantization, i.e. :pq)" % lines[3])
        for what, ln in (("coarse", lines[6]), ("residual", lines[7])):
            if ln.split(".")[-1] != "SqEuclidean":
                raise NotImplementedError("%s distance %r is not supported (only SqEuclidean)" % (what, ln))
        U, I, T = lines[4], lines[5], lines[8]
        if U != "UInt8":
            raise NotImplementedError("quantization element type %s (only UInt8)" % U)
        if I not in _I_TYPES or T not in _T_TYPES:
            raise NotImplementedError("index type %s / element type %s" % (I, T))
        tdt = np.dtype(_T_TYPES[T]).newbyteorder("<")
        idt = np.dtype(_I_TYPES[I]).newbyteorder("<")
        cent = np.frombuffer(f.read(tdt.itemsize * nrows * nclusters), tdt).reshape(nclusters, nrows)
        labels = np.zeros((m, k), np.uint8)
        cbs = np.zeros((m, k, dsub), np.float32)
        for i in range(m):
            labels[i] = np.frombuffer(f.read(k), np.uint8)
            cbs[i] = np.frombuffer(f.read(tdt.itemsize * k * dsub), tdt).reshape(dsub, k).T
        if quantizers_only:
            return dict(centroids=cent.astype(np.float32), codebooks=cbs, labels=labels, index_type=np.dtype(_I_TYPES[I]), T=T, n=n)
        rot = np.frombuffer(f.read(tdt.itemsize * nrows * nrows), tdt).reshape(nrows, nrows)
        # (rot[i] = column i of the quantizer's rotation, persistency.jl:62-64: never read by knn_search; returned for :opq files)
        offsets = np.zeros(nclusters + 1, np.int64)
        ids_l, codes_l = [], []
        for l in range(nclusters):
            clsize = int(np.frombuffer(f.read(8), "<i8")[0])
            ids_l.append(np.frombuffer(f.read(idt.itemsize * clsize), idt))
            codes_l.append(np.frombuffer(f.read(m * clsize), np.uint8).reshape(clsize, m))
            offsets[l + 1] = offsets[l] + clsize
        ids = np.concatenate(ids_l) if ids_l else np.zeros(0, idt)
        codes = np.concatenate(codes_l) if codes_l else np.zeros((0, m), np.uint8)
    return dict(rot=rot.astype(np.float32), centroids=cent.astype(np.float32), codebooks=cbs, labels=labels, offsets=offsets,
                codes=codes, ids=ids.astype(np.uint32), index_type=np.dtype(_I_TYPES[I]), T=T, n=n)
