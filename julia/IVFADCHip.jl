# IVFADCHip.jl -- GPU methods for IVFADC.jl's knn_search hot path (libivfadc_hip.so, MI355X / gfx950).
#
# Drop-in: `include("IVFADCHip.jl")` after `using IVFADC` (or add the body as src/hip.jl and include it last in
# src/IVFADC.jl).  It adds MORE SPECIFIC methods of the reference's own generic functions -- knn_search
# (src/index.jl:204-273), push! (src/utils.jl:114-145) -- for the element types the HIP library implements
# (U = UInt8, T = Float32, SqEuclidean for both distances, NaiveQuantizer); every other index keeps the CPU methods.
# Every C symbol is declared in include/ivfadc_hip.h, which cites the reference interface it replaces.
#
# The same text is shown in INTEGRATION.md section 3 (tests/test_abi.py checks that the two stay identical and that every
# symbol ccall'ed here is exported by the library).  julia is not part of the build image, so this file is exercised
# only through those checks; the calls themselves are the ones the Python/ctypes tests drive.
module IVFADCHip

using IVFADC
using IVFADC: IVFADCIndex, NaiveQuantizer
import IVFADC: knn_search
import Base: push!
using Distances
using QuantizedArrays
import LinearAlgebra

export hip_sync!, hip_release!

# GPU methods for the knn_search hot path; everything else falls through to the CPU methods.
const LIBIVFADC = get(ENV, "IVFADC_HIP_LIB", "libivfadc_hip.so")

mutable struct HipHandle
    ptr::Ptr{Cvoid}
end

function _check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:ivfadc_last_error, LIBIVFADC), Cstring, ()))
    rc == 1 ? throw(AssertionError(msg)) : error("ivfadc_hip status $rc: $msg")
end

const GpuIndex = IVFADCIndex{UInt8,I,Distances.SqEuclidean,Distances.SqEuclidean,Float32,
                             NaiveQuantizer{Distances.SqEuclidean,Float32}} where {I<:Unsigned}

# weak keys: an index that becomes garbage takes its entry -- and, through the handle's finalizer, its device copy -- with it
const _handles = WeakKeyDict{Any,HipHandle}()

# The residual quantizer of a :pq index carries an identity rotation; :opq carries a real one (QuantizedArrays), which neither
# ivfadc_append's encoder nor the device tables apply: such an index keeps the CPU methods (the native loaders refuse it too).
_gpu_ok(ivfadc::GpuIndex) = ivfadc.residual_quantizer.rot == LinearAlgebra.I

"Free the device copy of `ivfadc` now (it is also freed when the index is collected)."
function hip_release!(ivfadc::GpuIndex)
    h = pop!(_handles, ivfadc, nothing)
    h === nothing || finalize(h)
    return nothing
end

"Upload (or refresh) the device copy of `ivfadc`; call again after pop!/delete_from_index!."
function hip_sync!(ivfadc::GpuIndex; device::Int=0)
    _gpu_ok(ivfadc) || error("IVFADCHip: the residual quantizer carries a rotation (:opq); this index is served by the CPU methods")
    cq, rq = ivfadc.coarse_quantizer, ivfadc.residual_quantizer
    d, kc = size(cq.vectors)
    m = length(rq.codebooks); ksub = length(rq.codebooks[1].codes)
    h = get!(_handles, ivfadc) do
        cbs = reduce(hcat, [vec(cb.vectors) for cb in rq.codebooks])        # m blocks of dsub×ksub, column-major
        labels = reduce(vcat, [cb.codes for cb in rq.codebooks])            # m×ksub
        out = Ref{Ptr{Cvoid}}(C_NULL)
        _check(ccall((:ivfadc_create, LIBIVFADC), Cint,
                     (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{UInt8}),
                     out, device, d, kc, m, ksub, cq.vectors, cbs, labels))
        hh = HipHandle(out[])
        finalizer(hh) do x
            x.ptr == C_NULL || ccall((:ivfadc_destroy, LIBIVFADC), Cvoid, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        hh
    end
    offsets = Int64[0; cumsum(length(l.idxs) for l in ivfadc.inverse_index)]
    codes = isempty(ivfadc.inverse_index) ? UInt8[] :
            reduce(vcat, (reduce(vcat, l.codes; init=UInt8[]) for l in ivfadc.inverse_index))   # n×m, list order
    ids = UInt32.(reduce(vcat, (l.idxs for l in ivfadc.inverse_index)))                           # 0-based (index.jl:189)
    _check(ccall((:ivfadc_set_lists, LIBIVFADC), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{UInt32}),
                 h.ptr, offsets, codes, ids))
    return h
end

# knn_search, batch (index.jl:261-273).  Asserts are raised BEFORE the ccall, as in index.jl:210-211.
function knn_search(ivfadc::GpuIndex{I}, points::Vector{Vector{Float32}}, k::Int; w::Int=1) where {I}
    _gpu_ok(ivfadc) || return invoke(knn_search, Tuple{IVFADCIndex,Vector{Vector{Float32}},Int}, ivfadc, points, k; w=w)   # :opq -> CPU
    @assert k >= 1 "Number of neighbors must be k >= 1"
    @assert w >= 1 "Number of clusters to search in must be w >= 1"
    h = get(() -> hip_sync!(ivfadc), _handles, ivfadc)
    nq = length(points)
    q = reduce(hcat, points)                                   # d×nq column-major
    ids = Matrix{UInt32}(undef, k, nq); dists = Matrix{Float32}(undef, k, nq); counts = Vector{Int32}(undef, nq)
    _check(ccall((:ivfadc_search, LIBIVFADC), Cint,
                 (Ptr{Cvoid}, Int64, Ptr{Float32}, Cint, Cint, Ptr{UInt32}, Ptr{Float32}, Ptr{Int32}),
                 h.ptr, nq, q, k, min(w, size(ivfadc.coarse_quantizer, 2)), ids, dists, counts))
    return [I.(ids[1:counts[i], i]) for i in 1:nq], [dists[1:counts[i], i] for i in 1:nq]
end

knn_search(ivfadc::GpuIndex, point::Vector{Float32}, k::Int; w::Int=1) =
    first.(knn_search(ivfadc, [point], k; w=w))

# push! (utils.jl:114-145): encode on the GPU, keep the Julia lists as the source of truth.
function push!(ivfadc::GpuIndex{I}, point::Vector{Float32}) where {I}
    _gpu_ok(ivfadc) || return invoke(push!, Tuple{IVFADCIndex,Vector{Float32}}, ivfadc, point)   # :opq: the CPU encoder applies the rotation
    nrows, nvectors = size(ivfadc)
    @assert nrows == length(point) "Adding to index requires $nrows-element vectors"
    @assert QuantizedArrays.TYPE_TO_BITS[I] >= log2(nvectors + 1) "Cannot index, exceeding index capacity"
    h = get(() -> hip_sync!(ivfadc), _handles, ivfadc)
    m = length(ivfadc.residual_quantizer.codebooks)
    lst = Ref{Int32}(0); code = Vector{UInt8}(undef, m); id = UInt32[nvectors]
    _check(ccall((:ivfadc_append, LIBIVFADC), Cint,
                 (Ptr{Cvoid}, Int64, Ptr{Float32}, Ptr{UInt32}, Ref{Int32}, Ptr{UInt8}),
                 h.ptr, 1, point, id, lst, code))
    push!(ivfadc.inverse_index[lst[] + 1].idxs, I(nvectors))
    push!(ivfadc.inverse_index[lst[] + 1].codes, code)
    return nothing
end

end # module
