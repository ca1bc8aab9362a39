# IVFADCHip.jl -- GPU methods for IVFADC.jl's knn_search hot path (libivfadc_hip.so, MI355X / gfx950).
#
# Drop-in: `include("IVFADCHip.jl")` after `using IVFADC` (or add the body as src/hip.jl and include it last in
# src/IVFADC.jl).  It adds MORE SPECIFIC methods of the reference's own generic functions -- knn_search
# (src/index.jl:204-273), push! / pushfirst! (src/utils.jl:114-145), pop! / popfirst! (src/utils.jl:29-68),
# delete_from_index! (src/utils.jl:90-105) -- for the element types the HIP library implements
# (U = UInt8, T = Float32, SqEuclidean for both distances, NaiveQuantizer); every other index keeps the CPU methods.
# Every C symbol is declared in include/ivfadc_hip.h, which cites the reference interface it replaces.
#
# The same text is shown in INTEGRATION.md section 3 (tests/test_abi.py checks that the two stay identical and that every
# symbol ccall'ed here is exported by the library).  julia is not part of the build image, so this file is exercised
# only through those checks; the calls themselves are the ones the Python/ctypes tests drive.
module IVFADCHip

using IVFADC
using IVFADC: IVFADCIndex, NaiveQuantizer
import IVFADC: knn_search, delete_from_index!
import Base: push!, pushfirst!, pop!, popfirst!
using Distances
using QuantizedArrays
import LinearAlgebra

export hip_sync!, hip_release!

# GPU methods for the knn_search hot path; everything else falls through to the CPU methods.
const LIBIVFADC = get(ENV, "IVFADC_HIP_LIB", "libivfadc_hip.so")

# The C ABI this file was written for (include/ivfadc_hip.h: IVFADC_ABI_VERSION).  Checked once, before the first handle is made:
# a library with other prototypes must not be called through these ccall signatures.
const ABI_VERSION = 4
const _abi_checked = Ref(false)
function _check_abi()
    _abi_checked[] && return
    v = ccall((:ivfadc_abi_version, LIBIVFADC), Cint, ())
    v == ABI_VERSION || error("IVFADCHip: $LIBIVFADC has ABI version $v, this shim was written for $ABI_VERSION")
    _abi_checked[] = true
    return
end

# Page-locked pack buffers of a handle (ivfadc_host_alloc).  knn_search takes a Vector of Vectors (index.jl:261-265), so the queries
# must be packed into one d×nq matrix anyway: they are packed straight into memory the GPU reads, and the results are unpacked straight
# out of memory the final kernel wrote -- the library then stages nothing (include/ivfadc_hip.h, "Page-locked host memory").
mutable struct PinnedBlock
    ptr::Ptr{Cvoid}
    bytes::Int
end
PinnedBlock() = PinnedBlock(C_NULL, 0)

function _ensure!(b::PinnedBlock, bytes::Int)
    bytes <= b.bytes && return b.ptr
    b.ptr == C_NULL || _check(ccall((:ivfadc_host_free, LIBIVFADC), Cint, (Ptr{Cvoid},), b.ptr))
    b.ptr = C_NULL; b.bytes = 0
    want = bytes + bytes ÷ 2
    out = Ref{Ptr{Cvoid}}(C_NULL)
    _check(ccall((:ivfadc_host_alloc, LIBIVFADC), Cint, (Csize_t, Ref{Ptr{Cvoid}}), want, out))
    b.ptr = out[]; b.bytes = want
    return b.ptr
end

function _release!(b::PinnedBlock)
    b.ptr == C_NULL || ccall((:ivfadc_host_free, LIBIVFADC), Cint, (Ptr{Cvoid},), b.ptr)
    b.ptr = C_NULL; b.bytes = 0
    return
end

mutable struct HipHandle
    ptr::Ptr{Cvoid}
    q::PinnedBlock        # d×nq Float32
    ids::PinnedBlock      # k×nq UInt32
    dists::PinnedBlock    # k×nq Float32
    counts::PinnedBlock   # nq Int32
end
HipHandle(ptr::Ptr{Cvoid}) = HipHandle(ptr, PinnedBlock(), PinnedBlock(), PinnedBlock(), PinnedBlock())

# views of a handle's pack buffers for a call on nq queries (valid until the next call grows them)
function _io(h::HipHandle, d::Int, k::Int, nq::Int)
    q = unsafe_wrap(Array, Ptr{Float32}(_ensure!(h.q, 4 * d * max(nq, 1))), (d, nq))
    ids = unsafe_wrap(Array, Ptr{UInt32}(_ensure!(h.ids, 4 * k * max(nq, 1))), (k, nq))
    dists = unsafe_wrap(Array, Ptr{Float32}(_ensure!(h.dists, 4 * k * max(nq, 1))), (k, nq))
    counts = unsafe_wrap(Array, Ptr{Int32}(_ensure!(h.counts, 4 * max(nq, 1))), (nq,))
    return q, ids, dists, counts
end

function _check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:ivfadc_last_error, LIBIVFADC), Cstring, ()))
    rc == 1 ? throw(AssertionError(msg)) : error("ivfadc_hip status $rc: $msg")
end

const GpuIndex = IVFADCIndex{UInt8,I,Distances.SqEuclidean,Distances.SqEuclidean,Float32,
                             NaiveQuantizer{Distances.SqEuclidean,Float32}} where {I<:Unsigned}

# IVFADCIndex is an immutable struct (src/index.jl:39): it can carry no finalizer, and a WeakKeyDict would compare its mutable fields by
# CONTENT.  Its inverse_index field is a Vector -- mutable, identity-stable, and exactly as long-lived as the index that holds it -- so
# the registry keys on that object's identity (objectid) and a finalizer on the vector drops the entry: when an index becomes garbage
# its lists do, the entry goes, and the HipHandle's finalizer frees the device replica and the pack buffers.  Code that builds and
# rebuilds indexes therefore leaks nothing; hip_release!(ivfadc) frees a replica at once.
const _handles = Dict{UInt,HipHandle}()
# A finalizer may run at any allocation -- in the middle of get! / pop! on _handles inside hip_sync! / hip_release! -- so it must not
# touch the Dict itself: it only QUEUES the key (a lock-free push onto a vector guarded by a SpinLock that no allocating code holds),
# and the queue is drained, under the registry's lock, at the next hip_sync! / hip_release! (the pattern of the Julia manual's
# "finalizers and locks").  _handles is only ever touched with _registry_lock held.
const _registry_lock = ReentrantLock()
const _dead_lock = Base.Threads.SpinLock()
const _dead_keys = UInt[]
_key(ivfadc::GpuIndex) = objectid(ivfadc.inverse_index)
function _drop_handle(lists)
    k = objectid(lists)
    lock(_dead_lock)
    try
        push!(_dead_keys, k)
    finally
        unlock(_dead_lock)
    end
    return nothing
end
# with _registry_lock held
function _drain_dead!()
    ks = UInt[]
    lock(_dead_lock)
    try
        append!(ks, _dead_keys); empty!(_dead_keys)
    finally
        unlock(_dead_lock)
    end
    for k in ks
        h = pop!(_handles, k, nothing)
        h === nothing || finalize(h)
    end
    return nothing
end

# The residual quantizer of a :pq index carries an identity rotation; :opq carries a real one (QuantizedArrays).  knn_search never reads
# it (src/index.jl:204-258): an :opq index is SEARCHED on the GPU like any other.  quantize_data -- push! / pushfirst! -- does: those
# keep the reference's CPU methods on such an index and drop the device copy, so that the next search uploads the edited lists.
_gpu_ok(ivfadc::GpuIndex) = ivfadc.residual_quantizer.rot == LinearAlgebra.I

"Free the device copy of `ivfadc` now; the next GPU call uploads the Julia lists afresh."
function hip_release!(ivfadc::GpuIndex)
    lock(_registry_lock) do
        _drain_dead!()
        h = pop!(_handles, _key(ivfadc), nothing)
        h === nothing || finalize(h)
    end
    return nothing
end

"Upload (or refresh) the device copy of `ivfadc` from the Julia lists."
function hip_sync!(ivfadc::GpuIndex; device::Int=0)
    cq, rq = ivfadc.coarse_quantizer, ivfadc.residual_quantizer
    d, kc = size(cq.vectors)
    m = length(rq.codebooks); ksub = length(rq.codebooks[1].codes)
    _check_abi()
    h = lock(_registry_lock) do
      _drain_dead!()
      get!(_handles, _key(ivfadc)) do
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
            _release!(x.q); _release!(x.ids); _release!(x.dists); _release!(x.counts)
        end
        finalizer(_drop_handle, ivfadc.inverse_index)
        hh
      end
    end
    offsets = Int64[0; cumsum(length(l.idxs) for l in ivfadc.inverse_index)]
    codes = isempty(ivfadc.inverse_index) ? UInt8[] :
            reduce(vcat, (reduce(vcat, l.codes; init=UInt8[]) for l in ivfadc.inverse_index))   # n×m, list order
    ids = UInt32.(reduce(vcat, (l.idxs for l in ivfadc.inverse_index)))                           # 0-based (index.jl:189)
    _check(ccall((:ivfadc_set_lists, LIBIVFADC), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{UInt32}),
                 h.ptr, offsets, codes, ids))
    return h
end

function _handle(ivfadc::GpuIndex)
    h = lock(_registry_lock) do
        _drain_dead!()
        get(_handles, _key(ivfadc), nothing)
    end
    return h === nothing ? hip_sync!(ivfadc) : h
end

# Every mutator edits the device copy IN PLACE next to the Julia lists.  Should a device edit fail half way, the handle is dropped, so
# that the next search uploads the Julia lists (the source of truth) afresh: a stale device copy cannot be searched.
function _on_device(f, ivfadc::GpuIndex)
    try
        return f()
    catch
        hip_release!(ivfadc)
        rethrow()
    end
end

# knn_search, batch (index.jl:261-273).  Asserts are raised BEFORE the ccall, as in index.jl:210-211.
function knn_search(ivfadc::GpuIndex{I}, points::Vector{Vector{Float32}}, k::Int; w::Int=1) where {I}
    @assert k >= 1 "Number of neighbors must be k >= 1"
    @assert w >= 1 "Number of clusters to search in must be w >= 1"
    h = _handle(ivfadc)
    nq = length(points)
    nq == 0 && return Vector{I}[], Vector{Float32}[]
    d = size(ivfadc.coarse_quantizer, 1)
    q, ids, dists, counts = _io(h, d, k, nq)
    for i in 1:nq
        @assert length(points[i]) == d "Searching requires $d-element vectors"
        copyto!(q, (i - 1) * d + 1, points[i], 1, d)           # d×nq column-major, packed once, into page-locked memory
    end
    _check(ccall((:ivfadc_search, LIBIVFADC), Cint,
                 (Ptr{Cvoid}, Int64, Ptr{Float32}, Cint, Cint, Ptr{UInt32}, Ptr{Float32}, Ptr{Int32}),
                 h.ptr, nq, q, k, min(w, size(ivfadc.coarse_quantizer, 2)), ids, dists, counts))
    return [I.(ids[1:counts[i], i]) for i in 1:nq], [dists[1:counts[i], i] for i in 1:nq]
end

knn_search(ivfadc::GpuIndex, point::Vector{Float32}, k::Int; w::Int=1) =
    first.(knn_search(ivfadc, [point], k; w=w))

# A run of consecutive batches: `[knn_search(ivfadc, b, k; w=w) for b in batches]` as ONE native call (ivfadc_search_batches).  Inside,
# every batch is searched with its successor already named (the serving-loop hint, ivfadc_set_next_queries, on buffers the library
# owns), so the successor's coarse search runs behind the batch's scan launch.  Same results, batch by batch.
function knn_search(ivfadc::GpuIndex{I}, batches::Vector{Vector{Vector{Float32}}}, k::Int; w::Int=1) where {I}
    @assert k >= 1 "Number of neighbors must be k >= 1"
    @assert w >= 1 "Number of clusters to search in must be w >= 1"
    h = _handle(ivfadc)
    sizes = Int64[length(b) for b in batches]
    total = sum(sizes)
    total == 0 && return [(Vector{I}[], Vector{Float32}[]) for _ in batches]
    d = size(ivfadc.coarse_quantizer, 1)
    q, ids, dists, counts = _io(h, d, k, total)
    at = 0
    for b in batches, p in b                                                 # d×total, the batches back to back
        @assert length(p) == d "Searching requires $d-element vectors"
        copyto!(q, at * d + 1, p, 1, d)
        at += 1
    end
    _check(ccall((:ivfadc_search_batches, LIBIVFADC), Cint,
                 (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float32}, Cint, Cint, Ptr{UInt32}, Ptr{Float32}, Ptr{Int32}),
                 h.ptr, length(sizes), sizes, q, k, min(w, size(ivfadc.coarse_quantizer, 2)), ids, dists, counts))
    ends = cumsum(sizes)
    return [([I.(ids[1:counts[i], i]) for i in (e - s + 1):e], [dists[1:counts[i], i] for i in (e - s + 1):e])
            for (s, e) in zip(sizes, ends)]
end

# push! / pushfirst! (utils.jl:114-145): encode on the GPU and append in place on the device; the Julia lists get the same edit.
function _gpu_push!(ivfadc::GpuIndex{I}, point::Vector{Float32}, position::Symbol) where {I}
    nrows, nvectors = size(ivfadc)
    @assert nrows == length(point) "Adding to index requires $nrows-element vectors"
    @assert QuantizedArrays.TYPE_TO_BITS[I] >= log2(nvectors + 1) "Cannot index, exceeding index capacity"
    h = _handle(ivfadc)
    m = length(ivfadc.residual_quantizer.codebooks)
    (vecid, shift) = position == :first ? (0, 1) : (nvectors, 0)            # utils.jl:139
    lst = Ref{Int32}(0); code = Vector{UInt8}(undef, m); id = UInt32[vecid]
    _on_device(ivfadc) do
        shift == 0 || _check(ccall((:ivfadc_shift_ids, LIBIVFADC), Cint, (Ptr{Cvoid}, Int32), h.ptr, shift))   # _shift_up_inverse_index!
        _check(ccall((:ivfadc_append, LIBIVFADC), Cint,
                     (Ptr{Cvoid}, Int64, Ptr{Float32}, Ptr{UInt32}, Ref{Int32}, Ptr{UInt8}),
                     h.ptr, 1, point, id, lst, code))
    end
    if shift != 0
        for l in ivfadc.inverse_index
            l.idxs .+= one(I)                                                # utils.jl:2-6
        end
    end
    push!(ivfadc.inverse_index[lst[] + 1].idxs, I(vecid))
    push!(ivfadc.inverse_index[lst[] + 1].codes, code)
    return nothing
end

function push!(ivfadc::GpuIndex, point::Vector{Float32})
    if !_gpu_ok(ivfadc)     # :opq: the CPU encoder applies the rotation; the device copy is dropped and re-uploaded by the next search
        r = invoke(push!, Tuple{IVFADCIndex,Vector{Float32}}, ivfadc, point)
        hip_release!(ivfadc)
        return r
    end
    return _gpu_push!(ivfadc, point, :last)
end

function pushfirst!(ivfadc::GpuIndex, point::Vector{Float32})
    if !_gpu_ok(ivfadc)
        r = invoke(pushfirst!, Tuple{IVFADCIndex,Vector{Float32}}, ivfadc, point)
        hip_release!(ivfadc)
        return r
    end
    return _gpu_push!(ivfadc, point, :first)
end

# pop! / popfirst! / delete_from_index! (utils.jl:29-105): the reference's own method edits the Julia lists (and makes its assertions
# and the reconstruction); ivfadc_delete_ids then removes the same 0-based ids from the device copy in place -- stable within every
# list, every surviving id lowered by the number of removed ids below it (_shift_inverse_index!, utils.jl:11-27).
function _gpu_delete!(ivfadc::GpuIndex, ids::Vector{UInt32})
    h = lock(_registry_lock) do
        _drain_dead!()
        get(_handles, _key(ivfadc), nothing)
    end
    h === nothing && return nothing                  # no device copy yet: the next search uploads the edited lists
    _on_device(ivfadc) do
        _check(ccall((:ivfadc_delete_ids, LIBIVFADC), Cint, (Ptr{Cvoid}, Int64, Ptr{UInt32}, Ptr{Int64}),
                     h.ptr, length(ids), ids, C_NULL))
    end
    return nothing
end

function pop!(ivfadc::GpuIndex)
    n = length(ivfadc)
    rec = invoke(pop!, Tuple{IVFADCIndex}, ivfadc)                  # asserts n > 0; removes id n - 1, shifts nothing
    _gpu_delete!(ivfadc, UInt32[n - 1])
    return rec
end

function popfirst!(ivfadc::GpuIndex)
    rec = invoke(popfirst!, Tuple{IVFADCIndex}, ivfadc)             # removes id 0, lowers every other id by one
    _gpu_delete!(ivfadc, UInt32[0])
    return rec
end

function delete_from_index!(ivfadc::GpuIndex, points::Vector{<:Integer})
    invoke(delete_from_index!, Tuple{IVFADCIndex,Vector{<:Integer}}, ivfadc, points)   # `points` are 1-based (utils.jl:93)
    _gpu_delete!(ivfadc, UInt32.(unique(points) .- 1))
    return nothing
end

end # module
