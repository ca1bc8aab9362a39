# make_fixture.jl -- cross-implementation fixtures from the REAL IVFADC.jl (the only route to a pinned oracle).
#
# Run where Julia and IVFADC.jl v0.1.4 are installed (not possible in the build image: no julia, no registry):
#     julia --project=/path/to/IVFADC.jl tools/julia/make_fixture.jl tests/golden
# It writes, for two seeded indexes,
#     julia_<name>.bin          save_ivfadc_index(...)            (src/persistency.jl:1-78; read natively by ivfadc_load_index)
#     julia_<name>_queries.f32  the queries, d x nq Float32 column-major
#     julia_<name>_knn.txt      one line per (w, query): "w q count id... | dist..."  from knn_search (src/index.jl:204-273)
# tests/test_julia_fixture.py consumes them WHEN PRESENT: the index is loaded by the native reader and every id must match
# exactly, every Float32 distance within 1e-4 relative.  The k-means inside IVFADCIndex(...) is unseeded in the reference
# (Clustering.kmeans), so the script seeds the global RNG; whatever index comes out is saved and is the fixture.
using Random
using IVFADC

outdir = length(ARGS) >= 1 ? ARGS[1] : "tests/golden"
mkpath(outdir)

function dump(name, ivfadc, queries::Vector{Vector{Float32}}, K, ws)
    save_ivfadc_index(joinpath(outdir, "julia_$(name).bin"), ivfadc)
    open(joinpath(outdir, "julia_$(name)_queries.f32"), "w") do io
        for q in queries
            write(io, q)
        end
    end
    open(joinpath(outdir, "julia_$(name)_knn.txt"), "w") do io
        println(io, "# K=$(K) d=$(length(queries[1])) nq=$(length(queries))  (ids 0-based as stored, index.jl:189)")
        for w in ws
            ids, dists = knn_search(ivfadc, queries, K, w=w)
            for (qi, (i, dd)) in enumerate(zip(ids, dists))
                println(io, w, " ", qi - 1, " ", length(i), " ", join(Int.(i), " "), " | ", join(repr.(Float32.(dd)), " "))
            end
        end
    end
end

# (1) the reference's own known-answer data (test/search.jl:27-30), as Float32
Random.seed!(20260101)
data1 = Float32[0    0     0     1    1  1    1    1  20  20    20    20    20;
                0.1  0.11  0.12  8   10  15   14   16  5   5.1   5.2   5.4   5.5]
ix1 = IVFADCIndex(data1, kc=3, k=8, m=2, index_type=UInt16)
dump("search_jl", ix1, [Float32[1, 10], Float32[0, 0], Float32[20, 5]], 3, (1, 2))

# (2) 50 x 1000 (README shape): kc=100, k=256, m=10
Random.seed!(20260102)
data2 = rand(Float32, 50, 1000)
ix2 = IVFADCIndex(data2, kc=100, k=256, m=10, index_type=UInt16)
qs = [data2[:, i] .+ 0.01f0 .* rand(Float32, 50) for i in 1:64]
dump("readme_50x1000", ix2, qs, 3, (1, 8))
println("fixtures written to ", outdir)
