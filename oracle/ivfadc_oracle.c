/*
 * ivfadc_oracle.c -- CPU restatement of IVFADC.jl's knn_search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ivfadc.jl_amd/ (the product) may
 * include, link, load or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and there only as the checker / the
 * reported CPU baseline.
 *
 * What it restates (reference paths are relative to /root/reference):
 *   src/coarsequantizers.jl:33-37   coarse_search(::NaiveQuantizer, point, w)
 *   src/coarsequantizers.jl:40-45   _closest_cluster_residuals
 *   src/index.jl:204-258            knn_search (single query)
 *   src/index.jl:261-273            knn_search (batch; serial map)
 *   src/utils.jl:148-161            _encode_point (push! path)
 *
 * Pinning status: the reference holds NO numeric golden vectors for this path
 * (its tests draw unseeded rand() data) and Julia plus its five registry
 * dependencies are absent from this image, so NUMERIC PARITY WITH THE JULIA
 * IMPLEMENTATION IS UNPINNED.  What is pinned: the reference's only
 * result-level test, test/search.jl:26-49 (set-level known answers on 2x13
 * hand-made data) -- tests/test_oracle.py replays it through this file -- and
 * the type / assertion conventions of test/search.jl:11-21.
 *
 * Arithmetic that lives in third-party packages whose source is not under
 * /root/reference (Project.toml:13-19; no Manifest, so patch versions are
 * unpinned) is restated from its published behaviour:
 *   Distances.jl ^0.10   colwise(SqEuclidean(), A, b)[j] = sum_i abs2(A[i,j]-b[i])
 *                        (call sites index.jl:234, coarsequantizers.jl:34)
 *   DataStructures.jl ^0.18  SortedMultiDict: ordered by key, equal keys keep
 *                        insertion order, last() is the maximum (index.jl:225-257);
 *                        LittleDict(keys, vals): label -> value (index.jl:235)
 *   QuantizedArrays.jl ^0.1.6  rowrange(n, m, i) = contiguous slice of sub-space
 *                        i when n % m == 0 (index.jl:233); quantize_data = per
 *                        sub-space argmin over codewords (utils.jl:158)
 *
 * Canonical float order (the reference's @simd order is unspecified): Float32,
 * sums run sequentially in ascending index, one rounding per operation, no
 * FMA contraction.  Build with -ffp-contract=off and without -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_OK 0
#define ORA_ERR_ASSERT 1
#define ORA_ERR_NOMEM 2

/* ---- Distances.colwise(SqEuclidean(), A[d x ncols], b[d]) ---------------- */
static void colwise_sqeuclidean(int d, int ncols, const float *A, const float *b, float *out)
{
    for (int j = 0; j < ncols; ++j) {
        const float *col = A + (size_t)j * d;
        float s = 0.0f;
        for (int i = 0; i < d; ++i) {
            float t = col[i] - b[i];
            s = s + t * t;
        }
        out[j] = s;
    }
}

/* ---- sortperm(dists)[1:w]: stable, so ties go to the lower column index -- */
typedef struct { float key; int idx; } ora_pair_t;

static int pair_cmp(const void *pa, const void *pb)
{
    const ora_pair_t *a = (const ora_pair_t *)pa, *b = (const ora_pair_t *)pb;
    if (a->key < b->key) return -1;
    if (a->key > b->key) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* coarsequantizers.jl:33-37.  clusters are returned 0-based here. */
int ora_coarse_search(int d, int kc, const float *centroids, const float *point, int w,
                      int32_t *out_clusters, float *out_dists)
{
    if (w < 1 || w > kc) return ORA_ERR_ASSERT;
    float *dist = (float *)malloc(sizeof(float) * (size_t)kc);
    ora_pair_t *perm = (ora_pair_t *)malloc(sizeof(ora_pair_t) * (size_t)kc);
    if (!dist || !perm) { free(dist); free(perm); return ORA_ERR_NOMEM; }
    colwise_sqeuclidean(d, kc, centroids, point, dist);
    for (int c = 0; c < kc; ++c) { perm[c].key = dist[c]; perm[c].idx = c; }
    qsort(perm, (size_t)kc, sizeof(ora_pair_t), pair_cmp);
    for (int j = 0; j < w; ++j) { out_clusters[j] = perm[j].idx; out_dists[j] = perm[j].key; }
    free(dist); free(perm);
    return ORA_OK;
}

/* coarsequantizers.jl:40-45: residuals[:, j] = point .- vectors[:, cl_j] */
void ora_residuals(int d, const float *centroids, const float *point, int w,
                   const int32_t *clusters, float *out_residuals /* d x w */)
{
    for (int j = 0; j < w; ++j) {
        const float *c = centroids + (size_t)clusters[j] * d;
        for (int i = 0; i < d; ++i) out_residuals[(size_t)j * d + i] = point[i] - c[i];
    }
}

/* ---- SortedMultiDict{T,I} used as a bounded max-heap (index.jl:225,247-254) */
typedef struct { float *keys; uint32_t *vals; int len; } ora_smd_t;

static void smd_push(ora_smd_t *s, float key, uint32_t val)
{
    /* insert AFTER every existing entry whose key is <= key */
    int pos = s->len;
    while (pos > 0 && s->keys[pos - 1] > key) {
        s->keys[pos] = s->keys[pos - 1];
        s->vals[pos] = s->vals[pos - 1];
        --pos;
    }
    s->keys[pos] = key;
    s->vals[pos] = val;
    s->len++;
}

/* Source of the codes of one inverted list: either stored arrays or the
 * counter-based synthetic generator (see ora_synth_code below). */
typedef struct {
    const int64_t *offsets;   /* kc+1, in points */
    const uint8_t *codes;     /* n x m, list order, m bytes per point; NULL => synthetic */
    const uint32_t *ids;      /* n, 0-based; NULL => id == global position */
    uint64_t synth_seed;
} ora_lists_t;

static inline uint64_t ora_mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

/* byte b of the code of the point at global position g (synthetic indexes):
 * linear byte index B = g*m + b; 8 bytes come out of one 64-bit hash word.  */
uint8_t ora_synth_code(uint64_t seed, uint64_t g, int m, int b)
{
    uint64_t B = g * (uint64_t)m + (uint64_t)b;
    uint64_t word = ora_mix64(seed + (B >> 3) * 0x9E3779B97F4A7C15ull);
    return (uint8_t)(word >> (8 * (B & 7)));
}

/* index.jl:204-258, one query.  Returns the number of neighbours found. */
static int knn_single(int d, int kc, int m, int ksub,
                      const float *centroids, const float *codebooks, const uint8_t *labels,
                      const ora_lists_t *L, const float *point, int K, int w,
                      uint32_t *out_ids, float *out_dists, int *out_count)
{
    if (K < 1) return ORA_ERR_ASSERT;                 /* index.jl:210 */
    if (w < 1) return ORA_ERR_ASSERT;                 /* index.jl:211 */
    if (d % m != 0) return ORA_ERR_ASSERT;            /* rowrange only restated for d % m == 0 */
    if (w > kc) w = kc;                               /* index.jl:216 */
    const int dsub = d / m;

    int32_t *cl = (int32_t *)malloc(sizeof(int32_t) * (size_t)w);
    float *cdist = (float *)malloc(sizeof(float) * (size_t)w);
    float *resid = (float *)malloc(sizeof(float) * (size_t)d * (size_t)w);
    float *table = (float *)malloc(sizeof(float) * (size_t)m * 256);
    float *diffs = (float *)malloc(sizeof(float) * (size_t)ksub);
    ora_smd_t nb;
    nb.keys = (float *)malloc(sizeof(float) * ((size_t)K + 1));
    nb.vals = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)K + 1));
    nb.len = 0;
    if (!cl || !cdist || !resid || !table || !diffs || !nb.keys || !nb.vals) {
        free(cl); free(cdist); free(resid); free(table); free(diffs); free(nb.keys); free(nb.vals);
        return ORA_ERR_NOMEM;
    }

    int rc = ora_coarse_search(d, kc, centroids, point, w, cl, cdist);     /* :219 */
    if (rc == ORA_OK) {
        ora_residuals(d, centroids, point, w, cl, resid);                  /* :220 */
        float maxdist = 0.0f;                                              /* :226 */
        for (int j = 0; j < w; ++j) {                                      /* :228 */
            const float dc = cdist[j];                                     /* :229 */
            for (int i = 0; i < m; ++i) {                                  /* :232-236 */
                const float *cb = codebooks + (size_t)i * dsub * ksub;
                colwise_sqeuclidean(dsub, ksub, cb, resid + (size_t)j * d + (size_t)i * dsub, diffs);
                for (int c = 0; c < ksub; ++c)      /* LittleDict(codes, diffs) */
                    table[(size_t)i * 256 + labels[(size_t)i * ksub + c]] = diffs[c];
            }
            const int64_t lo = L->offsets[cl[j]], hi = L->offsets[cl[j] + 1];   /* :240 */
            for (int64_t p = lo; p < hi; ++p) {                            /* :241 */
                float dd = dc;                                             /* :242 */
                for (int ii = 0; ii < m; ++ii) {                           /* :243-246 */
                    uint8_t code = L->codes ? L->codes[(size_t)p * m + ii]
                                            : ora_synth_code(L->synth_seed, (uint64_t)p, m, ii);
                    dd += table[(size_t)ii * 256 + code];
                }
                uint32_t id = L->ids ? L->ids[p] : (uint32_t)p;
                if (nb.len < K) {                                          /* :247-249 */
                    smd_push(&nb, dd, id);
                    maxdist = nb.keys[nb.len - 1];
                } else if (maxdist > dd) {                                 /* :250-253 */
                    nb.len--;                        /* delete!(lastindex) = drop the max */
                    smd_push(&nb, dd, id);
                    maxdist = nb.keys[nb.len - 1];
                }
            }
        }
        for (int i = 0; i < nb.len; ++i) { out_ids[i] = nb.vals[i]; out_dists[i] = nb.keys[i]; }  /* :257 */
        *out_count = nb.len;
    }
    free(cl); free(cdist); free(resid); free(table); free(diffs); free(nb.keys); free(nb.vals);
    return rc;
}

/* index.jl:261-273.  queries: d x nq column-major.  Outputs K slots per query
 * (slots past out_counts[q] are left untouched).  nthreads <= 1 reproduces the
 * reference's serial loop; nthreads > 1 is the Threads.@threads variant the
 * comment at index.jl:269 hints at (used only as a timed CPU baseline).      */
int ora_knn_search(int d, int kc, int m, int ksub,
                   const float *centroids, const float *codebooks, const uint8_t *labels,
                   const int64_t *offsets, const uint8_t *codes, const uint32_t *ids,
                   uint64_t synth_seed,
                   int nq, const float *queries, int K, int w,
                   uint32_t *out_ids, float *out_dists, int32_t *out_counts, int nthreads)
{
    if (K < 1 || w < 1) return ORA_ERR_ASSERT;
    ora_lists_t L = { offsets, codes, ids, synth_seed };
    int rc_all = ORA_OK;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int q = 0; q < nq; ++q) {
        int cnt = 0;
        int rc = knn_single(d, kc, m, ksub, centroids, codebooks, labels, &L,
                            queries + (size_t)q * d, K, w,
                            out_ids + (size_t)q * K, out_dists + (size_t)q * K, &cnt);
        out_counts[q] = cnt;
        if (rc != ORA_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    (void)nthreads;
    return rc_all;
}

/* utils.jl:148-161 _encode_point, for npts points (d x npts column-major).
 * out_list: 0-based cluster; out_codes: m labels per point.
 * quantize_data is third-party: restated as, per sub-space, the codeword with
 * the smallest SqEuclidean distance to the residual slice, first minimum on
 * ties (Julia argmin/findmin convention).  Parity of that tie rule: unpinned. */
int ora_encode_points(int d, int kc, int m, int ksub,
                      const float *centroids, const float *codebooks, const uint8_t *labels,
                      int npts, const float *points, int32_t *out_list, uint8_t *out_codes)
{
    if (d % m != 0) return ORA_ERR_ASSERT;
    const int dsub = d / m;
    float *resid = (float *)malloc(sizeof(float) * (size_t)d);
    float *diffs = (float *)malloc(sizeof(float) * (size_t)ksub);
    if (!resid || !diffs) { free(resid); free(diffs); return ORA_ERR_NOMEM; }
    int rc = ORA_OK;
    for (int p = 0; p < npts && rc == ORA_OK; ++p) {
        const float *pt = points + (size_t)p * d;
        int32_t cl; float cd;
        rc = ora_coarse_search(d, kc, centroids, pt, 1, &cl, &cd);          /* utils.jl:154 */
        if (rc != ORA_OK) break;
        ora_residuals(d, centroids, pt, 1, &cl, resid);                     /* utils.jl:157 */
        for (int i = 0; i < m; ++i) {                                       /* utils.jl:158 */
            colwise_sqeuclidean(dsub, ksub, codebooks + (size_t)i * dsub * ksub,
                                resid + (size_t)i * dsub, diffs);
            int best = 0;
            for (int c = 1; c < ksub; ++c) if (diffs[c] < diffs[best]) best = c;
            out_codes[(size_t)p * m + i] = labels[(size_t)i * ksub + best];
        }
        out_list[p] = cl;
    }
    free(resid); free(diffs);
    return rc;
}

/* Fill n x m synthetic code bytes for positions [g0, g0+n) -- lets a test
 * materialise a slice of a synthetic index and compare it with the device's. */
void ora_synth_fill(uint64_t seed, uint64_t g0, uint64_t n, int m, uint8_t *out)
{
    for (uint64_t g = 0; g < n; ++g)
        for (int b = 0; b < m; ++b) out[g * (uint64_t)m + b] = ora_synth_code(seed, g0 + g, m, b);
}

int ora_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
