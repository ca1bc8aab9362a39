// smallq.hip.h -- the latency path: a handful of queries (the reference's primary entry is ONE query, w = 1: src/index.jl:204-208).
// Included by kernels.hip.h, namespace ivf.
//
// The batch kernels are built for throughput: a coarse launch, a top-w launch, a scan launch whose work items walk a query's
// probes (query-major) or lists (list-major), and a merge launch -- four dependent launches and, for one query, a single
// workgroup doing eight table builds in sequence.  Here ONE launch does everything, (query, probe, chunk)-parallel:
//
//   workgroup (q, j, c):  coarse distances of query q (every workgroup of the query computes them for itself -- kc * d sub / mul /
//                         add in the reference's order, coarsequantizers.jl:34 -- when kc <= SQ_COARSE_INSIDE; for a larger
//                         coarse quantizer the row comes from the exact coarse kernel, one launch earlier)
//                      -> top-w (stable: ties to the lower cell, coarsequantizers.jl:35-36), identical in every workgroup of q
//                      -> probe j alone: residual, one f32 table set (index.jl:232-236), points [c CH, (c + 1) CH) of the list
//                         (index.jl:240-246) through the four wave selectors, merged to a partial top-K in HBM
//                      -> arrival counter of q; the LAST workgroup to arrive merges the w * nch partial results (index.jl:247-257)
//                         and writes ids, distances and the count.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, the fence-free row): the one storing wave writes its keys with
// agent-scope (sc1, write-through) stores and waits for them (vmcnt(0)), the workgroup meets at its barrier, lane 0 adds to the
// query's counter with an agent-scope atomic; the workgroup whose add returns w * nch - 1 is the last one and reads the other
// workgroups' keys with agent-scope (sc1) loads, behind its own add and a workgroup barrier.  No workgroup ever waits for another,
// so the launch drains whatever the dispatch order is.  A query that is ONE workgroup (w = 1, short list) skips all of it.
#pragma once

constexpr int SQ_COARSE_INSIDE = 2048;   // largest coarse quantizer whose search is repeated in every workgroup of a query

struct SqArgs {
    IndexView ix;
    const float *queries;
    int nq, w, K, nch;       // nch chunks of CH points per probe
    u32 CH;
    const float *cdist;      // [nq][kc] exact coarse distances (kc > SQ_COARSE_INSIDE), else null
    const float *centroids_t;   // [d / 4][kc][4]: the centroids regrouped for the search inside the launch (cdist == null)
    u64 *part_keys;          // [nq][w * nch][K]
    u32 *part_cnt;           // [nq][w * nch]
    u32 *arrive;             // [nq], zero between launches
    u32 *out_ids;
    float *out_dists;
    int *out_counts;
    u64 *scanned_points;
};

// Exact coarse distances for a HANDFUL of queries (the launch in front of sq_kernel): a lane owns a (centroid, query) pair and walks
// the regrouped centroids [d / 4][kc][4] -- a wave's load is 1 KB coalesced, no LDS staging, no barrier --, the sum in the reference's
// order (i ascending; sub, mul, add: coarsequantizers.jl:34).  The chain is short (3 d dependent operations); what a lane waits for is
// its centroid, so sixteen 16-byte groups (64 dimensions) are requested at once.  Against coarse_sgpr_kernel's tiles (stage 32 KB,
// barrier, 128-step sum for 16 queries of which one is real) this is half the time for one query.
template <int NG_FIX>
__global__ __launch_bounds__(256) void coarse_lane_kernel(const float4 *__restrict__ ct, const float *__restrict__ Q, float *__restrict__ out,
                                                          int nq, int kc, int d)
{
    const int c = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    if (c >= kc || q >= nq) return;
    const float *qv = Q + (size_t)q * d;
    const int ng = d >> 2;     // d % 4 == 0
    float acc = 0.f;
    if constexpr (NG_FIX > 0) {
        // d = 4 NG_FIX (<= 128): the whole centroid is requested at once -- ONE trip to L2 instead of two; the sum stays the
        // reference's single chain (i ascending; sub, mul, add: the adds cannot be split across lanes bit for bit)
        float4 x[NG_FIX];
#pragma unroll
        for (int u = 0; u < NG_FIX; ++u) x[u] = ct[(size_t)u * kc + c];
#pragma unroll
        for (int u = 0; u < NG_FIX; ++u) {
            const float4 qq = *(const float4 *)(qv + 4 * u);
            float t = x[u].x - qq.x; acc = acc + t * t;
            t = x[u].y - qq.y; acc = acc + t * t;
            t = x[u].z - qq.z; acc = acc + t * t;
            t = x[u].w - qq.w; acc = acc + t * t;
        }
    } else {
    for (int g0 = 0; g0 < ng; g0 += 16) {
        float4 x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = ct[(size_t)(g0 + u < ng ? g0 + u : g0) * kc + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (g0 + u < ng) {   // uniform
                const float4 qq = *(const float4 *)(qv + 4 * (g0 + u));
                float t = x[u].x - qq.x; acc = acc + t * t;
                t = x[u].y - qq.y; acc = acc + t * t;
                t = x[u].z - qq.z; acc = acc + t * t;
                t = x[u].w - qq.w; acc = acc + t * t;
            }
        }
    }
    }
    out[(size_t)q * kc + c] = acc;
}

template <int M, int DS>
__global__ __launch_bounds__(256) void sq_kernel(const SqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const IndexView &ix = a.ix;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = (M > 0) ? M : ix.m;
    const int K = a.K, w = a.w, nch = a.nch, per_q = w * nch;
    const int q = blockIdx.x / per_q, rem = blockIdx.x - q * per_q;
    const int j = rem / nch, c = rem - j * nch;
    const LdsCarve L = carve_lds<1, true>(smem_raw, m, ix.d, 64);
    // behind the carve: probes of the query (64 entries each: cell, coarse distance, visit-order base), scratch of the row selection,
    // then the row of coarse distances
    int *s_list = (int *)(L.sthr + STHR_WORDS);
    float *s_dc = (float *)(s_list + 64);
    u32 *s_base = (u32 *)(s_dc + 64);
    u64 *wbound = (u64 *)(s_base + 64);          // [4]
    u32 *ccnt = (u32 *)(wbound + 4);             // [2]: candidate counter, "last arriver" flag
    float *s_row = (float *)(((size_t)(ccnt + 4) + 15) & ~(size_t)15);   // 16-byte rows: the selection reads float4
    const float *qv = a.queries + (size_t)q * ix.d;

    // ---- coarse distances of the query
    const float *row = a.cdist ? a.cdist + (size_t)q * ix.kc : s_row;
    if (!a.cdist) {
        // a lane owns a centroid; centroids_t = [d / 4][kc][4] (adjacent centroids in adjacent 16-byte groups), so a wave's load is 1 KB
        // coalesced -- a lane walking its own row of the [kc][d] matrix touches 64 cache lines per load instruction (measured: 15 us
        // for kc = 1024, d = 128 against 9 us for the separate kernel).  Four centroids per thread at a time, four independent sums, each
        // in the reference's order (i ascending; sub, mul, add).
        const float4 *ct = (const float4 *)a.centroids_t;
        for (int c0 = tid; c0 < ix.kc; c0 += 1024) {
            const int c1 = c0 + 256 < ix.kc ? c0 + 256 : c0, c2 = c0 + 512 < ix.kc ? c0 + 512 : c0, c3 = c0 + 768 < ix.kc ? c0 + 768 : c0;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            auto acc4 = [](float s, const float4 &cv, const float4 &qq) {
                float t = cv.x - qq.x; s = s + t * t;
                t = cv.y - qq.y; s = s + t * t;
                t = cv.z - qq.z; s = s + t * t;
                t = cv.w - qq.w; s = s + t * t;
                return s;
            };
#pragma unroll 4
            for (int g = 0; g < (ix.d >> 2); ++g) {     // d % 4 == 0
                const float4 qq = *(const float4 *)(qv + 4 * g);
                const float4 *row4 = ct + (size_t)g * ix.kc;
                const float4 x0 = row4[c0], x1 = row4[c1], x2 = row4[c2], x3 = row4[c3];
                a0 = acc4(a0, x0, qq); a1 = acc4(a1, x1, qq); a2 = acc4(a2, x2, qq); a3 = acc4(a3, x3, qq);
            }
            s_row[c0] = a0;
            if (c0 + 256 < ix.kc) s_row[c1] = a1;
            if (c0 + 512 < ix.kc) s_row[c2] = a2;
            if (c0 + 768 < ix.kc) s_row[c3] = a3;
        }
    }
    if (tid == 0) arm_bound<1>(L.sthr, 0, KEY_MAX);
    __syncthreads();

    // ---- top-w (the query-major prologue's selection, exact distances)
    WSel<true> ws;
    ws.init(KEY_MAX, nullptr, 64, w);
    bool have = false;
    if (w <= SHORT_ROW_MAXK && ix.kc <= 8192 && (ix.kc & 3) == 0) {
        if (ix.kc <= 2048) have = select_row_short<false, 2>(ws, row, ix.kc, w, wv, lane, tid, L.xch, wbound, ccnt);
        else if (ix.kc <= 4096) have = select_row_short<false, 4>(ws, row, ix.kc, w, wv, lane, tid, L.xch, wbound, ccnt);
        else have = select_row_short<false, 8>(ws, row, ix.kc, w, wv, lane, tid, L.xch, wbound, ccnt);
    }
    if (!have) {
        ws.init(KEY_MAX, nullptr, 64, w);
        select_row<false, 4>(ws, row, ix.kc, w, wv, lane, L.sthr);
        const int wc = ws.finish(w, lane);
        ws.store(L.xch + (size_t)wv * 64, wc, lane);
        if (lane == 0) L.scnt[wv] = wc;
        __syncthreads();
        if (wv == 0) merge_waves(ws, L.xch, (size_t)64, L.scnt, 1, w, KEY_MAX, 0, lane);
    }
    if (wv == 0) {
        const int fc = ws.finish(w, lane);       // == min(w, kc)
        u32 len = 0;
        int l = 0;
        if (lane < fc) {
            l = (int)(u32)ws.top;
            len = ix.list_len[l];
        }
        u32 incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane < fc) {
            s_list[lane] = l;
            s_dc[lane] = __uint_as_float((u32)(ws.top >> 32));
            s_base[lane] = incl - len;
        }
        const u32 total = __shfl(incl, 63);      // by every lane: a shuffle under EXEC = lane 0 would read an inactive lane 63
        if (lane == 0) {
            arm_bound<1>(L.sthr, 0, KEY_MAX);    // re-armed for the scan
            if (j == 0 && c == 0) atomicAdd(a.scanned_points + (size_t)(q & 63) * 8, (u64)total);
        }
    }
    __syncthreads();

    // ---- probe j, chunk c
    const int l = s_list[j];
    const float dcv = s_dc[j];
    const u32 sb = s_base[j];
    const u32 len = ix.list_len[l];
    const u32 p0 = (u32)c * a.CH, p1 = min(len, p0 + a.CH);
    WSel<true> sel[1];
    sel[0].init(KEY_MAX, nullptr, 64, K);
    int fc = 0;
    if (p0 < len) {   // uniform
        const uint8_t *cbase = ix.codes + ix.list_codeoff[l];
        CodeRegs<M, ppl_of<M, 1>()> cr;
        scan_prefetch(cr, cbase, p0, p1, wv, lane);
        const int qi[1] = {q}, li[1] = {l};
        build_residuals<1>(ix, a.queries, qi, li, L.resid, tid);
        __syncthreads();
        build_tables_t<1, DS, TAB_SEP>(ix, m, L.resid, L.tab, tid);
        __syncthreads();
        const float dc1[1] = {dcv};
        const u32 sb1[1] = {sb};
        scan_range<M, 1>(L.tab, 0u, cbase, ix.cs, m, p0, p1, dc1, sb1, 1, sel, K, wv, lane, cr, L.sthr, 0);
        const int mycnt = sel[0].finish(K, lane);
        __syncthreads();              // the exchange area aliases the tables
        sel[0].store(L.xch + (size_t)wv * L.xcap, mycnt, lane);
        if (lane == 0) L.scnt[wv] = mycnt;
        __syncthreads();
        if (wv == 0) {
            merge_waves(sel[0], L.xch, (size_t)L.xcap, L.scnt, 1, K, KEY_MAX, 0, lane);
            fc = sel[0].finish(K, lane);
        }
    }
    if (per_q == 1) {   // one workgroup is the whole query: nothing to hand over
        if (wv == 0) {
            sel[0].for_each(fc, lane, [&](int i, u64 key) { emit_result(key, i, q, w, K, s_list, s_base, ix.list_pos, ix.ids, a.out_ids, a.out_dists); });
            if (lane == 0) a.out_counts[q] = fc;
        }
        return;
    }
    const size_t slot = (size_t)q * per_q + rem;
    if (wv == 0) {
        // write-through (sc1) stores of the partial result, drained before the arrival is counted; the last arriver reads with sc1 loads:
        // the hand-off form of MI355X_MICROARCH.md that needs neither a release nor an acquire fence (1.7 - 6.5 us each)
        u64 *dst = a.part_keys + slot * K;
        sel[0].for_each(fc, lane, [&](int i, u64 key) { __hip_atomic_store(&dst[i], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
        if (lane == 0) __hip_atomic_store(&a.part_cnt[slot], (u32)fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        const u32 prev = __hip_atomic_fetch_add(&a.arrive[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = prev == (u32)per_q - 1u;
        ccnt[1] = last ? 1u : 0u;
        if (last) __hip_atomic_store(&a.arrive[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every arrival of this launch is in: re-armed
    }
    __syncthreads();
    if (ccnt[1] == 0u || wv != 0) return;   // uniform per wave

    // ---- last arriver, wave 0: K smallest of the w * nch partial results, visit order -> stored id (index.jl:248,252,257)
    WSel<true> fin;
    fin.init(KEY_MAX, nullptr, 64, K);
    const int T = per_q * K;
    for (int e0 = 0; e0 < T; e0 += 64) {
        const int e = e0 + lane;
        bool pred = e < T;
        u64 key = KEY_MAX;
        if (pred) {
            const int s = e / K, i = e - s * K;
            const size_t sl = (size_t)q * per_q + s;
            pred = (u32)i < __hip_atomic_load(&a.part_cnt[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pred) key = __hip_atomic_load(&a.part_keys[sl * K + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pred = pred && key < fin.thr();
        fin.push(pred, key, K, lane);
    }
    const int nres = fin.finish(K, lane);
    fin.for_each(nres, lane, [&](int i, u64 key) { emit_result(key, i, q, w, K, s_list, s_base, ix.list_pos, ix.ids, a.out_ids, a.out_dists); });
    if (lane == 0) a.out_counts[q] = nres;
}
