// generic.hip.h -- the path for what the selection kernels do not cover (K > 2048 or w > 2048; also selectable with
// ivfadc_set_tuning(h, -2, 0) as an independent second implementation for cross-checks).
//
// Same semantics, no selection structures: every (query, probed point) pair gets its key
// f32_bits(distance) << 32 | visit order written out, the keys of a query are sorted (rocPRIM segmented radix sort),
// and the first K are the result (index.jl:225-257: the K smallest under (distance, visit order)).  The coarse top-w
// is taken the same way: sort the kc keys f32_bits(distance) << 32 | cluster of a row, keep the first w
// (coarsequantizers.jl:33-37, stable sortperm => ties to the lower cluster id).  Any m, ksub, labels, K and w.
#pragma once
#include "kernels.hip.h"

namespace ivf {

// keys of one row of coarse distances
__global__ __launch_bounds__(256) void gen_row_keys_kernel(const float *__restrict__ cdist, int64_t total, int kc, u64 *__restrict__ keys)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        keys[e] = ((u64)__float_as_uint(cdist[e]) << 32) | (u32)(e % kc);
}

// The w best clusters of every query from its sorted row; visit-order base of each probe (exclusive prefix of the list
// lengths in probe order) and the query's number of probed points.  One workgroup per query.
__global__ __launch_bounds__(256) void gen_probes_kernel(const u64 *__restrict__ sorted, int kc, int w, const u32 *__restrict__ list_len,
                                                         int *__restrict__ probe_list, float *__restrict__ probe_dc,
                                                         u32 *__restrict__ probe_base, u32 *__restrict__ totals,
                                                         u64 *__restrict__ scanned_points)
{
    __shared__ u32 s_part[256];
    __shared__ u32 s_run;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int j0 = 0; j0 < w; j0 += 256) {
        const int j = j0 + tid;
        u32 len = 0;
        int l = 0;
        float dc = 0.0f;
        if (j < w) {
            const u64 key = sorted[(size_t)q * kc + j];
            l = (int)(u32)key;
            dc = __uint_as_float((u32)(key >> 32));
            len = list_len[l];
        }
        s_part[tid] = len;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {   // inclusive scan (Hillis-Steele)
            const u32 v = tid >= off ? s_part[tid - off] : 0u;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const u32 run = s_run;
        if (j < w) {
            probe_list[(size_t)q * w + j] = l;
            probe_dc[(size_t)q * w + j] = dc;
            probe_base[(size_t)q * w + j] = run + s_part[tid] - len;
        }
        __syncthreads();
        if (tid == 0) s_run = run + s_part[255];
        __syncthreads();
    }
    if (tid == 0) {
        totals[q] = s_run;
        atomicAdd(scanned_points + (size_t)(q & 63) * 8, (u64)s_run);
    }
}

// ADC distance of every point of one probed list: residual (coarsequantizers.jl:40-45), table (index.jl:232-236),
// d = dc; d += tab_ii[code[ii]] in ascending ii (index.jl:240-246).  One workgroup per (probe, query).
__global__ __launch_bounds__(256) void gen_dump_kernel(const IndexView ix, const float *__restrict__ queries, int w,
                                                       const int *__restrict__ probe_list, const float *__restrict__ probe_dc,
                                                       const u32 *__restrict__ probe_base, const u32 *__restrict__ key_off,
                                                       u64 *__restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *resid = (float *)smem_raw;                       // [d]
    float *tab = resid + (((size_t)ix.d + 3) & ~(size_t)3);   // [m][256]
    const int j = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const int l = probe_list[(size_t)q * w + j];
    const u32 len = ix.list_len[l];
    if (len == 0) return;   // uniform
    for (int i = tid; i < ix.d; i += 256) resid[i] = queries[(size_t)q * ix.d + i] - ix.centroids[(size_t)l * ix.d + i];
    __syncthreads();
    for (int e = tid; e < ix.m * 256; e += 256) {
        const int ii = e >> 8, c = e & 255;
        if (c < ix.ksub) {
            const float *cw = ix.codebooks + ((size_t)ii * ix.ksub + c) * ix.dsub;
            const float *rr = resid + (size_t)ii * ix.dsub;
            float sum = 0.0f;
            for (int t = 0; t < ix.dsub; ++t) {
                const float df = cw[t] - rr[t];
                sum = sum + df * df;
            }
            tab[ii * 256 + (ix.identity_labels ? c : (int)ix.labels[ii * ix.ksub + c])] = sum;
        }
    }
    __syncthreads();
    const float dc = probe_dc[(size_t)q * w + j];
    const u32 base = probe_base[(size_t)q * w + j];
    const uint8_t *codes = ix.codes + ix.list_codeoff[l];
    u64 *dst = keys + key_off[q] + base;
    for (u32 p = tid; p < len; p += 256) {
        const uint8_t *cp = codes + (size_t)p * ix.cs;
        float dist = dc;
        for (int ii = 0; ii < ix.m; ++ii) dist = dist + tab[ii * 256 + cp[ii]];
        dst[p] = ((u64)__float_as_uint(dist) << 32) | (base + p);
    }
}

// first K sorted keys of every query -> ids and distances (index.jl:248,252,257)
__global__ __launch_bounds__(256) void gen_emit_kernel(const u64 *__restrict__ sorted, const u32 *__restrict__ key_off,
                                                       const u32 *__restrict__ totals, int w, int K, const int *__restrict__ probe_list,
                                                       const u32 *__restrict__ probe_base, const int64_t *__restrict__ list_pos,
                                                       const u32 *__restrict__ ids, u32 *__restrict__ out_ids, float *__restrict__ out_dists,
                                                       int *__restrict__ out_counts)
{
    const int q = blockIdx.x;
    const u32 tot = totals[q];
    const int cnt = tot < (u32)K ? (int)tot : K;
    for (int i = threadIdx.x; i < cnt; i += 256)
        emit_result(sorted[key_off[q] + i], i, q, w, K, probe_list + (size_t)q * w, probe_base + (size_t)q * w, list_pos, ids, out_ids,
                    out_dists);
    if (threadIdx.x == 0) out_counts[q] = cnt;
}

}  // namespace ivf
