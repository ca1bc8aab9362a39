// train.hip.h -- device-side index training: k-means++ seeding + Lloyd iterations.
//
// Counterpart of the training half of the reference constructor (/root/reference/src/index.jl:127-147):
// Clustering.kmeans(data, kc; init=:kmpp) for the coarse quantizer and QuantizedArrays.build_quantizer(residuals;
// k, m, method=:pq), i.e. one k-means per sub-space.  Both are third-party and unseeded in the reference, so this
// matches them statistically, not bit for bit.  The assignment step reuses the exact
// coarse-distance kernel of the search path; the update step sums in 64-bit fixed point with integer atomics, so
// the result does not depend on the order in which points arrive: training is deterministic for a given seed.
#pragma once
#include "kernels.hip.h"

namespace ivf {

static __device__ __forceinline__ u64 tr_hash(u64 a, u64 b, u64 c)
{
    return mix64(a + 0x9E3779B97F4A7C15ull * (b + 1) + 0xD1B54A32D192ED03ull * (c + 1));
}

// max |x| over an n x dcols window of a row-major matrix with leading dimension ld (one value per block)
__global__ __launch_bounds__(256) void tr_maxabs_kernel(const float *__restrict__ x, int64_t n, int dcols, int ld,
                                                        float *__restrict__ blockmax)
{
    __shared__ float sm[256];
    float v = 0.0f;
    const int64_t total = n * dcols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t p = e / dcols;
        const int i = (int)(e - p * dcols);
        v = fmaxf(v, fabsf(x[p * ld + i]));
    }
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}

// k-means++ seeding over a strided subsample of S points (sample s = point (s * n) / S).
// update: distance of every sample to the newest centre, running minimum, per-block sums in a fixed tree order.
__global__ __launch_bounds__(256) void tr_kmpp_update_kernel(const float *__restrict__ x, int64_t n, int dcols, int ld, int S,
                                                             const float *__restrict__ centre, float *__restrict__ mind,
                                                             double *__restrict__ partial, int first)
{
    __shared__ double sm[256];
    const int s = blockIdx.x * 256 + threadIdx.x;
    double mine = 0.0;
    if (s < S) {
        const int64_t p = ((int64_t)s * n) / S;
        const float *row = x + p * ld;
        float acc = 0.0f;
        for (int i = 0; i < dcols; ++i) {
            const float t = row[i] - centre[i];
            acc = acc + t * t;
        }
        const float m = first ? acc : fminf(mind[s], acc);
        mind[s] = m;
        mine = (double)m;
    }
    sm[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// pick: centre j = sample drawn with probability proportional to mind (D^2 weighting); one thread, fixed order.
__global__ void tr_kmpp_pick_kernel(const float *__restrict__ x, int64_t n, int dcols, int ld, int S,
                                    const float *__restrict__ mind, const double *__restrict__ partial, int nblocks, u64 seed,
                                    int j, float *__restrict__ centres)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int s;
    if (j == 0) {
        s = (int)(tr_hash(seed, 0, 0) % (u64)S);
    } else {
        double total = 0.0;
        for (int b = 0; b < nblocks; ++b) total += partial[b];
        const double u = (double)(tr_hash(seed, (u64)j, 1) >> 11) * (1.0 / 9007199254740992.0);
        if (total <= 0.0) {
            s = (int)(tr_hash(seed, (u64)j, 2) % (u64)S);
        } else {
            const double r = u * total;
            double run = 0.0;
            int b = 0;
            while (b < nblocks - 1 && run + partial[b] <= r) { run += partial[b]; ++b; }
            s = b * 256;
            const int hi = min(S, b * 256 + 256);
            while (s < hi - 1 && run + (double)mind[s] <= r) { run += (double)mind[s]; ++s; }
        }
    }
    const int64_t p = ((int64_t)s * n) / S;
    for (int i = 0; i < dcols; ++i) centres[(size_t)j * dcols + i] = x[p * ld + i];
}

// update step, part 1: fixed-point sums (order-independent) and counts
__global__ __launch_bounds__(256) void tr_accumulate_kernel(const float *__restrict__ x, int64_t n, int dcols, int ld,
                                                            const int *__restrict__ assign, double scale,
                                                            long long *__restrict__ acc, u32 *__restrict__ counts)
{
    const int64_t total = n * dcols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t p = e / dcols;
        const int i = (int)(e - p * dcols);
        const int c = assign[p];
        const long long v = __double2ll_rn((double)x[p * ld + i] * scale);
        atomicAdd((unsigned long long *)&acc[(size_t)c * dcols + i], (unsigned long long)v);
        if (i == 0) atomicAdd(&counts[c], 1u);
    }
}

// update step, part 2: means; an empty cluster restarts from a pseudo-random point; flags any change
__global__ __launch_bounds__(256) void tr_finalize_kernel(const float *__restrict__ x, int64_t n, int dcols, int ld, int k,
                                                          const long long *__restrict__ acc, const u32 *__restrict__ counts,
                                                          double inv_scale, u64 seed, int iter, float *__restrict__ centres,
                                                          int *__restrict__ changed)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= k * dcols) return;
    const int c = e / dcols, i = e - c * dcols;
    float v;
    if (counts[c] > 0) {
        v = (float)((double)acc[e] * inv_scale / (double)counts[c]);
    } else {
        const int64_t p = (int64_t)(tr_hash(seed, (u64)iter + 7777, (u64)c) % (u64)n);
        v = x[p * ld + i];
    }
    if (__float_as_uint(v) != __float_as_uint(centres[e])) {
        centres[e] = v;
        *changed = 1;
    }
}

// residuals of every point against its centroid (index.jl:168-175)
__global__ __launch_bounds__(256) void tr_residual_kernel(const float *__restrict__ x, int64_t n, int d, const int *__restrict__ assign,
                                                          const float *__restrict__ centres, float *__restrict__ out)
{
    const int64_t total = n * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t p = e / d;
        const int i = (int)(e - p * d);
        out[e] = x[e] - centres[(size_t)assign[p] * d + i];
    }
}

}  // namespace ivf
