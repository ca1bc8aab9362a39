// wave_sort.hip -- correctness + latency of the 64-lane bitonic sort built on DPP / permlane-swap lane exchanges
// against the ds_bpermute (__shfl_xor) version.   hipcc --offload-arch=gfx950 -O3 -o wave_sort wave_sort.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define IVF_SORT_ONLY 1
#include "../../ivfadc.jl_amd/csrc/wave_sort.hip.h"

static __device__ __attribute__((noinline)) u64 sort_bpermute(u64 v, int lane)
{
#pragma unroll 1
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            const u32 plo = __shfl_xor((u32)v, j);
            const u32 phi = __shfl_xor((u32)(v >> 32), j);
            const u64 pv = ((u64)phi << 32) | plo;
            const bool keep_min = (((lane & k) == 0) == ((lane & j) == 0));
            const bool p_less = pv < v;
            v = (keep_min == p_less) ? pv : v;
        }
    }
    return v;
}

template <int WHICH> __global__ void k(const u64 *in, u64 *out, u64 *cycles, int reps)
{
    const int lane = threadIdx.x & 63;
    u64 v = in[blockIdx.x * blockDim.x + threadIdx.x];
    const u64 t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        v = WHICH ? ivf::wave_sort64(v, lane) : sort_bpermute(v, lane);
        if (r + 1 < reps) v = v * 0x9E3779B97F4A7C15ull + (u64)lane;   // re-scramble (dependent)
    }
    const u64 t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const int blocks = 1024, threads = 256, n = blocks * threads;
    std::vector<u64> h(n), o(n);
    u64 s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (i % 7 == 0) ? (s >> 60) : s; }
    u64 *din, *dout, *dc;
    hipMalloc(&din, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&dc, blocks * 8);
    hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        if (which) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, din, dout, dc, 1);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, din, dout, dc, 1);
        hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < n / 64; ++w) {
            std::vector<u64> e(h.begin() + w * 64, h.begin() + w * 64 + 64);
            std::sort(e.begin(), e.end());
            for (int i = 0; i < 64; ++i) bad += e[i] != o[w * 64 + i];
        }
        // latency: one wave alone, 64 dependent sorts
        if (which) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, 64);
        else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, din, dout, dc, 64);
        u64 c1 = 0;
        hipMemcpy(&c1, dc, 8, hipMemcpyDeviceToHost);
        // throughput-ish: 16 waves per CU everywhere
        if (which) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, din, dout, dc, 64);
        else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, din, dout, dc, 64);
        u64 c2 = 0;
        hipMemcpy(&c2, dc, 8, hipMemcpyDeviceToHost);
        printf("%s: mismatches=%d  cycles/sort alone=%.0f  with 16 waves/CU=%.0f\n", which ? "dpp+permlane-swap" : "ds_bpermute", bad,
               (double)c1 / 64, (double)c2 / 64);
    }
    return 0;
}
