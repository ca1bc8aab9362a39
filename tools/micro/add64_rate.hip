// v_lshl_add_u64 (one-instruction 64-bit add) against two v_add_u32: SIMD cycles per instruction at 1-4 waves per SIMD (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 add64_rate.hip -o add64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(256) void k(u64 *out, u64 *cyc, int iters, unsigned seed)
{
    u64 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = (u64)seed * 3 + i * 7 + threadIdx.x; }
    u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 1) {
                    unsigned lo = (unsigned)a[i], hi = (unsigned)(a[i] >> 32);
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"((unsigned)b[i]));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(hi) : "v"((unsigned)(b[i] >> 32)));
                    a[i] = ((u64)hi << 32) | lo;
                }
                if (MODE == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(*(unsigned *)&a[i]) : "v"((unsigned)b[i]));
            }
    }
    u64 t1 = __builtin_readcyclecounter();
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int w)
{
    const int grid = 256 * w, iters = 2000;
    u64 *out, *cyc;
    hipMalloc(&out, (size_t)grid * 256 * 8); hipMalloc(&cyc, (size_t)grid * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1u); hipDeviceSynchronize(); }
    u64 *h = new u64[grid];
    hipMemcpy(h, cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; ++i) mean += (double)h[i]; mean /= grid;
    const double per = mean / (iters * 32.0);
    printf("%-34s waves/SIMD=%d  cycles per 64-bit add per wave=%6.2f  SIMD cycles per 64-bit add=%5.2f\n", name, w, per, per / w);
    delete[] h; hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 4; ++w) { run<0>("v_lshl_add_u64", w); run<1>("2 x v_add_u32", w); run<2>("v_pk_add_u16 (32 bits only)", w); }
    return 0;
}
