// Integer / byte-permute VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction for the ops the narrow-field list scan
// is made of, at 1/2/3/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 ivalu_rate.hip -o ivalu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;

template <int MODE>
__global__ __launch_bounds__(256) void k(u32 *out, unsigned long long *cyc, int iters, u32 seed)
{
    u32 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 3 + i * 7 + threadIdx.x; }
    const u32 c = seed * 5 + 1, sel = 0x07020500u + (seed & 3);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(sel));
                if (MODE == 2) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if (MODE == 3) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "s"(3u));
                if (MODE == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if (MODE == 5) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if (MODE == 7) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 8) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));
                if (MODE == 10) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 11) asm volatile("v_min_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));
                if (MODE == 13) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    u32 s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int wgs_per_cu)
{
    int ncu = 256;
    int grid = ncu * wgs_per_cu;
    u32 *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = new unsigned long long[grid];
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; ++i) mean += h[i]; mean /= grid;
    const double insts = (double)iters * 32;            // per wave
    printf("%-22s waves/SIMD=%d  cycles/inst/wave=%6.2f  SIMD cycles per inst=%5.2f  (%.1f G wave-inst/s chip)\n", name,
           wgs_per_cu, mean / insts, mean / insts / wgs_per_cu, insts * grid * 4 / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc); delete[] h;
}

int main()
{
    for (int w : {1, 2, 3, 4}) {
        run<0>("v_add_u32", w); run<1>("v_perm_b32", w); run<2>("v_bfi_b32", w); run<3>("v_lshlrev_b32_sdwa", w); run<4>("v_add3_u32", w);
        run<5>("v_or_b32", w); run<6>("v_and_or_b32", w); run<7>("v_lshl_add_u32", w); run<8>("v_pk_add_u16", w); run<9>("v_fma_f32", w);
        run<10>("v_min_f32", w); run<11>("v_min_f32_dpp", w); run<12>("v_cndmask_b32", w); run<13>("v_cvt_u32_f32", w);
    }
    return 0;
}
