// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction for scalar vs packed FP32,
// at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){a[i], a[i] + 1.f}; }
    const float b = seed * 0.5f;
    const v2f pb = (v2f){b, b + 0.25f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = a[i] + b;                       // v_add_f32
                if (MODE == 1) a[i] = a[i] * b;                       // v_mul_f32
                if (MODE == 2) p[i] = p[i] + pb;                      // v_pk_add_f32
                if (MODE == 3) p[i] = p[i] * pb;                      // v_pk_mul_f32
                if (MODE == 4) a[i] = __builtin_fmaf(a[i], b, b);     // v_fma_f32
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int wgs_per_cu)
{
    int ncu = 256;
    int grid = ncu * wgs_per_cu;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = new unsigned long long[grid];
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; ++i) mean += h[i]; mean /= grid;
    const double insts = (double)iters * 32;            // per wave
    // waves per SIMD = wgs_per_cu (4 waves per WG over 4 SIMDs)
    printf("%-14s waves/SIMD=%d  cycles/inst/wave=%6.2f  SIMD cycles per inst=%5.2f  wall=%.3f ms  (%.1f G wave-inst/s chip)\n", name,
           wgs_per_cu, mean / insts, mean / insts / wgs_per_cu, ms, insts * grid * 4 / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc); delete[] h;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_f32", w); run<1>("v_mul_f32", w); run<4>("v_fma_f32", w); run<2>("v_pk_add_f32", w); run<3>("v_pk_mul_f32", w);
    }
    return 0;
}
