// mfma4_probe.hip -- facts the lower-bound table build (kernels.hip.h, lbq_* / qscan_lb_kernel) relies on, probed on the device:
//   1. operand / result layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave)
//   2. rounding of v_cvt_pk_u8_f32 (saturating? nearest or truncating?)
//   3. gather rate of ds_read_u8 from 256-byte tables (the u8 lookup of the filter scan) vs ds_read_b32 from 1 KB tables
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4_probe mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

static unsigned short f2bf(float x)
{
    unsigned b;
    memcpy(&b, &x, 4);
    return (unsigned short)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}

__global__ void mfma_kernel(const s4 *a, const s4 *b, f4 *c)
{
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}

__global__ void mfma_f32_kernel(const float *a, const float *b, f4 *c)
{
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}

__global__ void cvt_kernel(const float *in, unsigned *out, unsigned *out2, int n)
{
    const int i = threadIdx.x;
    if (i < n) {
        unsigned r = 0xAABBCC00u;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
        out[i] = r;
        out2[i] = (unsigned)in[i];   // v_cvt_u32_f32
    }
}

// ---- gather rates: every lane looks up NL random codes per iteration in m tables
template <int MODE>   // 0: ds_read_u8, tables of 256 B; 1: ds_read_b32, tables of 1 KB; 2: ds_read_u8 with 16 u8 tables interleaved per dword column
__global__ __launch_bounds__(256) void gather_kernel(const unsigned *codes, unsigned *out, int iters, int m)
{
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x;
    const int tb = MODE == 1 ? 1024 : 256;
    for (int i = tid; i < m * tb; i += 256) smem[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    unsigned cw[12];
    for (int k = 0; k < 12; ++k) cw[k] = codes[(blockIdx.x * 256 + tid) * 12 + k];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ii = k * 4 + b;
                const unsigned byte = (cw[k] >> (8 * b)) & 0xffu;
                if (MODE == 1) acc += *(const unsigned *)(smem + ii * 1024 + byte * 4);
                else acc += smem[ii * 256 + byte];
            }
            cw[k] = cw[k] * 1664525u + 1013904223u + acc;
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}

int main()
{
    // ---- 1. layout
    std::vector<unsigned short> ha(64 * 4), hb(64 * 4);
    std::vector<float> fa(64 * 4), fb(64 * 4);
    srand(1);
    for (int i = 0; i < 256; ++i) {
        fa[i] = (float)((rand() % 17) - 8);
        fb[i] = (float)((rand() % 13) - 6);
        ha[i] = f2bf(fa[i]);
        hb[i] = f2bf(fb[i]);
    }
    s4 *da, *db;
    f4 *dc;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 1024);
    hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dc);
    std::vector<float> hc(256);
    hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
    // hypothesis H1: block = lane / 4; A lane (4 blk + i) holds row i (k = 0..3); B lane (4 blk + j) holds column j; D lane (4 blk + j)
    // register i = sum_k A[i][k] B[k][j]
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4, i = r;
            float e1 = 0.f, e2 = 0.f;
            for (int k = 0; k < 4; ++k) {
                e1 += fa[(blk * 4 + i) * 4 + k] * fb[(blk * 4 + j) * 4 + k];
                e2 += fa[(blk * 4 + j) * 4 + k] * fb[(blk * 4 + i) * 4 + k];   // H2: transposed roles
            }
            if (hc[l * 4 + r] != e1) ++bad1;
            if (hc[l * 4 + r] != e2) ++bad2;
        }
    printf("mfma_f32_4x4x4_16b_bf16 layout: H1 (D[lane=4b+j][reg i] = sum_k A[lane 4b+i][k] B[lane 4b+j][k]) mismatches=%d; H2 (transposed) mismatches=%d\n", bad1, bad2);

    {   // v_mfma_f32_4x4x1_16b_f32: one k-step, same block structure
        float *fa1, *fb1;
        hipMalloc(&fa1, 256); hipMalloc(&fb1, 256);
        std::vector<float> a1(64), b1(64);
        for (int i = 0; i < 64; ++i) { a1[i] = (float)((rand() % 17) - 8); b1[i] = (float)((rand() % 13) - 6); }
        hipMemcpy(fa1, a1.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(fb1, b1.data(), 256, hipMemcpyHostToDevice);
        mfma_f32_kernel<<<1, 64>>>(fa1, fb1, dc);
        hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r)
                if (hc[l * 4 + r] != a1[(l / 4) * 4 + r] * b1[l]) ++bad;
        printf("mfma_f32_4x4x1_16b_f32 layout: D[lane=4b+j][reg i] = A[lane 4b+i] B[lane 4b+j] mismatches=%d\n", bad);
    }
    // ---- 2. cvt
    const float tv[] = {-3.f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.49f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.9f, 255.0f, 255.4f, 255.6f, 300.f, 1e9f, 0.999f, 1.0f};
    const int nt = sizeof(tv) / 4;
    float *di;
    unsigned *dout, *dout2;
    hipMalloc(&di, 256); hipMalloc(&dout, 256); hipMalloc(&dout2, 256);
    hipMemcpy(di, tv, nt * 4, hipMemcpyHostToDevice);
    cvt_kernel<<<1, 64>>>(di, dout, dout2, nt);
    unsigned ho[64], ho2[64];
    hipMemcpy(ho, dout, nt * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ho2, dout2, nt * 4, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32 (byte 0 of 0xAABBCC00) and v_cvt_u32_f32:\n");
    for (int i = 0; i < nt; ++i) printf("  %12.4f -> pk_u8 0x%08x (%u)   cvt_u32 %u\n", tv[i], ho[i], ho[i] & 255u, ho2[i]);

    // ---- 3. gather rates
    const int nblk = 256 * 4, iters = 200;
    std::vector<unsigned> hcodes((size_t)nblk * 256 * 12);
    for (auto &v : hcodes) v = (unsigned)rand() * 2654435761u + (unsigned)rand();
    unsigned *dcodes, *dres;
    hipMalloc(&dcodes, hcodes.size() * 4); hipMalloc(&dres, (size_t)nblk * 256 * 4);
    hipMemcpy(dcodes, hcodes.data(), hcodes.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        const size_t lds = mode == 1 ? 48 * 1024 : 48 * 256;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) gather_kernel<0><<<nblk, 256, lds>>>(dcodes, dres, iters, 48);
            else gather_kernel<1><<<nblk, 256, lds>>>(dcodes, dres, iters, 48);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double lookups = (double)nblk * 256 * 48 * iters;
            if (rep == 2)
                printf("gather %s: %.3f ms, %.1f lookups/clk/CU at 2.4 GHz nominal (256 CUs), LDS %zu B per workgroup\n",
                       mode == 0 ? "ds_read_u8 from 256-B tables" : "ds_read_b32 from 1-KB tables", ms, lookups / (ms * 1e-3) / 2.4e9 / 256.0, lds);
        }
    }
    return 0;
}
