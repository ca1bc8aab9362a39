// LDS gather-rate microbenchmark (gfx950): table lookups per clock and CU for the access forms the scan kernels use.
//
//   random   the round-1 layouts: tab[ii][code][QG] -- the bank (group) of a lookup is code mod (banks / entry width),
//            so the 16 / 32 lanes of a ds_read service group collide at random (2.7-way for b128, 3.5-way for b32);
//   striped  the round-2 layouts: lane l looks sub-quantizer (t + l) mod m up at step t and every sub-quantizer owns a
//            bank stripe, so the lanes of a service group can only collide inside their own stripe (or not at all when
//            the stripe is as wide as the lanes that share it).
//
// Addresses are fixed per lane and step (8 steps, reused every iteration): the loop measures the LDS array, not the
// address arithmetic.  Results feed profiles/lds_roof.json (bench.py's roofline_lds).
// Build: hipcc --offload-arch=gfx950 -O3 lds_gather.hip -o lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum Mode {
    B32_RANDOM = 0,        // m = 8:  addr = s * 1024 + c * 4
    B32_STRIPE4 = 1,       // m = 8:  4-bank stripe per sub-quantizer: addr = (c >> 2) * 128 + s * 16 + (c & 3) * 4
    B32_M16_RANDOM = 2,    // m = 16: addr = s * 1024 + c * 4
    B32_M16_STRIPE2 = 3,   // m = 16: 2-bank stripe: addr = (c >> 1) * 128 + s * 8 + (c & 1) * 4
    B32_M16_2COPY = 4,     // m = 16: two copies, conflict-free: addr = c * 128 + s * 8 + copy * 4
    B64_RANDOM = 5,        // QG = 2: addr = s * 2048 + c * 8
    B128_RANDOM = 6,       // QG = 4: addr = s * 4096 + c * 16
    B128_STRIPED = 7,      // QG = 4, two copies (or QG = 8 split over lane pairs): addr = c * 256 + s * 32 + copy * 16
    B128_STRIPE_NOCOPY = 8,// QG = 4, one copy: addr = (c >> 1) * 256 + s * 32 + (c & 1) * 16
    B32_SEQ = 9,           // conflict-free reference: addr = lane * 4
    B128_SEQ = 10,         // conflict-free reference: addr = lane * 16
    B64_STRIPE_Q16 = 11,   // QG = 4 as four 16-bit fields (m = 8 integer filter): addr = c * 64 + s * 8
    B64_SEQ = 12,          // conflict-free reference: addr = lane * 8
    NMODES = 13
};

static const char *mode_name[NMODES] = {"b32 random (m=8)", "b32 stripe x4 (m=8)", "b32 random (m=16)", "b32 stripe x2 (m=16)",
                                        "b32 2 copies (m=16)", "b64 random (QG=2)", "b128 random (QG=4)", "b128 striped, 2 copies",
                                        "b128 striped, 1 copy", "b32 sequential", "b128 sequential",
                                        "b64 striped, 4 x u16 (QG=4)", "b64 sequential"};

template <int W> struct Ld;
template <> struct Ld<4> { static __device__ __forceinline__ float ld(unsigned a) { return *(const __attribute__((address_space(3))) float *)(size_t)a; } };
template <> struct Ld<8> { static __device__ __forceinline__ float ld(unsigned a) { v2f v = *(const __attribute__((address_space(3))) v2f *)(size_t)a; return v.x + v.y; } };
template <> struct Ld<16> { static __device__ __forceinline__ float ld(unsigned a) { v4f v = *(const __attribute__((address_space(3))) v4f *)(size_t)a; return (v.x + v.y) + (v.z + v.w); } };

template <int W>
__global__ __launch_bounds__(256) void k(const unsigned *__restrict__ addr_in, float *out, unsigned long long *cyc, int iters, unsigned lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (unsigned i = threadIdx.x * 4; i < lds_bytes; i += 1024) *(float *)(smem + i) = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    unsigned a[8];
    for (int t = 0; t < 8; ++t) a[t] = addr_in[((size_t)blockIdx.x * 8 + t) * 256 + threadIdx.x] & (lds_bytes - 1);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            asm volatile("" : "+v"(a[t]));   // the address is opaque to the optimiser: the read stays in the loop
            acc[t] += Ld<W>::ld(a[t]);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static unsigned address(int mode, int lane, int t, unsigned c)
{
    switch (mode) {
    case B32_RANDOM: return (unsigned)(t & 7) * 1024 + c * 4;
    case B32_STRIPE4: { unsigned s = (t + lane) & 7; return (c >> 2) * 128 + s * 16 + (c & 3) * 4; }
    case B32_M16_RANDOM: return (unsigned)(t & 15) * 1024 + c * 4;
    case B32_M16_STRIPE2: { unsigned s = (t + lane) & 15; return (c >> 1) * 128 + s * 8 + (c & 1) * 4; }
    case B32_M16_2COPY: { unsigned s = (t + lane) & 15; return c * 128 + s * 8 + ((lane >> 4) & 1) * 4; }
    case B64_RANDOM: return (unsigned)(t & 7) * 2048 + c * 8;
    case B128_RANDOM: return (unsigned)(t & 7) * 4096 + c * 16;
    case B128_STRIPED: { unsigned s = (t + lane) & 7; return c * 256 + s * 32 + ((lane >> 3) & 1) * 16; }
    case B128_STRIPE_NOCOPY: { unsigned s = (t + lane) & 7; return (c >> 1) * 256 + s * 32 + (c & 1) * 16; }
    case B32_SEQ: return (unsigned)lane * 4 + (unsigned)t * 256;
    case B64_STRIPE_Q16: { unsigned s = (t + lane) & 7; return c * 64 + s * 8; }
    case B64_SEQ: return (unsigned)lane * 8 + (unsigned)t * 512;
    default: return (unsigned)lane * 16 + (unsigned)t * 1024;
    }
}

int main(int argc, char **argv)
{
    const int ncu = 256, iters = 4000;
    const bool json = argc > 1 && !strcmp(argv[1], "--json");
    float *out; unsigned long long *cyc; unsigned *addr;
    const int maxgrid = ncu * 8;
    hipMalloc(&out, (size_t)maxgrid * 256 * 4); hipMalloc(&cyc, (size_t)maxgrid * 8); hipMalloc(&addr, (size_t)maxgrid * 8 * 256 * 4);
    std::vector<unsigned> h((size_t)maxgrid * 8 * 256);
    if (json) printf("{\n");
    bool first = true;
    for (int mode = 0; mode < NMODES; ++mode) {
        const int W = (mode == B64_RANDOM || mode == B64_STRIPE_Q16 || mode == B64_SEQ) ? 8 : (mode >= B128_RANDOM && mode != B32_SEQ) ? 16 : 4;
        const int QPL = (mode == B64_STRIPE_Q16 || mode == B64_SEQ) ? 4 : W / 4;   // query entries per lane lookup (16-bit fields: four in 8 bytes)
        const unsigned full = (mode == B128_STRIPED) ? 65536u : (mode == B128_RANDOM || mode == B128_STRIPE_NOCOPY || mode == B32_M16_2COPY) ? 32768u
                              : (mode == B32_M16_RANDOM || mode == B32_M16_STRIPE2 || mode == B64_RANDOM || mode == B64_STRIPE_Q16) ? 16384u : 8192u;
        for (int wgs : {1, 2, 3, 4}) {
            // LDS per workgroup: the layout's real size when that many workgroups fit a CU, else the largest power of
            // two that does (the code range shrinks, the banking does not change)
            unsigned lds = full;
            while ((size_t)lds * wgs > (160u << 10)) lds >>= 1;
            const int grid = ncu * wgs;
            srand(1234 + mode);
            for (int b = 0; b < grid; ++b)
                for (int t = 0; t < 8; ++t)
                    for (int l = 0; l < 256; ++l) h[((size_t)b * 8 + t) * 256 + l] = address(mode, l & 63, t, (unsigned)(rand() & 255));
            hipMemcpy(addr, h.data(), (size_t)grid * 8 * 256 * 4, hipMemcpyHostToDevice);
            auto launch = [&]() {
                if (W == 4) { hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), lds, 0, addr, out, cyc, iters, lds); }
                else if (W == 8) { hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), lds, 0, addr, out, cyc, iters, lds); }
                else { hipFuncSetAttribute((const void *)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), lds, 0, addr, out, cyc, iters, lds); }
            };
            launch();
            hipDeviceSynchronize();
            launch();
            hipDeviceSynchronize();
            std::vector<unsigned long long> hc(grid);
            hipMemcpy(hc.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (int i = 0; i < grid; ++i) mean += (double)hc[i];
            mean /= grid;
            // lookups per clock and CU: every wave-instruction serves 64 lanes x (W / 4) query entries
            const double per_wg = (double)iters * 8 * 256 * QPL;
            const double rate = per_wg * wgs / mean;
            if (!json)
                printf("%-26s waves/SIMD=%d lds/WG=%6u  cycles/wave-inst=%6.2f  lane-lookups/clk/CU=%6.2f  query-lookups/clk/CU=%6.2f\n", mode_name[mode], wgs,
                       lds, mean / (iters * 8.0), rate / QPL, rate);
            else {
                printf("%s  \"%s @%d waves/SIMD\": {\"lane_lookups_per_clk_cu\": %.2f, \"query_lookups_per_clk_cu\": %.2f, \"cycles_per_wave_inst\": %.2f}",
                       first ? "" : ",\n", mode_name[mode], wgs, rate / QPL, rate, mean / (iters * 8.0));
                first = false;
            }
        }
    }
    if (json) printf("\n}\n");
    return 0;
}
