// Kernels reading page-locked host memory in place, coherent (fine-grained, uncached on the GPU) against non-coherent (coarse-grained: cached in
// L2, visible at kernel boundaries): a 512 KB block read 16 times by one launch (what an exhaustive coarse kernel does with a 1024 x 128 query
// batch), and whether a launch ever sees the PREVIOUS contents after the host rewrote the block.  Build: hipcc --offload-arch=gfx950 -O3 host_nc.hip -o host_nc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every workgroup sums the whole block (n16 uint4) `reps` times, strided over reps so that all 16 passes are real reads
__global__ __launch_bounds__(256) void reread(const uint4 *__restrict__ src, size_t n16, int reps, unsigned long long *out)
{
    unsigned long long s = 0;
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)((blockIdx.x + r) % gridDim.x) * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
            const uint4 v = src[i];
            s += v.x + v.y + v.z + v.w;
        }
    atomicAdd(out, s);
}
static void poll(hipStream_t s) { while (hipStreamQuery(s) == hipErrorNotReady) { } }

int main()
{
    const size_t B = 1024 * 128 * 4, n16 = B / 16;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *dout;
    CK(hipMalloc(&dout, 8));
    void *coh, *nc, *dev;
    CK(hipHostMalloc(&coh, B, hipHostMallocDefault));
    CK(hipHostMalloc(&nc, B, hipHostMallocNonCoherent));
    CK(hipMalloc(&dev, B));
    void *regc = aligned_alloc(4096, B), *regn = aligned_alloc(4096, B);
    memset(regc, 1, B); memset(regn, 1, B);
    CK(hipHostRegister(regc, B, hipHostRegisterDefault));
    hipError_t er = hipHostRegister(regn, B, hipExtHostRegisterCoarseGrained);
    printf("hipHostRegister(coarse-grained): %s\n", hipGetErrorString(er));
    struct { const char *name; void *p; } bufs[] = {{"device memory", dev}, {"host, coherent (hipHostMallocDefault)", coh}, {"host, non-coherent (hipHostMallocNonCoherent)", nc},
                                                    {"registered, default", regc}, {"registered, coarse-grained", er == hipSuccess ? regn : nullptr}};
    for (auto &b : bufs) {
        if (!b.p) continue;
        if (b.p != dev) memset(b.p, 3, B); else CK(hipMemset(dev, 3, B));
        for (int reps : {1, 16}) {
            for (int w = 0; w < 5; ++w) { hipLaunchKernelGGL(reread, dim3(256), dim3(256), 0, s, (const uint4 *)b.p, n16, reps, dout); poll(s); }
            const double t0 = now();
            const int N = 100;
            for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(reread, dim3(256), dim3(256), 0, s, (const uint4 *)b.p, n16, reps, dout); poll(s); }
            printf("%-46s 512 KB read %2d x by one launch: %7.1f us\n", b.name, reps, (now() - t0) / N * 1e6);
        }
    }
    // staleness: host rewrites the block, a launch sums it; the sum must be that of the NEW contents, every time
    for (auto &b : bufs) {
        if (!b.p || b.p == dev) continue;
        int stale = 0;
        unsigned long long hs = 0;
        for (int it = 0; it < 300; ++it) {
            const unsigned v = 1u + (unsigned)it;
            unsigned *w = (unsigned *)b.p;
            for (size_t i = 0; i < B / 4; ++i) w[i] = v;
            CK(hipMemsetAsync(dout, 0, 8, s));
            hipLaunchKernelGGL(reread, dim3(256), dim3(256), 0, s, (const uint4 *)b.p, n16, 4, dout);
            CK(hipMemcpyAsync(&hs, dout, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (hs != (unsigned long long)v * (B / 4) * 4) ++stale;
        }
        printf("%-46s stale launches after a host rewrite: %d of 300\n", b.name, stale);
    }
    return 0;
}
