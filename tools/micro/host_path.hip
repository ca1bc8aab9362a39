// Host <-> device primitives of the host-pointer search path, timed one by one (gfx950 box): what a batch of 1024 x 128 f32 queries
// (512 KB in) and its packed results (84 KB out) cost by each route.  Build: hipcc --offload-arch=gfx950 -O3 host_path.hip -o host_path -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <sys/mman.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void spin_kernel(long long cycles, int *flag_host)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (flag_host && threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(flag_host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

template <class F> static double timeit(F f, int reps = 200, int warm = 20)
{
    for (int i = 0; i < warm; ++i) f();
    const double t0 = now();
    for (int i = 0; i < reps; ++i) f();
    return (now() - t0) / reps * 1e6;
}

static void poll(hipStream_t s) { while (hipStreamQuery(s) == hipErrorNotReady) { } }

int main()
{
    const size_t QB = 1024 * 128 * 4, OB = 1024 * 84;
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *dq, *dout, *pin, *pout, *pinwc;
    CK(hipMalloc(&dq, 16 * QB)); CK(hipMalloc(&dout, 16 * OB));
    CK(hipHostMalloc(&pin, 16 * QB, hipHostMallocDefault));
    CK(hipHostMalloc(&pout, 16 * OB, hipHostMallocDefault));
    CK(hipHostMalloc(&pinwc, 16 * QB, hipHostMallocWriteCombined));
    // pageable source, 16 batches (8 MB: larger than L2 of a core, inside L3)
    std::vector<char> page(16 * QB, 1), pageout(16 * OB, 0);
    void *reg = mmap(nullptr, 16 * QB, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(reg, 2, 16 * QB);
    void *regout = mmap(nullptr, 16 * OB, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(regout, 0, 16 * OB);
    double t0 = now();
    CK(hipHostRegister(reg, 16 * QB, hipHostRegisterDefault));
    printf("hipHostRegister(8 MB)                      %8.1f us\n", (now() - t0) * 1e6);
    t0 = now();
    CK(hipHostRegister(regout, 16 * OB, hipHostRegisterDefault));
    printf("hipHostRegister(1.3 MB)                    %8.1f us\n", (now() - t0) * 1e6);
    {
        void *tmp = mmap(nullptr, QB, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        memset(tmp, 1, QB);
        t0 = now(); CK(hipHostRegister(tmp, QB, hipHostRegisterDefault)); double a = (now() - t0) * 1e6;
        t0 = now(); CK(hipHostUnregister(tmp)); double b = (now() - t0) * 1e6;
        printf("hipHostRegister / Unregister(512 KB)       %8.1f / %.1f us\n", a, b);
        hipPointerAttribute_t at;
        t0 = now(); for (int i = 0; i < 1000; ++i) (void)hipPointerGetAttributes(&at, pin);
        printf("hipPointerGetAttributes(pinned)            %8.2f us\n", (now() - t0) * 1e3);
        t0 = now(); for (int i = 0; i < 1000; ++i) (void)hipPointerGetAttributes(&at, page.data());
        printf("hipPointerGetAttributes(pageable)          %8.2f us\n", (now() - t0) * 1e3);
        (void)hipGetLastError();
    }
    int b = 0;
    auto nb = [&]() { b = (b + 1) & 15; return (size_t)b; };
    printf("memcpy 512 KB pageable -> pinned           %8.1f us\n", timeit([&] { size_t o = nb() * QB; memcpy((char *)pin + o, page.data() + o, QB); }));
    printf("memcpy 512 KB pageable -> pinned WC        %8.1f us\n", timeit([&] { size_t o = nb() * QB; memcpy((char *)pinwc + o, page.data() + o, QB); }));
    printf("memcpy 84 KB pinned -> pageable            %8.1f us\n", timeit([&] { size_t o = nb() * OB; memcpy(pageout.data() + o, (char *)pout + o, OB); }));
    printf("H2D 512 KB pinned, async + poll            %8.1f us\n", timeit([&] { size_t o = nb() * QB; CK(hipMemcpyAsync((char *)dq + o, (char *)pin + o, QB, hipMemcpyHostToDevice, s)); poll(s); }));
    printf("H2D 512 KB registered, async + poll        %8.1f us\n", timeit([&] { size_t o = nb() * QB; CK(hipMemcpyAsync((char *)dq + o, (char *)reg + o, QB, hipMemcpyHostToDevice, s)); poll(s); }));
    printf("H2D 512 KB pageable, async + poll          %8.1f us\n", timeit([&] { size_t o = nb() * QB; CK(hipMemcpyAsync((char *)dq + o, page.data() + o, QB, hipMemcpyHostToDevice, s)); poll(s); }));
    printf("H2D 512 KB by kernel from pinned (64 wg)   %8.1f us\n", timeit([&] { size_t o = nb() * QB; hipLaunchKernelGGL(copy16, dim3(64), dim3(256), 0, s, (const uint4 *)((char *)pin + o), (uint4 *)((char *)dq + o), QB / 16); poll(s); }));
    printf("H2D 512 KB by kernel from pinned (256 wg)  %8.1f us\n", timeit([&] { size_t o = nb() * QB; hipLaunchKernelGGL(copy16, dim3(256), dim3(128), 0, s, (const uint4 *)((char *)pin + o), (uint4 *)((char *)dq + o), QB / 16); poll(s); }));
    printf("H2D 512 KB by kernel from registered       %8.1f us\n", timeit([&] { size_t o = nb() * QB; hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const uint4 *)((char *)reg + o), (uint4 *)((char *)dq + o), QB / 16); poll(s); }));
    printf("D2D 512 KB by kernel (launch floor)        %8.1f us\n", timeit([&] { size_t o = nb() * QB; hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const uint4 *)((char *)dq + o), (uint4 *)((char *)dq + ((b + 1) & 15) * QB), QB / 16); poll(s); }));
    printf("D2H 84 KB pinned, async + poll             %8.1f us\n", timeit([&] { size_t o = nb() * OB; CK(hipMemcpyAsync((char *)pout + o, (char *)dout + o, OB, hipMemcpyDeviceToHost, s)); poll(s); }));
    printf("D2H 84 KB registered, async + poll         %8.1f us\n", timeit([&] { size_t o = nb() * OB; CK(hipMemcpyAsync((char *)regout + o, (char *)dout + o, OB, hipMemcpyDeviceToHost, s)); poll(s); }));
    printf("D2H 84 KB pageable, async + poll           %8.1f us\n", timeit([&] { size_t o = nb() * OB; CK(hipMemcpyAsync(pageout.data() + o, (char *)dout + o, OB, hipMemcpyDeviceToHost, s)); poll(s); }));
    printf("D2H 84 KB by kernel to pinned              %8.1f us\n", timeit([&] { size_t o = nb() * OB; hipLaunchKernelGGL(copy16, dim3(21), dim3(256), 0, s, (const uint4 *)((char *)dout + o), (uint4 *)((char *)pout + o), OB / 16); poll(s); }));
    printf("D2H 84 KB by kernel to registered          %8.1f us\n", timeit([&] { size_t o = nb() * OB; hipLaunchKernelGGL(copy16, dim3(21), dim3(256), 0, s, (const uint4 *)((char *)dout + o), (uint4 *)((char *)regout + o), OB / 16); poll(s); }));
    // the chain of a blocking call: H2D, a 30 us kernel, D2H; by copies, by kernels; completion by stream poll vs a host flag
    int *flag; CK(hipHostMalloc(&flag, 64, hipHostMallocDefault)); *flag = 0;
    const long long cyc30 = 3000;   // wall_clock64 ticks at 100 MHz: 30 us
    printf("chain: H2D copy, 30 us kernel, D2H copy, poll          %8.1f us\n", timeit([&] {
        size_t o = nb() * QB, oo = (size_t)b * OB;
        CK(hipMemcpyAsync((char *)dq + o, (char *)pin + o, QB, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc30, (int *)nullptr);
        CK(hipMemcpyAsync((char *)pout + oo, (char *)dout + oo, OB, hipMemcpyDeviceToHost, s)); poll(s); }));
    printf("chain: H2D kernel, 30 us kernel, D2H kernel, poll      %8.1f us\n", timeit([&] {
        size_t o = nb() * QB, oo = (size_t)b * OB;
        hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const uint4 *)((char *)pin + o), (uint4 *)((char *)dq + o), QB / 16);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc30, (int *)nullptr);
        hipLaunchKernelGGL(copy16, dim3(21), dim3(256), 0, s, (const uint4 *)((char *)dout + oo), (uint4 *)((char *)pout + oo), OB / 16); poll(s); }));
    printf("chain: H2D kernel, 30 us kernel w/ host flag, spin     %8.1f us\n", timeit([&] {
        size_t o = nb() * QB;
        hipLaunchKernelGGL(copy16, dim3(128), dim3(256), 0, s, (const uint4 *)((char *)pin + o), (uint4 *)((char *)dq + o), QB / 16);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc30, flag);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == 0) { }
        *flag = 0; }));
    printf("30 us kernel alone, poll                               %8.1f us\n", timeit([&] { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc30, (int *)nullptr); poll(s); }));
    printf("30 us kernel alone, hipStreamSynchronize               %8.1f us\n", timeit([&] { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc30, (int *)nullptr); CK(hipStreamSynchronize(s)); }));
    return 0;
}
