#include "kernels.hip.h"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
int main() {
    const int nq = 300, kc = 1000, d = 96;
    std::vector<float> Q(nq*d), C(kc*d), cn(kc), out(nq*kc);
    srand(1);
    for (auto& v : Q) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : C) v = (rand() / (float)RAND_MAX) * 3 - 1;   // asymmetric
    for (int c = 0; c < kc; ++c) { double s = 0; for (int i = 0; i < d; ++i) s += (double)C[c*d+i]*C[c*d+i]; cn[c] = (float)s; }
    float *dQ, *dC, *dn, *dout;
    hipMalloc(&dQ, Q.size()*4); hipMalloc(&dC, C.size()*4); hipMalloc(&dn, cn.size()*4); hipMalloc(&dout, out.size()*4);
    hipMemcpy(dQ, Q.data(), Q.size()*4, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size()*4, hipMemcpyHostToDevice);
    hipMemcpy(dn, cn.data(), cn.size()*4, hipMemcpyHostToDevice);
    dim3 grid((kc + 127)/128, (nq + 127)/128);
    hipLaunchKernelGGL(ivf::coarse_mfma_kernel, grid, dim3(256), 0, 0, dQ, dC, dn, dout, nq, kc, d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out.data(), dout, out.size()*4, hipMemcpyDeviceToHost);
    double maxerr = 0; int bad = 0;
    for (int q = 0; q < nq; ++q) for (int c = 0; c < kc; ++c) {
        double s = 0, qn = 0; for (int i = 0; i < d; ++i) { s += (double)Q[q*d+i]*C[c*d+i]; }
        double ref = (double)cn[c] - 2*s;
        double err = fabs(ref - out[q*kc+c]);
        if (err > maxerr) maxerr = err;
        if (err > 1e-3) ++bad;
    }
    printf("hip=%d maxerr=%g bad=%d (scores ~ %g)\n", (int)e, maxerr, bad, out[5]);
    return bad != 0;
}
