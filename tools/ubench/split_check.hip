// split_pair (dsg_split.hpp) against the plain formulation on random and extreme values:  hipcc --offload-arch=gfx950 -O3 -I diffsg_amd/csrc -o split_check split_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "dsg_split.hpp"
using namespace dsg;
__global__ void k(const float* x, unsigned* hi, unsigned* lo, unsigned* hr, unsigned* lr, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float v0 = x[2 * i], v1 = x[2 * i + 1];
    unsigned a, b;
    split_pair(v0, v1, a, b);
    hi[i] = a; lo[i] = b;
    const hp2 p = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    const float r0 = v0 - (float)p[0], r1 = v1 - (float)p[1];
    const _Float16 q0 = (_Float16)r0, q1 = (_Float16)r1;          // round to nearest even
    unsigned short s0 = __builtin_bit_cast(unsigned short, q0), s1 = __builtin_bit_cast(unsigned short, q1);
    hr[i] = __builtin_bit_cast(unsigned, p); lr[i] = (unsigned)s0 | ((unsigned)s1 << 16);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int m = i % 8;
        const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
        x[i] = m == 0 ? u : m == 1 ? u * 1e-3f : m == 2 ? u * 1e3f : m == 3 ? u * 6e4f : m == 4 ? u * 1e-6f : m == 5 ? u * 16.f : m == 6 ? ldexpf(u, (rand() % 40) - 24) : 0.f;
    }
    float* dx; unsigned *a, *b, *c, *d;
    hipMalloc(&dx, n * 4); hipMalloc(&a, n * 2); hipMalloc(&b, n * 2); hipMalloc(&c, n * 2); hipMalloc(&d, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, a, b, c, d, n);
    std::vector<unsigned> ha(n / 2), hb(n / 2), hc(n / 2), hd(n / 2);
    hipMemcpy(ha.data(), a, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 2, hipMemcpyDeviceToHost);
    long bad_hi = 0, bad_lo = 0; int shown = 0;
    for (int i = 0; i < n / 2; ++i) {
        if (ha[i] != hc[i]) ++bad_hi;
        if (hb[i] != hd[i]) { ++bad_lo; if (shown++ < 10) printf("x = %g %g  hi %08x  lo %08x  ref lo %08x\n", x[2 * i], x[2 * i + 1], ha[i], hb[i], hd[i]); }
    }
    printf("pairs %d  hi mismatches %ld  lo mismatches %ld\n", n / 2, bad_hi, bad_lo);
    return 0;
}
