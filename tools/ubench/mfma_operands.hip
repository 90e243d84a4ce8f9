// Does the issue rate of v_mfma_f32_32x32x16_f16 depend on WHICH registers feed it?  One wave per SIMD (and two), 48 MFMAs per
// iteration over 4 accumulators, term-major like the M phase of dsg_panel.hpp:
//   mode 0  every MFMA reads the same A and B registers                    (tools/ubench/overlap2.hip's stream: 32 cycles)
//   mode 1  A rotates over 8 register quads, B fixed
//   mode 2  A rotates over 8 quads, B over 8 quads                          (the M phase's operand pattern)
//   mode 3  as 2, but accumulators in AGPRs ("+a")
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_operands mfma_operands.hip ; run: ./mfma_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF(c, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MFA(c, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float* out, long long* cyc) {
    extern __shared__ char lds_hold[];
    const int wave = threadIdx.x >> 6;
    h8 a[8], b[8];
    for (int q = 0; q < 8; ++q)
        for (int i = 0; i < 8; ++i) { a[q][i] = (_Float16)((threadIdx.x & 63) * 0.01f + i + q); b[q][i] = (_Float16)(i * 0.25f - 1.f + 0.1f * q); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) { MF(c0, a[0], b[0]); MF(c1, a[0], b[0]); MF(c2, a[0], b[0]); MF(c3, a[0], b[0]); }
            } else if (MODE == 1) {
                MF(c0, a[0], b[0]); MF(c1, a[1], b[0]); MF(c2, a[2], b[0]); MF(c3, a[3], b[0]);
                MF(c0, a[0], b[0]); MF(c1, a[1], b[0]); MF(c2, a[2], b[0]); MF(c3, a[3], b[0]);
                MF(c0, a[4], b[0]); MF(c1, a[5], b[0]); MF(c2, a[6], b[0]); MF(c3, a[7], b[0]);
            } else if (MODE == 2) {
                MF(c0, a[0], b[s]); MF(c1, a[1], b[s]); MF(c2, a[2], b[s]); MF(c3, a[3], b[s]);
                MF(c0, a[0], b[4 + s]); MF(c1, a[1], b[4 + s]); MF(c2, a[2], b[4 + s]); MF(c3, a[3], b[4 + s]);
                MF(c0, a[4], b[s]); MF(c1, a[5], b[s]); MF(c2, a[6], b[s]); MF(c3, a[7], b[s]);
            } else {
                MFA(c0, a[0], b[s]); MFA(c1, a[1], b[s]); MFA(c2, a[2], b[s]); MFA(c3, a[3], b[s]);
                MFA(c0, a[0], b[4 + s]); MFA(c1, a[1], b[4 + s]); MFA(c2, a[2], b[4 + s]); MFA(c3, a[3], b[4 + s]);
                MFA(c0, a[4], b[s]); MFA(c1, a[5], b[s]); MFA(c2, a[6], b[s]); MFA(c3, a[7], b[s]);
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float sres = 0.f;
    for (int i = 0; i < 16; ++i) sres += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

template <int MODE>
void run(int threads, const char* what) {
    const int blocks = 256, iters = 200;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 512 * sizeof(float)); hipMalloc(&cyc, blocks * 8 * sizeof(long long));
    hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 96 * 1024, 0, iters, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> v;
    for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < threads / 64; ++w) v.push_back(h[bI * 8 + w]);
    std::sort(v.begin(), v.end());
    const double per = (double)v[v.size() / 2] / (iters * 48.0);
    printf("%-40s waves/SIMD %d: %.1f cycles per MFMA per wave (%.1f per SIMD)\n", what, threads / 256, per, per / (threads / 256));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>(threads, "same A, same B");
        run<1>(threads, "A rotates, B fixed");
        run<2>(threads, "A and B rotate (M phase pattern)");
        run<3>(threads, "as above, accumulators in AGPRs");
    }
    return 0;
}
