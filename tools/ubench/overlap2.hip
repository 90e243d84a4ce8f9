// Do v_mfma_f32_32x32x16_f16 and ordinary VALU instructions overlap on ONE SIMD of gfx950 -- with controlled placement.
//
// Round 1's overlap.hip split roles by wave parity inside a 512-thread workgroup; the waves of a workgroup are dealt over
// the four SIMDs in turn, so odd and even waves sat on DIFFERENT SIMDs and the test measured nothing about one SIMD.
// Here: one workgroup per CU (it declares 96 KiB of LDS, so a second cannot be resident), 256 threads = one wave per
// SIMD, 512 threads = two (waves w and w + 4 share a SIMD).  Every instruction is an `asm volatile`, so the stream is
// exactly what is written: per loop iteration 4 MFMAs (4 accumulators), each followed by K VALU instructions
// (12 independent chains).  Reported: shader cycles (s_memtime) per MFMA gap, median over all waves, and wall time.
//   mode 0  every wave runs the interleaved stream (K VALU per MFMA)
//   mode 1  two waves per SIMD, waves 0-3 MFMA only, waves 4-7 VALU only (K per MFMA of the partner): cross-wave overlap
//   mode 2  VALU only (K per "gap", no MFMA): the VALU issue cost alone
// kind 0: v_fma_f32   kind 1: one v_exp_f32 in every 4 VALU, rest v_fma_f32   kind 2: v_pk_fma_f32 (counted as ONE VALU)
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap2 overlap2.hip ; run: ./overlap2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(ka), "v"(kb))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pka), "v"(pkb))

template <int K, int KIND>
__device__ __forceinline__ void valu_group(float (&v)[12], f2 (&p)[6], int& slot, float ka, float kb, f2 pka, f2 pkb) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int s = (slot + i) % 12;
        if (KIND == 2) PKFMA(p[s % 6]);
        else if (KIND == 1 && (i & 3) == 3) EXP(v[s]);
        else FMA(v[s]);
    }
    slot = (slot + K) % 12;
}

template <int K, int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, long long* cyc) {
    extern __shared__ char lds_hold[];
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float v[12];
    f2 p[6];
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 6; ++i) p[i] = f2{v[2 * i], v[2 * i + 1]};
    const float ka = 0.999f, kb = 0.001f;
    const f2 pka = {0.999f, 0.999f}, pkb = {0.001f, 0.001f};
    const bool do_mfma = mode == 0 || (mode == 1 && wave < 4);
    const bool do_valu = mode == 0 || mode == 2 || (mode == 1 && wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
            int slot = 0;
            MFMA(c0); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
            MFMA(c1); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
            MFMA(c2); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
            MFMA(c3); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
            int slot = 0;
            valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
            valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb); valu_group<K, KIND>(v, p, slot, ka, kb, pka, pkb);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 6; ++i) s += p[i].x + p[i].y;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s + lds_hold[threadIdx.x];
}

template <int K, int KIND>
static void run(int mode, int threads, float* out, long long* cyc_d) {
    const int iters = 4000, blocks = 256;
    const size_t lds = 96 * 1024;
    static bool once = false;
    hipFuncSetAttribute((const void*)k<K, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)once;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, KIND>), dim3(blocks), dim3(threads), lds, 0, mode, 200, out, cyc_d);
    hipMemset(cyc_d, 0, blocks * 8 * sizeof(long long));
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, KIND>), dim3(blocks), dim3(threads), lds, 0, mode, iters, out, cyc_d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks * 8);
    hipMemcpy(c.data(), cyc_d, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> lo, hi;   // waves 0-3, waves 4-7
    for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < threads / 64; ++w) (w < 4 ? lo : hi).push_back(c[bI * 8 + w]);
    std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
    const double gaps = 4.0 * iters;
    const double mlo = lo[lo.size() / 2] / gaps, mhi = hi.empty() ? 0.0 : hi[hi.size() / 2] / gaps;
    printf("kind %d mode %d waves/SIMD %d K %2d : cycles/gap waves0-3 %7.2f  waves4-7 %7.2f   wall %.3f ms (%.1f ns/gap => %.2f GHz)\n", KIND, mode,
           threads / 256, K, mlo, mhi, ms, ms * 1e6 / gaps, (mhi > mlo ? mhi : mlo) / (ms * 1e6 / gaps));
}

template <int K, int KIND>
static void sweep(float* out, long long* cyc) {
    run<K, KIND>(0, 256, out, cyc);
    run<K, KIND>(0, 512, out, cyc);
    if (K > 0) { run<K, KIND>(1, 512, out, cyc); run<K, KIND>(2, 256, out, cyc); run<K, KIND>(2, 512, out, cyc); }
}

int main() {
    float* out; hipMalloc(&out, 4);
    long long* cyc; hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    sweep<0, 0>(out, cyc);
    sweep<1, 0>(out, cyc); sweep<2, 0>(out, cyc); sweep<3, 0>(out, cyc); sweep<4, 0>(out, cyc); sweep<5, 0>(out, cyc);
    sweep<6, 0>(out, cyc); sweep<8, 0>(out, cyc); sweep<10, 0>(out, cyc); sweep<12, 0>(out, cyc); sweep<16, 0>(out, cyc);
    sweep<4, 1>(out, cyc); sweep<8, 1>(out, cyc); sweep<12, 1>(out, cyc);
    sweep<4, 2>(out, cyc); sweep<8, 2>(out, cyc);
    return 0;
}
