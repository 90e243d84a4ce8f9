// Which instructions of a partner wave's operand preparation keep the MFMA stream of the other wave on the SIMD from issuing?
// One workgroup per CU (96 KiB LDS), 512 threads: waves 0-3 issue only v_mfma_f32_32x32x16_f16 (4 accumulators), waves 4-7 (their
// SIMD partners) loop over ONE kind of instruction.  Reported: cycles per MFMA of the MFMA waves (32 = unimpeded) and cycles per
// instruction of the partner.  swap = 1: the MFMA stream runs on waves 4-7 (the younger waves) instead.
//   kind 0 v_fma_f32   1 v_exp_f32   2 v_rcp_f32   3 v_cvt_pkrtz_f16_f32   4 v_cvt_f32_f16   5 v_cvt_f32_f16 sdwa WORD_1
//   kind 6 v_sub_f32   7 ds_read_b128 (+ wait every 8)   8 s_nop 0   9 v_pk_fma_f32   10 v_mov_b32   11 s_mov_b32 (SALU)
// Build: hipcc --offload-arch=gfx950 -O3 -o overlap3 overlap3.hip ; run: ./overlap3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int KIND>
__device__ __forceinline__ void partner_block(float (&v)[12], f2 (&p)[6], unsigned lds_addr, uint4 (&q)[4]) {
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        const int s = i % 12;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[s]) : "v"(0.999f), "v"(0.001f));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[s]));
        if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[s]));
        if (KIND == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(v[s]) : "v"(v[(s + 1) % 12]), "v"(v[(s + 2) % 12]));
        if (KIND == 4) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[s]) : "v"(v[(s + 1) % 12]));
        if (KIND == 5) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[s]) : "v"(v[(s + 1) % 12]));
        if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[s]) : "v"(v[(s + 1) % 12]));
        if (KIND == 7) { asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(lds_addr)); if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (KIND == 8) asm volatile("s_nop 0");
        if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[s % 6]) : "v"(f2{0.999f, 0.999f}), "v"(f2{0.001f, 0.001f}));
        if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(v[s]) : "v"(v[(s + 1) % 12]));
        if (KIND == 11) { int t; asm volatile("s_mov_b32 %0, 5" : "=s"(t)); }
    }
}

template <int KIND>
__global__ __launch_bounds__(512) void k(int swap, int iters, float* out, long long* cyc) {
    extern __shared__ uint4 lds[];
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float v[12];
    f2 p[6];
    uint4 q[4] = {};
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + i) + 1.0f;
    for (int i = 0; i < 6; ++i) p[i] = f2{v[2 * i], v[2 * i + 1]};
    lds[threadIdx.x] = make_uint4(1, 2, 3, 4);
    const unsigned lds_addr = (unsigned)(threadIdx.x & 63) * 16u;
    const bool mf = swap ? wave >= 4 : wave < 4;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); }
        }
    } else {
        for (int it = 0; it < 4 * iters; ++it) partner_block<KIND>(v, p, lds_addr, q);     // runs longer than the MFMA waves
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 6; ++i) s += p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) s += (float)q[i].x;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* what) {
    const int blocks = 256, iters = 100;
    float* out; long long* cyc;
    (void)hipMalloc(&out, blocks * 512 * sizeof(float)); (void)hipMalloc(&cyc, blocks * 8 * sizeof(long long));
    (void)hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int swap = 0; swap < 2; ++swap) {
        (void)hipMemset(cyc, 0, blocks * 8 * sizeof(long long));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 96 * 1024, 0, swap, iters, out, cyc);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks * 8);
        (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<long long> m, o;
        for (int bI = 0; bI < blocks; ++bI)
            for (int w = 0; w < 8; ++w) ((swap ? w >= 4 : w < 4) ? m : o).push_back(h[bI * 8 + w]);
        std::sort(m.begin(), m.end()); std::sort(o.begin(), o.end());
        printf("%-28s MFMA on %s waves: %6.1f cycles per MFMA | partner %5.1f cycles per instruction\n", what, swap ? "younger" : "older  ",
               (double)m[m.size() / 2] / (iters * 48.0), (double)o[o.size() / 2] / (4.0 * iters * 48.0));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32"); run<1>("v_exp_f32"); run<2>("v_rcp_f32"); run<3>("v_cvt_pkrtz_f16_f32"); run<4>("v_cvt_f32_f16");
    run<5>("v_cvt_f32_f16 sdwa"); run<6>("v_sub_f32"); run<7>("ds_read_b128"); run<8>("s_nop 0"); run<9>("v_pk_fma_f32");
    run<10>("v_mov_b32"); run<11>("s_mov_b32");
    return 0;
}
