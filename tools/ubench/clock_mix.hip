// Which shader clock does the chip hold under which instruction mix?  (round 5: the panel kernel and the bare f16 MFMA probe both run at
// 1.72 GHz, the exact-float32 block kernel at 2.44 GHz -- is the lower clock a property of ANY use of the f16 matrix pipe, or of its duty?)
// Every SIMD runs W waves of: [one MFMA of kind MF] + NV dependent-free v_fma_f32 per slot; the clock = shader cycles (s_memtime) over the
// constant 100 MHz counter (s_memrealtime) across the whole loop, wave 0 of block 0; also the wall time per slot.
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_mix clock_mix.hip ; run: ./clock_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MF: 0 none, 1 v_mfma_f32_32x32x16_f16, 2 v_mfma_f32_32x32x2_f32, 3 v_mfma_f32_16x16x32_f16 ; EVERY: one MFMA per EVERY slots
template <int MF, int NV, int EVERY>
__global__ __launch_bounds__(512) void k_clk(int iters, float* out, unsigned long long* clk) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c[2] = {{0}, {0}};
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 d4[2] = {{0}, {0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i) + 1.0f;
    const float k1 = 0.999f, k0 = 0.001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s % EVERY == 0) {
                if (MF == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[(s / EVERY) & 1]) : "v"(a), "v"(b));
                else if (MF == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[(s / EVERY) & 1]) : "v"(k1), "v"(k0));
                else if (MF == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d4[(s / EVERY) & 1]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(s * NV + i) % 8]) : "v"(k1), "v"(k0));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c[0][i] + c[1][i];
    for (int i = 0; i < 4; ++i) acc += d4[0][i] + d4[1][i];
    for (int i = 0; i < 8; ++i) acc += v[i];
    if (acc == 123.456f) out[threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MF, int NV, int EVERY>
void run(const char* tag, int iters) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_clk<MF, NV, EVERY>), dim3(256), dim3(512), 0, 0, iters, out, clk);     // 8 waves per CU: two per SIMD
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);          // 100 MHz counter: 10 ns per tick
    printf("%-52s %7.2f ms   shader clock %.3f GHz   %.1f cycles per slot (two waves per SIMD)\n", tag, ms, ghz, (double)h[0] / ((double)iters * 16) / 2);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run<0, 6, 1>("v_fma_f32 only (6 per slot)", 40000);
    run<0, 6, 1>("v_fma_f32 only (6 per slot), again", 40000);
    run<1, 0, 1>("f16 32x32x16 MFMA only", 40000);
    run<1, 6, 1>("f16 MFMA + 6 v_fma per slot", 30000);
    run<1, 12, 1>("f16 MFMA + 12 v_fma per slot", 20000);
    run<1, 12, 2>("f16 MFMA every 2nd slot, 12 v_fma per slot", 20000);
    run<1, 12, 4>("f16 MFMA every 4th slot, 12 v_fma per slot", 20000);
    run<1, 12, 8>("f16 MFMA every 8th slot, 12 v_fma per slot", 20000);
    run<1, 12, 16>("f16 MFMA every 16th slot, 12 v_fma per slot", 20000);
    run<3, 12, 4>("f16 16x16x32 MFMA every 4th slot, 12 v_fma", 20000);
    run<2, 0, 1>("f32 32x32x2 MFMA only", 20000);
    run<2, 6, 1>("f32 MFMA + 6 v_fma per slot", 20000);
    run<0, 12, 1>("v_fma_f32 only (12 per slot)", 20000);
    return 0;
}
