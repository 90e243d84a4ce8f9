// Do MFMA (v_mfma_f32_32x32x16_f16) and ordinary VALU instructions of DIFFERENT waves on one SIMD overlap?
//   mode 0: every wave issues MFMAs only;  mode 1: every wave VALU only;  mode 2: odd waves MFMA, even waves VALU;
//   mode 3: every wave alternates 1 MFMA / 8 VALU (same totals per SIMD as mode 2);
//   mode 4: VALU only, the same 64 FMAs per wave-iteration as 32 v_pk_fma_f32 (is the packed form full rate?).
// 8 waves per CU (2 per SIMD), 256 CUs.  Build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1)) || mode == 3;
    const bool do_valu = mode == 1 || (mode == 2 && !(wave & 1)) || mode == 3;
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7};
    const f2 ka = {1.0001f, 1.0001f}, kb = {0.5f, 0.5f};
    for (int it = 0; it < iters; ++it) {
        if (mode == 4) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                p0 = __builtin_elementwise_fma(p0, ka, kb); p1 = __builtin_elementwise_fma(p1, ka, kb);
                p2 = __builtin_elementwise_fma(p2, ka, kb); p3 = __builtin_elementwise_fma(p3, ka, kb);
            }
        } else if (mode == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {   // half the work of each kind per wave: same totals per SIMD as mode 2
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
                }
            }
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                    v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* out; hipMalloc(&out, 4);
    const int iters = 20000;
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<256, 512>>>(mode, 100, out);
        hipEventRecord(e0);
        k<<<256, 512>>>(mode, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: mode 0: 2 waves x 8 MFMA/iter; mode 1: 2 waves x 64 VALU/iter; mode 2: 8 MFMA + 64 VALU; mode 3: 2 x (4 MFMA + 32 VALU)
        printf("mode %d: %.3f ms  (%.1f ns per iteration)\n", mode, ms, ms * 1e6 / iters);
    }
    return 0;
}
