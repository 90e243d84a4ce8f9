// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 under the issue patterns of the block kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float b0 = lane * 0.001f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    float4 wv[4];
    for (int i = 0; i < 4; ++i) wv[i] = *(const float4*)(w + (i * 64 + lane) * 4);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 4 dependent per accumulator, registers only
            for (int nt = 0; nt < 4; ++nt) { MF(acc[nt], wv[nt].x, b0); MF(acc[nt], wv[nt].y, b1); MF(acc[nt], wv[nt].z, b2); MF(acc[nt], wv[nt].w, b3); }
        } else if (MODE == 1) {  // rotate
            for (int nt = 0; nt < 4; ++nt) MF(acc[nt], wv[nt].x, b0);
            for (int nt = 0; nt < 4; ++nt) MF(acc[nt], wv[nt].y, b1);
            for (int nt = 0; nt < 4; ++nt) MF(acc[nt], wv[nt].z, b2);
            for (int nt = 0; nt < 4; ++nt) MF(acc[nt], wv[nt].w, b3);
        } else if (MODE == 2) {  // with streamed weight loads (prefetch distance 1), L2-resident 64 KiB
            float4 wn[4];
            for (int i = 0; i < 4; ++i) wn[i] = *(const float4*)(w + ((((it + 1) & 15) * 4 + i) * 64 + lane) * 4);
            for (int nt = 0; nt < 4; ++nt) { MF(acc[nt], wv[nt].x, b0); MF(acc[nt], wv[nt].y, b1); MF(acc[nt], wv[nt].z, b2); MF(acc[nt], wv[nt].w, b3); }
            for (int i = 0; i < 4; ++i) wv[i] = wn[i];
        } else if (MODE == 3) {  // single accumulator, fully dependent
            for (int nt = 0; nt < 4; ++nt) { MF(acc[0], wv[nt].x, b0); MF(acc[0], wv[nt].y, b1); MF(acc[0], wv[nt].z, b2); MF(acc[0], wv[nt].w, b3); }
        }
        b0 += 1e-9f;
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *w, *out; long long* cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int blocks : {256, 512, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, w, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, w, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, w, out, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, w, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double nm = 16.0 * iters;
            printf("blocks=%4d (%.1f waves/SIMD) mode=%d: %.3f ms, clock64 %.1f ticks/MFMA, wall %.1f ns/MFMA/wave -> %.1f TFLOP/s\n", blocks,
                   blocks / 256.0, mode, ms, c / nm, ms * 1e6 / nm, blocks * 4.0 * nm * 2 * 2048 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
