// Micro-benchmark 2: the inner k-group pattern of the block kernels (weight fragment stream + activation VALU + 16 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float silu(float v) {
    const float t = v * -1.44269504f; const float e = fmaf(v, -1.9259630e-8f, fmaf(v, -1.44269504f, -t));
    const float p = __builtin_amdgcn_exp2f(t); return v * __builtin_amdgcn_rcpf(1.0f + fmaf(p * e, 0.693147f, p));
}
// NLOAD: W loads per group (4) + extra (gamma/beta) ; ACT: VALU activation ; LDSW: weights read from LDS instead of global
template <int EXTRA, bool ACT, bool SB, bool LDSW>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, const float* __restrict__ gb, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lw[16 * 4 * 256];  // 64 KiB: 16 groups x 4 tiles
    const int lane = threadIdx.x & 63, h = lane >> 5;
    if (LDSW) { for (int i = threadIdx.x; i < 16 * 4 * 256; i += 256) lw[i] = w[i]; __syncthreads(); }
    f32x16 acc[4], in;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 16; ++r) in[r] = lane * 0.01f + r;
    float4 wn[4], gm = {1, 1, 1, 1}, bt = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) wn[i] = LDSW ? *(float4*)&lw[(i * 64 + lane) * 4] : ld4(w + (i * 64 + lane) * 4);
    for (int it = 0; it < iters; ++it) {
        float4 wc[4];
        for (int i = 0; i < 4; ++i) wc[i] = wn[i];
        const float4 g0 = gm, b0v = bt;
        const int g = (it + 1) & 15;
        for (int i = 0; i < 4; ++i) wn[i] = LDSW ? *(float4*)&lw[((g * 4 + i) * 64 + lane) * 4] : ld4(w + ((g * 4 + i) * 64 + lane) * 4);
        if (EXTRA) { gm = ld4(gb + 8 * g + 4 * h); bt = ld4(gb + 512 + 8 * g + 4 * h); }
        if (SB) __builtin_amdgcn_sched_barrier(0);
        float b[4];
        const int q = (it & 3) * 4;
        for (int p = 0; p < 4; ++p) {
            float x = in[q + p];
            if (ACT) x = silu(fmaf((x - 0.5f) * 0.9f, (&g0.x)[p], (&b0v.x)[p]));
            b[p] = x;
        }
        for (int nt = 0; nt < 4; ++nt) { MF(acc[nt], wc[nt].x, b[0]); MF(acc[nt], wc[nt].y, b[1]); MF(acc[nt], wc[nt].z, b[2]); MF(acc[nt], wc[nt].w, b[3]); }
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int EXTRA, bool ACT, bool SB, bool LDSW>
void run(const char* name, const float* w, const float* gb, float* out) {
    const int iters = 1600;
    for (int blocks : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<EXTRA, ACT, SB, LDSW>), dim3(blocks), dim3(256), 0, 0, w, gb, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %.0f wave/SIMD: %.3f ms  %.1f cyc/MFMA/SIMD  %.1f TFLOP/s\n", name, blocks / 256.0, ms,
               ms * 1e-3 * 2.4e9 / (16.0 * iters * blocks / 256.0), blocks * 4.0 * 16 * iters * 4096.0 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float *w, *gb, *out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&gb, 1 << 16); hipMemset(gb, 0, 1 << 16); hipMalloc(&out, 4096 * 256 * 4);
    run<0, false, false, false>("W loads only", w, gb, out);
    run<1, false, false, false>("W + gamma/beta loads", w, gb, out);
    run<1, true, false, false>("W + g/b loads + act", w, gb, out);
    run<1, true, true, false>("W + g/b loads + act + schedbar", w, gb, out);
    run<0, true, true, false>("W loads + act + schedbar", w, gb, out);
    run<0, false, false, true>("W from LDS", w, gb, out);
    run<0, true, true, true>("W from LDS + act + schedbar", w, gb, out);
    return 0;
}
