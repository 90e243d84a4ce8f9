// Micro-benchmark 3: 2x fp16 split on v_mfma_f32_32x32x16_f16 (3 MFMAs per tile per 16 k) + activation/split VALU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (acc), 0, 0, 0)
__device__ __forceinline__ float silu(float v) {
    const float p = __builtin_amdgcn_exp2f(v * -1.44269504f); return v * __builtin_amdgcn_rcpf(1.0f + p);
}
template <bool ACT, int NT>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ w, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NT], in;
    for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 16; ++r) in[r] = lane * 0.01f + r;
    uint4 wn[NT][2];
    for (int i = 0; i < NT; ++i) for (int pl = 0; pl < 2; ++pl) wn[i][pl] = w[(i * 2 + pl) * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        uint4 wc[NT][2];
        for (int i = 0; i < NT; ++i) for (int pl = 0; pl < 2; ++pl) wc[i][pl] = wn[i][pl];
        const int g = (it + 1) & 15;
        for (int i = 0; i < NT; ++i) for (int pl = 0; pl < 2; ++pl) wn[i][pl] = w[((g * NT + i) * 2 + pl) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        h8 b1, b2;
        const int q = (it & 1) * 8;
        for (int p = 0; p < 8; p += 2) {
            float x0 = in[q + p], x1 = in[q + p + 1];
            if (ACT) { x0 = silu(fmaf((x0 - 0.5f) * 0.9f, 1.01f, 0.1f)) * 256.f; x1 = silu(fmaf((x1 - 0.5f) * 0.9f, 1.01f, 0.1f)) * 256.f; }
            h2 hi = __builtin_amdgcn_cvt_pkrtz(x0, x1);
            const float r0 = x0 - (float)hi[0], r1 = x1 - (float)hi[1];
            h2 lo = __builtin_amdgcn_cvt_pkrtz(r0, r1);
            b1[p] = (_Float16)hi[0]; b1[p + 1] = (_Float16)hi[1]; b2[p] = (_Float16)lo[0]; b2[p + 1] = (_Float16)lo[1];
        }
        for (int nt = 0; nt < NT; ++nt) {
            const h8 w1 = __builtin_bit_cast(h8, wc[nt][0]), w2 = __builtin_bit_cast(h8, wc[nt][1]);
            MF(acc[nt], w1, b1); MF(acc[nt], w1, b2); MF(acc[nt], w2, b1);
        }
    }
    float s = 0; for (int i = 0; i < NT; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool ACT, int NT>
void run(const char* name, const uint4* w, float* out) {
    const int iters = 3200;
    for (int blocks : {256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<ACT, NT>), dim3(blocks), dim3(256), 0, 0, w, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // one iteration = 16 k x 32 out x NT tiles x 32 rows of fp32-equivalent MACs
        printf("%-22s NT=%d %.0f wave/SIMD: %.3f ms  %.0f cyc per k16-step per SIMD-wave-slot  %.1f fp32-equiv TFLOP/s\n", name, NT, blocks / 256.0, ms,
               ms * 1e-3 * 2.4e9 / (iters * (blocks / 256.0)), blocks * 4.0 * iters * NT * 2.0 * 16 * 32 * 32 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    uint4* w; float* out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&out, 4096 * 256 * 4);
    run<false, 4>("split only", w, out);
    run<true, 4>("act + split", w, out);
    run<true, 2>("act + split", w, out);
    run<true, 1>("act + split", w, out);
    return 0;
}
