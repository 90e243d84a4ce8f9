// The narrow ResidualBlock in a 16-ROW HALF-TILE layout on v_mfma_f32_16x16x32_f16 (round 6, design experiment -- not in the library):
// a wave carries TWO independent 16-row half-tiles; lane l = 16 q + j holds, per half-tile, features 16 t + 4 q + r (r = 0..3) of row j in
// accumulator register r of out tile t -- the D layout of the 16x16x32 MFMA -- and feeds them back as its k-slots 8 q .. 8 q + 7 of the next
// product (slots e < 4: tile 0's registers, e >= 4: tile 1's, or zero for a 16-wide tensor; the packed weights use the same slot map), so a
// chain of Linears still never moves data across lanes.  Per stage and half-tile a lane transforms N / 4 values (the 32-row layout: N / 2),
// the row statistics take two cross-lane steps (v_permlane16_swap, v_permlane32_swap), and the two half-tiles of a wave are independent
// chains the scheduler can interleave.  This file measures ONE down block (N = 16 or 32, identity shortcut) against tools/ubench/narrow_block.hip:
// same work per row (3 x LayerNorm + SiLU + split + 3-term f16 MFMA + un-scale, condition term, residual, statistics), LDS-resident
// planes and vectors, W waves per SIMD.  Timing only: the operands are synthetic.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQ_N=16 -DQ_WAVES=16 -o narrow_q16 tools/ubench/narrow_q16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#ifndef Q_N
#define Q_N 16
#endif
#ifndef Q_WAVES
#define Q_WAVES 16
#endif
#ifndef Q_HT
#define Q_HT 2            // half-tiles per wave
#endif
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = Q_N, NTQ = N / 16;          // out tiles of 16 features
constexpr float kEps = 1e-5f, kAct = 16.0f;

__device__ __forceinline__ float red4(float v) {            // sum over the four lanes (q = 0..3) that share a row
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float w = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 0"
        : "=&v"(l) : "v"(v0), "v"(v1), "v"(hi));
    lo = l;
}
// one stage for HT half-tiles: in x[ht][NTQ] -> out acc[ht][NTQ]; planes: per out tile (hi, lo) 64 lanes x 8 halfs; vectors per lane feature quad
template <int HT>
__device__ __forceinline__ void stage(f32x4 (&acc)[HT][NTQ], const f32x4 (&x)[HT][NTQ], const uint4* __restrict__ planes, const float* __restrict__ gam,
                                      const float* __restrict__ bet, const float* __restrict__ bias, const float inv, const int lane) {
    const int q = lane >> 4;
    uint4 whi[NTQ], wlo[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) { whi[t] = planes[(2 * t) * 64 + lane]; wlo[t] = planes[(2 * t + 1) * 64 + lane]; }
    f32x4 g[NTQ], b[NTQ], bs[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
        g[t] = *reinterpret_cast<const f32x4*>(gam + 16 * t + 4 * q); b[t] = *reinterpret_cast<const f32x4*>(bet + 16 * t + 4 * q);
        bs[t] = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * q);
    }
    h8 bhi[HT], blo[HT];
#pragma unroll
    for (int s = 0; s < HT; ++s) {
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) sm += (x[s][t][0] + x[s][t][1]) + (x[s][t][2] + x[s][t][3]);
        const float mean = red4(sm) * (1.0f / N);
        float qq = 0.f;
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x[s][t][r] - mean; qq = fmaf(d, d, qq); }
        const float rstd = __builtin_amdgcn_rsqf(red4(qq) * (1.0f / N) + kEps), c = rstd, d0 = -mean * rstd;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = fmaf(fmaf(x[s][t][r], c, d0), g[t][r], b[t][r]);
                const float p = __builtin_amdgcn_exp2f(u * -1.44269504088896341f);
                v[4 * t + r] = u * __builtin_amdgcn_rcpf(fmaf(p, 1.0f / kAct, 1.0f / kAct));
            }
        unsigned hh[4] = {0u, 0u, 0u, 0u}, ll[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 2 * NTQ; ++k) split_pair(v[2 * k], v[2 * k + 1], hh[k], ll[k]);
        const uint4 uh = {hh[0], hh[1], hh[2], hh[3]}, ul = {ll[0], ll[1], ll[2], ll[3]};
        bhi[s] = __builtin_bit_cast(h8, uh); blo[s] = __builtin_bit_cast(h8, ul);
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // term-major over (half-tile, out tile): consecutive MFMAs never share an accumulator
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, whi[t]), bhi[s], z, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, whi[t]), blo[s], acc[s][t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wlo[t]), bhi[s], acc[s][t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][t][r] = fmaf(acc[s][t][r], inv, bs[t][r]);
}

template <int HT>
__global__ __launch_bounds__(1024, 4) void k_q16(const uint4* __restrict__ image, int n_u4, const float* __restrict__ cond, int reps, int nht, long long* cyc,
                                                  float* sink, size_t cold_stride) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < n_u4; i += blockDim.x) lds[i] = image[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const int wave_g = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int ht0 = wave_g * HT;
    if (ht0 >= nht) return;
    const float* const vec = reinterpret_cast<const float*>(lds + 6 * NTQ * 64 * 1);     // behind the three stages' planes
    f32x4 x[HT][NTQ];
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[s][t][r] = 0.01f * ((lane * 7 + r * 3 + s * 5 + t) % 41) - 0.2f;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
        // the block's condition embedding: row-dependent, read once per block and half-tile (cold memory every repetition)
        f32x4 cv[HT][NTQ];
#pragma unroll
        for (int s = 0; s < HT; ++s)
#pragma unroll
            for (int t = 0; t < NTQ; ++t)
                cv[s][t] = *reinterpret_cast<const f32x4*>(cond + (size_t)rep * cold_stride + (((size_t)(ht0 + s) * NTQ + t) * 64 + lane) * 4);
        f32x4 h1[HT][NTQ], h2[HT][NTQ], o[HT][NTQ];
        stage<HT>(h1, x, lds + 0 * 2 * NTQ * 64, vec + 0, vec + 32, vec + 192, 1e-4f, lane);
        stage<HT>(h2, h1, lds + 1 * 2 * NTQ * 64, vec + 64, vec + 96, vec + 224, 1e-4f, lane);
#pragma unroll
        for (int s = 0; s < HT; ++s)
#pragma unroll
            for (int t = 0; t < NTQ; ++t) h2[s][t] += cv[s][t];
        stage<HT>(o, h2, lds + 2 * 2 * NTQ * 64, vec + 128, vec + 160, vec + 256, 1e-4f, lane);
#pragma unroll
        for (int s = 0; s < HT; ++s)
#pragma unroll
            for (int t = 0; t < NTQ; ++t) x[s][t] = (o[s][t] + x[s][t]) * 0.5f;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < HT; ++s)
#pragma unroll
        for (int t = 0; t < NTQ; ++t) acc += x[s][t][0] + x[s][t][1] + x[s][t][2] + x[s][t][3];
    sink[(size_t)wave_g * 64 + lane] = acc + (float)(q + j);
    if (lane == 0) cyc[wave_g] = t1 - t0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 100;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, waves = Q_WAVES, nwaves = cus * waves, nht = nwaves * Q_HT;
    auto dalloc = [](size_t bytes) { void* p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
    const int n_u4 = 6 * NTQ * 64 + 128;
    std::vector<unsigned short> himg((size_t)n_u4 * 8);
    for (size_t i = 0; i < himg.size(); ++i) himg[i] = (unsigned short)(0x2000 + (i * 37) % 0x0c00);
    std::vector<float> hv(512);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = 0.5f + 0.001f * (i % 97);
    uint4* image = (uint4*)dalloc((size_t)n_u4 * 16);
    (void)hipMemcpy(image, himg.data(), (size_t)6 * NTQ * 64 * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy((char*)image + (size_t)6 * NTQ * 64 * 16, hv.data(), 128 * 16, hipMemcpyHostToDevice);
    const size_t cold_stride = (size_t)nht * NTQ * 256;
    float* cond = (float*)dalloc((size_t)reps * cold_stride * 4);
    long long* cyc = (long long*)dalloc((size_t)nwaves * 8);
    float* sink = (float*)dalloc((size_t)nwaves * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 30; ++it) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_q16<Q_HT>, dim3(cus), dim3(64 * waves), 0, 0, image, n_u4, cond, reps, nht, cyc, sink, cold_stride);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<long long> hc(nwaves);
    (void)hipMemcpy(hc.data(), cyc, (size_t)nwaves * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto c : hc) m += (double)c; m /= nwaves;
    const double rows_per_wave = 16.0 * Q_HT;
    printf("q16 N=%d half-tiles/wave=%d waves/CU=%d: %.0f cycles per block per wave = %.0f per 32 rows; kernel %.1f us = %.2f us per block-round (%d rows per CU per round)\n",
           N, Q_HT, waves, m / reps, m / reps * 32.0 / rows_per_wave, best * 1e3, best * 1e3 / reps, (int)(rows_per_wave * waves));
    return 0;
}
