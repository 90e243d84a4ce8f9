// One narrow ResidualBlock body (the code of k_fused_narrow_lds: LDS-resident planes and vectors, running tensor in registers) in a loop,
// 16 waves per CU = 4 per SIMD, every wave its own 32-row tile: what does ONE block cost under the load the real kernel runs under, and what
// does it wait for?  (round 6: per-operator stamps put a 16-wide block at 12-15 k cycles and a 32-wide one at 14-20 k under 4 waves per SIMD,
// 5-8 x its vector-issue time.)   Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I diffsg_amd/csrc -DNB_N=16 -DNB_SCLIN=0 -o narrow_block tools/ubench/narrow_block.hip
#include "dsg_split.hpp"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
using namespace dsg;
#ifndef NB_N
#define NB_N 16
#endif
#ifndef NB_SCLIN
#define NB_SCLIN 0
#endif
#ifndef NB_WAVES
#define NB_WAVES 16
#endif
#ifndef NB_SK
#define NB_SK 0
#endif
#ifndef NB_PC
#define NB_PC 0
#endif
constexpr int N = NB_N;
constexpr bool SCLIN = NB_SCLIN != 0;

__global__ __launch_bounds__(1024, 4) void k_block(const BlockArgsH ah_in, const uint4* __restrict__ image, int n_u4, int reps, int ntiles, long long* cyc,
                                                   float* sink, size_t cold_stride_c, size_t cold_stride_s, const BlockArgsH* __restrict__ tab,
                                                   const NarrowLdsOp* __restrict__ ltab) {
    __shared__ uint4 lds[kNarrowLdsU4];
    for (int i = threadIdx.x; i < n_u4; i += blockDim.x) lds[i] = image[i];
    __syncthreads();
    const float* const ldsf = reinterpret_cast<const float*>(lds);
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (tile >= ntiles) return;
    BlockArgsH b = ah_in;
    // image layout (uint4 units): W1 [0, 1024) | W2 [1024, 1536) | W3 [1536, 2048) | Wsc [2048, 3072) | vectors from float offset 4 * 3072
    b.W1h = lds; b.W2h = lds + 1024; b.W3h = lds + 1536; b.Wsch = lds + 2048;
    const float* v = ldsf + 4 * 3072;
    b.b.gamma1 = v; b.b.beta1 = v + 96; b.b.gamma2 = v + 192; b.b.beta2 = v + 224; b.b.gamma3 = v + 256; b.b.beta3 = v + 288;
    b.b.c2 = v + 320; b.b.c3 = v + 352; b.b.tbias = v + 384;
    globalize<true>(b);
    f32x16 x[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[0][r] = (r < N / 2) ? 0.01f * ((lane * 7 + r * 3) % 41) - 0.2f : 0.f;
    float xmean = 0.f, xm2 = 1.0f;
    {
        float m, q;
        acc_stats<N, 1>(x, lane >> 5, m, q);
        xmean = m; xm2 = q;
    }
    const int lane_id = lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
#ifdef NB_REC
        {   // as the operator loop of k_fused_narrow_lds: the record of THIS operator is read from the table now
            b = tab[r & 3];
            const NarrowLdsOp lo = ltab[r & 3];
            b.W1h = lds + lo.w1; b.W2h = lds + lo.w2; b.W3h = lds + lo.w3; b.Wsch = lds + lo.wsc;
            b.b.gamma1 = ldsf + lo.g1; b.b.beta1 = ldsf + lo.b1; b.b.gamma2 = ldsf + lo.g2; b.b.beta2 = ldsf + lo.b2;
            b.b.gamma3 = ldsf + lo.g3; b.b.beta3 = ldsf + lo.b3; b.b.c2 = ldsf + lo.c2; b.b.c3 = ldsf + lo.c3; b.b.tbias = ldsf + lo.tb;
            globalize<true>(b);
        }
#endif
        if (b.b.in1.groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NB_COLD
        b.b.cond_pre = ah_in.b.cond_pre + (size_t)r * cold_stride_c;       // every repetition reads memory nobody has touched: HBM latency, as in the step
        if (SCLIN) { b.b.in1.data = ah_in.b.in1.data + (size_t)r * cold_stride_s; }
        globalize_io(b.b);
#endif
        #ifdef NB_STORE
        resblock_body_h<N, SCLIN, true, true, false, false, NB_PC, NB_SK>(b, tile, lane, &x, &xmean, &xm2, true, nullptr, 0);
#else
        resblock_body_h<N, SCLIN, true, true, false, false, NB_PC, NB_SK>(b, tile, lane, &x, &xmean, &xm2, false, nullptr, 0);
#endif
        // keep the values bounded: the block is a residual map
#pragma unroll
        for (int k = 0; k < 16; ++k) x[0][k] *= 0.5f;
        xmean *= 0.5f; xm2 *= 0.25f;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = xmean + xm2;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += x[0][k];
    sink[(size_t)tile * 64 + lane] = s;
    if (lane == 0) cyc[tile] = t1 - t0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int waves = NB_WAVES, ntiles = cus * waves, tpp = ntiles / 2;
    const int NG = N / 8;
    // buffers
    auto dalloc = [](size_t bytes) { void* p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
    std::vector<float> hv(4096);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = 0.5f + 0.001f * (i % 97);
    std::vector<unsigned short> himg((size_t)3072 * 8);
    for (size_t i = 0; i < himg.size(); ++i) himg[i] = (unsigned short)(0x2000 + (i * 37) % 0x0c00);   // small positive halfs
    uint4* image = (uint4*)dalloc((3072 + 1024) * 16);
    (void)hipMemcpy(image, himg.data(), himg.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy((char*)image + 3072 * 16, hv.data(), 1024 * 16, hipMemcpyHostToDevice);
#ifdef NB_COLD
    const size_t coldx = reps;
#else
    const size_t coldx = 1;
#endif
    float* skip = (float*)dalloc(coldx * (size_t)ntiles * NG * 256 * 4);
    float* skip_st = (float*)dalloc((size_t)ntiles * 32 * 2 * 4);
    float* cond = (float*)dalloc(coldx * (size_t)tpp * NG * 256 * 4);
    float* kc = (float*)dalloc(64);
    const float hkc[4] = {1e-4f, 1e-4f, 1e-4f, 0.f};
    (void)hipMemcpy(kc, hkc, 16, hipMemcpyHostToDevice);
    std::vector<float> hs((size_t)ntiles * NG * 256, 0.25f), hst((size_t)ntiles * 64);
    for (size_t i = 0; i < hst.size(); i += 2) { hst[i] = 0.25f; hst[i + 1] = 0.5f; }
    (void)hipMemcpy(skip, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(skip_st, hst.data(), hst.size() * 4, hipMemcpyHostToDevice);
    long long* cyc = (long long*)dalloc((size_t)ntiles * 8);
    float* sink = (float*)dalloc((size_t)ntiles * 64 * 4);
    BlockArgsH a;
    memset(&a, 0, sizeof a);
    a.b.in0.groups = NG; a.b.in0.width = N;
    if (SCLIN) { a.b.in1.data = skip; a.b.in1.stats = skip_st; a.b.in1.groups = NG; a.b.in1.width = N; }
    a.b.cond_pre = cond;
    a.b.out = (float*)dalloc((size_t)ntiles * NG * 256 * 4); a.b.out_stats = (float*)dalloc((size_t)ntiles * 64 * 4);
    int* rf = (int*)dalloc(64); a.b.range_flag = rf;
    a.b.ntiles = ntiles; a.b.tiles_per_pass = tpp; a.b.uncond_tiles = tpp; a.b.nrows = tpp * 32;
    const float n0 = N, n1 = SCLIN ? N : 0;
    a.b.chan_w = n0 * n1 / (n0 + n1); a.b.chan_f = n1 / (n0 + n1); a.b.inv_nin = 1.0f / (n0 + n1);
    a.kc = kc; a.m1 = kc; a.m2 = kc; a.m3 = kc; a.msc = kc;
#ifdef NB_NOCOND
    a.b.uncond_tiles = ntiles;
#endif
    BlockArgsH* tab = (BlockArgsH*)dalloc(4 * sizeof(BlockArgsH));
    NarrowLdsOp* ltab = (NarrowLdsOp*)dalloc(4 * sizeof(NarrowLdsOp));
    {
        BlockArgsH t4[4] = {a, a, a, a};
        NarrowLdsOp l; memset(&l, 0, sizeof l);
        l.w1 = 0; l.w2 = 1024; l.w3 = 1536; l.wsc = 2048;
        const unsigned vb = 4 * 3072;
        l.g1 = vb; l.b1 = vb + 96; l.g2 = vb + 192; l.b2 = vb + 224; l.g3 = vb + 256; l.b3 = vb + 288; l.c2 = vb + 320; l.c3 = vb + 352; l.tb = vb + 384;
        NarrowLdsOp l4[4] = {l, l, l, l};
        (void)hipMemcpy(tab, t4, sizeof t4, hipMemcpyHostToDevice);
        (void)hipMemcpy(ltab, l4, sizeof l4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 40; ++it) {       // ~20 ms of launches: the clock of an idle box ramps over tens of milliseconds
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_block, dim3(cus), dim3(64 * waves), 0, 0, a, image, 3072 + 1024, reps, ntiles, cyc, sink, (size_t)tpp * NG * 256, (size_t)ntiles * NG * 256, tab, ltab);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<long long> hc(ntiles);
    (void)hipMemcpy(hc.data(), cyc, (size_t)ntiles * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto c : hc) m += (double)c; m /= ntiles;
    printf("N=%d SCLIN=%d PC=%d cold=%d waves/CU=%d reps=%d: %.0f s_memtime ticks per block per wave (under load), kernel %.1f us = %.2f us per block-round\n", N,
           (int)SCLIN, (int)NB_PC, (int)(coldx > 1), waves, reps, m / reps, best * 1e3, best * 1e3 / reps);
    return 0;
}
