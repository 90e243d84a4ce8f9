// 128-wide ResidualBlocks for LARGE launches: the four waves of a workgroup share every weight plane through LDS.
//
// Why (profiles/r01i_pmc_summary.txt): with one wave per 32-row tile fetching its own weight planes
// (k_resblock_h), a 256 -> 128 up block moves 384 KiB of packed planes from L2 per tile -- 8 of a wave's 10 vector loads per
// k16-step, every wave of a workgroup fetching the same bytes: 1.5 GB of L2 -> CU traffic per launch, ~64 GB/s per CU, which
// is what a CU can pull out of its XCD's L2 (MI355X guide: 66-73 GB/s per CU for L2-served rows).  The kernel was bound by
// that, not by its MFMA (27 % busy) or VALU issue.  Here a workgroup = 4 waves = 4 row tiles streams each plane ONCE into
// an LDS ring with LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no register rotation of prefetched
// operands) and all four waves read it with ds_read_b128: 4x less L2 traffic, 256 B/clk of LDS bandwidth instead of
// 64 B/clk of vector-L1 return path.  The private operands of a wave (its tile of the input tensors, the precomputed
// condition embedding) travel through the same ring, so that no compiler-visible vector load sits between two stages
// (hipcc's own s_waitcnt for such a load would drain the DMA queue: guide, "three .s-level traps").
//
// Ring: kRingChunks slots of 8 KiB; a chunk = 8 pieces of 1 KiB (= one wave-wide b128 access); wave w issues pieces 2w and
// 2w+1 of every chunk (one asm statement: two DMAs, the second through the instruction offset, which moves the global AND the
// LDS address).  The producer runs kRingDist chunks ahead of the consumer, across stage boundaries.  An EVENT consumes 1-3
// chunks: counted `s_waitcnt vmcnt(2 * chunks still in flight)`, raw s_barrier (every wave's pieces have landed; every wave
// is done with the previous event's slots), issue as many chunks as the event consumes, read.  A shared (weight) chunk holds
// the hi/lo planes of ONE k16-step for the 4 out tiles; a private chunk holds two 8-feature groups per wave.
//
// Arithmetic per element is that of resblock_body_h (same packed planes, same scales, same accumulation order per
// accumulator), except that -log2(e) is folded into the staged LayerNorm vectors (one VALU less per element: the SiLU's
// exponent argument is the LayerNorm output itself).
#pragma once
#include "dsg_split.hpp"

namespace dsg {

constexpr int kRingChunks = 8;     // 64 KiB; two workgroups per CU
constexpr int kRingDist = 6;       // chunks in flight ahead of the consumer; an event consumes 2: 6 + 2 = the ring
constexpr int kChunkU4 = 512;      // uint4 per chunk
constexpr int kWideMaxChunks = 96; // chunk program length bound (2*16 + 8 + 8 + 8 + 2*16 + 8)
constexpr int kWideVec = 2 * kLnLdsW1 + 10 * 128;   // floats: LN1 | LN2 | LN3 (gamma', beta'), time bias, c2, c3, epilogue LN + bias

// two 1 KiB pieces: global (per-lane address) -> LDS (wave-uniform address, lane-linear); the instruction offset of the second
// DMA moves the global AND the LDS address
__device__ __forceinline__ void glds_pair(const void* gsrc_lane, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

// The chunk program of one block (+ optional Linear epilogue), as seen by ONE wave: where chunk c comes from.  Evaluated once
// per wave at kernel start, one chunk per lane, into an LDS table; the producer then reads one 8-byte entry per chunk (the
// first version decoded the chunk index with scalar branches at every issue: 3 400 SALU instructions and 1 200 branches per
// wave against 4 300 VALU, all of it in the in-order issue stream of the wave that also has to feed the matrix core).
struct WideProg {
    int cA, cB, cC, cD, cE, cF;   // cumulative ends of: stage 1 (x, W1 per step) | stage 2 (W2) | condition embedding (private) |
                                  // stage 3 (W3) | shortcut (x, Wsc per step) or residual re-read (private) | epilogue Linear
    int ks0;                      // k16-steps of in0 (the rest of stage 1 reads in1)
    const float *x0, *x1, *cp;    // this wave's tile of in0 / in1 / cond_pre
    const uint4 *w1, *w2, *w3, *wsc, *wl;   // this wave's out tile of each packed matrix (step 0, hi plane)
    int lin_spc;                  // epilogue Linear: k16-steps per chunk (4 / pow2(NTO))
    bool sclin;
    int dbg;                      // measurement: 32 = the second read of x hits one hot KiB, 64 = so does the first read of in0

    __device__ __forceinline__ const void* source(int c) const {   // c may differ per lane
        const int iA = c, iB = c - cA, iC = c - cB, iD = c - cC, iE = c - cD, iF = c - cE;
        const int S1 = iA >> 1, SE = iE >> 1;
        const void* xa = S1 < ks0 ? (const void*)(x0 + ((dbg & 64) ? 0 : (size_t)S1 * 512)) : (const void*)(x1 + (size_t)(S1 - ks0) * 512);
        const void* xe = (dbg & 32) ? (const void*)x0 : (SE < ks0 ? (const void*)(x0 + (size_t)SE * 512) : (const void*)(x1 + (size_t)(SE - ks0) * 512));
        const void* sA = (iA & 1) ? (const void*)(w1 + (size_t)S1 * 128) : xa;
        const void* sE = sclin ? ((iE & 1) ? (const void*)(wsc + (size_t)SE * 128) : xe) : (const void*)(x0 + (size_t)iE * 512);
        const void* r = wl + (size_t)iF * lin_spc * 128;
        r = c < cE ? sE : r;
        r = c < cD ? (const void*)(w3 + (size_t)iD * 128) : r;
        r = c < cC ? (const void*)(cp + (size_t)iC * 512) : r;
        r = c < cB ? (const void*)(w2 + (size_t)iB * 128) : r;
        r = c < cA ? sA : r;
        return r;
    }
};

// Measurement switches (BlockLinArgsH::dbg, env DSG_WIDE_DBG read by the host launcher; results are WRONG with any of them
// set -- they exist to time the kernel with one ingredient removed): 1 no barrier, 2 no DMA wait, 16 no DMA issue, 32 / 64 hot-line sources
// (the 4 = no MFMA and 8 = no LayerNorm / SiLU / split VALU switches of the sweep in DESIGN.md 3.2 were removed with it).  Compiled out unless the library is built with
// -DDSG_WIDE_DBG_ENABLE=127 (tools/dbg_sweep.sh does that and restores the production build afterwards).
#ifndef DSG_WIDE_DBG_ENABLE
#define DSG_WIDE_DBG_ENABLE 0
#endif
struct WideRing {
    int dbg;
    const uint4* rd;                 // the ring as ordinary LDS (+ lane)
    const unsigned long long* tab;   // this wave's chunk-source table (LDS)
    unsigned long long nsrc[2];      // sources of the next two chunks to issue (read one event ahead)
    unsigned long long lane16;       // lane * 16
    unsigned lds_w;                  // LDS byte address of this wave's first piece of slot 0
    int gc, pi;                      // chunks consumed / issued so far
    int total;
};

__device__ __forceinline__ void ring_issue(WideRing& r, unsigned long long src) {
    glds_pair(reinterpret_cast<const void*>(src + r.lane16), r.lds_w + (unsigned)(r.pi & (kRingChunks - 1)) * 8192u);
    ++r.pi;
}

// Consume two chunks: returns the slot of the first one (the second is the next slot, modulo the ring).
__device__ __forceinline__ int wide_event(WideRing& r) {
    const int fl = r.pi - r.gc;                       // chunks issued and not consumed: kRingDist until the program runs out
    if (!(r.dbg & 2)) {
        if (fl == kRingDist) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // 4 chunks (8 DMAs of this wave) stay in flight
        else if (fl == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(r.dbg & 1)) __builtin_amdgcn_s_barrier();
    if (r.pi < r.total) {                             // chunk counts of all segments are even: two at a time
        if (!(r.dbg & 16)) { ring_issue(r, r.nsrc[0]); ring_issue(r, r.nsrc[1]); } else r.pi += 2;
        const int n0 = r.pi < kWideMaxChunks - 1 ? r.pi : kWideMaxChunks - 2;
        r.nsrc[0] = r.tab[n0]; r.nsrc[1] = r.tab[n0 + 1];
    }
    const int s = r.gc & (kRingChunks - 1);
    r.gc += 2;
    return s;
}
__device__ __forceinline__ int ring_next(int s) { return (s + 1) & (kRingChunks - 1); }

// 16 * silu(u) from u' = -log2(e) * u:  p = 2^u' = e^-u;  16 u / (1 + p) = u' / ((1 + p) * (-log2(e) / 16))
__device__ __forceinline__ f32x2 silu_scaled_l2(f32x2 up) {
    constexpr float k = -1.44269504088896341f / kActScale;
    const f32x2 p = {__builtin_amdgcn_exp2f(up.x), __builtin_amdgcn_exp2f(up.y)};
    const f32x2 q = __builtin_elementwise_fma(p, pk2(k), pk2(k));
    const f32x2 r = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    return up * r;
}
// gv / bv: gamma' and beta' of the 8 features (already times -log2 e); packed-f32 arithmetic (dsg_split.hpp: same bits)
__device__ __forceinline__ void act8_l2(float (&v)[8], const float (&x)[8], float c, float d, const float4 g0, const float4 b0, const float4 g1,
                                        const float4 b1) {
    const f32x2 cc = pk2(c), dd = pk2(d);
    const f32x2 r0 = silu_scaled_l2(__builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{x[0], x[1]}, cc, dd), f32x2{g0.x, g0.y}, f32x2{b0.x, b0.y}));
    const f32x2 r1 = silu_scaled_l2(__builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{x[2], x[3]}, cc, dd), f32x2{g0.z, g0.w}, f32x2{b0.z, b0.w}));
    const f32x2 r2 = silu_scaled_l2(__builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{x[4], x[5]}, cc, dd), f32x2{g1.x, g1.y}, f32x2{b1.x, b1.y}));
    const f32x2 r3 = silu_scaled_l2(__builtin_elementwise_fma(__builtin_elementwise_fma(f32x2{x[6], x[7]}, cc, dd), f32x2{g1.z, g1.w}, f32x2{b1.z, b1.w}));
    v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y; v[4] = r2.x; v[5] = r2.y; v[6] = r3.x; v[7] = r3.y;
}

template <int NT>
__device__ __forceinline__ void ring_wfrag(HFrag<NT>& w, const uint4* slot /* + lane */) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w.hi[nt] = slot[(2 * nt) * 64]; w.lo[nt] = slot[(2 * nt + 1) * 64]; }
}

template <int NT>
__device__ __forceinline__ void wacc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}

// B operand of one k16-step: split(16 silu(LN(x))) or split(x)
struct BOp { h8 hi, lo; };
template <bool LNACT>
__device__ __forceinline__ BOp wide_prep(const float (&x)[8], const float* gamma, const float* beta, int S, float c, float d, int h) {
    float v[8];
    if (LNACT) {
        const float4 g0 = ld4(gamma + 16 * S + 4 * h), b0 = ld4(beta + 16 * S + 4 * h);
        const float4 g1 = ld4(gamma + 16 * S + 8 + 4 * h), b1 = ld4(beta + 16 * S + 8 + 4 * h);
        act8_l2(v, x, c, d, g0, b0, g1, b1);
    } else {
#pragma unroll
        for (int q = 0; q < 8; q += 2) { const f32x2 t = f32x2{x[q], x[q + 1]} * pk2(kRawScale); v[q] = t.x; v[q + 1] = t.y; }
    }
    BOp b;
    split8(v, b.hi, b.lo);
    return b;
}
template <bool FIRST>
__device__ __forceinline__ void wide_mma(f32x16 (&acc)[4], const HFrag<4>& w, const BOp& b) {
    if (FIRST) mfma_step_h0<4>(acc, w, b.hi, b.lo); else mfma_step_h<4>(acc, w, b.hi, b.lo);
}
// register-fed stage over the ring: out = W * split(16 silu(LN(in))), two chunks (= two k16-steps x 4 out tiles) per event.
// (A form software-pipelined by one step -- the operand of step S+1 computed next to the MFMAs of step S, with
// sched_group_barrier asking for MFMA / 7 VALU / MFMA ... -- measured 105.3 vs 103.8 us on the dominant launch: the kernel is
// not bound by the order of a wave's own VALU and MFMA instructions.  Taken out again.)
__device__ __forceinline__ void wide_stage_reg(f32x16 (&out)[4], const f32x16 (&in)[4], WideRing& r, const float* gamma, const float* beta,
                                               float mean, float rstd, int h) {
    const float c = rstd, d = -mean * rstd;
#pragma unroll
    for (int E = 0; E < 4; ++E) {
        const int s0 = wide_event(r);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int S = 2 * E + half, s = half ? ring_next(s0) : s0;
            HFrag<4> w;
            ring_wfrag<4>(w, r.rd + s * kChunkU4);
            const int t = S >> 1, r0 = 8 * (S & 1);
            const float x[8] = {in[t][r0], in[t][r0 + 1], in[t][r0 + 2], in[t][r0 + 3], in[t][r0 + 4], in[t][r0 + 5], in[t][r0 + 6], in[t][r0 + 7]};
            const BOp b = wide_prep<true>(x, gamma, beta, S, c, d, h);
            if (S == 0) wide_mma<true>(out, w, b); else wide_mma<false>(out, w, b);
        }
    }
}

// memory-fed stage over the ring: per k16-step one private chunk (this wave's two groups of x) and one weight chunk.
// FIRST: the chain starts here (accumulators undefined before).
template <bool LNACT, bool FIRST>
__device__ __forceinline__ void wide_mem_step(f32x16 (&acc)[4], int S, WideRing& r, int wave, const float* gamma, const float* beta, float c, float d,
                                              int h) {
    const int s0 = wide_event(r), s1 = ring_next(s0);
    const uint4* xs = r.rd + s0 * kChunkU4 + (2 * wave) * 64;
    const float4 xa = __builtin_bit_cast(float4, xs[0]), xb = __builtin_bit_cast(float4, xs[64]);
    HFrag<4> w;
    ring_wfrag<4>(w, r.rd + s1 * kChunkU4);
    const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    const BOp b = wide_prep<LNACT>(x, gamma, beta, S, c, d, h);
    wide_mma<FIRST>(acc, w, b);
}
template <bool LNACT, bool ZERO>
__device__ __forceinline__ void wide_stage_mem(f32x16 (&acc)[4], int steps, WideRing& r, int wave, const float* gamma,
                                               const float* beta, float mean, float rstd, int h) {
    const float c = rstd, d = -mean * rstd;
    wide_mem_step<LNACT, ZERO>(acc, 0, r, wave, gamma, beta, c, d, h);
    for (int S = 1; S < steps; ++S) wide_mem_step<LNACT, false>(acc, S, r, wave, gamma, beta, c, d, h);
}

// acc += private tensor (16 groups of this wave's tile), two chunks (4 groups = one accumulator tile) per event
__device__ __forceinline__ void wide_add_private(f32x16 (&acc)[4], WideRing& r, int wave, bool take) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s0 = wide_event(r), s1 = ring_next(s0);
        if (take) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4* src = r.rd + (q < 2 ? s0 : s1) * kChunkU4 + (2 * wave + (q & 1)) * 64;
                const float4 cv = __builtin_bit_cast(float4, src[0]);
                acc[e][4 * q + 0] += cv.x; acc[e][4 * q + 1] += cv.y; acc[e][4 * q + 2] += cv.z; acc[e][4 * q + 3] += cv.w;
            }
        }
    }
}

// The vector reads go out four at a time AHEAD of the arithmetic (a scheduling barrier keeps the group together): written as "read 4,
// use 4" hipcc kept that order and reused the destination registers, i.e. sixteen LDS round trips in a row, each fully exposed, three
// times per tile of a 128-wide block (k_panel128_h: 51 of its 63 full lgkmcnt drains).
template <int NT>
__device__ __forceinline__ void acc_unscale_add_lds(f32x16 (&acc)[NT], float inv, const float* vec, int h) {
    constexpr int NB = 1;                    // out tiles per batch (two: 20-196 B of scratch in the panel kernels)
#pragma unroll
    for (int n0 = 0; n0 < NT; n0 += NB) {
        float4 b[4 * NB];
#pragma unroll
        for (int k = 0; k < 4 * NB; ++k)
            if (n0 + k / 4 < NT) b[k] = ld4(vec + 32 * (n0 + k / 4) + 8 * (k % 4) + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4 * NB; ++k)
            if (n0 + k / 4 < NT) {
                const int nt = n0 + k / 4, q = k % 4;
                acc[nt][4 * q + 0] = fmaf(acc[nt][4 * q + 0], inv, b[k].x); acc[nt][4 * q + 1] = fmaf(acc[nt][4 * q + 1], inv, b[k].y);
                acc[nt][4 * q + 2] = fmaf(acc[nt][4 * q + 2], inv, b[k].z); acc[nt][4 * q + 3] = fmaf(acc[nt][4 * q + 3], inv, b[k].w);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// EPI: 0 = block only, 1 = + raw Linear (Down/Upsample), 2 = + final (LayerNorm + SiLU + Linear, row-major out)
template <bool SCLIN, int EPI, int NTO>
__global__ __launch_bounds__(256, 2) void k_wide128_h(const BlockLinArgsH A) {
    constexpr int N = 128, NT = 4, NG = 16;
    constexpr int NTOP = NTO <= 1 ? 1 : (NTO == 2 ? 2 : 4), SPC = 4 / NTOP;       // epilogue Linear: k16-steps per chunk
    __shared__ uint4 lds[kRingChunks * kChunkU4 + kWideVec / 4 + 4 * kWideMaxChunks / 2];
    float* const vec = reinterpret_cast<float*>(lds + kRingChunks * kChunkU4);
    unsigned long long* const tab_all = reinterpret_cast<unsigned long long*>(lds + kRingChunks * kChunkU4 + kWideVec / 4);
    float* const g1v = vec, * const b1v = vec + kLnLdsW1, * const v2 = vec + 2 * kLnLdsW1;
    float* const g2v = v2, * const b2v = v2 + 128, * const g3v = v2 + 256, * const b3v = v2 + 384, * const tbv = v2 + 512, * const c2v = v2 + 640,
         * const c3v = v2 + 768, * const gLv = v2 + 896, * const bLv = v2 + 1024, * const biasLv = v2 + 1152;
    const BlockArgsH& ah = A.b;
    const BlockArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_raw = blockIdx.x * 4 + wave;
    const bool live = tile_raw < a.ntiles;                 // idle waves of the last workgroup still move their pieces and meet the barriers
    const int tile = live ? tile_raw : a.ntiles - 1;
    const int ptile = tile % a.tiles_per_pass;
    const int ks0 = a.in0.groups >> 1, ks1 = a.in1.groups >> 1, KS1 = ks0 + ks1;
    const int last_tile = blockIdx.x * 4 + 3 < a.ntiles ? blockIdx.x * 4 + 3 : a.ntiles - 1;
    const bool wg_cond = last_tile >= a.uncond_tiles, my_cond = tile >= a.uncond_tiles;
    constexpr float kL2 = -1.44269504088896341f;
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x31);

    // ---- per-feature vectors -> LDS (LayerNorm vectors times -log2 e)
    {
        const int n1 = ln1_extent(a);
        for (int i = threadIdx.x; i < n1; i += 256) { g1v[i] = a.gamma1[i] * kL2; b1v[i] = a.beta1[i] * kL2; }
        if (threadIdx.x < 128) {
            const int i = threadIdx.x;
            g2v[i] = a.gamma2[i] * kL2; b2v[i] = a.beta2[i] * kL2; g3v[i] = a.gamma3[i] * kL2; b3v[i] = a.beta3[i] * kL2;
            c2v[i] = a.c2[i]; c3v[i] = a.c3[i];
            if (!a.ts) tbv[i] = a.tbias[(size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride + i];
            if (EPI == 2) { gLv[i] = A.l.l.gamma[i] * kL2; bLv[i] = A.l.l.beta[i] * kL2; }
            if (EPI != 0) biasLv[i] = i < NTO * 32 ? A.l.l.bias[i] : 0.f;
        }
    }
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x32);
    // ---- LN1 statistics (Chan merge of the producers' (mean, M2)), as resblock_body_h
    float mean1, rstd1;
    {
        const float2 s0 = reinterpret_cast<const float2*>(a.in0.stats)[(size_t)seg_tile(a.in0, tile) * 32 + j];
        float mean = s0.x, m2 = s0.y;
        if (a.in1.groups) {
            const float2 s1 = reinterpret_cast<const float2*>(a.in1.stats)[(size_t)seg_tile(a.in1, tile) * 32 + j];
            const float dd = s1.x - mean;
            m2 = m2 + s1.y + dd * dd * a.chan_w;
            mean = mean + dd * a.chan_f;
        }
        mean1 = mean;
        rstd1 = rsqrtf(m2 * a.inv_nin + kLnEps);
        if (SCLIN) range_check(a.range_flag, mean, m2);
    }
    const float inv1 = ah.kc[0], inv2 = ah.kc[1], inv3 = ah.kc[2];
    int entry = 0;
    if (a.ts) {
        int row = ptile * 32 + j;
        row = row < a.nrows ? row : a.nrows - 1;
        entry = a.ts[row];
    }

    // ---- chunk program
    WideProg p;
    p.ks0 = ks0; p.sclin = SCLIN; p.lin_spc = SPC; p.dbg = A.dbg & DSG_WIDE_DBG_ENABLE;
    p.cA = 2 * KS1; p.cB = p.cA + 8; p.cC = p.cB + (wg_cond ? 8 : 0); p.cD = p.cC + 8; p.cE = p.cD + (SCLIN ? 2 * KS1 : 8);
    p.cF = p.cE + (EPI != 0 ? 8 / SPC : 0);
    p.x0 = a.in0.data + (size_t)seg_tile(a.in0, tile) * a.in0.groups * 256;
    p.x1 = a.in1.groups ? a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 : p.x0;
    p.cp = a.cond_pre + (size_t)ptile * NG * 256;
    p.w1 = ah.W1h + (size_t)wave * KS1 * 128; p.w2 = ah.W2h + (size_t)wave * 8 * 128; p.w3 = ah.W3h + (size_t)wave * 8 * 128;
    p.wsc = SCLIN ? ah.Wsch + (size_t)wave * KS1 * 128 : p.w1;
    if (EPI != 0) {
        const int nt = wave % NTOP, sl = wave / NTOP;                     // wave -> (out tile, step within the chunk); tiles >= NTO: reload tile 0
        p.wl = A.l.Wh + (size_t)(nt < NTO ? nt : 0) * 8 * 128 + (size_t)sl * 128;
    } else {
        p.wl = p.w1;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds;
    WideRing r;
    r.rd = lds + lane; r.gc = 0; r.pi = 0; r.total = p.cF; r.dbg = A.dbg & DSG_WIDE_DBG_ENABLE;
    r.lds_w = lds0 + (unsigned)wave * 2048u;
    r.lane16 = (unsigned long long)lane * 16ull;
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x33);
    {   // this wave's chunk-source table: one chunk per lane
        unsigned long long* tab = tab_all + wave * kWideMaxChunks;
        for (int c = lane; c < kWideMaxChunks; c += 64)
            tab[c] = reinterpret_cast<unsigned long long>(p.source(c < p.cF ? c : p.cF - 1));
        r.tab = tab;
    }
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x34);
    __syncthreads();                                        // the staged vectors and tables are visible; no DMA is in flight yet
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x35);
#pragma unroll
    for (int i = 0; i < kRingDist; ++i) ring_issue(r, r.tab[i]);
    r.nsrc[0] = r.tab[kRingDist]; r.nsrc[1] = r.tab[kRingDist + 1];

    // ---- stage 1
    f32x16 acc1[NT];
    wide_stage_mem<true, true>(acc1, KS1, r, wave, g1v, b1v, mean1, rstd1, h);
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x36);
    if (a.ts) acc_unscale_add<NT>(acc1, inv1, a.tbias + (size_t)entry * a.tb_stride, h);
    else acc_unscale_add_lds<NT>(acc1, inv1, tbv, h);
    if (a.save_h1 && live) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h1 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc1[G >> 2][4 * (G & 3)], acc1[G >> 2][4 * (G & 3) + 1], acc1[G >> 2][4 * (G & 3) + 2], acc1[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 2
    f32x16 acc2[NT];
    {
        float mean, m2;
        acc_stats<N, NT>(acc1, h, mean, m2);
        wide_stage_reg(acc2, acc1, r, g2v, b2v, mean, rsqrtf(m2 * (1.0f / N) + kLnEps), h);
        acc_unscale_add_lds<NT>(acc2, inv2, c2v, h);
    }
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x37);
    if (wg_cond) wide_add_private(acc2, r, wave, my_cond);
    if (a.save_h2 && live) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h2 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc2[G >> 2][4 * (G & 3)], acc2[G >> 2][4 * (G & 3) + 1], acc2[G >> 2][4 * (G & 3) + 2], acc2[G >> 2][4 * (G & 3) + 3]));
    }

    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x38);
    // ---- stage 3 (+ shortcut in the same scaled accumulator)
    f32x16 (&acc3)[NT] = acc1;
    {
        float mean, m2;
        acc_stats<N, NT>(acc2, h, mean, m2);
        wide_stage_reg(acc3, acc2, r, g3v, b3v, mean, rsqrtf(m2 * (1.0f / N) + kLnEps), h);
    }
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x39);
    if (SCLIN) {
        wide_stage_mem<false, false>(acc3, KS1, r, wave, nullptr, nullptr, 0.f, 1.f, h);
        acc_unscale_add_lds<NT>(acc3, inv3, c3v, h);
    } else {
        acc_unscale_add_lds<NT>(acc3, inv3, c3v, h);
        wide_add_private(acc3, r, wave, true);
    }

    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x3a);
    // ---- statistics + store
    float xmean, xm2;
    acc_stats<N, NT>(acc3, h, xmean, xm2);
    if ((EPI == 0 || A.store_block_out) && live && !(A.dbg & DSG_WIDE_DBG_ENABLE & 64)) {
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(xmean, xm2);
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2], acc3[G >> 2][4 * (G & 3) + 3]));
    }
    DSG_STAMP(SCLIN && EPI == 0 && wave == 0 && blockIdx.x == 0, 0x3b);
    if (EPI == 0) return;
    if (EPI == 1) range_check(a.range_flag, xmean, xm2);

    // ---- epilogue Linear over the ring: a chunk holds SPC k16-steps x NTOP out tiles
    const LinArgs& la = A.l.l;
    f32x16 acc[NTO];
    {
        const float c = EPI == 2 ? rsqrtf(xm2 * la.inv_in_w + kLnEps) : 1.f, d = -xmean * c;
        int sl0 = 0;
#pragma unroll
        for (int ch = 0; ch < 8 / SPC; ++ch) {
            const int s = (ch & 1) ? ring_next(sl0) : (sl0 = wide_event(r));
#pragma unroll
            for (int sl = 0; sl < SPC; ++sl) {
                const int S = ch * SPC + sl, t = S >> 1, r0 = 8 * (S & 1);
                HFrag<NTO> w;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    w.hi[nt] = r.rd[s * kChunkU4 + ((sl * NTOP + nt) * 2) * 64];
                    w.lo[nt] = r.rd[s * kChunkU4 + ((sl * NTOP + nt) * 2 + 1) * 64];
                }
                const float x[8] = {acc3[t][r0], acc3[t][r0 + 1], acc3[t][r0 + 2], acc3[t][r0 + 3], acc3[t][r0 + 4], acc3[t][r0 + 5], acc3[t][r0 + 6],
                                    acc3[t][r0 + 7]};
                float v[8];
                if (EPI == 2) {
                    const float4 g0 = ld4(gLv + 16 * S + 4 * h), b0 = ld4(bLv + 16 * S + 4 * h);
                    const float4 g1 = ld4(gLv + 16 * S + 8 + 4 * h), b1 = ld4(bLv + 16 * S + 8 + 4 * h);
                    act8_l2(v, x, c, d, g0, b0, g1, b1);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q += 2) { const f32x2 t = f32x2{x[q], x[q + 1]} * pk2(kRawScale); v[q] = t.x; v[q + 1] = t.y; }
                }
                h8 bhi, blo;
                split8(v, bhi, blo);
                if (S == 0) mfma_step_h0<NTO>(acc, w, bhi, blo); else mfma_step_h<NTO>(acc, w, bhi, blo);
            }
        }
    }
    acc_unscale_add_lds<NTO>(acc, A.l.kc[EPI == 2 ? 1 : 0], biasLv, h);
    if (!live) return;
    if (EPI == 1) {
        const int NGo = (la.out_width + 7) / 8;
        float m, qq;
        lin_out_stats<NTO>(acc, h, la.out_width, la.inv_out_w, m, qq);
        if (h == 0) reinterpret_cast<float2*>(la.out_stats)[(size_t)tile * 32 + j] = make_float2(m, qq);
#pragma unroll
        for (int G = 0; G < NTO * 4; ++G)
            if (G < NGo)
                st4(la.out + ((size_t)tile * NGo + G) * 256 + lane * 4,
                    make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
    } else {
        const int pass = tile / la.tiles_per_pass, row = ptile * 32 + j;
        if (row < la.nrows) {
            float* o = la.out_rm + ((size_t)pass * la.nrows + row) * la.out_width;
            if ((la.out_width & 3) == 0) {
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G) {
                    const int f = 8 * G + 4 * h;
                    if (f < la.out_width)
                        st4(o + f, make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
                }
            } else {
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = 8 * G + 4 * h + q;
                        if (f < la.out_width) o[f] = acc[G >> 2][4 * (G & 3) + q];
                    }
            }
        }
    }
}

}  // namespace dsg
