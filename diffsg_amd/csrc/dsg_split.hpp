// Wide ResidualBlocks on the real matrix cores: float32-accurate GEMMs as a 2-way fp16 split on
// v_mfma_f32_32x32x16_f16 with float32 accumulation.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the float32 VECTOR rate (157 TFLOP/s) and, measured here
// (tools/ubench/chain_rate.hip), does not overlap with VALU / VMEM issue from the same SIMD: the activation VALU of
// a block simply adds to its MFMA time (91 cycles per MFMA instead of 64).  The f16 MFMA is 16x faster per MAC and
// does overlap, so three of them -- hi*hi + hi*lo + lo*hi of operands split as x = hi + lo, hi = fp16(x),
// lo = fp16(x - hi) -- cost 96 cycles per 32x32x16 block instead of 512, at 22 significant bits per operand
// (2^-22 = 2.4e-7 relative; float32 itself is 6e-8) and exact float32 accumulation.  Operands are pre-scaled by
// powers of two so that the lo parts stay in fp16's normal range (weights by 2^e per layer from max|W|, activations
// by 16); the accumulator is un-scaled with one fma (which also adds the bias).  tools/ubench/split_rate.hip:
// 367 float32-equivalent TFLOP/s for the stage pattern of a 128-wide block, against 110 on the f32 MFMA.
//
// Layouts are those of dsg_kernels.hpp (same fragment tensors in HBM, same accumulator order): the C/D map of the
// 32x32x16 MFMA equals the 32x32x2 one, and its B operand of k16-step S is exactly fragment groups 2S and 2S+1 of the
// lane (8 halfs), so the accumulator-is-the-next-operand chaining carries over unchanged.  Packed weights: per
// (out tile, k16-step) two planes (hi, lo) of 64 lanes x 8 halfs.
#pragma once
#include "dsg_kernels.hpp"
#include "dsg_narrow8.hpp"

namespace dsg {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));

constexpr float kActScale = 16.0f;       // B operands that are LayerNorm+SiLU outputs (bounded by ~sqrt(width))
constexpr float kRawScale = 1.0f;        // B operands that are raw residual-stream values
// Raw operands (Linear shortcut, Down/Upsample, feature_proj) go through v_cvt_pkrtz unnormalised: beyond +-65504 the hi part
// saturates (round-toward-zero never gives inf) and the product is silently wrong.  Every kernel that splits a raw operand
// bounds its rows -- |mean| + sqrt(M2) >= max|x| from the LayerNorm statistics it has anyway -- and raises the handle's range
// flag above this limit; dsg_range_status reports it (the caller then switches to DSG_PRECISION_F32_MFMA).
constexpr float kRawLimit = 6.0e4f / kRawScale;
__device__ __forceinline__ void range_check(int* flag, float mean, float m2) {
    if (flag && fabsf(mean) + sqrtf(m2) > kRawLimit) *flag = 1;
}

// Weight scale exponent of one Linear from max|W|: max|W| * 2^e in [4096, 8192).
__device__ __forceinline__ int scale_exp(float maxabs) {
    int e = (int)floorf(log2f(8192.0f / fmaxf(maxabs, 1e-30f)));
    return e < -8 ? -8 : (e > 24 ? 24 : e);
}
// lin3 and the Linear shortcut accumulate into one chain: (2^e3 * kActScale) must equal (2^esc * kRawScale).
__device__ __forceinline__ int scale_exp_lin3(float m3, float msc) {
    const int e3 = scale_exp(m3), esc = scale_exp(msc) - 4;
    return e3 < esc ? e3 : esc;
}

#define DSG_MFMA_H(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (acc), 0, 0, 0)


// wp: packed planes of this k16-step for out tile 0, + lane; tile stride in uint4
template <int NT>
__device__ __forceinline__ void load_hfrag(HFrag<NT>& w, const uint4* __restrict__ wp, size_t nt_stride) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w.hi[nt] = wp[nt * nt_stride]; w.lo[nt] = wp[nt * nt_stride + 64]; }
}

// Two float32 values -> their packed hi halves (fp16, round toward zero) and packed lo halves fp16(v - hi): THREE instructions.
// v_fma_mixlo_f16 / v_fma_mixhi_f16 take the float32 value and the half it was rounded to (a source operand read as fp16), subtract in
// float32 and write the fp16 result straight into the low / high half of the destination.  Until round 4 the lo pair was
// "convert hi back (2), subtract (2), pack (1)" in the compiler-scheduled kernels and "v_fma_mix_f32 (2), pack (1)" in the panel kernel:
// 6 and 4 vector instructions per pair -- on kernels that sit at ~80 % of the vector issue port inside their MFMA loops.
// The remainder v - hi has at most 13 significant bits, two more than fp16 holds: it is rounded to nearest even here (the
// pack instruction truncated), so hi + lo is the 22-bit operand as before, half a unit of its last place closer to v on average.
// Inline asm operands are values: a product feeding the split is rounded to float32 first in EVERY kernel form (hipcc cannot contract it
// into the subtraction), so a row's bits do not depend on the form that computes it.
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
    // ONE statement, closed by a wait state: v_fma_mixhi_f16 writes half a register, and on gfx940+ a vector instruction that reads a
    // register the instruction just before it wrote in part (dst_sel / op_sel destination) gets the old contents ("dst_sel forwarding"
    // hazard).  hipcc pads that for instructions it emits itself, not for inline asm: without the s_nop the narrow kernels, where
    // the consumer sometimes follows directly, returned rows off by 1e-3 (tools/dbg_split.py; the panel kernel, whose split
    // pieces sit between MFMAs, never showed it).
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 0"
        : "=&v"(l) : "v"(v0), "v"(v1), "v"(hi));
    lo = l;
}
// The same split WITHOUT the closing wait state, for call sites whose next instruction is known not to read `lo`: the pieces of
// panel_pipe_s sit between two MFMA statements and their results are first read a k16-step later (round 5: the wait state is an issue
// slot like any other -- 192 per tile of a 128-wide up block -- in a kernel that the round's microbenchmark shows to be issue-bound).
__device__ __forceinline__ void split_pair_nowait(float v0, float v1, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l) : "v"(v0), "v"(v1), "v"(hi));
    lo = l;
}
// 8 float32 values (already multiplied by the activation scale) -> hi / lo half vectors; hi + lo carries 22 bits
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
    unsigned hh[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], hh[q], ll[q]);
    const uint4 uh = {hh[0], hh[1], hh[2], hh[3]}, ul = {ll[0], ll[1], ll[2], ll[3]};
    hi = __builtin_bit_cast(h8, uh); lo = __builtin_bit_cast(h8, ul);
}

template <int NT>
__device__ __forceinline__ void mfma_step_h(f32x16 (&acc)[NT], const HFrag<NT>& w, const h8 bhi, const h8 blo) {
    // term-major: the three products of one accumulator are NT instructions apart, so no MFMA waits on the one issued just
    // before it (per accumulator the order hi*hi, hi*lo, lo*hi is unchanged: same bits)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) DSG_MFMA_H(acc[nt], __builtin_bit_cast(h8, w.hi[nt]), bhi);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) DSG_MFMA_H(acc[nt], __builtin_bit_cast(h8, w.hi[nt]), blo);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) DSG_MFMA_H(acc[nt], __builtin_bit_cast(h8, w.lo[nt]), bhi);
}

// first k16-step of a chain: the hi*hi products start from the inline constant 0 (no zero-fill of the 16 * NT accumulators)
template <int NT>
__device__ __forceinline__ void mfma_step_h0(f32x16 (&acc)[NT], const HFrag<NT>& w, const h8 bhi, const h8 blo) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w.hi[nt]), bhi, z, 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) DSG_MFMA_H(acc[nt], __builtin_bit_cast(h8, w.hi[nt]), blo);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) DSG_MFMA_H(acc[nt], __builtin_bit_cast(h8, w.lo[nt]), bhi);
}

// kActScale * silu(u): the scale rides in the denominator (u * rcp((1 + e)/16)). exp(-u) is one v_exp_f32 of the rounded
// product u * log2(e): its rounding moves exp by at most |u| * 4e-8 relative, which only matters where silu is already
// ~0 (a compensated form was measured: no accuracy gain on any parity case, 4% of the wide operator's time)
__device__ __forceinline__ float silu_scaled(float u) {
    const float p = __builtin_amdgcn_exp2f(u * -1.44269504088896341f);
    return u * __builtin_amdgcn_rcpf(fmaf(p, 1.0f / kActScale, 1.0f / kActScale));
}

// Two values per instruction: gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 at the rate of their scalar forms (measured,
// tools/ubench/pkf32.hip: 5.6 cycles per pair against 10.8), each element rounded exactly like the scalar instruction - the
// packed forms below return the same bits as silu_scaled / the scalar LayerNorm expression, at 3/4 of the VALU time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk2(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 silu_scaled2(f32x2 u) {
    const f32x2 m = u * pk2(-1.44269504088896341f);
    const f32x2 p = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
    const f32x2 q = __builtin_elementwise_fma(p, pk2(1.0f / kActScale), pk2(1.0f / kActScale));
    const f32x2 r = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
    return u * r;
}
// LayerNorm as x*c + d (c = rstd, d = -mean*rstd), affine, SiLU, activation scale
__device__ __forceinline__ f32x2 act2(f32x2 x, f32x2 c, f32x2 d, f32x2 g, f32x2 b) {
    return silu_scaled2(__builtin_elementwise_fma(__builtin_elementwise_fma(x, c, d), g, b));
}

__device__ __forceinline__ void act8(float (&v)[8], const float4 x0, const float4 x1, float c, float d, const float4 g0, const float4 b0,
                                     const float4 g1, const float4 b1) {
    const f32x2 cc = pk2(c), dd = pk2(d);
    const f32x2 r0 = act2(f32x2{x0.x, x0.y}, cc, dd, f32x2{g0.x, g0.y}, f32x2{b0.x, b0.y});
    const f32x2 r1 = act2(f32x2{x0.z, x0.w}, cc, dd, f32x2{g0.z, g0.w}, f32x2{b0.z, b0.w});
    const f32x2 r2 = act2(f32x2{x1.x, x1.y}, cc, dd, f32x2{g1.x, g1.y}, f32x2{b1.x, b1.y});
    const f32x2 r3 = act2(f32x2{x1.z, x1.w}, cc, dd, f32x2{g1.z, g1.w}, f32x2{b1.z, b1.w});
    v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y; v[4] = r2.x; v[5] = r2.y; v[6] = r3.x; v[7] = r3.y;
}

// Register-fed stage (stages 2 and 3): B = split(16 * silu(LN(in))).  An odd group count (N = 8, 4) pairs the last group
// with the accumulator's zero padding (rows >= N of the producer's packed weights and biases are zero).
template <int N, int NT, int NTI = NT, bool ZERO = false>
__device__ __forceinline__ void chain_from_acc_h(f32x16 (&out)[NT], const f32x16 (&in)[NTI], const uint4* __restrict__ wp,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float mean, float rstd,
                                                 int lane, int h, size_t nt_stride_override = 0, const HFrag<NT>* w0 = nullptr) {
    constexpr int KS = ((N + 7) / 8 + 1) / 2;
    const size_t nt_stride = nt_stride_override ? nt_stride_override : (size_t)KS * 128;
    const float c = rstd, d = -mean * rstd;
    HFrag<NT> wn;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gn0, bn0, gn1 = z4, bn1 = z4;
    if (w0) wn = *w0;                 // the first step's planes were requested earlier (narrow run)
    else load_hfrag<NT>(wn, wp + lane, nt_stride);
    gn0 = ld4(gamma + 4 * h); bn0 = ld4(beta + 4 * h);
    // the second quad of a step is padding when 16 S + 8 >= N (N = 8, 24, ...): x = 0, gamma = beta = 0 there, so the operand
    // is exactly 0 - neither loaded nor transformed
    if (8 < N) { gn1 = ld4(gamma + 8 + 4 * h); bn1 = ld4(beta + 8 + 4 * h); }
#pragma unroll
    for (int S = 0; S < KS; ++S) {
        const HFrag<NT> wc = wn;
        const float4 g0 = gn0, b0 = bn0, g1 = gn1, b1 = bn1;
        if (S + 1 < KS) {
            load_hfrag<NT>(wn, wp + (size_t)(S + 1) * 128 + lane, nt_stride);
            gn0 = ld4(gamma + 16 * (S + 1) + 4 * h); bn0 = ld4(beta + 16 * (S + 1) + 4 * h);
            if (16 * (S + 1) + 8 < N) { gn1 = ld4(gamma + 16 * (S + 1) + 8 + 4 * h); bn1 = ld4(beta + 16 * (S + 1) + 8 + 4 * h); }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int t = S >> 1, r0 = 8 * (S & 1);
        float v[8];
        if (16 * S + 8 < N) {
            act8(v, make_float4(in[t][r0], in[t][r0 + 1], in[t][r0 + 2], in[t][r0 + 3]),
                 make_float4(in[t][r0 + 4], in[t][r0 + 5], in[t][r0 + 6], in[t][r0 + 7]), c, d, g0, b0, g1, b1);
        } else {
            const f32x2 q0 = act2(f32x2{in[t][r0], in[t][r0 + 1]}, pk2(c), pk2(d), f32x2{g0.x, g0.y}, f32x2{b0.x, b0.y});
            const f32x2 q1 = act2(f32x2{in[t][r0 + 2], in[t][r0 + 3]}, pk2(c), pk2(d), f32x2{g0.z, g0.w}, f32x2{b0.z, b0.w});
            v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y;
            v[4] = v[5] = v[6] = v[7] = 0.f;
        }
        h8 bhi, blo;
        split8(v, bhi, blo);
        if (ZERO && S == 0) mfma_step_h0<NT>(out, wc, bhi, blo);   // `out` is undefined on entry: the chain starts from the constant 0
        else mfma_step_h<NT>(out, wc, bhi, blo);
    }
}

// Register-fed RAW stage (no LayerNorm / SiLU): Linear shortcut and the plain Linears of the narrow run when their input
// lives in registers.  `groups` (runtime, <= 4*NT) real groups; the missing group of an odd count is accumulator padding.
template <int NT, int NTI = NT, bool ZERO = false>
__device__ __forceinline__ void chain_raw_from_reg_h(f32x16 (&out)[NT], const f32x16 (&in)[NTI], int groups, const uint4* __restrict__ wp,
                                                     size_t nt_stride, int lane, const HFrag<NT>* w0 = nullptr) {
    const int steps = (groups + 1) >> 1;
    if constexpr (NT * NTI <= 2) {
        // every plane of the chain first, pinned above the arithmetic: left to itself hipcc sank each load to its first use -- the hi plane
        // in front of the step's first MFMA, the lo plane in front of its third, a full wait behind each -- two exposed L2 round trips
        // per k16-step of a Linear that is a handful of MFMAs (round 5, disassembly of k_fused_narrow_h; one tile's latency is the whole
        // small-batch step).  At most 32 plane registers; the wide pair kernels (NT * NTI > 2) keep the step-by-step form.
        HFrag<NT> wall[2 * NTI];
#pragma unroll
        for (int S = 0; S < 2 * NTI; ++S)
            if (S < steps) {
                if (S == 0 && w0) wall[0] = *w0;
                else load_hfrag<NT>(wall[S], wp + (size_t)S * 128 + lane, nt_stride);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int S = 0; S < 2 * NTI; ++S) {
            if (S < steps) {
                const int t = S >> 1, r0 = 8 * (S & 1);
                float v[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) v[p] = kRawScale * in[t][r0 + p];
                h8 bhi, blo;
                split8(v, bhi, blo);
                if (ZERO && S == 0) mfma_step_h0<NT>(out, wall[S], bhi, blo);
                else mfma_step_h<NT>(out, wall[S], bhi, blo);
            }
        }
        return;
    }
#pragma unroll
    for (int S = 0; S < 2 * NTI; ++S) {
        if (S < steps) {
            HFrag<NT> w;
            if (S == 0 && w0) w = *w0;
            else load_hfrag<NT>(w, wp + (size_t)S * 128 + lane, nt_stride);
            const int t = S >> 1, r0 = 8 * (S & 1);
            float v[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) v[p] = kRawScale * in[t][r0 + p];
            h8 bhi, blo;
            split8(v, bhi, blo);
            if (ZERO && S == 0) mfma_step_h0<NT>(out, w, bhi, blo);
            else mfma_step_h<NT>(out, w, bhi, blo);
        }
    }
}

// Memory-fed stage over one fragment tensor (runtime k16-step loop, prefetch distance 1).  `groups` may be odd: the
// missing group of the last step reads as zero (its packed weights are zero too).
// MAXS > 0 (narrow blocks: a segment has at most MAXS k16-steps): the step loop is unrolled, so the prefetched operands are
// renamed instead of rotated through v_mov (52 copies per iteration of the runtime loop - a tenth of a narrow block's VALU).
// FIXG > 0: the segment's group count is known at compile time (the skip input of a narrow up block is as wide as the block): the
// per-step conditions below fold away -- with a runtime count every array element is a conditional definition, and hipcc keeps the
// whole set of prefetched operands alive through the merges (spilled, next to the float32 section's larger live state).
// the activations of a <= 32-wide segment requested AHEAD of its chain (round 6: the skip input of a narrow up block -- a row-dependent read
// that was issued right where the chain starts: an exposed L2 / HBM round trip in front of stage 1's second half and another in front of the shortcut)
struct SegPre { float4 x0[2], x1[2]; };
template <int FIXG>
__device__ __forceinline__ void seg_prefetch(SegPre& p, const float* __restrict__ xp /* tile base + lane * 4 */) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int S = 0; S < 2; ++S) {
        p.x0[S] = z4; p.x1[S] = z4;
        if (2 * S < FIXG) p.x0[S] = ld4(xp + (size_t)(2 * S) * 256);
        if (2 * S + 1 < FIXG) p.x1[S] = ld4(xp + (size_t)(2 * S + 1) * 256);
    }
}

template <int NT, bool LNACT, int MAXS = 0, int FIXG = 0>
__device__ __forceinline__ void chain_from_mem_h(f32x16 (&acc)[NT], const float* __restrict__ xp, int groups_rt, const uint4* __restrict__ wp,
                                                 size_t nt_stride, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float mean, float rstd, const HFrag<NT>* w0 = nullptr, const SegPre* xpre = nullptr) {
    if (FIXG > 0 && groups_rt != FIXG) __builtin_trap();
    const int groups = FIXG > 0 ? FIXG : groups_rt;
    const int steps = (groups + 1) >> 1;
    if (steps <= 0) return;
    if constexpr (MAXS > 0) {
        if (steps > MAXS) __builtin_trap();             // a narrow block's segments are at most 32 features wide (plan builder)
        if (xpre) {
            // the activations are in registers already (requested a stage ago): planes and LayerNorm vectors step by step from LDS -- all
            // steps' operands at once (below) cost 48 registers the 32-wide up block does not have beside the prefetched tensor
            const float c = rstd, d = -mean * rstd;
#pragma unroll
            for (int S = 0; S < MAXS; ++S) {
                if (S < steps) {
                    HFrag<NT> w;
                    if (S == 0 && w0) w = *w0;
                    else load_hfrag<NT>(w, wp + (size_t)S * 128, nt_stride);
                    float v[8];
                    if (LNACT) {
                        const float4 g0 = ld4(gamma + 16 * S), b0 = ld4(beta + 16 * S), g1 = ld4(gamma + 16 * S + 8), b1 = ld4(beta + 16 * S + 8);
                        act8(v, xpre->x0[S], xpre->x1[S], c, d, g0, b0, g1, b1);
                    } else {
                        const float4 x0 = xpre->x0[S], x1 = xpre->x1[S];
                        v[0] = kRawScale * x0.x; v[1] = kRawScale * x0.y; v[2] = kRawScale * x0.z; v[3] = kRawScale * x0.w;
                        v[4] = kRawScale * x1.x; v[5] = kRawScale * x1.y; v[6] = kRawScale * x1.z; v[7] = kRawScale * x1.w;
                    }
                    h8 bhi, blo;
                    split8(v, bhi, blo);
                    mfma_step_h<NT>(acc, w, bhi, blo);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        const float c = rstd, d = -mean * rstd;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        HFrag<NT> w[MAXS];
        float4 x0[MAXS], x1[MAXS], g0[MAXS], b0[MAXS], g1[MAXS], b1[MAXS];
#pragma unroll
        for (int S = 0; S < MAXS; ++S) {                   // every load of the segment first (a step beyond `steps` is never read)
            if (S < steps) {
                if (S == 0 && w0) w[0] = *w0;
                else load_hfrag<NT>(w[S], wp + (size_t)S * 128, nt_stride);
                x0[S] = ld4(xp + (size_t)(2 * S) * 256);
                x1[S] = (2 * S + 1 < groups) ? ld4(xp + (size_t)(2 * S + 1) * 256) : z4;
                if (LNACT) { g0[S] = ld4(gamma + 16 * S); b0[S] = ld4(beta + 16 * S); g1[S] = ld4(gamma + 16 * S + 8); b1[S] = ld4(beta + 16 * S + 8); }
            }
        }
#pragma unroll
        for (int S = 0; S < MAXS; ++S) {
            if (S < steps) {
                float v[8];
                if (LNACT) {
                    act8(v, x0[S], x1[S], c, d, g0[S], b0[S], g1[S], b1[S]);
                } else {
                    v[0] = kRawScale * x0[S].x; v[1] = kRawScale * x0[S].y; v[2] = kRawScale * x0[S].z; v[3] = kRawScale * x0[S].w;
                    v[4] = kRawScale * x1[S].x; v[5] = kRawScale * x1[S].y; v[6] = kRawScale * x1[S].z; v[7] = kRawScale * x1[S].w;
                }
                h8 bhi, blo;
                split8(v, bhi, blo);
                mfma_step_h<NT>(acc, w[S], bhi, blo);
            }
        }
        return;
    }
    const float c = rstd, d = -mean * rstd;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    HFrag<NT> wn;
    float4 xn0, xn1, gn0 = z4, bn0 = z4, gn1 = z4, bn1 = z4;
    if (w0) wn = *w0;
    else load_hfrag<NT>(wn, wp, nt_stride);
    xn0 = ld4(xp);
    xn1 = groups > 1 ? ld4(xp + 256) : z4;
    if (LNACT) { gn0 = ld4(gamma); bn0 = ld4(beta); gn1 = ld4(gamma + 8); bn1 = ld4(beta + 8); }
    for (int S = 0; S < steps; ++S) {
        const HFrag<NT> wc = wn;
        const float4 x0 = xn0, x1 = xn1, g0 = gn0, b0 = bn0, g1 = gn1, b1 = bn1;
        if (S + 1 < steps) {
            load_hfrag<NT>(wn, wp + (size_t)(S + 1) * 128, nt_stride);
            xn0 = ld4(xp + (size_t)(2 * S + 2) * 256);
            xn1 = (2 * S + 3 < groups) ? ld4(xp + (size_t)(2 * S + 3) * 256) : z4;
            if (LNACT) {
                gn0 = ld4(gamma + 16 * (S + 1)); bn0 = ld4(beta + 16 * (S + 1));
                gn1 = ld4(gamma + 16 * (S + 1) + 8); bn1 = ld4(beta + 16 * (S + 1) + 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float v[8];
        if (LNACT) {
            act8(v, x0, x1, c, d, g0, b0, g1, b1);
        } else {
            v[0] = kRawScale * x0.x; v[1] = kRawScale * x0.y; v[2] = kRawScale * x0.z; v[3] = kRawScale * x0.w;
            v[4] = kRawScale * x1.x; v[5] = kRawScale * x1.y; v[6] = kRawScale * x1.z; v[7] = kRawScale * x1.w;
        }
        h8 bhi, blo;
        split8(v, bhi, blo);
        mfma_step_h<NT>(acc, wc, bhi, blo);
    }
}

// The same chain for ONE wave on a latency-bound path (k_unet_tile's plain Linears: nothing else runs on the SIMD, and a k16-step's
// arithmetic is a fraction of an L2 round trip): the k16-steps in batches of B, every load of a batch issued before its first MFMA --
// one exposed round trip per batch instead of one per step.  Same products in the same order: same bits.
template <int NT, bool LNACT, int B>
__device__ __forceinline__ void chain_from_mem_h_batched(f32x16 (&acc)[NT], const float* __restrict__ xp, int groups, const uint4* __restrict__ wp,
                                                         size_t nt_stride, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float mean, float rstd) {
    const int steps = (groups + 1) >> 1;
    const float c = rstd, d = -mean * rstd;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int S0 = 0; S0 < steps; S0 += B) {
        HFrag<NT> w[B];
        float4 x0[B], x1[B], g0[B], b0[B], g1[B], b1[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int S = S0 + i;
            x0[i] = z4; x1[i] = z4; g0[i] = z4; b0[i] = z4; g1[i] = z4; b1[i] = z4;
            if (S < steps) {
                load_hfrag<NT>(w[i], wp + (size_t)S * 128, nt_stride);
                x0[i] = ld4(xp + (size_t)(2 * S) * 256);
                if (2 * S + 1 < groups) x1[i] = ld4(xp + (size_t)(2 * S + 1) * 256);
                if (LNACT) { g0[i] = ld4(gamma + 16 * S); b0[i] = ld4(beta + 16 * S); g1[i] = ld4(gamma + 16 * S + 8); b1[i] = ld4(beta + 16 * S + 8); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < B; ++i) {
            if (S0 + i < steps) {
                float v[8];
                if (LNACT) {
                    act8(v, x0[i], x1[i], c, d, g0[i], b0[i], g1[i], b1[i]);
                } else {
                    v[0] = kRawScale * x0[i].x; v[1] = kRawScale * x0[i].y; v[2] = kRawScale * x0[i].z; v[3] = kRawScale * x0[i].w;
                    v[4] = kRawScale * x1[i].x; v[5] = kRawScale * x1[i].y; v[6] = kRawScale * x1[i].z; v[7] = kRawScale * x1[i].w;
                }
                h8 bhi, blo;
                split8(v, bhi, blo);
                mfma_step_h<NT>(acc, w[i], bhi, blo);
            }
        }
    }
}

// acc <- acc * inv + vec  (un-scale and add the per-feature vector, padded to NT*32).  NQ = real 8-feature groups: the padded
// groups of a narrow block (N < 32) hold exact zeros (zero weight rows, zero vector padding) and are left alone - a quarter
// of the vector reads and FMAs of an 8-wide block's three stages
template <int NT, int NQ = NT * 4>
__device__ __forceinline__ void acc_unscale_add(f32x16 (&acc)[NT], float inv, const float* __restrict__ vec, int h) {
    if constexpr (NT == 1) {
        // one accumulator tile (the narrow operators: latency-bound): the vector's groups in one batch of loads pinned above the
        // multiply-adds -- hipcc otherwise reuses one register quad: load, full wait, four FMAs, four times in a row
        float4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < NQ) b[q] = ld4(vec + 8 * q + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= NQ) continue;
            acc[0][4 * q + 0] = fmaf(acc[0][4 * q + 0], inv, b[q].x); acc[0][4 * q + 1] = fmaf(acc[0][4 * q + 1], inv, b[q].y);
            acc[0][4 * q + 2] = fmaf(acc[0][4 * q + 2], inv, b[q].z); acc[0][4 * q + 3] = fmaf(acc[0][4 * q + 3], inv, b[q].w);
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * nt + q >= NQ) continue;
            const float4 b = ld4(vec + 32 * nt + 8 * q + 4 * h);
            acc[nt][4 * q + 0] = fmaf(acc[nt][4 * q + 0], inv, b.x); acc[nt][4 * q + 1] = fmaf(acc[nt][4 * q + 1], inv, b.y);
            acc[nt][4 * q + 2] = fmaf(acc[nt][4 * q + 2], inv, b.z); acc[nt][4 * q + 3] = fmaf(acc[nt][4 * q + 3], inv, b.w);
        }
}

// the same with the vector already in registers (requested at the top of a narrow block: dsg_split.hpp resblock_body_h, PRE)
template <int NT, int NQ = NT * 4>
__device__ __forceinline__ void acc_unscale_add_reg(f32x16 (&acc)[NT], float inv, const float4 (&b)[NT * 4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * nt + q >= NQ) continue;
            const float4 v = b[4 * nt + q];
            acc[nt][4 * q + 0] = fmaf(acc[nt][4 * q + 0], inv, v.x); acc[nt][4 * q + 1] = fmaf(acc[nt][4 * q + 1], inv, v.y);
            acc[nt][4 * q + 2] = fmaf(acc[nt][4 * q + 2], inv, v.z); acc[nt][4 * q + 3] = fmaf(acc[nt][4 * q + 3], inv, v.w);
        }
}
template <int NT, int NQ = NT * 4>
__device__ __forceinline__ void load_vec4(float4 (&b)[NT * 4], const float* __restrict__ vec, int h) {
#pragma unroll
    for (int q = 0; q < NT * 4; ++q)
        if (q < NQ) b[q] = ld4(vec + 8 * q + 4 * h);
}

struct BlockArgsH {
    BlockArgs b;             // tensors, biases, LayerNorm parameters, time table, cond_pre: as the f32 kernel
    const uint4* W1h;        // packed planes [NT][KS1][2][64], KS1 = ceil(g0/2) + ceil(g1/2)
    const uint4* W2h;        // [NT][N/16][2][64]
    const uint4* W3h;
    const uint4* Wsch;       // [NT][KS1][2][64] or null
    const float* m1;         // device scalars: max|W| of lin1, lin2, lin3, shortcut (written by k_maxabs at bind time)
    const float* m2;
    const float* m3;
    const float* msc;
    const float* kc;         // device: { 2^-e1 / 16, 2^-e2 / 16, 2^-e3 / 16 } of this block, computed ONCE at bind time (by k_pack_h)
                             // from the same max|W| -- scale_exp is a log2 + floor + ldexp per stage, per wave, per block otherwise
};
// (dsg_kernels.hpp, as_global) LDS == true: the planes and vectors were redirected to an LDS image, only the rest is global
template <bool LDS>
__device__ __forceinline__ void globalize(BlockArgsH& a) {
    globalize_io(a.b);
    a.m1 = as_global(a.m1); a.m2 = as_global(a.m2); a.m3 = as_global(a.m3); a.msc = as_global(a.msc); a.kc = as_global(a.kc);
    if (!LDS) { globalize_params(a.b); a.W1h = as_global(a.W1h); a.W2h = as_global(a.W2h); a.W3h = as_global(a.W3h); a.Wsch = as_global(a.Wsch); }
}

// Needs cond_pre: the condition embedding Wc silu(cond*mask) is precomputed per call (sampling) or per step (training)
// and added here, never multiplied.
// XIN: in0 is not read from memory but handed over in registers `xr` with its row statistics (narrow run).
// XOUT: the output goes (back) into `xr`; it is stored only when `store_out` (something outside this wave reads it).
constexpr int kLnLdsW1 = 272, kLnLdsN = 128;   // LDS copy of the LayerNorm vectors: stage-1 width (<= 256 + pad), block width

template <int N, bool SCLIN, bool XIN = false, bool XOUT = XIN, bool PRE = false, bool LDSLN = false, int PC = 0, int SK = 0>
__device__ __forceinline__ void resblock_body_h(const BlockArgsH& ah, const int tile, const int lane, f32x16 (*xr)[(N + 31) / 32] = nullptr,
                                                float* xr_mean = nullptr, float* xr_m2 = nullptr, bool store_out = true,
                                                const float* lnp = nullptr, int entry_pre = -1) {
    constexpr int NG = (N + 7) / 8, NT = (N + 31) / 32;
    const BlockArgs& a = ah.b;
    const int h = lane >> 5, j = lane & 31;
    // LayerNorm affine vectors: global, or the block's LDS copy (stage_ln_params) - a third of the wide kernels' vector
    // memory instructions are these broadcast reads
    const float* const gamma1 = LDSLN ? lnp : a.gamma1;
    const float* const beta1 = LDSLN ? lnp + kLnLdsW1 : a.beta1;
    const float* const gamma2 = LDSLN ? lnp + 2 * kLnLdsW1 : a.gamma2;
    const float* const beta2 = LDSLN ? lnp + 2 * kLnLdsW1 + kLnLdsN : a.beta2;
    const float* const gamma3 = LDSLN ? lnp + 2 * kLnLdsW1 + 2 * kLnLdsN : a.gamma3;
    const float* const beta3 = LDSLN ? lnp + 2 * kLnLdsW1 + 3 * kLnLdsN : a.beta3;
    const int ptile = tile % a.tiles_per_pass;
    const int ks0 = (a.in0.groups + 1) >> 1, ks1 = (a.in1.groups + 1) >> 1, KS1 = ks0 + ks1;

    DSG_STAMP(XIN && tile == 0, 0x11);
    // ---- LN1 statistics (Chan merge of the producers' (mean, M2))
    float mean1, rstd1;
    {
        float mean, m2;
        if (XIN) { mean = *xr_mean; m2 = *xr_m2; }
        else { const float2 s0 = reinterpret_cast<const float2*>(a.in0.stats)[(size_t)seg_tile(a.in0, tile) * 32 + j]; mean = s0.x; m2 = s0.y; }
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
    // Narrow run, small launches (PRE): a stage is one or two k16-steps, so the first step of every chain is an exposed L2
    // round trip unless its planes are requested before the previous stage's arithmetic: request all six of them now.  (Costs
    // ~40 VGPRs: the large-launch form of the narrow kernel keeps four waves per SIMD instead.)
    // entry of the time table: the step index (sampling) or ts[row] (training) -- the same for every block of a run: the fused
    // narrow kernel reads it once (entry_pre); otherwise two DEPENDENT round trips (index, then the row) sit behind stage 1
    int entry = entry_pre;
    if (entry < 0) {
        entry = 0;
        if (a.ts) {
            int row = ptile * 32 + j;
            row = row < a.nrows ? row : a.nrows - 1;
            entry = a.ts[row];
        } else if (a.step_ptr) {
            entry = *a.step_ptr;
        }
    }
    HFrag<NT> p1a, p1b, p2, p3, psa, psb;
    float4 vtb[NT * 4], vc2[NT * 4], vc3[NT * 4], vcp[NT * 4];
    if (PRE) {
        // the per-feature vectors and the condition embedding as well: each was a round trip of its own right where it is added
        load_vec4<NT, NG>(vtb, a.tbias + (size_t)entry * a.tb_stride, h);
        load_vec4<NT, NG>(vc2, a.c2, h);
        load_vec4<NT, NG>(vc3, a.c3, h);
    }
    if (PRE || PC == 1) {
        // PC (round 6, the LDS form of the narrow run): the block's condition embedding is a row-dependent stream from HBM (329 MB per step
        // over all blocks: nothing of it is cached) -- requested here, two stages before it is added
        if (tile >= a.uncond_tiles) {
            const float* cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
#pragma unroll
            for (int G = 0; G < NG; ++G) vcp[G] = ld4(cp + (size_t)G * 256);
        }
    }
    if (PRE) {
        const size_t s1 = (size_t)KS1 * 128, s2 = (size_t)(((N + 7) / 8 + 1) / 2) * 128;
        load_hfrag<NT>(p1a, ah.W1h + lane, s1);
        if (a.in1.groups) load_hfrag<NT>(p1b, ah.W1h + (size_t)ks0 * 128 + lane, s1);
        load_hfrag<NT>(p2, ah.W2h + lane, s2);
        load_hfrag<NT>(p3, ah.W3h + lane, s2);
        if (SCLIN) {
            load_hfrag<NT>(psa, ah.Wsch + lane, s1);
            if (a.in1.groups) load_hfrag<NT>(psb, ah.Wsch + (size_t)ks0 * 128 + lane, s1);
        }
    }

    // Round 6, the LDS form of the narrow run: SK = 1 requests the skip input of a 16-wide up block AHEAD of its two chains (at the block
    // top for stage 1, under stage 3 for the shortcut) -- row-dependent reads that were issued right where the chains start.  The 32-wide
    // up block has no 16 registers to spare (80 B of scratch, slower); for it a "Linear shortcut first, skip read once" order was built and
    // measured: -7 % in the isolated block (tools/ubench/narrow_block.hip), +3 % in the kernel (profiles/r06_narrow_block_forms.txt): not kept.
    constexpr bool SKP = SK == 1 && XIN && N == 16;
    SegPre skp;
    if (SKP && a.in1.groups) seg_prefetch<(N == 16 ? NG : 1)>(skp, a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 + lane * 4);
    DSG_STAMP(XIN && tile == 0, 0x12);
    // ---- stage 1
    f32x16 acc1[NT];
    if (!XIN) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nt][r] = 0.f;
    }
    {
        const size_t nt_stride = (size_t)KS1 * 128;
        if (XIN)    // in0 has the block's own width N here (down / middle: in = N; up: cat(N, N))
            chain_from_acc_h<N, NT, NT, true>(acc1, *xr, ah.W1h, gamma1, beta1, mean1, rstd1, lane, h, nt_stride, PRE ? &p1a : nullptr);
        else
            chain_from_mem_h<NT, true, (N <= 32 ? 2 : 0)>(acc1, a.in0.data + (size_t)seg_tile(a.in0, tile) * a.in0.groups * 256 + lane * 4, a.in0.groups, ah.W1h + lane, nt_stride,
                                       gamma1 + 4 * h, beta1 + 4 * h, mean1, rstd1);
        if (a.in1.groups)
            chain_from_mem_h<NT, true, (N <= 32 ? 2 : 0), (XIN && N <= 32 ? NG : 0)>(acc1, a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 + lane * 4, a.in1.groups,
                                       ah.W1h + (size_t)ks0 * 128 + lane, nt_stride, gamma1 + 8 * a.in0.groups + 4 * h,
                                       beta1 + 8 * a.in0.groups + 4 * h, mean1, rstd1, PRE ? &p1b : nullptr, SKP ? &skp : nullptr);
        DSG_STAMP(XIN && tile == 0, 0x13);
        if (PRE) acc_unscale_add_reg<NT, NG>(acc1, inv1, vtb);
        else acc_unscale_add<NT, NG>(acc1, inv1, a.tbias + (size_t)entry * a.tb_stride, h);
    }
    DSG_STAMP(XIN && tile == 0, 0x14);
    if (a.save_h1) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h1 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc1[G >> 2][4 * (G & 3)], acc1[G >> 2][4 * (G & 3) + 1], acc1[G >> 2][4 * (G & 3) + 2],
                            acc1[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 2
    f32x16 acc2[NT];
    if (PC == 2) {      // the 32-wide blocks of the LDS form: no registers for the embedding during stage 1 -- requested here, one stage ahead
        if (tile >= a.uncond_tiles) {
            const float* cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
#pragma unroll
            for (int G = 0; G < NG; ++G) vcp[G] = ld4(cp + (size_t)G * 256);
        }
    }
    {
        float mean, m2;
        acc_stats<N, NT>(acc1, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        chain_from_acc_h<N, NT, NT, true>(acc2, acc1, ah.W2h, gamma2, beta2, mean, rstd, lane, h, 0, PRE ? &p2 : nullptr);
        DSG_STAMP(XIN && tile == 0, 0x15);
        if (PRE) acc_unscale_add_reg<NT, NG>(acc2, inv2, vc2);
        else acc_unscale_add<NT, NG>(acc2, inv2, a.c2, h);
    }
    if (tile >= a.uncond_tiles) {
        const float* cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            const float4 cv = (PRE || PC != 0) ? vcp[G] : ld4(cp + (size_t)G * 256);
            acc2[G >> 2][4 * (G & 3) + 0] += cv.x; acc2[G >> 2][4 * (G & 3) + 1] += cv.y;
            acc2[G >> 2][4 * (G & 3) + 2] += cv.z; acc2[G >> 2][4 * (G & 3) + 3] += cv.w;
        }
    }
    if (a.save_h2) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h2 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc2[G >> 2][4 * (G & 3)], acc2[G >> 2][4 * (G & 3) + 1], acc2[G >> 2][4 * (G & 3) + 2],
                            acc2[G >> 2][4 * (G & 3) + 3]));
    }

    DSG_STAMP(XIN && tile == 0, 0x16);
    // ---- stage 3 (+ shortcut in the same scaled accumulator)
    f32x16 (&acc3)[NT] = acc1;
    if (SKP && SCLIN && a.in1.groups)       // the shortcut's read of the skip input: on its way under stage 3
        seg_prefetch<(N == 16 ? NG : 1)>(skp, a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 + lane * 4);
    {
        float mean, m2;
        acc_stats<N, NT>(acc2, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        chain_from_acc_h<N, NT, NT, true>(acc3, acc2, ah.W3h, gamma3, beta3, mean, rstd, lane, h, 0, PRE ? &p3 : nullptr);
    }
    DSG_STAMP(XIN && tile == 0, 0x17);
    if (SCLIN) {
        const size_t nt_stride = (size_t)KS1 * 128;
        if (XIN)
            chain_raw_from_reg_h<NT>(acc3, *xr, a.in0.groups, ah.Wsch, nt_stride, lane, PRE ? &psa : nullptr);
        else
            chain_from_mem_h<NT, false, (N <= 32 ? 2 : 0)>(acc3, a.in0.data + (size_t)seg_tile(a.in0, tile) * a.in0.groups * 256 + lane * 4, a.in0.groups, ah.Wsch + lane, nt_stride,
                                        nullptr, nullptr, 0.f, 1.f);
        if (a.in1.groups)
            chain_from_mem_h<NT, false, (N <= 32 ? 2 : 0), (XIN && N <= 32 ? NG : 0)>(acc3, a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 + lane * 4, a.in1.groups,
                                        ah.Wsch + (size_t)ks0 * 128 + lane, nt_stride, nullptr, nullptr, 0.f, 1.f, PRE ? &psb : nullptr, SKP ? &skp : nullptr);
        if (PRE) acc_unscale_add_reg<NT, NG>(acc3, inv3, vc3);
        else acc_unscale_add<NT, NG>(acc3, inv3, a.c3, h);
    } else {
        if (PRE) acc_unscale_add_reg<NT, NG>(acc3, inv3, vc3);
        else acc_unscale_add<NT, NG>(acc3, inv3, a.c3, h);
        if (XIN) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc3[nt] += (*xr)[nt];
        } else {
            const float* xp = a.in0.data + (size_t)seg_tile(a.in0, tile) * NG * 256 + lane * 4;
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                const float4 xv = ld4(xp + (size_t)G * 256);
                acc3[G >> 2][4 * (G & 3) + 0] += xv.x; acc3[G >> 2][4 * (G & 3) + 1] += xv.y;
                acc3[G >> 2][4 * (G & 3) + 2] += xv.z; acc3[G >> 2][4 * (G & 3) + 3] += xv.w;
            }
        }
    }

    DSG_STAMP(XIN && tile == 0, 0x18);
    // ---- statistics + store (XREG: hand the tensor on in registers; store only if something else reads it)
    {
        float mean, m2;
        acc_stats<N, NT>(acc3, h, mean, m2);
        if (XOUT) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) (*xr)[nt] = acc3[nt];
            *xr_mean = mean; *xr_m2 = m2;
        }
        DSG_STAMP(XIN && tile == 0, 0x19);
        if (!XOUT || store_out) {
            if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(mean, m2);
#pragma unroll
            for (int G = 0; G < NG; ++G)
                st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                    make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2],
                                acc3[G >> 2][4 * (G & 3) + 3]));
        }
    }
}

// Floats of gamma1 / beta1 a block's stage 1 can touch: whole k16-steps of each input segment (the odd group of in0 reads
// on into in1's vector, or - without in1 - into the zero padding of the packed buffer; both finite, both times zero weights)
__device__ __forceinline__ int ln1_extent(const BlockArgs& a) {
    return a.in1.groups ? 8 * a.in0.groups + 16 * ((a.in1.groups + 1) >> 1) : 16 * ((a.in0.groups + 1) >> 1);
}

// Copy the block's three (gamma, beta) pairs into LDS: [gamma1 | beta1] kLnLdsW1 floats each, then [gamma2 | beta2 |
// gamma3 | beta3] kLnLdsN each.  n1 = the floats stage 1 can touch (8 * groups of in0 + 16 * steps of in1).
__device__ __forceinline__ void stage_ln_params(float* __restrict__ lds, const BlockArgs& a, int N) {
    const int n1 = ln1_extent(a), n2 = 16 * (((N + 7) / 8 + 1) / 2);
    for (int i = threadIdx.x; i < n1; i += blockDim.x) { lds[i] = a.gamma1[i]; lds[kLnLdsW1 + i] = a.beta1[i]; }
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        float* q = lds + 2 * kLnLdsW1;
        q[i] = a.gamma2[i]; q[kLnLdsN + i] = a.beta2[i]; q[2 * kLnLdsN + i] = a.gamma3[i]; q[3 * kLnLdsN + i] = a.beta3[i];
    }
    __syncthreads();
}

template <int N, bool SCLIN>
__global__ __launch_bounds__(256, 2) void k_resblock_h(const BlockArgsH ah) {
    __shared__ float lnp[2 * kLnLdsW1 + 4 * kLnLdsN];
    stage_ln_params(lnp, ah.b, N);
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= ah.b.ntiles) return;
    resblock_body_h<N, SCLIN, false, false, false, true>(ah, tile, lane, nullptr, nullptr, nullptr, true, lnp);
}

// ---------------------------------------------------------------------------------------------
// Cooperative form of the wide blocks for SMALL launches (fewer row tiles than SIMDs: 512-row evaluation chunks, the
// 8 192-row BASELINE config, 32 768-row training steps).  One wave per tile leaves most of the chip idle there and a block
// is then as slow as ONE wave walking through all of it (VALU and MFMA of a wave do not overlap).  Here the N/32 waves of
// a tile each own one 32-feature slice of every stage: a wave transforms (LayerNorm, SiLU, hi/lo split) only its share
// of a stage's input and publishes the operand halves in LDS, all waves read all of them (lane-contiguous b128, the MFMA
// B-operand image as it is) and multiply by their own slice of the weights; the LayerNorm statistics of the next stage
// are merged from the per-slice (mean, M2) with Chan's formula.  Same packed weights, same HBM layouts, same arithmetic per
// element; only the order of the additions inside a row statistic differs from k_resblock_h.
// Workgroup = 4 waves = one 128-wide tile or two 64-wide tiles.
// ---------------------------------------------------------------------------------------------
constexpr int kCoopLdsU4 = 4096;      // 64 KiB: per tile slot the transformed image and (Linear shortcut) the raw image

__device__ __forceinline__ void coop_publish(uint4* __restrict__ img, int S, int lane, const h8 hi, const h8 lo) {
    img[(size_t)(2 * S) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    img[(size_t)(2 * S + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}
// A wave's share of a stage is 3 MFMAs per k16-step: far too little to hide an L2 round trip per step, so ALL weight planes
// of a stage are requested up front (before the barrier that publishes the operands: the fetch overlaps the other waves'
// transforms) and the chain then runs out of registers.  KSM = compile-time bound of the step count.
template <int KSM>
__device__ __forceinline__ void coop_load_w(HFrag<1> (&wf)[KSM], const uint4* __restrict__ wp, int KS, int S0 = 0) {
#pragma unroll
    for (int S = 0; S < KSM; ++S)
        if (S0 + S < KS) load_hfrag<1>(wf[S], wp + (size_t)(S0 + S) * 128, 0);
}
template <int KSM>
__device__ __forceinline__ void coop_mma(f32x16 (&acc)[1], const uint4* __restrict__ img, const HFrag<1> (&wf)[KSM], int KS, int lane, int S0 = 0) {
#pragma unroll
    for (int S = 0; S < KSM; ++S)
        if (S0 + S < KS) {
            const h8 bhi = __builtin_bit_cast(h8, img[(size_t)(2 * (S0 + S)) * 64 + lane]), blo = __builtin_bit_cast(h8, img[(size_t)(2 * (S0 + S) + 1) * 64 + lane]);
            mfma_step_h<1>(acc, wf[S], bhi, blo);
        }
}
// A chain of up to 2 * KSM steps out of KSM plane registers: behind the MFMAs of step S the planes of step S + KSM are requested into the
// registers step S has just released, so the second batch arrives while the first is still being multiplied (the plain form asked for
// it behind the LAST step of the first batch: an exposed L2 round trip of 8 KiB per wave in the middle of every 2N-wide chain).
template <int KSM>
__device__ __forceinline__ void coop_mma_refill(f32x16 (&acc)[1], const uint4* __restrict__ img, HFrag<1> (&wf)[KSM], const uint4* __restrict__ wp,
                                                int KS, int lane) {
#pragma unroll
    for (int S = 0; S < KSM; ++S)
        if (S < KS) {
            const h8 bhi = __builtin_bit_cast(h8, img[(size_t)(2 * S) * 64 + lane]), blo = __builtin_bit_cast(h8, img[(size_t)(2 * S + 1) * 64 + lane]);
            mfma_step_h<1>(acc, wf[S], bhi, blo);
            if (S + KSM < KS) load_hfrag<1>(wf[S], wp + (size_t)(S + KSM) * 128, 0);
        }
#pragma unroll
    for (int S = 0; S < KSM; ++S)
        if (S + KSM < KS) {
            const h8 bhi = __builtin_bit_cast(h8, img[(size_t)(2 * (S + KSM)) * 64 + lane]), blo = __builtin_bit_cast(h8, img[(size_t)(2 * (S + KSM) + 1) * 64 + lane]);
            mfma_step_h<1>(acc, wf[S], bhi, blo);
        }
}

// A chain of KSM steps whose plane registers are needed again right behind it (the next out tile of the same stage, the next stage): behind
// the MFMAs of step S the planes of the NEXT chain's step S go into the registers step S has released -- requested KSM - 1 steps before
// the next chain starts instead of behind this chain's last MFMA.  (k_resblock_bwd_c, round 5: with the request behind the chain and 256
// registers in use hipcc reloaded every plane of the following chain right in front of its MFMAs -- sixteen exposed round trips.)
template <int KSM>
__device__ __forceinline__ void coop_mma_reload(f32x16 (&acc)[1], const uint4* __restrict__ img, HFrag<1> (&wf)[KSM], const uint4* __restrict__ wp_next,
                                                int lane) {
#pragma unroll
    for (int S = 0; S < KSM; ++S) {
        const h8 bhi = __builtin_bit_cast(h8, img[(size_t)(2 * S) * 64 + lane]), blo = __builtin_bit_cast(h8, img[(size_t)(2 * S + 1) * 64 + lane]);
        mfma_step_h<1>(acc, wf[S], bhi, blo);
        load_hfrag<1>(wf[S], wp_next + (size_t)S * 128, 0);
    }
}

// (mean, M2) of a row over N = 32 * NT features from the per-slice (mean, M2) in `st` (Chan, equal counts)
template <int NT>
__device__ __forceinline__ void coop_merge_stats(const float2* __restrict__ st /* [NT][32] of this tile slot */, int j, float& mean, float& m2) {
    float2 p[NT];
    float ms = 0.f;
#pragma unroll
    for (int w = 0; w < NT; ++w) { p[w] = st[w * 32 + j]; ms += p[w].x; }
    mean = ms * (1.0f / NT);
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < NT; ++w) { const float d = p[w].x - mean; q += p[w].y + 32.0f * d * d; }
    m2 = q;
}

// workgroup barriers inside resblock_coop_body: a wave that sits a block out (k_unet_tile: the upper waves of a 64-wide block) meets
// exactly these and nothing else
constexpr int kCoopBarriers = 7;

// The body: `tile_raw` = the tile of this wave's slot (>= ntiles: an idle slot that only meets the barriers and stores nothing),
// `slot` / `w` = the wave's tile slot and 32-feature slice, `img` / `stats` = the workgroup's LDS (kCoopLdsU4 uint4, 4 x 32 float2).
// `vlds` (7 * N floats of LDS): the per-feature vectors every stage reads -- gamma2 | beta2 | gamma3 | beta3 | c2 | c3 | this step's
// time-bias row -- staged by the workgroup at the top and read from LDS behind the first barrier.  Round 5 (tools/tile_stamps.py): read
// from global memory right where they are used, each was an exposed L2 round trip on the tile's critical path (the transforms of stages
// 2 and 3 took 5x their arithmetic); likewise the condition embedding is requested a stage early and the second batch of the planes
// of a 2N-wide input refills the registers of the first behind each step's MFMAs.
template <int N, bool SCLIN>
__device__ __forceinline__ void resblock_coop_body(const BlockArgsH& ah, const int tile_raw, const int slot, const int w, uint4* __restrict__ img,
                                                   float2* __restrict__ stats, float* __restrict__ vlds) {
    constexpr int NG = N / 8, NT = N / 32, TPW = 4 / NT, KS = NG / 2;
    constexpr int kSlotU4 = kCoopLdsU4 / TPW;
    const BlockArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const bool live = tile_raw < a.ntiles;            // idle slots of the last workgroup still meet the barriers
    const int tile = live ? tile_raw : a.ntiles - 1;
    const int ptile = tile % a.tiles_per_pass;
    uint4* const Bimg = img + slot * kSlotU4;
    uint4* const Rimg = Bimg + kSlotU4 / 2;
    float2* const st = stats + slot * NT * 32;
    // The input tensors are exactly N wide (in0, and in1 of an up block's concat): the host launches this body for those shapes only
    // (launch_res_h, the tile table), so the k16-step counts are compile-time constants.  With run-time counts every plane load of
    // coop_load_w / coop_mma_refill is conditional, hipcc cannot count the loads in flight, and each k16-step of a refilled chain waited
    // with vmcnt(0) -- for the REFILL requested one step earlier, a full L2 round trip per step (round 5, disassembly).
    constexpr int ks0 = NG / 2, ks1 = SCLIN ? NG / 2 : 0, KS1 = ks0 + ks1;
    if (a.in0.groups != NG || a.in1.groups != (SCLIN ? NG : 0)) __builtin_trap();
    // (`vlds` is NOT optional: a pointer that is "the LDS copy or the global original" is a generic pointer to hipcc, every read through
    // it a FLAT instruction, and every wait behind one drains the wave's global loads as well -- the weight planes in flight.  Round 5,
    // tools/isa_lint.py: 28 FLAT loads per block in k_resblock_c, 224 in k_unet_tile, each followed by vmcnt(0) lgkmcnt(0).)
    const bool tb_lds = !a.ts;                          // one time-bias row for the whole launch (sampling); per-row entries stay in memory
    {
        // thread t stages one float4: 7 vectors x N / 4 quads (N = 128: 224 threads; N = 64: 112, the two waves that run a 64-wide block)
        constexpr int per = N / 4;
        const int t = threadIdx.x, v = t / per, o = (t % per) * 4;
        if (t < 7 * per && (v < 6 || tb_lds)) {
            const float* src = v == 0 ? a.gamma2 : v == 1 ? a.beta2 : v == 2 ? a.gamma3 : v == 3 ? a.beta3 : v == 4 ? a.c2 : v == 5 ? a.c3
                                      : a.tbias + (size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride;
            *reinterpret_cast<float4*>(vlds + v * N + o) = ld4(src + o);
        }
    }
    const float* const g2p = vlds, * const b2p = vlds + N, * const g3p = vlds + 2 * N, * const b3p = vlds + 3 * N, * const c2p = vlds + 4 * N,
               * const c3p = vlds + 5 * N;

    // ---- LN1 statistics (Chan merge of the producers' (mean, M2)), as k_resblock_h
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
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- stage 1 operands: this wave transforms k16-steps w, w + NT, ... of the (concatenated) input
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2001);
    constexpr int KS1M = SCLIN ? NG : NG / 2;           // concat input of an up block: 2N wide
    constexpr int KSB = KS1M < 8 ? KS1M : 8;            // planes in flight per batch (8 steps = 64 VGPRs)
    HFrag<1> wf1[KSB];
    float4 cvp[4] = {z4, z4, z4, z4};                   // condition embedding of this wave's slice (conditional tiles), added behind stage 2
    {
        const float c = rstd1, d = -mean1 * rstd1;
        constexpr int MY = KS1M / NT;                   // k16-steps per wave (upper bound)
        float4 x0[MY], x1[MY], g0[MY], b0[MY], g1[MY], b1[MY];
#pragma unroll
        for (int i = 0; i < MY; ++i) {                  // every load of the stage first
            const int S = w + i * NT;
            x0[i] = z4; x1[i] = z4; g0[i] = z4; b0[i] = z4; g1[i] = z4; b1[i] = z4;
            if (S < KS1) {
                const bool first = S < ks0;
                const Seg& sg = first ? a.in0 : a.in1;
                const int Sl = first ? S : S - ks0;
                const float* xp = sg.data + (size_t)seg_tile(sg, tile) * sg.groups * 256 + lane * 4;
                x0[i] = ld4(xp + (size_t)(2 * Sl) * 256);
                if (2 * Sl + 1 < sg.groups) x1[i] = ld4(xp + (size_t)(2 * Sl + 1) * 256);
                const int gbase = (first ? 0 : 8 * a.in0.groups) + 16 * Sl + 4 * h;
                g0[i] = ld4(a.gamma1 + gbase); b0[i] = ld4(a.beta1 + gbase); g1[i] = ld4(a.gamma1 + gbase + 8); b1[i] = ld4(a.beta1 + gbase + 8);
            }
        }
        // the stage's planes BEHIND the tile's own inputs (loads return in order: the transform below needs only the inputs)
        coop_load_w<KSB>(wf1, ah.W1h + (size_t)w * KS1 * 128 + lane, KS1);
        if (tile >= a.uncond_tiles) {
            const float* cp = a.cond_pre + ((size_t)ptile * NG + 4 * w) * 256 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) cvp[q] = ld4(cp + (size_t)q * 256);
        }
#pragma unroll
        for (int i = 0; i < MY; ++i) {
            const int S = w + i * NT;
            if (S < KS1) {
                float v[8];
                act8(v, x0[i], x1[i], c, d, g0[i], b0[i], g1[i], b1[i]);
                h8 hi, lo;
                split8(v, hi, lo);
                coop_publish(Bimg, S, lane, hi, lo);
                if (SCLIN) {
                    const float r[8] = {kRawScale * x0[i].x, kRawScale * x0[i].y, kRawScale * x0[i].z, kRawScale * x0[i].w,
                                        kRawScale * x1[i].x, kRawScale * x1[i].y, kRawScale * x1[i].z, kRawScale * x1[i].w};
                    split8(r, hi, lo);
                    coop_publish(Rimg, S, lane, hi, lo);
                }
            }
        }
    }
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2002);
    __syncthreads();
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2003);

    // ---- stage 1: this wave's 32 output features
    f32x16 acc1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
    if (KS1M > KSB) coop_mma_refill<KSB>(acc1, Bimg, wf1, ah.W1h + (size_t)w * KS1 * 128 + lane, KS1, lane);   // 2N-wide input: two batches
    else coop_mma<KSB>(acc1, Bimg, wf1, KS1, lane);
    HFrag<1> wf2[KS];                                   // next stage's planes: requested now, used after two barriers
    coop_load_w<KS>(wf2, ah.W2h + (size_t)w * KS * 128 + lane, KS);
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2004);
    {
        int entry = 0;
        if (a.ts) {
            int row = ptile * 32 + j;
            row = row < a.nrows ? row : a.nrows - 1;
            entry = a.ts[row];
        } else if (a.step_ptr) {
            entry = *a.step_ptr;
        }
        if (tb_lds) {
            // through an explicit LDS pointer: as two calls of acc_unscale_add hipcc merged the branches into one body behind a select of
            // the two pointers -- a generic pointer again, four FLAT loads
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) const f32x4_ lds_f4;
            lds_f4* const tp = (lds_f4*)(vlds + 6 * N + 32 * w + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_ b = tp[2 * q];
                acc1[0][4 * q + 0] = fmaf(acc1[0][4 * q + 0], inv1, b.x); acc1[0][4 * q + 1] = fmaf(acc1[0][4 * q + 1], inv1, b.y);
                acc1[0][4 * q + 2] = fmaf(acc1[0][4 * q + 2], inv1, b.z); acc1[0][4 * q + 3] = fmaf(acc1[0][4 * q + 3], inv1, b.w);
            }
        } else {
            acc_unscale_add<1>(acc1, inv1, a.tbias + (size_t)entry * a.tb_stride + 32 * w, h);
        }
    }
    if (a.save_h1 && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            st4(a.save_h1 + ((size_t)tile * NG + 4 * w + q) * 256 + lane * 4,
                make_float4(acc1[0][4 * q], acc1[0][4 * q + 1], acc1[0][4 * q + 2], acc1[0][4 * q + 3]));
    }
    {
        float m, q;
        acc_stats<32, 1>(acc1, h, m, q);
        if (h == 0) st[w * 32 + j] = make_float2(m, q);
    }
    __syncthreads();                 // statistics published; every wave is also done reading the stage-1 image
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2005);

    // ---- stage 2
    f32x16 acc2[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][r] = 0.f;
    {
        float mean, m2;
        coop_merge_stats<NT>(st, j, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        const float c = rstd, d = -mean * rstd;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int S = 2 * w + half, r0 = 8 * half;
            const float4 g0 = ld4(g2p + 16 * S + 4 * h), b0 = ld4(b2p + 16 * S + 4 * h);
            const float4 g1 = ld4(g2p + 16 * S + 8 + 4 * h), b1 = ld4(b2p + 16 * S + 8 + 4 * h);
            float v[8];
            act8(v, make_float4(acc1[0][r0], acc1[0][r0 + 1], acc1[0][r0 + 2], acc1[0][r0 + 3]),
                 make_float4(acc1[0][r0 + 4], acc1[0][r0 + 5], acc1[0][r0 + 6], acc1[0][r0 + 7]), c, d, g0, b0, g1, b1);
            h8 hi, lo;
            split8(v, hi, lo);
            coop_publish(Bimg, S, lane, hi, lo);
        }
    }
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2006);
    __syncthreads();
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2007);
    coop_mma<KS>(acc2, Bimg, wf2, KS, lane);
    HFrag<1> wf3[KS];
    coop_load_w<KS>(wf3, ah.W3h + (size_t)w * KS * 128 + lane, KS);
    HFrag<1> wfs[SCLIN ? KSB : 1];
    if (SCLIN) coop_load_w<(SCLIN ? KSB : 1)>(wfs, ah.Wsch + (size_t)w * KS1 * 128 + lane, KS1);
    acc_unscale_add<1>(acc2, inv2, c2p + 32 * w, h);
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2008);
    if (tile >= a.uncond_tiles) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 cv = cvp[q];
            acc2[0][4 * q + 0] += cv.x; acc2[0][4 * q + 1] += cv.y; acc2[0][4 * q + 2] += cv.z; acc2[0][4 * q + 3] += cv.w;
        }
    }
    if (a.save_h2 && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            st4(a.save_h2 + ((size_t)tile * NG + 4 * w + q) * 256 + lane * 4,
                make_float4(acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]));
    }
    {
        float m, q;
        acc_stats<32, 1>(acc2, h, m, q);
        if (h == 0) st[w * 32 + j] = make_float2(m, q);
    }
    __syncthreads();
    DSG_STAMP(tile_raw == 0 && w == 0, 0x2009);

    // ---- stage 3 (+ shortcut in the same scaled accumulator)
    f32x16 (&acc3)[1] = acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[0][r] = 0.f;
    {
        float mean, m2;
        coop_merge_stats<NT>(st, j, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        const float c = rstd, d = -mean * rstd;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int S = 2 * w + half, r0 = 8 * half;
            const float4 g0 = ld4(g3p + 16 * S + 4 * h), b0 = ld4(b3p + 16 * S + 4 * h);
            const float4 g1 = ld4(g3p + 16 * S + 8 + 4 * h), b1 = ld4(b3p + 16 * S + 8 + 4 * h);
            float v[8];
            act8(v, make_float4(acc2[0][r0], acc2[0][r0 + 1], acc2[0][r0 + 2], acc2[0][r0 + 3]),
                 make_float4(acc2[0][r0 + 4], acc2[0][r0 + 5], acc2[0][r0 + 6], acc2[0][r0 + 7]), c, d, g0, b0, g1, b1);
            h8 hi, lo;
            split8(v, hi, lo);
            coop_publish(Bimg, S, lane, hi, lo);
        }
    }
    DSG_STAMP(tile_raw == 0 && w == 0, 0x200a);
    __syncthreads();
    DSG_STAMP(tile_raw == 0 && w == 0, 0x200b);
    coop_mma<KS>(acc3, Bimg, wf3, KS, lane);
    if (SCLIN) {
        if (KS1M > KSB) coop_mma_refill<(SCLIN ? KSB : 1)>(acc3, Rimg, wfs, ah.Wsch + (size_t)w * KS1 * 128 + lane, KS1, lane);
        else coop_mma<(SCLIN ? KSB : 1)>(acc3, Rimg, wfs, KS1, lane);
        acc_unscale_add<1>(acc3, inv3, c3p + 32 * w, h);
    } else {
        acc_unscale_add<1>(acc3, inv3, c3p + 32 * w, h);
        const float* xp = a.in0.data + ((size_t)seg_tile(a.in0, tile) * NG + 4 * w) * 256 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xv = ld4(xp + (size_t)q * 256);
            acc3[0][4 * q + 0] += xv.x; acc3[0][4 * q + 1] += xv.y; acc3[0][4 * q + 2] += xv.z; acc3[0][4 * q + 3] += xv.w;
        }
    }
    // ---- output statistics (merged by the first wave of the tile) + store
    DSG_STAMP(tile_raw == 0 && w == 0, 0x200c);
    {
        float m, q;
        acc_stats<32, 1>(acc3, h, m, q);
        __syncthreads();             // the stage-3 statistics in `st` have been read by everyone
        if (h == 0) st[w * 32 + j] = make_float2(m, q);
    }
    __syncthreads();
    DSG_STAMP(tile_raw == 0 && w == 0, 0x200d);
    if (live) {
        if (w == 0 && h == 0) {
            float mean, m2;
            coop_merge_stats<NT>(st, j, mean, m2);
            reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(mean, m2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            st4(a.out + ((size_t)tile * NG + 4 * w + q) * 256 + lane * 4,
                make_float4(acc3[0][4 * q], acc3[0][4 * q + 1], acc3[0][4 * q + 2], acc3[0][4 * q + 3]));
    }
}

template <int N, bool SCLIN>
__global__ __launch_bounds__(256) void k_resblock_c(const BlockArgsH ah) {
    constexpr int NT = N / 32, TPW = 4 / NT;
    __shared__ uint4 img[kCoopLdsU4];
    __shared__ float2 stats[4 * 32];
    __shared__ float4 vecs[7 * N / 4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave / NT, w = wave % NT;
    resblock_coop_body<N, SCLIN>(ah, blockIdx.x * TPW + slot, slot, w, img, stats, reinterpret_cast<float*>(vecs));
}

// A wide block followed by the Linear that consumes it (Down/Upsample: raw; final: LayerNorm + SiLU, row-major out),
// fused: the block's output stays in registers, which saves one launch, one store (unless it is a skip) and one reload.
struct LinArgsH;
// Row statistics (mean, M2 over the true width) of a Linear's output held in NT accumulator tiles.  With a run-time width every
// element is a compare + add + select and the whole sum ONE dependent chain of selects (disassembly of k_fused_narrow_lds, round 5:
// 5 instructions per element, 64 elements per lane at NT = 4, twice).  A width that is a multiple of 8 makes the mask a function of the
// 8-feature group alone: W > 0 is that compile-time form -- same elements in the same order, same bits.  W = 0: run-time width.
template <int NT, int W>
__device__ __forceinline__ void lin_out_stats_w(const f32x16 (&acc)[NT], const int h, const int width, const float inv_w, float& m, float& q) {
    float s = 0.f;
#pragma unroll
    for (int G = 0; G < NT * 4; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (W > 0 ? 8 * G < W : 8 * G + 4 * h + p < width) s += acc[G >> 2][4 * (G & 3) + p];
    m = xhalf_sum(s) * inv_w;
    float qq = 0.f;
#pragma unroll
    for (int G = 0; G < NT * 4; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (W > 0 ? 8 * G < W : 8 * G + 4 * h + p < width) { const float d = acc[G >> 2][4 * (G & 3) + p] - m; qq = fmaf(d, d, qq); }
    q = xhalf_sum(qq);
}
template <int NT>
__device__ __forceinline__ void lin_out_stats(const f32x16 (&acc)[NT], const int h, const int width, const float inv_w, float& m, float& q) {
    if (width == 32 * NT) lin_out_stats_w<NT, 32 * NT>(acc, h, width, inv_w, m, q);              // every shipped wide Linear
    else if (NT == 1 && width == 16) lin_out_stats_w<NT, (NT == 1 ? 16 : 0)>(acc, h, width, inv_w, m, q);
    else if (NT == 1 && width == 8) lin_out_stats_w<NT, (NT == 1 ? 8 : 0)>(acc, h, width, inv_w, m, q);
    else lin_out_stats_w<NT, 0>(acc, h, width, inv_w, m, q);
}

template <int NTO, int NTI, bool FINAL>
__device__ __forceinline__ void linear_epilogue_h(const LinArgsH& ah, int tile, int lane, const f32x16 (&x)[NTI], float xmean, float xm2);

// Plain Linear on the split path (feature_proj, Down/Upsample, final): same contract as linear_body.
struct LinArgsH {
    LinArgs l;
    const uint4* Wh;   // [NT][ceil(KG/2)][2][64]
    const float* m;    // max|W|
    const float* kc;   // device: { 2^-e (raw operand), 2^-e / 16 (LayerNorm + SiLU operand) } (written by k_pack_h)
};
template <bool LDS>      // LDS: Wh and the bias were redirected to an LDS image
__device__ __forceinline__ void globalize(LinArgsH& a) {
    globalize_io(a.l);
    a.m = as_global(a.m); a.kc = as_global(a.kc);
    a.l.W = as_global(a.l.W); a.l.gamma = as_global(a.l.gamma); a.l.beta = as_global(a.l.beta);
    if (!LDS) { a.l.bias = as_global(a.l.bias); a.Wh = as_global(a.Wh); }
}

// Bind-time constants of the split path: the un-scale factors of every block and Linear from max|W| (same arithmetic as the
// kernels used per wave before: scale_exp / scale_exp_lin3 + ldexp).
struct OpConstDesc { int w1, w2, w3, wsc; };   // parameter indices of lin1, lin2, lin3, shortcut (-1: none); Linear: w1 only (written by k_pack_h)

template <int NTO, int NTI, bool FINAL>
__device__ __forceinline__ void linear_epilogue_h(const LinArgsH& ah, int tile, int lane, const f32x16 (&x)[NTI], float xmean, float xm2) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int KS = (a.in_groups + 1) >> 1;
    const size_t nt_stride = (size_t)KS * 128;
    f32x16 acc[NTO];
    if (FINAL) {
        const float rstd = rsqrtf(xm2 * a.inv_in_w + kLnEps);
        chain_from_acc_h<NTI * 32, NTO, NTI, true>(acc, x, ah.Wh, a.gamma, a.beta, xmean, rstd, lane, h, nt_stride);
    } else {
        range_check(a.range_flag, xmean, xm2);
        chain_raw_from_reg_h<NTO, NTI, true>(acc, x, a.in_groups, ah.Wh, nt_stride, lane);
    }
    acc_unscale_add<NTO>(acc, ah.kc[FINAL ? 1 : 0], a.bias, h);
    if (!FINAL) {
        const int NG = (a.out_width + 7) / 8;
        float m, q;
        lin_out_stats<NTO>(acc, h, a.out_width, a.inv_out_w, m, q);
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
        for (int G = 0; G < NTO * 4; ++G)
            if (G < NG)
                st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                    make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
    } else {
        const int ptile = tile % a.tiles_per_pass, pass = tile / a.tiles_per_pass, row = ptile * 32 + j;
        if (row < a.nrows) {
            float* o = a.out_rm + ((size_t)pass * a.nrows + row) * a.out_width;
            if ((a.out_width & 3) == 0) {
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G) {
                    const int f = 8 * G + 4 * h;
                    if (f < a.out_width)
                        st4(o + f, make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
                }
            } else {
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int f = 8 * G + 4 * h + p;
                        if (f < a.out_width) o[f] = acc[G >> 2][4 * (G & 3) + p];
                    }
            }
        }
    }
}

struct BlockLinArgsH { BlockArgsH b; LinArgsH l; int store_block_out; int dbg; };   // dbg: dsg_wide.hpp measurement switches (0 in production)

template <int N, bool SCLIN, int NTO, bool FINAL>
__global__ __launch_bounds__(256, 2) void k_resblock_lin_h(const BlockLinArgsH a) {
    constexpr int NT = (N + 31) / 32;
    __shared__ float lnp[2 * kLnLdsW1 + 4 * kLnLdsN];
    stage_ln_params(lnp, a.b.b, N);
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= a.b.b.ntiles) return;
    f32x16 x[NT];
    float xmean = 0.f, xm2 = 0.f;
    resblock_body_h<N, SCLIN, false, true, false, true>(a.b, tile, lane, &x, &xmean, &xm2, a.store_block_out != 0, lnp);
    linear_epilogue_h<NTO, NT, FINAL>(a.l, tile, lane, x, xmean, xm2);
}

// DEEP (k_unet_tile): the fragment-input chain in batches of loads (chain_from_mem_h_batched)
template <int NT, int INMODE, int OUTMODE, bool LNACT, bool DEEP = false>
__device__ __forceinline__ void linear_body_h(const LinArgsH& ah, const int tile, const int lane) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int ptile = tile % a.tiles_per_pass;
    const int pass = tile / a.tiles_per_pass;
    const int row = ptile * 32 + j;
    const int KG = a.in_groups, KS = (KG + 1) >> 1;
    const size_t nt_stride = (size_t)KS * 128;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    float mean = 0.f, rstd = 1.f;
    if (LNACT) {
        const float2 s = reinterpret_cast<const float2*>(a.in.stats)[(size_t)seg_tile(a.in, tile) * 32 + j];
        mean = s.x;
        rstd = rsqrtf(s.y * a.inv_in_w + kLnEps);
    } else if (INMODE == IN_FRAG && a.range_flag && a.in.stats) {
        const float2 s = reinterpret_cast<const float2*>(a.in.stats)[(size_t)seg_tile(a.in, tile) * 32 + j];
        range_check(a.range_flag, s.x, s.y);
    }
    float vmax = 0.f;
    if (INMODE == IN_FRAG) {
        if constexpr (DEEP)
            chain_from_mem_h_batched<NT, LNACT, 2>(acc, a.in.data + (size_t)seg_tile(a.in, tile) * KG * 256 + lane * 4, KG, ah.Wh + lane,
                                                                          nt_stride, LNACT ? a.gamma + 4 * h : nullptr, LNACT ? a.beta + 4 * h : nullptr, mean, rstd);
        else
        chain_from_mem_h<NT, LNACT>(acc, a.in.data + (size_t)seg_tile(a.in, tile) * KG * 256 + lane * 4, KG, ah.Wh + lane, nt_stride,
                                    LNACT ? a.gamma + 4 * h : nullptr, LNACT ? a.beta + 4 * h : nullptr, mean, rstd);
    } else {
        // rows of 16-byte quads up to 128 wide (MSR-80c: 80): EVERY quad of the row is requested before the first step (round 5: with
        // one step's two loads issued per step, feature_proj was five dependent HBM round trips long -- 26 us at 65 536 rows for 55 MB)
        constexpr int kPreS = 8;
        if ((a.in_width & 3) == 0 && KS <= kPreS) {
            float4 xq[2 * kPreS];
#pragma unroll
            for (int q = 0; q < 2 * kPreS; ++q) {
                const int f = 8 * q + 4 * h;
                xq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < 2 * KS && row < a.nrows && f < a.in_width) xq[q] = ld4(a.in_rm + (size_t)row * a.in_width + f);
            }
#pragma unroll
            for (int S = 0; S < kPreS; ++S) {
                if (S < KS) {
                    HFrag<NT> wc;
                    load_hfrag<NT>(wc, ah.Wh + (size_t)S * 128 + lane, nt_stride);
                    const float4 x0 = xq[2 * S], x1 = xq[2 * S + 1];
                    const float v[8] = {kRawScale * x0.x, kRawScale * x0.y, kRawScale * x0.z, kRawScale * x0.w,
                                        kRawScale * x1.x, kRawScale * x1.y, kRawScale * x1.z, kRawScale * x1.w};
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) vmax = fmaxf(vmax, fabsf(v[jj]));
                    h8 bhi, blo;
                    split8(v, bhi, blo);
                    mfma_step_h<NT>(acc, wc, bhi, blo);
                }
            }
        } else
        for (int S = 0; S < KS; ++S) {
            HFrag<NT> wc;
            load_hfrag<NT>(wc, ah.Wh + (size_t)S * 128 + lane, nt_stride);
            float v[8];
            if ((a.in_width & 3) == 0) {   // 16-byte aligned quads: two float4 loads per lane and step
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = 8 * (2 * S + q) + 4 * h;
                    float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < a.nrows && f < a.in_width) x4 = ld4(a.in_rm + (size_t)row * a.in_width + f);
                    v[4 * q] = kRawScale * x4.x; v[4 * q + 1] = kRawScale * x4.y; v[4 * q + 2] = kRawScale * x4.z; v[4 * q + 3] = kRawScale * x4.w;
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int f = 8 * (2 * S + (jj >> 2)) + 4 * h + (jj & 3);
                    v[jj] = (row < a.nrows && f < a.in_width) ? kRawScale * a.in_rm[(size_t)row * a.in_width + f] : 0.f;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) vmax = fmaxf(vmax, fabsf(v[jj]));
            h8 bhi, blo;
            split8(v, bhi, blo);
            mfma_step_h<NT>(acc, wc, bhi, blo);
        }
        if (a.range_flag && vmax > kRawLimit * kRawScale) *a.range_flag = 1;
    }
    acc_unscale_add<NT>(acc, ah.kc[LNACT ? 1 : 0], a.bias, h);

    if (OUTMODE == OUT_FRAG) {
        const int NG = (a.out_width + 7) / 8;
        float m, q;
        lin_out_stats<NT>(acc, h, a.out_width, a.inv_out_w, m, q);
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
        for (int G = 0; G < NT * 4; ++G)
            if (G < NG)
                st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                    make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2],
                                acc[G >> 2][4 * (G & 3) + 3]));
    } else {
        if (row < a.nrows) {
            float* o = a.out_rm + ((size_t)pass * a.nrows + row) * a.out_width;
            if ((a.out_width & 3) == 0) {
#pragma unroll
                for (int G = 0; G < NT * 4; ++G) {
                    const int f = 8 * G + 4 * h;
                    if (f < a.out_width)
                        st4(o + f, make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2],
                                               acc[G >> 2][4 * (G & 3) + 3]));
                }
            } else {
#pragma unroll
                for (int G = 0; G < NT * 4; ++G)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int f = 8 * G + 4 * h + p;
                        if (f < a.out_width) o[f] = acc[G >> 2][4 * (G & 3) + p];
                    }
            }
        }
    }
}

template <int NT, int INMODE, int OUTMODE, bool LNACT>
__global__ __launch_bounds__(256) void k_linear_h(const LinArgsH a) {
    if (INMODE == IN_ROWMAJOR && a.l.advance_step && blockIdx.x == 0 && threadIdx.x == 0) *a.l.advance_step -= 1;
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= a.l.ntiles) return;
    linear_body_h<NT, INMODE, OUTMODE, LNACT>(a, tile, lane);
}

// The narrow run on the split path (see k_fused_narrow), with the running tensor kept in REGISTERS between operators:
// every operator here is at most 32 wide (one accumulator tile), so `x` + its (mean, M2) are 18 registers.  An operator
// stores its output only when `store_out` (skip tensors and the last operator); skip inputs still come from memory.
struct FusedOpH {
    int kind, N, sclin, store_out;
    BlockArgsH b;
    LinArgsH l;
};

// Linear with register input (in width <= 32) and register output (out width <= 32)
__device__ __forceinline__ void linear_reg_h(const LinArgsH& ah, const int tile, const int lane, f32x16 (&x)[1], float& xmean, float& xm2,
                                             bool store_out) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int KS = (a.in_groups + 1) >> 1;
    f32x16 acc[1];
    range_check(a.range_flag, xmean, xm2);
    chain_raw_from_reg_h<1, 1, true>(acc, x, a.in_groups, ah.Wh, (size_t)KS * 128, lane);
    acc_unscale_add<1>(acc, ah.kc[0], a.bias, h);
    const int NG = (a.out_width + 7) / 8;
    float m, q;
    lin_out_stats<1>(acc, h, a.out_width, a.inv_out_w, m, q);
    x[0] = acc[0]; xmean = m; xm2 = q;
    if (store_out) {
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
        for (int G = 0; G < 4; ++G)
            if (G < NG) st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4, make_float4(acc[0][4 * G], acc[0][4 * G + 1], acc[0][4 * G + 2], acc[0][4 * G + 3]));
    }
}

// Linear with register input (in width <= 32) and a 32 * NTO wide output that leaves the run: the Upsample behind the last narrow
// block (k_fused_narrow_lds, operator kind 2) -- always stored, with its row statistics
template <int NTO>
__device__ __forceinline__ void linear_reg_out_h(const LinArgsH& ah, const int tile, const int lane, const f32x16 (&x)[1], float xmean, float xm2) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int KS = (a.in_groups + 1) >> 1;
    f32x16 acc[NTO];
    range_check(a.range_flag, xmean, xm2);
    chain_raw_from_reg_h<NTO, 1, true>(acc, x, a.in_groups, ah.Wh, (size_t)KS * 128, lane);
    acc_unscale_add<NTO>(acc, ah.kc[0], a.bias, h);
    const int NG = (a.out_width + 7) / 8;
    float m, q;
    lin_out_stats<NTO>(acc, h, a.out_width, a.inv_out_w, m, q);
    if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
    for (int G = 0; G < 4 * NTO; ++G)
        if (G < NG)
            st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
}

// The Linear that OPENS the narrow run (Downsample 64 -> 32 of the shipped nets): input from a fragment tensor in memory (<= 8 groups), output
// (<= 32 wide) in registers.  Round 6 (profiles/r06_narrow_op_stamps.txt): as "linear_body_h, wait for its stores, read the result back" the
// operator took 20 000 cycles of a 175 000-cycle launch -- five dependent memory round trips (its k16-step loop prefetches one step ahead,
// then store -> reload) for a handful of MFMAs.  Here every input group is requested at once -- by the LDS kernel BEFORE it stages its image,
// so the loads fly under the staging -- and the result never leaves the registers (it is stored only because it is a skip tensor).
// Same operands, same products in the same order, same statistics as linear_body_h: same bits.
constexpr int kLinPreG = 8;
struct LinPre { float4 x[kLinPreG]; float2 st; };
__device__ __forceinline__ void linear_mem_prefetch(LinPre& p, const LinArgs& a /* globalized */, const int tile, const int lane) {
    const int j = lane & 31;
    const size_t t = (size_t)seg_tile(a.in, tile);
    p.st = make_float2(0.f, 0.f);
    if (a.in.stats) p.st = reinterpret_cast<const float2*>(a.in.stats)[t * 32 + j];
    const float* xp = a.in.data + t * a.in_groups * 256 + lane * 4;
#pragma unroll
    for (int G = 0; G < kLinPreG; ++G) {       // unconditional loads: a group that does not exist re-reads the last one (and is zeroed below)
        const int g = G < a.in_groups ? G : a.in_groups - 1;
        p.x[G] = ld4(xp + (size_t)g * 256);
    }
}
__device__ __forceinline__ void linear_pre_to_reg_h(const LinArgsH& ah, const LinPre& p, const int tile, const int lane, f32x16 (&x)[1], float& xmean,
                                                    float& xm2, const bool store_out) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int groups = a.in_groups, steps = (groups + 1) >> 1;
    const size_t nt_stride = (size_t)steps * 128;
    if (a.range_flag && a.in.stats) range_check(a.range_flag, p.st.x, p.st.y);
    f32x16 acc[1];
#pragma unroll
    for (int S = 0; S < kLinPreG / 2; ++S) {
        if (S < steps) {
            HFrag<1> w;
            load_hfrag<1>(w, ah.Wh + (size_t)S * 128 + lane, nt_stride);
            const float4 x0 = p.x[2 * S];
            float4 x1 = p.x[2 * S + 1];
            if (2 * S + 1 >= groups) x1 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float v[8] = {kRawScale * x0.x, kRawScale * x0.y, kRawScale * x0.z, kRawScale * x0.w,
                                kRawScale * x1.x, kRawScale * x1.y, kRawScale * x1.z, kRawScale * x1.w};
            h8 bhi, blo;
            split8(v, bhi, blo);
            if (S == 0) mfma_step_h0<1>(acc, w, bhi, blo);
            else mfma_step_h<1>(acc, w, bhi, blo);
        }
    }
    acc_unscale_add<1>(acc, ah.kc[0], a.bias, h);
    const int NG = (a.out_width + 7) / 8;
    float m, q;
    lin_out_stats<1>(acc, h, a.out_width, a.inv_out_w, m, q);
    x[0] = acc[0]; xmean = m; xm2 = q;
    if (store_out) {
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
        for (int G = 0; G < 4; ++G)
            if (G < NG) st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4, make_float4(acc[0][4 * G], acc[0][4 * G + 1], acc[0][4 * G + 2], acc[0][4 * G + 3]));
    }
}

// What the training forward keeps of the float32 section for the backward pass: every operator's output with its row statistics and the
// blocks' pre-LayerNorm tensors, at the places the matrix-core forms write them (8-wide fragment tensors: one float4 per lane).
struct V8TableSave {
    const FusedOpH* ops;          // the section's first operator (the Downsample Linear); block blk is ops[1 + blk]
    int tile, lane;
    bool on;
    __device__ __forceinline__ void put(float* dst, const float (&v)[4]) const {
        st4(as_global(dst) + (size_t)tile * 256 + lane * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
    __device__ __forceinline__ void stats(float* dst, float mean, float m2) const {
        if (lane < 32) reinterpret_cast<float2*>(as_global(dst))[(size_t)tile * 32 + lane] = make_float2(mean, m2);
    }
    __device__ __forceinline__ void h1(int blk, const float (&v)[4]) const { if (on && ops[1 + blk].b.b.save_h1) put(ops[1 + blk].b.b.save_h1, v); }
    __device__ __forceinline__ void h2(int blk, const float (&v)[4]) const { if (on && ops[1 + blk].b.b.save_h2) put(ops[1 + blk].b.b.save_h2, v); }
    __device__ __forceinline__ void out(int blk, const float (&v)[4], float mean, float m2) const {
        if (on) { put(ops[1 + blk].b.b.out, v); stats(ops[1 + blk].b.b.out_stats, mean, m2); }
    }
    __device__ __forceinline__ void lin_down(const float (&v)[4], float mean, float m2) const {
        if (on) { put(ops[0].l.l.out, v); stats(ops[0].l.l.out_stats, mean, m2); }
    }
};

// V8NB > 0: operators [v8_at, v8_at + v8_nops) of the run are the 8-wide bottom of the net (n_blocks = V8NB) and run on the vector unit in
// float32 from the section's image `v8_img` (dsg_narrow8.hpp; global, or staged in LDS by the caller: V8LDS); v8_store: the training forward (every tensor of the section is stored).
// The run for ONE wave and its tile (k_fused_narrow_h below; k_unet_tile runs it on the first wave of a tile's workgroup).
// V8LDS (k_unet_tile): the section's image and its blocks' slices of the step's time-table row were staged in LDS by the workgroup, at
// `v8_lds` ([V8SecL::SIZE floats | (2 V8NB + 3) x 32 floats]); the caller guarantees one time-table row per launch (no per-row entries).
// Read from global memory the section waits for an L2 round trip in front of every product (~25 full drains per tile, disassembly).
// V8LDS = 2 (k_fused_narrow_h): the image alone is in LDS at `v8_lds`; the time-table rows stay in global memory (training: an entry per row).
template <bool PRE, int V8NB, int V8LDS = 0>
__device__ __forceinline__ void narrow_run_body(const FusedOpH* __restrict__ ops, const int nops, const int tile, const int lane, const int v8_at,
                                                const int v8_nops, const float* __restrict__ v8_img, const int v8_store,
                                                const float* v8_lds = nullptr) {
    f32x16 x[1];
    float xmean = 0.f, xm2 = 0.f;
    bool have_x = false;
    int entry = -1;                   // time-table entry of this wave's rows: read once, by the first block of the run
    const int lane_id = lane;
    int i = 0;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
    const int stop = (V8NB > 0 && part == 0 && v8_at >= 0) ? v8_at : nops;
    if (V8NB > 0 && part == 1 && v8_at >= 0) {
        // the float32 section, between the two runs of the operator loop (as in k_fused_narrow_lds)
        int lane = lane_id;
        if (!PRE || V8NB > 0) asm volatile("" : "+v"(lane));
        const int h = lane >> 5, j = lane & 31;
        const LinArgs& la = ops[i].l.l;
        const BlockArgs& b0 = ops[i + 1].b.b;
        const BlockArgs& b1 = ops[i + 2].b.b;
        if (!have_x) {                               // the section opens the run: its 16-wide input comes from memory
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (G < 2) v = ld4(as_global(la.in.data) + ((size_t)seg_tile(la.in, tile) * 2 + G) * 256 + lane * 4);
                x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
            }
            have_x = true;
        }
        if (entry < 0) {
            entry = 0;
            if (b0.ts) {
                int row = (tile % b0.tiles_per_pass) * 32 + j;
                row = row < b0.nrows ? row : b0.nrows - 1;
                entry = as_global(b0.ts)[row];
            } else if (b0.step_ptr) {
                entry = *as_global(b0.step_ptr);
            }
        }
        const V8Sec sc{as_global(b0.cond_pre), (long long)(b1.cond_pre - b0.cond_pre), b0.tiles_per_pass, b0.uncond_tiles};
        const float xi[8] = {x[0][0], x[0][1], x[0][2], x[0][3], x[0][4], x[0][5], x[0][6], x[0][7]};
        float xo[8];
        const V8TableSave sv{ops + i, tile, lane, v8_store != 0};
        if constexpr (V8LDS == 1) {
            v8_lf* const S = (v8_lf*)v8_lds;
            v8_section<(V8NB > 0 ? V8NB : 2)>(S, S + V8SecL<(V8NB > 0 ? V8NB : 2)>::SIZE, sc, tile, lane, xi, xo, xmean, xm2, V8NoSave{});
        } else if constexpr (V8LDS == 2) {
            const float* const tb0 = as_global(b0.tbias) + (size_t)entry * b0.tb_stride;
            v8_section<(V8NB > 0 ? V8NB : 2)>((v8_lf*)v8_lds, tb0, sc, tile, lane, xi, xo, xmean, xm2, sv);
        } else {
            const float* const tb0 = as_global(b0.tbias) + (size_t)entry * b0.tb_stride;  // this lane's row of the time table, the first block's slice
            v8_section<(V8NB > 0 ? V8NB : 2)>(v8_img, tb0, sc, tile, lane, xi, xo, xmean, xm2, sv);
        }
        x[0] = f32x16{xo[0], xo[1], xo[2], xo[3], xo[4], xo[5], xo[6], xo[7], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const FusedOpH& lz = ops[i + v8_nops - 1];
        if (lz.store_out) {                          // the Upsample output is read from memory by somebody (training: the backward pass)
            const LinArgs& a = lz.l.l;
            if (h == 0) reinterpret_cast<float2*>(as_global(a.out_stats))[(size_t)tile * 32 + j] = make_float2(xmean, xm2);
#pragma unroll
            for (int G = 0; G < 2; ++G)
                st4(as_global(a.out) + ((size_t)tile * 2 + G) * 256 + lane * 4, make_float4(x[0][4 * G], x[0][4 * G + 1], x[0][4 * G + 2], x[0][4 * G + 3]));
        }
        i += v8_nops;
    }
#pragma unroll 1
    for (; i < stop; ++i) {
        // opaque per operator (see k_fused_narrow_lds): keeps the lane-derived indices of every operator body inside its body
        int lane = lane_id;
        if (!PRE || V8NB > 0) asm volatile("" : "+v"(lane));
        const int h = lane >> 5, j = lane & 31;
        const FusedOpH& op = ops[i];
        if (entry < 0 && op.kind == 0) {
            const BlockArgs& b0 = op.b.b;
            entry = 0;
            if (b0.ts) {
                int row = (tile % b0.tiles_per_pass) * 32 + j;
                row = row < b0.nrows ? row : b0.nrows - 1;
                entry = as_global(b0.ts)[row];
            } else if (b0.step_ptr) {
                entry = *as_global(b0.step_ptr);
            }
        }
        if (op.kind == 0) {
            BlockArgsH b = op.b;              // every pointer of the record is global (dsg_kernels.hpp, as_global)
            globalize<false>(b);
            if (!have_x) {  // first operator of the run: bring its (<= 32 wide) input into registers once
                const Seg& s0 = b.b.in0;
                const float2 st = reinterpret_cast<const float2*>(s0.stats)[(size_t)seg_tile(s0, tile) * 32 + j];
                xmean = st.x; xm2 = st.y;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < s0.groups) v = ld4(s0.data + ((size_t)seg_tile(s0, tile) * s0.groups + G) * 256 + lane * 4);
                    x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
                }
                have_x = true;
            }
            // skip tensors were stored by this wave earlier in the run: make sure those stores have landed
            if (b.b.in1.groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool st = op.store_out != 0;
            const int N = (V8NB > 0 && op.N < 16) ? 16 : op.N;     // with the float32 section every 8-wide block is inside it
            if (op.sclin) {
                switch (N) {
                    case 4: if (V8NB == 0) resblock_body_h<4, true, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    case 8: if (V8NB == 0) resblock_body_h<8, true, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    case 16: resblock_body_h<16, true, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    default: resblock_body_h<32, true, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                }
            } else {
                switch (N) {
                    case 4: if (V8NB == 0) resblock_body_h<4, false, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    case 8: if (V8NB == 0) resblock_body_h<8, false, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    case 16: resblock_body_h<16, false, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                    default: resblock_body_h<32, false, true, true, PRE>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, entry); break;
                }
            }
        } else if (!have_x || op.l.l.in_groups > 4) {
            // Linear whose input is wider than one tile (the entry of the run): memory in, memory out, then reload
            LinArgsH l = op.l;
            globalize<false>(l);
            linear_body_h<1, IN_FRAG, OUT_FRAG, false>(l, tile, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const LinArgs& a = l.l;
            const int NG = (a.out_width + 7) / 8;
            const float2 st = reinterpret_cast<const float2*>(a.out_stats)[(size_t)tile * 32 + j];
            xmean = st.x; xm2 = st.y;
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (G < NG) v = ld4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4);
                x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
            }
            have_x = true;
        } else {
            LinArgsH l = op.l;
            globalize<false>(l);
            linear_reg_h(l, tile, lane, x, xmean, xm2, op.store_out != 0);
        }
    }
    }
}

template <bool PRE, int V8NB>
__global__ __launch_bounds__(256, PRE ? 2 : 4) void k_fused_narrow_h(const FusedOpH* __restrict__ ops, int nops, int ntiles, int v8_at, int v8_nops,
                                                                     const float* __restrict__ v8_img, int v8_store) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if constexpr (V8NB > 0) {
        // the float32 section's image (11-14 KiB) into LDS, once per workgroup: read from global memory the section waits for an L2 round
        // trip in front of every product (round 5: ~25 full drains per tile in the disassembly; k_unet_tile stages it the same way)
        __shared__ float4 v8s[V8SecL<V8NB>::SIZE / 4];
        if (v8_at >= 0) {
            for (int i = threadIdx.x; i < V8SecL<V8NB>::SIZE / 4; i += 256) v8s[i] = ld4(v8_img + 4 * i);
            __syncthreads();
        }
        if (tile >= ntiles) return;
        narrow_run_body<PRE, V8NB, 2>(ops, nops, tile, lane, v8_at, v8_nops, v8_img, v8_store, reinterpret_cast<const float*>(v8s));
    } else {
        if (tile >= ntiles) return;
        narrow_run_body<PRE, V8NB>(ops, nops, tile, lane, v8_at, v8_nops, v8_img, v8_store);
    }
}

// ---------------------------------------------------------------------------------------------
// The narrow run for LARGE launches with the run's packed planes and per-feature vectors RESIDENT IN LDS.
//
// Why (profiles/r02d_pmc_summary.txt, k_fused_narrow_h<false>): 730 vector-memory reads and 572 scalar loads per wave, 59 % of wave
// cycles in s_waitcnt -- every stage of every narrow block is one or two k16-steps behind "operator record -> planes / vectors ->
// MFMA".  The planes of the whole run are 210 KiB for MSR (74 KiB down + middle, 136 KiB up), its vectors 25 KiB: the host cuts
// the run into PHASES that fit 150 KiB of LDS (dsg_api.hip, plan_narrow_lds), one launch per phase; a 16-wave workgroup (one per
// CU: 4 waves per SIMD as before, one row tile per wave) copies the phase's image once, then every wave walks its tile through
// the phase's operators with every plane and vector a ds_read away.  The running tensor crosses a phase boundary through memory
// (the boundary operator stores, the next phase's first operator reloads: the paths the fused kernel already has).
// The block and Linear bodies are the ones of k_fused_narrow_h: the operator record's weight / vector pointers are REPLACED by
// pointers derived from the __shared__ array inside the kernel, so that after inlining hipcc knows their address space and emits
// ds_read (round 1 handed generic pointers through the record and got flat loads of the LDS aperture -- slower than the L2 hits).
// ---------------------------------------------------------------------------------------------
constexpr int kNarrowLdsU4 = 9472;       // 148 KiB image
struct NarrowLdsOp {                     // LDS offsets (uint4 units for planes, floats for vectors) of one operator of the phase
    unsigned w1, w2, w3, wsc;            // planes (a Linear: w1 only)
    unsigned g1, b1, g2, b2, g3, b3, c2, c3, tb;   // vectors (a Linear: c2 = bias); tb: the time-bias row of the CURRENT step
    int store_out;                       // the table's flag, or 1 at a phase boundary
};
// plan time: the static part of a phase's image (planes, vectors) is gathered ONCE into one contiguous device buffer, so that a
// workgroup stages it with one flat copy loop (a per-entry loop -- ~110 entries, each a dependent round trip -- cost 60 us per
// launch); per step only the phase's slice of the time-table row is added behind it
struct NarrowLdsCopy { const void* src; unsigned dst_u4, n_u4; };
__global__ void k_narrow_image_build(const NarrowLdsCopy* __restrict__ copies, uint4* __restrict__ image) {
    const NarrowLdsCopy c = copies[blockIdx.x];
    if ((reinterpret_cast<unsigned long long>(c.src) & 15ull) == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(c.src);
        for (unsigned i = threadIdx.x; i < c.n_u4; i += blockDim.x) image[c.dst_u4 + i] = src[i];
    } else {    // a raw parameter tensor (the float32 section copies nn.Linear weights as they are) that is only 4-byte aligned
        const float* src = reinterpret_cast<const float*>(c.src);
        float* dst = reinterpret_cast<float*>(image + c.dst_u4);
        for (unsigned i = threadIdx.x; i < 4 * c.n_u4; i += blockDim.x) dst[i] = src[i];
    }
}
struct NarrowPhaseArgs {
    const uint4* image; unsigned n_u4;      // static image of the phase (LDS offset 0)
    const float* tb; unsigned tb_u4;         // its slice of time-table row 0 (+ step * tb_stride floats), LDS offset n_u4
    int op_lo, op_hi;
    // the float32 section (dsg_narrow8.hpp: the 8-wide bottom of the net on the vector unit) when it lies in this phase: operators
    // [v8_at, v8_at + v8_nops), its image at LDS float offset v8_sec, its first block's time-bias slice at v8_tb, v8_store: the tensor
    // that leaves the section crosses a phase boundary through memory.  v8_at < 0: none.
    int v8_at, v8_nops, v8_store;
    unsigned v8_sec, v8_tb;
};

constexpr int kNarrowMaxPhases = 4;
struct NarrowPhases { NarrowPhaseArgs p[kNarrowMaxPhases]; int n; };

// Round 6: ONE launch for the whole run.  The phases (images that fit the LDS) are staged one after the other by the same workgroup, a
// workgroup barrier on either side; the running tensor stays in registers across a phase boundary (rounds 2-5: one launch per phase, the
// tensor through memory -- per-operator stamps, profiles/r06_narrow_op_stamps.txt: 10 000 + 5 500 cycles of staging, a launch ramp per
// phase and a 5 000-cycle reload in front of the second phase's first operator).  The run's opening Linear requests its whole input
// before the first image is staged (linear_mem_prefetch).
template <int V8NB>    // n_blocks of the float32 section the plan may contain (0: none; one instance per shape keeps one copy of the section in the kernel)
__global__ __launch_bounds__(1024, 4) void k_fused_narrow_lds(const FusedOpH* __restrict__ ops, const NarrowLdsOp* __restrict__ lops, const NarrowPhases P,
                                                              int ntiles, const int* step_ptr, int tb_stride) {
    __shared__ uint4 lds[kNarrowLdsU4];
    const int lane = threadIdx.x & 63;
    const int tile_raw = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const bool active = tile_raw < ntiles;                 // every wave of the workgroup meets the staging barriers
    const int tile = active ? tile_raw : ntiles - 1;
#ifdef DSG_CYCLE_STAMPS
    DSG_STAMP(blockIdx.x == 0 && threadIdx.x < 64, 0x400);
#endif
    const float* const ldsf = reinterpret_cast<const float*>(lds);
    f32x16 x[1];
    float xmean = 0.f, xm2 = 0.f;
    bool have_x = false;
    const int lane_id = lane;
    // the run's opening operator, when it is a Linear fed from memory: its input is requested before the first barrier
    LinPre pre;
    const int first = P.p[0].op_lo;
    const bool pre_lin = ops[first].kind == 1 && ops[first].l.l.in_groups <= kLinPreG && !(V8NB > 0 && P.p[0].v8_at == first);
    // the phase's image: every thread copies 16-byte pieces, one piece in flight per thread.  More loads in flight per thread
    // are SLOWER here (same box, us per launch: 1 or 2 pieces 75.3, 4 pieces 80.7, all ten 84.3): 256 workgroups pull the same
    // 75-150 KiB from L2 at the same moment, and the burst costs more than the round trips it saves.
    auto stage = [&](const NarrowPhaseArgs& ph) {
        const uint4* tbs = reinterpret_cast<const uint4*>(ph.tb + (size_t)(step_ptr ? *step_ptr : 0) * tb_stride);
        for (unsigned i = threadIdx.x; i < ph.n_u4; i += blockDim.x) lds[i] = ph.image[i];
        for (unsigned i = threadIdx.x; i < ph.tb_u4; i += blockDim.x) lds[ph.n_u4 + i] = tbs[i];
    };
    stage(P.p[0]);
    if (pre_lin) {      // behind the image's loads (in front of them the nine requests per lane doubled the staging time), under the barrier
        LinArgs l0 = ops[first].l.l;
        globalize_io(l0);
        linear_mem_prefetch(pre, l0, tile, lane);
    }
    __syncthreads();
    DSG_STAMP(tile_raw == 0, 0x401);
    int i_first = first;
    if (pre_lin) {
        // the run's opening Linear, OUTSIDE the operator loops (inside them its 34 prefetch registers would stay live across the whole run)
        DSG_STAMP(tile_raw == 0, 0x410 + first);
        LinArgsH l = ops[first].l;
        const NarrowLdsOp lo = lops[first];
        l.Wh = lds + lo.w1; l.l.bias = ldsf + lo.c2;
        globalize<true>(l);
        linear_pre_to_reg_h(l, pre, tile, lane, x, xmean, xm2, active && lo.store_out != 0);
        have_x = true;
        i_first = first + 1;
    }
#pragma unroll 1
    for (int pi = 0; pi < P.n; ++pi) {
    const NarrowPhaseArgs& ph = P.p[pi];
    if (pi > 0) {
        __syncthreads();              // every wave is done with the previous image
        stage(ph);
        __syncthreads();
        DSG_STAMP(tile_raw == 0, 0x401);
    }
    const int op_hi = ph.op_hi;
    if (!active) continue;
    // The float32 section sits BETWEEN two runs of the operator loop, not inside it (`part`): as one more case of the loop body it made
    // hipcc keep two 16-register tuples of the matrix-core operators in scratch memory.
    int i = pi == 0 ? i_first : ph.op_lo;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
    const int stop = (V8NB > 0 && part == 0 && ph.v8_at >= 0) ? ph.v8_at : op_hi;
    if (V8NB > 0 && part == 1 && ph.v8_at >= 0) {
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5, j = lane & 31;
        const FusedOpH& op = ops[i];
        DSG_STAMP(tile_raw == 0, 0x480);
        {
            // the 8-wide bottom of the net on the vector unit: Downsample 16 -> 8 ... Upsample 8 -> 16, skip tensors in registers
            const LinArgs& la = op.l.l;
            if (!have_x) {                           // first operator of the run: its 16-wide input comes from memory
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < 2) v = ld4(as_global(la.in.data) + ((size_t)seg_tile(la.in, tile) * 2 + G) * 256 + lane * 4);
                    x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
                }
                have_x = true;
            }
            const BlockArgs& b0 = ops[i + 1].b.b;
            const BlockArgs& b1 = ops[i + 2].b.b;
            const V8Sec sc{as_global(b0.cond_pre), (long long)(b1.cond_pre - b0.cond_pre), b0.tiles_per_pass, b0.uncond_tiles};
            v8_lf* const S = (v8_lf*)(ldsf + ph.v8_sec);
            v8_lf* const tb0 = (v8_lf*)(ldsf + ph.v8_tb);
            const float xi[8] = {x[0][0], x[0][1], x[0][2], x[0][3], x[0][4], x[0][5], x[0][6], x[0][7]};
            float xo[8];
            v8_section<(V8NB > 0 ? V8NB : 2)>(S, tb0, sc, tile, lane, xi, xo, xmean, xm2, V8NoSave{});
            x[0] = f32x16{xo[0], xo[1], xo[2], xo[3], xo[4], xo[5], xo[6], xo[7], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ph.v8_store) {                       // something outside this wave reads the section's 16-wide output
                const LinArgs& lz = ops[i + ph.v8_nops - 1].l.l;
                if (h == 0) reinterpret_cast<float2*>(as_global(lz.out_stats))[(size_t)tile * 32 + j] = make_float2(xmean, xm2);
#pragma unroll
                for (int G = 0; G < 2; ++G)
                    st4(as_global(lz.out) + ((size_t)tile * 2 + G) * 256 + lane * 4, make_float4(x[0][4 * G], x[0][4 * G + 1], x[0][4 * G + 2], x[0][4 * G + 3]));
            }
            i += ph.v8_nops;
        }
    }
#pragma unroll 1
    for (; i < stop; ++i) {
        // the lane index as an opaque value per operator: hipcc otherwise hoists every lane-derived index and comparison of every operator
        // body (`8 G + 4 h + p`, `... < width`: ~40 registers) out of this loop, keeps them live across all of it and spills them
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5, j = lane & 31;
        DSG_STAMP(tile_raw == 0, 0x410 + i);
        const FusedOpH& op = ops[i];
        const NarrowLdsOp lo = lops[i];
        if (op.kind == 0) {
            BlockArgsH b = op.b;
            b.W1h = lds + lo.w1; b.W2h = lds + lo.w2; b.W3h = lds + lo.w3; b.Wsch = lds + lo.wsc;
            b.b.gamma1 = ldsf + lo.g1; b.b.beta1 = ldsf + lo.b1; b.b.gamma2 = ldsf + lo.g2; b.b.beta2 = ldsf + lo.b2;
            b.b.gamma3 = ldsf + lo.g3; b.b.beta3 = ldsf + lo.b3; b.b.c2 = ldsf + lo.c2; b.b.c3 = ldsf + lo.c3;
            b.b.tbias = ldsf + lo.tb;            // the staged row of this step: entry 0
            globalize<true>(b);
            if (!have_x) {  // first operator of the run: bring its (<= 32 wide) input into registers once
                const Seg& s0 = b.b.in0;
                const float2 st = reinterpret_cast<const float2*>(s0.stats)[(size_t)seg_tile(s0, tile) * 32 + j];
                xmean = st.x; xm2 = st.y;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < s0.groups) v = ld4(s0.data + ((size_t)seg_tile(s0, tile) * s0.groups + G) * 256 + lane * 4);
                    x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
                }
                have_x = true;
            }
            // skip tensors were stored by this wave earlier in the run: make sure this wave's stores have landed
            if (b.b.in1.groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool st = lo.store_out != 0;
            // with the float32 section in the plan every 8-wide block of the net is inside it (dims[-1] = 8): those kernels carry no
            // matrix-core form of the 4- and 8-wide blocks
            const int N = (V8NB > 0 && op.N < 16) ? 16 : op.N;
            if (op.sclin) {
                switch (N) {
                    case 4: if (V8NB == 0) resblock_body_h<4, true, true, true, false, false, 1>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    case 8: if (V8NB == 0) resblock_body_h<8, true, true, true, false, false, 1>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    case 16: resblock_body_h<16, true, true, true, false, false, 1, 1>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    default: resblock_body_h<32, true, true, true, false, false, 2, 0>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                }
            } else {
                switch (N) {
                    case 4: if (V8NB == 0) resblock_body_h<4, false, true, true, false, false, 1>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    case 8: if (V8NB == 0) resblock_body_h<8, false, true, true, false, false, 1>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    case 16: resblock_body_h<16, false, true, true, false, false, 1, 0>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                    default: resblock_body_h<32, false, true, true, false, false, 2, 0>(b, tile, lane, &x, &xmean, &xm2, st, nullptr, 0); break;
                }
            }
        } else if (op.kind == 2) {
            // the 64-wide Linear that consumes the run's last tensor (Upsample): input from registers, output to memory
            LinArgsH l = op.l;
            l.Wh = lds + lo.w1; l.l.bias = ldsf + lo.c2;
            globalize<true>(l);
            linear_reg_out_h<2>(l, tile, lane, x, xmean, xm2);
        } else {
            LinArgsH l = op.l;
            l.Wh = lds + lo.w1; l.l.bias = ldsf + lo.c2;
            globalize<true>(l);
            if (!have_x || l.l.in_groups > 4) {
                // any other Linear whose input is wider than one tile or that opens the run: memory in, memory out, then reload
                linear_body_h<1, IN_FRAG, OUT_FRAG, false>(l, tile, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const LinArgs& a = l.l;
                const int NG = (a.out_width + 7) / 8;
                const float2 st = reinterpret_cast<const float2*>(a.out_stats)[(size_t)tile * 32 + j];
                xmean = st.x; xm2 = st.y;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < NG) v = ld4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4);
                    x[0][4 * G] = v.x; x[0][4 * G + 1] = v.y; x[0][4 * G + 2] = v.z; x[0][4 * G + 3] = v.w;
                }
                have_x = true;
            } else {
                linear_reg_h(l, tile, lane, x, xmean, xm2, lo.store_out != 0);
            }
        }
    }
    }
    }
    { const int lane = lane_id; DSG_STAMP(tile_raw == 0, 0x4ff); }
}

// ---------------------------------------------------------------------------------------------
// Bind-time helpers: max|W| per weight tensor, and the fp16 plane packing
//   dst[((nt*KS + S)*2 + plane)*64 + lane] = 8 halfs: plane(W[32nt + (lane&31)][col(2S + (jj>>2), 4(lane>>5) + (jj&3))] * 2^e)
// with every K segment padded to an EVEN number of 8-feature groups.
// ---------------------------------------------------------------------------------------------
// grid (tensors, kMaxabsSlices): a slice of a tensor per block, combined by atomicMax on the bit pattern (values are >= 0, so
// the patterns order like the values); `out` is zeroed by the caller.  One block per tensor made the launch as long as the
// serial walk over the largest weight matrix.
constexpr int kMaxabsSlices = 8;
__global__ __launch_bounds__(256) void k_maxabs(const float* const* __restrict__ ptrs, const long long* __restrict__ numel,
                                                const int* __restrict__ out_idx, float* __restrict__ out) {
    __shared__ float sm[4];
    const float* p = ptrs[blockIdx.x];
    const long long n = numel[blockIdx.x];
    const long long per = (n + kMaxabsSlices - 1) / kMaxabsSlices, lo = blockIdx.y * per, hi = lo + per < n ? lo + per : n;
    float m = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (v > 0.f) atomicMax(reinterpret_cast<unsigned*>(out + out_idx[blockIdx.x]), __float_as_uint(v));
    }
}

// ---------------------------------------------------------------------------------------------
// Condition embeddings of ALL blocks in one launch: ce_b = Wc_b silu(cond * mask) (UNetCF.py:93), the input is the same
// for every block, so a wave splits its tile's condition fragments once (<= 8 k16-steps, registers) and walks the list of
// 32-wide output tiles of every block: 3 MFMAs per k16-step, unscale, store in the block's fragment region.
// ---------------------------------------------------------------------------------------------
struct CondTile {
    float* out;           // block region [tile][NG_b][256] + first group of this output tile
    const float* m;       // max|Wc_b|
    int groups;           // groups of this output tile that exist (1..4)
    int ng_block;         // NG_b (tile stride of the region in groups)
};
constexpr int kCondMaxSteps = 8;   // cond_dim <= 128

__global__ __launch_bounds__(256) void k_cond_embed_h(const float* __restrict__ condfrag, int CG, const uint4* __restrict__ W,
                                                      const CondTile* __restrict__ tiles_tab, int nout, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= ntiles) return;
    const int KS = (CG + 1) >> 1;
    h8 bhi[kCondMaxSteps], blo[kCondMaxSteps];
#pragma unroll
    for (int S = 0; S < kCondMaxSteps; ++S) {
        if (S < KS) {
            const float4 x0 = ld4(condfrag + ((size_t)tile * CG + 2 * S) * 256 + lane * 4);
            const float4 x1 = 2 * S + 1 < CG ? ld4(condfrag + ((size_t)tile * CG + 2 * S + 1) * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float v[8] = {kActScale * x0.x, kActScale * x0.y, kActScale * x0.z, kActScale * x0.w,
                          kActScale * x1.x, kActScale * x1.y, kActScale * x1.z, kActScale * x1.w};
            split8(v, bhi[S], blo[S]);
        }
    }
    // gridDim.y waves share one row tile: each takes every gridDim.y-th output tile (the operand split above is repeated per
    // wave: 10 k-groups, cheap), so that small launches still fill the SIMDs
    const int part = blockIdx.y, nparts = gridDim.y;
    const size_t ot_stride = (size_t)KS * 128;
    uint4 wh[kCondMaxSteps], wl[kCondMaxSteps];
    auto loadw = [&](int ot) {
#pragma unroll
        for (int S = 0; S < kCondMaxSteps; ++S)
            if (S < KS) { wh[S] = W[ot * ot_stride + (size_t)S * 128 + lane]; wl[S] = W[ot * ot_stride + (size_t)S * 128 + 64 + lane]; }
    };
    if (part < nout) loadw(part);
    // the tile's record and the weight maximum behind its pointer are requested one output tile ahead as well (round 5: record ->
    // pointer -> max|W| -> un-scale was a dependent chain of two round trips in front of every tile's stores)
    CondTile ctn = tiles_tab[part < nout ? part : 0];
    float mn = *as_global(ctn.m);
    for (int ot = part; ot < nout; ot += nparts) {
        const CondTile ct = ctn;
        const float mcur = mn;
        if (ot + nparts < nout) { ctn = tiles_tab[ot + nparts]; mn = *as_global(ctn.m); }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int S = 0; S < kCondMaxSteps; ++S)
            if (S < KS) {
                const h8 whi = __builtin_bit_cast(h8, wh[S]), wlo = __builtin_bit_cast(h8, wl[S]);
                DSG_MFMA_H(acc, whi, bhi[S]);
                DSG_MFMA_H(acc, whi, blo[S]);
                DSG_MFMA_H(acc, wlo, bhi[S]);
            }
        if (ot + nparts < nout) loadw(ot + nparts);     // next tile's planes land under this tile's MFMAs and stores
        const float inv = ldexpf(1.0f / kActScale, -scale_exp(mcur));
        float* o = as_global(ct.out) + (size_t)tile * ct.ng_block * 256 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < ct.groups) st4(o + (size_t)q * 256, make_float4(acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv));
    }
}

struct PackHDesc {
    const float* W;       // [N][Ktot]
    uint4* dst;
    const float* m_self;  // max|W| of this tensor
    const float* m_pair;  // lin3 <-> shortcut partner or null
    int role;             // 0: standalone, 1: lin3 with shortcut partner, 2: shortcut with lin3 partner
    int N, Ktot, w0, w1, NT;
    int transposed;       // 1: A operand = W^T (rows = input features in padded group order, k = output features): data gradients
    long long total;      // uint4 elements of dst
    long long blk_begin;
};

// Blocks [0, ceil(nopc / 256)) also write the un-scale constants of the operators (they need the same final
// max|W| words as the packing, so they ride in this launch instead of one of their own).
__global__ __launch_bounds__(256) void k_pack_h(const PackHDesc* __restrict__ descs, int ndesc, const float* __restrict__ maxabs,
                                                const OpConstDesc* __restrict__ opc_desc, int nopc, float* __restrict__ opc_out) {
    {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < nopc) {
            const OpConstDesc c = opc_desc[i];
            if (c.w2 < 0) {          // Linear
                const int e = scale_exp(maxabs[c.w1]);
                opc_out[4 * i + 0] = ldexpf(1.0f / kRawScale, -e); opc_out[4 * i + 1] = ldexpf(1.0f / kActScale, -e); opc_out[4 * i + 2] = 0.f; opc_out[4 * i + 3] = 0.f;
            } else {
                const int e1 = scale_exp(maxabs[c.w1]), e2 = scale_exp(maxabs[c.w2]);
                const int e3 = c.wsc >= 0 ? scale_exp_lin3(maxabs[c.w3], maxabs[c.wsc]) : scale_exp(maxabs[c.w3]);
                opc_out[4 * i + 0] = ldexpf(1.0f / kActScale, -e1); opc_out[4 * i + 1] = ldexpf(1.0f / kActScale, -e2);
                opc_out[4 * i + 2] = ldexpf(1.0f / kActScale, -e3); opc_out[4 * i + 3] = 0.f;
            }
        }
    }
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk_begin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackHDesc d = descs[lo];
    const long long idx = ((long long)blockIdx.x - d.blk_begin) * 256 + threadIdx.x;
    if (idx >= d.total) return;
    int e;
    if (d.role == 0) e = scale_exp(*d.m_self);
    else if (d.role == 1) e = scale_exp_lin3(*d.m_self, *d.m_pair);
    else e = scale_exp_lin3(*d.m_pair, *d.m_self) + 4;
    const float sc = ldexpf(1.0f, e);
    const int lane = idx & 63, plane = (idx >> 6) & 1;
    const long long ts = idx >> 7;
    const int g0 = (d.w0 + 7) / 8, g1 = (d.w1 + 7) / 8;
    const int h = lane >> 5;
    h8 out;
    if (d.transposed) {
        const int KS = ((d.N + 7) / 8 + 1) >> 1;
        const int S = ts % KS, nt = ts / KS;
        // row of W^T = input feature in padded group order (each concat segment padded to whole groups)
        const int r = 32 * nt + (lane & 31), G = r >> 3, e = r & 7;
        int col = -1;
        if (G < g0) { if (8 * G + e < d.w0) col = 8 * G + e; }
        else if (G < g0 + g1) { const int c = 8 * (G - g0) + e; if (c < d.w1) col = d.w0 + c; }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int kf = 8 * (2 * S + (jj >> 2)) + 4 * h + (jj & 3);
            float v = 0.f;
            if (col >= 0 && kf < d.N) v = d.W[(size_t)kf * d.Ktot + col] * sc;
            const _Float16 vh = (_Float16)v;
            out[jj] = plane == 0 ? vh : (_Float16)(v - (float)vh);
        }
    } else {
    const int ks0 = (g0 + 1) >> 1, KS = ks0 + ((g1 + 1) >> 1);
    const int S = ts % KS, nt = ts / KS;
    const int n = 32 * nt + (lane & 31);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int p = jj & 3;
        int col = -1;
        if (S < ks0) {
            const int g = 2 * S + (jj >> 2), f = 8 * g + 4 * h + p;
            if (g < g0 && f < d.w0) col = f;
        } else {
            const int g = 2 * (S - ks0) + (jj >> 2), f = 8 * g + 4 * h + p;
            if (g < g1 && f < d.w1) col = d.w0 + f;
        }
        float v = 0.f;
        if (col >= 0 && n < d.N) v = d.W[(size_t)n * d.Ktot + col] * sc;
        const _Float16 vh = (_Float16)v;
        out[jj] = plane == 0 ? vh : (_Float16)(v - (float)vh);
    }
    }
    d.dst[idx] = __builtin_bit_cast(uint4, out);
}

}  // namespace dsg
