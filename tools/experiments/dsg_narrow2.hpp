// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 2): the narrow run with TWO row tiles per wave.  Parity-green (27 sampling tests
// incl. the NU checkpoint and the 65 536-row oracle check passed with it wired in), but SLOWER than k_fused_narrow_h: 219.8 us
// at 2 waves/SIMD, 199.6 us at 3 waves/SIMD (168 VGPRs) against 180 us.  The narrow run is latency-bound per dependency chain
// (SQ_ACTIVE_INST_ANY summed over a SIMD's four waves is 72 % of ONE wave's cycles): two tiles in one wave at half the
// occupancy are the same number of chains per SIMD, and what the shared loads save in instructions does not show.  It also
// needs `struct NarrowBases` / `rebase` (base + laundered offset -> global instead of flat loads), which was measured on its
// own with k_fused_narrow_h: 461 flat_load -> 461 global_load, -12 % static VALU, and 209 us instead of 180 -- dropped too.
//
// k_fused_narrow_h walks one tile per wave through the 22 narrow operators of MSR-80c.  Its PMC summary
// (profiles/r02b_pmc_summary.txt): 11 800 VALU, 590 scalar-memory and 780 vector-memory instructions per wave for 333 MFMAs,
// 63 % of wave cycles in s_waitcnt -- every stage of every block is one or two k16-steps behind a dependent chain of loads
// (operator record -> weight planes / LayerNorm vectors / bias -> MFMA -> row statistics -> next stage), and four waves per SIMD
// are not enough to cover it.  Everything in that chain except the activations themselves is the same for every row tile:
// here a wave owns TP = 2 consecutive tiles and reads the operator record, the weight planes, the LayerNorm vectors, the bias /
// time rows and the scale constants ONCE for both; the two tiles' arithmetic is independent, so the scheduler has two
// dependency chains to interleave inside a wave (ILP instead of occupancy: 2 waves per SIMD).  Arithmetic per element is that
// of resblock_body_h<N, SCLIN, XIN, XOUT> / linear_reg_h (same planes, scales, accumulation order): results are bit-identical
// to the one-tile form.  Sampling only (no saved tensors, time row chosen by the step): the training forward keeps
// k_fused_narrow_h.
#pragma once
#include "dsg_split.hpp"

namespace dsg {

constexpr int kTP = 2;

struct HF1 { uint4 hi, lo; };
__device__ __forceinline__ HF1 ld_hf1(const uint4* __restrict__ wp /* + lane */) { return HF1{wp[0], wp[64]}; }

template <bool FIRST>
__device__ __forceinline__ void mma1(f32x16& acc, const HF1& w, const h8 bhi, const h8 blo) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (FIRST) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w.hi), bhi, z, 0, 0, 0);
    else DSG_MFMA_H(acc, __builtin_bit_cast(h8, w.hi), bhi);
    DSG_MFMA_H(acc, __builtin_bit_cast(h8, w.hi), blo);
    DSG_MFMA_H(acc, __builtin_bit_cast(h8, w.lo), bhi);
}

// Register-fed chain over both tiles: out[tp] (+)= W * split(16 silu(LN(in[tp]))) or W * split(in[tp]); N = width of `in`.
template <int N, bool LNACT, bool ZERO>
__device__ __forceinline__ void chain_reg2(f32x16 (&out)[kTP], const f32x16 (&in)[kTP], const uint4* __restrict__ wp /* step 0, + lane */,
                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float (&mean)[kTP],
                                           const float (&rstd)[kTP], int h) {
    constexpr int KS = ((N + 7) / 8 + 1) / 2;
#pragma unroll
    for (int S = 0; S < KS; ++S) {
        const HF1 w = ld_hf1(wp + (size_t)S * 128);
        const bool two = 16 * S + 8 < N;             // the second quad of the step is padding otherwise: operand exactly 0
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g0 = z4, b0 = z4, g1 = z4, b1 = z4;
        if (LNACT) {
            g0 = ld4(gamma + 16 * S + 4 * h); b0 = ld4(beta + 16 * S + 4 * h);
            if (two) { g1 = ld4(gamma + 16 * S + 8 + 4 * h); b1 = ld4(beta + 16 * S + 8 + 4 * h); }
        }
#pragma unroll
        for (int tp = 0; tp < kTP; ++tp) {
            const int r0 = 8 * S;
            float v[8];
            if (LNACT) {
                const float c = rstd[tp], d = -mean[tp] * rstd[tp];
                v[0] = silu_scaled(fmaf(fmaf(in[tp][r0], c, d), g0.x, b0.x)); v[1] = silu_scaled(fmaf(fmaf(in[tp][r0 + 1], c, d), g0.y, b0.y));
                v[2] = silu_scaled(fmaf(fmaf(in[tp][r0 + 2], c, d), g0.z, b0.z)); v[3] = silu_scaled(fmaf(fmaf(in[tp][r0 + 3], c, d), g0.w, b0.w));
                if (two) {
                    v[4] = silu_scaled(fmaf(fmaf(in[tp][r0 + 4], c, d), g1.x, b1.x)); v[5] = silu_scaled(fmaf(fmaf(in[tp][r0 + 5], c, d), g1.y, b1.y));
                    v[6] = silu_scaled(fmaf(fmaf(in[tp][r0 + 6], c, d), g1.z, b1.z)); v[7] = silu_scaled(fmaf(fmaf(in[tp][r0 + 7], c, d), g1.w, b1.w));
                } else { v[4] = v[5] = v[6] = v[7] = 0.f; }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = kRawScale * in[tp][r0 + q];
            }
            h8 bhi, blo;
            split8(v, bhi, blo);
            if (ZERO && S == 0) mma1<true>(out[tp], w, bhi, blo); else mma1<false>(out[tp], w, bhi, blo);
        }
    }
}

// Memory-fed chain over both tiles (skip tensors: `groups` 8-feature groups, at most 4): xp[tp] = tile base + lane * 4.
template <bool LNACT>
__device__ __forceinline__ void chain_mem2(f32x16 (&acc)[kTP], const float* const (&xp)[kTP], int groups, const uint4* __restrict__ wp,
                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float (&mean)[kTP],
                                           const float (&rstd)[kTP]) {
    const int steps = (groups + 1) >> 1;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int S = 0; S < 2; ++S) {
        if (S < steps) {
            const HF1 w = ld_hf1(wp + (size_t)S * 128);
            float4 g0 = z4, b0 = z4, g1 = z4, b1 = z4;
            if (LNACT) { g0 = ld4(gamma + 16 * S); b0 = ld4(beta + 16 * S); g1 = ld4(gamma + 16 * S + 8); b1 = ld4(beta + 16 * S + 8); }
            float4 x0[kTP], x1[kTP];
#pragma unroll
            for (int tp = 0; tp < kTP; ++tp) {
                x0[tp] = ld4(xp[tp] + (size_t)(2 * S) * 256);
                x1[tp] = (2 * S + 1 < groups) ? ld4(xp[tp] + (size_t)(2 * S + 1) * 256) : z4;
            }
#pragma unroll
            for (int tp = 0; tp < kTP; ++tp) {
                float v[8];
                if (LNACT) act8(v, x0[tp], x1[tp], rstd[tp], -mean[tp] * rstd[tp], g0, b0, g1, b1);
                else {
                    v[0] = kRawScale * x0[tp].x; v[1] = kRawScale * x0[tp].y; v[2] = kRawScale * x0[tp].z; v[3] = kRawScale * x0[tp].w;
                    v[4] = kRawScale * x1[tp].x; v[5] = kRawScale * x1[tp].y; v[6] = kRawScale * x1[tp].z; v[7] = kRawScale * x1[tp].w;
                }
                h8 bhi, blo;
                split8(v, bhi, blo);
                mma1<false>(acc[tp], w, bhi, blo);
            }
        }
    }
}

// acc[tp] <- acc[tp] * inv + vec (NQ real 8-feature groups), the vector read once for both tiles
template <int NQ>
__device__ __forceinline__ void unscale_add2(f32x16 (&acc)[kTP], float inv, const float* __restrict__ vec, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= NQ) continue;
        const float4 b = ld4(vec + 8 * q + 4 * h);
#pragma unroll
        for (int tp = 0; tp < kTP; ++tp) {
            acc[tp][4 * q + 0] = fmaf(acc[tp][4 * q + 0], inv, b.x); acc[tp][4 * q + 1] = fmaf(acc[tp][4 * q + 1], inv, b.y);
            acc[tp][4 * q + 2] = fmaf(acc[tp][4 * q + 2], inv, b.z); acc[tp][4 * q + 3] = fmaf(acc[tp][4 * q + 3], inv, b.w);
        }
    }
}

// One ResidualBlock of the narrow run on both tiles; x / xmean / xm2 in and out (registers); stored only if store_out.
template <int N, bool SCLIN>
__device__ __forceinline__ void narrow_block2(const BlockArgsH& ah, const int (&tile)[kTP], const bool (&live)[kTP], int lane, f32x16 (&x)[kTP],
                                              float (&xmean)[kTP], float (&xm2)[kTP], bool store_out) {
    constexpr int NG = (N + 7) / 8;
    const BlockArgs& a = ah.b;
    const int h = lane >> 5, j = lane & 31;
    const int ks0 = (a.in0.groups + 1) >> 1, ks1 = (a.in1.groups + 1) >> 1, KS1 = ks0 + ks1;
    (void)ks1; (void)KS1;
    float mean1[kTP], rstd1[kTP];
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp) {
        float mean = xmean[tp], m2 = xm2[tp];
        if (a.in1.groups) {
            const float2 s1 = reinterpret_cast<const float2*>(a.in1.stats)[(size_t)seg_tile(a.in1, tile[tp]) * 32 + j];
            const float dd = s1.x - mean;
            m2 = m2 + s1.y + dd * dd * a.chan_w;
            mean = mean + dd * a.chan_f;
        }
        mean1[tp] = mean;
        rstd1[tp] = rsqrtf(m2 * a.inv_nin + kLnEps);
        if (SCLIN) range_check(a.range_flag, mean, m2);
    }
    const float inv1 = ah.kc[0], inv2 = ah.kc[1], inv3 = ah.kc[2];
    const float* skip[kTP];
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp)
        skip[tp] = a.in1.groups ? a.in1.data + (size_t)seg_tile(a.in1, tile[tp]) * a.in1.groups * 256 + lane * 4 : a.in0.data;

    // ---- stage 1
    f32x16 acc1[kTP];
    chain_reg2<N, true, true>(acc1, x, ah.W1h + lane, a.gamma1, a.beta1, mean1, rstd1, h);
    if (a.in1.groups)
        chain_mem2<true>(acc1, skip, a.in1.groups, ah.W1h + (size_t)ks0 * 128 + lane, a.gamma1 + 8 * a.in0.groups + 4 * h,
                         a.beta1 + 8 * a.in0.groups + 4 * h, mean1, rstd1);
    unscale_add2<NG>(acc1, inv1, a.tbias + (size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride, h);

    // ---- stage 2
    f32x16 acc2[kTP];
    {
        float mean[kTP], rstd[kTP];
#pragma unroll
        for (int tp = 0; tp < kTP; ++tp) {
            float m2;
            acc_stats<N, 1>(reinterpret_cast<const f32x16(&)[1]>(acc1[tp]), h, mean[tp], m2);
            rstd[tp] = rsqrtf(m2 * (1.0f / N) + kLnEps);
        }
        chain_reg2<N, true, true>(acc2, acc1, ah.W2h + lane, a.gamma2, a.beta2, mean, rstd, h);
        unscale_add2<NG>(acc2, inv2, a.c2, h);
    }
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp)
        if (tile[tp] >= a.uncond_tiles) {
            const float* cp = a.cond_pre + (size_t)(tile[tp] % a.tiles_per_pass) * NG * 256 + lane * 4;
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                const float4 cv = ld4(cp + (size_t)G * 256);
                acc2[tp][4 * G + 0] += cv.x; acc2[tp][4 * G + 1] += cv.y; acc2[tp][4 * G + 2] += cv.z; acc2[tp][4 * G + 3] += cv.w;
            }
        }

    // ---- stage 3 (+ shortcut in the same scaled accumulator)
    f32x16 (&acc3)[kTP] = acc1;
    {
        float mean[kTP], rstd[kTP];
#pragma unroll
        for (int tp = 0; tp < kTP; ++tp) {
            float m2;
            acc_stats<N, 1>(reinterpret_cast<const f32x16(&)[1]>(acc2[tp]), h, mean[tp], m2);
            rstd[tp] = rsqrtf(m2 * (1.0f / N) + kLnEps);
        }
        chain_reg2<N, true, true>(acc3, acc2, ah.W3h + lane, a.gamma3, a.beta3, mean, rstd, h);
    }
    if (SCLIN) {
        const float zero[kTP] = {0.f, 0.f}, one[kTP] = {1.f, 1.f};
        chain_reg2<N, false, false>(acc3, x, ah.Wsch + lane, nullptr, nullptr, zero, one, h);
        if (a.in1.groups) chain_mem2<false>(acc3, skip, a.in1.groups, ah.Wsch + (size_t)ks0 * 128 + lane, nullptr, nullptr, zero, one);
        unscale_add2<NG>(acc3, inv3, a.c3, h);
    } else {
        unscale_add2<NG>(acc3, inv3, a.c3, h);
#pragma unroll
        for (int tp = 0; tp < kTP; ++tp) acc3[tp] += x[tp];
    }

    // ---- statistics, hand over in registers, store if something outside this wave reads it
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp) {
        float mean, m2;
        acc_stats<N, 1>(reinterpret_cast<const f32x16(&)[1]>(acc3[tp]), h, mean, m2);
        x[tp] = acc3[tp]; xmean[tp] = mean; xm2[tp] = m2;
        if (store_out && live[tp]) {
            if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile[tp] * 32 + j] = make_float2(mean, m2);
#pragma unroll
            for (int G = 0; G < NG; ++G)
                st4(a.out + ((size_t)tile[tp] * NG + G) * 256 + lane * 4,
                    make_float4(acc3[tp][4 * G], acc3[tp][4 * G + 1], acc3[tp][4 * G + 2], acc3[tp][4 * G + 3]));
        }
    }
}

// Linear with register input (<= 32 wide) and register output (<= 32 wide) on both tiles (Down/Upsample inside the run)
__device__ __forceinline__ void linear_reg2(const LinArgsH& ah, const int (&tile)[kTP], const bool (&live)[kTP], int lane, f32x16 (&x)[kTP],
                                            float (&xmean)[kTP], float (&xm2)[kTP], bool store_out) {
    const LinArgs& a = ah.l;
    const int h = lane >> 5, j = lane & 31;
    const int steps = (a.in_groups + 1) >> 1;
    f32x16 acc[kTP];
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp) range_check(a.range_flag, xmean[tp], xm2[tp]);
#pragma unroll
    for (int S = 0; S < 2; ++S) {
        if (S < steps) {
            const HF1 w = ld_hf1(ah.Wh + (size_t)S * 128 + lane);
#pragma unroll
            for (int tp = 0; tp < kTP; ++tp) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = kRawScale * x[tp][8 * S + q];
                h8 bhi, blo;
                split8(v, bhi, blo);
                if (S == 0) mma1<true>(acc[tp], w, bhi, blo); else mma1<false>(acc[tp], w, bhi, blo);
            }
        }
    }
    unscale_add2<4>(acc, ah.kc[0], a.bias, h);
    const int NG = (a.out_width + 7) / 8;
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp) {
        float s = 0.f;
#pragma unroll
        for (int G = 0; G < 4; ++G)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (8 * G + 4 * h + p < a.out_width) s += acc[tp][4 * G + p];
        const float m = xhalf_sum(s) * a.inv_out_w;
        float q = 0.f;
#pragma unroll
        for (int G = 0; G < 4; ++G)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (8 * G + 4 * h + p < a.out_width) { const float d = acc[tp][4 * G + p] - m; q = fmaf(d, d, q); }
        q = xhalf_sum(q);
        x[tp] = acc[tp]; xmean[tp] = m; xm2[tp] = q;
        if (store_out && live[tp]) {
            if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile[tp] * 32 + j] = make_float2(m, q);
#pragma unroll
            for (int G = 0; G < 4; ++G)
                if (G < NG) st4(a.out + ((size_t)tile[tp] * NG + G) * 256 + lane * 4, make_float4(acc[tp][4 * G], acc[tp][4 * G + 1], acc[tp][4 * G + 2], acc[tp][4 * G + 3]));
        }
    }
}

__global__ __launch_bounds__(256, 3) void k_fused_narrow2_h(const FusedOpH* __restrict__ ops, int nops, int ntiles, const NarrowBases nb) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (w * kTP >= ntiles) return;
    const int h = lane >> 5, j = lane & 31;
    (void)h;
    int tile[kTP];
    bool live[kTP];
#pragma unroll
    for (int tp = 0; tp < kTP; ++tp) { live[tp] = w * kTP + tp < ntiles; tile[tp] = live[tp] ? w * kTP + tp : ntiles - 1; }
    f32x16 x[kTP];
    float xmean[kTP] = {0.f, 0.f}, xm2[kTP] = {0.f, 0.f};
    bool have_x = false;
    for (int i = 0; i < nops; ++i) {
        FusedOpH op = ops[i];
        rebase(op.b, nb); rebase(op.l, nb);
        if (op.kind == 0) {
            if (!have_x) {  // first operator of the run: bring its (<= 32 wide) input into registers once
                const Seg& s0 = op.b.b.in0;
#pragma unroll
                for (int tp = 0; tp < kTP; ++tp) {
                    const float2 st = reinterpret_cast<const float2*>(s0.stats)[(size_t)seg_tile(s0, tile[tp]) * 32 + j];
                    xmean[tp] = st.x; xm2[tp] = st.y;
#pragma unroll
                    for (int G = 0; G < 4; ++G) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (G < s0.groups) v = ld4(s0.data + ((size_t)seg_tile(s0, tile[tp]) * s0.groups + G) * 256 + lane * 4);
                        x[tp][4 * G] = v.x; x[tp][4 * G + 1] = v.y; x[tp][4 * G + 2] = v.z; x[tp][4 * G + 3] = v.w;
                    }
                }
                have_x = true;
            }
            // skip tensors were stored by this wave earlier in the run: make sure those stores have landed
            if (op.b.b.in1.groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const bool st = op.store_out != 0;
            if (op.sclin) {
                switch (op.N) {
                    case 4: narrow_block2<4, true>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    case 8: narrow_block2<8, true>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    case 16: narrow_block2<16, true>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    default: narrow_block2<32, true>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                }
            } else {
                switch (op.N) {
                    case 4: narrow_block2<4, false>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    case 8: narrow_block2<8, false>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    case 16: narrow_block2<16, false>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                    default: narrow_block2<32, false>(op.b, tile, live, lane, x, xmean, xm2, st); break;
                }
            }
        } else if (!have_x || op.l.l.in_groups > 4) {
            // Linear whose input is wider than one tile (the entry of the run): memory in, memory out, then reload
#pragma unroll
            for (int tp = 0; tp < kTP; ++tp)
                if (live[tp] || tp == 0) linear_body_h<1, IN_FRAG, OUT_FRAG, false>(op.l, tile[tp], lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const LinArgs& a = op.l.l;
            const int NG = (a.out_width + 7) / 8;
#pragma unroll
            for (int tp = 0; tp < kTP; ++tp) {
                const float2 st = reinterpret_cast<const float2*>(a.out_stats)[(size_t)tile[tp] * 32 + j];
                xmean[tp] = st.x; xm2[tp] = st.y;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < NG) v = ld4(a.out + ((size_t)tile[tp] * NG + G) * 256 + lane * 4);
                    x[tp][4 * G] = v.x; x[tp][4 * G + 1] = v.y; x[tp][4 * G + 2] = v.z; x[tp][4 * G + 3] = v.w;
                }
            }
            have_x = true;
        } else {
            linear_reg2(op.l, tile, live, lane, x, xmean, xm2, op.store_out != 0);
        }
    }
}

}  // namespace dsg
