// The 8-wide bottom of the U-Net (Downsample 16 -> 8, n_blocks DownBlocks, the two MiddleBlocks, n_blocks + 1 UpBlocks 16 -> 8,
// Upsample 8 -> 16; UNetCF.py:278-311 at dims[-1] = 8) on the VECTOR unit, exact float32, inside the fused narrow run.
//
// Why (profiles/r03f_pmc_summary.txt k_fused_narrow_lds): on the matrix-core path an 8-wide block is 470 vector
// instructions and 12 padded 32x32x16 MFMAs (an 8 x 8 product uses 1/8 of a 32 x 16 A tile, times three for the hi/lo split) for
// 3 x 64 useful multiply-adds per row -- LayerNorm, SiLU and the fp16 split of 32 padded features, the cross-lane statistics, the
// un-scaling.  The fragment layout of an 8-wide tensor is already "half a row per lane": lane l = 32 h + j holds features
// 4h .. 4h+3 of row j.  So here each lane computes ITS four output features of every Linear in float32 FMAs:
//   * the eight activated inputs of a row are made visible to both of its lanes with four v_permlane32_swap (v8_gather);
//   * the weights are the raw nn.Linear matrices (row-major [out][in]: a lane's four rows are contiguous), staged in LDS with the
//     rest of the phase image, read as two-address broadcast ds_read_b128;
//   * no operand scaling, no fp16 planes, no range flag: the arithmetic is the reference's float32 (fused multiply-adds in feature order).
// The section's skip tensors (UNetCF.py:333-340, 350-351: the Downsample output and every DownBlock output, popped by the UpBlocks) never
// leave the registers.  ~190 (down / middle) and ~300 (up) vector instructions per block and row tile instead of 470 + 12 MFMAs and
// two memory round trips per skip tensor.
#pragma once
#include "dsg_kernels.hpp"

namespace dsg {

typedef float v8f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float v8_lf;
typedef __attribute__((address_space(3))) const v8f4 v8_lf4;

// The image of the section is read through `PF`: an LDS pointer (v8_lf*: the LDS form of the narrow run stages it with the rest of its
// phase) or a global one (const float*: small launches and the training forward read the 11 KiB image from L1 / L2).
__device__ __forceinline__ v8f4 v8_ld4(v8_lf* p) { return *(v8_lf4*)p; }
__device__ __forceinline__ v8f4 v8_ld4(const float* p) { const float4 v = ld4(p); return v8f4{v.x, v.y, v.z, v.w}; }

// float offsets inside one block's record (the host lays the pieces out in this order: dsg_api.hip, plan of the narrow run)
template <bool UP> struct V8BlockL {
    static constexpr int K1 = UP ? 16 : 8;
    static constexpr int W1 = 0, W2 = W1 + 8 * K1, W3 = W2 + 64, WSC = W3 + 64, G1 = WSC + (UP ? 128 : 0), B1 = G1 + K1, G2 = B1 + K1, B2 = G2 + 8,
                         G3 = B2 + 8, B3 = G3 + 8, C2 = B3 + 8, C3 = C2 + 8, SIZE = C3 + 8;
};
// the section: [Downsample W 8x16 | bias 8] [NB down] [2 middle] [NB + 1 up] [Upsample W 16x8 | bias 16]
template <int NB> struct V8SecL {
    static constexpr int LIND_W = 0, LIND_B = 128, DOWN = 136, MID = DOWN + NB * V8BlockL<false>::SIZE, UP = MID + 2 * V8BlockL<false>::SIZE,
                         LINU_W = UP + (NB + 1) * V8BlockL<true>::SIZE, LINU_B = LINU_W + 128, SIZE = LINU_B + 16;
    static constexpr int NOPS = 2 * NB + 5;        // operators of the plan the section replaces
};
constexpr int kV8TbStride = 32;                    // a block's slice of the time-table row is padded to 32 floats (pad32(N))

// silu(LayerNorm(x)): the LayerNorm as x * c + d (c = rstd, d = -mean * rstd) then the affine pair, as every other forward kernel here
__device__ __forceinline__ float v8_act(float x, float c, float d, float g, float b) {
    const float u = fmaf(fmaf(x, c, d), g, b);
    return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896341f));
}
// lane (h, j) holds features 4h .. 4h+3 of row j: afterwards BOTH lanes of the row hold all eight, in feature order.
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second: on two copies of a register it leaves
// [lower | lower] in the first and [upper | upper] in the second.  The four copies are made first (the swap may not read a register the
// instruction before it wrote: hipcc otherwise puts an s_nop in front of every swap).
__device__ __forceinline__ void v8_gather(const float (&own)[4], float (&full)[8]) {
    float c[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) asm volatile("v_mov_b32 %0, %1" : "=v"(c[p]) : "v"(own[p]));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[p]), __float_as_uint(c[p]), false, false);
        full[p] = __uint_as_float(r[0]); full[4 + p] = __uint_as_float(r[1]);
    }
}
// A lane's four rows of a row-major matrix (LDW floats per row), columns k0 .. k0 + 3: the first half of a product is requested at the top
// of its stage (in flight under the LayerNorm / SiLU arithmetic), the second half behind the gather (in flight under the first half's
// multiply-adds) -- four ds_read_b128 per request instead of two with a wait behind each pair, and never more than 32 weight registers live
struct V8W { v8f4 r[4]; };
template <int LDW, typename PF>
__device__ __forceinline__ V8W v8_wload(PF wl, int k0) {
    V8W w;
#pragma unroll
    for (int o = 0; o < 4; ++o) w.r[o] = v8_ld4(wl + o * LDW + k0);
    return w;
}
// acc[o] += sum_{k < 4} W[row_o][k0 + k] * v[k]: each output's sum in feature order, the four outputs advance together (independent chains)
__device__ __forceinline__ void v8_dot4(float (&acc)[4], const V8W& w, const float* v) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = fmaf(w.r[o][k], v[k], acc[o]);
}
// acc += W[:, k0 .. k0 + 7] v with the first half already in registers
template <int LDW, typename PF>
__device__ __forceinline__ void v8_dot8(float (&acc)[4], PF wl, int k0, const V8W& w0, const float (&v)[8]) {
    const V8W w1 = v8_wload<LDW>(wl, k0 + 4);
    v8_dot4(acc, w0, v);
    v8_dot4(acc, w1, v + 4);
}
// 1 / sqrt(v) for v >= 1e-5 (a variance plus the LayerNorm epsilon): one v_rsq_f32 -- the same bits rsqrtf() returns there, without its
// denormal-range scaling (five more instructions per call)
__device__ __forceinline__ float v8_rsq(float v) { return __builtin_amdgcn_rsqf(v); }
// (mean, M2) of a row over its 8 features; the same sums in the same order as acc_stats<8, 1>
__device__ __forceinline__ void v8_stats(const float (&v)[4], float& mean, float& m2) {
    const float s = ((v[0] + v[1]) + v[2]) + v[3];
    mean = xhalf_sum(s) * 0.125f;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) { const float d = v[p] - mean; q = fmaf(d, d, q); }
    m2 = xhalf_sum(q);
}

// One ResidualBlock of width 8 (UNetCF.py:83-95).  x (+ its row statistics) in, x out; UP: the input is cat(x, sk) and the shortcut a Linear.
// P: the block's LDS record; tb: its slice of the current step's time-table row (b1 + Wt silu(temb) + bt); cp: this lane's four values of the
// block's precomputed condition embedding Wc silu(cond) (null on an unconditional tile: silu(0) = 0, only the bias bc, which sits in c2).
// sv: what the training forward keeps for the backward pass (h1, h2: the pre-LayerNorm tensors of stages 2 and 3); a no-op otherwise.
template <bool UP, typename PF, typename PT, typename Save>       // PF: the image's pointer type, PT: the time-table row's (LDS or global each)
__device__ __forceinline__ void v8_block(PF P, PT tb, const int h, float (&x)[4], float& xmean, float& xm2, const float (&sk)[4], float smean,
                                         float sm2, const float* __restrict__ cp, const Save& sv, const int blk) {
    using L = V8BlockL<UP>;
    constexpr int K1 = L::K1;
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cp) cv = ld4(cp);                                   // requested first, added behind stage 2
    // ---- stage 1: h1 = W1 silu(LN1(cat(x, sk))) + [b1 + time bias]
    float mean = xmean, m2 = xm2;
    if (UP) {                                               // Chan merge of the two 8-feature statistics (n0 = n1 = 8)
        const float dd = smean - mean;
        m2 = m2 + sm2 + dd * dd * 4.0f;
        mean = mean + dd * 0.5f;
    }
    float h1[4];
    __builtin_amdgcn_sched_barrier(0);      // the section is one long basic block of loads nothing stores to: without these fences hipcc
                                            // requests the weights of later stages early and spills what the current stage needs
    {
        const PF W1 = P + L::W1 + 4 * h * K1;
        const V8W wx = v8_wload<K1>(W1, 0);
        const float c = v8_rsq(m2 * (1.0f / K1) + kLnEps), d = -mean * c;
        const v8f4 g = v8_ld4(P + L::G1 + 4 * h), b = v8_ld4(P + L::B1 + 4 * h), t = v8_ld4(tb + 4 * h);
        const float a[4] = {v8_act(x[0], c, d, g[0], b[0]), v8_act(x[1], c, d, g[1], b[1]), v8_act(x[2], c, d, g[2], b[2]), v8_act(x[3], c, d, g[3], b[3])};
        float f[8];
        v8_gather(a, f);
        h1[0] = t[0]; h1[1] = t[1]; h1[2] = t[2]; h1[3] = t[3];
        V8W ws;
        if (UP) ws = v8_wload<K1>(W1, 8);
        v8_dot8<K1>(h1, W1, 0, wx, f);
        if (UP) {
            const v8f4 gs = v8_ld4(P + L::G1 + 8 + 4 * h), bs = v8_ld4(P + L::B1 + 8 + 4 * h);
            const float as[4] = {v8_act(sk[0], c, d, gs[0], bs[0]), v8_act(sk[1], c, d, gs[1], bs[1]), v8_act(sk[2], c, d, gs[2], bs[2]),
                                 v8_act(sk[3], c, d, gs[3], bs[3])};
            v8_gather(as, f);
            v8_dot8<K1>(h1, W1, 8, ws, f);
        }
    }
    sv.h1(blk, h1);
    // ---- stage 2: h2 = W2 silu(LN2(h1)) + [b2 + bc] + Wc silu(cond)
    float h2[4];
    __builtin_amdgcn_sched_barrier(0);
    {
        const PF W2 = P + L::W2 + 4 * h * 8;
        const V8W w = v8_wload<8>(W2, 0);
        float mu, q;
        v8_stats(h1, mu, q);
        const float c = v8_rsq(q * 0.125f + kLnEps), d = -mu * c;
        const v8f4 g = v8_ld4(P + L::G2 + 4 * h), b = v8_ld4(P + L::B2 + 4 * h), c2 = v8_ld4(P + L::C2 + 4 * h);
        const float a[4] = {v8_act(h1[0], c, d, g[0], b[0]), v8_act(h1[1], c, d, g[1], b[1]), v8_act(h1[2], c, d, g[2], b[2]), v8_act(h1[3], c, d, g[3], b[3])};
        float f[8];
        v8_gather(a, f);
        h2[0] = c2[0]; h2[1] = c2[1]; h2[2] = c2[2]; h2[3] = c2[3];
        v8_dot8<8>(h2, W2, 0, w, f);
        h2[0] += cv.x; h2[1] += cv.y; h2[2] += cv.z; h2[3] += cv.w;
    }
    sv.h2(blk, h2);
    // ---- stage 3: out = W3 silu(LN3(h2)) + b3 + shortcut(cat(x, sk))
    float o[4];
    __builtin_amdgcn_sched_barrier(0);
    {
        const PF W3 = P + L::W3 + 4 * h * 8;
        const V8W w = v8_wload<8>(W3, 0);
        float mu, q;
        v8_stats(h2, mu, q);
        const float c = v8_rsq(q * 0.125f + kLnEps), d = -mu * c;
        const v8f4 g = v8_ld4(P + L::G3 + 4 * h), b = v8_ld4(P + L::B3 + 4 * h), c3 = v8_ld4(P + L::C3 + 4 * h);
        const float a[4] = {v8_act(h2[0], c, d, g[0], b[0]), v8_act(h2[1], c, d, g[1], b[1]), v8_act(h2[2], c, d, g[2], b[2]), v8_act(h2[3], c, d, g[3], b[3])};
        float f[8];
        v8_gather(a, f);
        o[0] = c3[0]; o[1] = c3[1]; o[2] = c3[2]; o[3] = c3[3];
        V8W sx;
        if (UP) sx = v8_wload<16>(P + L::WSC + 4 * h * 16, 0);
        v8_dot8<8>(o, W3, 0, w, f);
        if (UP) {                                           // Linear shortcut over the raw concat (c3 = b3 + b_shortcut)
            const PF WS = P + L::WSC + 4 * h * 16;
            v8_gather(x, f);
            const V8W ss = v8_wload<16>(WS, 8);
            v8_dot8<16>(o, WS, 0, sx, f);
            v8_gather(sk, f);
            v8_dot8<16>(o, WS, 8, ss, f);
        } else {
            o[0] += x[0]; o[1] += x[1]; o[2] += x[2]; o[3] += x[3];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    v8_stats(o, xmean, xm2);
    x[0] = o[0]; x[1] = o[1]; x[2] = o[2]; x[3] = o[3];
    sv.out(blk, x, xmean, xm2);
}

struct V8NoSave {                 // sampling: nothing of the section leaves the registers
    __device__ __forceinline__ void h1(int, const float (&)[4]) const {}
    __device__ __forceinline__ void h2(int, const float (&)[4]) const {}
    __device__ __forceinline__ void out(int, const float (&)[4], float, float) const {}
    __device__ __forceinline__ void lin_down(const float (&)[4], float, float) const {}
};

struct V8Sec {                    // what the section needs beyond its LDS image
    const float* cond_pre;        // condition embedding of the section's FIRST block, [tiles per pass][64][4]
    long long cp_stride;          // floats from one block's embedding to the next block's
    int tiles_per_pass, uncond_tiles;
};

// xin: this lane's eight values of the 16-wide tensor that enters the section (fragment groups 0 and 1); xout: the same of the 16-wide
// tensor that leaves it, with its row statistics.  (Plain arrays, not the f32x16 of the matrix-core operators: with a partially read and
// element-wise rewritten 16-register vector in the interface hipcc kept two whole tuples alive across the section and spilled both.)
// sv.lin_down / sv.out(blk, ...): the Downsample output and block blk's output (blk counts the section's blocks from 0), for the backward pass.
template <int NB, typename PF, typename PT, typename Save>
__device__ __forceinline__ void v8_section(PF S, PT tb0, const V8Sec& sc, const int tile, const int lane, const float (&xin)[8], float (&xout)[8],
                                           float& xmean, float& xm2, const Save& sv) {
    using L = V8SecL<NB>;
    using BD = V8BlockL<false>;
    using BU = V8BlockL<true>;
    const int h = lane >> 5;
    const int ptile = tile >= sc.tiles_per_pass ? tile - sc.tiles_per_pass : tile;
    const float* cp = tile >= sc.uncond_tiles ? sc.cond_pre + (size_t)ptile * 256 + lane * 4 : nullptr;
    float x[4], sk[NB + 1][4], skm[NB + 1], skq[NB + 1];
    float mean, m2;
    {   // Downsample 16 -> 8 (plain Linear on the raw tensor, UNetCF.py:230-241)
        const float o0[4] = {xin[0], xin[1], xin[2], xin[3]}, o1[4] = {xin[4], xin[5], xin[6], xin[7]};
        float f0[8], f1[8];
        v8_gather(o0, f0);
        v8_gather(o1, f1);
        const v8f4 b = v8_ld4(S + L::LIND_B + 4 * h);
        x[0] = b[0]; x[1] = b[1]; x[2] = b[2]; x[3] = b[3];
        const PF W = S + L::LIND_W + 4 * h * 16;
        v8_dot8<16>(x, W, 0, v8_wload<16>(W, 0), f0);
        v8_dot8<16>(x, W, 8, v8_wload<16>(W, 8), f1);
        v8_stats(x, mean, m2);
        sv.lin_down(x, mean, m2);
    }
    const float none[4] = {0.f, 0.f, 0.f, 0.f};
    int blk = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sk[b][0] = x[0]; sk[b][1] = x[1]; sk[b][2] = x[2]; sk[b][3] = x[3]; skm[b] = mean; skq[b] = m2;
        v8_block<false>(S + L::DOWN + b * BD::SIZE, tb0 + blk * kV8TbStride, h, x, mean, m2, none, 0.f, 0.f, cp ? cp + blk * sc.cp_stride : nullptr, sv, blk);
        ++blk;
    }
    sk[NB][0] = x[0]; sk[NB][1] = x[1]; sk[NB][2] = x[2]; sk[NB][3] = x[3]; skm[NB] = mean; skq[NB] = m2;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        v8_block<false>(S + L::MID + b * BD::SIZE, tb0 + blk * kV8TbStride, h, x, mean, m2, none, 0.f, 0.f, cp ? cp + blk * sc.cp_stride : nullptr, sv, blk);
        ++blk;
    }
#pragma unroll
    for (int b = 0; b <= NB; ++b) {
        v8_block<true>(S + L::UP + b * BU::SIZE, tb0 + blk * kV8TbStride, h, x, mean, m2, sk[NB - b], skm[NB - b], skq[NB - b],
                       cp ? cp + blk * sc.cp_stride : nullptr, sv, blk);
        ++blk;
    }
    {   // Upsample 8 -> 16: this lane's output features are 4h .. 4h+3 (group 0) and 8 + 4h .. 8 + 4h+3 (group 1)
        float f[8];
        v8_gather(x, f);
        const v8f4 b0 = v8_ld4(S + L::LINU_B + 4 * h), b1 = v8_ld4(S + L::LINU_B + 8 + 4 * h);
        float y0[4] = {b0[0], b0[1], b0[2], b0[3]}, y1[4] = {b1[0], b1[1], b1[2], b1[3]};
        const PF W0 = S + L::LINU_W + 4 * h * 8;
        const PF W1 = S + L::LINU_W + (8 + 4 * h) * 8;
        v8_dot8<8>(y0, W0, 0, v8_wload<8>(W0, 0), f);
        v8_dot8<8>(y1, W1, 0, v8_wload<8>(W1, 0), f);
        // row statistics over 16 features, summed in fragment order as the matrix-core kernels do (linear_reg_h)
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) s += y0[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) s += y1[p];
        const float m = xhalf_sum(s) * (1.0f / 16);
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const float d = y0[p] - m; q = fmaf(d, d, q); }
#pragma unroll
        for (int p = 0; p < 4; ++p) { const float d = y1[p] - m; q = fmaf(d, d, q); }
        xmean = m; xm2 = xhalf_sum(q);
#pragma unroll
        for (int p = 0; p < 4; ++p) { xout[p] = y0[p]; xout[4 + p] = y1[p]; }
    }
}

}  // namespace dsg
