// DiffSG denoiser hot path -- gfx950 (MI355X / CDNA4) device code.
//
// Every activation lives in the "fragment layout" of v_mfma_f32_32x32x2_f32 so that a chain of
// Linear layers never moves data across lanes:
//
//   a wave owns a TILE of 32 batch rows.  Lane l = 32*h + j holds row j; a width-w tensor is w/8
//   GROUPS of 8 features; group G is one float4 per lane holding features 8G + 4h + {0,1,2,3}.
//
//   * as MFMA B operand  (B[k][j], lane holds k = h):  element p of group G is k-step p of G;
//   * the MFMA result D[i][j] for an output tile nt has out-feature i = (r&3) + 8(r>>2) + 4h in
//     accumulator register r, i.e. registers 4q..4q+3 of tile nt ARE group 4nt+q of the output.
//
//   So the accumulator of one Linear is, register for register, the B operand of the next one, and
//   LayerNorm / SiLU / bias / residual are plain per-register VALU work (row statistics need one
//   exchange between the two lane halves).  Weights are pre-packed (k_pack_grouped) into the
//   matching A-operand order so that every weight fetch is one coalesced 1 KiB float4 wave load.
//
// In HBM a tensor is [tile][group][lane(64)][4] floats (1 KiB per tile-group, fully coalesced) plus
// per-row LayerNorm statistics (mean, M2) written by the producer.
//
// Reference semantics: /root/reference/ddpm_opt/UNetCF.py (cited per kernel below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;          // batch rows per wave tile
constexpr int kWavesPerBlock = 4;  // one wave per SIMD
constexpr float kLnEps = 1e-5f;    // nn.LayerNorm default (UNetCF.py:60)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Swish, UNetCF.py:14: x * sigmoid(x).  exp(-v) = 2^(-v log2 e) on v_exp_f32 (1 ulp) with the rounding error of the
// product and the low part of log2 e folded back in (2^(t+e) ~ 2^t (1 + e ln 2)), so the result stays within ~2 ulp of
// expf() at a third of its VALU cost; a bare v_exp_f32(-v*log2e) is |v|*6e-8 off, which at omega = 500 tripled the
// end-to-end deviation from the float64 trajectory.
__device__ __forceinline__ float fast_exp_neg(float v) {
    constexpr float c_hi = -1.44269504088896341f, c_lo = -1.925963033500671e-08f;  // -log2(e) split
    const float t = v * c_hi;
    const float e = fmaf(v, c_lo, fmaf(v, c_hi, -t));
    const float p = __builtin_amdgcn_exp2f(t);
    return fmaf(p * e, 0.693147180559945f, p);
}
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + fast_exp_neg(v)); }

// v[lane] + v[lane ^ 32] in every lane.  gfx950's v_permlane32_swap exchanges the upper half of one register with the lower half of
// another at VALU speed; the shuffle it replaces is a ds_bpermute through the LDS crossbar (~150 cycles behind an lgkmcnt wait, and
// every LayerNorm statistic of every stage sits on it).  Same two addends in every lane: same bits.
__device__ __forceinline__ unsigned xhalf_swap_lo(unsigned v, unsigned& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    hi = r[1];
    return r[0];
}
__device__ __forceinline__ float xhalf_sum(float v) {
    unsigned hi;
    const unsigned lo = xhalf_swap_lo(__float_as_uint(v), hi);
    return __uint_as_float(lo) + __uint_as_float(hi);
}
__device__ __forceinline__ unsigned xhalf_max(unsigned m) {
    unsigned hi;
    const unsigned lo = xhalf_swap_lo(m, hi);
    return lo > hi ? lo : hi;
}

#define DSG_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)

// Weight fragments of one k-group for NT output tiles: wp points at packed[(0*KG + g)*256 + lane*4].
// Loads are issued one k-group AHEAD of the MFMAs that consume them (software prefetch): hipcc otherwise emits
// load -> s_waitcnt vmcnt(0) -> 4 MFMAs per tile, exposing a full L2 round trip every 256 MFMA cycles.
template <int NT>
__device__ __forceinline__ void load_wfrag(float4 (&w)[NT], const float* __restrict__ wp, size_t nt_stride) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w[nt] = ld4(wp + nt * nt_stride);
}

// One k-group (4 k-steps) into NT output tiles from preloaded fragments.
template <int NT>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[NT], const float4 (&w)[NT], float b0, float b1, float b2, float b3) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        DSG_MFMA(acc[nt], w[nt].x, b0);
        DSG_MFMA(acc[nt], w[nt].y, b1);
        DSG_MFMA(acc[nt], w[nt].z, b2);
        DSG_MFMA(acc[nt], w[nt].w, b3);
    }
}

__device__ __forceinline__ float4 ln_silu4(float4 x, float mean, float rstd, float4 gm, float4 bt) {
    return make_float4(silu(fmaf((x.x - mean) * rstd, gm.x, bt.x)), silu(fmaf((x.y - mean) * rstd, gm.y, bt.y)),
                       silu(fmaf((x.z - mean) * rstd, gm.z, bt.z)), silu(fmaf((x.w - mean) * rstd, gm.w, bt.w)));
}

// Stream `groups` k-groups of a fragment tensor from memory into the chain (runtime loop, prefetch distance 1).
//   xp     tile base of the tensor + lane*4 (group stride 256)
//   wp     packed weights of the first of these groups + lane*4 (group stride 256, tile stride nt_stride)
//   gamma  LayerNorm parameters of the first group + 4h (LNACT), group stride 8
template <int NT, bool LNACT>
__device__ __forceinline__ void chain_from_mem(f32x16 (&acc)[NT], const float* __restrict__ xp, int groups, const float* __restrict__ wp,
                                               size_t nt_stride, const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float mean, float rstd) {
    if (groups <= 0) return;
    if constexpr (NT == 1) {
        // one accumulator tile (the narrow run, <= 4 groups per tensor): "current" and "next" sets, copied every group -- 16 moves per
        // group, but the fused narrow kernel has no registers for two full sets (the ping-pong form below spilled there)
        float4 wn[NT], xn, gmn = make_float4(0.f, 0.f, 0.f, 0.f), btn = gmn;
        load_wfrag<NT>(wn, wp, nt_stride);
        xn = ld4(xp);
        if (LNACT) { gmn = ld4(gamma); btn = ld4(beta); }
        for (int g = 0; g < groups; ++g) {
            float4 wc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
            float4 xv = xn;
            const float4 gm = gmn, bt = btn;
            if (g + 1 < groups) {
                load_wfrag<NT>(wn, wp + (size_t)(g + 1) * 256, nt_stride);
                xn = ld4(xp + (size_t)(g + 1) * 256);
                if (LNACT) { gmn = ld4(gamma + 8 * (g + 1)); btn = ld4(beta + 8 * (g + 1)); }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs (hipcc otherwise sinks it)
            if (LNACT) xv = ln_silu4(xv, mean, rstd, gm, bt);
            mfma_group<NT>(acc, wc, xv.x, xv.y, xv.z, xv.w);
        }
        return;
    }
    // Two operand sets, the loop unrolled by two: each set is loaded and consumed in the same registers (ping-pong).  The round-1
    // form kept "current" and "next" sets and copied next -> current every group: 28 register moves per group at NT = 4 -- and hipcc
    // copied them back at the loop end -- more instructions than a raw group's arithmetic (round 5, disassembly).
    float4 w0[NT], w1[NT], x0, x1, gm0 = make_float4(0.f, 0.f, 0.f, 0.f), bt0 = gm0, gm1 = gm0, bt1 = gm0;
    load_wfrag<NT>(w0, wp, nt_stride);
    x0 = ld4(xp);
    if (LNACT) { gm0 = ld4(gamma); bt0 = ld4(beta); }
    int g = 0;
    for (; g + 1 < groups; g += 2) {
        load_wfrag<NT>(w1, wp + (size_t)(g + 1) * 256, nt_stride);
        x1 = ld4(xp + (size_t)(g + 1) * 256);
        if (LNACT) { gm1 = ld4(gamma + 8 * (g + 1)); bt1 = ld4(beta + 8 * (g + 1)); }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs (hipcc otherwise sinks it)
        {
            float4 xv = x0;
            if (LNACT) xv = ln_silu4(xv, mean, rstd, gm0, bt0);
            mfma_group<NT>(acc, w0, xv.x, xv.y, xv.z, xv.w);
        }
        if (g + 2 < groups) {
            load_wfrag<NT>(w0, wp + (size_t)(g + 2) * 256, nt_stride);
            x0 = ld4(xp + (size_t)(g + 2) * 256);
            if (LNACT) { gm0 = ld4(gamma + 8 * (g + 2)); bt0 = ld4(beta + 8 * (g + 2)); }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            float4 xv = x1;
            if (LNACT) xv = ln_silu4(xv, mean, rstd, gm1, bt1);
            mfma_group<NT>(acc, w1, xv.x, xv.y, xv.z, xv.w);
        }
    }
    if (g < groups) {                       // odd count: the last group sits in set 0
        float4 xv = x0;
        if (LNACT) xv = ln_silu4(xv, mean, rstd, gm0, bt0);
        mfma_group<NT>(acc, w0, xv.x, xv.y, xv.z, xv.w);
    }
}

// The same chain with the group count known at compile time (the >= 64-wide blocks: N / 8 groups per input tensor), fully unrolled:
// the D + 1 operand sets are compile-time slots, so the prefetch rotates by renaming.  As a runtime loop hipcc rotated them with
// register moves -- 56 v_mov_b32 per group, twice the LayerNorm + SiLU arithmetic of a raw group -- and waited for the NEXT group's
// loads a third of the way into the current group's MFMAs (round 5, disassembly of k_resblock<128, true>: 3 600 moves per tile).
// Same products in the same order as chain_from_mem: same bits.
template <int NT, bool LNACT, int GROUPS, int D = 1>
__device__ __forceinline__ void chain_from_mem_fixed(f32x16 (&acc)[NT], const float* __restrict__ xp, const float* __restrict__ wp,
                                                     size_t nt_stride, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float mean, float rstd) {
    float4 wb[D + 1][NT], xb[D + 1], gb[D + 1], bb[D + 1];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < GROUPS) {
            load_wfrag<NT>(wb[d], wp + (size_t)d * 256, nt_stride);
            xb[d] = ld4(xp + (size_t)d * 256);
            if (LNACT) { gb[d] = ld4(gamma + 8 * d); bb[d] = ld4(beta + 8 * d); }
        }
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        if (g + D < GROUPS) {
            constexpr int M = D + 1;
            const int s = (g + D) % M;
            load_wfrag<NT>(wb[s], wp + (size_t)(g + D) * 256, nt_stride);
            xb[s] = ld4(xp + (size_t)(g + D) * 256);
            if (LNACT) { gb[s] = ld4(gamma + 8 * (g + D)); bb[s] = ld4(beta + 8 * (g + D)); }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs
        float4 xv = xb[g % (D + 1)];
        if (LNACT) xv = ln_silu4(xv, mean, rstd, gb[g % (D + 1)], bb[g % (D + 1)]);
        mfma_group<NT>(acc, wb[g % (D + 1)], xv.x, xv.y, xv.z, xv.w);
    }
}
// FIX > 0: the caller has checked that the tensor has exactly FIX groups (k_resblock picks the body per launch: a check in here
// made hipcc join the two forms' accumulators with 32 - 64 register moves behind every chain)
template <int NT, bool LNACT, int FIX>
__device__ __forceinline__ void chain_from_mem_n(f32x16 (&acc)[NT], const float* __restrict__ xp, int groups, const float* __restrict__ wp,
                                                 size_t nt_stride, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float mean, float rstd) {
    if constexpr (FIX > 0) chain_from_mem_fixed<NT, LNACT, FIX>(acc, xp, wp, nt_stride, gamma, beta, mean, rstd);
    else chain_from_mem<NT, LNACT>(acc, xp, groups, wp, nt_stride, gamma, beta, mean, rstd);
}

// acc <- per-feature vector (padded to NT*32 floats) in accumulator order.
template <int NT>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[NT], const float* __restrict__ vec, int h) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = ld4(vec + 32 * nt + 8 * q + 4 * h);
            acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
        }
}

template <int NT>
__device__ __forceinline__ void acc_add(f32x16 (&acc)[NT], const float* __restrict__ vec, int h) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = ld4(vec + 32 * nt + 8 * q + 4 * h);
            acc[nt][4 * q + 0] += b.x; acc[nt][4 * q + 1] += b.y; acc[nt][4 * q + 2] += b.z; acc[nt][4 * q + 3] += b.w;
        }
}

// Row statistics of an accumulator-resident tensor of true width N: mean and M2 = sum (x-mean)^2.
template <int N, int NT>
__device__ __forceinline__ void acc_stats(const f32x16 (&acc)[NT], int h, float& mean, float& m2) {
    constexpr int NG = (N + 7) / 8;
    float s = 0.f;
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (8 * G + 4 * h + p < N) s += acc[G >> 2][4 * (G & 3) + p];
    mean = xhalf_sum(s) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (8 * G + 4 * h + p < N) {
                const float d = acc[G >> 2][4 * (G & 3) + p] - mean;
                q = fmaf(d, d, q);
            }
    m2 = xhalf_sum(q);
}

// B operands for a chain whose input is an accumulator: act = silu(LN(acc)), then MFMA into `out`.
// ResidualBlock stages 2 and 3 (UNetCF.py:92,94).
// First-group operands of a register-fed chain, issued by the caller well before the chain starts (narrow blocks are
// latency-bound: this takes one L2 round trip per stage off the critical path).
template <int NT>
struct ChainHead { float4 w[NT]; float4 gm, bt; };

template <int N, int NT>
__device__ __forceinline__ void chain_head_load(ChainHead<NT>& hd, const float* __restrict__ wp, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int lane, int h) {
    constexpr int NG = (N + 7) / 8;
    load_wfrag<NT>(hd.w, wp + lane * 4, (size_t)NG * 256);
    hd.gm = ld4(gamma + 4 * h);
    hd.bt = ld4(beta + 4 * h);
}

template <int N, int NT>
__device__ __forceinline__ void chain_from_acc(f32x16 (&out)[NT], const f32x16 (&in)[NT], const float* __restrict__ wp,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float mean, float rstd, int lane, int h, const ChainHead<NT>* head = nullptr) {
    constexpr int NG = (N + 7) / 8;
    const size_t nt_stride = (size_t)NG * 256;
    float4 wn[NT], gmn, btn;
    if (head) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wn[nt] = head->w[nt];
        gmn = head->gm; btn = head->bt;
    } else {
        load_wfrag<NT>(wn, wp + lane * 4, nt_stride);
        gmn = ld4(gamma + 4 * h);
        btn = ld4(beta + 4 * h);
    }
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        float4 wc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wc[nt] = wn[nt];
        const float4 gm = gmn, bt = btn;
        if (G + 1 < NG) {
            load_wfrag<NT>(wn, wp + (size_t)(G + 1) * 256 + lane * 4, nt_stride);
            gmn = ld4(gamma + 8 * (G + 1) + 4 * h);
            btn = ld4(beta + 8 * (G + 1) + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs
        const float4 b = ln_silu4(make_float4(in[G >> 2][4 * (G & 3) + 0], in[G >> 2][4 * (G & 3) + 1], in[G >> 2][4 * (G & 3) + 2],
                                              in[G >> 2][4 * (G & 3) + 3]), mean, rstd, gm, bt);
        mfma_group<NT>(out, wc, b.x, b.y, b.z, b.w);
    }
}

// Register-fed segment of a chain whose weights have tile stride `nt_stride` (stage 1 / shortcut of a block whose in0 is the narrow
// run's running tensor, <= 32 wide: one accumulator tile): every load of the segment first, then LayerNorm + SiLU (LNACT) or the raw
// values, group by group in the order chain_from_mem walks them -- same products, same order, same bits.
template <int NT, bool LNACT>
__device__ __forceinline__ void chain_from_reg(f32x16 (&out)[NT], const f32x16& in, int groups, const float* __restrict__ wp /* + lane*4 */,
                                               size_t nt_stride, const float* __restrict__ gamma /* + 4h */, const float* __restrict__ beta,
                                               float mean, float rstd) {
    float4 w[4][NT], gm[4], bt[4];
#pragma unroll
    for (int G = 0; G < 4; ++G)
        if (G < groups) {
            load_wfrag<NT>(w[G], wp + (size_t)G * 256, nt_stride);
            if (LNACT) { gm[G] = ld4(gamma + 8 * G); bt[G] = ld4(beta + 8 * G); }
        }
#pragma unroll
    for (int G = 0; G < 4; ++G)
        if (G < groups) {
            float4 b = make_float4(in[4 * G], in[4 * G + 1], in[4 * G + 2], in[4 * G + 3]);
            if (LNACT) b = ln_silu4(b, mean, rstd, gm[G], bt[G]);
            mfma_group<NT>(out, w[G], b.x, b.y, b.z, b.w);
        }
}

// A fragment-layout tensor in HBM.
// Cycle stamps for measurement builds (-DDSG_CYCLE_STAMPS, tools/cycle_stamps.sh): (cycle counter << 16 | tag) appended to a device array by
// lane 0 of the selected wave; dsg_stamps_fetch (dsg_api.hip) reads and clears it.  Compiled out otherwise.
#ifdef DSG_CYCLE_STAMPS
__device__ unsigned long long dsg_stamp_buf[8192];
__device__ int dsg_stamp_n;
#define DSG_STAMP(cond, tag)                                                                      \
    do {                                                                                          \
        if (cond) {                                                                               \
            const unsigned long long t_ = __builtin_readcyclecounter();                           \
            if (lane == 0) {                                                                      \
                const int k_ = atomicAdd(&dsg_stamp_n, 1);                                        \
                if (k_ < 8192) dsg_stamp_buf[k_] = (t_ << 16) | (unsigned long long)(tag);        \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#else
#define DSG_STAMP(cond, tag) do {} while (0)
#endif

// hi / lo half planes of NT out tiles of one k16-step of a packed split-f16 weight matrix (dsg_split.hpp, dsg_train_split.hpp)
template <int NT>
struct HFrag { uint4 hi[NT], lo[NT]; };

struct Seg {
    const float* data;   // [tiles][groups][64][4]
    const float* stats;  // [tiles*32][2] = (mean, M2) per row, may be null when unused
    int groups;          // ceil(width/8)
    int width;           // true feature count
    int wrap;            // 0, or tiles per pass: both CFG passes read the first pass's tiles (feature_proj(y) is the same
                         // tensor for the unconditional and the conditional pass; split path, reverse loop only)
};
__device__ __forceinline__ int seg_tile(const Seg& s, int tile) { return (s.wrap && tile >= s.wrap) ? tile - s.wrap : tile; }

// A pointer that a kernel reads from a descriptor TABLE in memory (operator tables of the fused narrow kernels, the weight-gradient
// descriptors) is a generic pointer to hipcc, and every access through it becomes a FLAT instruction.  A FLAT access counts on the
// LDS counter as well as on the vector-memory counter and may return out of order with LDS data, so hipcc waits for ALL of a wave's
// outstanding memory operations at every LDS read that follows one -- the prefetches of these latency-bound kernels stopped
// overlapping with anything that touched LDS.  as_global() declares such a pointer global: an address-space cast there and back
// with an empty asm in between (hipcc folds a bare round trip away; behind the asm it infers the address space of every access
// and also keeps the base in scalar registers: `global_load v, v_off, s[base]`).  No instruction is emitted.  The pointer must be
// wave-uniform (a descriptor field is) and must not have been redirected to LDS; globalize() does it for a descriptor's pointers.
template <typename T> __device__ __forceinline__ T* as_global(T* p) {
    auto g = (__attribute__((address_space(1))) T*)p;
    asm("" : "+s"(g));
    return (T*)g;
}
__device__ __forceinline__ void globalize(Seg& s) { s.data = as_global(s.data); s.stats = as_global(s.stats); }

// ---------------------------------------------------------------------------------------------
// ResidualBlock forward (UNetCF.py:83-95), one wave per 32-row tile:
//   h1 = W1 silu(LN1(x)) + [b1 + time bias]           time bias row chosen by step (sampling) or ts[row]
//   h2 = W2 silu(LN2(h1)) + [b2 + bc] + Wc silu(cond)  cond term skipped for tiles < uncond_tiles
//   out = W3 silu(LN3(h2)) + b3 + shortcut(x)          shortcut = identity or Linear(cat) (UNetCF.py:72-75)
// x is the concatenation of up to two HBM tensors (skip concat, UNetCF.py:351, never materialised).
// ---------------------------------------------------------------------------------------------
struct BlockArgs {
    Seg in0, in1;
    const float* W1;      // packed [NT][KG][256], KG = in0.groups + in1.groups
    const float* gamma1;  // [KG*8] group order
    const float* beta1;
    const float* tbias;   // [entries][tb_stride]; this block's slice starts at tbias (+ entry*tb_stride)
    const int* step_ptr;  // device step counter (sampling) or null
    const int* ts;        // per-row entry index (training / generic forward) or null
    int tb_stride;
    const float* W2;      // packed [NT][NG][256]
    const float* gamma2;
    const float* beta2;
    const float* c2;      // b2 + bc, padded NT*32
    const float* Wc;      // packed [NT][CG][256]
    const float* condfrag;  // [tiles_per_pass][CG][64][4] = silu(cond * mask)
    int cond_groups;
    const float* cond_pre;  // sampling: Wc silu(cond) of THIS block, precomputed once per call [tiles_per_pass][NG][64][4]
                            // (cond does not change over the T steps), or null -> run the GEMM here
    const float* W3;
    const float* gamma3;
    const float* beta3;
    const float* c3;      // b3 (+ b_shortcut)
    const float* Wsc;     // packed [NT][KG][256] or null (identity)
    float* out;           // [tiles][NG][64][4]
    float* out_stats;     // [tiles*32][2]
    float* save_h1;       // training: pre-LN2 / pre-LN3 tensors for the backward pass, or null
    float* save_h2;
    int ntiles;           // all passes
    int tiles_per_pass;
    int uncond_tiles;     // tiles [0, uncond_tiles) skip the condition term (Swish(0) = 0)
    int nrows;            // valid rows per pass
    int* range_flag;      // split path: set to 1 when a RAW operand row may exceed fp16's range (dsg_range_status), or null
    // host-side constants of the LN1 statistics merge (Chan) over the concat: n0*n1/(n0+n1), n1/(n0+n1), 1/(n0+n1) -- three
    // float divisions per wave and block otherwise (~30 VALU; the narrow blocks have ~170 VALU of real work)
    float chan_w, chan_f, inv_nin;
};
// tensors, indices and flags (what stays in global memory when a kernel keeps the block's planes and vectors in LDS) / the rest
__device__ __forceinline__ void globalize_io(BlockArgs& a) {
    globalize(a.in0); globalize(a.in1);
    a.step_ptr = as_global(a.step_ptr); a.ts = as_global(a.ts); a.condfrag = as_global(a.condfrag); a.cond_pre = as_global(a.cond_pre);
    a.out = as_global(a.out); a.out_stats = as_global(a.out_stats); a.save_h1 = as_global(a.save_h1); a.save_h2 = as_global(a.save_h2);
    a.range_flag = as_global(a.range_flag);
}
__device__ __forceinline__ void globalize_params(BlockArgs& a) {
    a.W1 = as_global(a.W1); a.gamma1 = as_global(a.gamma1); a.beta1 = as_global(a.beta1); a.tbias = as_global(a.tbias);
    a.W2 = as_global(a.W2); a.gamma2 = as_global(a.gamma2); a.beta2 = as_global(a.beta2); a.c2 = as_global(a.c2); a.Wc = as_global(a.Wc);
    a.W3 = as_global(a.W3); a.gamma3 = as_global(a.gamma3); a.beta3 = as_global(a.beta3); a.c3 = as_global(a.c3); a.Wsc = as_global(a.Wsc);
}

// XIN / XOUT (round 5, the exact path's narrow run): in0 is handed over in registers `xr` with its row statistics instead of read from
// memory / the output is handed on the same way and stored only if `store_out` (a skip tensor, the run's last tensor).  N <= 32.
template <int N, bool SCLIN, bool XIN = false, bool XOUT = false, bool FIXED = false>
__device__ __forceinline__ void resblock_body(const BlockArgs& a, const int tile, const int lane, f32x16* xr = nullptr, float* xr_mean = nullptr,
                                              float* xr_m2 = nullptr, const bool store_out = true) {
    constexpr int NG = (N + 7) / 8, NT = (N + 31) / 32;
    constexpr int FIXG = FIXED ? N / 8 : 0;        // FIXED: in0 (and in1 of a concat) have exactly N / 8 groups (chain_from_mem_fixed)
    static_assert(!XIN || NT == 1, "register input: one accumulator tile");      // XOUT: xr[NT] (k_resblock_lin hands a wide block's output on)
    const int h = lane >> 5, j = lane & 31;
    const int ptile = tile % a.tiles_per_pass;
    const int KG = a.in0.groups + a.in1.groups;

    // ---- LN1 statistics from the producers' per-row (mean, M2), combined over the concat (Chan)
    float mean1, rstd1;
    {
        float2 s0;
        if constexpr (XIN) s0 = make_float2(*xr_mean, *xr_m2);
        else s0 = reinterpret_cast<const float2*>(a.in0.stats)[(size_t)tile * 32 + j];
        float mean = s0.x, m2 = s0.y, n = (float)a.in0.width;
        if (a.in1.groups) {
            const float2 s1 = reinterpret_cast<const float2*>(a.in1.stats)[(size_t)tile * 32 + j];
            const float n1 = (float)a.in1.width, nt_ = n + n1;
            const float d = s1.x - mean;
            m2 = m2 + s1.y + d * d * (n * n1 / nt_);
            mean = mean + d * (n1 / nt_);
            n = nt_;
        }
        mean1 = mean;
        rstd1 = rsqrtf(m2 / n + kLnEps);
    }

    // narrow blocks: issue the first operands of stages 2 and 3 now
    constexpr bool EARLY = NT == 1;
    ChainHead<NT> head2, head3;
    if (EARLY) {
        chain_head_load<N, NT>(head2, a.W2, a.gamma2, a.beta2, lane, h);
        chain_head_load<N, NT>(head3, a.W3, a.gamma3, a.beta3, lane, h);
    }

    // ---- stage 1
    f32x16 acc1[NT];
    {
        int entry = 0;
        if (a.ts) {
            int row = ptile * 32 + j;
            row = row < a.nrows ? row : a.nrows - 1;
            entry = a.ts[row];
        } else if (a.step_ptr) {
            entry = *a.step_ptr;
        }
        acc_init<NT>(acc1, a.tbias + (size_t)entry * a.tb_stride, h);
    }
    {
        const size_t nt_stride = (size_t)KG * 256;
        if constexpr (XIN) chain_from_reg<NT, true>(acc1, (*xr), a.in0.groups, a.W1 + lane * 4, nt_stride, a.gamma1 + 4 * h, a.beta1 + 4 * h, mean1, rstd1);
        else
        chain_from_mem_n<NT, true, FIXG>(acc1, a.in0.data + (size_t)tile * a.in0.groups * 256 + lane * 4, a.in0.groups, a.W1 + lane * 4, nt_stride,
                                 a.gamma1 + 4 * h, a.beta1 + 4 * h, mean1, rstd1);
        if (a.in1.groups)
            chain_from_mem_n<NT, true, FIXG>(acc1, a.in1.data + (size_t)tile * a.in1.groups * 256 + lane * 4, a.in1.groups,
                                     a.W1 + (size_t)a.in0.groups * 256 + lane * 4, nt_stride, a.gamma1 + 8 * a.in0.groups + 4 * h,
                                     a.beta1 + 8 * a.in0.groups + 4 * h, mean1, rstd1);
    }
    if (a.save_h1) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h1 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc1[G >> 2][4 * (G & 3)], acc1[G >> 2][4 * (G & 3) + 1], acc1[G >> 2][4 * (G & 3) + 2],
                            acc1[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 2 (+ condition embedding accumulated into the same chain)
    f32x16 acc2[NT];
    {
        float mean, m2;
        acc_stats<N, NT>(acc1, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        acc_init<NT>(acc2, a.c2, h);
        chain_from_acc<N, NT>(acc2, acc1, a.W2, a.gamma2, a.beta2, mean, rstd, lane, h, EARLY ? &head2 : nullptr);
    }
    if (tile >= a.uncond_tiles) {
        if (a.cond_pre) {
            const float* cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                const float4 cv = ld4(cp + (size_t)G * 256);
                acc2[G >> 2][4 * (G & 3) + 0] += cv.x; acc2[G >> 2][4 * (G & 3) + 1] += cv.y;
                acc2[G >> 2][4 * (G & 3) + 2] += cv.z; acc2[G >> 2][4 * (G & 3) + 3] += cv.w;
            }
        } else {
            chain_from_mem<NT, false>(acc2, a.condfrag + (size_t)ptile * a.cond_groups * 256 + lane * 4, a.cond_groups, a.Wc + lane * 4,
                                      (size_t)a.cond_groups * 256, nullptr, nullptr, 0.f, 1.f);
        }
    }
    if (a.save_h2) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h2 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc2[G >> 2][4 * (G & 3)], acc2[G >> 2][4 * (G & 3) + 1], acc2[G >> 2][4 * (G & 3) + 2],
                            acc2[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 3 (+ shortcut)
    f32x16 (&acc3)[NT] = acc1;  // h1 is dead
    {
        float mean, m2;
        acc_stats<N, NT>(acc2, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        acc_init<NT>(acc3, a.c3, h);
        chain_from_acc<N, NT>(acc3, acc2, a.W3, a.gamma3, a.beta3, mean, rstd, lane, h, EARLY ? &head3 : nullptr);
    }
    if (SCLIN) {
        const size_t nt_stride = (size_t)KG * 256;
        if constexpr (XIN) chain_from_reg<NT, false>(acc3, (*xr), a.in0.groups, a.Wsc + lane * 4, nt_stride, nullptr, nullptr, 0.f, 1.f);
        else
        chain_from_mem_n<NT, false, FIXG>(acc3, a.in0.data + (size_t)tile * a.in0.groups * 256 + lane * 4, a.in0.groups, a.Wsc + lane * 4, nt_stride,
                                  nullptr, nullptr, 0.f, 1.f);
        if (a.in1.groups)
            chain_from_mem_n<NT, false, FIXG>(acc3, a.in1.data + (size_t)tile * a.in1.groups * 256 + lane * 4, a.in1.groups,
                                      a.Wsc + (size_t)a.in0.groups * 256 + lane * 4, nt_stride, nullptr, nullptr, 0.f, 1.f);
    } else if constexpr (XIN) {
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            acc3[0][4 * G + 0] += (*xr)[4 * G + 0]; acc3[0][4 * G + 1] += (*xr)[4 * G + 1];
            acc3[0][4 * G + 2] += (*xr)[4 * G + 2]; acc3[0][4 * G + 3] += (*xr)[4 * G + 3];
        }
    } else {
        const float* xp = a.in0.data + (size_t)tile * NG * 256 + lane * 4;
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            const float4 xv = ld4(xp + (size_t)G * 256);
            acc3[G >> 2][4 * (G & 3) + 0] += xv.x; acc3[G >> 2][4 * (G & 3) + 1] += xv.y;
            acc3[G >> 2][4 * (G & 3) + 2] += xv.z; acc3[G >> 2][4 * (G & 3) + 3] += xv.w;
        }
    }

    // ---- store + statistics for the consumer's LayerNorm
    {
        float mean, m2;
        acc_stats<N, NT>(acc3, h, mean, m2);
        if constexpr (XOUT) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xr[nt] = acc3[nt];
            *xr_mean = mean; *xr_m2 = m2;
        }
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

// FIXED (chosen by the host, launch_res): in0 -- and in1 of a concat -- have exactly N / 8 groups, as in every shipped net: the unrolled
// chains.  A kernel of its own per form: both bodies in one kernel cost the larger register count for both and spills at N = 64.
// Waves per SIMD: the unrolled chains leave hipcc free to take 332 / 192 registers at N = 128 / 64 -- one / two waves per SIMD where
// the runtime-loop form runs two / four; bounded to what that form has.
template <int N, bool SCLIN, bool FIXED = false>
__global__ __launch_bounds__(256, !FIXED ? 1 : (N >= 128 ? 2 : 4)) void k_resblock(const BlockArgs a) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= a.ntiles) return;
    resblock_body<N, SCLIN, false, false, FIXED>(a, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Plain Linear with optional LayerNorm+SiLU prologue and row-major I/O at the network boundary:
//   feature_proj (UNetCF.py:328)   row-major y[B][D] -> fragment x0          (both passes read the same y)
//   Down/Upsample (UNetCF.py:230-257)  fragment -> fragment
//   final (UNetCF.py:356)          fragment -> LN -> SiLU -> Linear -> row-major eps[pass][B][D]
// ---------------------------------------------------------------------------------------------
struct LinArgs {
    Seg in;                 // fragment input (IN_FRAG)
    const float* in_rm;     // row-major input [nrows][in_width] (IN_ROWMAJOR)
    int in_width;
    int in_groups;
    const float* W;         // packed [NT][KG][256]
    const float* bias;      // padded NT*32
    const float* gamma;     // LN prologue (LNACT)
    const float* beta;
    float* out;             // fragment out
    float* out_stats;
    float* out_rm;          // row-major out [npass][nrows][out_width]
    int out_width;
    int ntiles, tiles_per_pass, nrows;
    int* advance_step;      // reverse loop: the step's first operator (feature_proj, which does not read the step index)
                            // moves it on - no kernel of its own, and nothing else is running that could read it
    int* range_flag;        // split path: see BlockArgs::range_flag
    float inv_in_w, inv_out_w;   // 1 / in_width, 1 / out_width (host)
};
__device__ __forceinline__ void globalize_io(LinArgs& a) {
    globalize(a.in); a.in_rm = as_global(a.in_rm); a.out = as_global(a.out); a.out_stats = as_global(a.out_stats); a.out_rm = as_global(a.out_rm);
    a.advance_step = as_global(a.advance_step); a.range_flag = as_global(a.range_flag);
}
__device__ __forceinline__ void globalize_params(LinArgs& a) {
    a.W = as_global(a.W); a.bias = as_global(a.bias); a.gamma = as_global(a.gamma); a.beta = as_global(a.beta);
}

enum { IN_FRAG = 0, IN_ROWMAJOR = 1 };
enum { OUT_FRAG = 0, OUT_ROWMAJOR = 1 };

// The output side of a Linear: fragment layout with the row statistics of the true width, or row-major rows at the network boundary.
template <int NT, int OUTMODE>
__device__ __forceinline__ void linear_store(const LinArgs& a, const int tile, const int lane, const f32x16 (&acc)[NT]) {
    const int h = lane >> 5, j = lane & 31;
    const int ptile = tile % a.tiles_per_pass;
    const int pass = tile / a.tiles_per_pass;
    const int row = ptile * 32 + j;
    if (OUTMODE == OUT_FRAG) {
        const int NG = (a.out_width + 7) / 8;
        // statistics over the true width
        float s = 0.f;
#pragma unroll
        for (int G = 0; G < NT * 4; ++G)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (8 * G + 4 * h + p < a.out_width) s += acc[G >> 2][4 * (G & 3) + p];
        const float m = xhalf_sum(s) / (float)a.out_width;
        float q = 0.f;
#pragma unroll
        for (int G = 0; G < NT * 4; ++G)
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (8 * G + 4 * h + p < a.out_width) {
                    const float d = acc[G >> 2][4 * (G & 3) + p] - m;
                    q = fmaf(d, d, q);
                }
        q = xhalf_sum(q);
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
            if ((a.out_width & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out_rm) & 15) == 0) {       // 16-byte stores (as the split path's epilogue): a quarter of the store instructions
#pragma unroll
                for (int G = 0; G < NT * 4; ++G) {
                    const int f = 8 * G + 4 * h;
                    if (f < a.out_width)
                        st4(o + f, make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
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
__device__ __forceinline__ void linear_body(const LinArgs& a, const int tile, const int lane) {
    const int h = lane >> 5, j = lane & 31;
    const int ptile = tile % a.tiles_per_pass;
    const int pass = tile / a.tiles_per_pass;
    const int row = ptile * 32 + j;
    const int KG = a.in_groups;
    const size_t nt_stride = (size_t)KG * 256;

    f32x16 acc[NT];
    acc_init<NT>(acc, a.bias, h);

    float mean = 0.f, rstd = 1.f;
    if (LNACT) {
        const float2 s = reinterpret_cast<const float2*>(a.in.stats)[(size_t)tile * 32 + j];
        mean = s.x;
        rstd = rsqrtf(s.y / (float)a.in.width + kLnEps);
    }
    if (INMODE == IN_FRAG) {
        chain_from_mem<NT, LNACT>(acc, a.in.data + (size_t)tile * KG * 256 + lane * 4, KG, a.W + lane * 4, nt_stride,
                                  LNACT ? a.gamma + 4 * h : nullptr, LNACT ? a.beta + 4 * h : nullptr, mean, rstd);
    } else {
        // rows of a multiple of 4 floats (MSR-80c: 80): one 16-byte load per lane and group instead of four 4-byte ones (a wave's
        // load touches 32 rows = 32 cache lines either way; round 5: feature_proj 53 us at 65 536 rows on the exact path)
        const bool vec4 = (a.in_width & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in_rm) & 15) == 0;
        const bool rowok = row < a.nrows;
        const float* rp = a.in_rm + (size_t)(rowok ? row : 0) * a.in_width;
        // (weights one group ahead in ping-pong sets, as chain_from_mem, measured SLOWER here: feature_proj 40.4 -> 45.4 us at 65 536 rows)
        float4 xn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec4 && rowok && 4 * h < a.in_width) xn = ld4(rp + 4 * h);
        for (int g = 0; g < KG; ++g) {
            float4 wc[NT];
            load_wfrag<NT>(wc, a.W + (size_t)g * 256 + lane * 4, nt_stride);
            float v[4];
            if (vec4) {
                v[0] = xn.x; v[1] = xn.y; v[2] = xn.z; v[3] = xn.w;
                xn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowok && 8 * (g + 1) + 4 * h < a.in_width && g + 1 < KG) xn = ld4(rp + 8 * (g + 1) + 4 * h);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int f = 8 * g + 4 * h + p;
                    v[p] = (rowok && f < a.in_width) ? rp[f] : 0.f;
                }
            }
            mfma_group<NT>(acc, wc, v[0], v[1], v[2], v[3]);
        }
    }

    linear_store<NT, OUTMODE>(a, tile, lane, acc);
}

// (Without a waves-per-SIMD bound hipcc gave these kernels 174-302 registers -- up to 128 of them accumulator registers for 64
// accumulator values -- and one or two waves per SIMD: a handful of MFMAs per memory round trip then runs at 1 TB/s.  Round 5.)
template <int NT, int INMODE, int OUTMODE, bool LNACT>
__global__ __launch_bounds__(256, NT >= 4 ? 3 : 4) void k_linear(const LinArgs a) {
    if (INMODE == IN_ROWMAJOR && a.advance_step && blockIdx.x == 0 && threadIdx.x == 0) *a.advance_step -= 1;
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= a.ntiles) return;
    linear_body<NT, INMODE, OUTMODE, LNACT>(a, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Exact path, large launches: a >= 64-wide block and the Linear that consumes it (Down / Upsample, or `final`: LayerNorm + SiLU +
// Linear) in one launch -- the Linear reads the block's accumulators instead of a stored tensor (round 5; the split path has had
// the pairs since round 2).  Same operands, same MFMA order, same statistics as the two launches: same bits
// (test_exact_path_pair_kernels_are_the_two_launches_bit_for_bit).
// ---------------------------------------------------------------------------------------------
template <int N, bool SCLIN, int NTO, bool FINAL>
__global__ __launch_bounds__(256, N >= 128 ? 2 : 4) void k_resblock_lin(const BlockArgs a, const LinArgs l, const int store_block_out) {
    constexpr int NT = N / 32, NG = N / 8;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= a.ntiles) return;
    f32x16 x[NT];
    float xmean, xm2;
    resblock_body<N, SCLIN, false, true, true>(a, tile, lane, x, &xmean, &xm2, store_block_out != 0);
    f32x16 acc[NTO];
    acc_init<NTO>(acc, l.bias, h);
    const float rstd = FINAL ? rsqrtf(xm2 / (float)N + kLnEps) : 1.f;      // as linear_body computes it from the stored statistics
    const size_t nt_stride = (size_t)NG * 256;
    const float* wp = l.W + lane * 4;
    // weights one group ahead in two slots -- except at N = 64 with >= 3 out tiles, where the second slot does not fit the 128
    // registers of four waves per SIMD (164 bytes of scratch): there the group's own loads, covered by the SIMD's other waves
    constexpr int PF = (N == 64 && NTO >= 3) ? 0 : 1;
    float4 wb[PF + 1][NTO], gb[PF + 1], bb[PF + 1];
    if (PF) {
        load_wfrag<NTO>(wb[0], wp, nt_stride);
        if (FINAL) { gb[0] = ld4(l.gamma + 4 * h); bb[0] = ld4(l.beta + 4 * h); }
    }
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        const int GL = G + PF, sl = PF ? (GL & 1) : 0, sc = PF ? (G & 1) : 0;
        if (GL < NG) {
            load_wfrag<NTO>(wb[sl], wp + (size_t)GL * 256, nt_stride);
            if (FINAL) { gb[sl] = ld4(l.gamma + 8 * GL + 4 * h); bb[sl] = ld4(l.beta + 8 * GL + 4 * h); }
        }
        __builtin_amdgcn_sched_barrier(0);
        float4 b = make_float4(x[G >> 2][4 * (G & 3) + 0], x[G >> 2][4 * (G & 3) + 1], x[G >> 2][4 * (G & 3) + 2], x[G >> 2][4 * (G & 3) + 3]);
        if (FINAL) b = ln_silu4(b, xmean, rstd, gb[sc], bb[sc]);
        mfma_group<NTO>(acc, wb[sc], b.x, b.y, b.z, b.w);
        if (!PF) __builtin_amdgcn_sched_barrier(0);
    }
    linear_store<NTO, FINAL ? OUT_ROWMAJOR : OUT_FRAG>(l, tile, lane, acc);
}

// ---------------------------------------------------------------------------------------------
// The narrow middle of the U-Net (the longest run of modules whose outputs are <= 32 wide) as ONE
// launch: a wave walks its tile through the whole run of operators.  Tensors still round-trip through their
// fragment buffers (the wave re-reads what it wrote itself: tile-local, L2-hot), so this removes ~25 dependent
// kernel boundaries and their fill/drain per reverse step, not the arithmetic.  Inference only.
// ---------------------------------------------------------------------------------------------
struct FusedOp {
    int kind;     // 0: ResidualBlock, 1: Linear
    int N;        // block width / Linear out width
    int sclin;
    int pad;
    BlockArgs b;
    LinArgs l;
};

// Linear with register input (in width <= 32) and register output (out width <= 32): the exact path's in-run Down/Upsample
__device__ __forceinline__ void linear_reg(const LinArgs& a, const int tile, const int lane, f32x16& x, float& xmean, float& xm2, const bool store_out) {
    const int h = lane >> 5, j = lane & 31;
    f32x16 acc[1];
    acc_init<1>(acc, a.bias, h);
    chain_from_reg<1, false>(acc, x, a.in_groups, a.W + lane * 4, (size_t)a.in_groups * 256, nullptr, nullptr, 0.f, 1.f);
    const int NG = (a.out_width + 7) / 8;
    float s = 0.f;
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (8 * G + 4 * h + p < a.out_width) s += acc[0][4 * G + p];
    const float m = xhalf_sum(s) / (float)a.out_width;
    float q = 0.f;
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (8 * G + 4 * h + p < a.out_width) { const float d = acc[0][4 * G + p] - m; q = fmaf(d, d, q); }
    q = xhalf_sum(q);
    x = acc[0]; xmean = m; xm2 = q;
    if (store_out) {
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(m, q);
#pragma unroll
        for (int G = 0; G < 4; ++G)
            if (G < NG) st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4, make_float4(acc[0][4 * G], acc[0][4 * G + 1], acc[0][4 * G + 2], acc[0][4 * G + 3]));
    }
}

// `pad` of an entry: bit 1 = the entry is a link of a CHAIN (the narrow run: its input is the previous entry's output, handed over in
// registers; round 5 -- the round-1 form stored every tensor, drained its stores and re-read them: ~27 exposed round trips per tile),
// bit 0 = store the output anyway (a skip tensor, the run's last tensor).  pad = 0: an independent operator, memory in, memory out
// (the condition-embedding Linears of run_cond_embed share this kernel).
__global__ __launch_bounds__(256, 4) void k_fused_narrow(const FusedOp* __restrict__ ops, int nops, int ntiles) {
    // (36 bytes of scratch per lane: 8 spilled registers under the 128-register bound.  The opaque-lane-per-operator construction that
    // freed k_fused_narrow_lds of its spills makes it worse here -- 79 scratch instructions against 40, round 5.)
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= ntiles) return;
    f32x16 x;
    float xmean = 0.f, xm2 = 0.f;
    bool have_x = false;
#pragma unroll 1
    for (int i = 0; i < nops; ++i) {
        const FusedOp& op = ops[i];
        const bool chain = (op.pad & 2) != 0, st = (op.pad & 1) != 0;
        if (op.kind == 0) {
            BlockArgs b = op.b;               // every pointer of the record is global (as_global)
            globalize_io(b); globalize_params(b);
            if (chain) {
                if (!have_x) {                // first link: bring its (<= 32 wide) input into registers once
                    const float2 s0 = reinterpret_cast<const float2*>(b.in0.stats)[(size_t)tile * 32 + j];
                    xmean = s0.x; xm2 = s0.y;
#pragma unroll
                    for (int G = 0; G < 4; ++G) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (G < b.in0.groups) v = ld4(b.in0.data + ((size_t)tile * b.in0.groups + G) * 256 + lane * 4);
                        x[4 * G] = v.x; x[4 * G + 1] = v.y; x[4 * G + 2] = v.z; x[4 * G + 3] = v.w;
                    }
                    have_x = true;
                }
                // skip tensors were stored by this wave earlier in the run: make sure those stores have landed
                if (b.in1.groups) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (op.sclin) {
                    switch (op.N) {
                        case 4: resblock_body<4, true, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        case 8: resblock_body<8, true, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        case 16: resblock_body<16, true, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        default: resblock_body<32, true, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                    }
                } else {
                    switch (op.N) {
                        case 4: resblock_body<4, false, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        case 8: resblock_body<8, false, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        case 16: resblock_body<16, false, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                        default: resblock_body<32, false, true, true>(b, tile, lane, &x, &xmean, &xm2, st); break;
                    }
                }
                continue;
            }
            if (op.sclin) {
                switch (op.N) {
                    case 4: resblock_body<4, true>(b, tile, lane); break;
                    case 8: resblock_body<8, true>(b, tile, lane); break;
                    case 16: resblock_body<16, true>(b, tile, lane); break;
                    default: resblock_body<32, true>(b, tile, lane); break;
                }
            } else {
                switch (op.N) {
                    case 4: resblock_body<4, false>(b, tile, lane); break;
                    case 8: resblock_body<8, false>(b, tile, lane); break;
                    case 16: resblock_body<16, false>(b, tile, lane); break;
                    default: resblock_body<32, false>(b, tile, lane); break;
                }
            }
        } else {
            LinArgs l = op.l;
            globalize_io(l); globalize_params(l);
            if (chain && have_x && l.in_groups <= 4) { linear_reg(l, tile, lane, x, xmean, xm2, st); continue; }
            linear_body<1, IN_FRAG, OUT_FRAG, false>(l, tile, lane);
            if (chain) {
                // the Linear that enters the run (its input is wider than one tile): memory in, memory out, then reload
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int NG = (l.out_width + 7) / 8;
                const float2 s0 = reinterpret_cast<const float2*>(l.out_stats)[(size_t)tile * 32 + j];
                xmean = s0.x; xm2 = s0.y;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G < NG) v = ld4(l.out + ((size_t)tile * NG + G) * 256 + lane * 4);
                    x[4 * G] = v.x; x[4 * G + 1] = v.y; x[4 * G + 2] = v.z; x[4 * G + 3] = v.w;
                }
                have_x = true;
                continue;
            }
        }
        // the next operator of THIS wave may read what it just stored (same tile): drain the stores first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------
// Weight packing: nn.Linear.weight [N][Ktot] row-major -> MFMA A-fragment order
//   kind 0  packed[((nt*KG + g)*64 + lane)*4 + p] = W[32nt + (lane&31)][col(g) + 4(lane>>5) + p]
//   kind 1  the transposed operator (data gradient dX = W^T g):
//           packed[((ot*NGin + g)*64 + lane)*4 + p] = W[8g + 4(lane>>5) + p][col(32ot + (lane&31))]
//   kind 2  per-feature vectors (biases, LayerNorm gamma/beta, merged biases a+b) padded in group order
// The K axis may be the concatenation of two tensors (widths w0, w1) that are padded to a multiple of 8
// separately, matching the two input segments of k_resblock.
// ---------------------------------------------------------------------------------------------
// One launch packs everything (dsg_bind_weights runs after every optimizer step): block -> descriptor by binary search.
struct PackDesc {
    const float* a;       // weight [N][Ktot], or vector
    const float* b;       // optional second vector (added)
    float* dst;
    int kind;             // 0: A-operand pack, 1: transposed pack (data gradient), 2: padded vector
    int N, Ktot, w0, w1, T;   // T = number of 32-wide output tiles (kind 0/1) ; npad (kind 2)
    long long total;      // elements of dst
    long long blk_begin;  // first block of this descriptor
};

__device__ __forceinline__ float pack_elem(const PackDesc& d, long long idx) {
    if (d.kind == 2) {
        const int g0 = (d.w0 + 7) / 8;
        const int g = (int)(idx >> 3), e = (int)(idx & 7);
        int k, ok;
        if (g < g0) { k = 8 * g + e; ok = k < d.w0; }
        else { const int kk = 8 * (g - g0) + e; ok = kk < d.w1; k = d.w0 + kk; }
        float v = 0.f;
        if (ok) { v = d.a[k]; if (d.b) v += d.b[k]; }
        return v;
    }
    const int p = idx & 3, lane = (idx >> 2) & 63;
    const long long tg = idx >> 8;
    const int g0 = (d.w0 + 7) / 8;
    if (d.kind == 0) {
        const int KG = g0 + (d.w1 + 7) / 8;
        const int g = tg % KG, nt = tg / KG;
        const int n = 32 * nt + (lane & 31);
        int k, ok;
        if (g < g0) { k = 8 * g + 4 * (lane >> 5) + p; ok = k < d.w0; }
        else { const int kk = 8 * (g - g0) + 4 * (lane >> 5) + p; ok = kk < d.w1; k = d.w0 + kk; }
        return (ok && n < d.N && k < d.Ktot) ? d.a[(size_t)n * d.Ktot + k] : 0.f;
    }
    const int NGin = (d.N + 7) / 8;
    const int g = tg % NGin, ot = tg / NGin;
    const int n = 8 * g + 4 * (lane >> 5) + p;
    const int o = 32 * ot + (lane & 31);
    const int og = o >> 3, e = o & 7;
    int col, ok;
    if (og < g0) { col = 8 * og + e; ok = col < d.w0; }
    else { const int c = 8 * (og - g0) + e; ok = c < d.w1; col = d.w0 + c; }
    return (ok && n < d.N && col < d.Ktot) ? d.a[(size_t)n * d.Ktot + col] : 0.f;
}

constexpr int kPackPerBlock = 256 * 8;

// `zero` (nzero floats): the max|W| words of the split path, cleared here for the k_maxabs launch that follows on the same stream (a
// hipMemsetAsync of its own was a 5 us launch in every training step's re-pack)
__global__ __launch_bounds__(256) void k_pack_grouped(const PackDesc* __restrict__ descs, int ndesc, float* __restrict__ zero, int nzero) {
    {
        const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
        if (gi < nzero) zero[gi] = 0.f;
    }
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk_begin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackDesc d = descs[lo];
    const long long base = ((long long)blockIdx.x - d.blk_begin) * kPackPerBlock;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long long idx = base + it * 256 + threadIdx.x;
        if (idx < d.total) d.dst[idx] = pack_elem(d, idx);
    }
}

// ---------------------------------------------------------------------------------------------
// Time path, hoisted out of the per-row work (it depends only on the step / ts value):
//   TimeEmbedding.forward (UNetCF.py:35-44)  -> st[e] = Swish(temb(t_e))   (Swish from UNetCF.py:91)
//   per block b:  tb[e][off_b + n] = lin1.bias[n] + time_emb.bias[n] + time_emb.weight[n] . st[e]
// so that stage 1 of every ResidualBlock starts its accumulator from tb (UNetCF.py:90-91).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Stage 1: emb = [sin, cos](t * freq), h1 = Swish(lin1 emb).  grid = (entries, ceil(td / 16)); one wave per output.
__global__ __launch_bounds__(256) void k_time_embed1(const float* __restrict__ tvals, const float* __restrict__ freq, int half,
                                                     const float* __restrict__ W1, const float* __restrict__ b1, int td,
                                                     float* __restrict__ h1s, float* __restrict__ save_emb, float* __restrict__ save_h1pre) {
    extern __shared__ float sm[];  // [2*half]
    const int ent = blockIdx.x;
    const float t = tvals[ent];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float ang = t * freq[i];
        sm[i] = sinf(ang);
        sm[half + i] = cosf(ang);
        if (save_emb && blockIdx.y == 0) { save_emb[(size_t)ent * 2 * half + i] = sm[i]; save_emb[(size_t)ent * 2 * half + half + i] = sm[half + i]; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K1 = 2 * half;
    for (int n = blockIdx.y * 16 + wave; n < td && n < (int)(blockIdx.y + 1) * 16; n += 4) {
        float s = 0.f;
        for (int k = lane; k < K1; k += 64) s = fmaf(W1[(size_t)n * K1 + k], sm[k], s);
        s = wave_sum(s) + b1[n];
        if (lane == 0) {
            h1s[(size_t)ent * td + n] = silu(s);
            if (save_h1pre) save_h1pre[(size_t)ent * td + n] = s;
        }
    }
}

// Stage 2: st = Swish(lin2 h1).  Same grid.
__global__ __launch_bounds__(256) void k_time_embed2(const float* __restrict__ h1s, const float* __restrict__ W2, const float* __restrict__ b2,
                                                     int td, float* __restrict__ st, float* __restrict__ save_tpre) {
    extern __shared__ float sm[];  // [td]
    const int ent = blockIdx.x;
    for (int i = threadIdx.x; i < td; i += blockDim.x) sm[i] = h1s[(size_t)ent * td + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = blockIdx.y * 16 + wave; n < td && n < (int)(blockIdx.y + 1) * 16; n += 4) {
        float s = 0.f;
        for (int k = lane; k < td; k += 64) s = fmaf(W2[(size_t)n * td + k], sm[k], s);
        s = wave_sum(s) + b2[n];
        if (lane == 0) {
            st[(size_t)ent * td + n] = silu(s);
            if (save_tpre) save_tpre[(size_t)ent * td + n] = s;
        }
    }
}

struct TimeBlockDesc {
    const float* Wt;   // [N][td]
    const float* bt;   // [N]
    const float* b1;   // lin1.bias [N]
    int N;
    int off;           // offset of this block's slice inside a table row (multiple of 32)
};

__global__ __launch_bounds__(256) void k_time_table(const float* __restrict__ st, int td, const TimeBlockDesc* __restrict__ blocks,
                                                    int nblocks, float* __restrict__ tb, int tb_stride) {
    extern __shared__ float sm[];
    const int ent = blockIdx.x;
    for (int i = threadIdx.x; i < td; i += blockDim.x) sm[i] = st[(size_t)ent * td + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // gridDim.y == 8 * nblocks: an eighth of one block's rows per workgroup (the table is tiny: the kernel is bound by the
    // length of one wave's chain of dependent dot products, so it is spread thin)
    {
        const int b = blockIdx.y >> 3, part = blockIdx.y & 7;
        if (b < nblocks) {
            const TimeBlockDesc d = blocks[b];
            const int npad = (d.N + 31) / 32 * 32, per = npad / 8;
            for (int n = part * per + wave; n < (part + 1) * per; n += nw) {
                float v = 0.f;
                if (n < d.N) {
                    float s0 = 0.f, s1 = 0.f;
                    int k = lane;
                    for (; k + 64 < td; k += 128) {
                        s0 = fmaf(d.Wt[(size_t)n * td + k], sm[k], s0);
                        s1 = fmaf(d.Wt[(size_t)n * td + k + 64], sm[k + 64], s1);
                    }
                    for (; k < td; k += 64) s0 = fmaf(d.Wt[(size_t)n * td + k], sm[k], s0);
                    v = wave_sum(s0 + s1) + d.bt[n] + d.b1[n];
                }
                if (lane == 0) tb[(size_t)ent * tb_stride + d.off + n] = v;
            }
        }
    }
}

// condfrag[tile][g][lane][p] = silu(cond[row][8g+4h+p] * mask[row])   (UNetCF.py:330 and :93)
__global__ void k_cond_frag(const float* __restrict__ cond, const float* __restrict__ mask, int nrows, int C, int CG,
                            float* __restrict__ out, int ntiles) {
    const size_t total = (size_t)ntiles * CG * 256;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int p = idx & 3, lane = (idx >> 2) & 63;
        const size_t tg = idx >> 8;
        const int g = tg % CG;
        const int tile = tg / CG;
        const int row = tile * 32 + (lane & 31), f = 8 * g + 4 * (lane >> 5) + p;
        float v = 0.f;
        if (row < nrows && f < C) {
            v = cond[(size_t)row * C + f];
            if (mask) v *= mask[row];
            v = silu(v);
        }
        out[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Reverse-step update (DDPM.sample, classifier_free_MSR.py:129-134), op order kept:
//   eps = (1+omega)*eps1 - omega*eps0 ;  y <- (y - c1*eps)*c2 + c3*z
// z comes from a caller tensor (parity mode) or from Philox4x32-10 + Box-Muller (throughput mode).
// coef[step] = {c1 = betas/sqrt(1-acp), c2 = 1/sqrt(alpha), c3 = (1-acp[i-1])/(1-acp[i]), has_noise}.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void normal4(uint64_t seed, uint32_t stream, uint64_t idx4, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)idx4, (uint32_t)(idx4 >> 32), stream, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u0 = ((r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float s, c;
    sincosf(6.28318530717958647692f * u1, &s, &c);
    z[0] = ra * c; z[1] = ra * s;
    sincosf(6.28318530717958647692f * u3, &s, &c);
    z[2] = rb * c; z[3] = rb * s;
}

// Per-call parameters live in device memory so that the captured per-step graph does not depend on them.
struct CallParams {
    const float* z;       // [T-2][n] noise of steps T-1 .. 2, or null (Philox)
    const float* coef;    // [T][4]
    float omega;
    int T;
    unsigned long long seed;
    float* rec_y;         // record_denoise_path (MSR.py:139-141): [T][n] y after each step (after the renorm), or null
    float* rec_eps;       // [T][n] guided eps of each step, or null
    // chunked call (dsg_sample_chunked): the batch is a sequence of independent sample() calls of `chunk_n4` quads each (the last
    // one may be shorter): chunk c draws from Philox stream seeds[c] with chunk-local element indices, exactly as its own call would
    size_t chunk_n4;                      // 0: one call
    const unsigned long long* seeds;      // device [chunks]
};
__device__ __forceinline__ void chunk_of(const CallParams& cp, size_t i4, unsigned long long& seed, size_t& local) {
    seed = cp.seed; local = i4;
    if (cp.chunk_n4) { const size_t c = i4 / cp.chunk_n4; seed = cp.seeds[c]; local = i4 - c * cp.chunk_n4; }
}

__global__ void k_set_call(int* step, CallParams* dst, const CallParams cp, int start) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { *step = start; *dst = cp; }
}

struct UpdateArgs {
    const float* eps;     // [2][n] : eps0 then eps1
    float* y;             // [n] in/out
    const CallParams* cp;
    const int* step_ptr;
    size_t n;
    int record_y;             // steps without the renorm: this launch also records y (otherwise k_record after the renorm)
};

__global__ __launch_bounds__(256) void k_update(const UpdateArgs a) {
    const int step = *a.step_ptr;
    const CallParams cp = *a.cp;
    const float c1 = cp.coef[4 * step], c2 = cp.coef[4 * step + 1], c3 = cp.coef[4 * step + 2];
    const bool noisy = cp.coef[4 * step + 3] != 0.f;
    const float omega = cp.omega, w1 = 1.0f + cp.omega;
    const size_t n4 = (a.n + 3) / 4;
    const float* zrow = (cp.z && noisy) ? cp.z + (size_t)(cp.T - 1 - step) * a.n : nullptr;
    float* rec = cp.rec_eps ? cp.rec_eps + (size_t)(cp.T - 1 - step) * a.n : nullptr;
    float* recy = (a.record_y && cp.rec_y) ? cp.rec_y + (size_t)(cp.T - 1 - step) * a.n : nullptr;
    // whole quads, 16-byte aligned everywhere: one dwordx4 per stream and thread (same arithmetic per element)
    const bool quads = (a.n & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(a.eps) | reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(zrow) |
                         reinterpret_cast<uintptr_t>(rec) | reinterpret_cast<uintptr_t>(recy)) & 15) == 0;
    for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * blockDim.x) {
        float zz[4] = {0.f, 0.f, 0.f, 0.f};
        if (noisy && !zrow) { unsigned long long sd; size_t li; chunk_of(cp, i4, sd, li); normal4(sd, (uint32_t)step, li, zz); }
        if (quads) {
            const float4 e0 = ld4(a.eps + i4 * 4), e1 = ld4(a.eps + a.n + i4 * 4), yv = ld4(a.y + i4 * 4);
            if (zrow) { const float4 zv = ld4(zrow + i4 * 4); zz[0] = zv.x; zz[1] = zv.y; zz[2] = zv.z; zz[3] = zv.w; }
            const float e0a[4] = {e0.x, e0.y, e0.z, e0.w}, e1a[4] = {e1.x, e1.y, e1.z, e1.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
            float ea[4], out[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ea[p] = __fsub_rn(__fmul_rn(w1, e1a[p]), __fmul_rn(omega, e0a[p]));
                const float v = __fmul_rn(__fsub_rn(ya[p], __fmul_rn(c1, ea[p])), c2);
                out[p] = noisy ? __fadd_rn(v, __fmul_rn(c3, zz[p])) : v;
            }
            if (rec) st4(rec + i4 * 4, make_float4(ea[0], ea[1], ea[2], ea[3]));
            st4(a.y + i4 * 4, make_float4(out[0], out[1], out[2], out[3]));
            if (recy) st4(recy + i4 * 4, make_float4(out[0], out[1], out[2], out[3]));
            continue;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const size_t i = i4 * 4 + p;
            if (i < a.n) {
                if (zrow) zz[p] = zrow[i];
                const float e = __fsub_rn(__fmul_rn(w1, a.eps[a.n + i]), __fmul_rn(omega, a.eps[i]));
                if (rec) rec[i] = e;
                const float v = __fmul_rn(__fsub_rn(a.y[i], __fmul_rn(c1, e)), c2);
                const float o = noisy ? __fadd_rn(v, __fmul_rn(c3, zz[p])) : v;
                a.y[i] = o;
                if (recy) recy[i] = o;
            }
        }
    }
}

// y_T ~ N(0, 1) on device (throughput mode; the reference draws it on the host, MSR.py:115)
__global__ void k_randn(float* __restrict__ y, size_t n, unsigned long long seed, unsigned stream) {
    const size_t n4 = (n + 3) / 4;
    for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * blockDim.x) {
        float zz[4];
        normal4(seed, stream, i4, zz);
        for (int p = 0; p < 4; ++p)
            if (i4 * 4 + p < n) y[i4 * 4 + p] = zz[p];
    }
}

// start state of a chunked call: chunk c = k_randn(seed[c]) on its own elements
__global__ void k_randn_chunked(float* __restrict__ y, size_t n, const unsigned long long* __restrict__ seeds, size_t chunk_n4, unsigned stream) {
    const size_t n4 = (n + 3) / 4;
    for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i4 / chunk_n4;
        float zz[4];
        normal4(seeds[c], stream, i4 - c * chunk_n4, zz);
        for (int p = 0; p < 4; ++p)
            if (i4 * 4 + p < n) y[i4 * 4 + p] = zz[p];
    }
}

// device-side trajectory ring (replaces the reference's per-step .cpu().numpy(), MSR.py:139-141); no-op when disabled
__global__ void k_record(const float* __restrict__ y, size_t n, const CallParams* __restrict__ cp, const int* __restrict__ step_ptr) {
    float* dst = cp->rec_y;
    if (!dst) return;
    dst += (size_t)(cp->T - 1 - *step_ptr) * n;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = y[i];
}


// ---------------------------------------------------------------------------------------------
// Early-step renormalisation (MSR.py:136-137): y <- (y - mean(y)) / sqrt(var(y)), unbiased var over ALL B*D
// elements.  Deterministic two-pass reduction with float64 partials (fixed grid, fixed order).
// ---------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 512;

__device__ __forceinline__ double block_sum(double v, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    __syncthreads();
    return t;
}

// Chunked form (gridDim.y = chunks, dsg_sample_chunked): block (x, c) does for chunk c -- elements [c * chunk_n, min(n, (c + 1) * chunk_n))
// -- exactly what block x of a one-call launch does for the whole tensor: same strides, same order of additions, so a chunk's
// statistics are bit-identical to those of its own sample() call.
__device__ __forceinline__ void renorm_chunk(const float*& y, size_t& n, size_t chunk_n) {
    if (chunk_n) { const size_t lo = (size_t)blockIdx.y * chunk_n; y += lo; n = n - lo < chunk_n ? n - lo : chunk_n; }
}
// every thread visits whole 16-byte quads (the tensor is 21 MB at the bench size: scalar loads ran these passes at 1.7 TB/s), then
// the < 4 elements of a ragged tail; the order is fixed, so the reductions stay deterministic
template <typename F>
__device__ __forceinline__ void renorm_visit(const float* y, size_t n, F&& f) {
    const size_t n4 = (reinterpret_cast<uintptr_t>(y) & 15) ? 0 : n / 4;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (size_t i4 = t; i4 < n4; i4 += st) { const float4 v = ld4(y + 4 * i4); f(v.x); f(v.y); f(v.z); f(v.w); }
    for (size_t i = 4 * n4 + t; i < n; i += st) f(y[i]);
}
// Round 4: ONE pass for both moments (sum y and sum y^2 in float64, fixed block partials: deterministic); the variance is formed in
// float64 as (sum y^2 - n mean^2) / (n - 1) by k_renorm_apply.  In float64 that loses ~(1 + mean^2 / var) x 1e-16 relative, far below the
// float32 result it is rounded to; it replaces the two-pass form (k_renorm_sum, then k_renorm_sqdiff: one more launch and one more read
// of y on each of the four early steps of every call).
__global__ __launch_bounds__(256) void k_renorm_sum(const float* y, size_t n, double* __restrict__ part, double* __restrict__ part2, size_t chunk_n = 0) {
    __shared__ double sm[4];
    renorm_chunk(y, n, chunk_n); part += (size_t)blockIdx.y * kRedBlocks; part2 += (size_t)blockIdx.y * kRedBlocks;
    double s = 0.0, q = 0.0;
    renorm_visit(y, n, [&](float v) { const double d = (double)v; s += d; q += d * d; });
    s = block_sum(s, sm);
    q = block_sum(q, sm);
    if (threadIdx.x == 0) { part[blockIdx.x] = s; part2[blockIdx.x] = q; }
}

// Sharded form of the early-step renorm (dsg_set_renorm_hook): the shard's (sum y, sum y^2, count) in float64 -- the caller
// sums the three over the ranks -- then the same standardisation from the reduced moments.
__global__ __launch_bounds__(256) void k_renorm_moments(const float* __restrict__ y, size_t n, double* __restrict__ part, double* __restrict__ part2) {
    __shared__ double sm[4];
    double s = 0.0, q = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)y[i];
        s += v; q += v * v;
    }
    s = block_sum(s, sm);
    __syncthreads();
    q = block_sum(q, sm);
    if (threadIdx.x == 0) { part[blockIdx.x] = s; part2[blockIdx.x] = q; }
}
__global__ void k_renorm_moments_final(const double* __restrict__ part, const double* __restrict__ part2, size_t n, double* __restrict__ stats3) {
    double tot = 0.0, tot2 = 0.0;
    for (int i = 0; i < kRedBlocks; ++i) { tot += part[i]; tot2 += part2[i]; }
    stats3[0] = tot; stats3[1] = tot2; stats3[2] = (double)n;
}
__global__ __launch_bounds__(256) void k_renorm_apply_stats(float* __restrict__ y, size_t n, const double* __restrict__ stats3) {
    const double N = stats3[2], m = stats3[0] / N;
    const float mean = (float)m;
    const double var = (stats3[1] - N * m * m) / (N - 1.0);     // one-pass form: a (near-)constant y can round a hair below zero -> clamp
    const float sd = sqrtf((float)(var > 0.0 ? var : 0.0));         // (the two-pass form of MSR.py:136-137 gives exactly 0 there)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (y[i] - mean) / sd;
}
// ... and, fused, the trajectory record of the renormalised y (k_record's job on these steps): `cp` / `step_ptr` null = no record
__global__ __launch_bounds__(256) void k_renorm_apply(float* y, size_t n, const double* __restrict__ part,
                                                      const double* __restrict__ part2, size_t chunk_n = 0,
                                                      const CallParams* __restrict__ cp = nullptr, const int* __restrict__ step_ptr = nullptr) {
    float* rec = nullptr;
    if (cp && cp->rec_y) rec = cp->rec_y + (size_t)(cp->T - 1 - *step_ptr) * n + (chunk_n ? (size_t)blockIdx.y * chunk_n : 0);
    { const float* yc = y; renorm_chunk(yc, n, chunk_n); y = const_cast<float*>(yc); }
    part += (size_t)blockIdx.y * kRedBlocks; part2 += (size_t)blockIdx.y * kRedBlocks;
    double tot = 0.0, tot2 = 0.0;
    for (int i = 0; i < kRedBlocks; ++i) { tot += part[i]; tot2 += part2[i]; }
    const double mean_d = tot / (double)n;
    const float mean = (float)mean_d;
    const double var = (tot2 - tot * mean_d) / (double)(n - 1);                     // unbiased, as torch.var (MSR.py:137)
    const float sd = sqrtf((float)(var > 0.0 ? var : 0.0));                        // clamp: see k_renorm_apply_stats
    const size_t n4 = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(rec)) & 15) ? 0 : n / 4;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (size_t i4 = t; i4 < n4; i4 += st) {
        const float4 v = ld4(y + 4 * i4);
        const float4 o = make_float4((v.x - mean) / sd, (v.y - mean) / sd, (v.z - mean) / sd, (v.w - mean) / sd);
        st4(y + 4 * i4, o);
        if (rec) st4(rec + 4 * i4, o);
    }
    for (size_t i = 4 * n4 + t; i < n; i += st) {
        const float o = (y[i] - mean) / sd;
        y[i] = o;
        if (rec) rec[i] = o;
    }
}

// EMA (ema.py:11-12): avg = decay*avg + (1-decay)*p over one flat range
// (`om` = 1 - decay evaluated in double on the host, then rounded, as torch does with the Python scalar)
__global__ void k_ema(float* __restrict__ avg, const float* __restrict__ p, float decay, float om, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        avg[i] = __fadd_rn(__fmul_rn(decay, avg[i]), __fmul_rn(om, p[i]));
}

// Adam (classifier_free_MSR.py:209, torch.optim.Adam defaults; no amsgrad) over ONE flat range, element for element the arithmetic of
// torch's fused kernel (ATen/native/cuda/fused_adam_utils.cuh, adam_math<float, float, 4, ORIGINAL, false>): the hyper-parameters are
// doubles, so the two moment updates and the `+ eps` are evaluated in double and rounded once; the bias corrections are formed in
// double from pow(beta, step) and handed on as floats; step size, square root, quotient and the parameter update are float operations.
// torch's kernel gives a block 65 536 elements of a tensor: the 1.6 M-element flat parameter vector runs on 26 of 256 CUs (45 us); this
// one is a plain grid-stride loop over 16-byte pieces (memory-bound: 4 reads + 3 writes per element).
// dsg_range_status_stream: take the flag (read and clear in ONE atomic step) and hand it to the host through a pinned word
__global__ void k_flag_take(int* __restrict__ flag, int* __restrict__ host_word) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int v = atomicExch(flag, 0);
        __atomic_store_n(host_word, v, __ATOMIC_RELAXED);
        __threadfence_system();
    }
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    size_t n;
    double lr, beta1, beta2, weight_decay, eps;
    float step;                 // the step count AFTER this update (1, 2, ...), as torch's float step tensor holds it
    int maximize;
    // dsg_adam_step_dyn (the update inside a captured graph): learning rate and step count from device memory (null: the values above)
    const double* lr_ptr; const float* step_ptr;
};
__device__ __forceinline__ void adam_one(float& param, float grad, float& exp_avg, float& exp_avg_sq, const AdamArgs& a, float bias_correction1,
                                         float bias_correction2_sqrt) {
    if (a.maximize) grad = -grad;
    if (a.weight_decay != 0) grad += param * a.weight_decay;
    exp_avg = a.beta1 * exp_avg + (1 - a.beta1) * grad;
    exp_avg_sq = a.beta2 * exp_avg_sq + (1 - a.beta2) * grad * grad;
    const float step_size = a.lr / bias_correction1;
    const float denom = (sqrtf(exp_avg_sq) / bias_correction2_sqrt) + a.eps;
    param -= step_size * exp_avg / denom;
}
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    if (a.lr_ptr) a.lr = *a.lr_ptr;
    if (a.step_ptr) a.step = *a.step_ptr;
    const double bc1 = 1 - pow(a.beta1, (double)a.step), bc2 = 1 - pow(a.beta2, (double)a.step);
    const float bias_correction1 = (float)bc1, bias_correction2_sqrt = (float)sqrt(bc2);
    const size_t n4 = a.n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 p = ld4(a.p + 4 * i), m = ld4(a.m + 4 * i), v = ld4(a.v + 4 * i);
        const float4 g = ld4(a.g + 4 * i);
        adam_one(p.x, g.x, m.x, v.x, a, bias_correction1, bias_correction2_sqrt);
        adam_one(p.y, g.y, m.y, v.y, a, bias_correction1, bias_correction2_sqrt);
        adam_one(p.z, g.z, m.z, v.z, a, bias_correction1, bias_correction2_sqrt);
        adam_one(p.w, g.w, m.w, v.w, a, bias_correction1, bias_correction2_sqrt);
        st4(a.p + 4 * i, p); st4(a.m + 4 * i, m); st4(a.v + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = 4 * n4 + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adam_one(p, a.g[i], m, v, a, bias_correction1, bias_correction2_sqrt);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Box calibration probes (dsg_box_calibrate, bench.py `box`): a dependent-MFMA loop on every SIMD and a float4 copy.
// ---------------------------------------------------------------------------------------------
// NV = vector instructions (every sixth a transcendental) behind each MFMA: 0 = the bare matrix pipe, 6 = the instruction mix of this
// library's block kernels (6.5 vector instructions per MFMA), whose clock under load is not the bare pipe's.
template <int NV>
__global__ __launch_bounds__(256) void k_calib_mfma(int iters, float* __restrict__ sink) {
    typedef _Float16 h8c __attribute__((ext_vector_type(8)));
    h8c a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {             // non-trivial operands: the clock a box holds depends on the data (all-zero operands run faster)
        a[i] = (_Float16)(0.37f + 0.013f * (float)((threadIdx.x * 7 + i * 3) & 63));
        b[i] = (_Float16)(-0.81f + 0.021f * (float)((threadIdx.x * 5 + i) & 31));
    }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48; ++r) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[r & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = (r * NV + i) % 12;
                if (i == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(0.999f), "v"(0.001f));
            }
        }
        // keep the sums finite over thousands of iterations without leaving the matrix pipe idle for long
        if ((it & 63) == 63) { c[0] *= 1e-6f; c[1] *= 1e-6f; c[2] *= 1e-6f; c[3] *= 1e-6f; }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMA's results are read below
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc += c[0][i] + c[1][i] + c[2][i] + c[3][i];
#pragma unroll
    for (int i = 0; i < 12; ++i) sacc += v[i];
    sink[blockIdx.x * 256 + threadIdx.x] = sacc;
}
__global__ __launch_bounds__(256) void k_calib_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace dsg
