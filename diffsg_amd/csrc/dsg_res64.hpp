// 64-wide ResidualBlocks for LARGE launches: the WHOLE block's weight planes live in LDS for the whole launch.
//
// Why (profiles/r02d_pmc_summary.txt): k_resblock_h<64,*> gives every wave its own copy of every plane from L2 -- 161
// vector-memory reads per wave, 48 % of wave cycles in s_waitcnt, 0.16 of the matrix-core peak; five such launches are a quarter
// of a reverse step.  A 64-wide block's planes are 96 KiB (128 -> 64 up block: W1 32, W2 16, W3 16, shortcut 32) or 48 KiB (down
// block): they fit the CU's 160 KiB of LDS next to nothing else.  So: one 8-wave workgroup per CU stages the planes and the
// per-feature vectors ONCE (plain loads + ds_write, then the launch's only barrier), then every wave walks its row tiles alone
// -- no weight stream, no ring, no barrier, no counted waits.  The wave's inputs are ordinary global loads into REGISTERS, one
// tile ahead: a 64-wide tile is 32 (down) or 64 (up) registers, the accumulator sets are 32 each, so the concat input is read ONCE
// and serves stage 1 (LayerNorm + SiLU) and the shortcut / residual from registers (the old kernel read it twice), and the next
// tile's input arrives under the current tile's shortcut stage.  Everything is compiler-visible code: hipcc counts every wait.
// Arithmetic per element, packed planes, scales and accumulation order are those of resblock_body_h<64, *>.
#pragma once
#include "dsg_panel.hpp"

namespace dsg {

constexpr int kR64Waves = 8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;

// LDS image (uint4): W1 [2][KS1][2][64] | W2 [2][4][2][64] | W3 | Wsc [2][KS1][2][64] (SCLIN) | vectors (floats)
template <bool SCLIN, int NTO = 0> struct R64Layout {
    static constexpr int KS1 = SCLIN ? 8 : 4;
    static constexpr int W1 = 0, W2 = W1 + 2 * KS1 * 128, W3 = W2 + 2 * 4 * 128, WSC = W3 + 2 * 4 * 128, WL = WSC + (SCLIN ? 2 * KS1 * 128 : 0),
                         VEC = WL + NTO * 4 * 128;     // WL: planes of the consuming Linear (K = 64: 4 steps), NTO out tiles
    // vectors: gamma1', beta1' (16 * KS1 each) | gamma2', beta2', gamma3', beta3' (64 each) | time bias, c2, c3 (64 each)
    static constexpr int G1 = 0, B1 = 16 * KS1, G2 = 2 * 16 * KS1, B2 = G2 + 64, G3 = B2 + 64, B3 = G3 + 64, TB = B3 + 64, C2 = TB + 64, C3 = C2 + 64,
                         BL = C3 + 64, NV = BL + 32 * NTO;
    static constexpr int TOTAL_U4 = VEC + (NV + 3) / 4;
};

// B operand of a step from eight values: LayerNorm (vectors from LDS, already times -log2 e) + SiLU + split, or the raw split
template <bool LNACT>
__device__ __forceinline__ void r64_prep(const float (&x)[8], const float* gv, const float* bv, int S, float c, float d, int h, h8& hi, h8& lo) {
    const BOp o = panel_prep<LNACT>(x, gv, bv, S, c, d, h);
    hi = o.hi; lo = o.lo;
}

// ---- software-pipelined stage: the operand of step S + 1 is prepared BETWEEN the MFMAs of step S.
// hipcc emits a stage written step by step (prep, then 4 plane reads, then 6 MFMAs) exactly in that order: the plane reads' LDS
// round trip and the 6 x 32 MFMA cycles are then exposed once per step in every wave (profiles/r03a: matrix core busy 18 %, VALU
// 34 %, the union 45 % of the launch).  Here the planes and LayerNorm vectors of step S + 1 are requested first (region A), and
// one scheduling region (B) holds the 6 MFMAs of step S and the ~64 VALU instructions of step S + 1's preparation, interleaved by
// sched_group_barrier requests (1 MFMA, then a sixth of the VALU work).
struct R64Planes { u32x4 h0, l0, h1, l1; };
__device__ __forceinline__ R64Planes r64_planes(lds_cu4* w, int tile_stride) { return R64Planes{w[0], w[64], w[tile_stride], w[tile_stride + 64]}; }
struct R64Vec { f32x4 g0, b0, g1, b1; };
__device__ __forceinline__ R64Vec r64_vec(const float* gv, const float* bv, int S, int h) {
    lds_cf4* const gl = (lds_cf4*)(gv + 4 * h);
    lds_cf4* const bl = (lds_cf4*)(bv + 4 * h);
    return R64Vec{gl[4 * S], bl[4 * S], gl[4 * S + 2], bl[4 * S + 2]};
}
template <bool FIRST>
__device__ __forceinline__ void r64_mma_p(f32x16 (&acc)[2], const R64Planes& p, const h8 bhi, const h8 blo) {
    if (FIRST) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, p.h0), bhi, z, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, p.h1), bhi, z, 0, 0, 0);
    } else {
        DSG_MFMA_H(acc[0], __builtin_bit_cast(h8, p.h0), bhi);
        DSG_MFMA_H(acc[1], __builtin_bit_cast(h8, p.h1), bhi);
    }
    DSG_MFMA_H(acc[0], __builtin_bit_cast(h8, p.h0), blo);
    DSG_MFMA_H(acc[1], __builtin_bit_cast(h8, p.h1), blo);
    DSG_MFMA_H(acc[0], __builtin_bit_cast(h8, p.l0), bhi);
    DSG_MFMA_H(acc[1], __builtin_bit_cast(h8, p.l1), bhi);
}
// the preparation with its vectors in registers and no scheduling barriers of its own (same arithmetic as panel_prep)
template <bool LNACT>
__device__ __forceinline__ void r64_prep_v(const float (&x)[8], const R64Vec& vv, float c, float d, h8& hi, h8& lo) {
    float v[8];
    if (LNACT) {
        constexpr float kk = -1.44269504088896341f / kActScale;
        const float g[8] = {vv.g0[0], vv.g0[1], vv.g0[2], vv.g0[3], vv.g1[0], vv.g1[1], vv.g1[2], vv.g1[3]};
        const float b[8] = {vv.b0[0], vv.b0[1], vv.b0[2], vv.b0[3], vv.b1[0], vv.b1[1], vv.b1[2], vv.b1[3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float u = fmaf(fmaf(x[q], c, d), g[q], b[q]);
            v[q] = u * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(u), kk, kk));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[q] * kRawScale;
    }
    split8(v, hi, lo);
}
// KS steps; getx(S, x) fills the eight input values of step S; `first` = the accumulators start from zero
template <bool LNACT, bool FIRST, int KS, typename GetX, typename Extra>
__device__ __forceinline__ void r64_stage(f32x16 (&acc)[2], lds_cu4* w, int tile_stride, const float* gv, const float* bv, float cc, float dd, int h,
                                          GetX&& getx, Extra&& extra) {
    R64Planes pc = r64_planes(w, tile_stride);
    h8 bh, bl;
    {
        R64Vec v0{};
        if (LNACT) v0 = r64_vec(gv, bv, 0, h);
        float x[8];
        getx(0, x);
        r64_prep_v<LNACT>(x, v0, cc, dd, bh, bl);
    }
#pragma unroll
    for (int S = 0; S < KS; ++S) {
        R64Planes pn = pc;
        R64Vec vn{};
        h8 nh = bh, nl = bl;
        if (S + 1 < KS) {                               // region A: requests of step S + 1
            pn = r64_planes(w + (S + 1) * 128, tile_stride);
            if (LNACT) vn = r64_vec(gv, bv, S + 1, h);
        }
        extra(S);                                       // the caller's own requests (next tile's input)
        __builtin_amdgcn_sched_barrier(0);
        if (S + 1 < KS) {                               // region B: MFMAs of step S + preparation of step S + 1
            float x[8];
            getx(S + 1, x);
            r64_prep_v<LNACT>(x, vn, cc, dd, nh, nl);
        }
        if (FIRST && S == 0) r64_mma_p<true>(acc, pc, bh, bl); else r64_mma_p<false>(acc, pc, bh, bl);
        if (S + 1 < KS) {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, LNACT ? 11 : 4, 0);     // a sixth of the preparation
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        pc = pn; bh = nh; bl = nl;
    }
}

template <int NQ>
__device__ __forceinline__ void r64_unscale_add(f32x16 (&acc)[2], float inv, const float* vec, int h) {
    lds_cf4* const v = (lds_cf4*)(vec + 4 * h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = v[8 * nt + 2 * q];
            acc[nt][4 * q + 0] = fmaf(acc[nt][4 * q + 0], inv, b[0]); acc[nt][4 * q + 1] = fmaf(acc[nt][4 * q + 1], inv, b[1]);
            acc[nt][4 * q + 2] = fmaf(acc[nt][4 * q + 2], inv, b[2]); acc[nt][4 * q + 3] = fmaf(acc[nt][4 * q + 3], inv, b[3]);
        }
}

struct R64Tile {                // where a row tile's operands live
    const float *x0, *x1, *cp;  // fragment tensors (+ lane * 4)
    const float2 *st0, *st1;    // row statistics (+ j)
    bool cond;
};

// NTO > 0: + the raw Linear that consumes the block (Upsample 64 -> 32 * NTO), its planes resident too; `store_block_out`: the
// block's own output is a skip tensor somebody else reads
template <bool SCLIN, int NTO>
__global__ __launch_bounds__(512, 2) void k_res64_lds(const BlockLinArgsH A, const int ngroups) {
    using L = R64Layout<SCLIN, NTO>;
    const BlockArgsH& ah = A.b;
    constexpr int N = 64, NT = 2, NG = 8, KS1 = L::KS1, XG = SCLIN ? 16 : 8;       // XG: 8-feature groups of the (concatenated) input
    __shared__ uint4 lds[L::TOTAL_U4];
    float* const vec = reinterpret_cast<float*>(lds + L::VEC);
    const BlockArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr float kL2 = -1.44269504088896341f;

    // ---- the block's planes and vectors -> LDS, once per launch.  EVERY load is issued before the first LDS write: written as
    // `for (i...) lds[i] = src[i]` hipcc emits load -> s_waitcnt vmcnt(0) -> ds_write per iteration, 13 dependent L2 round trips (1 us of
    // the 5-6 us every launch costs before its first tile: profiles/r03d_fixed_cost.txt)
    {
        constexpr int N1 = 2 * KS1 * 128 / 512, N2 = 2 * 4 * 128 / 512, NL = (NTO * 4 * 128 + 511) / 512;
        uint4 r1[N1], r2[N2], r3[N2], rs[SCLIN ? N1 : 1], rl[NTO > 0 ? NL : 1];
        float g1 = 0.f, b1 = 0.f, bl = 0.f, v6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, tb = 0.f;
        const int tid = threadIdx.x;
#pragma unroll
        for (int k = 0; k < N1; ++k) r1[k] = ah.W1h[tid + 512 * k];
#pragma unroll
        for (int k = 0; k < N2; ++k) { r2[k] = ah.W2h[tid + 512 * k]; r3[k] = ah.W3h[tid + 512 * k]; }
        if (SCLIN) {
#pragma unroll
            for (int k = 0; k < N1; ++k) rs[k] = ah.Wsch[tid + 512 * k];
        }
        if (NTO > 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k) {              // (whole 512-piece rounds unconditionally: a conditionally written array element sends
                rl[k] = make_uint4(0u, 0u, 0u, 0u);     // the array to scratch memory)
                if (512 * (k + 1) <= NTO * 4 * 128 || tid + 512 * k < NTO * 4 * 128) rl[k] = A.l.Wh[tid + 512 * k];
            }
            if (tid < 32 * NTO) bl = A.l.l.bias[tid];
        }
        if (tid < 16 * KS1) { g1 = a.gamma1[tid]; b1 = a.beta1[tid]; }
        if (tid < 64) {
            v6[0] = a.gamma2[tid]; v6[1] = a.beta2[tid]; v6[2] = a.gamma3[tid]; v6[3] = a.beta3[tid]; v6[4] = a.c2[tid]; v6[5] = a.c3[tid];
            tb = a.tbias[(size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride + tid];
        }
#pragma unroll
        for (int k = 0; k < N1; ++k) lds[L::W1 + tid + 512 * k] = r1[k];
#pragma unroll
        for (int k = 0; k < N2; ++k) { lds[L::W2 + tid + 512 * k] = r2[k]; lds[L::W3 + tid + 512 * k] = r3[k]; }
        if (SCLIN) {
#pragma unroll
            for (int k = 0; k < N1; ++k) lds[L::WSC + tid + 512 * k] = rs[k];
        }
        if (NTO > 0) {
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (512 * (k + 1) <= NTO * 4 * 128 || tid + 512 * k < NTO * 4 * 128) lds[L::WL + tid + 512 * k] = rl[k];
            if (tid < 32 * NTO) vec[L::BL + tid] = bl;
        }
        if (tid < 16 * KS1) { vec[L::G1 + tid] = g1 * kL2; vec[L::B1 + tid] = b1 * kL2; }
        if (tid < 64) {
            vec[L::G2 + tid] = v6[0] * kL2; vec[L::B2 + tid] = v6[1] * kL2; vec[L::G3 + tid] = v6[2] * kL2; vec[L::B3 + tid] = v6[3] * kL2;
            vec[L::C2 + tid] = v6[4]; vec[L::C3 + tid] = v6[5];
            vec[L::TB + tid] = tb;
        }
    }
    const float inv1 = ah.kc[0], inv2 = ah.kc[1], inv3 = ah.kc[2];
    float invL = 0.f;
    if (NTO > 0) invL = A.l.kc[0];
    __syncthreads();
    lds_cu4* const wl = (lds_cu4*)(lds + L::WL) + lane;
    lds_cu4* const w1 = (lds_cu4*)(lds + L::W1) + lane;
    lds_cu4* const w2 = (lds_cu4*)(lds + L::W2) + lane;
    lds_cu4* const w3 = (lds_cu4*)(lds + L::W3) + lane;
    lds_cu4* const wsc = (lds_cu4*)(lds + L::WSC) + lane;

    auto tile_of = [&](int g) -> R64Tile {
        const int traw = g * kR64Waves + wave;
        const int tile = traw < a.ntiles ? traw : a.ntiles - 1;
        const int ptile = tile >= a.tiles_per_pass ? tile - a.tiles_per_pass : tile;
        const int t0 = seg_tile(a.in0, tile), t1 = seg_tile(a.in1, tile);
        R64Tile t;
        t.x0 = a.in0.data + (size_t)t0 * 8 * 256 + lane * 4;
        t.st0 = reinterpret_cast<const float2*>(a.in0.stats) + (size_t)t0 * 32 + j;
        t.x1 = SCLIN ? a.in1.data + (size_t)t1 * 8 * 256 + lane * 4 : t.x0;
        t.st1 = SCLIN ? reinterpret_cast<const float2*>(a.in1.stats) + (size_t)t1 * 32 + j : t.st0;
        t.cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
        t.cond = tile >= a.uncond_tiles;
        return t;
    };
    auto xsrc = [&](const R64Tile& t, int G) -> const float* { return SCLIN && G >= 8 ? t.x1 + (size_t)(G - 8) * 256 : t.x0 + (size_t)G * 256; };

    // the first tile's input
    float4 xc[XG];
    float2 s0c, s1c;
    {
        const R64Tile t = tile_of(blockIdx.x < ngroups ? blockIdx.x : 0);
#pragma unroll
        for (int G = 0; G < XG; ++G) xc[G] = ld4(xsrc(t, G));
        s0c = *t.st0; s1c = *t.st1;
    }

    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        // the planes in LDS never change inside this loop: without a compiler barrier hipcc hoists ALL the plane reads out of it (loop-
        // invariant loads) and spills 700 registers
        asm volatile("" ::: "memory");
        const int tile_raw = g * kR64Waves + wave;
        const bool live = tile_raw < a.ntiles;
        const int tile = live ? tile_raw : a.ntiles - 1;
        const R64Tile cur = tile_of(g);
        const R64Tile nxt = tile_of(g + gridDim.x < ngroups ? g + gridDim.x : g);
        // condition embedding of this tile: requested now, added after stage 2
        float4 cpv[NG];
        if (cur.cond) {
#pragma unroll
            for (int G = 0; G < NG; ++G) cpv[G] = ld4(cur.cp + (size_t)G * 256);
        }

        // ---- LN1 statistics (Chan merge over the concat)
        float mean1, rstd1;
        {
            float mean = s0c.x, m2 = s0c.y;
            if (SCLIN) {
                const float dd = s1c.x - mean;
                m2 = m2 + s1c.y + dd * dd * a.chan_w;
                mean = mean + dd * a.chan_f;
            }
            mean1 = mean;
            rstd1 = rsqrtf(m2 * a.inv_nin + kLnEps);
            if (SCLIN) range_check(a.range_flag, mean, m2);
        }

        // ---- stage 1
        f32x16 acc1[NT];
        {
            const float cc = rstd1, dd = -mean1 * rstd1;
            r64_stage<true, true, KS1>(acc1, w1, KS1 * 128, vec + L::G1, vec + L::B1, cc, dd, h,
                                       [&](int S, float (&x)[8]) {
                                           x[0] = xc[2 * S].x; x[1] = xc[2 * S].y; x[2] = xc[2 * S].z; x[3] = xc[2 * S].w;
                                           x[4] = xc[2 * S + 1].x; x[5] = xc[2 * S + 1].y; x[6] = xc[2 * S + 1].z; x[7] = xc[2 * S + 1].w;
                                       }, [](int) {});
        }
        r64_unscale_add<8>(acc1, inv1, vec + L::TB, h);

        // ---- stage 2
        f32x16 acc2[NT];
        {
            float mean, m2;
            acc_stats<N, NT>(acc1, h, mean, m2);
            const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), cc = rstd, dd = -mean * rstd;
            r64_stage<true, true, 4>(acc2, w2, 4 * 128, vec + L::G2, vec + L::B2, cc, dd, h,
                                     [&](int S, float (&x)[8]) {
                                         const int t = S >> 1, r0 = 8 * (S & 1);
#pragma unroll
                                         for (int q = 0; q < 8; ++q) x[q] = acc1[t][r0 + q];
                                     }, [](int) {});
            r64_unscale_add<8>(acc2, inv2, vec + L::C2, h);
        }
        if (cur.cond) {
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                acc2[G >> 2][4 * (G & 3) + 0] += cpv[G].x; acc2[G >> 2][4 * (G & 3) + 1] += cpv[G].y;
                acc2[G >> 2][4 * (G & 3) + 2] += cpv[G].z; acc2[G >> 2][4 * (G & 3) + 3] += cpv[G].w;
            }
        }

        // ---- stage 3
        f32x16 (&acc3)[NT] = acc1;
        {
            float mean, m2;
            acc_stats<N, NT>(acc2, h, mean, m2);
            const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), cc = rstd, dd = -mean * rstd;
            r64_stage<true, true, 4>(acc3, w3, 4 * 128, vec + L::G3, vec + L::B3, cc, dd, h,
                                     [&](int S, float (&x)[8]) {
                                         const int t = S >> 1, r0 = 8 * (S & 1);
#pragma unroll
                                         for (int q = 0; q < 8; ++q) x[q] = acc2[t][r0 + q];
                                     }, [](int) {});
        }

        // ---- shortcut / residual from the SAME registers stage 1 read; the next tile's input is requested step by step behind it
        float4 xn[XG];
        float2 s0n, s1n;
        s0n = *nxt.st0; s1n = *nxt.st1;
        if (SCLIN) {
            r64_stage<false, false, KS1>(acc3, wsc, KS1 * 128, nullptr, nullptr, 0.f, 0.f, h,
                                         [&](int S, float (&x)[8]) {
                                             x[0] = xc[2 * S].x; x[1] = xc[2 * S].y; x[2] = xc[2 * S].z; x[3] = xc[2 * S].w;
                                             x[4] = xc[2 * S + 1].x; x[5] = xc[2 * S + 1].y; x[6] = xc[2 * S + 1].z; x[7] = xc[2 * S + 1].w;
                                         },
                                         [&](int S) { xn[2 * S] = ld4(xsrc(nxt, 2 * S)); xn[2 * S + 1] = ld4(xsrc(nxt, 2 * S + 1)); });
            r64_unscale_add<8>(acc3, inv3, vec + L::C3, h);
        } else {
            r64_unscale_add<8>(acc3, inv3, vec + L::C3, h);
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                acc3[G >> 2][4 * (G & 3) + 0] += xc[G].x; acc3[G >> 2][4 * (G & 3) + 1] += xc[G].y;
                acc3[G >> 2][4 * (G & 3) + 2] += xc[G].z; acc3[G >> 2][4 * (G & 3) + 3] += xc[G].w;
                xn[G] = ld4(xsrc(nxt, G));
            }
        }

        // ---- statistics + store
        float xmean, xm2;
        acc_stats<N, NT>(acc3, h, xmean, xm2);
        // (the lane index as an opaque value at the two store sites: hipcc otherwise forms the per-lane 64-bit output addresses in the
        // prologue, keeps them live across the whole tile loop and spills them in the 256-register variants)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        if (live && (NTO == 0 || A.store_block_out)) {
            if (lane_o < 32) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + lane_o] = make_float2(xmean, xm2);
#pragma unroll
            for (int G = 0; G < NG; ++G)
                st4(a.out + ((size_t)tile * NG + G) * 256 + lane_o * 4,
                    make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2], acc3[G >> 2][4 * (G & 3) + 3]));
        }
        if constexpr (NTO > 0) {
            // ---- the consuming raw Linear: K = 64 (4 k16-steps) from the block's output in registers
            range_check(a.range_flag, xmean, xm2);
            const LinArgs& la = A.l.l;
            f32x16 accL[NTO];
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const int t = S >> 1, r0 = 8 * (S & 1);
                const float x[8] = {acc3[t][r0], acc3[t][r0 + 1], acc3[t][r0 + 2], acc3[t][r0 + 3], acc3[t][r0 + 4], acc3[t][r0 + 5], acc3[t][r0 + 6], acc3[t][r0 + 7]};
                h8 bhi, blo;
                r64_prep<false>(x, nullptr, nullptr, 0, 0.f, 0.f, h, bhi, blo);
                HFrag<NTO> w;
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const u32x4 hh = wl[(nt * 4 + S) * 128], ll = wl[(nt * 4 + S) * 128 + 64];
                    w.hi[nt] = __builtin_bit_cast(uint4, hh); w.lo[nt] = __builtin_bit_cast(uint4, ll);
                }
                if (S == 0) mfma_step_h0<NTO>(accL, w, bhi, blo); else mfma_step_h<NTO>(accL, w, bhi, blo);
            }
            {
                lds_cf4* const v = (lds_cf4*)(vec + L::BL + 4 * h);
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b = v[8 * nt + 2 * q];
                        accL[nt][4 * q + 0] = fmaf(accL[nt][4 * q + 0], invL, b[0]); accL[nt][4 * q + 1] = fmaf(accL[nt][4 * q + 1], invL, b[1]);
                        accL[nt][4 * q + 2] = fmaf(accL[nt][4 * q + 2], invL, b[2]); accL[nt][4 * q + 3] = fmaf(accL[nt][4 * q + 3], invL, b[3]);
                    }
            }
            if (live) {
                const int NGo = (la.out_width + 7) / 8;
                float sm = 0.f;
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (8 * G + 4 * h + qq < la.out_width) sm += accL[G >> 2][4 * (G & 3) + qq];
                const float m = xhalf_sum(sm) * la.inv_out_w;
                float sq = 0.f;
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (8 * G + 4 * h + qq < la.out_width) { const float dv = accL[G >> 2][4 * (G & 3) + qq] - m; sq = fmaf(dv, dv, sq); }
                sq = xhalf_sum(sq);
                if (lane_o < 32) reinterpret_cast<float2*>(la.out_stats)[(size_t)tile * 32 + lane_o] = make_float2(m, sq);
#pragma unroll
                for (int G = 0; G < NTO * 4; ++G)
                    if (G < NGo)
                        st4(la.out + ((size_t)tile * NGo + G) * 256 + lane_o * 4,
                            make_float4(accL[G >> 2][4 * (G & 3)], accL[G >> 2][4 * (G & 3) + 1], accL[G >> 2][4 * (G & 3) + 2], accL[G >> 2][4 * (G & 3) + 3]));
            }
        }
#pragma unroll
        for (int G = 0; G < XG; ++G) xc[G] = xn[G];
        s0c = s0n; s1c = s1n;
    }
}

// ---- two consecutive down-64 blocks (no Linear shortcut, no consuming Linear) in ONE launch: both blocks' planes are resident
// (2 x 48 KiB), the first block's output is stored (it is a skip tensor) AND handed to the second block in registers -- a launch's
// fixed cost (5-6 us, profiles/r03d_fixed_cost.txt) and one read of that tensor less per step.  Arithmetic as k_res64_lds<false, 0>.
__global__ __launch_bounds__(512, 2) void k_res64_dual(const BlockArgsH A0, const BlockArgsH A1, const int ngroups) {
    using L = R64Layout<false, 0>;
    constexpr int N = 64, NT = 2, NG = 8, KS1 = 4;
    __shared__ uint4 lds[2 * L::TOTAL_U4];
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr float kL2 = -1.44269504088896341f;
    const int tid = threadIdx.x;
    {   // every load of both blocks first, then the LDS writes
        uint4 r1[2][2], r2[2][2], r3[2][2];
        float g1[2] = {0.f, 0.f}, b1[2] = {0.f, 0.f}, v6[2][6], tb[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const BlockArgsH& ah = k == 0 ? A0 : A1;
            const BlockArgs& a = ah.b;
#pragma unroll
            for (int q = 0; q < 2; ++q) { r1[k][q] = ah.W1h[tid + 512 * q]; r2[k][q] = ah.W2h[tid + 512 * q]; r3[k][q] = ah.W3h[tid + 512 * q]; }
            if (tid < 16 * KS1) { g1[k] = a.gamma1[tid]; b1[k] = a.beta1[tid]; }
#pragma unroll
            for (int q = 0; q < 6; ++q) v6[k][q] = 0.f;
            if (tid < 64) {
                v6[k][0] = a.gamma2[tid]; v6[k][1] = a.beta2[tid]; v6[k][2] = a.gamma3[tid]; v6[k][3] = a.beta3[tid]; v6[k][4] = a.c2[tid]; v6[k][5] = a.c3[tid];
                tb[k] = a.tbias[(size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride + tid];
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint4* const lb = lds + k * L::TOTAL_U4;
            float* const vec = reinterpret_cast<float*>(lb + L::VEC);
#pragma unroll
            for (int q = 0; q < 2; ++q) { lb[L::W1 + tid + 512 * q] = r1[k][q]; lb[L::W2 + tid + 512 * q] = r2[k][q]; lb[L::W3 + tid + 512 * q] = r3[k][q]; }
            if (tid < 16 * KS1) { vec[L::G1 + tid] = g1[k] * kL2; vec[L::B1 + tid] = b1[k] * kL2; }
            if (tid < 64) {
                vec[L::G2 + tid] = v6[k][0] * kL2; vec[L::B2 + tid] = v6[k][1] * kL2; vec[L::G3 + tid] = v6[k][2] * kL2; vec[L::B3 + tid] = v6[k][3] * kL2;
                vec[L::C2 + tid] = v6[k][4]; vec[L::C3 + tid] = v6[k][5];
                vec[L::TB + tid] = tb[k];
            }
        }
    }
    __syncthreads();
    const BlockArgs& a0 = A0.b;
    auto x0_of = [&](int g) -> const float* {
        const int traw = g * kR64Waves + wave;
        const int tile = traw < a0.ntiles ? traw : a0.ntiles - 1;
        return a0.in0.data + (size_t)seg_tile(a0.in0, tile) * 8 * 256 + lane * 4;
    };
    auto st0_of = [&](int g) -> const float2* {
        const int traw = g * kR64Waves + wave;
        const int tile = traw < a0.ntiles ? traw : a0.ntiles - 1;
        return reinterpret_cast<const float2*>(a0.in0.stats) + (size_t)seg_tile(a0.in0, tile) * 32 + j;
    };
    float4 xc[NG];
    float2 sc;
    {
        const float* xp = x0_of(blockIdx.x < ngroups ? blockIdx.x : 0);
#pragma unroll
        for (int G = 0; G < NG; ++G) xc[G] = ld4(xp + (size_t)G * 256);
        sc = *st0_of(blockIdx.x < ngroups ? blockIdx.x : 0);
    }
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        asm volatile("" ::: "memory");                   // as k_res64_lds: keeps the plane reads inside the loop
        const int tile_raw = g * kR64Waves + wave;
        const bool live = tile_raw < a0.ntiles;
        const int tile = live ? tile_raw : a0.ntiles - 1;
        const int ptile = tile >= a0.tiles_per_pass ? tile - a0.tiles_per_pass : tile;
        const bool cond = tile >= a0.uncond_tiles;
        const int gn = g + gridDim.x < ngroups ? g + gridDim.x : g;
        float4 xn[NG];
        float2 sn = sc;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const BlockArgsH& ah = k == 0 ? A0 : A1;
            const BlockArgs& a = ah.b;
            lds_cu4* const w1 = (lds_cu4*)(lds + k * L::TOTAL_U4 + L::W1) + lane;
            lds_cu4* const w2 = (lds_cu4*)(lds + k * L::TOTAL_U4 + L::W2) + lane;
            lds_cu4* const w3 = (lds_cu4*)(lds + k * L::TOTAL_U4 + L::W3) + lane;
            const float* const vec = reinterpret_cast<const float*>(lds + k * L::TOTAL_U4 + L::VEC);
            const float inv1 = ah.kc[0], inv2 = ah.kc[1], inv3 = ah.kc[2];
            float4 cpv[NG];
            if (cond) {
                const float* cp = a.cond_pre + (size_t)ptile * NG * 256 + lane * 4;
#pragma unroll
                for (int G = 0; G < NG; ++G) cpv[G] = ld4(cp + (size_t)G * 256);
            }
            const float rstd1 = rsqrtf(sc.y * a.inv_nin + kLnEps), mean1 = sc.x;
            f32x16 acc1[NT];
            r64_stage<true, true, KS1>(acc1, w1, KS1 * 128, vec + L::G1, vec + L::B1, rstd1, -mean1 * rstd1, h,
                                       [&](int S, float (&x)[8]) {
                                           x[0] = xc[2 * S].x; x[1] = xc[2 * S].y; x[2] = xc[2 * S].z; x[3] = xc[2 * S].w;
                                           x[4] = xc[2 * S + 1].x; x[5] = xc[2 * S + 1].y; x[6] = xc[2 * S + 1].z; x[7] = xc[2 * S + 1].w;
                                       }, [](int) {});
            r64_unscale_add<8>(acc1, inv1, vec + L::TB, h);
            f32x16 acc2[NT];
            {
                float mean, m2;
                acc_stats<N, NT>(acc1, h, mean, m2);
                const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
                r64_stage<true, true, 4>(acc2, w2, 4 * 128, vec + L::G2, vec + L::B2, rstd, -mean * rstd, h,
                                         [&](int S, float (&x)[8]) {
                                             const int t = S >> 1, r0 = 8 * (S & 1);
#pragma unroll
                                             for (int q = 0; q < 8; ++q) x[q] = acc1[t][r0 + q];
                                         }, [](int) {});
                r64_unscale_add<8>(acc2, inv2, vec + L::C2, h);
            }
            if (cond) {
#pragma unroll
                for (int G = 0; G < NG; ++G) {
                    acc2[G >> 2][4 * (G & 3) + 0] += cpv[G].x; acc2[G >> 2][4 * (G & 3) + 1] += cpv[G].y;
                    acc2[G >> 2][4 * (G & 3) + 2] += cpv[G].z; acc2[G >> 2][4 * (G & 3) + 3] += cpv[G].w;
                }
            }
            f32x16 (&acc3)[NT] = acc1;
            {
                float mean, m2;
                acc_stats<N, NT>(acc2, h, mean, m2);
                const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
                r64_stage<true, true, 4>(acc3, w3, 4 * 128, vec + L::G3, vec + L::B3, rstd, -mean * rstd, h,
                                         [&](int S, float (&x)[8]) {
                                             const int t = S >> 1, r0 = 8 * (S & 1);
#pragma unroll
                                             for (int q = 0; q < 8; ++q) x[q] = acc2[t][r0 + q];
                                         }, [](int) {});
            }
            r64_unscale_add<8>(acc3, inv3, vec + L::C3, h);
            if (k == 1) {                              // the next tile's input of the FIRST block, behind the residual add
                const float* xp = x0_of(gn);
#pragma unroll
                for (int G = 0; G < NG; ++G) xn[G] = ld4(xp + (size_t)G * 256);
                sn = *st0_of(gn);
            }
#pragma unroll
            for (int G = 0; G < NG; ++G) {
                acc3[G >> 2][4 * (G & 3) + 0] += xc[G].x; acc3[G >> 2][4 * (G & 3) + 1] += xc[G].y;
                acc3[G >> 2][4 * (G & 3) + 2] += xc[G].z; acc3[G >> 2][4 * (G & 3) + 3] += xc[G].w;
            }
            float xmean, xm2;
            acc_stats<N, NT>(acc3, h, xmean, xm2);
            if (live) {
                if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(xmean, xm2);
#pragma unroll
                for (int G = 0; G < NG; ++G)
                    st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                        make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2], acc3[G >> 2][4 * (G & 3) + 3]));
            }
            if (k == 0) {                              // the second block's input: this block's output, in registers
#pragma unroll
                for (int G = 0; G < NG; ++G) xc[G] = make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2], acc3[G >> 2][4 * (G & 3) + 3]);
                sc = make_float2(xmean, xm2);
            }
        }
#pragma unroll
        for (int G = 0; G < NG; ++G) xc[G] = xn[G];
        sc = sn;
    }
}


}  // namespace dsg
