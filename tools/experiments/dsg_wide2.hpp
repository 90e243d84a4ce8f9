// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 2): 128-wide ResidualBlocks, ping-pong form -- one 8-wave workgroup per CU,
// its two halves one phase apart.  Parity-green (the sampling goldens under the large-launch policy and the 16 417- / 65 536-row
// oracle checks passed with it wired in behind DSG_WIDE_FORM=1), but SLOWER than k_wide128_h: up.17.res 158.5 vs 111.6 us
// (dsg_time_op, same box), down.0.res 93.6 vs ~75.  Why: every phase boundary is a barrier of ALL the waves of the CU, 96 per
// block instead of 44 among 4 of 8 waves, and the latency of one boundary (counted vmcnt, s_barrier, the unit-table read and the
// DMA issue: ~400 cycles) is exposed to both halves at once -- with two independent 4-wave workgroups per CU the other workgroup
// keeps the SIMDs busy while one sits in its barrier.  Kept as the record of that measurement; to try it again include it
// after dsg_wide.hpp and dispatch k_wide128p_h with grid ntiles / 8, block 512.
//
// k_wide128_h (dsg_wide.hpp) left MFMA time, VALU time and the wait for data ADDING UP: the PMC counters show
// the matrix core busy 35 % of the launch and VALU executing beside it in only 28 % of that time
// (SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES, profiles/r02a_pmc_summary.txt) -- the two waves of a SIMD, from
// two independent workgroups that each run in barrier lockstep, do not stay out of phase.  tools/ubench/overlap2.hip
// measures what out-of-phase partners get: an MFMA-only wave keeps its 32 cycles per MFMA beside a VALU-only wave that
// runs at 90 % of its solo rate.  This kernel makes the phases explicit:
//
//   * a workgroup is 8 waves = 8 row tiles; waves w and w + 4 share a SIMD (the platform guide's placement rule);
//   * the block is a sequence of PHASES, alternately V (fetch this step's operands from the LDS ring, LayerNorm + SiLU +
//     hi/lo split: VALU and LDS only) and M (the step's 12 MFMAs: nothing else);
//   * every phase boundary is one s_barrier for all 8 waves; waves 4-7 run the SAME code one barrier later (they pass one
//     extra barrier before the program, waves 0-3 one after it), so in every interval one half is in a V phase while the
//     other is in an M phase, on every SIMD, by construction.
//
// Data movement is the LDS-DMA ring of dsg_wide.hpp scaled to the workgroup: 16 slots x 8 KiB, a W unit = the 8 planes of one
// k16-step (one piece per wave), an X2 unit = two 8-feature groups of every wave's own tile (16 KiB, one two-DMA statement
// per wave).  All bookkeeping is keyed on the interval index t, identical in all 8 waves: before barrier t every wave has
// waited (counted vmcnt) for its pieces of the units of phase t; after it the units of phases <= t-2 are free (waves 0-3
// read phase t-1 in interval t-1, waves 4-7 in interval t... which is this one: so t-2), and every wave issues the next
// units while they fit.  The unit list and the cumulative (DMA, chunk, unit) counts per phase are built once per wave / per
// workgroup in LDS.  The condition embedding and the residual input ride in the V phases of stages 2 / 3 and are added to
// the accumulator IN THE SCALED DOMAIN between two steps' MFMAs (an exact power-of-two scaling), so the ring never carries
// more than three chunks per phase.
#pragma once
#include "dsg_wide.hpp"

namespace dsg {

constexpr int kPRing = 16;                 // chunks
constexpr int kPRingBytes = kPRing * 8192;
constexpr int kPMaxUnits = 160;

struct PProg {                             // one block (+ epilogue Linear) as units and phases; same for all waves except addresses
    int ks0, KS1;
    bool sclin, cond;
    int epi_steps;                         // k16-steps of the epilogue Linear (0: none)
    const float *x0, *x1, *cp;             // this wave's tile of in0 / in1 / cond_pre
    const uint4 *w1, *w2, *w3, *wsc, *wl;  // this wave's plane of each packed matrix, step 0 (wave w: out tile w >> 1, plane w & 1)
    int lin_ok;                            // this wave's plane exists in the epilogue Linear (out tile < NTO)

    // Phases: stage 1: V(k) = {X2, W}, M(k) = {} ; stage 2: V = {W (+ P)}, M ; stage 3: V = {W (+ R)}, M ; shortcut: V = {X2, W}, M ;
    // epilogue: V = {W}, M.
    // unit index -> (phase step decode) ; returns source address with bit 0 = pair (two groups / two DMAs, 2 chunks)
    __device__ __forceinline__ unsigned long long unit_source(int u) const {
        const void* src; bool pair;
        const int nA = 2 * KS1, nB = nA + (cond ? 16 : 8), nC = nB + (sclin ? 8 : 16), nD = nC + (sclin ? 2 * KS1 : 0);
        auto xsrc = [&](int S) { return S < ks0 ? (const void*)(x0 + (size_t)S * 512) : (const void*)(x1 + (size_t)(S - ks0) * 512); };
        if (u < nA) { const int S = u >> 1; pair = !(u & 1); src = pair ? xsrc(S) : (const void*)(w1 + (size_t)S * 128); }
        else if (u < nB) {
            const int i = u - nA;
            if (cond) { const int S = i >> 1; pair = (i & 1); src = pair ? (const void*)(cp + (size_t)S * 512) : (const void*)(w2 + (size_t)S * 128); }
            else { pair = false; src = w2 + (size_t)i * 128; }
        } else if (u < nC) {
            const int i = u - nB;
            if (sclin) { pair = false; src = w3 + (size_t)i * 128; }
            else { const int S = i >> 1; pair = (i & 1); src = pair ? (const void*)(x0 + (size_t)S * 512) : (const void*)(w3 + (size_t)S * 128); }
        } else if (u < nD) { const int i = u - nC, S = i >> 1; pair = !(i & 1); src = pair ? xsrc(S) : (const void*)(wsc + (size_t)S * 128); }
        else { pair = false; src = wl + (size_t)(u - nD) * 128; }
        return reinterpret_cast<unsigned long long>(src) | (pair ? 1ull : 0ull);
    }
};

struct PRing {
    const uint4* rd;                       // ring as ordinary LDS, + lane
    const unsigned* utab;                  // this wave's unit sources (lo, hi words)
    unsigned long long ns0, ns1;           // next two unit sources (wave-uniform: read through readfirstlane)
    unsigned voff;                         // lane * 16
    unsigned lds0;                         // LDS byte address of the ring
    int wave, half;
    int nu;                                // units
    int pu;                                // next unit to issue
    int pos_issue;                         // chunk position of the next unit (= chunks = DMAs this wave has issued)
    int cert;                              // chunks through the phase that must have landed at the next barrier
    int freec;                             // chunks through the newest phase both halves are done with
    int h1, h2;                            // chunk counts of the two phases before this wave's current one
};

// two / one 1 KiB pieces: global (wave-uniform base + per-lane byte offset) -> LDS
__device__ __forceinline__ void glds_pair_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_one_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void p_wait_vm(int n) {   // wave-uniform
    if (n >= 8) {
        if (n >= 12) {
            if (n >= 14) { if (n >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
            else { if (n >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        } else {
            if (n >= 10) { if (n >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
            else { if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        }
    } else {
        if (n >= 4) {
            if (n >= 6) { if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else { if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        } else {
            if (n >= 2) { if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else { if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
    }
}

__device__ __forceinline__ unsigned long long p_tab(const PRing& r, int u) {
    const int i = u < kPMaxUnits ? u : kPMaxUnits - 1;
    const unsigned lo = __builtin_amdgcn_readfirstlane(r.utab[2 * i]), hi = __builtin_amdgcn_readfirstlane(r.utab[2 * i + 1]);
    return ((unsigned long long)hi << 32) | lo;
}

// issue the units that fit: chunks through `freec` are free
__device__ __forceinline__ void p_issue(PRing& r) {
    while (r.pu < r.nu) {
        const bool pair = (r.ns0 & 1ull) != 0;
        const int nch = pair ? 2 : 1;
        if (r.pos_issue + nch - r.freec > kPRing) break;
        const void* src = reinterpret_cast<const void*>(r.ns0 & ~1ull);
        const unsigned dst = r.lds0 + (((unsigned)r.pos_issue * 8192u + (unsigned)r.wave * (pair ? 2048u : 1024u)) & (unsigned)(kPRingBytes - 1));
        if (pair) glds_pair_s(r.voff, src, dst);
        else glds_one_s(r.voff, src, dst);
        r.pos_issue += nch; ++r.pu;
        r.ns0 = r.ns1;
        r.ns1 = p_tab(r, r.pu + 1);
    }
}

// One interval boundary.  n_own / n_next: chunks of the units of the phase this wave is about to run and of the one after it.
// All 8 waves do the same bookkeeping at the same barrier: before it, the units of the phase waves 0-3 are about to run
// (= the one after the phase waves 4-7 are about to run) have landed; after it, the phases both halves have finished are free.
__device__ __forceinline__ void p_sync(PRing& r, int n_own, int n_next) {
    r.cert += r.half ? n_next : n_own;
    const int allowed = r.pos_issue - r.cert;
    p_wait_vm(allowed < 0 ? 0 : allowed);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    r.freec += r.half ? r.h1 : r.h2;
    r.h2 = r.h1; r.h1 = n_own;
    p_issue(r);
}

__device__ __forceinline__ const uint4* p_x2(const PRing& r, int pos) {   // this wave's two groups of an X2 unit at chunk position pos
    return r.rd + ((((unsigned)pos * 8192u + (unsigned)r.wave * 2048u) & (unsigned)(kPRingBytes - 1)) >> 4);
}
__device__ __forceinline__ const uint4* p_chunk(const PRing& r, int pos) {
    return r.rd + ((((unsigned)pos * 8192u) & (unsigned)(kPRingBytes - 1)) >> 4);
}
__device__ __forceinline__ void p_wfrag(HFrag<4>& w, const PRing& r, int pos) {
    const uint4* s = p_chunk(r, pos);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { w.hi[nt] = s[(2 * nt) * 64]; w.lo[nt] = s[(2 * nt + 1) * 64]; }
}

template <bool SCLIN, int EPI, int NTO>
__global__ __launch_bounds__(512, 2) void k_wide128p_h(const BlockLinArgsH A) {
    constexpr int N = 128, NT = 4, NG = 16;
    __shared__ uint4 lds[kPRingBytes / 16 + kWideVec / 4 + 8 * kPMaxUnits / 2];
    float* const vec = reinterpret_cast<float*>(lds + kPRingBytes / 16);
    unsigned long long* const utab_all = reinterpret_cast<unsigned long long*>(lds + kPRingBytes / 16 + kWideVec / 4);
    float* const g1v = vec, * const b1v = vec + kLnLdsW1, * const v2 = vec + 2 * kLnLdsW1;
    float* const g2v = v2, * const b2v = v2 + 128, * const g3v = v2 + 256, * const b3v = v2 + 384, * const tbv = v2 + 512, * const c2v = v2 + 640,
         * const c3v = v2 + 768, * const gLv = v2 + 896, * const bLv = v2 + 1024, * const biasLv = v2 + 1152;
    const BlockArgsH& ah = A.b;
    const BlockArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave >> 2;
    const int tile_raw = blockIdx.x * 8 + wave;
    const bool live = tile_raw < a.ntiles;
    const int tile = live ? tile_raw : a.ntiles - 1;
    const int ptile = tile % a.tiles_per_pass;
    const int ks0 = a.in0.groups >> 1, ks1 = a.in1.groups >> 1, KS1 = ks0 + ks1;
    const int last_tile = blockIdx.x * 8 + 7 < a.ntiles ? blockIdx.x * 8 + 7 : a.ntiles - 1;
    const bool wg_cond = last_tile >= a.uncond_tiles, my_cond = tile >= a.uncond_tiles;
    constexpr float kL2 = -1.44269504088896341f;

    // ---- per-feature vectors -> LDS (LayerNorm vectors times -log2 e)
    {
        const int n1 = ln1_extent(a);
        for (int i = threadIdx.x; i < n1; i += 512) { g1v[i] = a.gamma1[i] * kL2; b1v[i] = a.beta1[i] * kL2; }
        if (threadIdx.x < 128) {
            const int i = threadIdx.x;
            g2v[i] = a.gamma2[i] * kL2; b2v[i] = a.beta2[i] * kL2; g3v[i] = a.gamma3[i] * kL2; b3v[i] = a.beta3[i] * kL2;
            c2v[i] = a.c2[i]; c3v[i] = a.c3[i];
            if (!a.ts) tbv[i] = a.tbias[(size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride + i];
            if (EPI == 2) { gLv[i] = A.l.l.gamma[i] * kL2; bLv[i] = A.l.l.beta[i] * kL2; }
            if (EPI != 0) biasLv[i] = i < NTO * 32 ? A.l.l.bias[i] : 0.f;
        }
    }
    // ---- LN1 statistics (Chan merge of the producers' (mean, M2))
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

    // ---- program: unit sources of this wave, cumulative counts per phase
    PProg p;
    p.ks0 = ks0; p.KS1 = KS1; p.sclin = SCLIN; p.cond = wg_cond; p.epi_steps = EPI != 0 ? 8 : 0;
    p.x0 = a.in0.data + (size_t)seg_tile(a.in0, tile) * a.in0.groups * 256;
    p.x1 = a.in1.groups ? a.in1.data + (size_t)seg_tile(a.in1, tile) * a.in1.groups * 256 : p.x0;
    p.cp = a.cond_pre + (size_t)ptile * NG * 256;
    const int wnt = wave >> 1, wpl = wave & 1;                                 // this wave's out tile and plane of a W unit
    p.w1 = ah.W1h + ((size_t)wnt * KS1 * 2 + wpl) * 64; p.w2 = ah.W2h + ((size_t)wnt * 8 * 2 + wpl) * 64;
    p.w3 = ah.W3h + ((size_t)wnt * 8 * 2 + wpl) * 64;
    p.wsc = SCLIN ? ah.Wsch + ((size_t)wnt * KS1 * 2 + wpl) * 64 : p.w1;
    p.wl = EPI != 0 ? A.l.Wh + ((size_t)(wnt < NTO ? wnt : 0) * 8 * 2 + wpl) * 64 : p.w1;
    PRing r;
    {
        unsigned long long* utab = utab_all + wave * kPMaxUnits;
        const int nu = 2 * KS1 + (wg_cond ? 16 : 8) + (SCLIN ? 8 + 2 * KS1 : 16) + (EPI != 0 ? 8 : 0);
        r.nu = nu;
        for (int u = lane; u < kPMaxUnits; u += 64) utab[u] = p.unit_source(u < nu ? u : nu - 1);
        r.utab = reinterpret_cast<const unsigned*>(utab);
    }
    r.rd = lds + lane; r.wave = wave; r.half = half; r.pu = 0; r.pos_issue = 0; r.cert = 0; r.freec = 0; r.h1 = 0; r.h2 = 0;
    r.voff = (unsigned)lane * 16u;
    r.lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds;
    __syncthreads();
    r.ns0 = p_tab(r, 0); r.ns1 = p_tab(r, 1);
    p_issue(r);                                                                // fill the ring
    const int nS2 = wg_cond ? 3 : 1, nS3 = SCLIN ? 1 : 3;                      // chunks of a V phase of stage 2 / stage 3
    if (half) p_sync(r, 0, 3);                                                 // waves 4-7 run one interval behind
    int pos = 0;                                                               // chunk position of this wave's current phase

    // ---- stage 1: V = x (ring) -> LayerNorm + SiLU + split, W1 planes -> registers ; M = 12 MFMAs
    f32x16 acc1[NT];
    {
        const float c = rstd1, d = -mean1 * rstd1;
        for (int S = 0; S < KS1; ++S) {
            p_sync(r, 3, 0);
            const uint4* xs = p_x2(r, pos);
            const float4 xa = __builtin_bit_cast(float4, xs[0]), xb = __builtin_bit_cast(float4, xs[64]);
            HFrag<4> w;
            p_wfrag(w, r, pos + 2);
            pos += 3;
            const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const BOp b = wide_prep<true>(x, g1v, b1v, S, c, d, h);
            p_sync(r, 0, S + 1 < KS1 ? 3 : nS2);
            if (S == 0) wide_mma<true>(acc1, w, b); else wide_mma<false>(acc1, w, b);
        }
    }
    if (a.ts) acc_unscale_add<NT>(acc1, inv1, a.tbias + (size_t)entry * a.tb_stride, h);
    else acc_unscale_add_lds<NT>(acc1, inv1, tbv, h);
    if (a.save_h1 && live) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h1 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc1[G >> 2][4 * (G & 3)], acc1[G >> 2][4 * (G & 3) + 1], acc1[G >> 2][4 * (G & 3) + 2], acc1[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 2 (the condition embedding rides in its V phases: added scaled by 1 / inv2, an exact power of two)
    f32x16 acc2[NT];
    {
        float mean, m2;
        acc_stats<N, NT>(acc1, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), c = rstd, d = -mean * rstd;
        const float sc2 = 1.0f / inv2;
        if (wg_cond) wacc_zero<NT>(acc2);
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            p_sync(r, nS2, 0);
            HFrag<4> w;
            p_wfrag(w, r, pos);
            if (wg_cond) {
                const uint4* cs = p_x2(r, pos + 1);
                const float4 ca = __builtin_bit_cast(float4, cs[0]), cb = __builtin_bit_cast(float4, cs[64]);
                if (my_cond) {
                    const int G = 2 * S;
                    acc2[G >> 2][4 * (G & 3) + 0] += ca.x * sc2; acc2[G >> 2][4 * (G & 3) + 1] += ca.y * sc2;
                    acc2[G >> 2][4 * (G & 3) + 2] += ca.z * sc2; acc2[G >> 2][4 * (G & 3) + 3] += ca.w * sc2;
                    acc2[(G + 1) >> 2][4 * ((G + 1) & 3) + 0] += cb.x * sc2; acc2[(G + 1) >> 2][4 * ((G + 1) & 3) + 1] += cb.y * sc2;
                    acc2[(G + 1) >> 2][4 * ((G + 1) & 3) + 2] += cb.z * sc2; acc2[(G + 1) >> 2][4 * ((G + 1) & 3) + 3] += cb.w * sc2;
                }
            }
            const int t_ = S >> 1, r0 = 8 * (S & 1);
            const float x[8] = {acc1[t_][r0], acc1[t_][r0 + 1], acc1[t_][r0 + 2], acc1[t_][r0 + 3], acc1[t_][r0 + 4], acc1[t_][r0 + 5], acc1[t_][r0 + 6],
                                acc1[t_][r0 + 7]};
            pos += nS2;
            const BOp b = wide_prep<true>(x, g2v, b2v, S, c, d, h);
            p_sync(r, 0, S + 1 < 8 ? nS2 : nS3);
            if (S == 0 && !wg_cond) wide_mma<true>(acc2, w, b); else wide_mma<false>(acc2, w, b);
        }
        acc_unscale_add_lds<NT>(acc2, inv2, c2v, h);
    }
    if (a.save_h2 && live) {
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.save_h2 + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc2[G >> 2][4 * (G & 3)], acc2[G >> 2][4 * (G & 3) + 1], acc2[G >> 2][4 * (G & 3) + 2], acc2[G >> 2][4 * (G & 3) + 3]));
    }

    // ---- stage 3 (identity shortcut: the residual input rides in its V phases, scaled by 1 / inv3)
    f32x16 (&acc3)[NT] = acc1;
    {
        float mean, m2;
        acc_stats<N, NT>(acc2, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), c = rstd, d = -mean * rstd;
        const float sc3 = 1.0f / inv3;
        if (!SCLIN) wacc_zero<NT>(acc3);
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            p_sync(r, nS3, 0);
            HFrag<4> w;
            p_wfrag(w, r, pos);
            if (!SCLIN) {
                const uint4* cs = p_x2(r, pos + 1);
                const float4 ca = __builtin_bit_cast(float4, cs[0]), cb = __builtin_bit_cast(float4, cs[64]);
                const int G = 2 * S;
                acc3[G >> 2][4 * (G & 3) + 0] += ca.x * sc3; acc3[G >> 2][4 * (G & 3) + 1] += ca.y * sc3;
                acc3[G >> 2][4 * (G & 3) + 2] += ca.z * sc3; acc3[G >> 2][4 * (G & 3) + 3] += ca.w * sc3;
                acc3[(G + 1) >> 2][4 * ((G + 1) & 3) + 0] += cb.x * sc3; acc3[(G + 1) >> 2][4 * ((G + 1) & 3) + 1] += cb.y * sc3;
                acc3[(G + 1) >> 2][4 * ((G + 1) & 3) + 2] += cb.z * sc3; acc3[(G + 1) >> 2][4 * ((G + 1) & 3) + 3] += cb.w * sc3;
            }
            const int t_ = S >> 1, r0 = 8 * (S & 1);
            const float x[8] = {acc2[t_][r0], acc2[t_][r0 + 1], acc2[t_][r0 + 2], acc2[t_][r0 + 3], acc2[t_][r0 + 4], acc2[t_][r0 + 5], acc2[t_][r0 + 6],
                                acc2[t_][r0 + 7]};
            pos += nS3;
            const BOp b = wide_prep<true>(x, g3v, b3v, S, c, d, h);
            p_sync(r, 0, S + 1 < 8 ? nS3 : (SCLIN ? 3 : (EPI != 0 ? 1 : 0)));
            if (S == 0 && SCLIN) wide_mma<true>(acc3, w, b); else wide_mma<false>(acc3, w, b);
        }
    }
    if (SCLIN) {
        for (int S = 0; S < KS1; ++S) {
            p_sync(r, 3, 0);
            const uint4* xs = p_x2(r, pos);
            const float4 xa = __builtin_bit_cast(float4, xs[0]), xb = __builtin_bit_cast(float4, xs[64]);
            HFrag<4> w;
            p_wfrag(w, r, pos + 2);
            pos += 3;
            const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const BOp b = wide_prep<false>(x, nullptr, nullptr, S, 0.f, 0.f, h);
            p_sync(r, 0, S + 1 < KS1 ? 3 : (EPI != 0 ? 1 : 0));
            wide_mma<false>(acc3, w, b);
        }
    }
    acc_unscale_add_lds<NT>(acc3, inv3, c3v, h);

    // ---- statistics + store
    float xmean, xm2;
    acc_stats<N, NT>(acc3, h, xmean, xm2);
    if ((EPI == 0 || A.store_block_out) && live) {
        if (h == 0) reinterpret_cast<float2*>(a.out_stats)[(size_t)tile * 32 + j] = make_float2(xmean, xm2);
#pragma unroll
        for (int G = 0; G < NG; ++G)
            st4(a.out + ((size_t)tile * NG + G) * 256 + lane * 4,
                make_float4(acc3[G >> 2][4 * (G & 3)], acc3[G >> 2][4 * (G & 3) + 1], acc3[G >> 2][4 * (G & 3) + 2], acc3[G >> 2][4 * (G & 3) + 3]));
    }
    if (EPI == 0) {
        if (!half) p_sync(r, 0, 0);                                            // waves 0-3: the barrier waves 4-7 passed first
        return;
    }
    if (EPI == 1) range_check(a.range_flag, xmean, xm2);

    // ---- epilogue Linear: one W unit per step (planes of out tiles >= NTO are dummies)
    const LinArgs& la = A.l.l;
    f32x16 acc[NTO];
    {
        const float c = EPI == 2 ? rsqrtf(xm2 * la.inv_in_w + kLnEps) : 1.f, d = -xmean * c;
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            p_sync(r, 1, 0);
            const uint4* s = p_chunk(r, pos);
            pos += 1;
            HFrag<NTO> w;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) { w.hi[nt] = s[(2 * nt) * 64]; w.lo[nt] = s[(2 * nt + 1) * 64]; }
            const int t_ = S >> 1, r0 = 8 * (S & 1);
            const float x[8] = {acc3[t_][r0], acc3[t_][r0 + 1], acc3[t_][r0 + 2], acc3[t_][r0 + 3], acc3[t_][r0 + 4], acc3[t_][r0 + 5], acc3[t_][r0 + 6],
                                acc3[t_][r0 + 7]};
            BOp b;
            if (EPI == 2) b = wide_prep<true>(x, gLv, bLv, S, c, d, h); else b = wide_prep<false>(x, nullptr, nullptr, S, 0.f, 0.f, h);
            p_sync(r, 0, S + 1 < 8 ? 1 : 0);
            if (S == 0) mfma_step_h0<NTO>(acc, w, b.hi, b.lo); else mfma_step_h<NTO>(acc, w, b.hi, b.lo);
        }
    }
    if (!half) p_sync(r, 0, 0);
    acc_unscale_add_lds<NTO>(acc, A.l.kc[EPI == 2 ? 1 : 0], biasLv, h);
    if (!live) return;
    if (EPI == 1) {
        const int NGo = (la.out_width + 7) / 8;
        float s = 0.f;
#pragma unroll
        for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (8 * G + 4 * h + q < la.out_width) s += acc[G >> 2][4 * (G & 3) + q];
        const float m = xhalf_sum(s) * la.inv_out_w;
        float qq = 0.f;
#pragma unroll
        for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (8 * G + 4 * h + q < la.out_width) { const float dd = acc[G >> 2][4 * (G & 3) + q] - m; qq = fmaf(dd, dd, qq); }
        qq = xhalf_sum(qq);
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
