// 128-wide ResidualBlocks for LARGE launches, persistent form: one 8-wave workgroup per CU walks its tile groups, the weight
// planes of a block pass through LDS in 32 KiB PANELS that all eight waves read (8 row tiles per weight fetch).
//
// Why (profiles/r02d_*): k_wide128_h (4 waves, an 8 KiB chunk ring, one barrier per two chunks) met its
// workgroup at a barrier 44 times per block, paid a 12 000-cycle prologue and a 4 000-cycle store tail per 4 tiles (19 % of a
// workgroup's life) and sat at 0.29 of the matrix-core peak with the matrix core busy a third of the time.  Here
//   * a workgroup is PERSISTENT: per-feature vectors are staged once per launch, tile groups follow each other with the weight
//     and operand streams running across the seam (no prologue, no drain per group);
//   * a PANEL = 4 k16-steps x 4 out tiles x (hi, lo) = 32 KiB, two buffers: one barrier per panel (12 per up block instead of
//     44), the next panel's LDS-DMA issued right behind the barrier and in flight for a whole panel time;
//   * the PRIVATE operands of a wave (its tile's row statistics, input tensors, condition embedding) travel through per-wave
//     LDS slots (5 x 2 KiB, four items ahead) and need no barrier at all: only the issuing wave reads them;
//   * the request targets are static per call site, so the waits on the vector-memory counter are compile-time constants
//     (private items: vmcnt(6); a panel: the 8 / 2 / 0 ladder of panel_begin) -- no drain to zero inside the loop;
//   * the operand preparation (LayerNorm, SiLU, hi/lo split) of panel p + 1 sits BETWEEN the MFMAs of panel p (panel_pipe):
//     one MFMA per `asm volatile` slot, the VALU pieces pinned between the slots through "+v" operands.
// Arithmetic per element, packed planes, scales and accumulation order are those of k_wide128_h / resblock_body_h.
#pragma once
#include "dsg_wide.hpp"

namespace dsg {

constexpr int kPW = 8;                 // waves per workgroup (two per SIMD), ONE workgroup per CU
constexpr int kPanelU4 = 2048;         // uint4 per weight panel (32 KiB)
constexpr int kPrivU4 = 128;           // uint4 per private item (2 KiB = two 8-feature groups of one row tile)
constexpr int kPrivSlots = 5;          // per wave
constexpr int kPrivDist = 4;           // items in flight ahead of the consumer (a whole V phase: its four items are requested during
                                       // the previous one); the slot refilled is the one read a step earlier
constexpr int kPanelLdsU4 = 2 * kPanelU4 + kPW * kPrivSlots * kPrivU4 + kWideVec / 4;

// Measurement build (-DDSG_CYCLE_STAMPS): every wave of workgroup 0 keeps up to 128 (cycle, tag) stamps in LDS and dumps them at
// the end of the launch (no global traffic inside the pipelined loop); tools/panel_stamps.py reads them.
#ifndef DSG_STAMP_DBG
#define DSG_STAMP_DBG 0
#endif
#ifdef DSG_CYCLE_STAMPS
constexpr int kPanelStampU4 = kPW * 128 / 2;
#define DSG_PSTAMP(tag)                                                                                               \
    do {                                                                                                              \
        if (STAMPED && blockIdx.x == 0 && stamp_k < 128) {                                                           \
            const unsigned long long t_ = __builtin_readcyclecounter();                                               \
            if (lane == 0 && !(DSG_STAMP_DBG & 1)) stamp_lds[wave * 128 + stamp_k] = (t_ << 16) | (unsigned long long)(tag); \
            if (DSG_STAMP_DBG & 1) asm volatile("" :: "s"(t_));                                                       \
            ++stamp_k;                                                                                                \
        }                                                                                                             \
    } while (0)
#else
constexpr int kPanelStampU4 = 0;
#define DSG_PSTAMP(tag) do {} while (0)
#endif

// LDS-DMA with a scalar base and a per-lane byte offset; the instruction offset moves the global AND the LDS address
// (tools/ubench/glds_check.hip).  4 x 1 KiB / 2 x 1 KiB / two 256-B rows (one dword per lane).
__device__ __forceinline__ void glds_quad_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_pair_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// row statistics of a tile: 32 x (mean, M2) = 256 B of each input tensor -> lds_dst, lds_dst + 256
__device__ __forceinline__ void glds_stats_s(unsigned voff4, const void* s0, const void* s1, unsigned lds_dst) {
    unsigned keep;
    const void* s1m = reinterpret_cast<const char*>(s1) - 256;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, %2\n\t"
                 "global_load_lds_dword %1, %3 offset:256\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff4), "s"(s0), "s"(s1m), "s"(lds_dst) : "memory");
}

// The private operand stream of ONE wave: per tile [row statistics | stage 1: KS1 items | condition embedding: 8 items on a
// conditional tile | shortcut: KS1 items, or the residual re-read: 8 items], an item = 2 KiB (two 8-feature groups of the tile).
// The consumer code is unrolled, so every consume site knows at compile time which item sits kPrivDist places further down
// the stream: base pointer of the current (or, for a tile's last four sites, the next) tile + a constant.  The request for it
// is issued right there -- no iterator, no address decode (the first version kept a position and derived every address from
// it: ~230 scalar moves and ~100 branches per four items, more issue time than the arithmetic of the steps they feed).
// A stream that has run out keeps requesting (the same tile again): the operation counts stay regular and nothing reads the
// slots.
struct TilePtrs {
    const char *x0, *x1, *cp;      // this tile of in0, in1 (= in0 without a concat), the condition embedding
    const char *st0, *st1;         // its row statistics
    bool cond;
};
template <bool SCLIN>
__device__ __forceinline__ TilePtrs tile_ptrs(const BlockArgs& a, int g, int wave) {
    const int traw = g * kPW + wave;
    const int tile = traw < a.ntiles ? traw : a.ntiles - 1;
    const int ptile = tile >= a.tiles_per_pass ? tile - a.tiles_per_pass : tile;      // at most two passes
    const int t0 = seg_tile(a.in0, tile), t1 = seg_tile(a.in1, tile);
    TilePtrs t;
    t.x0 = reinterpret_cast<const char*>(a.in0.data + (size_t)t0 * 16 * 256);
    t.st0 = reinterpret_cast<const char*>(a.in0.stats + (size_t)t0 * 64);
    t.x1 = SCLIN ? reinterpret_cast<const char*>(a.in1.data + (size_t)t1 * 16 * 256) : t.x0;
    t.st1 = SCLIN ? reinterpret_cast<const char*>(a.in1.stats + (size_t)t1 * 64) : t.st0;
    t.cp = reinterpret_cast<const char*>(a.cond_pre + (size_t)ptile * 16 * 256);
    t.cond = tile >= a.uncond_tiles;
    return t;
}

struct PanelCtx {
    unsigned lane16, lane4;
    unsigned w_lds;            // LDS byte address of this wave's 4 KiB piece of weight buffer 0 (buffer 1: + 32 KiB)
    unsigned w_rd;             // LDS byte address of weight buffer 0 + lane * 16 (M phase)
    unsigned p_lds;            // LDS byte address of this wave's private slot 0
    const uint4* wrd;          // weight buffer 0 as ordinary LDS, + lane
    const uint4* prd;          // this wave's private slot 0, + lane
    int q;                     // weight panels consumed so far (buffer = q & 1)
    int islot, cslot;          // private slot the next item goes into / the next item is read from (0 .. kPrivSlots - 1)
};

// Counted waits.  The stream is regular: behind the item a consume site waits for, exactly kPrivDist - 1 = 3 younger items
// (2 DMA operations each) have been requested, plus possibly weight panels (4 each).  Vector-memory operations complete in
// order, so "at most 6 outstanding" implies the awaited item has landed -- and never waits for the three items behind it.
// A panel's own wait (panel_x) finds nothing younger than the panel in flight in a memory-fed stage: vmcnt(0).
// request one item (2 KiB at src, or the two statistics rows) into the next slot
__device__ __forceinline__ void priv_issue(PanelCtx& c, const char* src) {
    glds_pair_s(c.lane16, src, c.p_lds + (unsigned)c.islot * 2048u);
    c.islot = c.islot == kPrivSlots - 1 ? 0 : c.islot + 1;
}
__device__ __forceinline__ void priv_issue_stats(PanelCtx& c, const char* st0, const char* st1) {
    glds_stats_s(c.lane4, st0, st1, c.p_lds + (unsigned)c.islot * 2048u);
    c.islot = c.islot == kPrivSlots - 1 ? 0 : c.islot + 1;
}
// Consume the next private item: wait for it and hand back its slot.  The caller then requests the item kPrivDist places ahead
// (it goes into the slot read one step earlier).
__device__ __forceinline__ const uint4* priv_consume(PanelCtx& c) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    const uint4* rd = c.prd + c.cslot * kPrivU4;
    c.cslot = c.cslot == kPrivSlots - 1 ? 0 : c.cslot + 1;
    return rd;
}

template <int NT>
__device__ __forceinline__ void panel_wfrag(HFrag<NT>& w, const uint4* panel /* + lane */, int steps_per_tile, int s) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { w.hi[nt] = panel[(nt * steps_per_tile + s) * 128]; w.lo[nt] = panel[(nt * steps_per_tile + s) * 128 + 64]; }
}


// LayerNorm vectors of a step from LDS: `gl` / `bl` are this lane's address-space-3 pointers (vector + 4 h); with a compile-time S
// the offset folds into the ds_read instruction (through generic pointers hipcc kept four address registers per step alive
// across the whole tile loop and spilled them)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
// lo part of the split: v - float(half of `hi`), one v_fma_mix_f32 (f32 * 1.0 - f16): hipcc picks cvt + sub (two instructions) in
// this context.  Not volatile: an ordinary instruction for the scheduler.
__device__ __forceinline__ float sub_half_lo(float v, unsigned hi) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hi));
    return r;
}
__device__ __forceinline__ float sub_half_hi(float v, unsigned hi) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hi));
    return r;
}

// ASM_SPLIT: the hi/lo split exactly as panel_pipe does it (the lo part from the ROUNDED product through v_fma_mix; in the plain form
// hipcc contracts `u * r - hi` into one fma, which moves the last bit of lo): a tile whose first panel is prepared here on one
// occasion and inside panel_pipe on another must get the same bits (rows do not depend on their position in the batch)
template <bool LNACT, bool ASM_SPLIT = false>
__device__ __forceinline__ BOp panel_prep(const float (&x)[8], const float* gamma, const float* beta, int S, float c, float d, int h) {
    float v[8];
    if (LNACT) {
        lds_cf4* const gl = (lds_cf4*)(gamma + 4 * h);
        lds_cf4* const bl = (lds_cf4*)(beta + 4 * h);
        const f32x4 g0 = gl[4 * S], b0 = bl[4 * S], g1 = gl[4 * S + 2], b1 = bl[4 * S + 2];
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        // stage by stage over the eight values ("vertical"): hipcc otherwise walks the values pair by pair, and every
        // transcendental waits for the one issued just before it (exp -> fma -> rcp -> mul: four dependent stages per pair)
        constexpr float kk = -1.44269504088896341f / kActScale;
        float u[8], p[8], r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = fmaf(fmaf(x[q], c, d), g[q], b[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = __builtin_amdgcn_exp2f(u[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = fmaf(p[q], kk, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = __builtin_amdgcn_rcpf(r[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = u[q] * r[q];
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[q] * kRawScale;
    }
    BOp o;
    if (ASM_SPLIT) {
        unsigned hh[4], ll[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0 = v[2 * q], v1 = v[2 * q + 1];
            asm volatile("" : "+v"(v0), "+v"(v1));          // the rounded products, as values
            const unsigned a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
            hh[q] = a;
            ll[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(sub_half_lo(v0, a), sub_half_hi(v1, a)));
        }
        const uint4 uh = {hh[0], hh[1], hh[2], hh[3]}, ul = {ll[0], ll[1], ll[2], ll[3]};
        o.hi = __builtin_bit_cast(h8, uh); o.lo = __builtin_bit_cast(h8, ul);
        return o;
    }
    split8(v, o.hi, o.lo);
    return o;
}

// ---- V phase / M phase.  A panel's worth of work (4 k16-steps) is split into
//   V: the four B operands are prepared (LayerNorm, SiLU, hi/lo split: pure VALU + the LayerNorm vectors from LDS), and
//   M: 48 MFMAs over the panel with nothing but the panel's plane reads between them (the planes of step s+1 are requested
//      under the MFMAs of step s).
// Why: with the operand preparation of a step sitting between its LDS reads and its MFMAs, every step was a chain of exposed
// latencies (LDS round trip -> dependent VALU chain -> 12 MFMAs each waiting for its just-requested plane): the first
// panel kernel's PMC passes: 36 % of wave cycles in s_waitcnt and 30 % in issue stalls.  The M phase has no dependence on anything but LDS reads issued
// a step ahead; the V phase no MFMA to wait for.  The operands of four steps are 32 registers.
struct BOp4 { h8 hi[4], lo[4]; };
// The operands are COMPLETE here: without this the compiler sinks the whole preparation behind the panel's barrier, next to the
// MFMA stream that consumes it (cycle stamps: the "M phase" of both SIMD partners then contained the V phase's arithmetic)
// ... and the second half of a register-fed stage's input is "redefined" behind the first panel's MFMA stream, so that its
// preparation cannot be hoisted above that stream (32 more live registers exactly where the plane registers are needed)
__device__ __forceinline__ void v_phase_done(BOp4& b) {
    asm volatile("" : "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]), "+v"(b.hi[3]), "+v"(b.lo[0]), "+v"(b.lo[1]), "+v"(b.lo[2]), "+v"(b.lo[3]));
}

// B operands of steps S0..S0+3 of a register-fed stage from the accumulators of the previous one
__device__ __forceinline__ void v_phase_reg(BOp4& b, const f32x16 (&in)[4], int S0, const float* gamma, const float* beta, float cc, float dd, int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int S = S0 + s, t = S >> 1, r0 = 8 * (S & 1);
        const float x[8] = {in[t][r0], in[t][r0 + 1], in[t][r0 + 2], in[t][r0 + 3], in[t][r0 + 4], in[t][r0 + 5], in[t][r0 + 6], in[t][r0 + 7]};
        const BOp o = panel_prep<true>(x, gamma, beta, S, cc, dd, h);
        b.hi[s] = o.hi; b.lo[s] = o.lo;
    }
}

// 48 MFMAs: acc (+)= W[panel] * b, as ONE hand-ordered instruction stream (dsg_panel_mphase.inc, tools/gen/gen_mphase.py): two
// plane register sets alternate by step, the eight plane reads of step s+1 go out one behind each of the first eight MFMAs of
// step s, every MFMA waits (counted lgkmcnt: LDS returns in order) only for the plane it needs.  hipcc, given the same program
// as C++ with sched_group_barrier requests, re-ordered the MFMAs and kept every plane read right in front of its use
// (read -> wait -> MFMA, five exposed LDS round trips per step).
#include "dsg_panel_mphase.inc"
template <bool FIRST>
__device__ __forceinline__ void m_phase(f32x16 (&acc)[4], unsigned panel_addr /* LDS byte address of the panel + lane * 16 */, const BOp4& b) {
    uint4 pl[16];
    if (FIRST)
        asm volatile(DSG_MPHASE_ASM_FIRST
                     : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]),
                       [pl0] "=&v"(pl[0]), [pl1] "=&v"(pl[1]), [pl2] "=&v"(pl[2]), [pl3] "=&v"(pl[3]), [pl4] "=&v"(pl[4]), [pl5] "=&v"(pl[5]),
                       [pl6] "=&v"(pl[6]), [pl7] "=&v"(pl[7]), [pl8] "=&v"(pl[8]), [pl9] "=&v"(pl[9]), [pl10] "=&v"(pl[10]), [pl11] "=&v"(pl[11]),
                       [pl12] "=&v"(pl[12]), [pl13] "=&v"(pl[13]), [pl14] "=&v"(pl[14]), [pl15] "=&v"(pl[15])
                     : [ad] "v"(panel_addr), [bh0] "v"(b.hi[0]), [bh1] "v"(b.hi[1]), [bh2] "v"(b.hi[2]), [bh3] "v"(b.hi[3]),
                       [bl0] "v"(b.lo[0]), [bl1] "v"(b.lo[1]), [bl2] "v"(b.lo[2]), [bl3] "v"(b.lo[3])
                     : "memory");
    else
        asm volatile(DSG_MPHASE_ASM
                     : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]),
                       [pl0] "=&v"(pl[0]), [pl1] "=&v"(pl[1]), [pl2] "=&v"(pl[2]), [pl3] "=&v"(pl[3]), [pl4] "=&v"(pl[4]), [pl5] "=&v"(pl[5]),
                       [pl6] "=&v"(pl[6]), [pl7] "=&v"(pl[7]), [pl8] "=&v"(pl[8]), [pl9] "=&v"(pl[9]), [pl10] "=&v"(pl[10]), [pl11] "=&v"(pl[11]),
                       [pl12] "=&v"(pl[12]), [pl13] "=&v"(pl[13]), [pl14] "=&v"(pl[14]), [pl15] "=&v"(pl[15])
                     : [ad] "v"(panel_addr), [bh0] "v"(b.hi[0]), [bh1] "v"(b.hi[1]), [bh2] "v"(b.hi[2]), [bh3] "v"(b.hi[3]),
                       [bl0] "v"(b.lo[0]), [bl1] "v"(b.lo[1]), [bl2] "v"(b.lo[2]), [bl3] "v"(b.lo[3])
                     : "memory");
}

// ---- One MFMA per `asm volatile` slot (panel_pipe below): volatile statements keep their order, the plane reads are ordinary LDS
// loads placed BETWEEN the statements (a load cannot cross a volatile statement, so it stays in the slot it was written in; hipcc counts
// its lgkmcnt waits itself), and ordinary VALU code is pinned between two MFMAs by passing its inputs / results through the
// neighbouring statements as "+v" operands.
constexpr int kPlaneAhead = 8;     // MFMA slots between a plane's read and its use

// slot k of a panel: term t = (k / 4) % 3 (hi*bhi, hi*blo, lo*bhi), out tile nt = k % 4, step s = k / 12
__device__ __forceinline__ constexpr int slot_plane(int k) { return (((k % 4) * 4 + k / 12) * 2 + ((k / 4) % 3 == 2 ? 1 : 0)) * 64; }   // uint4 offset in the panel

// EPI: 0 = block only, 1 = + raw Linear (Down/Upsample), 2 = + final (LayerNorm + SiLU + Linear, row-major out)
// ---- panel_pipe: the MFMA stream of one panel WITH the operand preparation of the next panel between the MFMAs.
// 48 MFMA slots, 16 operand pairs (4 steps x 4 pairs of values): pair m takes slots 3m (LayerNorm + exp2), 3m + 1 (rcp, product)
// and 3m + 2 (hi/lo split), six / six / four VALU instructions, i.e. the five-or-so instructions a wave can issue under one
// MFMA (MI355X guide, "instructions hidden per MFMA gap").  The pieces are ordinary C++; what keeps each of them between its two
// MFMAs is that its inputs and its results pass through the neighbouring `asm volatile` statements as "+v" operands.

__device__ __forceinline__ void mfma_pin(f32x16& c, const uint4 a, const h8 b, float& p0, float& p1, float& p2, float& p3) {
    asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]" : [c] "+v"(c), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : [a] "v"(__builtin_bit_cast(h8, a)), [b] "v"(b));
}
__device__ __forceinline__ void mfma_pin0(f32x16& c, const uint4 a, const h8 b, float& p0, float& p1, float& p2, float& p3) {
    asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], 0" : [c] "=&v"(c), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : [a] "v"(__builtin_bit_cast(h8, a)), [b] "v"(b));
}

// PREP: what is prepared under this panel's MFMAs --
//   PIPE_REG   the next four steps (S0 .. S0 + 3) of a register-fed stage, from the accumulators `in` (LayerNorm + SiLU + split)
//   PIPE_LN    four steps whose input arrives through the wave's private slots (stage 1): LayerNorm + SiLU + split
//   PIPE_RAW   the same source, split only (Linear shortcut)
// For the two private-slot kinds `consume(j)` is called once per step, six MFMA slots before the step's first piece: it waits
// for the item, requests the one kPrivDist places ahead and returns the slot.
enum { PIPE_REG = 1, PIPE_LN = 2, PIPE_RAW = 3 };

template <bool FIRST, int PREP, typename Consume>
__device__ __forceinline__ void panel_pipe(f32x16 (&acc)[4], const uint4* pn /* + lane */, const BOp4& b, BOp4& bn, const f32x16 (&in)[4], int S0,
                                           const float* gamma, const float* beta, float cc, float dd, int h, Consume&& consume) {
    constexpr float kk = -1.44269504088896341f / kActScale;
    constexpr bool LN = PREP != PIPE_RAW, MEM = PREP != PIPE_REG;
    lds_cf4* const gl = (lds_cf4*)(gamma + 4 * h);
    lds_cf4* const bl = (lds_cf4*)(beta + 4 * h);
    uint4 pl[48];
    float cpin = cc, u0 = 0.f, u1 = 0.f, p0 = 0.f, p1 = 0.f, v0 = 0.f, v1 = 0.f;
    f32x4 gq[8], bq[8], xq[8];                       // per HALF step (two pairs): LayerNorm vectors, input values
    float hq[17], lq[17];                            // results (bit patterns of half pairs); [16] is a dummy pin for the first statement
    lds_cf4* slot[4];                                // private slot of each of the four steps (+ lane)
    hq[16] = 0.f; lq[16] = 0.f;
    auto load_vec = [&](int hs) {                    // half step hs = 2 * step + (0: values 0-3, 1: values 4-7)
        const int S = S0 + (hs >> 1);
        gq[hs] = gl[4 * S + 2 * (hs & 1)]; bq[hs] = bl[4 * S + 2 * (hs & 1)];
    };
    auto load_x = [&](int hs) { xq[hs] = slot[hs >> 1][(hs & 1) * 64]; };
#pragma unroll
    for (int k = 0; k < kPlaneAhead; ++k)
        if ((k / 4) % 3 != 1) pl[k] = pn[slot_plane(k)];
    if (MEM) { slot[0] = (lds_cf4*)consume(0); load_x(0); }
    if (LN) load_vec(0);
#pragma unroll
    for (int k = 0; k < 48; ++k) {
        const int s = k / 12, t = (k / 4) % 3, nt = k % 4;
        const int kp = t == 1 ? k - 4 : k;
        const h8 bb = t == 1 ? b.lo[s] : b.hi[s];
        const int m = k / 3, ph = k % 3, mp = m == 0 ? 16 : m - 1;
        // private item of the NEXT step: six slots before its first piece
        if (MEM && k % 12 == 6 && k / 12 < 3) slot[k / 12 + 1] = (lds_cf4*)consume(k / 12 + 1);
        // the statement: MFMA k; pins = what flows from the piece behind the previous MFMA into the piece behind this one
        if (ph == 0) { if (FIRST && k < 4) mfma_pin0(acc[nt], pl[kp], bb, cpin, hq[mp], lq[mp], v0); else mfma_pin(acc[nt], pl[kp], bb, cpin, hq[mp], lq[mp], v0); }
        else if (ph == 1) { if (FIRST && k < 4) mfma_pin0(acc[nt], pl[kp], bb, u0, u1, p0, p1); else mfma_pin(acc[nt], pl[kp], bb, u0, u1, p0, p1); }
        else { if (FIRST && k < 4) mfma_pin0(acc[nt], pl[kp], bb, v0, v1, u0, u1); else mfma_pin(acc[nt], pl[kp], bb, v0, v1, u0, u1); }
        // loads of later slots: planes kPlaneAhead slots ahead; vectors / values of the next half step at the first pair of this one
        const int kn = k + kPlaneAhead;
        if (kn < 48 && (kn / 4) % 3 != 1) pl[kn] = pn[slot_plane(kn)];
        if (ph == 0 && (m & 1) == 0 && m / 2 + 1 < 8) {
            if (LN) load_vec(m / 2 + 1);
            if (MEM) load_x(m / 2 + 1);
        }
        // the piece behind MFMA k
        const int S = S0 + (m >> 2), q = m & 3, tt = (S >> 1) & 3, r0 = 8 * (S & 1) + 2 * q, hs = m >> 1, e = 2 * (m & 1);
        if (ph == 0) {
            const float x0 = MEM ? xq[hs][e] : in[tt][r0], x1 = MEM ? xq[hs][e + 1] : in[tt][r0 + 1];
            if (LN) {
                u0 = fmaf(fmaf(x0, cpin, dd), gq[hs][e], bq[hs][e]);
                u1 = fmaf(fmaf(x1, cpin, dd), gq[hs][e + 1], bq[hs][e + 1]);
                p0 = __builtin_amdgcn_exp2f(u0); p1 = __builtin_amdgcn_exp2f(u1);
            } else {
                u0 = x0 * (cpin * kRawScale); u1 = x1 * (cpin * kRawScale);      // cpin == 1: keeps the piece behind this MFMA
            }
        } else if (ph == 1) {
            if (LN) {
                v0 = u0 * __builtin_amdgcn_rcpf(fmaf(p0, kk, kk));
                v1 = u1 * __builtin_amdgcn_rcpf(fmaf(p1, kk, kk));
            } else { v0 = u0; v1 = u1; }
        } else {
            const unsigned a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0, v1));
            const unsigned er = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(sub_half_lo(v0, a), sub_half_hi(v1, a)));
            hq[m] = __builtin_bit_cast(float, a); lq[m] = __builtin_bit_cast(float, er);
        }
    }
    // the last piece's results are complete here
    asm volatile("" : "+v"(hq[15]), "+v"(lq[15]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 hv = {hq[4 * j], hq[4 * j + 1], hq[4 * j + 2], hq[4 * j + 3]}, lv = {lq[4 * j], lq[4 * j + 1], lq[4 * j + 2], lq[4 * j + 3]};
        bn.hi[j] = __builtin_bit_cast(h8, hv); bn.lo[j] = __builtin_bit_cast(h8, lv);
    }
}

#ifndef DSG_PANEL_TRAIL_PRIO
#define DSG_PANEL_TRAIL_PRIO 1
#endif
template <bool FIRST>
__device__ __forceinline__ void M_PHASE(f32x16 (&acc)[4], unsigned pa, const BOp4& b, const uint4* lds_lane) {
    m_phase<FIRST>(acc, pa, b);
}
template <bool SCLIN, int EPI, int NTO>
__global__ __launch_bounds__(512, 2) void k_panel128_h(const BlockLinArgsH A, const int ngroups) {
    constexpr int N = 128, NT = 4, NG = 16;
    constexpr int KS1 = SCLIN ? 16 : 8, P1 = KS1 / 4, NE = SCLIN ? 16 : 8;   // k16-steps of stage 1; items of the shortcut / residual read
    constexpr int NTOP = NTO <= 1 ? 1 : (NTO == 2 ? 2 : 4), ESTEPS = 16 / NTOP;    // epilogue Linear: k16-steps per panel (8 needed)
    constexpr int EPANELS = EPI == 0 ? 0 : (ESTEPS >= 8 ? 1 : 2);
    constexpr int PA = P1, PB = PA + 2, PD = PB + 2, PE = PD + (SCLIN ? P1 : 0), NP = PE + EPANELS;
    __shared__ uint4 lds[kPanelLdsU4 + kPanelStampU4];
#ifdef DSG_CYCLE_STAMPS
    unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + kPanelLdsU4);
    int stamp_k = 0;
    constexpr bool STAMPED = SCLIN && EPI == 0;
#endif
    float* const vec = reinterpret_cast<float*>(lds + 2 * kPanelU4 + kPW * kPrivSlots * kPrivU4);
    float* const g1v = vec, * const b1v = vec + kLnLdsW1, * const v2 = vec + 2 * kLnLdsW1;
    float* const g2v = v2, * const b2v = v2 + 128, * const g3v = v2 + 256, * const b3v = v2 + 384, * const tbv = v2 + 512, * const c2v = v2 + 640,
         * const c3v = v2 + 768, * const gLv = v2 + 896, * const bLv = v2 + 1024, * const biasLv = v2 + 1152;
    const BlockArgsH& ah = A.b;
    const BlockArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = gridDim.x;
    // Waves w and w + 4 share a SIMD.  Both run the same program (every panel's MFMA stream carries the operand preparation of
    // the next panel between its MFMAs, panel_pipe); the younger half gets a static priority: at equal priority the older wave
    // of a SIMD wins every arbitration, finishes a panel ~1 400 cycles before its partner and idles at the barrier while the
    // partner runs alone (cycle stamps: 3 000 against 4 400 cycles per LayerNorm panel).  MI355X guide, "two waves per SIMD".
    const bool lead = wave < 4;
    if (!lead) __builtin_amdgcn_s_setprio(DSG_PANEL_TRAIL_PRIO);
    constexpr float kL2 = -1.44269504088896341f;

    // ---- per-feature vectors -> LDS, once per launch (LayerNorm vectors times -log2 e)
    {
        const int n1 = ln1_extent(a);
        for (int i = threadIdx.x; i < n1; i += 512) { g1v[i] = a.gamma1[i] * kL2; b1v[i] = a.beta1[i] * kL2; }
        if (threadIdx.x < 128) {
            const int i = threadIdx.x;
            g2v[i] = a.gamma2[i] * kL2; b2v[i] = a.beta2[i] * kL2; g3v[i] = a.gamma3[i] * kL2; b3v[i] = a.beta3[i] * kL2;
            c2v[i] = a.c2[i]; c3v[i] = a.c3[i];
            tbv[i] = a.tbias[(size_t)(a.step_ptr ? *a.step_ptr : 0) * a.tb_stride + i];
            if (EPI == 2) { gLv[i] = A.l.l.gamma[i] * kL2; bLv[i] = A.l.l.beta[i] * kL2; }
            if (EPI != 0) biasLv[i] = i < NTO * 32 ? A.l.l.bias[i] : 0.f;
        }
    }
    const float inv1 = ah.kc[0], inv2 = ah.kc[1], inv3 = ah.kc[2];
    float invL = 0.f;
    if (EPI != 0) invL = A.l.kc[EPI == 2 ? 1 : 0];
    __syncthreads();                   // no DMA is in flight yet: a plain barrier

    // ---- this wave's 4 KiB of every weight panel: (out tile, half of its four k16-steps)
    const int wnt = wave >> 1, wsub = wave & 1;
    const uint4* const w1p = ah.W1h + ((size_t)wnt * KS1) * 128 + wsub * 256;
    const uint4* const w2p = ah.W2h + ((size_t)wnt * 8) * 128 + wsub * 256;
    const uint4* const w3p = ah.W3h + ((size_t)wnt * 8) * 128 + wsub * 256;
    const uint4* const wsp = SCLIN ? ah.Wsch + ((size_t)wnt * KS1) * 128 + wsub * 256 : w1p;
    const uint4* wlp = w1p;
    if (EPI != 0) {
        // NTOP = 4: as the block's panels (tiles >= NTO reload tile 0); 2: (tile, quarter of its 8 steps); 1: quarter of tile 0
        if (NTOP == 4) wlp = A.l.Wh + ((size_t)(wnt < NTO ? wnt : 0) * 8) * 128 + wsub * 256;
        else if (NTOP == 2) wlp = A.l.Wh + ((size_t)(wave >> 2) * 8) * 128 + (wave & 3) * 256;
        else wlp = A.l.Wh + (wave & 3) * 256;
    }
    // panel p of the block program (compile-time p at every call site): the weights do not depend on the tile
    auto panel_src = [&](int p) -> const uint4* {
        if (p < PA) return w1p + (size_t)p * 512;
        if (p < PB) return w2p + (size_t)(p - PA) * 512;
        if (p < PD) return w3p + (size_t)(p - PB) * 512;
        if (p < PE) return wsp + (size_t)(p - PD) * 512;
        return wlp + (NTOP == 4 ? (size_t)(p - PE) * 512 : 0);
    };

    const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds;
    PanelCtx c;
    c.lane16 = (unsigned)lane * 16u; c.lane4 = (unsigned)lane * 4u;
    c.w_lds = lds0 + (unsigned)wave * 4096u;
    c.w_rd = lds0 + (unsigned)lane * 16u;
    c.p_lds = lds0 + 2u * kPanelU4 * 16u + (unsigned)wave * (kPrivSlots * 2048u);
    c.wrd = lds + lane;
    c.prd = lds + 2 * kPanelU4 + wave * (kPrivSlots * kPrivU4) + lane;
    c.q = 0; c.islot = 0; c.cslot = 0;

    // prologue: panel 0 into buffer 0 and the first four private items of this wave's first tile
    {
        const TilePtrs cur = tile_ptrs<SCLIN>(a, blockIdx.x < ngroups ? blockIdx.x : 0, wave);
        priv_issue_stats(c, cur.st0, cur.st1);
        priv_issue(c, cur.x0);
        priv_issue(c, cur.x0 + 2048);
        priv_issue(c, cur.x0 + 4096);
    }
    glds_quad_s(c.lane16, panel_src(0), c.w_lds);

    // One barrier per panel: my pieces of the panel have landed (counted wait: `since_w` DMA operations of the private stream
    // went out behind them); everyone is done with the other buffer; refill that one with the panel after this (`next`: the block
    // program repeats for the next tile group; the request behind the launch's last panel is redundant and drained at the end).
    int since_w = 8;                   // the four private requests of the prologue
    auto panel_begin = [&](int next) -> int {
        if (since_w >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (since_w >= 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int rd = c.q & 1;
        glds_quad_s(c.lane16, panel_src(next), c.w_lds + (unsigned)((c.q + 1) & 1) * (kPanelU4 * 16u));
        since_w = 0;
        ++c.q;
        return rd;
    };
    // sources of the concatenated input by k16-step (stage 1 and the shortcut / residual read the same tensors)
    auto xin = [&](const TilePtrs& t, int S) -> const char* { return SCLIN && S >= 8 ? t.x1 + (size_t)(S - 8) * 2048 : t.x0 + (size_t)S * 2048; };

    // XT (up block without an epilogue Linear): the first panel of a tile's stage 1 is prepared under the MFMAs of the PREVIOUS tile's
    // last shortcut panel (which has nothing of its own to prepare) instead of on its own -- 5 000 vector-only cycles per tile during
    // which the eight waves of the CU, in lock-step, all left the matrix cores idle.  Carried across the loop: the operands and the
    // LayerNorm constants of the coming tile.
    constexpr bool XT = SCLIN && EPI == 0;
    BOp4 bcar;
    float cc_car = 0.f, dd_car = 0.f;
    for (int g = blockIdx.x; g < ngroups; g += stride) {
        const int tile_raw = g * kPW + wave;
        const bool live = tile_raw < a.ntiles;          // idle waves of the last group still move their pieces and meet the barriers
        const int tile = live ? tile_raw : a.ntiles - 1;
        const int ptile = tile >= a.tiles_per_pass ? tile - a.tiles_per_pass : tile;
        const TilePtrs cur = tile_ptrs<SCLIN>(a, g, wave);
        const bool my_cond = cur.cond;
        const TilePtrs nxt = tile_ptrs<SCLIN>(a, g + stride < ngroups ? g + stride : g, wave);
        DSG_PSTAMP(0x01);
        // the item four places behind position `T` of the part that follows stage 1: condition embedding (conditional tiles),
        // then the shortcut / residual input, then the next tile's statistics and first steps
        auto issue_tail = [&](int T) {            // T counts from the first shortcut / residual item
            if (T < NE) priv_issue(c, xin(cur, T));
            else if (T == NE) priv_issue_stats(c, nxt.st0, nxt.st1);
            else priv_issue(c, nxt.x0 + (size_t)(T - NE - 1) * 2048);
            since_w += 2;
        };
        // stage 1, step S: wait for its item, request the one four places ahead
        auto consume_s1 = [&](int S) -> const uint4* {
            const uint4* rd = priv_consume(c);
            if (S + 4 < KS1) { priv_issue(c, xin(cur, S + 4)); since_w += 2; }
            else if (my_cond) { priv_issue(c, cur.cp + (size_t)(S + 4 - KS1) * 2048); since_w += 2; }
            else issue_tail(S + 4 - KS1);
            return rd;
        };
        auto consume_sc = [&](int S) -> const uint4* {        // shortcut, step S
            const uint4* rd = priv_consume(c);
            issue_tail(S + 4);
            return rd;
        };
        auto no_consume = [&](int) -> const uint4* { return nullptr; };

        // ---- LN1 statistics (Chan merge of the producers' (mean, M2)), as resblock_body_h; `t` = the tile they belong to
        auto ln1_stats = [&](const TilePtrs& t, float& cc, float& dd) {
            const uint4* rd = priv_consume(c);
            const float2* sp = reinterpret_cast<const float2*>(rd - lane);     // slot base
            const float2 s0 = sp[j];
            float mean = s0.x, m2 = s0.y;
            if (SCLIN) {
                const float2 s1 = sp[32 + j];
                const float dd_ = s1.x - mean;
                m2 = m2 + s1.y + dd_ * dd_ * a.chan_w;
                mean = mean + dd_ * a.chan_f;
            }
            priv_issue(c, t.x0 + 3 * 2048); since_w += 2;
            const float rstd = rsqrtf(m2 * a.inv_nin + kLnEps);
            if (SCLIN) range_check(a.range_flag, mean, m2);
            cc = rstd; dd = -mean * rstd;
        };

        // ---- stage 1: memory-fed, LayerNorm + SiLU.  Panel 0's operands are prepared on their own (first tile of the workgroup; every
        // tile where !XT) or arrive from the previous tile's last panel; the others under the MFMAs of the panel before them.
        f32x16 acc1[NT];
        {
            float cc, dd;
            BOp4 b;
            if (!XT || g == (int)blockIdx.x) {
                ln1_stats(cur, cc, dd);
                DSG_PSTAMP(0x10);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const uint4* rd = consume_s1(s4);
                    const float4 xa = __builtin_bit_cast(float4, rd[0]), xb = __builtin_bit_cast(float4, rd[64]);
                    const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                    const BOp o = panel_prep<true, XT>(x, g1v, b1v, s4, cc, dd, h);
                    b.hi[s4] = o.hi; b.lo[s4] = o.lo;
                }
                v_phase_done(b);
            } else {
                b = bcar; cc = cc_car; dd = dd_car;
            }
            DSG_PSTAMP(0x11);
#pragma unroll
            for (int p = 0; p < P1; ++p) {
                const int bi = panel_begin(p + 1);
                DSG_PSTAMP(0x14);
                if (p + 1 < P1) {
                    BOp4 bn;
                    if (p == 0) panel_pipe<true, PIPE_LN>(acc1, c.wrd + bi * kPanelU4, b, bn, acc1, 4 * (p + 1), g1v, b1v, cc, dd, h, [&](int jj) { return consume_s1(4 * (p + 1) + jj); });
                    else panel_pipe<false, PIPE_LN>(acc1, c.wrd + bi * kPanelU4, b, bn, acc1, 4 * (p + 1), g1v, b1v, cc, dd, h, [&](int jj) { return consume_s1(4 * (p + 1) + jj); });
                    b = bn;
                } else {
                    const unsigned pa = c.w_rd + (unsigned)bi * (kPanelU4 * 16u);
                    if (p == 0) M_PHASE<true>(acc1, pa, b, c.wrd); else M_PHASE<false>(acc1, pa, b, c.wrd);
                }
                DSG_PSTAMP(0x12);
            }
        }
        acc_unscale_add_lds<NT>(acc1, inv1, tbv, h);
        DSG_PSTAMP(0x13);

        // ---- stage 2: register-fed
        f32x16 acc2[NT];
        {
            float mean, m2;
            acc_stats<N, NT>(acc1, h, mean, m2);
            const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), cc = rstd, dd = -mean * rstd;
            BOp4 b, bn;
            DSG_PSTAMP(0x20);
            v_phase_reg(b, acc1, 0, g2v, b2v, cc, dd, h);
            v_phase_done(b);
            DSG_PSTAMP(0x21);
            int bi = panel_begin(PA + 1);
            DSG_PSTAMP(0x24);
            panel_pipe<true, PIPE_REG>(acc2, c.wrd + bi * kPanelU4, b, bn, acc1, 4, g2v, b2v, cc, dd, h, no_consume);
            DSG_PSTAMP(0x22);
            bi = panel_begin(PB);
            DSG_PSTAMP(0x24);
            M_PHASE<false>(acc2, c.w_rd + (unsigned)bi * (kPanelU4 * 16u), bn, c.wrd);
            DSG_PSTAMP(0x22);
            acc_unscale_add_lds<NT>(acc2, inv2, c2v, h);
        }
        if (my_cond) {                 // condition embedding of this tile: 8 private items, one accumulator tile per two
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int i = 2 * e + hf;
                    const uint4* rd = priv_consume(c);
                    const float4 cv0 = __builtin_bit_cast(float4, rd[0]), cv1 = __builtin_bit_cast(float4, rd[64]);
                    if (i + 4 < 8) { priv_issue(c, cur.cp + (size_t)(i + 4) * 2048); since_w += 2; }
                    else issue_tail(i + 4 - 8);
                    const int r0 = 8 * hf;
                    acc2[e][r0 + 0] += cv0.x; acc2[e][r0 + 1] += cv0.y; acc2[e][r0 + 2] += cv0.z; acc2[e][r0 + 3] += cv0.w;
                    acc2[e][r0 + 4] += cv1.x; acc2[e][r0 + 5] += cv1.y; acc2[e][r0 + 6] += cv1.z; acc2[e][r0 + 7] += cv1.w;
                }
            }
        }
        DSG_PSTAMP(0x23);

        // ---- stage 3 (+ shortcut in the same scaled accumulator).  The shortcut's operands (raw input, split only) are prepared
        // under the MFMAs of the panel before them, starting with stage 3's second panel.
        f32x16 (&acc3)[NT] = acc1;
        {
            float mean, m2;
            acc_stats<N, NT>(acc2, h, mean, m2);
            const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps), cc = rstd, dd = -mean * rstd;
            BOp4 b, bn;
            DSG_PSTAMP(0x30);
            v_phase_reg(b, acc2, 0, g3v, b3v, cc, dd, h);
            v_phase_done(b);
            DSG_PSTAMP(0x31);
            int bi = panel_begin(PB + 1);
            DSG_PSTAMP(0x34);
            panel_pipe<true, PIPE_REG>(acc3, c.wrd + bi * kPanelU4, b, bn, acc2, 4, g3v, b3v, cc, dd, h, no_consume);
            DSG_PSTAMP(0x32);
            bi = panel_begin((PD) % NP);
            DSG_PSTAMP(0x34);
            if (SCLIN) {
                panel_pipe<false, PIPE_RAW>(acc3, c.wrd + bi * kPanelU4, bn, b, acc3, 0, g3v, b3v, 1.f, 0.f, h, [&](int jj) { return consume_sc(jj); });
                DSG_PSTAMP(0x32);
#pragma unroll
                for (int p = 0; p < P1; ++p) {
                    bi = panel_begin((PD + p + 1) % NP);
                    DSG_PSTAMP(0x44);
                    if (p + 1 < P1) {
                        panel_pipe<false, PIPE_RAW>(acc3, c.wrd + bi * kPanelU4, b, bn, acc3, 0, g3v, b3v, 1.f, 0.f, h, [&](int jj) { return consume_sc(4 * (p + 1) + jj); });
                        b = bn;
                    } else if (XT) {
                        // the coming tile (the same one again behind the workgroup's last group: its requests are in flight either way)
                        ln1_stats(nxt, cc_car, dd_car);
                        panel_pipe<false, PIPE_LN>(acc3, c.wrd + bi * kPanelU4, b, bcar, acc3, 0, g1v, b1v, cc_car, dd_car, h, [&](int jj) {
                            const uint4* rd = priv_consume(c);
                            priv_issue(c, xin(nxt, jj + 4)); since_w += 2;
                            return rd;
                        });
                    } else {
                        M_PHASE<false>(acc3, c.w_rd + (unsigned)bi * (kPanelU4 * 16u), b, c.wrd);
                    }
                    DSG_PSTAMP(0x42);
                }
                acc_unscale_add_lds<NT>(acc3, inv3, c3v, h);
            } else {
                M_PHASE<false>(acc3, c.w_rd + (unsigned)bi * (kPanelU4 * 16u), bn, c.wrd);
                DSG_PSTAMP(0x32);
                acc_unscale_add_lds<NT>(acc3, inv3, c3v, h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * e + hf;
                        const uint4* rd = priv_consume(c);
                        const float4 xv0 = __builtin_bit_cast(float4, rd[0]), xv1 = __builtin_bit_cast(float4, rd[64]);
                        issue_tail(i + 4);
                        const int r0 = 8 * hf;
                        acc3[e][r0 + 0] += xv0.x; acc3[e][r0 + 1] += xv0.y; acc3[e][r0 + 2] += xv0.z; acc3[e][r0 + 3] += xv0.w;
                        acc3[e][r0 + 4] += xv1.x; acc3[e][r0 + 5] += xv1.y; acc3[e][r0 + 6] += xv1.z; acc3[e][r0 + 7] += xv1.w;
                    }
                }
            }
        }
        DSG_PSTAMP(0x43);

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
        DSG_PSTAMP(0x50);
        if (EPI == 0) continue;
        if (EPI == 1) range_check(a.range_flag, xmean, xm2);

        // ---- epilogue Linear: a panel holds ESTEPS k16-steps x NTOP out tiles
        const LinArgs& la = A.l.l;
        f32x16 acc[NTO];
        {
            const float cc = EPI == 2 ? rsqrtf(xm2 * la.inv_in_w + kLnEps) : 1.f, dd = -xmean * cc;
            const uint4* pn = nullptr;
#pragma unroll
            for (int S = 0; S < 8; ++S) {
                if (S % ESTEPS == 0) pn = c.wrd + panel_begin((PE + S / ESTEPS + 1) % NP) * kPanelU4;
                const int sl = S % ESTEPS, t = S >> 1, r0 = 8 * (S & 1);
                HFrag<NTO> w;
                panel_wfrag<NTO>(w, pn, ESTEPS, sl);
                const float x[8] = {acc3[t][r0], acc3[t][r0 + 1], acc3[t][r0 + 2], acc3[t][r0 + 3], acc3[t][r0 + 4], acc3[t][r0 + 5], acc3[t][r0 + 6],
                                    acc3[t][r0 + 7]};
                float v[8];
                if (EPI == 2) {
                    const float4 g0 = ld4(gLv + 16 * S + 4 * h), b0 = ld4(bLv + 16 * S + 4 * h);
                    const float4 g1 = ld4(gLv + 16 * S + 8 + 4 * h), b1 = ld4(bLv + 16 * S + 8 + 4 * h);
                    act8_l2(v, x, cc, dd, g0, b0, g1, b1);
                } else {
#pragma unroll
                    for (int qq = 0; qq < 8; qq += 2) { const f32x2 tt = f32x2{x[qq], x[qq + 1]} * pk2(kRawScale); v[qq] = tt.x; v[qq + 1] = tt.y; }
                }
                h8 bhi, blo;
                split8(v, bhi, blo);
                if (S == 0) mfma_step_h0<NTO>(acc, w, bhi, blo); else mfma_step_h<NTO>(acc, w, bhi, blo);
            }
        }
        acc_unscale_add_lds<NTO>(acc, invL, biasLv, h);
        if (!live) continue;
        if (EPI == 1) {
            const int NGo = (la.out_width + 7) / 8;
            float s = 0.f;
#pragma unroll
            for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    if (8 * G + 4 * h + qq < la.out_width) s += acc[G >> 2][4 * (G & 3) + qq];
            const float m = xhalf_sum(s) * la.inv_out_w;
            float sq = 0.f;
#pragma unroll
            for (int G = 0; G < NTO * 4; ++G)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    if (8 * G + 4 * h + qq < la.out_width) { const float dv = acc[G >> 2][4 * (G & 3) + qq] - m; sq = fmaf(dv, dv, sq); }
            sq = xhalf_sum(sq);
            if (h == 0) reinterpret_cast<float2*>(la.out_stats)[(size_t)tile * 32 + j] = make_float2(m, sq);
#pragma unroll
            for (int G = 0; G < NTO * 4; ++G)
                if (G < NGo)
                    st4(la.out + ((size_t)tile * NGo + G) * 256 + lane * 4,
                        make_float4(acc[G >> 2][4 * (G & 3)], acc[G >> 2][4 * (G & 3) + 1], acc[G >> 2][4 * (G & 3) + 2], acc[G >> 2][4 * (G & 3) + 3]));
        } else {
            const int pass = tile >= la.tiles_per_pass ? 1 : 0, row = ptile * 32 + j;
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
                        for (int qq = 0; qq < 4; ++qq) {
                            const int f = 8 * G + 4 * h + qq;
                            if (f < la.out_width) o[f] = acc[G >> 2][4 * (G & 3) + qq];
                        }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stream's last requests are redundant: nothing may land in LDS after the end
#ifdef DSG_CYCLE_STAMPS
    if (STAMPED && blockIdx.x == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = lane; k < 128; k += 64) dsg_stamp_buf[wave * 128 + k] = k < stamp_k ? stamp_lds[wave * 128 + k] : 0ull;
        if (threadIdx.x == 0) dsg_stamp_n = kPW * 128;
    }
#endif
}

}  // namespace dsg
