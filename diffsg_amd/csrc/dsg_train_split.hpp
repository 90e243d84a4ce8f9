// Training kernels on the split-f16 matrix-core path (gfx950): weight gradients as hi/lo half GEMMs with float32
// accumulation, 22-bit operands, dynamic power-of-two scaling of the gradient operand.
//
// Reference: torch autograd of every nn.Linear in UNet1D (UNetCF.py:318-356): dW[n][k] = sum_rows G[row][n] * A[row][k].
//
// k_wgrad_h replaces k_wgrad (dsg_train.hpp) when the handle runs in DSG_PRECISION_SPLIT_F16.  Same descriptors, units and
// slabs; what differs is the LDS image (two half planes per operand, [feature][row]) and the MFMA
// (v_mfma_f32_32x32x16_f16: i = out feature, j = in feature, k = 16 batch rows; 3 products hi*hi + hi*lo + lo*hi).
//   G  is scaled by 2^e, e chosen on the device from max|G| of the tensor (tracked by k_colsum, which runs first), so that
//      the largest element sits in [2^13, 2^14): no overflow whatever the loss scale, full 22 bits for everything
//      within 2^-14 of the maximum.
//   A  is scaled like the forward operands: kActScale for LayerNorm+SiLU outputs, kRawScale for raw tensors, 1 for one-hot.
#pragma once
#include "dsg_train.hpp"
#include "dsg_split.hpp"

namespace dsg {

constexpr int kWhTarget = 13;
constexpr int kWhLdsBytes = 43008;      // the one-out-tile form's images + vectors (wgrad_unit_h2); the wide form needs 33 792 B

__device__ __forceinline__ int wgrad_gexp(unsigned maxbits) {
    const int be = (int)((maxbits >> 23) & 0xff);              // biased exponent of max|G|; 0 = zero or denormal
    if (be == 0) return 0;
    const int e = kWhTarget - (be - 127);
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// ---- Units with 2 or 4 out tiles (NTp): one 32-row tile per barrier interval.  The kernel is bound by the number of instructions a wave issues per
// 32-row interval, not by the matrix cores (measured with the pieces switched off one at a time, profiles/r04_wgrad_switches.txt: MFMAs
// 5 %, operand transforms 37 %, loop skeleton and addressing 35 %), so this form issues about half of them:
//   * the image is ROW-major, [32 batch rows][128 features] f16 per plane, 256-byte rows with the 16-byte chunks XOR-swizzled
//     (cdna_hip_programming.md T10, image (b)): a lane of the fragment layout stores its four features of its row as ONE 8-byte
//     write per plane -- no lane exchange, no byte permutes -- and the MFMA operand (feature i, rows 16 s + 8 h + 0..7) is two
//     transposing reads (ds_read_b64_tr_b16) per plane, conflict-free;
//   * everything that does not depend on the tile is computed once per unit: item sources (base + tile * stride), write
//     offsets, read addresses (the k16-step and the plane are instruction offsets), LayerNorm vectors (staged in LDS);
//   * groups that belong to the image but hold nothing (padding of N or K to a tile) are zeroed once, not per interval.
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v lds_s4;
__device__ __forceinline__ unsigned wimg_off(unsigned row, unsigned ch) { return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u))); }
__device__ __forceinline__ h8 wimg_tr8(const char* p0, const char* p1) {
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)const_cast<char*>(p0));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)const_cast<char*>(p1));
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(h8, r);
}
constexpr unsigned kWimgGH = 0, kWimgGL = 8192, kWimgAH = 16384, kWimgAL = 24576, kWimgVec = 32768;   // byte offsets of the planes, + 1 KiB of vectors

struct WgradRegs { float4 g0[4], av[4]; float2 ms; int oh; };   // one interval's raw operands of a wave (k_wgrad_h, TB == 1 form)

// HASG1: the descriptor has a second gradient tensor; ONEHOT: its A operand is the time-table one-hot (nothing to read).  Compile time:
// as run-time cases behind unconditional redundant reads they cost a third (G1) / a half (one-hot) more L1 traffic.
template <bool HASG1, bool ONEHOT>
__device__ __forceinline__ void wgrad_unit_h1(const WgradDesc& d, const WgradUnit& un, char* __restrict__ img, f32x16 (&acc)[4], float gscale,
                                              int ntiles, int nchunks, int wave, int lane, int NTp, int KT, int g_lo, int ngr, int my_nt,
                                              int my_part, int rsplit) {
    const int h = lane >> 5, j = lane & 31;
    const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    const int t_lo = un.chunk * tiles_per_chunk;
    const int t_hi = (t_lo + tiles_per_chunk < ntiles) ? t_lo + tiles_per_chunk : ntiles;
    const int ngg = NTp * 4, nag = KT * 4;   // groups held by the images
    const int mode = ONEHOT ? (int)A_ONEHOT : d.amode;
    float* const gam = reinterpret_cast<float*>(img + kWimgVec);
    float* const bet = gam + 128;
    if (mode == A_LNSILU && threadIdx.x < 128) {
        const int f = threadIdx.x, ok = (f >> 3) < ngr;
        gam[f] = ok ? d.gamma[8 * g_lo + f] : 0.f;
        bet[f] = ok ? d.beta[8 * g_lo + f] : 0.f;
    }
    // this wave's items: feature group wave + 4 i of G and of the k-block of A; the write offset is the same for both.  An item that
    // does not exist (N or K end inside the tile) still LOADS -- from the wave's first item / the block's last group, a line that is
    // being read anyway -- so that the loads of an interval are one straight run of instructions; only its transform is skipped.
    const unsigned sw = (((unsigned)j & 3u) << 2) | (((unsigned)j >> 2) & 3u);
    unsigned wofs[4];
    bool gok[4], aok[4];
    size_t goff[4];                // uniform offsets (floats): the lane's part is added at the load (scalar base + vector offset)
    const float* ap[4];            // uniform
    size_t astr[4];
    const unsigned lane4 = lane * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave + 4 * i;
        wofs[i] = 256u * j + 16u * ((unsigned)g ^ sw) + 8u * h;
        gok[i] = g < d.NG && g < ngg;
        aok[i] = g < nag && g < ngr;
        goff[i] = (size_t)(gok[i] ? g : wave) * 256;
        const int G = g_lo + (g < ngr ? g : ngr - 1);
        const bool first = G < d.a0.groups;
        const Seg& sg = first ? d.a0 : d.a1;
        ap[i] = mode == A_ONEHOT ? d.G0 : sg.data + (size_t)(first ? G : G - d.a0.groups) * 256;   // one-hot: nothing to read, any valid line
        astr[i] = mode == A_ONEHOT ? 0 : (size_t)sg.groups * 256;
        // padding groups of the image: zero, once
        const uint2 z = make_uint2(0u, 0u);
        if (g < ngg && !gok[i]) { *reinterpret_cast<uint2*>(img + kWimgGH + wofs[i]) = z; *reinterpret_cast<uint2*>(img + kWimgGL + wofs[i]) = z; }
        if (g < nag && !aok[i]) { *reinterpret_cast<uint2*>(img + kWimgAH + wofs[i]) = z; *reinterpret_cast<uint2*>(img + kWimgAL + wofs[i]) = z; }
    }
    const size_t gstr = (size_t)d.NG * 256;
    // operand reads: lane 4 q + p of a 16-lane group supplies row q, features 4 p .. 4 p + 3 of its block; the lane's group takes
    // features 16 sub .. + 15 of the tile and rows 8 h + 4 e .. + 3 of the step (e = 0, 1: elements 0-3, 4-7 of the operand)
    const unsigned li = lane & 15, q = li >> 2, pp = li & 3, sub = (lane >> 4) & 1;
    const char* rb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        rb[e] = img + 256u * (8u * h + 4u * e + q) + 16u * ((2u * sub + (pp >> 1)) ^ ((2u * h + e) & 3u)) + 8u * (pp & 1u);
    const unsigned tq_g = 64u * ((unsigned)my_nt ^ q);
    unsigned tq_a[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) tq_a[kt] = 64u * ((unsigned)kt ^ q);

    // The second gradient tensor (a skip consumer's: few descriptors) is read one interval ahead only.  Every load of an interval is
    // unconditional, so that no register has two reaching definitions (hipcc resolved those with copies behind vmcnt(0) waits).
    float4 g1[HASG1 ? 4 : 1];
    auto fetch_g1 = [&](int t) {
        if constexpr (HASG1) {
            const float* const gb = d.G1 + (size_t)t * gstr;
#pragma unroll
            for (int i = 0; i < 4; ++i) g1[i] = ld4(gb + goff[i] + lane4);
        }
    };
    auto fetch = [&](WgradRegs& R, int t) {
        const float* const gb = d.G0 + (size_t)t * gstr;
#pragma unroll
        for (int i = 0; i < 4; ++i) R.g0[i] = ld4(gb + goff[i] + lane4);
        if constexpr (!ONEHOT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) R.av[i] = ld4(ap[i] + (size_t)t * astr[i] + lane4);
        }
        if (mode == A_ONEHOT) {
            const int row = t * 32 + j;
            R.oh = d.ts[row < d.nrows ? row : d.nrows - 1];
        }
        if (mode == A_LNSILU) R.ms = reinterpret_cast<const float2*>(d.rs)[(size_t)t * 32 + j];
    };
    auto put = [&](unsigned plane_hi, unsigned plane_lo, unsigned off, const float4 v) {
        unsigned h01, h23, l01, l23;
        split_pair(v.x, v.y, h01, l01);
        split_pair(v.z, v.w, h23, l23);
        *reinterpret_cast<uint2*>(img + plane_hi + off) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(img + plane_lo + off) = make_uint2(l01, l23);
    };
    auto stage = [&](const WgradRegs& R, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (gok[i]) {
                float4 v = R.g0[i];
                if constexpr (HASG1) v = make_float4(v.x + g1[i].x, v.y + g1[i].y, v.z + g1[i].z, v.w + g1[i].w);
                put(kWimgGH, kWimgGL, wofs[i], make_float4(v.x * gscale, v.y * gscale, v.z * gscale, v.w * gscale));
            }
        const bool live = t * 32 + j < d.nrows;        // forward tensors of padded rows are not zero
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (aok[i]) {
                float4 v;
                if (mode == A_ONEHOT) {
                    const int e = R.oh, f = 8 * (g_lo + wave + 4 * i) + 4 * h;
                    v = make_float4(e == f ? 1.f : 0.f, e == f + 1 ? 1.f : 0.f, e == f + 2 ? 1.f : 0.f, e == f + 3 ? 1.f : 0.f);
                } else if (mode == A_LNSILU) {
                    const float4 x = R.av[i];
                    const float4 gm = *reinterpret_cast<const float4*>(gam + 8 * (wave + 4 * i) + 4 * h);
                    const float4 bt = *reinterpret_cast<const float4*>(bet + 8 * (wave + 4 * i) + 4 * h);
                    const float c = R.ms.y, dd = -R.ms.x * R.ms.y;
                    v = make_float4(silu_scaled(fmaf(fmaf(x.x, c, dd), gm.x, bt.x)), silu_scaled(fmaf(fmaf(x.y, c, dd), gm.y, bt.y)),
                                    silu_scaled(fmaf(fmaf(x.z, c, dd), gm.z, bt.z)), silu_scaled(fmaf(fmaf(x.w, c, dd), gm.w, bt.w)));
                } else {
                    const float4 x = R.av[i];
                    v = make_float4(kRawScale * x.x, kRawScale * x.y, kRawScale * x.z, kRawScale * x.w);
                }
                if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
                put(kWimgAH, kWimgAL, wofs[i], v);
            }
    };
    // the MFMAs of an interval.  The k tiles go in pairs, a pair's four plane reads ahead of its six MFMAs; the second tile of a
    // pair may lie beyond KT: its accumulator then collects whatever the image holds there and is never written out.
    auto mma = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (rsplit == 2 && s != my_part) continue;      // two waves per out tile: one k16-step each
            const h8 ghi = wimg_tr8(rb[0] + tq_g + (kWimgGH + 4096u * s), rb[1] + tq_g + (kWimgGH + 4096u * s));
            const h8 glo = wimg_tr8(rb[0] + tq_g + (kWimgGL + 4096u * s), rb[1] + tq_g + (kWimgGL + 4096u * s));
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2)
                if (kp < KT) {
                    h8 ahi[2], alo[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        ahi[k] = wimg_tr8(rb[0] + tq_a[kp + k] + (kWimgAH + 4096u * s), rb[1] + tq_a[kp + k] + (kWimgAH + 4096u * s));
                        alo[k] = wimg_tr8(rb[0] + tq_a[kp + k] + (kWimgAL + 4096u * s), rb[1] + tq_a[kp + k] + (kWimgAL + 4096u * s));
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        DSG_MFMA_H(acc[kp + k], ghi, ahi[k]);
                        DSG_MFMA_H(acc[kp + k], ghi, alo[k]);
                        DSG_MFMA_H(acc[kp + k], glo, ahi[k]);
                    }
                }
        }
    };
    // Two register sets: the loads of interval t + 2 are issued when interval t's MFMAs start and are read a whole interval
    // later (one set gave them the MFMA phase only: ~1 000 cycles against >= 2 000 of memory latency).  Every fetch is
    // unconditional -- past the chunk's end it re-reads the last tile -- so that no register of a set has two reaching definitions.
    WgradRegs R0, R1;
#pragma unroll
    for (int i = 0; i < 4; ++i) R0.av[i] = R1.av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    R0.ms = R1.ms = make_float2(0.f, 0.f); R0.oh = R1.oh = -1;
    __syncthreads();                         // the LayerNorm vectors
    if (t_lo >= t_hi) return;
    const int t_last = t_hi - 1;
    auto clampt = [&](int t) { return t < t_last ? t : t_last; };
    fetch(R0, t_lo); fetch_g1(t_lo);
    fetch(R1, clampt(t_lo + 1));
    stage(R0, t_lo);
    __syncthreads();
    for (int t = t_lo; t < t_hi; t += 2) {
        // image: tile t; R1: tile t + 1; R0: free
        fetch(R0, clampt(t + 2));
        fetch_g1(clampt(t + 1));
        mma();
        __syncthreads();                     // everyone is done reading the image
        if (t + 1 < t_hi) stage(R1, t + 1);
        __syncthreads();
        // image: tile t + 1; R0: tile t + 2; R1: free
        fetch(R1, clampt(t + 3));
        fetch_g1(clampt(t + 2));
        if (t + 1 < t_hi) mma();
        __syncthreads();
        if (t + 2 < t_hi) stage(R0, t + 2);
        __syncthreads();
    }
}

// ---- NTp == 1 units (one out tile: the 4- to 32-wide Linears of the narrow blocks), the same construction over 64-row intervals: the
// four waves take one k16-step each.  G image [64 rows][32 features] with 72-byte rows (the 8-byte pad spreads a wave's row writes over
// the banks), A image as in the wide form.  KTM = 2: at most 64 in-features (every narrow block), four A items per wave; KTM = 4: eight.
constexpr unsigned kWimg2GH = 0, kWimg2GL = 4608, kWimg2AH = 9216, kWimg2AL = 25600, kWimg2Vec = 41984;   // 43 008 B in all
template <int KTM>
struct WgradRegs2 { float4 g0[2], av[2 * KTM]; float2 ms[2]; int oh[2]; };

template <int KTM, bool HASG1, bool ONEHOT>
__device__ __forceinline__ void wgrad_unit_h2(const WgradDesc& d, const WgradUnit& un, char* __restrict__ img, f32x16 (&acc)[4], float gscale,
                                              int ntiles, int nchunks, int wave, int lane, int KT, int g_lo, int ngr) {
    constexpr int NA = 2 * KTM;
    const int h = lane >> 5, j = lane & 31;
    const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    const int t_lo = un.chunk * tiles_per_chunk;
    const int t_hi = (t_lo + tiles_per_chunk < ntiles) ? t_lo + tiles_per_chunk : ntiles;
    const int nag = KT * 4;
    const int mode = ONEHOT ? (int)A_ONEHOT : d.amode;
    float* const gam = reinterpret_cast<float*>(img + kWimg2Vec);
    float* const bet = gam + 128;
    if (mode == A_LNSILU && threadIdx.x < 128) {
        const int f = threadIdx.x, ok = (f >> 3) < ngr;
        gam[f] = ok ? d.gamma[8 * g_lo + f] : 0.f;
        bet[f] = ok ? d.beta[8 * g_lo + f] : 0.f;
    }
    const unsigned sw = (((unsigned)j & 3u) << 2) | (((unsigned)j >> 2) & 3u);
    const unsigned lane4 = lane * 4;
    // G items: (group wave, tile i) ; A items: (group wave + 4 (i % KTM), tile i / KTM)
    unsigned gw[2], aw[NA];
    bool aok[NA];
    const float* ap[NA];
    size_t astr[NA];
    const bool gok = wave < d.NG;
    const size_t goff = (size_t)(gok ? wave : 0) * 256;
    const uint2 z = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gw[i] = 72u * (32u * i + j) + 16u * wave + 8u * h;
        if (!gok) { *reinterpret_cast<uint2*>(img + kWimg2GH + gw[i]) = z; *reinterpret_cast<uint2*>(img + kWimg2GL + gw[i]) = z; }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int gi = wave + 4 * (i % KTM), tt = i / KTM;
        aw[i] = 256u * (32u * tt + j) + 16u * ((unsigned)gi ^ sw) + 8u * h;
        aok[i] = gi < nag && gi < ngr;
        const int G = g_lo + (gi < ngr ? gi : ngr - 1);
        const bool first = G < d.a0.groups;
        const Seg& sg = first ? d.a0 : d.a1;
        ap[i] = mode == A_ONEHOT ? d.G0 : sg.data + (size_t)(first ? G : G - d.a0.groups) * 256;
        astr[i] = mode == A_ONEHOT ? 0 : (size_t)sg.groups * 256;
        if (gi < nag && !aok[i]) { *reinterpret_cast<uint2*>(img + kWimg2AH + aw[i]) = z; *reinterpret_cast<uint2*>(img + kWimg2AL + aw[i]) = z; }
    }
    const size_t gstr = (size_t)d.NG * 256;
    // operand reads of this wave's k16-step (s = wave)
    const unsigned li = lane & 15, q = li >> 2, pp = li & 3, sub = (lane >> 4) & 1;
    const char* rg[2];
    const char* rb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned row = 16u * wave + 8u * h + 4u * e + q;
        rg[e] = img + 72u * row + 32u * sub + 8u * pp;
        rb[e] = img + 256u * row + 16u * ((2u * sub + (pp >> 1)) ^ ((2u * h + e) & 3u)) + 8u * (pp & 1u);
    }
    unsigned tq_a[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) tq_a[kt] = 64u * ((unsigned)kt ^ q);

    const int t_last = t_hi - 1;
    auto clampt = [&](int t) { return t < t_last ? t : t_last; };
    float4 g1[HASG1 ? 2 : 1];
    auto fetch_g1 = [&](int t0) {
        if constexpr (HASG1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) g1[i] = ld4(d.G1 + (size_t)clampt(t0 + i) * gstr + goff + lane4);
        }
    };
    auto fetch = [&](WgradRegs2<KTM>& R, int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) R.g0[i] = ld4(d.G0 + (size_t)clampt(t0 + i) * gstr + goff + lane4);
        if constexpr (!ONEHOT) {
#pragma unroll
            for (int i = 0; i < NA; ++i) R.av[i] = ld4(ap[i] + (size_t)clampt(t0 + i / KTM) * astr[i] + lane4);
        }
        if (mode == A_ONEHOT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int row = clampt(t0 + i) * 32 + j; R.oh[i] = d.ts[row < d.nrows ? row : d.nrows - 1]; }
        }
        if (mode == A_LNSILU) {
#pragma unroll
            for (int i = 0; i < 2; ++i) R.ms[i] = reinterpret_cast<const float2*>(d.rs)[(size_t)clampt(t0 + i) * 32 + j];
        }
    };
    auto put = [&](unsigned plane_hi, unsigned plane_lo, unsigned off, const float4 v) {
        unsigned h01, h23, l01, l23;
        split_pair(v.x, v.y, h01, l01);
        split_pair(v.z, v.w, h23, l23);
        *reinterpret_cast<uint2*>(img + plane_hi + off) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(img + plane_lo + off) = make_uint2(l01, l23);
    };
    auto stage = [&](const WgradRegs2<KTM>& R, int t0) {
        if (gok) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float keep = t0 + i < t_hi ? 1.f : 0.f;        // a tile beyond the chunk: zero rows
                float4 v = R.g0[i];
                if constexpr (HASG1) v = make_float4(v.x + g1[i].x, v.y + g1[i].y, v.z + g1[i].z, v.w + g1[i].w);
                const float ks = keep * gscale;
                put(kWimg2GH, kWimg2GL, gw[i], make_float4(v.x * ks, v.y * ks, v.z * ks, v.w * ks));
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (aok[i]) {
                const int tt = i / KTM, tile = t0 + tt, gi = wave + 4 * (i % KTM);
                const bool live = tile < t_hi && tile * 32 + j < d.nrows;   // forward tensors of padded rows are not zero
                float4 v;
                if (mode == A_ONEHOT) {
                    const int e = R.oh[tt], f = 8 * (g_lo + gi) + 4 * h;
                    v = make_float4(e == f ? 1.f : 0.f, e == f + 1 ? 1.f : 0.f, e == f + 2 ? 1.f : 0.f, e == f + 3 ? 1.f : 0.f);
                } else if (mode == A_LNSILU) {
                    const float4 x = R.av[i];
                    const float4 gm = *reinterpret_cast<const float4*>(gam + 8 * gi + 4 * h);
                    const float4 bt = *reinterpret_cast<const float4*>(bet + 8 * gi + 4 * h);
                    const float c = R.ms[tt].y, dd = -R.ms[tt].x * R.ms[tt].y;
                    v = make_float4(silu_scaled(fmaf(fmaf(x.x, c, dd), gm.x, bt.x)), silu_scaled(fmaf(fmaf(x.y, c, dd), gm.y, bt.y)),
                                    silu_scaled(fmaf(fmaf(x.z, c, dd), gm.z, bt.z)), silu_scaled(fmaf(fmaf(x.w, c, dd), gm.w, bt.w)));
                } else {
                    const float4 x = R.av[i];
                    v = make_float4(kRawScale * x.x, kRawScale * x.y, kRawScale * x.z, kRawScale * x.w);
                }
                if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
                put(kWimg2AH, kWimg2AL, aw[i], v);
            }
    };
    auto mma = [&]() {
        const h8 ghi = wimg_tr8(rg[0] + kWimg2GH, rg[1] + kWimg2GH);
        const h8 glo = wimg_tr8(rg[0] + kWimg2GL, rg[1] + kWimg2GL);
#pragma unroll
        for (int kp = 0; kp < KTM; kp += 2)
            if (kp < KT) {
                h8 ahi[2], alo[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    ahi[k] = wimg_tr8(rb[0] + tq_a[kp + k] + kWimg2AH, rb[1] + tq_a[kp + k] + kWimg2AH);
                    alo[k] = wimg_tr8(rb[0] + tq_a[kp + k] + kWimg2AL, rb[1] + tq_a[kp + k] + kWimg2AL);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    DSG_MFMA_H(acc[kp + k], ghi, ahi[k]);
                    DSG_MFMA_H(acc[kp + k], ghi, alo[k]);
                    DSG_MFMA_H(acc[kp + k], glo, ahi[k]);
                }
            }
    };
    WgradRegs2<KTM> R0, R1;
    __syncthreads();                         // the LayerNorm vectors, the zeroed padding
    if (t_lo >= t_hi) return;
    fetch(R0, t_lo); fetch_g1(t_lo);
    fetch(R1, t_lo + 2);
    stage(R0, t_lo);
    __syncthreads();
    for (int t = t_lo; t < t_hi; t += 4) {
        // image: tiles t, t + 1; R1: t + 2, t + 3; R0: free
        fetch(R0, t + 4);
        fetch_g1(t + 2);
        mma();
        __syncthreads();
        if (t + 2 < t_hi) stage(R1, t + 2);
        __syncthreads();
        fetch(R1, t + 6);
        fetch_g1(t + 4);
        if (t + 2 < t_hi) mma();
        __syncthreads();
        if (t + 4 < t_hi) stage(R0, t + 4);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256, 2) void k_wgrad_h(const WgradDesc* __restrict__ descs, const WgradUnit* __restrict__ units,
                                                 const unsigned* __restrict__ gmax, float* __restrict__ slabs, size_t slab_stride, int ntiles,
                                                 int nchunks) {
    __shared__ __attribute__((aligned(16))) char img[kWhLdsBytes];
    const WgradUnit un = units[blockIdx.x];
    WgradDesc d = descs[un.desc];
    d.G0 = as_global(d.G0); d.G1 = as_global(d.G1); globalize(d.a0); globalize(d.a1);          // dsg_kernels.hpp, as_global
    d.rs = as_global(d.rs); d.gamma = as_global(d.gamma); d.beta = as_global(d.beta); d.ts = as_global(d.ts);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int NT = (d.N + 31) / 32;
    const int NTp = NT <= 1 ? 1 : (NT == 2 ? 2 : 4);
    const int rsplit = 4 / NTp;
    const int my_nt = wave % NTp, my_part = wave / NTp;
    const int g_lo = un.kblk * 16;
    const int g_hi = (g_lo + 16 < d.KG) ? g_lo + 16 : d.KG;
    const int ngr = g_hi - g_lo;
    const int KT = (ngr + 3) / 4;
    const int ge = wgrad_gexp(gmax[d.gmax_slot]);
    const float gscale = __int_as_float((127 + ge) << 23);
    const float ascale = d.amode == A_LNSILU ? kActScale : (d.amode == A_RAW ? kRawScale : 1.0f);
    const float unscale = __int_as_float((127 - ge) << 23) / ascale;

    f32x16 acc[4];
    acc_zero<4>(acc);
    // forms: (second gradient tensor) x (one-hot A operand: the time-table units, which have no second tensor)
    const bool g1 = d.G1 != nullptr, oh = d.amode == A_ONEHOT;
#define DSG_WG_ARGS2 d, un, img, acc, gscale, ntiles, nchunks, wave, lane, KT, g_lo, ngr
#define DSG_WG_ARGS1 d, un, img, acc, gscale, ntiles, nchunks, wave, lane, NTp, KT, g_lo, ngr, my_nt, my_part, rsplit
    if (NTp == 1 && KT <= 2) {
        if (oh) wgrad_unit_h2<2, false, true>(DSG_WG_ARGS2);
        else if (g1) wgrad_unit_h2<2, true, false>(DSG_WG_ARGS2);
        else wgrad_unit_h2<2, false, false>(DSG_WG_ARGS2);
    } else if (NTp == 1) {
        if (oh) wgrad_unit_h2<4, false, true>(DSG_WG_ARGS2);
        else if (g1) wgrad_unit_h2<4, true, false>(DSG_WG_ARGS2);
        else wgrad_unit_h2<4, false, false>(DSG_WG_ARGS2);
    } else {
        if (oh) wgrad_unit_h1<false, true>(DSG_WG_ARGS1);
        else if (g1) wgrad_unit_h1<true, false>(DSG_WG_ARGS1);
        else wgrad_unit_h1<false, false>(DSG_WG_ARGS1);
    }
#undef DSG_WG_ARGS1
#undef DSG_WG_ARGS2

    // ---- waves that split the rows of one n-tile add their partial tiles through LDS, then the first writes the slab
    if (rsplit > 1) {
        float* red = reinterpret_cast<float*>(img);  // <= 2 n-tiles x 4 k-tiles x 16 regs x 64 lanes = 32 KiB <= sizeof(img)
        for (int part = 1; part < rsplit; ++part) {
            if (my_part == part) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((my_nt * 4 + kt) * 16 + r) * 64 + lane] = acc[kt][r];
            }
            __syncthreads();
            if (my_part == 0) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][r] += red[((my_nt * 4 + kt) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (my_part == 0 && my_nt < NT) {
        float* out = slabs + (size_t)un.chunk * slab_stride + d.out_off;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < KT) {
                const int G = g_lo + 4 * kt + (j >> 3), e = j & 7;
                int col = -1;
                if (G < g_hi) {
                    if (d.amode == A_ONEHOT) { col = 8 * G + e; if (col >= d.onehot_n) col = -1; }
                    else if (G < d.a0.groups) { col = 8 * G + e; if (col >= d.a0.width) col = -1; }
                    else { const int c = 8 * (G - d.a0.groups) + e; col = c < d.a1.width ? d.a0.width + c : -1; }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * my_nt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (col >= 0 && n < d.N) out[(size_t)n * d.ld + col] = acc[kt][r] * unscale;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Activation gradients on the split path: dX = W^T dY as 3x v_mfma_f32_32x32x16_f16 per k16-step, with the gradient rows
// scaled one by one.  A row of dY lives in two lanes (j, j + 32); its largest element picks the power of two that puts the
// row's maximum into [2^11, 2^12), the weights carry the same 2^e as in the forward packs, and the accumulator is unscaled
// per lane afterwards -- exact powers of two, so the only rounding is the 22-bit operand split.
// ---------------------------------------------------------------------------------------------
struct BlockBwdArgsH {
    BlockBwdArgs b;
    const uint4* W3Th; const uint4* W2Th; const uint4* W1Th; const uint4* WscTh;   // transposed planes [OT][KS][2][64]
    const float* m1; const float* m2; const float* m3; const float* msc;          // max|W| (k_maxabs)
    unsigned* gmax_t; int gmax_ld;                    // per-tile max|G| words [slot][tile] (operand scales of k_wgrad_h)
    int slot_out, slot_h2, slot_h1;                   // slots of dout, dh2, dh1
};
__device__ __forceinline__ void globalize(BlockBwdArgsH& a) {
    globalize(a.b);
    a.W3Th = as_global(a.W3Th); a.W2Th = as_global(a.W2Th); a.W1Th = as_global(a.W1Th); a.WscTh = as_global(a.WscTh);
    a.m1 = as_global(a.m1); a.m2 = as_global(a.m2); a.m3 = as_global(a.m3); a.msc = as_global(a.msc); a.gmax_t = as_global(a.gmax_t);
}

template <int NT>
__device__ __forceinline__ void row_scale(const f32x16 (&g)[NT], float& s, float& sinv, unsigned& mbits) {
    unsigned m = 0u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned a = __float_as_uint(g[nt][r]) & 0x7fffffffu;
            m = m > a ? m : a;
        }
    m = xhalf_max(m);
    mbits = m;
    const int be = (int)(m >> 23);
    int e = be == 0 ? 0 : 11 - (be - 127);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    s = __int_as_float((127 + e) << 23);
    sinv = __int_as_float((127 - e) << 23);
}

// out[CH tiles] += planes x split(s * in): k over NG groups held in accumulator order, weight planes prefetched one step ahead
// `pre` (narrow blocks of the fused run): the planes of ALL k16-steps, requested at the top of the block -- with one or two steps per
// GEMM a prefetch distance of one step hides nothing, and four GEMMs per block each paid an L2 round trip (or two) of a lone wave
template <int NG, int NTI, int CH>
__device__ __forceinline__ void chain_scaled_chunk_h(f32x16* __restrict__ out, const f32x16 (&in)[NTI], const uint4* __restrict__ wp,
                                                     size_t nt_stride, int lane, float s, const HFrag<CH>* pre = nullptr) {
    constexpr int KS = (NG + 1) / 2;
    f32x16 (&o)[CH] = *reinterpret_cast<f32x16 (*)[CH]>(out);
    HFrag<CH> wn;
    if (pre) wn = pre[0];
    else load_hfrag<CH>(wn, wp + lane, nt_stride);
#pragma unroll
    for (int S = 0; S < KS; ++S) {
        HFrag<CH> wc = wn;
        if (S + 1 < KS) {
            if (pre) wn = pre[S + 1];
            else load_hfrag<CH>(wn, wp + (size_t)(S + 1) * 128 + lane, nt_stride);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int t = S >> 1, r0 = 8 * (S & 1);
        float v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = s * in[t][r0 + p];
        h8 bhi, blo;
        split8(v, bhi, blo);
        mfma_step_h<CH>(o, wc, bhi, blo);
    }
}

struct BwdGemmSplit {
    const BlockBwdArgsH& a;
    int lane;
    bool sclin;
    int tile;
    // all planes of GEMM `which` (NTout <= 4 out tiles, KS steps) into registers
    template <int NGin, int NTout>
    __device__ __forceinline__ void preload(HFrag<NTout> (&w)[(NGin + 1) / 2], int which) const {
        constexpr int KS = (NGin + 1) / 2;
        const uint4* wp = which == 3 ? a.W3Th : (which == 2 ? a.W2Th : (which == 1 ? a.W1Th : a.WscTh));
#pragma unroll
        for (int S = 0; S < KS; ++S) load_hfrag<NTout>(w[S], wp + (size_t)S * 128 + lane, (size_t)KS * 128);
    }
    template <int NGin, int NTin, int NTout>
    __device__ __forceinline__ void run(f32x16 (&out)[NTout], const f32x16 (&in)[NTin], int which, bool accumulate, int o0 = 0,
                                        const HFrag<(NTout < 4 ? NTout : 4)>* pre = nullptr) const {
        constexpr int CH = NTout < 4 ? NTout : 4;
        constexpr int KS = (NGin + 1) / 2;
        const uint4* wp = (which == 3 ? a.W3Th : (which == 2 ? a.W2Th : (which == 1 ? a.W1Th : a.WscTh))) + (size_t)o0 * KS * 128;
        int e;
        if (which == 3) e = sclin ? scale_exp_lin3(*a.m3, *a.msc) : scale_exp(*a.m3);
        else if (which == 2) e = scale_exp(*a.m2);
        else if (which == 1) e = scale_exp(*a.m1);
        else e = scale_exp_lin3(*a.m3, *a.msc) + 4;
        float s, sinv;
        unsigned m;
        row_scale<NTin>(in, s, sinv, m);
        if (which != 0) {       // the same rows are the G operand of this block's weight gradients: track the tensor maximum
            m = half_wave_max(m);
            if (lane == 0) a.gmax_t[(size_t)(which == 3 ? a.slot_out : (which == 2 ? a.slot_h2 : a.slot_h1)) * a.gmax_ld + tile] = m;
        }
        const float winv = __int_as_float((127 - e) << 23), wsc = __int_as_float((127 + e) << 23);
        if (accumulate) {
            const float pre = s * wsc;
#pragma unroll
            for (int nt = 0; nt < NTout; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[nt][r] *= pre;
        }
#pragma unroll
        for (int c0 = 0; c0 < NTout; c0 += CH)
            chain_scaled_chunk_h<NGin, NTin, CH>(&out[c0], in, wp + (size_t)c0 * KS * 128, (size_t)KS * 128, lane, s, (NTout <= 4 && c0 == 0) ? pre : nullptr);
        const float post = sinv * winv;
#pragma unroll
        for (int nt = 0; nt < NTout; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[nt][r] *= post;
    }
};

template <int N, bool SCLIN>
__global__ __launch_bounds__(256, 2) void k_resblock_bwd_h(const BlockBwdArgsH a) {
    // LayerNorm affine vectors from LDS (as in k_resblock_h): a quarter of this kernel's vector memory instructions
    __shared__ float lnp[2 * kLnLdsW1 + 4 * kLnLdsN];
    {
        const int n1 = 8 * (a.b.in0.groups + a.b.in1.groups), n2 = 8 * ((N + 7) / 8);
        for (int i = threadIdx.x; i < n1; i += blockDim.x) { lnp[i] = a.b.gamma1[i]; lnp[kLnLdsW1 + i] = a.b.beta1[i]; }
        float* q = lnp + 2 * kLnLdsW1;
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            q[i] = a.b.gamma2[i]; q[kLnLdsN + i] = a.b.beta2[i]; q[2 * kLnLdsN + i] = a.b.gamma3[i]; q[3 * kLnLdsN + i] = a.b.beta3[i];
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= a.b.ntiles) return;
    BlockBwdArgs b = a.b;
    b.gamma1 = lnp; b.beta1 = lnp + kLnLdsW1;
    b.gamma2 = lnp + 2 * kLnLdsW1; b.beta2 = b.gamma2 + kLnLdsN; b.gamma3 = b.gamma2 + 2 * kLnLdsN; b.beta3 = b.gamma2 + 3 * kLnLdsN;
    resblock_bwd_body<N, SCLIN>(b, BwdGemmSplit{a, lane, SCLIN, tile}, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Cooperative backward of the 64- and 128-wide blocks (mirror of k_resblock_c): the N/32 waves of a row tile each own one
// 32-feature slice of every gradient tensor.  One wave per tile (k_resblock_bwd_h) is a serial walk of ~600 MFMAs, ~12 000 VALU
// and a dozen exposed memory round trips: 70-105 us whatever the batch, 60 % of it waiting.  Here a wave does a quarter of
// the arithmetic, requests every saved tensor it will need at the start, and the row-wide quantities cross the slices
// through LDS:
//   row max of a gradient (the per-row power-of-two operand scale of the split GEMMs), LayerNorm statistics of h2 / h1
//   (per-slice (mean, M2), Chan merge), the two row sums of a LayerNorm backward, and the B-operand image itself (each
//   wave splits its slice into hi/lo halves, every wave reads all of it and multiplies by its own out tile of W^T).
// Same packed planes, scales, column-sum and max|G| outputs as k_resblock_bwd_h; only the order of the additions inside a
// row sum / statistic differs.  Requires in0 (and in1 for the concat blocks) exactly N wide.
// Workgroup = 4 waves = one 128-wide tile or two 64-wide tiles.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned slice_rowmax(const f32x16& v) {
    unsigned m = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned a = __float_as_uint(v[r]) & 0x7fffffffu;
        m = m > a ? m : a;
    }
    return xhalf_max(m);
}
// (mean, M2) of the 32 features of a slice per row
__device__ __forceinline__ void slice_stats(const f32x16& v, int h, float& mean, float& m2) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    mean = xhalf_sum(s) * (1.0f / 32);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; q = fmaf(d, d, q); }
    m2 = xhalf_sum(q);
}
__device__ __forceinline__ void rowscale_of(unsigned mbits, float& s, float& sinv) {
    const int be = (int)(mbits >> 23);
    int e = be == 0 ? 0 : 11 - (be - 127);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    s = __int_as_float((127 + e) << 23);
    sinv = __int_as_float((127 - e) << 23);
}
__device__ __forceinline__ void slice_load(f32x16& v, const float* __restrict__ p) {
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        const float4 t = ld4(p + (size_t)G * 256);
        v[4 * G] = t.x; v[4 * G + 1] = t.y; v[4 * G + 2] = t.z; v[4 * G + 3] = t.w;
    }
}
__device__ __forceinline__ void slice_store(const f32x16& v, float* __restrict__ p) {
#pragma unroll
    for (int G = 0; G < 4; ++G) st4(p + (size_t)G * 256, make_float4(v[4 * G], v[4 * G + 1], v[4 * G + 2], v[4 * G + 3]));
}
__device__ __forceinline__ void slice_colsum(const f32x16& v, float* __restrict__ dst, int lane, int h) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = v[r];
    colsum_store<16>(dst, 0, tile_colsum<16>(t, lane), lane, h);
}
// this wave's two k16-steps of the operand image: split(s * v)
__device__ __forceinline__ void slice_publish(uint4* __restrict__ img, int w, int lane, const f32x16& v, float s) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float q[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) q[p] = s * v[8 * half + p];
        h8 hi, lo;
        split8(q, hi, lo);
        coop_publish(img, 2 * w + half, lane, hi, lo);
    }
}
// LayerNorm + SiLU backward, first pass, on a 32-feature slice: d <- du * gamma; column sums of du and du * xhat; the slice's
// share of the two row sums.  gamma / beta point at the slice's first feature.
__device__ __forceinline__ void slice_ln_bwd1(f32x16& d, const f32x16& x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float mean, float rstd, int lane, int h, float* __restrict__ cs_beta, float* __restrict__ cs_gamma,
                                              bool live, float& s1, float& s2) {
    float db[16], dg[16];
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        const float4 gm = ld4(gamma + 8 * G + 4 * h), bt = ld4(beta + 8 * G + 4 * h);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float xh = (x[4 * G + p] - mean) * rstd;
            const float u = fmaf(xh, gmv[p], btv[p]);
            const float du = d[4 * G + p] * silu_grad(u);
            db[4 * G + p] = du;
            dg[4 * G + p] = du * xh;
            const float t = du * gmv[p];
            d[4 * G + p] = t;
            s1 += t;
            s2 = fmaf(t, xh, s2);
        }
    }
    const float cb = tile_colsum<16>(db, lane), cg = tile_colsum<16>(dg, lane);
    if (live) { colsum_store<16>(cs_beta, 0, cb, lane, h); colsum_store<16>(cs_gamma, 0, cg, lane, h); }
}
__device__ __forceinline__ void slice_ln_bwd2(f32x16& d, const f32x16& x, float mean, float rstd, float s1, float s2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float xh = (x[r] - mean) * rstd;
        d[r] = rstd * (d[r] - s1 - xh * s2);
    }
}

template <int N, bool SCLIN>
__global__ __launch_bounds__(256, 2) void k_resblock_bwd_c(const BlockBwdArgsH ah) {
    constexpr int NG = N / 8, NT = N / 32, TPW = 4 / NT, KS = NG / 2, DT = SCLIN ? 2 : 1;
    constexpr int NP = N, KP = DT * N;
    constexpr int kSlotU4 = 1024 / TPW;
    __shared__ uint4 img[1024];
    __shared__ unsigned rmax[4 * 32];
    __shared__ float2 rsum[4 * 32];
    __shared__ float2 lnst[2 * 4 * 32];
    const BlockBwdArgs& a = ah.b;
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave / NT, w = wave % NT;
    const int tile_raw = blockIdx.x * TPW + slot;
    const bool live = tile_raw < a.ntiles;            // idle slots of the last workgroup still meet the barriers
    const int tile = live ? tile_raw : a.ntiles - 1;
    uint4* const Bimg = img + slot * kSlotU4;
    unsigned* const rm = rmax + slot * NT * 32;
    float2* const rs = rsum + slot * NT * 32;
    float2* const st2 = lnst + slot * NT * 32;
    float2* const st1 = lnst + 128 + slot * NT * 32;
    float* const csb = a.cs + (size_t)tile * a.cs_stride;
    const size_t tS = ((size_t)tile * NG + 4 * w) * 256 + lane * 4;   // this wave's four groups of an N-wide tensor
    const float inv_n = 1.0f / N;

    // ---- every tensor this wave reads from HBM, requested up front
    f32x16 g, x2, x1, xin[DT];
    slice_load(g, a.gout_a + tS);
    slice_load(x2, a.h2 + tS);
    slice_load(x1, a.h1 + tS);
    slice_load(xin[0], a.in0.data + tS);
    if (SCLIN) slice_load(xin[DT - 1], a.in1.data + tS);
    if (a.gout_b) {
        f32x16 gb;
        slice_load(gb, a.gout_b + tS);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] += gb[r];
    }
    // LN1 statistics from the producers' per-row (mean, M2), combined over the concat (Chan), as k_resblock_bwd_h.  Evaluated where they
    // are used (here for the record the weight-gradient kernels read, and again in front of stage 1's LayerNorm backward: two L2-hot
    // loads and a dozen instructions) instead of carried across the whole kernel: three of the registers it does not have.
    auto ln1_stats = [&](float& mean1, float& rstd1, float& wtot) {
        int jj = j;
        asm volatile("" : "+v"(jj));                   // (opaque: keeps hipcc from merging the two evaluations back into one long-lived pair)
        const float2 s0 = reinterpret_cast<const float2*>(a.in0.stats)[(size_t)tile * 32 + jj];
        float mean = s0.x, m2 = s0.y, n = (float)a.in0.width;
        if (SCLIN) {
            const float2 s1 = reinterpret_cast<const float2*>(a.in1.stats)[(size_t)tile * 32 + jj];
            const float n1 = (float)a.in1.width, nt_ = n + n1;
            const float dd = s1.x - mean;
            m2 = m2 + s1.y + dd * dd * (n * n1 / nt_);
            mean = mean + dd * (n1 / nt_);
            n = nt_;
        }
        mean1 = mean; wtot = n;
        rstd1 = rsqrtf(m2 / n + kLnEps);
    };
    if (w == 0 && h == 0 && live) {
        float m_, r_, w_;
        ln1_stats(m_, r_, w_);
        reinterpret_cast<float2*>(a.rs1)[(size_t)tile * 32 + j] = make_float2(m_, r_);
    }
    // weight scales 2^-e (bind-time maxima): evaluated where they are used -- kept from here they are three of the registers this
    // kernel does not have (256 allocated, three spilled before)
    auto winv_of = [&](int stage) -> float {
        const int e = stage == 3 ? (SCLIN ? scale_exp_lin3(*ah.m3, *ah.msc) : scale_exp(*ah.m3)) : stage == 2 ? scale_exp(*ah.m2) : scale_exp(*ah.m1);
        return __int_as_float((127 - e) << 23);
    };

    // ---- dL/d(out): column sums, slice statistics of h2 / h1, row max
    if (live) slice_colsum(g, csb + 32 * w, lane, h);
    {
        float m, q;
        slice_stats(x2, h, m, q);
        if (h == 0) st2[w * 32 + j] = make_float2(m, q);
        slice_stats(x1, h, m, q);
        if (h == 0) st1[w * 32 + j] = make_float2(m, q);
        const unsigned mb = slice_rowmax(g);
        if (h == 0) rm[w * 32 + j] = mb;
    }
    HFrag<1> wf[KS];
    coop_load_w<KS>(wf, ah.W3Th + (size_t)w * KS * 128 + lane, KS);
    __syncthreads();                                                     // B0

    auto merged_rowmax = [&](int slot_g) {
        unsigned m = 0u;
#pragma unroll
        for (int q = 0; q < NT; ++q) { const unsigned v = rm[q * 32 + j]; m = m > v ? m : v; }
        if (w == 0) {      // the same rows are the G operand of this block's weight gradients: the tile's maximum sets their scale
            const unsigned t = half_wave_max(m);
            if (lane == 0 && live) ah.gmax_t[(size_t)slot_g * ah.gmax_ld + tile] = t;
        }
        return m;
    };
    float mean3, rstd3, mean2, rstd2;
    {
        float m2v;
        coop_merge_stats<NT>(st2, j, mean3, m2v);
        rstd3 = rsqrtf(m2v * inv_n + kLnEps);
        coop_merge_stats<NT>(st1, j, mean2, m2v);
        rstd2 = rsqrtf(m2v * inv_n + kLnEps);
        if (w == 0 && h == 0 && live) {
            reinterpret_cast<float2*>(a.rs3)[(size_t)tile * 32 + j] = make_float2(mean3, rstd3);
            reinterpret_cast<float2*>(a.rs2)[(size_t)tile * 32 + j] = make_float2(mean2, rstd2);
        }
    }
    float sg, sginv;
    rowscale_of(merged_rowmax(ah.slot_out), sg, sginv);
    slice_publish(Bimg, w, lane, g, sg);
    __syncthreads();                                                     // C0

    // ---- stage 3: d a3 = W3^T g (this wave's 32 features), LN3 / SiLU backward with h2
    f32x16 d[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) d[0][r] = 0.f;
    coop_mma<KS>(d, Bimg, wf, KS, lane);
    coop_load_w<KS>(wf, ah.W2Th + (size_t)w * KS * 128 + lane, KS);      // next stage's planes: requested now, used after three barriers
    {
        const float post = sginv * winv_of(3);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[0][r] *= post;
        float s1 = 0.f, s2 = 0.f;
        slice_ln_bwd1(d[0], x2, a.gamma3 + 32 * w, a.beta3 + 32 * w, mean3, rstd3, lane, h, csb + 3 * NP + 32 * w, csb + 4 * NP + 32 * w, live, s1, s2);
        s1 = xhalf_sum(s1); s2 = xhalf_sum(s2);
        if (h == 0) rs[w * 32 + j] = make_float2(s1, s2);
    }
    __syncthreads();                                                     // A1
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NT; ++q) { const float2 v = rs[q * 32 + j]; s1 += v.x; s2 += v.y; }
        slice_ln_bwd2(d[0], x2, mean3, rstd3, s1 * inv_n, s2 * inv_n);
        if (live) { slice_store(d[0], a.dh2 + tS); slice_colsum(d[0], csb + NP + 32 * w, lane, h); }
        const unsigned mb = slice_rowmax(d[0]);
        if (h == 0) rm[w * 32 + j] = mb;
    }
    __syncthreads();                                                     // B1
    float sd, sdinv;
    rowscale_of(merged_rowmax(ah.slot_h2), sd, sdinv);
    slice_publish(Bimg, w, lane, d[0], sd);
    __syncthreads();                                                     // C1

    // ---- stage 2: d a2 = W2^T dh2, LN2 / SiLU backward with h1
#pragma unroll
    for (int r = 0; r < 16; ++r) d[0][r] = 0.f;
    coop_mma_reload<KS>(d, Bimg, wf, ah.W1Th + (size_t)w * KS * 128 + lane, lane);      // + stage 1's planes, first out tile (the in0 slice)
    {
        const float post = sdinv * winv_of(2);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[0][r] *= post;
        float s1 = 0.f, s2 = 0.f;
        slice_ln_bwd1(d[0], x1, a.gamma2 + 32 * w, a.beta2 + 32 * w, mean2, rstd2, lane, h, csb + 5 * NP + 32 * w, csb + 6 * NP + 32 * w, live, s1, s2);
        s1 = xhalf_sum(s1); s2 = xhalf_sum(s2);
        if (h == 0) rs[w * 32 + j] = make_float2(s1, s2);
    }
    __syncthreads();                                                     // A2
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NT; ++q) { const float2 v = rs[q * 32 + j]; s1 += v.x; s2 += v.y; }
        slice_ln_bwd2(d[0], x1, mean2, rstd2, s1 * inv_n, s2 * inv_n);
        if (live) { slice_store(d[0], a.dh1 + tS); slice_colsum(d[0], csb + 2 * NP + 32 * w, lane, h); }
        const unsigned mb = slice_rowmax(d[0]);
        if (h == 0) rm[w * 32 + j] = mb;
    }
    __syncthreads();                                                     // B2
    rowscale_of(merged_rowmax(ah.slot_h1), sd, sdinv);
    slice_publish(Bimg, w, lane, d[0], sd);
    __syncthreads();                                                     // C2

    // ---- stage 1: d a1 = W1^T dh1 over the (concat) input: this wave's slice of in0 and, for an up block, of in1
    f32x16 dx[DT][1];
    float mean1, rstd1, wtot;
    ln1_stats(mean1, rstd1, wtot);
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int T = t * NT + w;                                    // 32-feature out tile of dL/dx
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[t][0][r] = 0.f;
            if (t + 1 < DT) coop_mma_reload<KS>(dx[t], Bimg, wf, ah.W1Th + (size_t)(T + NT) * KS * 128 + lane, lane);
            else if (SCLIN) coop_mma_reload<KS>(dx[t], Bimg, wf, ah.WscTh + (size_t)w * KS * 128 + lane, lane);
            else coop_mma<KS>(dx[t], Bimg, wf, KS, lane);
            const float post = sdinv * winv_of(1);
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[t][0][r] *= post;
            slice_ln_bwd1(dx[t][0], xin[t], a.gamma1 + 32 * T, a.beta1 + 32 * T, mean1, rstd1, lane, h, csb + 7 * NP + 32 * T,
                          csb + 7 * NP + KP + 32 * T, live, s1, s2);
        }
        s1 = xhalf_sum(s1); s2 = xhalf_sum(s2);
        if (h == 0) rs[w * 32 + j] = make_float2(s1, s2);
    }
    __syncthreads();                                                     // A3: sums published; every wave is done with the dh1 image
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NT; ++q) { const float2 v = rs[q * 32 + j]; s1 += v.x; s2 += v.y; }
        const float iw = 1.0f / wtot;
#pragma unroll
        for (int t = 0; t < DT; ++t) slice_ln_bwd2(dx[t][0], xin[t], mean1, rstd1, s1 * iw, s2 * iw);
    }
    // ---- shortcut: + Wsc^T g (Linear, the operand image of g once more) or + g (identity)
    if (SCLIN) {
        // (an opaque copy of the row scale: hipcc otherwise keeps the sixteen products g * sg of the FIRST publish alive until here --
        // they are values since the split takes rounded float32 operands -- and spills them; sixteen multiplies are cheaper)
        float sg_again = sg;
        asm volatile("" : "+v"(sg_again));
        slice_publish(Bimg, w, lane, g, sg_again);
        __syncthreads();                                                 // C3
        const int e0 = scale_exp_lin3(*ah.m3, *ah.msc) + 4;
        const float pre = sg * __int_as_float((127 + e0) << 23), post = sginv * __int_as_float((127 - e0) << 23);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[t][0][r] *= pre;
            if (t + 1 < DT) coop_mma_reload<KS>(dx[t], Bimg, wf, ah.WscTh + (size_t)(NT + w) * KS * 128 + lane, lane);
            else coop_mma<KS>(dx[t], Bimg, wf, KS, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) dx[t][0][r] *= post;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[0][0][r] += g[r];
    }
    if (live) {
        slice_store(dx[0][0], a.gin0 + tS);
        if (SCLIN) slice_store(dx[DT - 1][0], a.gin1 + tS);
    }
}

// ---------------------------------------------------------------------------------------------
// The narrow run of the backward pass in ONE launch (mirror of k_fused_narrow_h): a wave walks the run's operators in
// reverse order for its row tile.  Every gradient tensor is still stored (it is the G operand of a weight gradient) and
// the next operator reads it back from memory - written and read by the same wave, never read before it was written,
// so `s_waitcnt vmcnt(0)` between operators is all the ordering needed.  What goes away is a launch boundary per
// operator (27 of them in MSR-80c, each a drain, a launch and a cold start for ~10 us of latency-bound work).
// LayerNorm vectors come from global memory (the per-operator kernels stage them in LDS; here the operators of one
// workgroup's waves are not in step).
// ---------------------------------------------------------------------------------------------
struct FusedBwdOpH {
    int kind;        // 0 = residual block, 1 = plain Linear
    int N;           // block width
    int sclin;
    int ot;          // Linear: 32-feature tiles of its input gradient
    BlockBwdArgsH b;
    LinBwdArgs l;
};

__global__ __launch_bounds__(256, 2) void k_fused_narrow_bwd_h(const FusedBwdOpH* __restrict__ ops, int nops, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (tile >= ntiles) return;
    for (int i = 0; i < nops; ++i) {
        const FusedBwdOpH& op = ops[i];
        if (i) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous operator's stores (this wave's own) have landed
        if (op.kind == 0) {
            BlockBwdArgsH b = op.b;           // every pointer of the record is global (dsg_kernels.hpp, as_global)
            globalize(b);
            if (op.sclin) {
                switch (op.N) {
                    case 4: resblock_bwd_body<4, true, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, true, tile}, tile, lane); break;
                    case 8: resblock_bwd_body<8, true, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, true, tile}, tile, lane); break;
                    case 16: resblock_bwd_body<16, true, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, true, tile}, tile, lane); break;
                    default: resblock_bwd_body<32, true, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, true, tile}, tile, lane); break;
                }
            } else {
                switch (op.N) {
                    case 4: resblock_bwd_body<4, false, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, false, tile}, tile, lane); break;
                    case 8: resblock_bwd_body<8, false, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, false, tile}, tile, lane); break;
                    case 16: resblock_bwd_body<16, false, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, false, tile}, tile, lane); break;
                    default: resblock_bwd_body<32, false, BwdGemmSplit, true>(b.b, BwdGemmSplit{b, lane, false, tile}, tile, lane); break;
                }
            }
        } else {
            LinBwdArgs l = op.l;
            globalize(l);
            if (op.ot <= 1) linear_bwd_body<1, false>(l, tile, lane);
            else linear_bwd_body<2, false>(l, tile, lane);
        }
    }
}

}  // namespace dsg
