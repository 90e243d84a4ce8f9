// Training step kernels (gfx950): q_sample, eps-MSE loss, the backward of the fused ResidualBlock / Linear kernels
// of dsg_kernels.hpp, the grouped weight-gradient GEMM and the column-sum (bias / LayerNorm) gradients.
//
// Reference: DDPM.forward (classifier_free_MSR.py:100-112) + torch autograd through UNet1D (UNetCF.py:318-356).
//
// Data flow of one step (single pass, per-row ts):
//   k_qsample -> forward kernels (k_resblock with save_h1/save_h2) -> k_loss_grad
//   -> k_linear_bwd (final) -> k_resblock_bwd / k_linear_bwd in reverse op order   (activation gradients, registers)
//   -> k_wgrad (ONE grouped launch: dW = G^T A for every Linear, contraction over batch rows, LDS transposition)
//   -> k_colsum (ONE grouped launch: bias and LayerNorm gamma/beta gradients)
//   -> k_reduce_slabs (deterministic sum of the per-row-chunk partial slabs into the flat gradient bucket)
//   -> time-path backward on the [T x .] tables (k_small_gemm, tiny).
// All gradients land in ONE flat float32 buffer in state-dict order: the DP all-reduce bucket.
#pragma once
#include "dsg_kernels.hpp"

namespace dsg {

// ---------------------------------------------------------------------------------------------
// q_sample (MSR.py:103): y_t = sqrt_acp[ts]*y + sqrt_1m_acp[ts]*noise, written row-major (for feature_proj) and in
// fragment layout (A operand of feature_proj's weight gradient).
// ---------------------------------------------------------------------------------------------
__global__ void k_qsample(const float* __restrict__ y, const float* __restrict__ noise, const int* __restrict__ ts,
                          const float* __restrict__ sa, const float* __restrict__ sb, int nrows, int D, float* __restrict__ yt_rm,
                          float* __restrict__ yt_frag, int ntiles) {
    const int DG = (D + 7) / 8;
    const size_t total = (size_t)ntiles * DG * 256;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int p = idx & 3, lane = (idx >> 2) & 63;
        const size_t tg = idx >> 8;
        const int g = tg % DG, tile = tg / DG;
        const int row = tile * 32 + (lane & 31), f = 8 * g + 4 * (lane >> 5) + p;
        float v = 0.f;
        if (row < nrows && f < D) {
            const int t = ts[row];
            v = __fadd_rn(__fmul_rn(sa[t], y[(size_t)row * D + f]), __fmul_rn(sb[t], noise[(size_t)row * D + f]));
            yt_rm[(size_t)row * D + f] = v;
        }
        yt_frag[idx] = v;
    }
}

// The step's three draws on the device (dsg_train_draws): Philox4x32-10 keyed by seed, counter word 2 = 4*call + kind so that the
// three kinds and successive calls never share a counter.  noise: normal4 (Box-Muller, as the sampling loop's);
// ts = floor(u * T), u on a 24-bit grid;  mask = u < keep_prob.
// call_ptr (dsg_train_step_seeded_dyn: the step inside a captured graph): the call number is read from device memory -- a value baked into
// the launch would repeat the same draws at every replay; k_bump_u64 moves it on behind this kernel.
__global__ void k_train_draws(int* __restrict__ ts, float* __restrict__ noise, float* __restrict__ mask, int B, int D, int T, float keep,
                              unsigned long long seed, unsigned call_v, const unsigned long long* __restrict__ call_ptr) {
    const unsigned call = call_ptr ? (unsigned)*call_ptr : call_v;
    const size_t n = (size_t)B * D, n4 = (n + 3) / 4, b4 = ((size_t)B + 3) / 4;
    for (size_t i4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * blockDim.x) {
        if (noise) {
            float zz[4];
            normal4(seed, 4u * call, i4, zz);
            for (int p = 0; p < 4; ++p)
                if (i4 * 4 + p < n) noise[i4 * 4 + p] = zz[p];
        }
        if (i4 < b4) {
            uint32_t r[4], q[4];
            philox4x32_10((uint32_t)i4, (uint32_t)(i4 >> 32), 4u * call + 1u, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            philox4x32_10((uint32_t)i4, (uint32_t)(i4 >> 32), 4u * call + 2u, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), q);
            for (int p = 0; p < 4; ++p) {
                const size_t row = i4 * 4 + p;
                if (row >= (size_t)B) break;
                if (ts) { const int t = (int)(((r[p] >> 8) * (1.0f / 16777216.0f)) * (float)T); ts[row] = t < T ? t : T - 1; }
                if (mask) mask[row] = ((q[p] >> 8) * (1.0f / 16777216.0f)) < keep ? 1.0f : 0.0f;
            }
        }
    }
}

__global__ void k_bump_u64(unsigned long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1ull; }
__global__ void k_bump_f32(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1.0f; }

// loss = mean((noise - eps_hat)^2) (F.mse_loss, MSR.py:112); d_eps = 2 (eps_hat - noise) / (B*D) in fragment layout.
// Per-block float64 partial sums; k_loss_final reduces them in a fixed order.
__global__ __launch_bounds__(256) void k_loss_grad(const float* __restrict__ eps, const float* __restrict__ noise, int nrows, int D,
                                                   float* __restrict__ deps_frag, int ntiles, double* __restrict__ part) {
    __shared__ double sm[4];
    const int DG = (D + 7) / 8;
    const size_t total = (size_t)ntiles * DG * 256;
    const float scale = 2.0f / ((float)nrows * (float)D);
    double s = 0.0;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int p = idx & 3, lane = (idx >> 2) & 63;
        const size_t tg = idx >> 8;
        const int g = tg % DG, tile = tg / DG;
        const int row = tile * 32 + (lane & 31), f = 8 * g + 4 * (lane >> 5) + p;
        float v = 0.f;
        if (row < nrows && f < D) {
            const float d = eps[(size_t)row * D + f] - noise[(size_t)row * D + f];
            s += (double)d * (double)d;
            v = d * scale;
        }
        deps_frag[idx] = v;
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_loss_final(const double* __restrict__ part, int nparts, double denom, float* __restrict__ loss) {
    __shared__ double sm[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) t += part[i];     // fixed assignment and tree: deterministic
    t = block_sum(t, sm);
    if (threadIdx.x == 0) *loss = (float)(t / denom);
}

// ---------------------------------------------------------------------------------------------
// Column sums over the 32 rows of a tile, in registers (bias and LayerNorm gamma/beta gradients of the residual blocks).
// v[r], r < R (R a power of two <= 32), holds one value per row (lane & 31) for R different columns.  Afterwards lane l
// returns the sum over the 32 lanes of its half of column l & (R - 1).  Butterfly: each level halves the registers -- the
// lane keeps the half selected by one lane bit and adds its partner's copy of that half -- so the whole reduction costs
// about 3R VALU instead of 5R shuffles.
// ---------------------------------------------------------------------------------------------
// Everything at VALU speed (a lone wave waits ~100+ cycles on each ds_swizzle / ds_bpermute through the LDS crossbar, and a block's
// backward has ~40 of them in dependent chains):  xor 1, 2: DPP quad_perm;  xor 4: row_half_mirror then quad reverse (7 - i, then
// i ^ 3: together i ^ 4);  xor 8: row_ror 8;  xor 16 / 32: gfx950's v_permlane16_swap / v_permlane32_swap.
template <int X>
__device__ __forceinline__ int lane_xor_i(int v) {
    static_assert(X == 1 || X == 2 || X == 4 || X == 8, "lane_xor: DPP forms");
    if constexpr (X == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
    else if constexpr (X == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    else if constexpr (X == 4) return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true), 0x1B, 0xf, 0xf, true);
    else return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);
}
template <int X>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor_i<X>(__float_as_int(v))); }
// max over the 32 lanes of each half-wave (every lane gets it)
__device__ __forceinline__ unsigned half_wave_max(unsigned m) {
    unsigned t;
    t = (unsigned)lane_xor_i<1>((int)m); m = m > t ? m : t;
    t = (unsigned)lane_xor_i<2>((int)m); m = m > t ? m : t;
    t = (unsigned)lane_xor_i<4>((int)m); m = m > t ? m : t;
    t = (unsigned)lane_xor_i<8>((int)m); m = m > t ? m : t;
    const auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
template <int CNT, int X>
__device__ __forceinline__ void colsum_level(float* v, int lane) {
    if constexpr (X == 16) {
        // one swap does the whole level: odd 16-lane rows of `lo` trade places with even rows of `hi`, after which every lane holds
        // its own kept half in one register and its partner's copy of that half in the other (same two addends as the select form)
        if constexpr (CNT >= 2) {
#pragma unroll
            for (int i = 0; i < CNT / 2; ++i) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
                v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        } else {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[0]), false, false);
            v[0] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    } else if constexpr (CNT >= 2) {
        const bool b = lane & X;
#pragma unroll
        for (int i = 0; i < CNT / 2; ++i) {
            const float lo = v[2 * i], hi = v[2 * i + 1];
            v[i] = (b ? hi : lo) + lane_xor<X>(b ? lo : hi);
        }
    } else {
        v[0] += lane_xor<X>(v[0]);
    }
}
template <int R>
__device__ __forceinline__ float tile_colsum(float (&v)[R], int lane) {
    static_assert(R == 4 || R == 8 || R == 16 || R == 32, "tile_colsum: R");
    colsum_level<R, 1>(v, lane);
    colsum_level<(R >= 2 ? R / 2 : 1), 2>(v, lane);
    colsum_level<(R >= 4 ? R / 4 : 1), 4>(v, lane);
    colsum_level<(R >= 8 ? R / 8 : 1), 8>(v, lane);
    colsum_level<(R >= 16 ? R / 16 : 1), 16>(v, lane);
    return v[0];
}
// groups per reduction block of an NG-group tensor, and the store of a block's result: register r of the block is
// (group G0 + r/4, element r%4), i.e. feature 8*(G0 + r/4) + 4h + r%4 of the padded feature order
template <int NGX>
struct CsBlock { static constexpr int GB = NGX >= 8 ? 8 : (NGX >= 4 ? 4 : (NGX >= 2 ? 2 : 1)); };
template <int R>
__device__ __forceinline__ void colsum_store(float* __restrict__ dst, int G0, float s, int lane, int h) {
    const int r = lane & (R - 1);
    if ((lane & 31) < R) dst[8 * (G0 + (r >> 2)) + 4 * h + (r & 3)] = s;
}
// column sums of an accumulator-resident tensor (NG groups) -> dst[padded feature]
template <int NG, int NT>
__device__ __forceinline__ void acc_colsum_store(const f32x16 (&a)[NT], float* __restrict__ dst, int lane, int h) {
    constexpr int GB = CsBlock<NG>::GB;
#pragma unroll
    for (int G0 = 0; G0 < NG; G0 += GB) {
        float v[GB * 4];
#pragma unroll
        for (int Gl = 0; Gl < GB; ++Gl)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int G = G0 + Gl;
                v[4 * Gl + p] = G < NG ? a[G >> 2][4 * (G & 3) + p] : 0.f;
            }
        colsum_store<GB * 4>(dst, G0, tile_colsum<GB * 4>(v, lane), lane, h);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm + SiLU backward on an accumulator-resident gradient (in place):
//   in : da[f] = dL/d silu(u[f]),  x[f] = LN input,  u = xhat*gamma + beta
//   out: da[f] <- dL/dx[f] ;  du_out[f] = dL/du[f] (stored for the gamma/beta column sums)
// Row reductions run over the true width n_true; padded features are forced to zero.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_grad(float u) {
    // bare v_exp_f32: the product's rounding (|u| * 4e-8 relative on exp) is far below the gradient tolerance
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896341f));
    return sg * fmaf(u, 1.0f - sg, 1.0f);
}

// NG groups held in acc arrays da[], x[] (accumulator order).  W = true width.
// cs_beta / cs_gamma: this tile's column-sum vectors (sum_rows du, sum_rows du*xhat) in padded feature order.
template <int NG, int NT>
__device__ __forceinline__ void ln_silu_bwd_acc(f32x16 (&da)[NT], const f32x16 (&x)[NT], const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float mean, float rstd, int W, int h, int lane,
                                                float* __restrict__ cs_beta, float* __restrict__ cs_gamma) {
    constexpr int GB = CsBlock<NG>::GB;
    float s1 = 0.f, s2 = 0.f;
    // narrow operators (NG <= 4: the fused narrow backward, one wave per SIMD): every group's vectors in one batch of loads pinned above
    // the arithmetic -- read group by group each pair was an exposed round trip (load, full wait, four silu_grad, next pair: disassembly)
    constexpr bool PREV = NG <= 4;
    float4 gmA[PREV ? NG : 1], btA[PREV ? NG : 1];
    if constexpr (PREV) {
#pragma unroll
        for (int G = 0; G < NG; ++G) { gmA[G] = ld4(gamma + 8 * G + 4 * h); btA[G] = ld4(beta + 8 * G + 4 * h); }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int G0 = 0; G0 < NG; G0 += GB) {
        float db[GB * 4], dg[GB * 4];
#pragma unroll
        for (int Gl = 0; Gl < GB; ++Gl) {
            const int G = G0 + Gl;
            if (G < NG) {
                const float4 gm = PREV ? gmA[PREV ? G : 0] : ld4(gamma + 8 * G + 4 * h), bt = PREV ? btA[PREV ? G : 0] : ld4(beta + 8 * G + 4 * h);
                const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool ok = 8 * G + 4 * h + p < W;
                    const float xh = (x[G >> 2][4 * (G & 3) + p] - mean) * rstd;
                    const float u = fmaf(xh, gmv[p], btv[p]);
                    float dsu = da[G >> 2][4 * (G & 3) + p] * silu_grad(u);
                    asm volatile("" : "+v"(dsu));      // computed for every lane: as `ok ? ... : 0` hipcc branched around it per element, the beta load inside the branch
                    const float du = ok ? dsu : 0.f;
                    db[4 * Gl + p] = du;
                    dg[4 * Gl + p] = du * xh;
                    const float t = du * gmv[p];
                    da[G >> 2][4 * (G & 3) + p] = t;
                    s1 += t;
                    s2 = fmaf(t, xh, s2);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) { db[4 * Gl + p] = 0.f; dg[4 * Gl + p] = 0.f; }
            }
        }
        colsum_store<GB * 4>(cs_beta, G0, tile_colsum<GB * 4>(db, lane), lane, h);
        colsum_store<GB * 4>(cs_gamma, G0, tile_colsum<GB * 4>(dg, lane), lane, h);
    }
    s1 = xhalf_sum(s1) * (1.0f / W);
    s2 = xhalf_sum(s2) * (1.0f / W);
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool ok = 8 * G + 4 * h + p < W;
            const float xh = (x[G >> 2][4 * (G & 3) + p] - mean) * rstd;
            const float t = da[G >> 2][4 * (G & 3) + p];
            da[G >> 2][4 * (G & 3) + p] = ok ? rstd * (t - s1 - xh * s2) : 0.f;
        }
}

template <int NG, int NT>
__device__ __forceinline__ void acc_load(f32x16 (&a)[NT], const float* __restrict__ p /* tile base + lane*4 */) {
#pragma unroll
    for (int G = 0; G < NT * 4; ++G) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (G < NG) v = ld4(p + (size_t)G * 256);
        a[G >> 2][4 * (G & 3) + 0] = v.x; a[G >> 2][4 * (G & 3) + 1] = v.y;
        a[G >> 2][4 * (G & 3) + 2] = v.z; a[G >> 2][4 * (G & 3) + 3] = v.w;
    }
}
template <int NG, int NT>
__device__ __forceinline__ void acc_load_add(f32x16 (&a)[NT], const float* __restrict__ p) {
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        const float4 v = ld4(p + (size_t)G * 256);
        a[G >> 2][4 * (G & 3) + 0] += v.x; a[G >> 2][4 * (G & 3) + 1] += v.y;
        a[G >> 2][4 * (G & 3) + 2] += v.z; a[G >> 2][4 * (G & 3) + 3] += v.w;
    }
}
template <int NG, int NT>
__device__ __forceinline__ void acc_store(const f32x16 (&a)[NT], float* __restrict__ p) {
#pragma unroll
    for (int G = 0; G < NG; ++G)
        st4(p + (size_t)G * 256, make_float4(a[G >> 2][4 * (G & 3)], a[G >> 2][4 * (G & 3) + 1], a[G >> 2][4 * (G & 3) + 2],
                                             a[G >> 2][4 * (G & 3) + 3]));
}
template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&a)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[nt][r] = 0.f;
}

// out += Wt * in  where `in` is an accumulator-resident gradient of NGin groups (raw, no activation).
template <int NGin, int NTin, int NTout>
__device__ __forceinline__ void chain_raw_from_acc(f32x16 (&out)[NTout], const f32x16 (&in)[NTin], const float* __restrict__ wp,
                                                   int lane) {
    const size_t nt_stride = (size_t)NGin * 256;
    float4 wn[NTout];
    load_wfrag<NTout>(wn, wp + lane * 4, nt_stride);
#pragma unroll
    for (int G = 0; G < NGin; ++G) {
        float4 wc[NTout];
#pragma unroll
        for (int nt = 0; nt < NTout; ++nt) wc[nt] = wn[nt];
        if (G + 1 < NGin) load_wfrag<NTout>(wn, wp + (size_t)(G + 1) * 256 + lane * 4, nt_stride);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<NTout>(out, wc, in[G >> 2][4 * (G & 3) + 0], in[G >> 2][4 * (G & 3) + 1], in[G >> 2][4 * (G & 3) + 2],
                          in[G >> 2][4 * (G & 3) + 3]);
    }
}

// ---------------------------------------------------------------------------------------------
// ResidualBlock backward, one wave per 32-row tile (mirror of k_resblock).
// ---------------------------------------------------------------------------------------------
struct BlockBwdArgs {
    Seg in0, in1;            // forward inputs (+ their (mean, M2) statistics)
    const float* h1;         // saved pre-LN2 tensor [tiles][NG][256]
    const float* h2;         // saved pre-LN3 tensor
    const float* gout_a;     // dL/d(out) from the chain consumer
    const float* gout_b;     // dL/d(out) from the skip consumer, or null
    const float* W3T;        // packed transposed weights
    const float* W2T;
    const float* W1T;        // [OT1][NG][256], OT1 = ceil(8*KG/32)
    const float* WscT;       // or null (identity shortcut)
    const float* gamma1; const float* beta1;
    const float* gamma2; const float* beta2;
    const float* gamma3; const float* beta3;
    float* gin0;             // dL/d(in0) [tiles][g0][256]
    float* gin1;             // dL/d(in1) [tiles][g1][256] or null
    float* du1;                           // dL/du of stage 1, fragment layout: scratch of the two-half variant only
    float* dh1; float* dh2;               // dL/dh1, dL/dh2 (G operands of the weight gradients)
    float* rs1; float* rs2; float* rs3;   // per-row (mean, rstd) of LN1/2/3 for the weight-gradient A operands
    float* cs;               // this block's column-sum region: cs[tile * cs_stride + ...], layout in resblock_bwd_body
    size_t cs_stride;
    int ntiles;
};
// (dsg_kernels.hpp, as_global) for records read from an operator table in memory
__device__ __forceinline__ void globalize(BlockBwdArgs& a) {
    globalize(a.in0); globalize(a.in1);
    a.h1 = as_global(a.h1); a.h2 = as_global(a.h2); a.gout_a = as_global(a.gout_a); a.gout_b = as_global(a.gout_b);
    a.W3T = as_global(a.W3T); a.W2T = as_global(a.W2T); a.W1T = as_global(a.W1T); a.WscT = as_global(a.WscT);
    a.gamma1 = as_global(a.gamma1); a.beta1 = as_global(a.beta1); a.gamma2 = as_global(a.gamma2); a.beta2 = as_global(a.beta2);
    a.gamma3 = as_global(a.gamma3); a.beta3 = as_global(a.beta3);
    a.gin0 = as_global(a.gin0); a.gin1 = as_global(a.gin1); a.du1 = as_global(a.du1); a.dh1 = as_global(a.dh1); a.dh2 = as_global(a.dh2);
    a.rs1 = as_global(a.rs1); a.rs2 = as_global(a.rs2); a.rs3 = as_global(a.rs3); a.cs = as_global(a.cs);
}

// The four data-gradient GEMMs of a block go through a policy object (`gemm.run<NGin, NTin, NTout>(out, in, which, accumulate)`,
// which = 3, 2, 1 for W3^T, W2^T, W1^T and 0 for the Linear shortcut; o0 = first output tile): exact f32 MFMA here, split-f16 in dsg_train_split.hpp.
struct BwdGemmF32 {
    const BlockBwdArgs& a;
    int lane;
    template <int NGin, int NTin, int NTout>
    __device__ __forceinline__ void run(f32x16 (&out)[NTout], const f32x16 (&in)[NTin], int which, bool /*accumulate*/, int o0 = 0) const {
        const float* wp = which == 3 ? a.W3T : (which == 2 ? a.W2T : (which == 1 ? a.W1T : a.WscT));
        chain_raw_from_acc<NGin, NTin, NTout>(out, in, wp + (size_t)o0 * NGin * 256, lane);
    }
};

// SCLIN <=> the block has a concat input (up blocks): the stage-1 data gradient then spans 2*NG groups.
// HOIST (narrow blocks inside the fused backward run, one wave per tile and nothing else on the SIMD): every saved tensor the block
// reads -- h2, h1, the forward input, dL/d(out) for the shortcut -- is requested at the top and kept in registers, instead of
// a dependent memory round trip in front of each stage.
template <int N, bool SCLIN, class Gemm, bool HOIST = false>
__device__ __forceinline__ void resblock_bwd_body(const BlockBwdArgs& a, const Gemm& gemm, int tile, int lane) {
    constexpr int NG = (N + 7) / 8, NT = (N + 31) / 32;
    constexpr int KGT = SCLIN ? (2 * NG + 3) / 4 : NT;  // 32-feature output tiles of dL/dx
    const int h = lane >> 5, j = lane & 31;
    const int KG = a.in0.groups + a.in1.groups;
    const size_t tN = (size_t)tile * NG * 256 + lane * 4;

    // per-tile column sums (bias and LayerNorm gradients), padded feature order; gathered by k_cs_reduce:
    //   [dout | dh2 | dh1 | beta3 | gamma3 | beta2 | gamma2] x NP, then [beta1 | gamma1] x KP
    constexpr int NP = NT * 32, KP = KGT * 32;
    float* const csb = a.cs + (size_t)tile * a.cs_stride;
    DSG_STAMP(HOIST && tile == 0, 0x21);

    // ---- dL/d(out)
    static_assert(!HOIST || (NT == 1 && KGT <= 2), "HOIST: narrow blocks only");
    f32x16 g[NT];
    acc_load<NG, NT>(g, a.gout_a + tN);
    constexpr int HN = HOIST ? NT : 1, HK = HOIST ? KGT : 1;
    f32x16 xh2[HN], xh1[HN], xin[HK], gk[HN];
    if constexpr (HOIST) {
        acc_load<NG, NT>(xh2, a.h2 + tN);
        acc_load<NG, NT>(xh1, a.h1 + tN);
#pragma unroll
        for (int G = 0; G < KGT * 4; ++G) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (G < KG) {
                const bool first = G < a.in0.groups;
                const Seg& sg = first ? a.in0 : a.in1;
                const int gl = first ? G : G - a.in0.groups;
                v = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
            }
            xin[G >> 2][4 * (G & 3)] = v.x; xin[G >> 2][4 * (G & 3) + 1] = v.y; xin[G >> 2][4 * (G & 3) + 2] = v.z; xin[G >> 2][4 * (G & 3) + 3] = v.w;
        }
    }
    constexpr int KSB = (NG + 1) / 2;                 // k16-steps of the four transposed GEMMs (K = N)
    HFrag<HN> w3p[HOIST ? KSB : 1], w2p[HOIST ? KSB : 1];
    HFrag<HK> w1p[HOIST ? KSB : 1], wscp[HOIST ? KSB : 1];
    if constexpr (HOIST) {
        gemm.template preload<NG, NT>(w3p, 3);
        gemm.template preload<NG, NT>(w2p, 2);
        gemm.template preload<NG, KGT>(w1p, 1);
        if (SCLIN) gemm.template preload<NG, KGT>(wscp, 0);
    }
    if (a.gout_b) acc_load_add<NG, NT>(g, a.gout_b + tN);
    if constexpr (HOIST) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) gk[nt] = g[nt];
    }
    acc_colsum_store<NG, NT>(g, csb, lane, h);
    DSG_STAMP(HOIST && tile == 0, 0x22);

    // ---- stage 3: d a3 = W3^T g ; LN3/SiLU backward with h2
    f32x16 d[NT], x[NT];
    acc_zero<NT>(d);
    if constexpr (HOIST) gemm.template run<NG, NT, NT>(d, g, 3, false, 0, w3p);
    else gemm.template run<NG, NT, NT>(d, g, 3, false);
    DSG_STAMP(HOIST && tile == 0, 0x23);
    if constexpr (HOIST) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) x[nt] = xh2[nt];
    } else {
        acc_load<NG, NT>(x, a.h2 + tN);
    }
    {
        float mean, m2;
        acc_stats<N, NT>(x, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        if (h == 0) reinterpret_cast<float2*>(a.rs3)[(size_t)tile * 32 + j] = make_float2(mean, rstd);
        ln_silu_bwd_acc<NG, NT>(d, x, a.gamma3, a.beta3, mean, rstd, N, h, lane, csb + 3 * NP, csb + 4 * NP);
    }
    acc_store<NG, NT>(d, a.dh2 + tN);
    DSG_STAMP(HOIST && tile == 0, 0x24);
    acc_colsum_store<NG, NT>(d, csb + NP, lane, h);

    // ---- stage 2: d a2 = W2^T dh2 ; LN2/SiLU backward with h1
    f32x16 (&d1)[NT] = g;  // reuse: g is re-read from memory for the shortcut
    acc_zero<NT>(d1);
    if constexpr (HOIST) gemm.template run<NG, NT, NT>(d1, d, 2, false, 0, w2p);
    else gemm.template run<NG, NT, NT>(d1, d, 2, false);
    DSG_STAMP(HOIST && tile == 0, 0x25);
    if constexpr (HOIST) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) x[nt] = xh1[nt];
    } else {
        acc_load<NG, NT>(x, a.h1 + tN);
    }
    {
        float mean, m2;
        acc_stats<N, NT>(x, h, mean, m2);
        const float rstd = rsqrtf(m2 * (1.0f / N) + kLnEps);
        if (h == 0) reinterpret_cast<float2*>(a.rs2)[(size_t)tile * 32 + j] = make_float2(mean, rstd);
        ln_silu_bwd_acc<NG, NT>(d1, x, a.gamma2, a.beta2, mean, rstd, N, h, lane, csb + 5 * NP, csb + 6 * NP);
    }
    acc_store<NG, NT>(d1, a.dh1 + tN);
    DSG_STAMP(HOIST && tile == 0, 0x26);
    acc_colsum_store<NG, NT>(d1, csb + 2 * NP, lane, h);

    // ---- stage 1: d a1 = W1^T dh1 over the concat width; LN1/SiLU backward with x = cat(in0, in1)
    float mean1, rstd1, wtot;
    {
        const float2 s0 = reinterpret_cast<const float2*>(a.in0.stats)[(size_t)tile * 32 + j];
        float mean = s0.x, m2 = s0.y, n = (float)a.in0.width;
        if (a.in1.groups) {
            const float2 s1 = reinterpret_cast<const float2*>(a.in1.stats)[(size_t)tile * 32 + j];
            const float n1 = (float)a.in1.width, nt_ = n + n1;
            const float dd = s1.x - mean;
            m2 = m2 + s1.y + dd * dd * (n * n1 / nt_);
            mean = mean + dd * (n1 / nt_);
            n = nt_;
        }
        mean1 = mean; wtot = n;
        rstd1 = rsqrtf(m2 / n + kLnEps);
        if (h == 0) reinterpret_cast<float2*>(a.rs1)[(size_t)tile * 32 + j] = make_float2(mean1, rstd1);
    }
    if constexpr (KGT > 4) {
        // 256-wide concat input: dL/dx in two halves of 4 output tiles so that the kernel fits two waves per SIMD.
        // Pass A: GEMM half, dL/du (stored: it is also the LayerNorm gamma/beta operand) and the two row sums.
        // Pass B: dL/du re-read, LayerNorm backward, + shortcut GEMM half, store.
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int hb = 0; hb < KGT / 4; ++hb) {
            f32x16 dxh[4];
            acc_zero<4>(dxh);
            gemm.template run<NG, NT, 4>(dxh, d1, 1, false, 4 * hb);
#pragma unroll
            for (int B8 = 0; B8 < 2; ++B8) {
                float db[32], dg[32];
#pragma unroll
                for (int G8 = 0; G8 < 8; ++G8) {
                    const int Gl = 8 * B8 + G8, G = 16 * hb + Gl;
#pragma unroll
                    for (int p = 0; p < 4; ++p) { db[4 * G8 + p] = 0.f; dg[4 * G8 + p] = 0.f; }
                    if (G < KG) {
                        const bool first = G < a.in0.groups;
                        const Seg& sg = first ? a.in0 : a.in1;
                        const int gl = first ? G : G - a.in0.groups;
                        const float4 xv = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
                        const float4 gm = ld4(a.gamma1 + 8 * G + 4 * h), bt = ld4(a.beta1 + 8 * G + 4 * h);
                        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
                        float duv[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const bool ok = 8 * gl + 4 * h + p < sg.width;
                            const float xh = (xs[p] - mean1) * rstd1;
                            const float u = fmaf(xh, gmv[p], btv[p]);
                            float dsu = dxh[Gl >> 2][4 * (Gl & 3) + p] * silu_grad(u);
                            asm volatile("" : "+v"(dsu));      // computed for every lane: as `ok ? ... : 0` hipcc branched around it per element, the beta load inside the branch
                            const float du = ok ? dsu : 0.f;
                            duv[p] = du;
                            db[4 * G8 + p] = du;
                            dg[4 * G8 + p] = du * xh;
                            const float t = du * gmv[p];
                            s1 += t;
                            s2 = fmaf(t, xh, s2);
                        }
                        st4(a.du1 + ((size_t)tile * KG + G) * 256 + lane * 4, make_float4(duv[0], duv[1], duv[2], duv[3]));
                    }
                }
                colsum_store<32>(csb + 7 * NP, 16 * hb + 8 * B8, tile_colsum<32>(db, lane), lane, h);
                colsum_store<32>(csb + 7 * NP + KP, 16 * hb + 8 * B8, tile_colsum<32>(dg, lane), lane, h);
            }
        }
        s1 = xhalf_sum(s1) / wtot;
        s2 = xhalf_sum(s2) / wtot;
        f32x16 (&gg)[NT] = x;
        acc_load<NG, NT>(gg, a.gout_a + tN);
        if (a.gout_b) acc_load_add<NG, NT>(gg, a.gout_b + tN);
#pragma unroll
        for (int hb = 0; hb < KGT / 4; ++hb) {
            f32x16 dxh[4];
#pragma unroll
            for (int Gl = 0; Gl < 16; ++Gl) {
                const int G = 16 * hb + Gl;
                float o[4] = {0.f, 0.f, 0.f, 0.f};
                if (G < KG) {
                    const bool first = G < a.in0.groups;
                    const Seg& sg = first ? a.in0 : a.in1;
                    const int gl = first ? G : G - a.in0.groups;
                    const float4 xv = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
                    const float4 dv = ld4(a.du1 + ((size_t)tile * KG + G) * 256 + lane * 4);
                    const float4 gm = ld4(a.gamma1 + 8 * G + 4 * h);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, dus[4] = {dv.x, dv.y, dv.z, dv.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const bool ok = 8 * gl + 4 * h + p < sg.width;
                        const float xh = (xs[p] - mean1) * rstd1;
                        o[p] = ok ? rstd1 * (dus[p] * gmv[p] - s1 - xh * s2) : 0.f;
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) dxh[Gl >> 2][4 * (Gl & 3) + p] = o[p];
            }
            gemm.template run<NG, NT, 4>(dxh, gg, 0, true, 4 * hb);
#pragma unroll
            for (int Gl = 0; Gl < 16; ++Gl) {
                const int G = 16 * hb + Gl;
                if (G < KG) {
                    const float4 v = make_float4(dxh[Gl >> 2][4 * (Gl & 3)], dxh[Gl >> 2][4 * (Gl & 3) + 1], dxh[Gl >> 2][4 * (Gl & 3) + 2],
                                                 dxh[Gl >> 2][4 * (Gl & 3) + 3]);
                    if (G < a.in0.groups) st4(a.gin0 + ((size_t)tile * a.in0.groups + G) * 256 + lane * 4, v);
                    else st4(a.gin1 + ((size_t)tile * a.in1.groups + (G - a.in0.groups)) * 256 + lane * 4, v);
                }
            }
        }
    } else {
    f32x16 dx[KGT];
    acc_zero<KGT>(dx);
    if constexpr (HOIST) gemm.template run<NG, NT, KGT>(dx, d1, 1, false, 0, w1p);
    else gemm.template run<NG, NT, KGT>(dx, d1, 1, false);
    DSG_STAMP(HOIST && tile == 0, 0x27);
    {
        // pass 1: du, t = du*gamma, row sums (x streamed from memory, group by group); column sums of du and du*xhat per
        // block of groups
        constexpr int GB1 = KGT >= 2 ? 8 : 4;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int G0 = 0; G0 < KGT * 4; G0 += GB1) {
            float db[GB1 * 4], dg[GB1 * 4];
#pragma unroll
            for (int Gl = 0; Gl < GB1; ++Gl) {
                const int G = G0 + Gl;
#pragma unroll
                for (int p = 0; p < 4; ++p) { db[4 * Gl + p] = 0.f; dg[4 * Gl + p] = 0.f; }
                if (G < KG) {
                    const bool first = G < a.in0.groups;
                    const Seg& sg = first ? a.in0 : a.in1;
                    const int gl = first ? G : G - a.in0.groups;
                    float4 xv;
                    if constexpr (HOIST) xv = make_float4(xin[G >> 2][4 * (G & 3)], xin[G >> 2][4 * (G & 3) + 1], xin[G >> 2][4 * (G & 3) + 2], xin[G >> 2][4 * (G & 3) + 3]);
                    else xv = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
                    const float4 gm = ld4(a.gamma1 + 8 * G + 4 * h), bt = ld4(a.beta1 + 8 * G + 4 * h);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const bool ok = 8 * gl + 4 * h + p < sg.width;
                        const float xh = (xs[p] - mean1) * rstd1;
                        const float u = fmaf(xh, gmv[p], btv[p]);
                        float dsu = dx[G >> 2][4 * (G & 3) + p] * silu_grad(u);
                        asm volatile("" : "+v"(dsu));      // computed for every lane: as `ok ? ... : 0` hipcc branched around it per element, the beta load inside the branch
                        const float du = ok ? dsu : 0.f;
                        db[4 * Gl + p] = du;
                        dg[4 * Gl + p] = du * xh;
                        const float t = du * gmv[p];
                        dx[G >> 2][4 * (G & 3) + p] = t;
                        s1 += t;
                        s2 = fmaf(t, xh, s2);
                    }
                }
            }
            colsum_store<GB1 * 4>(csb + 7 * NP, G0, tile_colsum<GB1 * 4>(db, lane), lane, h);
            colsum_store<GB1 * 4>(csb + 7 * NP + KP, G0, tile_colsum<GB1 * 4>(dg, lane), lane, h);
        }
        s1 = xhalf_sum(s1) / wtot;
        s2 = xhalf_sum(s2) / wtot;
        // pass 2: dL/dx
#pragma unroll
        for (int G = 0; G < KGT * 4; ++G) {
            if (G < KG) {
                const bool first = G < a.in0.groups;
                const Seg& sg = first ? a.in0 : a.in1;
                const int gl = first ? G : G - a.in0.groups;
                float4 xv;
                if constexpr (HOIST) xv = make_float4(xin[G >> 2][4 * (G & 3)], xin[G >> 2][4 * (G & 3) + 1], xin[G >> 2][4 * (G & 3) + 2], xin[G >> 2][4 * (G & 3) + 3]);
                else xv = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool ok = 8 * gl + 4 * h + p < sg.width;
                    const float xh = (xs[p] - mean1) * rstd1;
                    const float t = dx[G >> 2][4 * (G & 3) + p];
                    dx[G >> 2][4 * (G & 3) + p] = ok ? rstd1 * (t - s1 - xh * s2) : 0.f;
                }
            }
        }
    }
    DSG_STAMP(HOIST && tile == 0, 0x28);
    // ---- shortcut: + Wsc^T g  (Linear) or + g (identity); g re-read from memory
    {
        f32x16 (&gg)[NT] = x;
        if constexpr (HOIST) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gg[nt] = gk[nt];
        } else {
            acc_load<NG, NT>(gg, a.gout_a + tN);
            if (a.gout_b) acc_load_add<NG, NT>(gg, a.gout_b + tN);
        }
        if (SCLIN) {
            if constexpr (HOIST) gemm.template run<NG, NT, KGT>(dx, gg, 0, true, 0, wscp);
            else gemm.template run<NG, NT, KGT>(dx, gg, 0, true);
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dx[nt] += gg[nt];
        }
    }
    DSG_STAMP(HOIST && tile == 0, 0x29);
    // ---- scatter dL/dx to the two input gradients
#pragma unroll
    for (int G = 0; G < KGT * 4; ++G) {
        if (G < KG) {
            const float4 v = make_float4(dx[G >> 2][4 * (G & 3)], dx[G >> 2][4 * (G & 3) + 1], dx[G >> 2][4 * (G & 3) + 2],
                                         dx[G >> 2][4 * (G & 3) + 3]);
            if (G < a.in0.groups) st4(a.gin0 + ((size_t)tile * a.in0.groups + G) * 256 + lane * 4, v);
            else st4(a.gin1 + ((size_t)tile * a.in1.groups + (G - a.in0.groups)) * 256 + lane * 4, v);
        }
    }
    }
}

template <int N, bool SCLIN>
__global__ __launch_bounds__(256) void k_resblock_bwd(const BlockBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= a.ntiles) return;
    resblock_bwd_body<N, SCLIN>(a, BwdGemmF32{a, lane}, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Plain Linear backward (data): gin = W^T gout ; with LNBWD (final layer) followed by LayerNorm+SiLU backward.
// ---------------------------------------------------------------------------------------------
struct LinBwdArgs {
    const float* gout_a;     // [tiles][NGout][256]
    const float* gout_b;     // or null
    int out_groups;          // groups of gout (N of the Linear)
    const float* WT;         // packed [OT][NGout][256]
    Seg in;                  // forward input (for LNBWD)
    const float* gamma; const float* beta;
    float* gin;              // [tiles][in.groups][256]
    float* du;               // LNBWD only
    float* rs;               // LNBWD only: (mean, rstd)
    int ntiles;
};
__device__ __forceinline__ void globalize(LinBwdArgs& a) {
    a.gout_a = as_global(a.gout_a); a.gout_b = as_global(a.gout_b); a.WT = as_global(a.WT); globalize(a.in);
    a.gamma = as_global(a.gamma); a.beta = as_global(a.beta); a.gin = as_global(a.gin); a.du = as_global(a.du); a.rs = as_global(a.rs);
}

template <int OT, bool LNBWD>
__device__ __forceinline__ void linear_bwd_body(const LinBwdArgs& a, int tile, int lane) {
    const int h = lane >> 5, j = lane & 31;
    const int NGo = a.out_groups, KG = a.in.groups;
    f32x16 dx[OT];
    acc_zero<OT>(dx);
    const size_t nt_stride = (size_t)NGo * 256;
    {
        // two operand sets, the loop unrolled by two (ping-pong, as chain_from_mem: the "current / next" form rotated its prefetch through
        // 4 OT + 4 register moves per group and copied them back at the loop end -- 725 of this kernel's 2 539 loop instructions at OT = 4)
        float4 w0[OT], w1[OT], g0, g1;
        auto load_g = [&](int g) {
            float4 gv = ld4(a.gout_a + ((size_t)tile * NGo + g) * 256 + lane * 4);
            if (a.gout_b) {
                const float4 gb = ld4(a.gout_b + ((size_t)tile * NGo + g) * 256 + lane * 4);
                gv.x += gb.x; gv.y += gb.y; gv.z += gb.z; gv.w += gb.w;
            }
            return gv;
        };
        load_wfrag<OT>(w0, a.WT + lane * 4, nt_stride);
        g0 = load_g(0);
        int g = 0;
        for (; g + 1 < NGo; g += 2) {
            load_wfrag<OT>(w1, a.WT + (size_t)(g + 1) * 256 + lane * 4, nt_stride);
            g1 = load_g(g + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group<OT>(dx, w0, g0.x, g0.y, g0.z, g0.w);
            if (g + 2 < NGo) {
                load_wfrag<OT>(w0, a.WT + (size_t)(g + 2) * 256 + lane * 4, nt_stride);
                g0 = load_g(g + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group<OT>(dx, w1, g1.x, g1.y, g1.z, g1.w);
        }
        if (g < NGo) mfma_group<OT>(dx, w0, g0.x, g0.y, g0.z, g0.w);
    }
    if (LNBWD) {
        const float2 s = reinterpret_cast<const float2*>(a.in.stats)[(size_t)tile * 32 + j];
        const float mean = s.x, rstd = rsqrtf(s.y / (float)a.in.width + kLnEps), invw = 1.0f / (float)a.in.width;
        if (h == 0) reinterpret_cast<float2*>(a.rs)[(size_t)tile * 32 + j] = make_float2(mean, rstd);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int G = 0; G < OT * 4; ++G) {
            if (G < KG) {
                const float4 xv = ld4(a.in.data + ((size_t)tile * KG + G) * 256 + lane * 4);
                const float4 gm = ld4(a.gamma + 8 * G + 4 * h), bt = ld4(a.beta + 8 * G + 4 * h);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
                float duv[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool ok = 8 * G + 4 * h + p < a.in.width;
                    const float xh = (xs[p] - mean) * rstd;
                    const float u = fmaf(xh, gmv[p], btv[p]);
                    float dsu = dx[G >> 2][4 * (G & 3) + p] * silu_grad(u);
                    asm volatile("" : "+v"(dsu));      // computed for every lane: as `ok ? ... : 0` hipcc branched around it per element, the beta load inside the branch
                    const float du = ok ? dsu : 0.f;
                    duv[p] = du;
                    const float t = du * gmv[p];
                    dx[G >> 2][4 * (G & 3) + p] = t;
                    s1 += t;
                    s2 = fmaf(t, xh, s2);
                }
                st4(a.du + ((size_t)tile * KG + G) * 256 + lane * 4, make_float4(duv[0], duv[1], duv[2], duv[3]));
            }
        }
        s1 = xhalf_sum(s1) * invw;
        s2 = xhalf_sum(s2) * invw;
#pragma unroll
        for (int G = 0; G < OT * 4; ++G) {
            if (G < KG) {
                const float4 xv = ld4(a.in.data + ((size_t)tile * KG + G) * 256 + lane * 4);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool ok = 8 * G + 4 * h + p < a.in.width;
                    const float xh = (xs[p] - mean) * rstd;
                    const float t = dx[G >> 2][4 * (G & 3) + p];
                    dx[G >> 2][4 * (G & 3) + p] = ok ? rstd * (t - s1 - xh * s2) : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int G = 0; G < OT * 4; ++G)
        if (G < KG)
            st4(a.gin + ((size_t)tile * KG + G) * 256 + lane * 4,
                make_float4(dx[G >> 2][4 * (G & 3)], dx[G >> 2][4 * (G & 3) + 1], dx[G >> 2][4 * (G & 3) + 2], dx[G >> 2][4 * (G & 3) + 3]));
}

// (two waves per SIMD as a bound: without one hipcc keeps dx in accumulation registers and moves every value in and out of them for the
// LayerNorm backward -- 657 v_accvgpr moves at OT = 4)
template <int OT, bool LNBWD>
__global__ __launch_bounds__(256, 2) void k_linear_bwd(const LinBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));  // wave-uniform: SGPR address math
    if (tile >= a.ntiles) return;
    linear_bwd_body<OT, LNBWD>(a, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradient: for every Linear,  dW[n][k] = sum_rows G[row][n] * A[row][k].
// The contraction runs over batch rows, which sit on the LANE axis of the fragment layout, so each 32-row tile of G
// and A is transposed through LDS ([feature][row] images, 36-float rows: conflict-free b32 scatter writes and b128
// reads) and fed to v_mfma_f32_32x32x2_f32 with i = out feature, j = in feature, k = row.
// One workgroup = (descriptor, block of <=128 input features, row chunk); partial results go to slab[chunk].
// ---------------------------------------------------------------------------------------------
enum { A_RAW = 0, A_LNSILU = 1, A_ONEHOT = 2 };

struct WgradDesc {
    const float* G0; const float* G1;   // gradient wrt the Linear's output (sum of two tensors if G1)
    int N, NG;                          // out features and groups of the G tensors
    int amode;
    Seg a0, a1;                         // A sources (fragment layout); a1.groups = 0 if absent
    const float* rs;                    // A_LNSILU: per-row (mean, rstd)
    const float* gamma; const float* beta;  // A_LNSILU: group-order padded
    const int* ts;                      // A_ONEHOT: per-row entry
    int onehot_n;                       // A_ONEHOT: number of entries (T)
    long long out_off;                  // offset inside a slab of dW[0][0]
    int ld;                             // row stride of dW (= Ktot)
    int KG;                             // total A groups
    int nrows;
    int gmax_slot;                      // split path: index of max|G| (tracked by k_colsum) that sets the operand scale
};
struct WgradUnit { int desc; int kblk; int chunk; int pad; };

// max|G| of a gradient tensor is tracked as one word per row tile (gmax_t[slot][tile], plain stores from the backward
// kernels, atomicMax from k_colsum's units) and reduced by k_gmax_reduce: thousands of waves updating or even just reading ONE
// word serialise on a single L2 channel for ~30 us per kernel.

constexpr int kWgLd = 36;      // floats per feature row of a 1-tile LDS image: 32 batch rows + 4 (conflict-free b128 reads)
constexpr int kWgImg = 128 * kWgLd;
constexpr int kWgTB = 4;       // small units (one out tile, <= 64 in features) take 4 row tiles per barrier interval
constexpr int kWgLdB = 32 * kWgTB + 4;
constexpr int kWgLdsFloats = 2 * 2 * kWgImg;   // 72 KiB: two (G, A) image pairs of the 1-tile form; the 4-tile form fits in it

// this wave's share of one row tile: G group `g` and A group index `gi` (relative to the unit's first A group)
__device__ __forceinline__ float4 wgrad_load_g(const WgradDesc& d, int tile, int g, int lane) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < d.NG) {
        v = ld4(d.G0 + ((size_t)tile * d.NG + g) * 256 + lane * 4);
        if (d.G1) {
            const float4 w = ld4(d.G1 + ((size_t)tile * d.NG + g) * 256 + lane * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    }
    return v;
}
__device__ __forceinline__ float4 wgrad_load_a(const WgradDesc& d, int tile, int G, int lane) {
    const int h = lane >> 5, j = lane & 31;
    float4 v;
    if (d.amode == A_ONEHOT) {
        const int row = tile * 32 + j;
        const int e = row < d.nrows ? d.ts[row] : -1;
        const int f = 8 * G + 4 * h;
        return make_float4(e == f ? 1.f : 0.f, e == f + 1 ? 1.f : 0.f, e == f + 2 ? 1.f : 0.f, e == f + 3 ? 1.f : 0.f);
    }
    const bool first = G < d.a0.groups;
    const Seg& sg = first ? d.a0 : d.a1;
    const int gl = first ? G : G - d.a0.groups;
    v = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
    if (d.amode == A_LNSILU) {
        const float2 ms = reinterpret_cast<const float2*>(d.rs)[(size_t)tile * 32 + j];
        const float4 gm = ld4(d.gamma + 8 * G + 4 * h), bt = ld4(d.beta + 8 * G + 4 * h);
        v = ln_silu4(v, ms.x, ms.y, gm, bt);
    }
    if (tile * 32 + j >= d.nrows) v = make_float4(0.f, 0.f, 0.f, 0.f);  // forward tensors of padded rows are not zero
    return v;
}
// lane (row j of tile `tt`, half h) scatters its 4 features of group g into a transposed image [feature][row]
__device__ __forceinline__ void wgrad_scatter(float* img, int ld, int g, int tt, int lane, const float4 v) {
    const int h = lane >> 5, j = lane & 31;
    float* p = img + (8 * g + 4 * h) * ld + 32 * tt + j;
    p[0] = v.x; p[ld] = v.y; p[2 * ld] = v.z; p[3 * ld] = v.w;
}

// One unit.  TB = row tiles per barrier interval; DB = double-buffered images (TB == 1) or single (TB == 4).
// MFMA: i = out feature (n-tile my_nt), j = in feature (k-tile kt), k = batch row.  Quad q covers rows 8q..8q+7: k-step s' of
// the quad pairs row 8q + s' (lane half 0) with row 8q + 4 + s' (half 1), so each lane needs 4 CONSECUTIVE rows of its
// feature: one ds_read_b128 per operand per 4 MFMAs.
template <int TB>
__device__ __forceinline__ void wgrad_unit(const WgradDesc& d, const WgradUnit& un, float* __restrict__ img, f32x16 (&acc)[4], int ntiles,
                                           int nchunks, int wave, int lane, int NTp, int KT, int g_lo, int ngr, int my_nt, int my_part,
                                           int rsplit) {
    constexpr int LD = TB == 1 ? kWgLd : kWgLdB;
    constexpr int NGW = TB == 1 ? 4 : 1 * TB;     // G float4 per wave per interval (TB==4: NTp == 1 -> 4 groups x 4 tiles / 4 waves)
    constexpr int NAW = TB == 1 ? 4 : 2 * TB;     // A float4 per wave per interval (TB==4: KT <= 2 -> 8 groups x 4 tiles / 4 waves)
    const int h = lane >> 5, j = lane & 31;
    const int gimg = TB == 1 ? kWgImg : 32 * LD;  // floats of the G image
    const int pair = TB == 1 ? 2 * kWgImg : 0;    // distance between the two buffers (TB == 1 only)
    const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    const int t_lo = un.chunk * tiles_per_chunk;
    const int t_hi = (t_lo + tiles_per_chunk < ntiles) ? t_lo + tiles_per_chunk : ntiles;
    float4 sg[NGW], sa[NAW];
    // item i of this wave: (group, tile-in-interval)
    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NGW; ++i) {
            const int item = wave + 4 * i, g = TB == 1 ? item : item % 4, tt = TB == 1 ? 0 : item / 4;
            sg[i] = (g < NTp * 4 && t0 + tt < t_hi) ? wgrad_load_g(d, t0 + tt, g, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NAW; ++i) {
            const int item = wave + 4 * i, gi = TB == 1 ? item : item % 8, tt = TB == 1 ? 0 : item / 8;
            sa[i] = (gi < KT * 4 && gi < ngr && t0 + tt < t_hi) ? wgrad_load_a(d, t0 + tt, g_lo + gi, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](float* G, float* A) {
#pragma unroll
        for (int i = 0; i < NGW; ++i) {
            const int item = wave + 4 * i, g = TB == 1 ? item : item % 4, tt = TB == 1 ? 0 : item / 4;
            if (g < NTp * 4) wgrad_scatter(G, LD, g, tt, lane, sg[i]);
        }
#pragma unroll
        for (int i = 0; i < NAW; ++i) {
            const int item = wave + 4 * i, gi = TB == 1 ? item : item % 8, tt = TB == 1 ? 0 : item / 8;
            if (gi < KT * 4) wgrad_scatter(A, LD, gi, tt, lane, sa[i]);
        }
    };
    if (t_lo < t_hi) { fetch(t_lo); stage(img, img + gimg); }
    __syncthreads();
    int buf = 0;
    for (int t0 = t_lo; t0 < t_hi; t0 += TB) {
        const float* Gimg = img + buf * pair;
        const float* Aimg = Gimg + gimg;
        const bool more = t0 + TB < t_hi;
        if (more) fetch(t0 + TB);
        const int nq = 4 * TB / rsplit, q_lo = my_part * nq;
        for (int q = q_lo; q < q_lo + nq; ++q) {
            const float4 av = *reinterpret_cast<const float4*>(Gimg + (32 * my_nt + j) * LD + 8 * q + 4 * h);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < KT) {
                    const float4 bv = *reinterpret_cast<const float4*>(Aimg + (32 * kt + j) * LD + 8 * q + 4 * h);
                    DSG_MFMA(acc[kt], av.x, bv.x);
                    DSG_MFMA(acc[kt], av.y, bv.y);
                    DSG_MFMA(acc[kt], av.z, bv.z);
                    DSG_MFMA(acc[kt], av.w, bv.w);
                }
        }
        if (TB == 1) {
            if (more) { float* nG = img + (buf ^ 1) * pair; stage(nG, nG + gimg); }
            __syncthreads();
            buf ^= 1;
        } else {
            __syncthreads();            // everyone is done reading the single image pair
            if (more) stage(img, img + gimg);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_wgrad(const WgradDesc* __restrict__ descs, const WgradUnit* __restrict__ units,
                                               float* __restrict__ slabs, size_t slab_stride, int ntiles, int nchunks) {
    __shared__ __attribute__((aligned(16))) float img[kWgLdsFloats];
    const WgradUnit un = units[blockIdx.x];
    WgradDesc d = descs[un.desc];
    d.G0 = as_global(d.G0); d.G1 = as_global(d.G1); globalize(d.a0); globalize(d.a1);          // dsg_kernels.hpp, as_global
    d.rs = as_global(d.rs); d.gamma = as_global(d.gamma); d.beta = as_global(d.beta); d.ts = as_global(d.ts);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int NT = (d.N + 31) / 32;
    const int NTp = NT <= 1 ? 1 : (NT == 2 ? 2 : 4);      // n-tiles padded to a divisor of 4
    const int rsplit = 4 / NTp;                           // waves sharing one n-tile split the row quads
    const int my_nt = wave % NTp, my_part = wave / NTp;
    const int g_lo = un.kblk * 16;
    const int g_hi = (g_lo + 16 < d.KG) ? g_lo + 16 : d.KG;
    const int ngr = g_hi - g_lo;                          // A groups handled here (<= 16)
    const int KT = (ngr + 3) / 4;

    f32x16 acc[4];
    acc_zero<4>(acc);
    // small units are latency-bound per barrier interval: give them 4 row tiles per interval
    if (NTp == 1 && KT <= 2) wgrad_unit<kWgTB>(d, un, img, acc, ntiles, nchunks, wave, lane, NTp, KT, g_lo, ngr, my_nt, my_part, rsplit);
    else wgrad_unit<1>(d, un, img, acc, ntiles, nchunks, wave, lane, NTp, KT, g_lo, ngr, my_nt, my_part, rsplit);

    // ---- waves that split the rows of one n-tile add their partial tiles through LDS, then the first writes the slab
    if (rsplit > 1) {
        float* red = img;  // <= 2 n-tiles x 4 k-tiles x 16 regs x 64 lanes = 32 KiB <= sizeof(img)
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
                // input feature of lane j in k-tile kt -> column of dW (the two A segments are padded separately)
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
                    if (col >= 0 && n < d.N) out[(size_t)n * d.ld + col] = acc[kt][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Grouped column sums over batch rows of fragment-layout tensors:
//   out[f]  = sum_rows (P0 + P1)[row][f]                          (bias gradients)
//   out2[f] = sum_rows P0[row][f] * xhat[row][f]                  (LayerNorm gamma gradient; xhat from X and rs)
// One wave per (descriptor, group, chunk); lanes accumulate their own (row, half) slice, one cross-lane reduction at
// the end.
// ---------------------------------------------------------------------------------------------
struct ColsumDesc {
    const float* P0; const float* P1;
    int groups;                 // groups of P
    Seg x0, x1;                 // xhat sources (null data when unused)
    const float* rs;
    int w0, w1;                 // true widths of the (up to two) feature segments: column mapping
    long long out_off;          // sum P      -> slab[out_off + col]   (or -1)
    long long out_off_b;        // second destination of the same sum (or -1)
    long long out2_off;         // sum P*xhat -> slab[out2_off + col]  (or -1)
    int gmax_slot;              // >= 0: also track max|P0 + P1| into gmax[slot] (operand scale of the split weight gradient)
};
struct ColsumUnit { int desc; int group; int chunk; int pad; };

__global__ __launch_bounds__(256) void k_colsum(const ColsumDesc* __restrict__ descs, const ColsumUnit* __restrict__ units, int nunits,
                                                float* __restrict__ slabs, size_t slab_stride, int ntiles, int nchunks, int nrows,
                                                unsigned* __restrict__ gmax_t, int gmax_ld) {
    const int u = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: the descriptor lives in scalar registers
    if (u >= nunits) return;
    const ColsumUnit un = units[u];
    ColsumDesc d = descs[un.desc];
    d.P0 = as_global(d.P0); d.P1 = as_global(d.P1); globalize(d.x0); globalize(d.x1); d.rs = as_global(d.rs);
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int G = un.group;
    const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    const int t_lo = un.chunk * tiles_per_chunk;
    const int t_hi = (t_lo + tiles_per_chunk < ntiles) ? t_lo + tiles_per_chunk : ntiles;
    const bool want2 = d.out2_off >= 0;
    const int g0x = d.x0.groups;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned mx = 0u;                 // max |P0 + P1| as float bits (orders like the value; NaN sorts above everything)
    const Seg& sg = G < g0x ? d.x0 : d.x1;
    const int gl = G < g0x ? G : G - g0x;
    // 4 tiles per trip, all loads issued before the first use: the wave is latency-bound otherwise
    for (int t0 = t_lo; t0 < t_hi; t0 += 4) {
        float4 p[4], w[4], xv[4];
        float2 ms[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tile = t0 + i < t_hi ? t0 + i : t_hi - 1;
            p[i] = ld4(d.P0 + ((size_t)tile * d.groups + G) * 256 + lane * 4);
            if (d.P1) w[i] = ld4(d.P1 + ((size_t)tile * d.groups + G) * 256 + lane * 4);
            if (want2) {
                xv[i] = ld4(sg.data + ((size_t)tile * sg.groups + gl) * 256 + lane * 4);
                ms[i] = reinterpret_cast<const float2*>(d.rs)[(size_t)tile * 32 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool live = t0 + i < t_hi && (t0 + i) * 32 + j < nrows;
            const float k = live ? 1.f : 0.f;
            if (want2) {
                q[0] = fmaf(k * p[i].x, (xv[i].x - ms[i].x) * ms[i].y, q[0]); q[1] = fmaf(k * p[i].y, (xv[i].y - ms[i].x) * ms[i].y, q[1]);
                q[2] = fmaf(k * p[i].z, (xv[i].z - ms[i].x) * ms[i].y, q[2]); q[3] = fmaf(k * p[i].w, (xv[i].w - ms[i].x) * ms[i].y, q[3]);
            }
            if (d.P1) { p[i].x += w[i].x; p[i].y += w[i].y; p[i].z += w[i].z; p[i].w += w[i].w; }
            s[0] = fmaf(k, p[i].x, s[0]); s[1] = fmaf(k, p[i].y, s[1]); s[2] = fmaf(k, p[i].z, s[2]); s[3] = fmaf(k, p[i].w, s[3]);
            if (live) {
                const unsigned a0 = __float_as_uint(p[i].x) & 0x7fffffffu, a1 = __float_as_uint(p[i].y) & 0x7fffffffu;
                const unsigned a2 = __float_as_uint(p[i].z) & 0x7fffffffu, a3 = __float_as_uint(p[i].w) & 0x7fffffffu;
                const unsigned m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
                const unsigned m = m01 > m23 ? m01 : m23;
                mx = mx > m ? mx : m;
            }
        }
    }
    if (d.gmax_slot >= 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)mx, o);
            mx = mx > t ? mx : t;
        }
        if (lane == 0 && mx) atomicMax(gmax_t + (size_t)d.gmax_slot * gmax_ld + (u % gmax_ld), mx);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            s[p] += __shfl_xor(s[p], o);
            q[p] += __shfl_xor(q[p], o);
        }
    if (j == 0) {
        const int gseg0 = (d.w0 + 7) / 8;
        float* out = slabs + (size_t)un.chunk * slab_stride;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int col;
            if (G < gseg0) { col = 8 * G + 4 * h + p; if (col >= d.w0) col = -1; }
            else { const int c = 8 * (G - gseg0) + 4 * h + p; col = c < d.w1 ? d.w0 + c : -1; }
            if (col >= 0) {
                if (d.out_off >= 0) out[d.out_off + col] = s[p];
                if (d.out_off_b >= 0) out[d.out_off_b + col] = s[p];
                if (want2) out[d.out2_off + col] = q[p];
            }
        }
    }
}

// Per-tile column sums of the residual blocks (written by resblock_bwd_body, one contiguous [tile][slot] region per
// block) -> per-chunk slabs, fixed order.  map[s] = (destination, second destination or -1, region base, region row
// stride) for every slot of every block; destination -1 = padding slot.
__global__ __launch_bounds__(256) void k_cs_reduce(const float* __restrict__ cs, const int4* __restrict__ map, int nslots,
                                                   float* __restrict__ slabs, size_t slab_stride, int ntiles, int nchunks) {
    const int sidx = blockIdx.x * 256 + threadIdx.x;
    if (sidx >= nslots) return;
    const int4 m = map[sidx];
    if (m.x < 0) return;
    const int c = blockIdx.y;
    const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
    const int t_lo = c * tiles_per_chunk;
    const int t_hi = (t_lo + tiles_per_chunk < ntiles) ? t_lo + tiles_per_chunk : ntiles;
    const float* p = cs + (size_t)(unsigned)m.z;
    const size_t st = (size_t)m.w;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int t = t_lo;
    for (; t + 3 < t_hi; t += 4) {
        a0 += p[(size_t)t * st];
        a1 += p[(size_t)(t + 1) * st];
        a2 += p[(size_t)(t + 2) * st];
        a3 += p[(size_t)(t + 3) * st];
    }
    for (; t < t_hi; ++t) a0 += p[(size_t)t * st];
    const float v = (a0 + a1) + (a2 + a3);
    float* out = slabs + (size_t)c * slab_stride;
    out[m.x] = v;
    if (m.y >= 0) out[m.y] = v;
}

// gmax[slot] = max over the per-tile words
__global__ __launch_bounds__(256) void k_gmax_reduce(const unsigned* __restrict__ gmax_t, int ld, unsigned* __restrict__ gmax) {
    __shared__ unsigned sm[4];
    unsigned m = 0u;
    for (int i = threadIdx.x; i < ld; i += 256) { const unsigned v = gmax_t[(size_t)blockIdx.x * ld + i]; m = m > v ? m : v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = m > t ? m : t; }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
        gmax[blockIdx.x] = a > b ? a : b;
    }
}

// grads[i] = sum_c slab[c][i]  (fixed order: deterministic)
__global__ void k_reduce_slabs(const float* __restrict__ slabs, size_t slab_stride, int nchunks, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float t = 0.f;
        for (int c = 0; c < nchunks; ++c) t += slabs[(size_t)c * slab_stride + i];
        out[i] = t;
    }
}

// The same sum over a LIST of ranges of the slab (element i of the concatenated ranges -> range by binary search over the prefix
// sums): the reduce at the end of a large training step leaves out the regions the time path writes directly
// (TimeEmbedding and the blocks' time_emb weights: ~40 % of MSR-80c's parameters, zero in every slab).
__global__ void k_reduce_ranges(const float* __restrict__ slabs, size_t slab_stride, int nchunks, float* __restrict__ out,
                                const long long* __restrict__ starts, const long long* __restrict__ prefix, int nranges, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nranges - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (prefix[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const size_t e = (size_t)(starts[lo] + (i - prefix[lo]));
        float t = 0.f;
        for (int c = 0; c < nchunks; ++c) t += slabs[(size_t)c * slab_stride + e];
        out[e] = t;
    }
}

// ... four elements per thread (16-byte accesses) where every range starts and ends on a multiple of four elements; `starts` /
// `prefix` / `total` then count float4 units
__global__ void k_reduce_ranges4(const float* __restrict__ slabs, size_t slab_stride, int nchunks, float* __restrict__ out,
                                 const long long* __restrict__ starts, const long long* __restrict__ prefix, int nranges, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nranges - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (prefix[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const size_t e = (size_t)(starts[lo] + (i - prefix[lo])) * 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < nchunks; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)c * slab_stride + e);
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4*>(out + e) = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Time-path backward on the [entries x .] tables: tiny dense products, one thread per output.
//   C[i][j] (+)= sum_l A[i*ai + l*al] * B[l*bl + j*bj]      (optional elementwise factor on A: swish'(Apre))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_gemm(const float* __restrict__ A, long long ai, long long al, const float* __restrict__ B,
                                                    long long bl, long long bj, float* __restrict__ C, long long ci, long long cj, int M, int N,
                                                    int L, int accumulate) {
    // 64 outputs per block, the contraction cut into 4 slices (one per wave) with 4 independent partial sums each: the
    // tables are tiny, so the kernel is bound by the length of one thread's dependent load chain, not by bandwidth
    __shared__ float part[4][64];
    const long long total = (long long)M * N;
    const int t = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int l0 = (int)((long long)L * sl / 4), l1 = (int)((long long)L * (sl + 1) / 4);
    for (long long base = blockIdx.x * 64LL; base < total; base += (long long)gridDim.x * 64) {
        const long long idx = base + t;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (idx < total) {
            const int jn = idx % N, im = idx / N;
            const float* a = A + im * ai;
            const float* b = B + jn * bj;
            int l = l0;
            for (; l + 3 < l1; l += 4) {
                s0 = fmaf(a[l * al], b[l * bl], s0);
                s1 = fmaf(a[(l + 1) * al], b[(l + 1) * bl], s1);
                s2 = fmaf(a[(l + 2) * al], b[(l + 2) * bl], s2);
                s3 = fmaf(a[(l + 3) * al], b[(l + 3) * bl], s3);
            }
            for (; l < l1; ++l) s0 = fmaf(a[l * al], b[l * bl], s0);
        }
        part[sl][t] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl == 0 && idx < total) {
            const int jn = idx % N, im = idx / N;
            const float v = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
            float* c = C + im * ci + jn * cj;
            *c = accumulate ? *c + v : v;
        }
        __syncthreads();
    }
}

// Time-path gradients over ALL blocks in two launches.  dTB rows (one per block output feature, `nrows_all` in total,
// each of length T) are contiguous; wt_row[r] / dst_row[r] give the matching time_emb.weight row and its gradient row.
//   d time_emb.weight[r][k] = sum_e dTB[r][e] * st[e][k]
__global__ void k_time_wgrad(const float* __restrict__ dtb, int T, const float* __restrict__ st, int td,
                             const long long* __restrict__ dst_row, float* __restrict__ G, int nrows_all) {
    const long long total = (long long)nrows_all * td;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = idx % td, r = idx / td;
        float s = 0.f;
        for (int e = 0; e < T; ++e) s = fmaf(dtb[(size_t)r * T + e], st[(size_t)e * td + k], s);
        G[dst_row[r] + k] = s;
    }
}
//   d st[e][k] = sum_r dTB[r][e] * Wt_row(r)[k]   -- rows split over blockIdx.y into partial sums (fixed order)
constexpr int kTimeChunks = 16;
__global__ void k_time_dgrad(const float* __restrict__ dtb, int T, const float* const* __restrict__ wt_row, int td,
                             float* __restrict__ part, int nrows_all) {
    const int per = (nrows_all + kTimeChunks - 1) / kTimeChunks;
    const int r_lo = blockIdx.y * per, r_hi = (r_lo + per < nrows_all) ? r_lo + per : nrows_all;
    const long long total = (long long)T * td;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = idx % td, e = idx / td;
        float s = 0.f;
        for (int r = r_lo; r < r_hi; ++r) s = fmaf(dtb[(size_t)r * T + e], wt_row[r][k], s);
        part[(size_t)blockIdx.y * total + idx] = s;
    }
}
// d_st[i] = (sum_c part[c][i]) * swish'(pre[i])
__global__ void k_time_dgrad_finish(const float* __restrict__ part, const float* __restrict__ pre, float* __restrict__ d_st, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float t = 0.f;
        for (int c = 0; c < kTimeChunks; ++c) t += part[(size_t)c * n + i];
        d_st[i] = t * silu_grad(pre[i]);
    }
}

// x[i] *= swish'(pre[i])
__global__ void k_mul_silu_grad(float* __restrict__ x, const float* __restrict__ pre, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] *= silu_grad(pre[i]);
}

// out[j] = sum_i X[i][j]  (i < M rows, row stride ld)
__global__ void k_col_sum_small(const float* __restrict__ X, int M, int N, long long ld, float* __restrict__ out) {
    for (int jn = blockIdx.x * blockDim.x + threadIdx.x; jn < N; jn += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += X[i * ld + jn];
        out[jn] = s;
    }
}

}  // namespace dsg
