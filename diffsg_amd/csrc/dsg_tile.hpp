// One launch per denoiser pass for SMALL batches: a 4-wave workgroup carries ONE 32-row tile through every operator of the U-Net.
//
// Why (round 5, profiles/r05_small_batch_kernel_stats_*.csv): BASELINE config 2 (MSR-3c, 8 192 rows, T = 1000) and the reference's own
// evaluation shape (512-row sample() calls, classifier_free_MSR.py:257, 273-279) are far below one wave per SIMD.  A reverse step
// there is 17 dependent launches, 250 us of kernel time with 8 us of gaps: the time is INSIDE the kernels, each of which is one tile's
// latency through one operator -- and begins with a dispatch, its kernel arguments and a cold round trip for statistics, inputs and
// weights (~2-3 us, seventeen times a step).  Rows are independent (UNetCF.py:318-356 has no cross-row operation), so nothing forces a
// launch boundary between operators: here the operator table of the whole net is walked inside the kernel.
//   * 64- and 128-wide blocks run the cooperative N-split body of k_resblock_c (resblock_coop_body: each wave a 32-feature slice of
//     every stage, operand images published once in LDS); a 64-wide block occupies two of the four waves, the others meet its barriers;
//   * the run of <= 32-wide operators is narrow_run_body (k_fused_narrow_h's) on the first wave, running tensor in registers, the 8-wide
//     bottom in float32 on the vector unit;
//   * the Linears outside the run (Down/Upsample at 64 / 128, final) are linear_body_h on the first wave: a handful of MFMAs each, not
//     worth a launch (they were 6-8 us apiece).
// Tensors cross operators through their fragment buffers exactly as between launches (written and re-read by the same workgroup: one
// CU, one L1; a workgroup barrier orders them), so every other kernel form reads and writes the same buffers and the per-operator
// arithmetic is bit for bit that of the separate launches.  feature_proj stays a launch of its own in front: it moves the device step
// counter on, which no workgroup of THIS kernel may do while another still reads it.
#pragma once
#include "dsg_split.hpp"

namespace dsg {

// kind of a table entry (FusedOpH::kind): 0 residual block, 1 Linear fragment -> fragment, 3 final (LayerNorm + SiLU + Linear, row-major out)
template <int NT>
__device__ __forceinline__ void tile_linear(const FusedOpH& op, const int tile, const int lane) {
    LinArgsH l = op.l;
    globalize<false>(l);
    if (op.kind == 3) linear_body_h<NT, IN_FRAG, OUT_ROWMAJOR, true, true>(l, tile, lane);
    else linear_body_h<NT, IN_FRAG, OUT_FRAG, false, true>(l, tile, lane);
}

// operators [lo, hi) outside the narrow run, one after the other
__device__ __forceinline__ void tile_wide_ops(const FusedOpH* __restrict__ ops, const int lo, const int hi, const int tile, const int wave, const int lane,
                                              uint4* __restrict__ img, float2* __restrict__ stats, float* __restrict__ vecs) {
#pragma unroll 1
    for (int i = lo; i < hi; ++i) {
        DSG_STAMP(blockIdx.x == 0 && wave == 0, 0x1000 + i);      // measurement builds only (tools/tile_stamps.py)
        const FusedOpH& op = ops[i];
        if (op.kind == 0) {
            // a wide block: all four waves (128) or the first two (64); its last barrier sits in front of the stores, the one below
            // behind them.  The record's pointers are declared global (as_global: the table is read from memory).
            BlockArgsH b = op.b;
            globalize<false>(b);
            if (op.N == 128) {
                if (op.sclin) resblock_coop_body<128, true>(b, tile, 0, wave, img, stats, vecs);
                else resblock_coop_body<128, false>(b, tile, 0, wave, img, stats, vecs);
            } else if (wave < 2) {
                if (op.sclin) resblock_coop_body<64, true>(b, tile, 0, wave, img, stats, vecs);
                else resblock_coop_body<64, false>(b, tile, 0, wave, img, stats, vecs);
            } else {
#pragma unroll
                for (int nb = 0; nb < kCoopBarriers; ++nb) __syncthreads();
            }
        } else if (wave == 0) {
            switch ((op.N + 31) >> 5) {
                case 1: tile_linear<1>(op, tile, lane); break;
                case 2: tile_linear<2>(op, tile, lane); break;
                case 3: tile_linear<3>(op, tile, lane); break;
                default: tile_linear<4>(op, tile, lane); break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // stores of this operator -> loads of the next (same workgroup: one CU, one L1)
    }
}

template <int V8NB>
__global__ __launch_bounds__(256, 2) void k_unet_tile(const FusedOpH* __restrict__ ops, const int op_lo, const int nops, const int ntiles, const int run_lo,
                                                      const int run_hi, const int v8_at, const int v8_nops, const float* __restrict__ v8_img) {
    __shared__ uint4 img[kCoopLdsU4];
    __shared__ float2 stats[4 * 32];
    __shared__ float4 vecs4[7 * 128 / 4];
    float* const vecs = reinterpret_cast<float*>(vecs4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;              // grid = ntiles
    const bool has_run = run_hi > run_lo;
    // the wide operators in front of the narrow run, the run (first wave), the wide operators behind it
    tile_wide_ops(ops, op_lo, has_run ? run_lo : nops, tile, wave, lane, img, stats, vecs);
    if (has_run) {
        DSG_STAMP(blockIdx.x == 0 && wave == 0, 0x1000 + run_lo);
        if (V8NB > 0 && v8_at >= 0) {
            // the float32 section's image (11-14 KiB) and its blocks' slices of this step's time-table row into the (now idle) image buffer:
            // the whole workgroup copies, one wave then walks the section out of LDS (the host sends launches with per-row time entries
            // to the per-operator kernels: tile_step_ok)
            constexpr int NBs = V8NB > 0 ? V8NB : 2, SZ4 = V8SecL<NBs>::SIZE / 4, TB4 = (2 * NBs + 3) * kV8TbStride / 4;
            static_assert(V8SecL<NBs>::SIZE % 4 == 0 && SZ4 + TB4 <= kCoopLdsU4, "the section's image fits the cooperative image buffer");
            const BlockArgs& b0 = ops[run_lo + v8_at + 1].b.b;
            const float* const tb = as_global(b0.tbias) + (size_t)(b0.step_ptr ? *as_global(b0.step_ptr) : 0) * b0.tb_stride;
            float4* const dst = reinterpret_cast<float4*>(img);
            for (int i = threadIdx.x; i < SZ4 + TB4; i += 256) dst[i] = i < SZ4 ? ld4(v8_img + 4 * i) : ld4(tb + 4 * (i - SZ4));
            __syncthreads();
        }
        if (wave == 0) narrow_run_body<true, V8NB, (V8NB > 0 ? 1 : 0)>(ops + run_lo, run_hi - run_lo, tile, lane, v8_at, v8_nops, v8_img, 0, reinterpret_cast<const float*>(img));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // the run's stored tensors (skips, its last output) are visible to the whole workgroup
        tile_wide_ops(ops, run_hi, nops, tile, wave, lane, img, stats, vecs);
    }
    DSG_STAMP(blockIdx.x == 0 && wave == 0, 0x1fff);
}

}  // namespace dsg
