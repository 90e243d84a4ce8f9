// What does ONE vector instruction of each kind cost in issue time on a gfx950 SIMD -- alone, and beside a stream of 32x32x16 MFMAs?
// (round 5: the 128-wide kernels are bound by instruction issue, profiles/r05_ubench_occupancy_mix.txt; which instructions are worth
// replacing -- packed float32 forms for the element-wise chain, fewer transcendentals -- depends on their cost relative to a plain FMA.)
// W waves per SIMD run the same stream: per slot [one MFMA if MF] + NV instructions of kind K on independent registers.
// Reported: cycles per slot per SIMD and the slope per instruction against the NV = 0 stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip ; run: ./valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// K: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 v_cvt_pkrtz_f16_f32, 5 v_fma_mixlo_f16, 6 v_mov_b32, 7 v_pk_mul_f32, 8 v_pk_add_f32,
//    9 v_accvgpr_read, 10 v_mul_f32 (VOP2), 11 v_fmac_f32 (VOP2), 12 v_add_f32 with an SGPR, 13 v_fmamk_f32 (literal)
template <int K, int NV, int MF>
__global__ __launch_bounds__(1024) void k_cost(int iters, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c[2] = {{0}, {0}};
    f32x4 d4[4] = {{0}, {0}, {0}, {0}};
    f32x2 p[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.001f * (threadIdx.x + i) + 1.0f; p[i] = f32x2{v[i], v[i] + 0.5f}; }
    f32x2 k1 = {0.999f, 0.998f}, k0 = {0.001f, 0.002f};
    float sg = 0.25f;
    asm volatile("s_mov_b32 %0, 0x3e800000" : "=s"(sg));
    unsigned long long sm = 0x5555555555555555ull;
    asm volatile("s_mov_b64 %0, %0" : "+s"(sm));
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (MF == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[s & 1]) : "v"(a), "v"(b));
            else if (MF == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[s & 1]) : "v"(k1[1]), "v"(k0[1]));
            else if (MF == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(d4[s & 3]) : "v"(k1[1]), "v"(k0[1]));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (s * NV + i) % 8;
                if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(k1), "v"(k0));
                else if (K == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else if (K == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[r]));
                else if (K == 4) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 5) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(k1[0]));
                else if (K == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(k1));
                else if (K == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(k0));
                else if (K == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(k1[0]));
                else if (K == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 12) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[r]) : "s"(sg));
                else if (K == 13) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(v[r]) : "v"(k0[0]));
                else if (K == 14) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[r]));
                else if (K == 15) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[r]));
                else if (K == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k0[0]));
                else if (K == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 18) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 19) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 20) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 21) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 22) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 23) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 24) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 25) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 26) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 1) % 8]));
                else if (K == 27) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 28) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 29) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 30) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 31) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
                else if (K == 32) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[r]));
                else if (K == 33) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 34) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[r]));
                else if (K == 35) asm volatile("v_mul_f32 %0, 0x3e800000, %0" : "+v"(v[r]));
                else if (K == 36) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[r]));
                else if (K == 37) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "s"(sm));
                else if (K == 38) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 39) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(v[r]) : "v"(k1[0]), "v"(k0[0]));
                else if (K == 40) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[r]) : "v"(k1[0]));
            }
            { int t; asm volatile("s_mov_b32 %0, 5" : "=s"(t)); }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 32 + wave] = t0; cyc[blockIdx.x * 32 + 16 + wave] = t1; }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
    for (int i = 0; i < 4; ++i) s += d4[i][0] + d4[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out; static long long* g_cyc;
static double median(std::vector<long long>& x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : (double)x[x.size() / 2]; }

template <int K, int NV, int MF>
double run(int W) {
    const int blocks = 256, iters = 200;
    (void)hipMemset(g_cyc, 0, blocks * 32 * sizeof(long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_cost<K, NV, MF>), dim3(blocks), dim3(256 * W), 0, 0, iters, g_out, g_cyc);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 32), m;
    (void)hipMemcpy(h.data(), g_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    for (int bI = 0; bI < blocks; ++bI) {
        long long a = h[bI * 32], e = h[bI * 32 + 16];
        for (int w = 0; w < 4 * W; ++w) { a = std::min(a, h[bI * 32 + w]); e = std::max(e, h[bI * 32 + 16 + w]); }
        m.push_back(e - a);
    }
    return median(m) / (iters * 12.0 * W);
}

template <int K>
void row(const char* name) {
    printf("%-24s", name);
    for (int W = 1; W <= 4; W *= 2) {
        const double a6 = run<K, 6, 0>(W), a12 = run<K, 12, 0>(W);
        printf("  alone W=%d: %5.2f", W, (a12 - a6) / 6.0);
    }
    for (int W = 2; W <= 4; W *= 2) {
        const double m0 = run<K, 0, 1>(W), m6 = run<K, 6, 1>(W), m12 = run<K, 12, 1>(W);
        printf("  | f16 MFMA W=%d: %5.1f %5.1f %5.1f (%5.2f / %5.2f)", W, m0, m6, m12, (m6 - m0) / 6.0, (m12 - m6) / 6.0);
    }
    {
        const int W = 2;
        const double m0 = run<K, 0, 2>(W), m6 = run<K, 6, 2>(W), m12 = run<K, 12, 2>(W);
        printf("  | f32 32x32x2 W=2: %5.1f %5.1f %5.1f (%5.2f / %5.2f)", m0, m6, m12, (m6 - m0) / 6.0, (m12 - m6) / 6.0);
        const double n0 = run<K, 0, 3>(W), n6 = run<K, 6, 3>(W), n12 = run<K, 12, 3>(W);
        printf("  | f32 16x16x4 W=2: %5.1f %5.1f %5.1f (%5.2f / %5.2f)", n0, n6, n12, (n6 - n0) / 6.0, (n12 - n6) / 6.0);
    }
    printf("\n");
}

int main() {
    (void)hipMalloc(&g_out, 256 * 1024 * sizeof(float));
    (void)hipMalloc(&g_cyc, 256 * 32 * sizeof(long long));
    printf("cycles of SIMD issue time per vector instruction (slope between 6 and 12 per slot); beside an MFMA per slot: slot cycles at 0 / 6 / 12 instructions per slot (slope per instruction 0->6 / 6->12)\n");
    row<0>("v_fma_f32");
    row<1>("v_pk_fma_f32");
    row<7>("v_pk_mul_f32");
    row<8>("v_pk_add_f32");
    row<10>("v_mul_f32 (VOP2)");
    row<11>("v_fmac_f32 (VOP2)");
    row<13>("v_fmamk_f32 (literal)");
    row<12>("v_add_f32 sgpr");
    row<2>("v_exp_f32");
    row<3>("v_rcp_f32");
    row<4>("v_cvt_pkrtz_f16_f32");
    row<5>("v_fma_mixlo_f16");
    row<6>("v_mov_b32");
    row<9>("v_accvgpr_read_b32");
    row<14>("v_cvt_f32_f16");
    row<15>("v_cvt_f16_f32");
    row<20>("v_cvt_pk_f16_f32 (gfx950)");
    row<16>("v_sub_f32");
    row<38>("v_fma_f32 neg modifier");
    row<39>("v_fma_f32 abs modifier");
    row<35>("v_mul_f32 literal");
    row<36>("v_add_f32 inline const");
    row<30>("v_max_f32");
    row<25>("v_med3_f32");
    row<31>("v_ldexp_f32");
    row<32>("v_rndne_f32");
    row<34>("v_rsq_f32");
    row<17>("v_and_b32");
    row<28>("v_add_u32");
    row<29>("v_lshl_add_u32");
    row<19>("v_perm_b32");
    row<18>("v_pack_b32_f16");
    row<21>("v_pk_fma_f16");
    row<40>("v_pk_add_f16");
    row<22>("v_fma_mix_f32");
    row<23>("v_dot2_f32_f16");
    row<33>("v_dot2c_f32_f16");
    row<24>("v_cndmask_b32 vcc");
    row<37>("v_cndmask_b32 sgpr pair");
    row<26>("v_permlane32_swap_b32");
    row<27>("v_mov_b32 dpp row_shr");
    return 0;
}
