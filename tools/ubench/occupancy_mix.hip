// How much of a wave's mixed MFMA + vector stream do MORE WAVES PER SIMD hide?  (round 5: the question behind an N-split / 16-row
// form of the 128-wide kernels, which would run 4 waves per SIMD at <= 128 registers instead of 2 at ~230.)
// One workgroup per CU (LDS-limited), W waves per SIMD (256 * W threads).  Every wave runs the same stream: per MFMA slot
// NV plain vector instructions (every 6th a transcendental), NL ds_read_b128 (one counted wait per slot), one SALU move.
// Reported: cycles per MFMA slot per SIMD (aggregate over the SIMD's waves; 32 = matrix pipe saturated for 32x32x16, 16 for 16x16x32).
//   mode 0: v_mfma_f32_32x32x16_f16     mode 1: v_mfma_f32_16x16x32_f16
// Second table: role split -- R waves per SIMD issue only MFMAs, V waves per SIMD only vector instructions: aggregate vector
// instruction rate beside a saturated matrix pipe.
// Build: hipcc --offload-arch=gfx950 -O3 -o occupancy_mix occupancy_mix.hip ; run: ./occupancy_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV, int NL>
__global__ __launch_bounds__(1024) void k_mix(int iters, float* out, long long* cyc) {
    extern __shared__ uint4 lds[];
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c[2] = {{0}, {0}};
    f32x4 d[4] = {{0}, {0}, {0}, {0}};
    float v[12];
    uint4 q[2] = {};
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + i) + 1.0f;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_uint4(1, 2, 3, 4);
    const unsigned lds_addr = (unsigned)(threadIdx.x & 63) * 16u;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            if (MODE == 0 || MODE == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[s & 1]) : "v"(a), "v"(b));
            else if (MODE == 2 || MODE == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[s & 1]) : "v"(a), "v"(b));
            else if (MODE == 5) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[s & 1]) : "v"(v[11]), "v"(v[10]));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d[s & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (s * NV + i) % 10;
                if (i % 6 == 5 && (MODE < 3 || MODE == 5)) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(0.999f), "v"(0.001f));
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i & 1]) : "v"(lds_addr), "n"(1024 * i));
            if (NL) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NL) : "memory");
            { int t; asm volatile("s_mov_b32 %0, 5" : "=s"(t)); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 32 + wave] = t0; cyc[blockIdx.x * 32 + 16 + wave] = t1; }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 2; ++i) s += (float)q[i].x;
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
    for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// role split: waves with (wave / 4) < R issue MFMAs only, the others vector instructions only (KIND 0 v_fma, 1 v_exp, 2 mix 5:1)
template <int MODE, int KIND>
__global__ __launch_bounds__(1024) void k_roles(int R, int iters, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x & 63) * 0.01f + i); b[i] = (_Float16)(i * 0.25f - 1.f); }
    f32x16 c[2] = {{0}, {0}};
    f32x4 d[4] = {{0}, {0}, {0}, {0}};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (threadIdx.x + i) + 1.0f;
    const bool mf = (wave >> 2) < R;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        for (int it = 0; it < 4 * iters; ++it) {          // outlasts the vector waves: they run beside a busy matrix pipe throughout
#pragma unroll
            for (int s = 0; s < 48; ++s) {
                if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[s & 1]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d[s & 3]) : "v"(a), "v"(b));
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 96; ++i) {
                const int r = i % 12;
                if (KIND == 1 || (KIND == 2 && i % 6 == 5)) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(0.999f), "v"(0.001f));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 32 + wave] = t0; cyc[blockIdx.x * 32 + 16 + wave] = t1; }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
    for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out; static long long* g_cyc;
static double median(std::vector<long long>& x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : (double)x[x.size() / 2]; }

template <int MODE, int NV, int NL>
void run_mix() {
    const int blocks = 256, iters = 200;
    (void)hipFuncSetAttribute((const void*)k_mix<MODE, NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const char* mn[] = {"32x32x16", "16x16x32", "32x32x16 acc in AGPRs", "32x32x16 fma only", "32x32x16 AGPR acc, fma only", "32x32x2 f32 (AGPR acc)"};
    printf("%s  %d vector + %d ds_read_b128 per MFMA slot:", mn[MODE], NV, NL);
    for (int W = 1; W <= 4; ++W) {
        (void)hipMemset(g_cyc, 0, blocks * 32 * sizeof(long long));
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_mix<MODE, NV, NL>), dim3(blocks), dim3(256 * W), 96 * 1024, 0, iters, g_out, g_cyc);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(blocks * 32), m;
        (void)hipMemcpy(h.data(), g_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int bI = 0; bI < blocks; ++bI) {            // span of the workgroup: first start to last end (older waves win arbitration and finish early)
            long long a = h[bI * 32], e = h[bI * 32 + 16];
            for (int w = 0; w < 4 * W; ++w) { a = std::min(a, h[bI * 32 + w]); e = std::max(e, h[bI * 32 + 16 + w]); }
            m.push_back(e - a);
        }
        // a SIMD's W waves issued W * iters * 12 slots in that span
        printf("  W=%d %6.1f", W, median(m) / (iters * 12.0 * W));
    }
    printf("   cycles per MFMA slot per SIMD\n");
}

template <int MODE, int KIND>
void run_roles() {
    const int blocks = 256, iters = 200;
    const char* kn[] = {"v_fma_f32", "v_exp_f32", "5 fma : 1 exp"};
    for (int R = 0; R <= 1; ++R)
        for (int V = 1; V <= 3; ++V) {
            const int W = R + V;
            (void)hipMemset(g_cyc, 0, blocks * 32 * sizeof(long long));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_roles<MODE, KIND>), dim3(blocks), dim3(256 * W), 0, 0, R, iters, g_out, g_cyc);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(blocks * 32), m, o;
            (void)hipMemcpy(h.data(), g_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            for (int bI = 0; bI < blocks; ++bI) {
                long long a = h[bI * 32], em = 0, eo = 0;
                for (int w = 0; w < 4 * W; ++w) { a = std::min(a, h[bI * 32 + w]); long long& e = (w >> 2) < R ? em : eo; e = std::max(e, h[bI * 32 + 16 + w]); }
                if (R) m.push_back(em - a);
                o.push_back(eo - a);
            }
            printf("%s  %-14s  %d MFMA-only + %d vector-only waves per SIMD: %6.1f cycles per MFMA | %5.2f cycles per vector instruction per SIMD (aggregate)\n",
                   MODE ? "16x16x32" : "32x32x16", kn[KIND], R, V, R ? median(m) / (4.0 * iters * 48.0) : 0.0, median(o) / (iters * 96.0 * V));
        }
}

int main() {
    (void)hipMalloc(&g_out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&g_cyc, 256 * 32 * sizeof(long long));
    run_mix<5, 0, 0>(); run_mix<5, 3, 0>(); run_mix<5, 5, 0>(); run_mix<5, 6, 0>(); run_mix<5, 12, 0>(); run_mix<5, 24, 0>();
    run_mix<0, 0, 0>(); run_mix<0, 3, 0>(); run_mix<0, 6, 0>(); run_mix<0, 6, 1>(); run_mix<0, 6, 2>(); run_mix<0, 8, 1>(); run_mix<0, 10, 1>();
    run_mix<0, 12, 0>(); run_mix<0, 24, 0>();
    run_mix<3, 6, 0>(); run_mix<3, 12, 0>(); run_mix<3, 24, 0>();
    run_mix<2, 6, 0>(); run_mix<2, 6, 1>(); run_mix<2, 12, 0>(); run_mix<4, 6, 0>(); run_mix<4, 12, 0>(); run_mix<4, 24, 0>();
    run_mix<1, 0, 0>(); run_mix<1, 3, 0>(); run_mix<1, 3, 1>(); run_mix<1, 4, 1>(); run_mix<1, 5, 1>();
    run_roles<0, 0>(); run_roles<0, 2>();
    return 0;
}
