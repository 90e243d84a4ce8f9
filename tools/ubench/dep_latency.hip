// How much instruction-level parallelism does ONE wave need to issue vector instructions at the SIMD's full rate, and what do W waves of a
// purely dependent stream reach together?  (round 6: the narrow run runs 4 waves per SIMD, each a dependent chain, at ~50 % of the vector
// issue rate -- profiles/r06_*.)  Stream: v_fma_f32 (K=0), v_exp_f32 (K=1), v_pk_fma_f32 (K=2), v_fma_mixlo_f16 (K=3), v_permlane32_swap (K=4)
// on D independent registers in rotation (dependency distance D), W waves per SIMD.  Reported: SIMD cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o dep_latency dep_latency.hip ; run: ./dep_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K, int D>
__global__ __launch_bounds__(1024) void k_dep(int iters, float* out, long long* cyc) {
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.001f * (threadIdx.x + i) + 1.0f; p[i] = f32x2{v[i], v[i] + 0.5f}; }
    const float k1 = 0.999f, k0 = 0.001f;
    const f32x2 q1 = {0.999f, 0.998f}, q0 = {0.001f, 0.002f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            const int r = s % D;
            if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1), "v"(k0));
            else if (K == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
            else if (K == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(q1), "v"(q0));
            else if (K == 3) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2" : "+v"(v[r]) : "v"(k1), "v"(k0));
            else if (K == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 4) % 8]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K, int D>
double run(int W, float* out, long long* cyc, int cus) {
    const int iters = 2000;
    hipLaunchKernelGGL((k_dep<K, D>), dim3(cus), dim3(256 * W), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_dep<K, D>), dim3(cus), dim3(256 * W), 0, 0, iters, out, cyc);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, cyc, sizeof(long long) * cus, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < cus; ++i) m += (double)h[i];
    m /= cus;
    return m / ((double)iters * 48 * W);       // SIMD cycles per instruction (W waves on each SIMD issue 48 * iters each)
}
template <int K> void row(const char* name, float* out, long long* cyc, int cus) {
    printf("%-22s", name);
    for (int W = 1; W <= 4; W *= 2)
        printf(" | W=%d: D=1 %6.2f  D=2 %6.2f  D=4 %6.2f  D=8 %6.2f", W, run<K, 1>(W, out, cyc, cus), run<K, 2>(W, out, cyc, cus), run<K, 4>(W, out, cyc, cus),
               run<K, 8>(W, out, cyc, cus));
    printf("\n");
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * 1024 * cus); hipMalloc(&cyc, sizeof(long long) * cus);
    printf("SIMD cycles per instruction (s_memtime ticks; one workgroup of 4 W waves per CU = W waves per SIMD), dependency distance D\n");
    row<0>("v_fma_f32", out, cyc, cus);
    row<1>("v_exp_f32", out, cyc, cus);
    row<2>("v_pk_fma_f32", out, cyc, cus);
    row<3>("v_fma_mixlo_f16", out, cyc, cus);
    row<4>("v_permlane32_swap", out, cyc, cus);
    return 0;
}
