// Throughput of packed-f32 VALU on gfx950: v_pk_fma_f32 against v_fma_f32 (one wave per SIMD, independent chains).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkf32 tools/ubench/pkf32.hip && /tmp/pkf32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, float a, float b) {
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 A = {a, a}, B = {b, b};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {        // 2 scalar fmas
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].y) : "v"(a), "v"(b));
            } else if (MODE == 1) { // 1 packed fma
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(A), "v"(B));
            } else if (MODE == 2) { // 2 scalar muls
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(a));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].y) : "v"(a));
            } else if (MODE == 3) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(A));
            } else if (MODE == 4) { // exp
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].y));
            } else if (MODE == 5) { // cvt_pkrtz
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(v[i].y));
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i].y) : "v"(a));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    const char* names[] = {"2x v_fma_f32", "1x v_pk_fma_f32", "2x v_mul_f32", "1x v_pk_mul_f32", "2x v_exp_f32", "2x v_cvt_pkrtz"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0001f, 0.5f); break;
            }
            hipDeviceSynchronize();
        }
        long long c[4];
        hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
        printf("%-18s per pair of values: %.2f counter ticks (x %d pairs per iteration)\n", names[mode], (double)c[0] / kIters / 8, 8);
    }
    // tick length: s_memtime runs at 100 MHz on gfx9; compare ratios only
    return 0;
}
