// Label generator of the MSR problem on the device: "LRH gradient descent" of the sum rate
//   max sum_c log2(1 + g_c * p_c)   s.t.  sum_c p_c = W
// Reference: utils/dataset_generate.py:247-255 (SUM_RATE_GRAD), :257-278 (alpha_calc), :280-313 (SUM_RATE_GEN); float64 as there.
//
// One wave per problem instance (row), M <= 128 channels, two per lane.  alpha_calc walks the channels by decreasing
// |grad|; here every channel finds the sum of the |grad| that precede it in that order (ties: lower index first) by one
// pass over the row's values in LDS -- O(M^2) compares per row and iteration, no sort -- and reads its alpha off that
// prefix: +sign before the running sum reaches half of the total, the fractional remainder on the crossing channel,
// -sign after it.  The reference's loop condition (ANY row with mean |grad| > eps) is a device flag chain: iteration n
// runs iff flag[n] is set and sets flag[n + 1]; no host round trip in the 149 iterations.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace dsg {

constexpr int kSrMaxM = 128;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void k_sumrate_init(double* __restrict__ schemes, long long n, double v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) schemes[i] = v;
}

// dry != 0: only evaluate the loop condition of the current schemes (the reference's check before the first iteration)
__global__ __launch_bounds__(256) void k_sumrate_iter(const double* __restrict__ gs, double* __restrict__ schemes, long long rows, int M,
                                                      double beta, double eps, const int* __restrict__ flag_in, int* __restrict__ flag_out,
                                                      int dry) {
    __shared__ double sg[4][kSrMaxM];
    __shared__ int any_sm;
    if (*flag_in == 0) return;                       // the reference's while loop has ended
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) any_sm = 0;
    __syncthreads();
    const long long r = blockIdx.x * 4LL + wave;
    const bool live = r < rows;
    const double ln2 = 0.693147180559945309417232121458;   // np.log(2)
    double g[2], s[2], gr[2], ga[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q;
        const bool ok = live && e < M;
        g[q] = ok ? gs[r * M + e] : 0.0;
        s[q] = ok ? schemes[r * M + e] : 0.0;
        gr[q] = ok ? g[q] / ((g[q] * s[q] + 1.0) * ln2) : 0.0;
        ga[q] = fabs(gr[q]);
        sg[wave][e] = ga[q];
    }
    const double total = wave_sum_d(ga[0] + ga[1]);
    if (live && lane == 0 && total / (double)M > eps) any_sm = 1;      // benign race: every writer stores 1
    if (!dry && live) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            if (e < M) {
                double before = 0.0;
                for (int k = 0; k < M; ++k) {
                    const double v = sg[wave][k];
                    if (v > ga[q] || (v == ga[q] && k < e)) before += v;
                }
                const double sgn = gr[q] > 0 ? 1.0 : -1.0, half = total / 2;
                double alpha;
                if (before + ga[q] < half) alpha = sgn;
                else if (before < half) alpha = (total - ga[q] - 2 * before) / ga[q] * sgn;
                else alpha = -sgn;
                schemes[r * M + e] = s[q] + beta * alpha * gr[q];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && any_sm && __hip_atomic_load(flag_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag_out, 1);
}

// rates[r] = sum_c log2(1 + schemes * gs)   (dataset_generate.py:312)
__global__ __launch_bounds__(256) void k_sumrate_rates(const double* __restrict__ gs, const double* __restrict__ schemes,
                                                       double* __restrict__ rates, long long rows, int M) {
    const int lane = threadIdx.x & 63;
    const long long r = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (r >= rows) return;
    double acc = 0.0;
    for (int e = lane; e < M; e += 64) acc += log2(1.0 + schemes[r * M + e] * gs[r * M + e]);
    acc = wave_sum_d(acc);
    if (lane == 0) rates[r] = acc;
}

}  // namespace dsg
