// Solution decoders and objective evaluators of the three problems, on the device.
//
// Reference: classifier_free_MSR.py:239-245 (custom_decoder), :287-288 (sum rate); classifier_free_CO.py:255-278 (cost_calc),
// :281-290 (customized_real_decoder); classifier_free_NU.py:267-276 (custom_decoder), :279-303 (rate_calc).
//
// All of them are row-local except the two global (min, max) pairs of the MSR / NU decoders, which are reduced in a fixed
// order (per-block partials, then every consumer folds the partials the same way): deterministic.  One wave per row with
// the lanes striding the columns where a row is wide (softmax rows), one thread per row where a row is a handful of
// scalars (cost / rate).  They are HBM-bound streaming kernels: algorithmic bytes = inputs read once + outputs written once.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace dsg {

constexpr int kEvalParts = 1024;   // at most this many partial (min, max) pairs in a global reduction (+1 slot for the result)

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// (min, max) over columns [c0, c1) of every row: per-block partials (waves stride the rows, lanes the columns), then one
// block folds the partials in a fixed order into part[0]
__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ y, long long rows, int D, int c0, int c1,
                                                        float2* __restrict__ part) {
    __shared__ float smin[4], smax[4];
    float lo = INFINITY, hi = -INFINITY;
    const int lane = threadIdx.x & 63;
    // lpr lanes per row (a power of two <= 64 fitted to the column range): narrow ranges put many rows in one wave
    const int w = c1 - c0;
    const int lpr = w >= 64 ? 64 : (w >= 16 ? 16 : (w >= 4 ? 4 : 1)), rpw = 64 / lpr;
    const int sub = lane & (lpr - 1), rsub = lane / lpr;
    for (long long r = (blockIdx.x * 4LL + (threadIdx.x >> 6)) * rpw + rsub; r < rows; r += (long long)gridDim.x * 4 * rpw)
        for (int c = c0 + sub; c < c1; c += lpr) {
            const float v = y[r * D + c];
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
    lo = wave_min_f(lo); hi = wave_max_f(hi);
    if (lane == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0)
        part[1 + blockIdx.x] = make_float2(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
}
__global__ __launch_bounds__(256) void k_minmax_final(float2* __restrict__ part, int nparts) {
    __shared__ float smin[4], smax[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { const float2 p = part[1 + i]; lo = fminf(lo, p.x); hi = fmaxf(hi, p.y); }
    lo = wave_min_f(lo); hi = wave_max_f(hi);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0)
        part[0] = make_float2(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
}

// out[r][:] = softmax_row(f(y[r][:])) with f = identity (MODE 0), or (y - lo)/(hi - lo) with the GLOBAL lo, hi (MODE 1: MSR),
// or MODE 2 (CO): identity softmax, rows whose entries are all < -10 become zero.
// LPR lanes share a row (1, 4, 16 or 64: the host picks the smallest that leaves <= kSoftEpl elements per lane), so a
// wave covers 64 / LPR rows and a row is read once into registers; the row reductions are xor-shuffles inside the LPR lanes.
constexpr int kSoftEpl = 16;
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int MODE, int LPR>
__global__ __launch_bounds__(256) void k_row_softmax(const float* __restrict__ y, float* __restrict__ out, long long rows, int D,
                                                     const float2* __restrict__ part) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const long long r = (blockIdx.x * 4LL + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = r < rows;
    float lo = 0.f, span = 1.f;
    if (MODE == 1) { const float2 mm = part[0]; lo = mm.x; span = mm.y - mm.x; }
    const float* yr = y + (live ? r : 0) * D;
    float v[kSoftEpl];
    float m = -INFINITY, ymax = -INFINITY;
#pragma unroll
    for (int k = 0; k < kSoftEpl; ++k) {
        const int c = sub + k * LPR;
        v[k] = -INFINITY;
        if (c < D) {
            const float raw = yr[c];
            v[k] = MODE == 1 ? (raw - lo) / span : raw;
            if (MODE == 2) ymax = fmaxf(ymax, raw);
        }
        m = fmaxf(m, v[k]);
    }
    m = group_max<LPR>(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kSoftEpl; ++k) {
        v[k] = (sub + k * LPR < D) ? expf(v[k] - m) : 0.f;
        s += v[k];
    }
    s = group_sum<LPR>(s);
    const bool dead = MODE == 2 && group_max<LPR>(ymax) < -10.0f;
    if (live) {
#pragma unroll
        for (int k = 0; k < kSoftEpl; ++k) {
            const int c = sub + k * LPR;
            if (c < D) out[r * D + c] = dead ? 0.f : v[k] / s;
        }
    }
}
// rows wider than 64 * kSoftEpl: one wave per row, the row streamed three times
template <int MODE>
__global__ __launch_bounds__(256) void k_row_softmax_wide(const float* __restrict__ y, float* __restrict__ out, long long rows, int D,
                                                          const float2* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const long long r = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (r >= rows) return;
    float lo = 0.f, span = 1.f;
    if (MODE == 1) { const float2 mm = part[0]; lo = mm.x; span = mm.y - mm.x; }
    const float* yr = y + r * D;
    float m = -INFINITY, ymax = -INFINITY;
    for (int c = lane; c < D; c += 64) {
        const float v = MODE == 1 ? (yr[c] - lo) / span : yr[c];
        m = fmaxf(m, v);
        if (MODE == 2) ymax = fmaxf(ymax, yr[c]);
    }
    m = wave_max_f(m);
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = MODE == 1 ? (yr[c] - lo) / span : yr[c];
        s += expf(v - m);
    }
    s = wave_sum_f(s);
    const bool dead = MODE == 2 && wave_max_f(ymax) < -10.0f;
    for (int c = lane; c < D; c += 64) {
        const float v = MODE == 1 ? (yr[c] - lo) / span : yr[c];
        out[r * D + c] = dead ? 0.f : expf(v - m) / s;
    }
}

// MSR objective: rate[r] = sum_c log2(1 + p[r][c] * g[r][c])   (MSR.py:287-288)
template <int LPR>
__global__ __launch_bounds__(256) void k_msr_rate(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ out,
                                                  long long rows, int D) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const long long r = (blockIdx.x * 4LL + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = r < rows;
    const long long base = (live ? r : 0) * D;
    float s = 0.f;
    for (int c = sub; c < D; c += LPR) s += log2f(1.0f + p[base + c] * g[base + c]);
    s = group_sum<LPR>(s);
    if (live && sub == 0) out[r] = s;
}

// CO objective (CO.py:255-278): nodes with Y > 0.1 are offloaded and share the unallocated remainder equally;
// cost = sum_nodes (1-D)*local + D*(transition + exec / share).  X row = [local, transition, exec] per node.
__global__ __launch_bounds__(256) void k_co_cost(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ out,
                                                 long long rows, int n) {
    const long long r = blockIdx.x * 256LL + threadIdx.x;
    if (r >= rows) return;
    const float* y = Y + r * n;
    const float* x = X + r * 3 * n;
    float ysum = 0.f;
    int dsum = 0;
    for (int i = 0; i < n; ++i)
        if (y[i] > 0.1f) { ysum += y[i]; ++dsum; }
    const float dden = dsum == 0 ? 0.00001f : (float)dsum;
    const float spread = (1.0f - ysum) / dden;
    float cost = 0.f;
    for (int i = 0; i < n; ++i) {
        const bool off = y[i] > 0.1f;
        const float share = y[i] + spread;
        cost += off ? x[3 * i + 1] + x[3 * i + 2] / share : x[3 * i];   // (1 - D) * local + D * (trans + exec / share), D in {0, 1}
    }
    out[r] = cost;
}

// NU decoder (NU.py:267-276): columns 0, 1 = UAV position, min-max scaled with the GLOBAL (lo, hi) of those two columns and
// stretched to the area; columns 2.. = power split, softmax * P_sum.
__global__ __launch_bounds__(256) void k_nu_decode(const float* __restrict__ y, float* __restrict__ out, long long rows, int D,
                                                   float width, float height, float p_sum, const float2* __restrict__ part) {
    // a row is 2 + K scalars (K <= a few dozen users): one thread per row
    const long long r = blockIdx.x * 256LL + threadIdx.x;
    if (r >= rows) return;
    const float2 mm = part[0];
    const float* yr = y + r * D;
    float m = -INFINITY;
    for (int c = 2; c < D; ++c) m = fmaxf(m, yr[c]);
    float s = 0.f;
    for (int c = 2; c < D; ++c) s += expf(yr[c] - m);
    for (int c = 2; c < D; ++c) out[r * D + c] = expf(yr[c] - m) / s * p_sum;
    out[r * D] = (yr[0] - mm.x) / (mm.y - mm.x) * width;
    out[r * D + 1] = (yr[1] - mm.x) / (mm.y - mm.x) * height;
}

// NU objective (NU.py:279-303): NOMA successive interference cancellation.  Users ordered by channel gain, strongest
// first; rank 0 sees only noise, rank r sees the summed power of ranks < r.  K = D - 2 <= 32 users, one thread per row.
constexpr int kNuMaxUsers = 32;
__global__ __launch_bounds__(256) void k_nu_rate(const float* __restrict__ Yd, const float* __restrict__ X, float* __restrict__ out,
                                                 long long rows, int K) {
    const long long r = blockIdx.x * 256LL + threadIdx.x;
    if (r >= rows) return;
    const float sigma_sq = 110.f, rou_0 = 60.f, H = 150.f;
    const float* yd = Yd + r * (K + 2);
    const float* x = X + r * 2 * K;
    float hg[kNuMaxUsers];
    int ord[kNuMaxUsers];
    for (int k = 0; k < K; ++k) {
        const float dx = x[2 * k] - yd[0], dy = x[2 * k + 1] - yd[1];
        hg[k] = sqrtf(rou_0 / (H * H + dx * dx + dy * dy));
        ord[k] = k;
    }
    // stable insertion sort, descending gain (the reference sorts -h ascending)
    for (int i = 1; i < K; ++i) {
        const int oi = ord[i];
        const float hi = hg[oi];
        int j = i - 1;
        while (j >= 0 && hg[ord[j]] < hi) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = oi;
    }
    float rate = 0.f, prev = 0.f;
    for (int rk = 0; rk < K; ++rk) {
        const int u = ord[rk];
        const float pw = yd[2 + u], h2 = hg[u] * hg[u];
        float sinr;
        if (rk == 0) sinr = pw * h2 / sigma_sq;
        else { prev += yd[2 + ord[rk - 1]]; sinr = pw / (prev + sigma_sq / h2); }
        rate += log2f(1.0f + sinr);
    }
    out[r] = rate;
}

}  // namespace dsg
