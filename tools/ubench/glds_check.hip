// Unit check of the LDS-DMA form dsg_wide.hpp relies on: `global_load_lds_dwordx4 voff, s[base:base+1] offset:1024`
// -- SGPR base + per-lane VGPR byte offset, and the instruction offset applied to BOTH the global address and the LDS
// address (second 1 KiB piece of a pair).  Each wave copies 2 KiB from global into LDS at a wave-specific slot, waits with a
// counted vmcnt, barriers, then every wave reads ALL slots back and writes them out; the host compares.
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_check glds_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void glds_pair(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst) {
    __shared__ uint4 ring[10 * 512];   // 80 KiB: the two chunks used sit at 56..72 KiB (the LDS address crosses 64 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)ring;
    // chunk 0: wave w copies pieces 2w, 2w+1 of src block 0; chunk 1: of src block 1 (+8 KiB)
    for (int c = 0; c < 2; ++c) {
        const uint4* base = src + (size_t)blockIdx.x * 1024 + c * 512 + wave * 128;
        glds_pair((unsigned)lane * 16u, base, lds0 + (unsigned)((7 + c) * 8192 + wave * 2048));
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // chunk 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    uint4 a[8];
    for (int p = 0; p < 8; ++p) a[p] = ring[7 * 512 + p * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint4 b[8];
    for (int p = 0; p < 8; ++p) b[p] = ring[8 * 512 + p * 64 + lane];
    if (wave == (blockIdx.x & 3)) {
        for (int p = 0; p < 8; ++p) {
            dst[(size_t)blockIdx.x * 1024 + p * 64 + lane] = a[p];
            dst[(size_t)blockIdx.x * 1024 + 512 + p * 64 + lane] = b[p];
        }
    }
}

int main() {
    const int blocks = 1024;
    const size_t n = (size_t)blocks * 1024;
    std::vector<uint4> h(n), o(n);
    for (size_t i = 0; i < n; ++i) h[i] = make_uint4((unsigned)i, (unsigned)(i * 7 + 1), (unsigned)(i ^ 0x5a5a5a5a), (unsigned)(i >> 3));
    uint4 *s, *d;
    hipMalloc(&s, n * sizeof(uint4)); hipMalloc(&d, n * sizeof(uint4));
    hipMemcpy(s, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
    hipMemset(d, 0, n * sizeof(uint4));
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, s, d);
    hipMemcpy(o.data(), d, n * sizeof(uint4), hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i)
        if (o[i].x != h[i].x || o[i].y != h[i].y || o[i].z != h[i].z || o[i].w != h[i].w) { if (bad < 5) printf("mismatch at %zu: got %u want %u\n", i, o[i].x, h[i].x); ++bad; }
    printf("glds_check: %zu mismatches of %zu\n", bad, n);
    return bad != 0;
}
