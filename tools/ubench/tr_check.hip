// ds_read_b64_tr_b16 against the [32 rows][128 features] f16 image of k_wgrad_h (dsg_train_split.hpp: wimg_off, wimg_wofs, wimg_rbase):
// 256-byte rows, 16-byte chunks XOR-swizzled (cdna_hip_programming.md T10, image (b)).  A lane of the fragment layout (row j = lane & 31,
// half h = lane >> 5, group g) stores its four features 8g + 4h .. + 3 as ONE 8-byte write; the MFMA operand of k16-step s, feature
// tile T is two transposed reads: lane (i = lane & 31, h) receives feature 32T + i of rows 16s + 8h + 0..7.
// Build: hipcc --offload-arch=gfx950 -O3 -o tr_check tr_check.hip ; run: ./tr_check  (prints the number of wrong elements: 0)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v lds_s4;
__device__ __forceinline__ unsigned wimg_off(unsigned row, unsigned ch) { return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u))); }
__global__ void k(int* bad, float* dump) {
    __shared__ __attribute__((aligned(16))) char img[8192];
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // two passes (f16 holds integers up to 2048 exactly): mode 0: value(row, feature) = (row & 15) * 128 + feature; mode 1: value = row
    const int li = lane & 15, q = li >> 2, p = li & 3, sub = (lane >> 4) & 1;
    int nbad = 0;
    for (int mode = 0; mode < 2; ++mode) {
    __syncthreads();
    for (int g = 0; g < 16; ++g) {
        _Float16 v[4];
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)(float)(mode ? j : (j & 15) * 128 + 8 * g + 4 * h + e);
        *reinterpret_cast<uint2*>(img + wimg_off(j, g) + 8 * h) = *reinterpret_cast<uint2*>(v);
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s)
        for (int T = 0; T < 4; ++T) {
            _Float16 got[8];
            for (int e = 0; e < 2; ++e) {
                const unsigned row = 16 * s + 8 * h + 4 * e + q;
                const unsigned addr = wimg_off(row, 4 * T + 2 * sub + (p >> 1)) + 8 * (p & 1);
                const s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + addr));
                *reinterpret_cast<s4v*>(got + 4 * e) = r;
            }
            for (int e = 0; e < 8; ++e) {
                const int row = 16 * s + 8 * h + e, feat = 32 * T + (lane & 31);
                const float want = (float)(mode ? row : (row & 15) * 128 + feat);
                if ((float)got[e] != want) ++nbad;
                if (mode == 0 && s == 0 && T == 1) dump[lane * 8 + e] = (float)got[e];
            }
        }
    }
    atomicAdd(bad, nbad);
}
int main() {
    int* bad; float* dump; hipMalloc(&bad, 4); hipMalloc(&dump, 64 * 8 * 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, bad, dump);
    int hb = -1; float hd[512]; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hd, dump, sizeof hd, hipMemcpyDeviceToHost);
    printf("wrong elements: %d of 8192\n", hb);
    if (hb) for (int l = 0; l < 64; l += 9) { printf("lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" %6.0f", hd[l * 8 + e]); printf("\n"); }
    return hb != 0;
}
