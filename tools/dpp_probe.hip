// Do the DPP controls give lane ^ 1, ^ 2, ^ 4, ^ 8 (the partners of the xor-butterfly in common.h: wave_sum) and a full
// 64-lane max in lane 63?   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp tools/dpp_probe.hip && /tmp/dpp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = dpp_i<0xB1>(l, l);                                     // quad_perm [1,0,3,2]
    out[1 * 64 + l] = dpp_i<0x4E>(l, l);                                     // quad_perm [2,3,0,1]
    { int t = dpp_i<0x104, 0xf, 0x5>(l, l); out[2 * 64 + l] = dpp_i<0x114, 0xf, 0xa>(t, l); }   // row_shl:4 banks 0,2 / row_shr:4 banks 1,3
    out[3 * 64 + l] = dpp_i<0x128>(l, l);                                    // row_ror:8
    int v = (l * 37 + 11) % 64;                                              // a permutation: max = 63 somewhere
    v = max(v, dpp_i<0xB1>(v, v)); v = max(v, dpp_i<0x4E>(v, v));
    v = max(v, dpp_i<0x141>(v, v)); v = max(v, dpp_i<0x140>(v, v));          // row_half_mirror, row_mirror
    v = max(v, dpp_i<0x142, 0xa>(v, v)); v = max(v, dpp_i<0x143, 0xc>(v, v));   // row_bcast15 rows 1,3; row_bcast31 rows 2,3
    out[4 * 64 + l] = v;
}
int main() {
    int* d; hipMalloc((void**)&d, 5 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, nullptr, d);
    int h[5 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int masks[4] = {1, 2, 4, 8};
    int bad = 0;
    for (int m = 0; m < 4; ++m) for (int l = 0; l < 64; ++l) if (h[m * 64 + l] != (l ^ masks[m])) { if (bad < 8) printf("xor %d lane %d got %d\n", masks[m], l, h[m * 64 + l]); ++bad; }
    if (h[4 * 64 + 63] != 63) { printf("max in lane 63 = %d\n", h[4 * 64 + 63]); ++bad; }
    printf(bad ? "DPP probe: %d mismatches\n" : "DPP probe: ok (xor 1, 2, 4, 8 partners; 64-lane max in lane 63)\n", bad);
    return bad != 0;
}
