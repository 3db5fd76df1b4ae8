// Do VALU instructions run under the shadow of MFMAs on gfx950 -- across two waves of one SIMD, and inside one wave?
// (tuning aid for the codec GEMM epilogue: its GELU / split arithmetic is as many SIMD cycles as its MFMAs)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvp tools/mfma_valu_probe.hip && /tmp/mvp
// Roles per wave: M = 4 independent chains of v_mfma_f32_32x32x16_bf16, V = 16 independent chains of v_fma_f32 (or packed
// v_pk_fma_f32, or v_exp_f32).  One block per CU; 4 waves = one per SIMD, 8 waves = two per SIMD (wave w and w+4 share one).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int VK>   // 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32
__device__ __forceinline__ void valu_step(float (&v)[16], float a, float b) {
    if (VK == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    } else if (VK == 1) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            f32x2_t t = {v[i], v[i + 1]};
            t = t * f32x2_t{a, a} + f32x2_t{b, b};
            v[i] = t.x; v[i + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    }
}

// mode 0: every wave M; 1: every wave V; 2: waves 0-3 M, waves 4-7 V; 3: every wave does both, interleaved in source
template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    f32x16_t acc[4];
    float v[16];
    bf16x8_t x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * threadIdx.x); y[i] = (__bf16)(0.002f * i); }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[c], 0, 0, 0);
                // a quarter of the VALU step behind each MFMA
                if (VK == 0) {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i) v[i] = __builtin_fmaf(v[i], a, b);
                } else if (VK == 1) {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; i += 2) { f32x2_t t = {v[i], v[i + 1]}; t = t * f32x2_t{a, a} + f32x2_t{b, b}; v[i] = t.x; v[i + 1] = t.y; }
                } else {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
                }
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[c], 0, 0, 0);
        }
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) valu_step<VK>(v, a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int VK>
static int run(const char* name, int threads, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(threads), 0, nullptr, out, iters, 1.0001f, 0.0001f);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-72s %8.1f us  (%.1f ns per iteration: 4 MFMA = 128 clk, 16 VALU)\n", name, best * 1000, best * 1e6 / iters);
    fflush(stdout);
    return 0;
}


// one wave per SIMD, KV VALU instructions behind every MFMA; DEP: they form ONE dependent chain instead of 16
template <int KV, bool DEP>
__global__ __launch_bounds__(256) void k2(float* out, int iters, float a, float b) {
    f32x16_t acc[4];
    float v[16];
    bf16x8_t x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * threadIdx.x); y[i] = (__bf16)(0.002f * i); }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[c], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KV; ++i) {
                const int r = DEP ? 0 : (c * KV + i) & 15;
                v[r] = __builtin_fmaf(v[r], a, b);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, KV, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}
template <int KV, bool DEP>
static int run2(float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((k2<KV, DEP>), dim3(256), dim3(256), 0, nullptr, out, iters, 1.0001f, 0.0001f);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("1 wave/SIMD, %2d %s v_fma_f32 behind each MFMA: %6.1f ns per 4 MFMA\n", KV, DEP ? "dependent  " : "independent", best * 1e6 / iters);
    fflush(stdout);
    return 0;
}

int main() {
    float* out; CK(hipMalloc((void**)&out, 4));
    run<0, 0>("MFMA only, 1 wave/SIMD", 256, out);
    run<0, 0>("MFMA only, 2 waves/SIMD", 512, out);
    run<1, 0>("v_fma_f32 only, 1 wave/SIMD", 256, out);
    run<1, 0>("v_fma_f32 only, 2 waves/SIMD", 512, out);
    run<2, 0>("one MFMA wave + one v_fma_f32 wave per SIMD", 512, out);
    run<3, 0>("one wave per SIMD, MFMA and v_fma_f32 interleaved", 256, out);
    run<3, 0>("two waves per SIMD, each MFMA and v_fma_f32 interleaved", 512, out);
    run<1, 1>("v_pk_fma_f32 only, 1 wave/SIMD", 256, out);
    run<2, 1>("one MFMA wave + one v_pk_fma_f32 wave per SIMD", 512, out);
    run<3, 1>("one wave per SIMD, MFMA and v_pk_fma_f32 interleaved", 256, out);
    run<1, 2>("v_exp_f32 only, 1 wave/SIMD", 256, out);
    run<2, 2>("one MFMA wave + one v_exp_f32 wave per SIMD", 512, out);
    run<3, 2>("one wave per SIMD, MFMA and v_exp_f32 interleaved", 256, out);
    run2<2, false>(out); run2<4, false>(out); run2<6, false>(out); run2<7, false>(out); run2<8, false>(out); run2<10, false>(out); run2<12, false>(out);
    run2<2, true>(out); run2<4, true>(out); run2<6, true>(out); run2<8, true>(out);
    return 0;
}
