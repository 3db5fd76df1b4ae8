// How fast can a kernel READ a 268 MB KV-sized buffer once on gfx950, by load flavour?  (tuning aid)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_probe tools/stream_probe.hip && gpurun_out/stream_probe
// Variants: (a) 16 x global_load_dwordx4 per lane into registers, non-temporal (what attn_scores / attn_pv do: one
// 16 KiB page per wave), (b) the same without nt, (c) LDS-DMA (global_load_lds_dwordx4, 16 KiB per wave into LDS, then one
// ds_read per piece), nt and default policy.  Each wave reads one 16 KiB "page"; grid = pages / 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_regs(const u32x4_t* __restrict__ p, unsigned* __restrict__ out, size_t pages) {
    const size_t pg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pg >= pages) return;
    const u32x4_t* q = p + pg * 1024 + (threadIdx.x & 63);
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = NT ? __builtin_nontemporal_load(q + j * 64) : q[j * 64];
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    if (s == 0x12345678u) out[0] = s;
}

template <int AUX>
__global__ __launch_bounds__(256) void k_dma(const u32x4_t* __restrict__ p, unsigned* __restrict__ out, size_t pages) {
    __shared__ __attribute__((aligned(16))) u32x4_t buf[4][1024];          // 16 KiB per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t pg = (size_t)blockIdx.x * 4 + wave;
    if (pg >= pages) return;
    const u32x4_t* q = p + pg * 1024 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(q + j * 64),
                                         (void __attribute__((address_space(3)))*)&buf[wave][j * 64], 16, 0, AUX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const u32x4_t v = buf[wave][j * 64 + lane]; s ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) out[0] = s;
}

int main() {
    const size_t bytes = 268435456, pages = bytes / 16384, copies = 6;
    std::vector<u32x4_t*> bufs(copies);
    for (auto& b : bufs) { CK(hipMalloc((void**)&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
    unsigned* out; CK(hipMalloc((void**)&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (size_t i = 0; i < copies; ++i) launch(bufs[i]);
        hipDeviceSynchronize();
        const int iters = 60;
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch(bufs[i % copies]);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.2f us per 268 MB  %6.2f TB/s\n", name, ms * 1000 / iters, bytes / (ms / iters * 1e-3) / 1e12);
        return 0;
    };
    const dim3 grid((unsigned)(pages / 4)), blk(256);
    run("registers, nt", [&](u32x4_t* b) { hipLaunchKernelGGL(k_regs<true>, grid, blk, 0, nullptr, b, out, pages); });
    run("registers, default policy", [&](u32x4_t* b) { hipLaunchKernelGGL(k_regs<false>, grid, blk, 0, nullptr, b, out, pages); });
    run("LDS-DMA, default policy", [&](u32x4_t* b) { hipLaunchKernelGGL(k_dma<0>, grid, blk, 0, nullptr, b, out, pages); });
    run("LDS-DMA, nt (aux 2)", [&](u32x4_t* b) { hipLaunchKernelGGL(k_dma<2>, grid, blk, 0, nullptr, b, out, pages); });
    CK(hipGetLastError());
    return 0;
}
