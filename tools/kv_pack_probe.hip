// Would a LOSSLESS 12-bit K page (sign + 7-bit mantissa byte, 4-bit exponent offset below a per-row base) make the decode
// attention's score pass faster?  (sizing aid for DESIGN.md section 9; the engine stores bf16 pages)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kvp tools/kv_pack_probe.hip && /tmp/kvp
// Both kernels do the score pass's work for 16 384 pages of 64 tokens x 128 dims (B=32 at 4k, 8 kv heads, one layer):
// a wave per page, a lane per token, q . k over 128 dims with v_dot2_f32_bf16, one bf16 score per token written.
//   plain : page = [d/8][token][8 bf16]                      16 x 16-byte loads per lane   (16 KiB per page)
//   packed: mantissa plane [d/16][token][16 x u8], exponent plane [d/32][token][32 x u4], base [token] u8
//           8 + 4 x 16-byte loads per lane + 1 byte                                      (12.06 KiB per page)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

struct Q { unsigned w[64]; };      // the query's 128 dims as bf16 pairs (uniform)

__global__ __launch_bounds__(256) void k_plain(const u32x4_t* __restrict__ pages, Q q, unsigned short* __restrict__ out, int npages) {
    const int pg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pg >= npages) return;
    const u32x4_t* p = pages + (size_t)pg * 1024 + lane;
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + j * 64);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        acc = dot2(v[j].x, q.w[4 * j], acc); acc = dot2(v[j].y, q.w[4 * j + 1], acc);
        acc = dot2(v[j].z, q.w[4 * j + 2], acc); acc = dot2(v[j].w, q.w[4 * j + 3], acc);
    }
    out[(size_t)pg * 64 + lane] = (unsigned short)(__builtin_bit_cast(unsigned, acc) >> 16);
}

// two values (mantissa bytes in bits 0-7 / 16-23 of P, exponent offsets in bits 0-3 / 16-19 of N2) -> bf16 pair
__device__ __forceinline__ unsigned unpack2(unsigned P, unsigned N2, unsigned base2) {
    const unsigned e = (base2 - N2) << 7;                         // per half: base - offset, into the exponent field
    return (P & 0x007f007fu) | ((P & 0x00800080u) << 8) | e;
}

__global__ __launch_bounds__(256) void k_packed(const u32x4_t* __restrict__ mant, const u32x4_t* __restrict__ expo,
                                                const unsigned char* __restrict__ base, Q q, unsigned short* __restrict__ out, int npages) {
    const int pg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pg >= npages) return;
    const u32x4_t* pm = mant + (size_t)pg * 512 + lane;           // 8 x 64 x 16 B
    const u32x4_t* pe = expo + (size_t)pg * 256 + lane;           // 4 x 64 x 16 B
    u32x4_t m[8], x[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = __builtin_nontemporal_load(pm + j * 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(pe + j * 64);
    const unsigned b = base[(size_t)pg * 64 + lane];
    const unsigned base2 = b | (b << 16);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                  // 16 values per mantissa vector, 8 bytes of offsets
        const unsigned mw[4] = {m[j].x, m[j].y, m[j].z, m[j].w};
        const unsigned ew[2] = {(j & 1) ? x[j >> 1].z : x[j >> 1].x, (j & 1) ? x[j >> 1].w : x[j >> 1].y};
#pragma unroll
        for (int c = 0; c < 4; ++c) {                              // 4 values per mantissa dword, 4 nibbles = 16 bits
            const unsigned M = mw[c], N = (ew[c >> 1] >> (16 * (c & 1))) & 0xffffu;
            const unsigned P0 = __builtin_amdgcn_perm(0u, M, 0x0c010c00u);            // [m0, 0, m1, 0]
            const unsigned P1 = __builtin_amdgcn_perm(0u, M, 0x0c030c02u);            // [m2, 0, m3, 0]
            const unsigned N0 = (N & 0xfu) | ((N & 0xf0u) << 12), N1 = ((N >> 8) & 0xfu) | ((N & 0xf000u) << 4);
            acc = dot2(unpack2(P0, N0, base2), q.w[8 * j + 2 * c], acc);
            acc = dot2(unpack2(P1, N1, base2), q.w[8 * j + 2 * c + 1], acc);
        }
    }
    out[(size_t)pg * 64 + lane] = (unsigned short)(__builtin_bit_cast(unsigned, acc) >> 16);
}


// 13-bit variant, lossless in practice: 5-bit exponent offset = nibble plane + one bit plane [token][16 B]; planes
// pre-arranged for cheap extraction (mantissa dword = [m0, m2, m1, m3]; nibble dword = values [0,2,4,6,1,3,5,7]; bit dword =
// values [0,2,..,30 | 1,3,..,31]), so a pair costs and / shift-and / shift-or only.  G query heads share the page (GQA).
template <int G>
__global__ __launch_bounds__(256) void k_packed13(const u32x4_t* __restrict__ mant, const u32x4_t* __restrict__ expo,
                                                  const u32x4_t* __restrict__ bits, const unsigned char* __restrict__ base, Q q,
                                                  unsigned short* __restrict__ out, int npages) {
    const int pg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pg >= npages) return;
    const u32x4_t* pm = mant + (size_t)pg * 512 + lane;
    const u32x4_t* pe = expo + (size_t)pg * 256 + lane;
    u32x4_t m[8], x[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = __builtin_nontemporal_load(pm + j * 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(pe + j * 64);
    const u32x4_t bw = __builtin_nontemporal_load(bits + (size_t)pg * 64 + lane);
    const unsigned b = base[(size_t)pg * 64 + lane];
    const unsigned base2 = b | (b << 16);
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
    const unsigned bwv[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                  // 16 values: 4 mantissa dwords, 2 nibble dwords, half a bit dword
        const unsigned mw[4] = {m[j].x, m[j].y, m[j].z, m[j].w};
        const unsigned ew[2] = {(j & 1) ? x[j >> 1].z : x[j >> 1].x, (j & 1) ? x[j >> 1].w : x[j >> 1].y};
        const unsigned bd = bwv[j >> 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                          // pair h of mantissa dword c = pair (2c + h) of these 16 values
                const int k = 2 * c + h;                           // pair index 0..7 within the 16 values
                const unsigned P = (mw[c] >> (8 * h)) & 0x00ff00ffu;
                const unsigned N2 = (ew[k >> 2] >> (4 * (k & 3))) & 0x000f000fu;
                const unsigned B2 = (bd >> (8 * (j & 1) + k)) & 0x00010001u;
                const unsigned off = N2 | (B2 << 4);
                const unsigned w = (P & 0x007f007fu) | ((P & 0x00800080u) << 8) | ((base2 - off) << 7);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = dot2(w, q.w[(8 * j + k + 17 * g) & 63], acc[g]);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) s += acc[g];
    out[(size_t)pg * 64 + lane] = (unsigned short)(__builtin_bit_cast(unsigned, s) >> 16);
}
template <int G>
__global__ __launch_bounds__(256) void k_plain_g(const u32x4_t* __restrict__ pages, Q q, unsigned short* __restrict__ out, int npages) {
    const int pg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pg >= npages) return;
    const u32x4_t* p = pages + (size_t)pg * 1024 + lane;
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + j * 64);
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[g] = dot2(v[j].x, q.w[(4 * j + 17 * g) & 63], acc[g]); acc[g] = dot2(v[j].y, q.w[(4 * j + 1 + 17 * g) & 63], acc[g]);
            acc[g] = dot2(v[j].z, q.w[(4 * j + 2 + 17 * g) & 63], acc[g]); acc[g] = dot2(v[j].w, q.w[(4 * j + 3 + 17 * g) & 63], acc[g]);
        }
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) s += acc[g];
    out[(size_t)pg * 64 + lane] = (unsigned short)(__builtin_bit_cast(unsigned, s) >> 16);
}

int main() {
    const int npages = 16384, copies = 6;
    std::vector<u32x4_t*> plain(copies), mant(copies), expo(copies);
    std::vector<unsigned char*> base(copies);
    for (int i = 0; i < copies; ++i) {
        CK(hipMalloc((void**)&plain[i], (size_t)npages * 16384)); CK(hipMemset(plain[i], 0x3c, (size_t)npages * 16384));
        CK(hipMalloc((void**)&mant[i], (size_t)npages * 8192)); CK(hipMemset(mant[i], 0x35, (size_t)npages * 8192));
        CK(hipMalloc((void**)&expo[i], (size_t)npages * 4096)); CK(hipMemset(expo[i], 0x21, (size_t)npages * 4096));
        CK(hipMalloc((void**)&base[i], (size_t)npages * 64)); CK(hipMemset(base[i], 0x7e, (size_t)npages * 64));
    }
    unsigned short* out; CK(hipMalloc((void**)&out, (size_t)npages * 64 * 2));
    Q q; for (int i = 0; i < 64; ++i) q.w[i] = 0x3c003c00u + i;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double mb, auto launch) {
        for (int i = 0; i < copies; ++i) launch(i);
        hipDeviceSynchronize();
        const int iters = 60;
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch(i % copies);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %7.2f us per layer pass  (%.0f MB -> %.2f TB/s)\n", name, ms * 1000 / iters, mb, mb / (ms / iters * 1e-3) / 1e6);
        return 0;
    };
    const dim3 grid(npages / 4), blk(256);
    run("bf16 pages (the engine's format)", npages * 16384 / 1e6, [&](int i) { hipLaunchKernelGGL(k_plain, grid, blk, 0, nullptr, plain[i], q, out, npages); });
    run("12-bit pages, unpacked in registers", npages * (8192 + 4096 + 64) / 1e6,
        [&](int i) { hipLaunchKernelGGL(k_packed, grid, blk, 0, nullptr, mant[i], expo[i], base[i], q, out, npages); });
    std::vector<u32x4_t*> bitp(copies);
    for (int i = 0; i < copies; ++i) { CK(hipMalloc((void**)&bitp[i], (size_t)npages * 1024)); CK(hipMemset(bitp[i], 0x11, (size_t)npages * 1024)); }
    run("bf16 pages, 2 query heads (GQA)", npages * 16384 / 1e6, [&](int i) { hipLaunchKernelGGL(k_plain_g<2>, grid, blk, 0, nullptr, plain[i], q, out, npages); });
    run("13-bit pages, 1 query head", npages * (8192 + 4096 + 1024 + 64) / 1e6,
        [&](int i) { hipLaunchKernelGGL(k_packed13<1>, grid, blk, 0, nullptr, mant[i], expo[i], bitp[i], base[i], q, out, npages); });
    run("13-bit pages, 2 query heads (GQA)", npages * (8192 + 4096 + 1024 + 64) / 1e6,
        [&](int i) { hipLaunchKernelGGL(k_packed13<2>, grid, blk, 0, nullptr, mant[i], expo[i], bitp[i], base[i], q, out, npages); });
    CK(hipGetLastError());
    return 0;
}
