// XY_Tokenizer decode path (RVQ codes -> 24 kHz waveform) in fp32 on gfx950.
//
// Replaces XY_Tokenizer.inference_detokenize (reference XY_Tokenizer/xy_tokenizer/model.py:104-128):
//   quantizer.decode_codes      nn/quantizer.py:345-364
//   post_rvq_adapter            nn/modules.py:568-640  (Transformer)
//   upsample                    nn/modules.py:502-515  (ConvTranspose1d k=s=4 == GEMM)
//   acoustic_decoder            nn/modules.py:386-423
//   enhanced_vocos              nn/modules.py:1398-1410,1135-1154 (backbone), :959-988,:737-792 (ISTFT head)
// The codec checkpoint is fp32 and is never cast by the reference, and the parity
// bar is waveform RMS <= 1e-4, so every contraction runs on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32): compute bound, 157 TFLOP/s peak.
//
// Activations are token-major [rows = (window, frame)][channels]; every Linear /
// Conv1d(k=1) / ConvTranspose1d is one GEMM with a fused epilogue.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mtts.h"
#include "common.h"

// ------------------------------------------------------------------------------------
// fp32 GEMM: C[M,N] = epi(A[M,K] * W^T), W is [N][K] (B_KN=false) or [K][N] (B_KN=true).
// 128x128 block tile, BK=16, 4 waves x (64x64) = 2x2 MFMA 32x32 tiles each.
// ------------------------------------------------------------------------------------
struct GemmF32Args {
    const float* A; const float* W; float* C;
    const float* bias; const float* gamma; const float* res;
    int M, N, K;
    long lda, ldw, ldc, ldres;
    int res_rows;          // residual row = m % res_rows (positional embedding); 0 = m
    float scale;           // applied after bias
    int act;               // 0 none, 1 GELU (erf)
    int batch_inner;       // z -> (zo = z / batch_inner, zi = z % batch_inner)
    long sAo, sAi, sWo, sWi, sCo, sCi;
    // pre-split operands / output (gemm_b3t_kernel): bf16 hi and lo planes in MFMA-fragment order (common.h xpack_off / wpack_off)
    const uint16_t *Ahi, *Alo, *Whi, *Wlo;
    uint16_t *Chi, *Clo;
};

#define GT 128
#define GK 16
#define GLD (GT + 4)

__constant__ int g_xcd_map = 1;      // MTTS_CODEC_XCD=0 switches the mapping below off (A/B measurements)
// XCD-aware block -> tile mapping for the 128 x 128-tile GEMMs.  Workgroups are dealt round-robin over the 8 XCDs (each
// with its own 4 MiB L2), so with the plain (blockIdx.x, blockIdx.y) order the ~64 blocks resident on one XCD work on
// tiles scattered over the whole output: every operand tile is pulled through 8 different L2s and the GEMM ends up bound
// by Infinity-Cache traffic (measured: 3.1 GB of operand fetches for a 24 000 x 4096 x 512 product = 5 TB/s, i.e. the
// 32 flop/B of a 128^2 fp32 tile, not the matrix cores).  Here XCD x = (linear id % 8) walks its own contiguous share
// of the tiles in an order where consecutive tiles form panels of 8 tile-rows (M) swept along N: the blocks resident on
// an XCD at any time cover ~8 x 8 tiles and share their operand rows in that XCD's L2.  Bijective for any grid.
__device__ __forceinline__ void xcd_tile(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int T = tiles_m * tiles_n, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int x = L & 7, sidx = L >> 3, q = T >> 3, r = T & 7;
    const int gidx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + sidx;      // position in the panel-major order
    constexpr int SM = 8;
    const int p = gidx / (SM * tiles_n);
    const int rows = min(SM, tiles_m - p * SM);
    const int idx = gidx - p * SM * tiles_n;
    tn = idx / rows;
    tm = p * SM + idx % rows;
}

template <bool B_KN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args g) {
    // double-buffered LDS tiles; the next tile is fetched into registers while the current one feeds the MFMAs
    __shared__ float As[2][GK][GLD];
    __shared__ float Ws[2][GK][GLD];
    const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
    const float* A = g.A + zo * g.sAo + zi * g.sAi;
    const float* W = g.W + zo * g.sWo + zi * g.sWi;
    float* C = g.C + zo * g.sCo + zi * g.sCi;
    int tm_ = blockIdx.y, tn_ = blockIdx.x;
    if (gridDim.z == 1 && g_xcd_map) xcd_tile((g.M + GT - 1) / GT, (g.N + GT - 1) / GT, tm_, tn_);
    const int m0 = tm_ * GT, n0 = tn_ * GT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2][4], rw[2][4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3;
            const int m = m0 + row, k = k0 + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[i][j] = 0.f;
            if (m < g.M) {
                const float* p = A + (long)m * g.lda + k;
                if (k + 3 < g.K) { const float4 t = *(const float4*)p; ra[i][0] = t.x; ra[i][1] = t.y; ra[i][2] = t.z; ra[i][3] = t.w; }
                else { for (int j = 0; j < 4; ++j) if (k + j < g.K) ra[i][j] = p[j]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) rw[i][j] = 0.f;
            if (!B_KN) {
                const int row = idx >> 2, kq = idx & 3;
                const int n = n0 + row, k = k0 + 4 * kq;
                if (n < g.N) {
                    const float* p = W + (long)n * g.ldw + k;
                    if (k + 3 < g.K) { const float4 t = *(const float4*)p; rw[i][0] = t.x; rw[i][1] = t.y; rw[i][2] = t.z; rw[i][3] = t.w; }
                    else { for (int j = 0; j < 4; ++j) if (k + j < g.K) rw[i][j] = p[j]; }
                }
            } else {
                const int kr = idx >> 5, nq = idx & 31;
                const int k = k0 + kr, n = n0 + 4 * nq;
                if (k < g.K) {
                    const float* p = W + (long)k * g.ldw + n;
                    if (n + 3 < g.N) { const float4 t = *(const float4*)p; rw[i][0] = t.x; rw[i][1] = t.y; rw[i][2] = t.z; rw[i][3] = t.w; }
                    else { for (int j = 0; j < 4; ++j) if (n + j < g.N) rw[i][j] = p[j]; }
                }
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][4 * kq + j][row] = ra[i][j];
            if (!B_KN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Ws[buf][4 * kq + j][row] = rw[i][j];
            } else {
                const int kr = idx >> 5, nq = idx & 31;
#pragma unroll
                for (int j = 0; j < 4; ++j) Ws[buf][kr][4 * nq + j] = rw[i][j];
            }
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
        const bool more = k0 + GK < g.K;
        if (more) fetch(k0 + GK);
#pragma unroll
        for (int ks = 0; ks < GK / 2; ++ks) {
            const int kk = 2 * ks + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][kk][wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Ws[cur][kk][wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // ---- epilogue: D[row m][col n], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
            const float gm = g.gamma ? g.gamma[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = (acc[i][j][r] + bv) * g.scale;
                if (g.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                v *= gm;
                if (g.res) v += g.res[(long)(g.res_rows ? m % g.res_rows : m) * g.ldres + n];
                C[(long)m * g.ldc + n] = v;
            }
        }
}

// ------------------------------------------------------------------------------------
// The same GEMM at 3 bf16 MFMAs per product tile ("bf16x3"): every fp32 operand is split on its way into LDS into
// hi = bf16(x) and lo = bf16(x - hi), and  a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulates in fp32 on
// v_mfma_f32_32x32x16_bf16 (the dropped terms are below 2^-16 relative).  The exact-f32 MFMA runs at 1/16 of the
// bf16 rate, so this is ~5x less matrix-core time per tile; the decoder's waveform moves by 1.5e-6 RMS against the
// reference (tolerance 1e-4; measured on the fixtures, oracle/ numerics emulation and the GPU tests).  Used for the
// decode direction only: the encoder's code ids come from an argmin and keep the exact kernel.
// W is [N][K].  LDS planes [k half][row][8 bf16]: a lane's MFMA operand (8 consecutive k of one row) is one 16-byte read.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {x, y};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);            // v_cvt_pk_bf16_f32 (RNE)
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// NJ = 32-column tiles per wave: 2 -> 128 x 128 block (3 blocks per CU; the one in use), 4 -> 128 x 256 block.
// GELU(erf) for the bf16x3 kernel's epilogue: erf by Abramowitz-Stegun 7.1.26 with the hardware exp (|error| < 7e-7,
// GELU within 4e-7 absolute: two orders below the kernel's own product error).  libm's erff costs ~30 instructions per
// element and was 38 % of the 512 -> 4096 Vocos GEMM; the exact kernel keeps it.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = v * 0.70710678118654752440f, ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * __expf(-ax * ax);
    return 0.5f * v * (1.0f + copysignf(e, x));
}

template <int NJ>
__global__ __launch_bounds__(256, 2) void gemm_b3_kernel(GemmF32Args g) {
    constexpr int BN = 64 * NJ;                  // block columns
    __shared__ __attribute__((aligned(16))) uint16_t Ah[2][2][GT][8], Al[2][2][GT][8], Wh[2][2][BN][8], Wl[2][2][BN][8];
    const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
    const float* A = g.A + zo * g.sAo + zi * g.sAi;
    const float* W = g.W + zo * g.sWo + zi * g.sWi;
    float* C = g.C + zo * g.sCo + zi * g.sCi;
    int tm_ = blockIdx.y, tn_ = blockIdx.x;
    if (gridDim.z == 1 && g_xcd_map && BN == GT) xcd_tile((g.M + GT - 1) / GT, (g.N + GT - 1) / GT, tm_, tn_);
    const int m0 = tm_ * GT, n0 = tn_ * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16_t acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2][4], rw[NJ][4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3;
            const int k = k0 + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[i][j] = 0.f;
            if (m0 + row < g.M) {
                const float* p = A + (long)(m0 + row) * g.lda + k;
                if (k + 3 < g.K) { const float4 t = *(const float4*)p; ra[i][0] = t.x; ra[i][1] = t.y; ra[i][2] = t.z; ra[i][3] = t.w; }
                else { for (int j = 0; j < 4; ++j) if (k + j < g.K) ra[i][j] = p[j]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3;
            const int k = k0 + 4 * kq;
#pragma unroll
            for (int j = 0; j < 4; ++j) rw[i][j] = 0.f;
            if (n0 + row < g.N) {
                const float* p = W + (long)(n0 + row) * g.ldw + k;
                if (k + 3 < g.K) { const float4 t = *(const float4*)p; rw[i][0] = t.x; rw[i][1] = t.y; rw[i][2] = t.z; rw[i][3] = t.w; }
                else { for (int j = 0; j < 4; ++j) if (k + j < g.K) rw[i][j] = p[j]; }
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3, kh = kq >> 1, off = 4 * (kq & 1);
            uint32_t h0, h1, l0, l1;
            split2(ra[i][0], ra[i][1], h0, l0);
            split2(ra[i][2], ra[i][3], h1, l1);
            *(u32x2_t*)&Ah[buf][kh][row][off] = u32x2_t{h0, h1};
            *(u32x2_t*)&Al[buf][kh][row][off] = u32x2_t{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 2, kq = idx & 3, kh = kq >> 1, off = 4 * (kq & 1);
            uint32_t h0, h1, l0, l1;
            split2(rw[i][0], rw[i][1], h0, l0);
            split2(rw[i][2], rw[i][3], h1, l1);
            *(u32x2_t*)&Wh[buf][kh][row][off] = u32x2_t{h0, h1};
            *(u32x2_t*)&Wl[buf][kh][row][off] = u32x2_t{l0, l1};
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    const int kh = lane >> 5, rl = lane & 31;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
        const bool more = k0 + GK < g.K;
        if (more) fetch(k0 + GK);
        u32x4_t ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *(const u32x4_t*)&Ah[cur][kh][wm * 64 + i * 32 + rl][0];
            al[i] = *(const u32x4_t*)&Al[cur][kh][wm * 64 + i * 32 + rl][0];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[j] = *(const u32x4_t*)&Wh[cur][kh][wn * 32 * NJ + j * 32 + rl][0];
            bl[j] = *(const u32x4_t*)&Wl[cur][kh][wn * 32 * NJ + j * 32 + rl][0];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&al[i], *(bf16x8_t*)&bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&ah[i], *(bf16x8_t*)&bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&ah[i], *(bf16x8_t*)&bh[j], acc[i][j], 0, 0, 0);
            }
        if (more) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * 32 * NJ + j * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
            const float gm = g.gamma ? g.gamma[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = (acc[i][j][r] + bv) * g.scale;
                if (g.act == 1) v = gelu_fast(v);
                v *= gm;
                if (g.res) v += g.res[(long)(g.res_rows ? m % g.res_rows : m) * g.ldres + n];
                C[(long)m * g.ldc + n] = v;
            }
        }
}

__device__ __forceinline__ void split1(float x, uint16_t& hi, uint16_t& lo) {
    uint32_t h, l;
    split2(x, 0.f, h, l);
    hi = (uint16_t)h;
    lo = (uint16_t)l;
}

// Epilogue of gemm_b3t_kernel.  D[n][row]: lane holds row = lane & 31 and n = 8q + 4(lane>>5) + j (register 4q + j).
// Per element, in this order: + bias, GELU, * gamma, + residual; then fp32 float4 or the next GEMM's hi / lo planes.
// The flags are template arguments so that a combination is straight-line code: the bias / gamma / residual loads of a
// wave tile are issued ahead of the arithmetic instead of one exposed L2 round trip per 4 outputs behind a branch.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t v) {          // gelu_fast on a pair (packed fp32 multiply / fma)
    const f32x2_t x = v * 0.70710678118654752440f;
    const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2_t den = ax * 0.3275911f + 1.0f;
    const f32x2_t t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};      // 1 ulp: inside the 7e-7 of the formula
    const f32x2_t poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const f32x2_t ex = {__expf(-ax.x * ax.x), __expf(-ax.y * ax.y)};
    const f32x2_t e = 1.0f - poly * ex;
    const f32x2_t se = {copysignf(e.x, x.x), copysignf(e.y, x.y)};
    return 0.5f * v * (1.0f + se);
}

template <int NA, int NB, bool ACT, bool GAMMA, bool RES, bool PLANES, bool NT_OUT = false>
__device__ __forceinline__ void b3t_epilogue(const GemmF32Args& g, f32x16_t (&acc)[NA][NB], int nt0, int rt0, int ntiles, int rtiles, int lane) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (nt0 + a >= ntiles) break;
        const int nb = (nt0 + a) * 32 + 4 * (lane >> 5);
        float4 bias[4], gam[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = min(nb + 8 * q, g.N - 4);                     // clamped: the load is unconditional, the store is not
            bias[q] = g.bias ? *(const float4*)(g.bias + n) : zero4;
            if (GAMMA) gam[q] = *(const float4*)(g.gamma + n);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (rt0 + b >= rtiles) break;
            const int m = (rt0 + b) * 32 + (lane & 31);
            const bool mok = m < g.M;
            const int mc = min(m, g.M - 1);
            float4 res[4];
            if (RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) res[q] = *(const float4*)(g.res + (long)mc * g.ldres + min(nb + 8 * q, g.N - 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + 8 * q;
                f32x2_t v0 = {acc[a][b][4 * q] + bias[q].x, acc[a][b][4 * q + 1] + bias[q].y};
                f32x2_t v1 = {acc[a][b][4 * q + 2] + bias[q].z, acc[a][b][4 * q + 3] + bias[q].w};
                if (ACT) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }
                if (GAMMA) { v0 *= f32x2_t{gam[q].x, gam[q].y}; v1 *= f32x2_t{gam[q].z, gam[q].w}; }
                if (RES) { v0 += f32x2_t{res[q].x, res[q].y}; v1 += f32x2_t{res[q].z, res[q].w}; }
                if (mok && n < g.N) {
                    if (PLANES) {
                        uint32_t h0, l0, h1, l1;
                        split2(v0.x, v0.y, h0, l0);
                        split2(v1.x, v1.y, h1, l1);
                        const size_t at = xpack_off(m, n, (int)g.ldc);
                        if (NT_OUT) {
                            __builtin_nontemporal_store(u32x2_t{h0, h1}, (u32x2_t*)(g.Chi + at));
                            __builtin_nontemporal_store(u32x2_t{l0, l1}, (u32x2_t*)(g.Clo + at));
                        } else {
                            *(u32x2_t*)(g.Chi + at) = u32x2_t{h0, h1};
                            *(u32x2_t*)(g.Clo + at) = u32x2_t{l0, l1};
                        }
                    } else {
                        *(float4*)(g.C + (long)m * g.ldc + n) = make_float4(v0.x, v0.y, v1.x, v1.y);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// bf16x3 GEMM on pre-split operands in MFMA-FRAGMENT ORDER, no LDS (the layout of the AR engine's GEMMs, common.h):
//   activations  [row/32][k/16][lane][8 bf16], lane = row%32 + 32*((k%16)/8)     (xpack_off; hi plane, then lo plane)
//   weights      [n/32][k/16][lane][8 bf16],   lane = n%32 + 32*((k%16)/8)       (wpack_off; split + packed once)
// so every operand fragment of a wave is ONE contiguous KiB straight into registers.  What bounded the LDS-staged
// kernels was bytes in flight (two blocks x 32 KB of LDS stage per CU against ~1.5 us of loaded L2 latency: SQ_WAIT_ANY
// 43 %, matrix cores 31 % busy, no bank conflicts, 88 % L2 hits); registers hold 4 k-steps x 8 fragments = 32 KiB per WAVE
// in flight.  Block = 128 x 128 outputs, 4 waves (2 x 2), a wave = 64 columns x 64 rows = 2 x 2 accumulators D[n][row];
// per 16-deep step 8 fragment loads and 12 MFMAs (w_hi*x_lo, w_lo*x_hi, w_hi*x_hi: the products of gemm_b3_kernel).
// Output: fp32 C (bias / GELU / gamma / residual, float4 per lane) or fragment-packed hi / lo planes for the next GEMM.
// ------------------------------------------------------------------------------------
// NA x NB = 32-wide tiles per wave along n / along rows (a wave owns 32 NA columns x 32 NB rows, a block 2 x 2 waves);
// U = k-steps per register set; OCC = waves per SIMD the register budget is cut for.
//   <2,2,2,2>: 212 VGPRs, two waves per SIMD, 683 operand bytes per MFMA through the vector L1;  <2,3,1,2>: 569;
//   <3,3,1,1>: 455, one wave per SIMD;  <4,3,1,1> / <3,4,1,1>: 398;  <4,4,1,1>: 341 (accumulators = 256 registers).
// The vector L1 returns 64 B/clk per CU and the four matrix pipes of a CU retire one MFMA per 8 clk between them, so
// 683 bytes per MFMA is L1-bound before it is MFMA-bound (measured: operand stream alone = 6.1 GB / 218 us = 64.7 B/clk/CU).
template <int NA, int NB, int U, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void gemm_b3t_kernel(GemmF32Args g) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ntiles = (g.N + 31) / 32, rtiles = (g.M + 31) / 32, KT = g.K / 16;
    int tm_ = blockIdx.y, tn_ = blockIdx.x;
    if (g_xcd_map) xcd_tile((g.M + 64 * NB - 1) / (64 * NB), (g.N + 64 * NA - 1) / (64 * NA), tm_, tn_);
    const int nt0 = tn_ * 2 * NA + (wave & 1) * NA, rt0 = tm_ * 2 * NB + (wave >> 1) * NB;
    if (nt0 >= ntiles || rt0 >= rtiles) return;
    f32x16_t acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const size_t xtile = (size_t)KT * 64;                    // one 32-row / 32-column tile, in 16-byte units
    // tiles past the edge re-read the last valid one (their accumulators are never stored)
    const u32x4_t *wh[NA], *wl[NA], *xh[NB], *xl[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const size_t t = (size_t)min(nt0 + a, ntiles - 1) * xtile;
        wh[a] = (const u32x4_t*)g.Whi + t + lane; wl[a] = (const u32x4_t*)g.Wlo + t + lane;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const size_t t = (size_t)min(rt0 + b, rtiles - 1) * xtile;
        xh[b] = (const u32x4_t*)g.Ahi + t + lane; xl[b] = (const u32x4_t*)g.Alo + t + lane;
    }
    // Two register sets of U k-steps each, ping-pong: the loads of one set are issued before the MFMAs of the other and
    // the scheduler is kept from interleaving them back into short-distance load/use pairs (sched_barrier): while a set
    // feeds the matrix cores the other set's (NA + NB) U KiB x 2 planes per wave are in flight.
    struct FragSet { u32x4_t ah[NA][U], al[NA][U], bh[NB][U], bl[NB][U]; };
    auto load = [&](FragSet& f, int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NA; ++t) { f.ah[t][u] = wh[t][(size_t)(i0 + u) * 64]; f.al[t][u] = wl[t][(size_t)(i0 + u) * 64]; }
#pragma unroll
            for (int t = 0; t < NB; ++t) { f.bh[t][u] = xh[t][(size_t)(i0 + u) * 64]; f.bl[t][u] = xl[t][(size_t)(i0 + u) * 64]; }
        }
    };
    auto compute = [&](FragSet& f) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.ah[a][u], *(bf16x8_t*)&f.bl[b][u], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.al[a][u], *(bf16x8_t*)&f.bh[b][u], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.ah[a][u], *(bf16x8_t*)&f.bh[b][u], acc[a][b], 0, 0, 0);
                }
    };
    FragSet fa, fb;                                  // KT is a multiple of 4 on this path (planes_ok)
    load(fa, 0);
    for (int i = 0; i < KT; i += 2 * U) {
        load(fb, i + U);
        __builtin_amdgcn_sched_barrier(0);
        compute(fa);
        __builtin_amdgcn_sched_barrier(0);
        load(fa, min(i + 2 * U, KT - U));              // (the last round re-reads its own tiles: no branch, so the
                                                       //  waits stay counted instead of falling back to vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
        compute(fb);
        __builtin_amdgcn_sched_barrier(0);
    }
    // one straight-line epilogue per combination the decoder uses (uniform switch, taken once)
    const int combo = (g.act == 1) | (g.gamma ? 2 : 0) | (g.res ? 4 : 0) | (g.Chi ? 8 : 0);
    switch (combo) {
    case 0: b3t_epilogue<NA, NB, false, false, false, false>(g, acc, nt0, rt0, ntiles, rtiles, lane); break;       // q/k/v
    case 4: b3t_epilogue<NA, NB, false, false, true, false>(g, acc, nt0, rt0, ntiles, rtiles, lane); break;        // out_proj, fc2
    case 6: b3t_epilogue<NA, NB, false, true, true, false>(g, acc, nt0, rt0, ntiles, rtiles, lane); break;         // Vocos pw2
    case 9:                                                                                                     // fc1, Vocos pw1
        if (g.batch_inner == 2) b3t_epilogue<NA, NB, true, false, false, true, true>(g, acc, nt0, rt0, ntiles, rtiles, lane);
        else b3t_epilogue<NA, NB, true, false, false, true>(g, acc, nt0, rt0, ntiles, rtiles, lane);
        break;
    default: break;                                       // gemm_planes refuses any other combination
    }
}

// fp32 weight [N][K] -> fragment-packed bf16 hi / lo planes (padding rows of the last 32-column tile stay zero)
__global__ void split_pack_w_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int N, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    const size_t at = wpack_off(n, k, K / 16);
    split1(w[i], hi[at], lo[at]);
}

// 0 = exact f32 MFMA everywhere, 1 = bf16x3 for the [N][K]-weight GEMMs (set per call path: decode 1, encode 0)
static thread_local int g_gemm_split = 0;

static void gemm_f32(hipStream_t st, bool b_kn, const float* A, const float* W, float* C, int M, int N, int K, long lda,
                     long ldw, long ldc, const float* bias = nullptr, int act = 0, const float* gamma = nullptr,
                     const float* res = nullptr, long ldres = 0, int res_rows = 0, float scale = 1.f, int batch = 1,
                     int batch_inner = 1, long sAo = 0, long sAi = 0, long sWo = 0, long sWi = 0, long sCo = 0,
                     long sCi = 0) {
    GemmF32Args g{A, W, C, bias, gamma, res, M, N, K, lda, ldw, ldc, ldres, res_rows, scale, act, batch_inner,
                  sAo, sAi, sWo, sWi, sCo, sCi, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, batch);
    // [K][N] weights (P.V of the codec attention, the inverse-DFT basis) are bound by their operand traffic, not by
    // the matrix cores (measured: 430 vs 442 us as bf16x3): they keep the exact kernel
    if (b_kn) hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, st, g);
    // (a 128 x 256 block, NJ = 4, was measured too: fewer operand bytes per MFMA but 2 blocks per CU instead of 3 --
    // faster only on the 512 -> 4096 Vocos expansion and slower end to end)
    else if (g_gemm_split) hipLaunchKernelGGL(gemm_b3_kernel<2>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, st, g);
}

// exact-f32 MFMA GEMM for the AR engine's fp32 mode (f32path.hip): C[M,N] (row stride ldc) = A[M,K] * W[N,K]^T
void mtts_gemm_f32_exact(hipStream_t st, const float* A, const float* W, float* C, int M, int N, int K, long ldc) {
    const int keep = g_gemm_split;
    g_gemm_split = 0;
    gemm_f32(st, false, A, W, C, M, N, K, K, K, ldc);
    g_gemm_split = keep;
}

// ------------------------------------------------------------------------------------
// Row kernels
// ------------------------------------------------------------------------------------
// RVQ: emb[row][d] = sum_q codebook_q[code_q[row]][d], q ascending (quantizer.py:357-361)
__global__ void rvq_gather_kernel(const int64_t* __restrict__ codes /*[nq][rows]*/, const float* const* __restrict__ cbs,
                                  float* __restrict__ emb, int nq, int rows, int dim, int cb_size, int* __restrict__ err) {
    const int row = blockIdx.x;
    for (int d = threadIdx.x; d < dim; d += blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < nq; ++q) {
            long c = codes[(long)q * rows + row];
            if (c < 0 || c >= cb_size) { if (d == 0) atomicExch(err, 1); c = 0; }
            s += cbs[q][c * dim + d];
        }
        emb[(long)row * dim + d] = s;
    }
}

__global__ void add_pe_kernel(float* x, const float* pe, long n, int T, int d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long row = i / d;
    x[i] += pe[(row % T) * d + (i % d)];
}

// LayerNorm over C channels, one wave per row; rows whose frame index >= len[b] are zeroed
// when lens != null (torch.where(attention_mask, h, 0), modules.py:409,626).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows, int C,
                                                        float eps, const int* __restrict__ lens, int T, long ldy,
                                                        uint16_t* __restrict__ ylo = nullptr) {
    // ylo != null: the output feeds a pre-split GEMM only -- write bf16 hi / lo planes (hi plane at `y`) instead of fp32
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * C;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; v[i] = c < C ? xr[c] : 0.f; s += v[i]; }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; const float d = v[i] - mu; if (c < C) q += d * d; }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    const bool zero = lens && (row % T) >= lens[row / T];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float o = zero ? 0.f : (v[i] - mu) * inv * w[c] + b[c];
            if (ylo) { const size_t at = xpack_off(row, c, (int)ldy); split1(o, ((uint16_t*)y)[at], ylo[at]); }
            else y[(long)row * ldy + c] = o;
        }
    }
}

// LayerNorm for wide rows (C up to 16384): one 256-thread block per row.
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y, int C, float eps) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += xr[c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mu = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float d = xr[c] - mu; q += d * d; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    const float inv = 1.0f / sqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[row * C + c] = (xr[c] - mu) * inv * w[c] + b[c];
}

// Masked softmax over keys, one wave per (b, head, query) row of S [.., T, ldT].
// VarLenAttention (modules.py:84-151): additive mask finfo.min where query or key is padding
// -> a padded query row is uniform over ALL T keys.  Pad columns [T, ldT) are zeroed.
__global__ __launch_bounds__(256) void softmax_mask_kernel(float* __restrict__ S, const int* __restrict__ lens, int heads,
                                                           int T, int ldT, long rows) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int tq = (int)(row % T);
    const int b = (int)(row / ((long)T * heads));
    const int len = lens[b];
    float* s = S + row * ldT;
    const bool qvalid = tq < len;
    const float NEGV = -3.4028234663852886e38f;
    if (ldT <= 8 * 256) {
        // the row stays in registers: one 16-byte read and one write per element, one exp (ldT is a multiple of 16)
        float4 v[8];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 256 * i + 4 * lane;
            v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (j < ldT) {
                const float4 t = *(const float4*)(s + j);
                v[i].x = j + 0 < T ? ((qvalid && j + 0 < len) ? t.x : t.x + NEGV) : -INFINITY;
                v[i].y = j + 1 < T ? ((qvalid && j + 1 < len) ? t.y : t.y + NEGV) : -INFINITY;
                v[i].z = j + 2 < T ? ((qvalid && j + 2 < len) ? t.z : t.z + NEGV) : -INFINITY;
                v[i].w = j + 3 < T ? ((qvalid && j + 3 < len) ? t.w : t.w + NEGV) : -INFINITY;
            }
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // columns [T, ldT) hold -inf: exp gives exactly 0 there
            v[i].x = expf(v[i].x - mx); v[i].y = expf(v[i].y - mx); v[i].z = expf(v[i].z - mx); v[i].w = expf(v[i].w - mx);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = 256 * i + 4 * lane;
            if (j < ldT) *(float4*)(s + j) = make_float4(v[i].x / sum, v[i].y / sum, v[i].z / sum, v[i].w / sum);
        }
        return;
    }
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) {
        float v = (qvalid && j < len) ? s[j] : s[j] + NEGV;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
        float v = (qvalid && j < len) ? s[j] : s[j] + NEGV;
        sum += expf(v - mx);
    }
    sum = wave_sum(sum);
    for (int j = lane; j < ldT; j += 64) {
        float o = 0.f;
        if (j < T) {
            float v = (qvalid && j < len) ? s[j] : s[j] + NEGV;
            o = expf(v - mx) / sum;
        }
        s[j] = o;
    }
}

// deconv1 epilogue: ConvTranspose1d(k=3, s=2) as GEMM G[B*T][3*C] + overlap-add + bias + GELU.
// out [B][2T+1][C]
__global__ void deconv_s2_kernel(const float* __restrict__ G, const float* __restrict__ bias, float* __restrict__ out,
                                 int T, int C) {
    const int p = blockIdx.x, b = blockIdx.y;
    const int P = 2 * T + 1;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = bias[c];
        if (p & 1) {
            v += G[((long)b * T + (p >> 1)) * 3 * C + C + c];
        } else {
            const int t = p >> 1;
            if (t < T) v += G[((long)b * T + t) * 3 * C + c];
            if (t >= 1) v += G[((long)b * T + t - 1) * 3 * C + 2 * C + c];
        }
        out[((long)b * P + p) * C + c] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}
// deconv2 epilogue: ConvTranspose1d(k=3, s=1): out[p] = sum_j G[p-j][j]; trimmed to Pout frames.
__global__ void deconv_s1_kernel(const float* __restrict__ G, const float* __restrict__ bias, float* __restrict__ out,
                                 int Pin, int Pout, int C) {
    const int p = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = bias[c];
        for (int j = 0; j < 3; ++j) {
            const int t = p - j;
            if (t >= 0 && t < Pin) v += G[((long)b * Pin + t) * 3 * C + j * C + c];
        }
        out[((long)b * Pout + p) * C + c] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}
// im2col for Conv1d(k=7, pad=3): X7[b,t][j*C + c] = x[b, t+j-3][c]
__global__ void im2col7_kernel(const float* __restrict__ x, float* __restrict__ X7, int T, int C) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int i = threadIdx.x; i < 7 * C; i += blockDim.x) {
        const int j = i / C, c = i % C;
        const int ts = t + j - 3;
        X7[((long)b * T + t) * 7 * C + i] = (ts >= 0 && ts < T) ? x[((long)b * T + ts) * C + c] : 0.f;
    }
}
// ConvNeXt front: depthwise Conv1d(k=7, pad=3, groups=C) + bias + LayerNorm(eps) (modules.py:1139-1146).
// dw is [7][C].  One wave per (b,t) row.
// The same for C = 512 (Vocos), R consecutive rows per wave: a lane owns 8 consecutive channels (two float4 per row, one
// 16-byte store per output plane), the 7 taps of its channels stay in registers and every input row is read once per
// wave (R + 6 rows for R outputs) instead of once per tap.  The convolution accumulates in the reference's tap order.
template <int R>
__global__ __launch_bounds__(256) void dwconv_ln512_kernel(const float* __restrict__ h, const float* __restrict__ dw,
                                                           const float* __restrict__ dwb, const float* __restrict__ lw,
                                                           const float* __restrict__ lb, float* __restrict__ y, int B, int T,
                                                           float eps, uint16_t* __restrict__ ylo) {
    constexpr int C = 512;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long rows = (long)B * T, r0 = ((long)blockIdx.x * 4 + wave) * R;
    if (r0 >= rows) return;
    const int c0 = lane * 8;
    float win[R + 6][8];
#pragma unroll
    for (int i = 0; i < R + 6; ++i) {
        const long r = r0 + i - 3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (r >= 0 && r < rows) { a = *(const float4*)(h + r * C + c0); b = *(const float4*)(h + r * C + c0 + 4); }
        win[i][0] = a.x; win[i][1] = a.y; win[i][2] = a.z; win[i][3] = a.w;
        win[i][4] = b.x; win[i][5] = b.y; win[i][6] = b.z; win[i][7] = b.w;
    }
    float w[7][8], pb[8], g[8], be[8];
    auto ld8 = [&](const float* p, float* o) {
        const float4 a = *(const float4*)(p + c0), b = *(const float4*)(p + c0 + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    };
#pragma unroll
    for (int j = 0; j < 7; ++j) ld8(dw + j * C, w[j]);
    ld8(dwb, pb); ld8(lw, g); ld8(lb, be);
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const long row = r0 + o;
        if (row >= rows) break;
        const int t = (int)(row % T);
        float v[8], s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = pb[c];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int ts = t + j - 3;
            if (ts >= 0 && ts < T) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += win[o + j][c] * w[j][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) s += v[c];
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float d = v[c] - mu; q += d * d; }
        const float inv = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float ov[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ov[c] = (v[c] - mu) * inv * g[c] + be[c];
        if (ylo) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) split2(ov[2 * c], ov[2 * c + 1], hi[c], lo[c]);
            const size_t at = xpack_off((int)row, c0, C);
            *(u32x4_t*)((uint16_t*)y + at) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
            *(u32x4_t*)(ylo + at) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        } else {
            *(float4*)(y + row * C + c0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            *(float4*)(y + row * C + c0 + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
        }
    }
}

__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* __restrict__ h, const float* __restrict__ dw,
                                                        const float* __restrict__ dwb, const float* __restrict__ lw,
                                                        const float* __restrict__ lb, float* __restrict__ y, int B, int T,
                                                        int C, float eps, uint16_t* __restrict__ ylo = nullptr) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)B * T) return;
    const int t = (int)(row % T);
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        float a = 0.f;
        if (c < C) {
            a = dwb[c];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int ts = t + j - 3;
                if (ts >= 0 && ts < T) a += h[(row + j - 3) * C + c] * dw[j * C + c];
            }
        }
        v[i] = a;
        s += a;
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; const float d = v[i] - mu; if (c < C) q += d * d; }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float o = (v[i] - mu) * inv * lw[c] + lb[c];
            if (ylo) { const size_t at = xpack_off((int)row, c, C); split1(o, ((uint16_t*)y)[at], ylo[at]); }
            else y[row * C + c] = o;
        }
    }
}
// ISTFT head front (modules.py:970-985): o[.., :nb] = log-magnitude, o[.., nb:] = phase
//   SP[row][k] = min(exp(mag),100)*cos(p), SP[row][nb+k] = ..*sin(p); pad columns zero.
__global__ void istft_prep_kernel(const float* __restrict__ o, float* __restrict__ SP, long rows, int nb, int ldo, int ldsp) {
    const long row = blockIdx.x;
    for (int k = threadIdx.x; k < ldsp; k += blockDim.x) {
        float v = 0.f;
        if (k < 2 * nb) {
            const int kk = k < nb ? k : k - nb;
            const float mag = fminf(expf(o[row * ldo + kk]), 100.0f);
            const float ph = o[row * ldo + nb + kk];
            v = k < nb ? mag * cosf(ph) : mag * sinf(ph);
        }
        SP[row * ldsp + k] = v;
    }
}
// window + overlap-add + trim + envelope normalisation (modules.py:769-790)
__global__ void istft_ola_kernel(const float* __restrict__ frames /*[B][T][n]*/, const float* __restrict__ window,
                                 float* __restrict__ wav /*[B][T*hop]*/, int T, int n, int hop) {
    const int b = blockIdx.y;
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (long)T * hop) return;
    const long pos = s + (n - hop) / 2;
    float acc = 0.f, env = 0.f;
    const int t_hi = (int)(pos / hop);
    for (int j = 0; j < (n + hop - 1) / hop; ++j) {
        const int t = t_hi - j;
        if (t < 0 || t >= T) continue;
        const int off = (int)(pos - (long)t * hop);
        if (off >= n) continue;
        const float w = window[off];
        acc += frames[((long)b * T + t) * n + off] * w;
        env += w * w;
    }
    wav[(long)b * T * hop + s] = acc / env;
}


// ------------------------------------------------------------------------------------
// Encode-side kernels (XY_Tokenizer.inference_tokenize, reference model.py:55-101)
// ------------------------------------------------------------------------------------
// STFT framing of torch.stft(center=True, pad_mode="reflect") over the 30 s zero-padded chunk
// (feature_extractor.py:78-104): frames[b,t,n] = x_reflect[t*hop + n - n_fft/2] * hann[n]
__global__ void mel_frames_kernel(const float* __restrict__ wav, const float* __restrict__ window, float* __restrict__ fr,
                                  int nsamp_in, int N, int nfr, int n_fft, int hop) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
        long i = (long)t * hop + n - n_fft / 2;
        if (i < 0) i = -i;
        if (i >= N) i = 2L * N - 2 - i;
        const float v = (i < nsamp_in) ? wav[(long)b * nsamp_in + i] : 0.f;
        fr[((long)b * nfr + t) * n_fft + n] = v * window[n];
    }
}
// |X|^2 from the [cos | sin] DFT GEMM output; columns padded to ldp with zeros
__global__ void power_kernel(const float* __restrict__ ri, float* __restrict__ pw, int nb, int ldri, int ldp) {
    const long row = blockIdx.x;
    for (int k = threadIdx.x; k < ldp; k += blockDim.x) {
        float v = 0.f;
        if (k < nb) { const float re = ri[row * ldri + k], im = ri[row * ldri + nb + k]; v = re * re + im * im; }
        pw[row * ldp + k] = v;
    }
}
// log10(clamp(mel,1e-10)); per-sample maximum (one block per sample)
__global__ __launch_bounds__(256) void logmel_max_kernel(float* __restrict__ mel, float* __restrict__ mx, long per_sample) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    float* p = mel + (long)b * per_sample;
    float m = -INFINITY;
    for (long i = threadIdx.x; i < per_sample; i += 256) {
        const float v = log10f(fmaxf(p[i], 1e-10f));
        p[i] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) mx[b] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ void logmel_norm_kernel(float* __restrict__ mel, const float* __restrict__ mx, long per_sample, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float m = mx[i / per_sample];
    mel[i] = (fmaxf(mel[i], m - 8.0f) + 4.0f) / 4.0f;
}
// generic im2col for Conv1d(k, stride, pad) on token-major data: X[b,t][j*C+c] = x[b, t*stride + j - pad][c]
__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ X, int Tin, int Tout, int C, int ksz,
                              int stride, int pad) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int i = threadIdx.x; i < ksz * C; i += blockDim.x) {
        const int j = i / C, c = i % C;
        const int ts = t * stride + j - pad;
        X[((long)b * Tout + t) * ksz * C + i] = (ts >= 0 && ts < Tin) ? x[((long)b * Tin + ts) * C + c] : 0.f;
    }
}
__global__ void silu_mul_kernel(float* __restrict__ g, const float* __restrict__ u, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = g[i];
    g[i] = (x / (1.0f + expf(-x))) * u[i];
}
// One RVQ stage (quantizer.py:154-191,:277-327, eval branch): nearest codebook entry of the masked
// residual by dist = (|r|^2 - 2 r.c) + |c|^2, first index on ties; residual -= codebook[idx] * mask.
// dot: [rows][K] from the GEMM.  One 256-thread block per row.
__global__ __launch_bounds__(256) void vq_argmin_update_kernel(const float* __restrict__ dot, const float* __restrict__ cb,
                                                               const float* __restrict__ cc, float* __restrict__ residual,
                                                               int64_t* __restrict__ codes, const int* __restrict__ lens,
                                                               int T, int K, int D) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    __shared__ float sh_rr;
    const long row = blockIdx.x;
    const bool valid = (int)(row % T) < lens[row / T];
    float* r = residual + row * D;
    float rr = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) rr += r[d] * r[d];
    rr = wave_sum(rr);
    if ((threadIdx.x & 63) == 0) shv[threadIdx.x >> 6] = rr;
    __syncthreads();
    if (threadIdx.x == 0) sh_rr = shv[0] + shv[1] + shv[2] + shv[3];
    __syncthreads();
    rr = valid ? sh_rr : 0.f;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < K; j += 256) {
        const float dt = valid ? dot[row * K + j] : 0.f;
        const float nd = -((rr - 2.0f * dt) + cc[j]);
        if (nd > best || (nd == best && j < bi)) { best = nd; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = best; shi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    best = shv[0]; bi = shi[0];
    for (int w = 1; w < 4; ++w)
        if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
    if (threadIdx.x == 0) codes[row] = bi;
    if (valid)
        for (int d = threadIdx.x; d < D; d += 256) r[d] -= cb[(long)bi * D + d];
}

// ------------------------------------------------------------------------------------
// Host object
// ------------------------------------------------------------------------------------
static thread_local char g_cerr[512] = "";
static int cfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cerr, sizeof(g_cerr), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* mtts_codec_last_error(void) { return g_cerr; }
#define CHK(x)                                                                                 \
    do {                                                                                       \
        hipError_t _e = (x);                                                                   \
        if (_e != hipSuccess) return cfail(MTTS_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct MttsCodec {
    MttsCodecConfig c;
    int device;
    std::map<std::string, float*> w;
    std::map<std::string, size_t> wn;
    float** d_cbs = nullptr;
    // workspace
    size_t cap_rows = 0;        // rows at the 100 Hz stage the workspace is sized for
    int cap_B = 0, cap_T = 0;
    float *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *bufD = nullptr, *bufE = nullptr, *big = nullptr, *scores = nullptr;
    float *melbuf = nullptr, *melmax = nullptr;
    int* d_lens2 = nullptr;
    std::vector<int> h_lens, h_lens4;
    int64_t* d_codes = nullptr;
    int *d_lens = nullptr, *d_lens4 = nullptr, *d_err = nullptr;
    int split_decode = 1;       // decode-direction GEMMs as bf16x3 (MTTS_CODEC_GEMM=f32: exact f32 MFMA)
    // bf16 hi/lo planes of the constant weights (made on first use after binding): key = the engine's fp32 copy
    std::map<const float*, uint16_t*> wplanes;
    int planes = 1;             // pre-split fragment-packed operands for the big decode-direction GEMMs (MTTS_CODEC_PLANES=0: off)
    int attn_packed = 1;        // fused attention on K / V packed once per layer (MTTS_CODEC_ATTN_PACKED=0: split per wave and tile)
    int nt_out = 1;             // nontemporal stores of the GELU GEMMs' output planes (393 MB per Vocos pw1 launch: -1.3 % per window; MTTS_CODEC_NT_OUT=0: off)
    int dw_rows = 4;            // dwconv_ln512_kernel: rows per wave (4: 31 us per launch at 8 windows; 8: 46 us, 2: 33 us)
    int tile = 0;               // gemm_b3t_kernel: MTTS_CODEC_TILE = NA NB U OCC as digits forces one variant (0: per shape)
    // Vocos pw1 -> GELU -> pw2 in one launch (codec_fused.hip): -1 = where it pays (fused_pw_pays), 0 = never, N = from N rows
    // per call up (MTTS_CODEC_FUSED_PW)
    long fused_pw_rows = -1;
    std::map<const float*, uint16_t*> wplanes_perm;   // W2 planes with the K order the fused kernel's second GEMM expects
};
void launch_split_pack_w2perm(hipStream_t st, const float* w2, uint16_t* hi, uint16_t* lo);
void launch_vocos_pw_fused(hipStream_t st, const uint16_t* xn_planes, long x_plane_elems, const uint16_t* w1_planes, const float* b1,
                           const uint16_t* w2perm_planes, long w_plane_elems, const float* b2, const float* gamma, float* h, int M);

extern "C" int32_t mtts_codec_create(const MttsCodecConfig* c, int32_t device, MttsCodec** out) {
    if (!c || !out) return cfail(MTTS_EINVAL, "null argument");
    if (c->adapter_dim % c->adapter_heads || c->dec_dim % c->dec_heads) return cfail(MTTS_EINVAL, "bad head split");
    if (c->adapter_dim > 1024 || c->dec_dim > 1024 || c->voc_dim > 1024) return cfail(MTTS_EINVAL, "row kernels hold <= 1024 channels");
    if (c->n_fft % 2 || c->hop < 1 || (c->n_fft - c->hop) % 2) return cfail(MTTS_EINVAL, "bad STFT geometry");
    if (c->enc_dim > 1024 || (c->enc_heads && c->enc_dim % c->enc_heads)) return cfail(MTTS_EINVAL, "bad encoder width");
    CHK(hipSetDevice(device));
    MttsCodec* k = new MttsCodec();
    k->c = *c;
    k->device = device;
    if (const char* m = getenv("MTTS_CODEC_GEMM")) k->split_decode = strcmp(m, "f32") != 0;
    if (const char* m = getenv("MTTS_CODEC_PLANES")) k->planes = atoi(m) != 0;
    if (const char* m = getenv("MTTS_CODEC_TILE")) k->tile = atoi(m);
    if (const char* m = getenv("MTTS_CODEC_DW_ROWS")) k->dw_rows = atoi(m);
    if (const char* m = getenv("MTTS_CODEC_NT_OUT")) k->nt_out = atoi(m);
    if (const char* m = getenv("MTTS_CODEC_ATTN_PACKED")) k->attn_packed = atoi(m);
    if (const char* m = getenv("MTTS_CODEC_FUSED_PW")) k->fused_pw_rows = atol(m);
    if (const char* m = getenv("MTTS_CODEC_XCD")) { const int v = atoi(m) != 0; CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_map), &v, sizeof(int))); }
    CHK(hipMalloc((void**)&k->d_err, 4));
    CHK(hipMemset(k->d_err, 0, 4));
    *out = k;
    return MTTS_OK;
}

extern "C" int32_t mtts_codec_destroy(MttsCodec* k) {
    if (!k) return MTTS_OK;
    hipSetDevice(k->device);
    hipDeviceSynchronize();
    for (auto& kv : k->w) hipFree(kv.second);
    for (auto& kv : k->wplanes) hipFree(kv.second);
    for (auto& kv : k->wplanes_perm) hipFree(kv.second);
    float* bufs[] = {k->bufA, k->bufB, k->bufC, k->bufD, k->bufE, k->big, k->scores, k->melbuf, k->melmax};
    for (float* p : bufs) if (p) hipFree(p);
    if (k->d_lens2) hipFree(k->d_lens2);
    if (k->d_codes) hipFree(k->d_codes);
    if (k->d_lens) hipFree(k->d_lens);
    if (k->d_lens4) hipFree(k->d_lens4);
    if (k->d_cbs) hipFree(k->d_cbs);
    hipFree(k->d_err);
    delete k;
    return MTTS_OK;
}

// Bind one fp32 tensor under the engine's role name (INTEGRATION.md lists the roles and how
// each is derived from the reference state dict); the engine keeps its own copy.
extern "C" int32_t mtts_codec_bind(MttsCodec* k, const char* role, const float* dev, int64_t n, void* stream) {
    if (!k || !role || !dev || n < 1) return cfail(MTTS_EINVAL, "bad argument");
    CHK(hipSetDevice(k->device));
    std::string r(role);
    if (k->w.count(r)) {
        auto pl = k->wplanes.find(k->w[r]);
        if (pl != k->wplanes.end()) { CHK(hipDeviceSynchronize()); hipFree(pl->second); k->wplanes.erase(pl); }
        auto pp = k->wplanes_perm.find(k->w[r]);
        if (pp != k->wplanes_perm.end()) { CHK(hipDeviceSynchronize()); hipFree(pp->second); k->wplanes_perm.erase(pp); }
        hipFree(k->w[r]); k->w.erase(r);
    }
    float* p = nullptr;
    CHK(hipMalloc((void**)&p, (size_t)n * 4));
    CHK(hipMemcpyAsync(p, dev, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    k->w[r] = p;
    k->wn[r] = (size_t)n;
    return MTTS_OK;
}

#define TRYC(x)            \
    do {                   \
        int _r = (x);      \
        if (_r) return _r; \
    } while (0)
static int need(MttsCodec* k, const std::string& name, size_t n, float** out) {
    auto it = k->w.find(name);
    if (it == k->w.end()) return cfail(MTTS_ESTATE, "codec tensor '%s' is not bound", name.c_str());
    if (k->wn[name] != n) return cfail(MTTS_EINVAL, "codec tensor '%s' has %zu elements, expected %zu", name.c_str(), k->wn[name], n);
    *out = it->second;
    return 0;
}
#define NEED(var, name, n)                          \
    float* var = nullptr;                           \
    do {                                            \
        int _r = need(k, (name), (size_t)(n), &var); \
        if (_r) return _r;                          \
    } while (0)

// Which tile shape: every variant computes the same bits (a tile only decides which wave owns an output element), so
// the choice is free to follow the shape.  A CU holds one 4-wave block per wave-per-SIMD of the variant and runs its
// share of the grid in sequence: time = ceil(blocks / 256) x (a KT + b + e [GELU epilogue]) with per-variant constants
// fitted to a rocprofv3 sweep of all variants over the decoder's ten GEMM shapes (profiles/r02_codec_tile_sweep.json;
// the model picks the measured-best variant on each of them).  Big tiles at one wave per SIMD need 341-455 operand
// bytes per MFMA from the vector L1 instead of 683 and win wherever K is long (pw2 470 -> 269 us, fc2 314 -> 150 us);
// 64 x 96 at two waves per SIMD wins the 512 -> 4096 expansion, whose time is its GELU / split / store epilogue.
static int b3t_choose(int M, int N, int K, int act) {
    struct Variant { int code, na, nb; double a, b, e; };
    // fitted at 8 windows per call (profiles/r02_codec_tile_sweep.json): the regime of long runs of tiles per CU
    static const Variant big[] = {{2222, 2, 2, 0.594, -3.5, 0.0}, {2312, 2, 3, 0.768, -2.1, -0.4}, {4221, 4, 2, 0.634, 9.6, 4.8},
                                  {3311, 3, 3, 0.693, 12.5, 2.4}, {3411, 3, 4, 0.862, 13.0, 8.1}, {4311, 4, 3, 0.875, 12.8, 9.9},
                                  {4411, 4, 4, 1.005, 25.9, 10.5}};
    // fitted over 1 / 2 / 4 / 8 windows per call (r02_codec_tile_sweep_small.json), with the 32- and 64-wide tiles that
    // fill the chip when a call has few rows (fc2 of one window: 18 blocks of 128 x 128, 72 of 64 x 64)
    static const Variant small[] = {{2222, 2, 2, 0.360, 8.1, -0.3}, {2312, 2, 3, 0.450, 10.8, 2.1}, {4221, 4, 2, 0.518, 14.8, 0.9},
                                    {3311, 3, 3, 0.566, 13.7, 3.0}, {3411, 3, 4, 0.672, 16.3, 4.5}, {4311, 4, 3, 0.673, 17.4, 6.1},
                                    {4411, 4, 4, 0.837, 22.8, 7.0}, {2122, 2, 1, 0.253, 3.7, -2.1}, {1222, 1, 2, 0.262, 3.4, -2.2},
                                    {1122, 1, 1, 0.155, 1.5, -1.1}};
    const bool large = (long)M * N >= (1L << 26);        // (the 512 -> 4096 expansion from 8 windows up: the second fit misjudges it)
    const Variant* v0 = large ? big : small;
    const int nv = large ? (int)(sizeof(big) / sizeof(big[0])) : (int)(sizeof(small) / sizeof(small[0]));
    int code = 0;
    double best = 1e30;
    for (int i = 0; i < nv; ++i) {
        const Variant& v = v0[i];
        const long blocks = (long)((N + 64 * v.na - 1) / (64 * v.na)) * ((M + 64 * v.nb - 1) / (64 * v.nb));
        const double t = (double)((blocks + 255) / 256) * (v.a * (K / 16) + v.b + (act == 1 ? v.e : 0.0));
        if (t < best) { best = t; code = v.code; }
    }
    return code;
}

// code = NA NB U OCC as decimal digits
static int b3t_launch(hipStream_t st, const GemmF32Args& g, int code) {
    auto grid = [&](int na, int nb) { return dim3((g.N + 64 * na - 1) / (64 * na), (g.M + 64 * nb - 1) / (64 * nb), 1); };
    switch (code) {
#define MTTS_B3T(NA, NB, U, OCC) \
    case NA * 1000 + NB * 100 + U * 10 + OCC: hipLaunchKernelGGL((gemm_b3t_kernel<NA, NB, U, OCC>), grid(NA, NB), dim3(256), 0, st, g); break;
        MTTS_B3T(2, 2, 2, 2) MTTS_B3T(2, 3, 1, 2) MTTS_B3T(4, 2, 2, 1) MTTS_B3T(3, 3, 1, 1) MTTS_B3T(3, 4, 1, 1) MTTS_B3T(4, 3, 1, 1)
        MTTS_B3T(2, 1, 2, 2) MTTS_B3T(1, 2, 2, 2) MTTS_B3T(1, 1, 2, 2)
        MTTS_B3T(4, 4, 1, 1)      // (2213, 2412, 2421, 3212, 3321, 3421, 4321, 4212 were measured too: never the best, removed)
#undef MTTS_B3T
    default: return cfail(MTTS_EINVAL, "gemm_planes: no kernel for tile code %d", code);
    }
    return 0;
}

// C[M,N] = epi(A * W^T) on pre-split operands (gemm_b3t_kernel).  A: fragment-packed bf16 hi plane at `a_planes`, lo plane
// pad32(M) rows further; W: the engine's fp32 weight, split + packed on first use.  Output: fp32 `C` (bias / GELU /
// gamma / residual as gemm_f32) or, with c_planes, fragment-packed planes in the same buffer (the next GEMM's A).
static bool planes_ok(const MttsCodec* k, int K, long lda) { return k->planes && g_gemm_split && K % 64 == 0 && lda == K; }
static inline long pad32(long r) { return (r + 31) / 32 * 32; }
// bf16 hi / lo planes of a constant weight [N][K], fragment-packed (made on first use; hi plane, lo plane pad32(N) * K further)
static int weight_planes(MttsCodec* k, hipStream_t st, const float* W, int N, int K, uint16_t** out) {
    auto it = k->wplanes.find(W);
    if (it != k->wplanes.end()) { *out = it->second; return 0; }
    uint16_t* wp = nullptr;
    const long wn = pad32(N) * (long)K;
    CHK(hipMalloc((void**)&wp, (size_t)wn * 4));
    CHK(hipMemsetAsync(wp, 0, (size_t)wn * 4, st));
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, wp, wp + wn, N, K);
    CHK(hipStreamSynchronize(st));           // once per weight: later calls may come in on another stream
    k->wplanes[W] = wp;
    *out = wp;
    return 0;
}
// the fused Vocos kernel's second weight: planes with its permuted K order (codec_fused.hip)
static int weight_planes_perm(MttsCodec* k, hipStream_t st, const float* W2, long elems, uint16_t** out) {
    auto it = k->wplanes_perm.find(W2);
    if (it != k->wplanes_perm.end()) { *out = it->second; return 0; }
    uint16_t* wp = nullptr;
    CHK(hipMalloc((void**)&wp, (size_t)elems * 4));
    launch_split_pack_w2perm(st, W2, wp, wp + elems);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(st));
    k->wplanes_perm[W2] = wp;
    *out = wp;
    return 0;
}
// The fused kernel runs ONE 64-row block per CU at a time (it owns all 160 KiB of LDS), so a call takes
// ceil(blocks / 256) rounds of ~330 us whatever the last round holds, while the two launches cost 0.0253 us per row
// (414 us per 16 384 rows).  Auto: the full rounds always go to the fused kernel; the last, partial round too when it is
// at least 78 % full (330 < 0.78 x 414 + two launch overheads), otherwise to the two launches (gemm_planes on a row range).
// Measured 1 .. 32 windows per call, profiles/r03_codec_fused_pw.json: never slower than the two launches, -8 % per
// window at 16 and 32.
static bool fused_pw_pays(long rows, long setting) {
    if (setting >= 0) return setting > 0 && rows >= setting;
    const long blocks = (rows + 63) / 64, last = blocks % 256;
    return last == 0 || last >= 200;
}
// `row0` / `plane_rows`: rows row0 .. row0 + M - 1 of operands whose planes were laid out for `plane_rows` rows (row0 a
// multiple of 32: a whole number of fragment tiles; 0 / 0 = all of them): the part of a Vocos call the fused kernel leaves.
static int gemm_planes(MttsCodec* k, hipStream_t st, const float* a_planes, long a_rows, const float* W, float* C, int M, int N,
                       int K, long lda, long ldc, const float* bias, int act, const float* gamma, const float* res, long ldres,
                       bool c_planes, long c_rows, long row0 = 0, long plane_rows = 0) {
    (void)a_rows; (void)c_rows; (void)lda;
    if (!plane_rows) plane_rows = M;
    if (row0 % 32) return cfail(MTTS_EINVAL, "gemm_planes: row0 must be a multiple of 32");
    if (N % 4) return cfail(MTTS_EINVAL, "gemm_planes: N must be a multiple of 4");
    const int combo = (act == 1) | (gamma ? 2 : 0) | (res ? 4 : 0) | (c_planes ? 8 : 0);          // the kernel's epilogues
    if (combo != 0 && combo != 4 && combo != 6 && combo != 9) return cfail(MTTS_EINVAL, "gemm_planes: no epilogue for flag combination %d", combo);
    uint16_t* wp = nullptr;
    const long wn = pad32(N) * (long)K;                      // elements per weight plane
    TRYC(weight_planes(k, st, W, N, K, &wp));
    // activation planes: hi plane first, lo plane pad32(M) rows further (both in fragment order, K = row length)
    const uint16_t* ah = (const uint16_t*)a_planes + row0 * K;
    uint16_t* ch = c_planes ? (uint16_t*)C + row0 * ldc : nullptr;
    GemmF32Args g{nullptr, nullptr, c_planes ? C : C + row0 * ldc, bias, gamma, res ? res + row0 * ldres : nullptr, M, N, K, (long)K, (long)K, ldc, ldres,
                  0, 1.f, act, 1, 0, 0, 0, 0, 0, 0,
                  ah, ah + pad32(plane_rows) * K, wp, wp + wn, ch, c_planes ? ch + pad32(plane_rows) * ldc : nullptr};
    int code = k->tile ? k->tile : b3t_choose(M, N, K, act);
    if (k->nt_out && combo == 9) g.batch_inner = 2;
    TRYC(b3t_launch(st, g, code));
    return 0;
}

static int ensure_workspace(MttsCodec* k, int B, int T) {
    if (B <= k->cap_B && T <= k->cap_T) return 0;
    B = std::max(B, k->cap_B);            // never shrink one dimension while growing the other
    T = std::max(T, k->cap_T);
    const MttsCodecConfig& c = k->c;
    float* bufs[] = {k->bufA, k->bufB, k->bufC, k->bufD, k->bufE, k->big, k->scores, k->melbuf, k->melmax};
    for (float* p : bufs) if (p) hipFree(p);
    if (k->d_codes) { hipFree(k->d_codes); hipFree(k->d_lens); hipFree(k->d_lens4); hipFree(k->d_lens2); }
    const int up = c.up_stride;
    const size_t r100 = (size_t)B * (2 * (size_t)T * up + 3) + 32;   // frames at the 100 Hz stage (+ deconv slack, + the fragment-packed planes' row padding)
    size_t wide = std::max<size_t>({(size_t)c.quant_out_dim, (size_t)3 * c.adapter_dim, (size_t)3 * c.dec_dim,
                                    (size_t)c.voc_dim, (size_t)7 * c.mel_bins, (size_t)c.n_fft + 16});
    size_t n_small = r100 * wide;
    size_t n_big = r100 * std::max<size_t>({(size_t)c.voc_inter, (size_t)c.dec_ffn, (size_t)c.adapter_ffn});
    const int Tdec = T * up;
    const size_t ldT = ((size_t)Tdec + 15) / 16 * 16;
    size_t n_sc = (size_t)B * std::max(c.dec_heads, c.adapter_heads) * (size_t)Tdec * ldT;
    // the fused attention parks K / V as four bf16 planes of 32-key tiles here (launch_codec_attn): 16 KiB per (b, head, tile)
    n_sc = std::max(n_sc, (size_t)B * std::max(c.dec_heads, c.adapter_heads) * (((size_t)Tdec + 31) / 32) * 4096);
    CHK(hipMalloc((void**)&k->bufA, n_small * 4));
    CHK(hipMalloc((void**)&k->bufB, n_small * 4));
    CHK(hipMalloc((void**)&k->bufC, n_small * 4));
    CHK(hipMalloc((void**)&k->bufD, n_small * 4));
    CHK(hipMalloc((void**)&k->bufE, n_small * 4));
    CHK(hipMalloc((void**)&k->melbuf, r100 * (size_t)c.mel_bins * 4));
    CHK(hipMalloc((void**)&k->melmax, (size_t)B * 4));
    CHK(hipMalloc((void**)&k->d_lens2, (size_t)B * 4));
    CHK(hipMalloc((void**)&k->big, n_big * 4));
    CHK(hipMalloc((void**)&k->scores, n_sc * 4));
    CHK(hipMalloc((void**)&k->d_codes, (size_t)c.nq * B * T * 8));
    CHK(hipMalloc((void**)&k->d_lens, (size_t)B * 4));
    CHK(hipMalloc((void**)&k->d_lens4, (size_t)B * 4));
    k->cap_B = B;
    k->cap_T = T;
    return 0;
}

// ------------------------------------------------------------------------------------
// Fused attention of the codec's transformer layers, decode direction (head_dim 64): softmax(QK^T * hd^-0.5 + mask) V
// without the [T][T] score matrix ever reaching HBM (the three-launch form moves ~3.4 GB per layer for 8 windows).
// One wave owns 32 queries and walks the keys 32 at a time, products as bf16x3 on v_mfma_f32_32x32x16_bf16:
//   S^T[key][q] = K . Q^T       (A = K rows, B = Q^T): lane = query, its 16 registers = 16 of the 32 keys
//   O^T[d][q]  += V^T . P^T     (A = V^T, B = P^T):    lane = query again
// so the softmax statistics of a query are in-lane (+ one exchange with lane^32), the running rescale of O is a
// per-lane scalar, and P feeds the second product straight from the registers it was computed in: the reduction
// over keys may visit them in any order as long as V^T uses the same one (key of operand slot (g, j) of step s is
// 16 s + 4 g + (j & 3) + 8 (j >> 2)).  fp32 running max / sum (online softmax, hardware exp), fp32 accumulators.
// Mask (VarLenAttention, modules.py:84-151): a valid query sees keys < len; a padded query row is uniform over all
// T keys.  grid = (ceil(T/128), heads, B), block 256 = 4 waves x 32 queries.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float4& a, const float4& b, u32x4_t& hi, u32x4_t& lo) {
    uint32_t h, l;
    split2(a.x, a.y, h, l); hi.x = h; lo.x = l;
    split2(a.z, a.w, h, l); hi.y = h; lo.y = l;
    split2(b.x, b.y, h, l); hi.z = h; lo.z = l;
    split2(b.z, b.w, h, l); hi.w = h; lo.w = l;
}

__global__ __launch_bounds__(256) void codec_attn_kernel(const float* __restrict__ qkv, float* __restrict__ att,
                                                         const int* __restrict__ lens, int T, int d, float scale,
                                                         uint16_t* __restrict__ att_lo = nullptr) {
    const int b = blockIdx.z, head = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 5, ql = lane & 31;
    const int q = blockIdx.x * 128 + wave * 32 + ql;
    if (blockIdx.x * 128 + wave * 32 >= T) return;
    const int len = lens[b];
    const bool qpad = q >= len;                         // padded (or out-of-range) query: uniform over all T keys
    const long ld = 3L * d;
    const float* base = qkv + (long)b * T * ld + head * 64;
    // Q^T operand: 8 consecutive d of this lane's query per 16-deep step, split once
    u32x4_t qh[4], qlo[4];
    {
        const float* qp = base + (long)min(q, T - 1) * ld + 8 * g;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 a = *(const float4*)(qp + 16 * s4), c = *(const float4*)(qp + 16 * s4 + 4);
            split8(a, c, qh[s4], qlo[s4]);
        }
    }
    f32x16_t o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int kend = qpad ? T : len;                    // every lane of a wave needs keys up to the largest bound
    int kmax = kend;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
    // software pipeline: the K rows of the next tile and the V values of the current one are requested before the
    // current tile's products and softmax, so their L2 latency overlaps the arithmetic (2 waves per SIMD only)
    float4 kreg[8];
    auto load_k = [&](int k0) {
        const float* kp = base + d + (long)min(k0 + ql, T - 1) * ld + 8 * g;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { kreg[2 * s4] = *(const float4*)(kp + 16 * s4); kreg[2 * s4 + 1] = *(const float4*)(kp + 16 * s4 + 4); }
    };
    load_k(0);
    for (int k0 = 0; k0 < kmax; k0 += 32) {
        // ---- V^T operand values of this tile: row d = 32 t + (lane&31), slot j <-> key k0 + 16 s2 + 4 g + (j&3) + 8 (j>>2)
        float vv[2][2][8];
        const bool tail = k0 + 32 > T;                                       // only the last tile clamps its keys
        const float* vrow = base + 2 * d + (long)(k0 + 4 * g) * ld + ql;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int koff = 16 * s2 + (j & 3) + 8 * (j >> 2);          // + 4 g + k0: this lane's key
                    const int key = tail ? min(k0 + 4 * g + koff, T - 1) - (k0 + 4 * g) : koff;
                    vv[s2][t][j] = vrow[(long)key * ld + 32 * t];
                }
        // ---- S^T tile: keys k0 + (lane&31) as A rows
        f32x16_t sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
        {
            u32x4_t kh[4], kl[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) split8(kreg[2 * s4], kreg[2 * s4 + 1], kh[s4], kl[s4]);
            if (k0 + 32 < kmax) load_k(k0 + 32);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&kl[s4], *(bf16x8_t*)&qh[s4], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&kh[s4], *(bf16x8_t*)&qlo[s4], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&kh[s4], *(bf16x8_t*)&qh[s4], sacc, 0, 0, 0);
            }
        }
        // ---- masked scores of this lane's query: register i <-> key k0 + (i&3) + 8*(i>>2) + 4*g
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * g;
            float v = qpad ? 0.f : sacc[i] * scale;
            if (key >= kend) v = -INFINITY;
            sacc[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = max_xor32(mx);
        const float m_new = fmaxf(m_run, mx);
        const float corr = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);    // no key yet: nothing to rescale
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pv = (sacc[i] == -INFINITY) ? 0.f : __expf(sacc[i] - m_new);
            sacc[i] = pv;
            ps += pv;
        }
        ps = add_xor32(ps);
        l_run = l_run * corr + ps;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[t][i] *= corr;
        // ---- O^T += V^T . P^T : two 16-key steps; P operand = registers 8s..8s+7 as they stand
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4_t ph, pl;
            split8(make_float4(sacc[8 * s2], sacc[8 * s2 + 1], sacc[8 * s2 + 2], sacc[8 * s2 + 3]),
                   make_float4(sacc[8 * s2 + 4], sacc[8 * s2 + 5], sacc[8 * s2 + 6], sacc[8 * s2 + 7]), ph, pl);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4_t vh, vl;
                split8(make_float4(vv[s2][t][0], vv[s2][t][1], vv[s2][t][2], vv[s2][t][3]),
                       make_float4(vv[s2][t][4], vv[s2][t][5], vv[s2][t][6], vv[s2][t][7]), vh, vl);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&vl, *(bf16x8_t*)&ph, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&vh, *(bf16x8_t*)&pl, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&vh, *(bf16x8_t*)&ph, o[t], 0, 0, 0);
            }
        }
    }
    if (q < T) {
        const float inv = 1.0f / l_run;
        const long ob = ((long)b * T + q) * d + head * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int col = head * 64 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * g;
                if (att_lo) {                                  // fragment-packed planes for the pre-split o_proj GEMM
                    const size_t at = xpack_off(b * T + q, col, d);
                    split1(o[t][i] * inv, ((uint16_t*)att)[at], att_lo[at]);
                } else att[ob - head * 64 + col] = o[t][i] * inv;
            }
    }
}

// ------------------------------------------------------------------------------------
// The same attention on K / V operands split and laid out ONCE per layer (attn_pack_kv_kernel) instead of by every
// wave for every key tile: per 32-key tile the kernel above spends ~200 of its ~400 VALU instructions per lane on the
// fp32 -> bf16 hi / lo split of K and V and 32 four-byte V loads, against 24 MFMAs (768 clk): VALU-bound 2 : 1.
// Per (b, head) and tile of 32 keys, 4 KiB per plane, MFMA-fragment order:
//   K  [s4 = d/16][lane = key%32 + 32 ((d%16)/8)][8 d]                                     (A rows of S^T = K . Q^T)
//   V^T [s2][t = d/32][lane = d%32 + 32 g][slot j <-> key 16 s2 + 4 g + (j&3) + 8 (j>>2)]    (A rows of O^T += V^T . P^T)
// planes: K hi, K lo, V hi, V lo, `plane_elems` bf16 each; keys >= T are zero.  Same split, same products, same order
// as codec_attn_kernel; the softmax runs in the log2 domain (one multiply less per score), so the two kernels agree
// to rounding, not bit for bit.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_pack_kv_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ planes,
                                                          size_t plane_elems, int T, int d, int tiles) {
    const int tile = blockIdx.x, head = blockIdx.y, b = blockIdx.z, lane = threadIdx.x, g = lane >> 5, ql = lane & 31;
    const long ld = 3L * d;
    const float* base = qkv + (long)b * T * ld + head * 64;
    const size_t tb = (((size_t)b * gridDim.y + head) * tiles + tile) * 2048;
    u32x4_t* Kh = (u32x4_t*)(planes + tb) + lane;
    u32x4_t* Kl = (u32x4_t*)(planes + plane_elems + tb) + lane;
    u32x4_t* Vh = (u32x4_t*)(planes + 2 * plane_elems + tb) + lane;
    u32x4_t* Vl = (u32x4_t*)(planes + 3 * plane_elems + tb) + lane;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int key = tile * 32 + ql;
        const float* kp = base + d + (long)min(key, T - 1) * ld + 8 * g;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 a = key < T ? *(const float4*)(kp + 16 * s4) : z, c = key < T ? *(const float4*)(kp + 16 * s4 + 4) : z;
            u32x4_t hi, lo;
            split8(a, c, hi, lo);
            Kh[s4 * 64] = hi;
            Kl[s4 * 64] = lo;
        }
    }
    const float* vrow = base + 2 * d + ql;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            auto at = [&](int j) {
                const int key = tile * 32 + 16 * s2 + 4 * g + (j & 3) + 8 * (j >> 2);
                return key < T ? vrow[(long)key * ld + 32 * t] : 0.f;
            };
            u32x4_t hi, lo;
            split8(make_float4(at(0), at(1), at(2), at(3)), make_float4(at(4), at(5), at(6), at(7)), hi, lo);
            Vh[(s2 * 2 + t) * 64] = hi;
            Vl[(s2 * 2 + t) * 64] = lo;
        }
}

__global__ __launch_bounds__(256, 2) void codec_attn_packed_kernel(const float* __restrict__ qkv, const uint16_t* __restrict__ planes,
                                                                size_t plane_elems, int tiles, float* __restrict__ att,
                                                                const int* __restrict__ lens, int T, int d, float scale,
                                                                uint16_t* __restrict__ att_lo, int heads, int nqb, int groups) {
    // 1-D grid.  Workgroups are dealt round-robin over the 8 XCDs: all query blocks of one (sequence, head) get the same
    // id % 8, so its K / V planes (768 KiB at T = 1500) are pulled through ONE L2 instead of eight.
    const int L = blockIdx.x, xcd = L & 7, sidx = L >> 3;
    const int qb = sidx % nqb, gi = (sidx / nqb) * 8 + xcd;
    if (gi >= groups) return;
    const int b = gi / heads, head = gi - b * heads;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 5, ql = lane & 31;
    const int q = qb * 128 + wave * 32 + ql;
    if (qb * 128 + wave * 32 >= T) return;
    const int len = lens[b];
    const bool qpad = q >= len;                         // padded (or out-of-range) query: uniform over all T keys
    const long ld = 3L * d;
    u32x4_t qh[4], qlo[4];
    {
        const float* qp = qkv + (long)b * T * ld + head * 64 + (long)min(q, T - 1) * ld + 8 * g;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 a = *(const float4*)(qp + 16 * s4), c = *(const float4*)(qp + 16 * s4 + 4);
            split8(a, c, qh[s4], qlo[s4]);
        }
    }
    f32x16_t o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int kend = qpad ? T : len;                    // every lane of a wave needs keys up to the largest bound
    int kmax = kend;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
    const size_t hb = ((size_t)b * heads + head) * tiles * 256;               // in 16-byte units; a tile = 256 of them per plane
    const u32x4_t* Kh = (const u32x4_t*)planes + hb + lane;
    const u32x4_t* Kl = (const u32x4_t*)(planes + plane_elems) + hb + lane;
    const u32x4_t* Vh = (const u32x4_t*)(planes + 2 * plane_elems) + hb + lane;
    const u32x4_t* Vl = (const u32x4_t*)(planes + 3 * plane_elems) + hb + lane;
    struct KV { u32x4_t kh[4], kl[4], vh[4], vl[4]; };
    auto load = [&](KV& f, int tile) {
        const size_t o4 = (size_t)min(tile, tiles - 1) * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) { f.kh[i] = Kh[o4 + i * 64]; f.kl[i] = Kl[o4 + i * 64]; f.vh[i] = Vh[o4 + i * 64]; f.vl[i] = Vl[o4 + i * 64]; }
    };
    // a padded query row attends uniformly: its scores are 0 (scale 0), its key bound is T.  Scores live in the log2
    // domain (scale * log2 e folded into one multiply, v_exp_f32 direct); pairs (v_pk_mul / v_pk_add) where possible.
    const float scale2 = qpad ? 0.f : scale * 1.44269504088896340736f;
    int kmin = kend;                                    // tiles that end at or before every lane's bound need no mask
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, __shfl_xor(kmin, off, 64));
    auto step = [&](KV& f, int k0, auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        f32x16_t sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.kl[s4], *(bf16x8_t*)&qh[s4], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.kh[s4], *(bf16x8_t*)&qlo[s4], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.kh[s4], *(bf16x8_t*)&qh[s4], sacc, 0, 0, 0);
        }
        // scores of this lane's query: register i <-> key k0 + (i&3) + 8*(i>>2) + 4*g.  exp2(-inf - m) = 0 does the
        // masking of the tiles around the key bounds (tile 0 always holds a valid key: the running maximum is finite
        // from then on).
        f32x2_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f32x2_t{sacc[2 * i], sacc[2 * i + 1]} * scale2;
        if (MASK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * g;
                if (key >= kend) v[i >> 1][i & 1] = -INFINITY;
            }
        }
        float mx = fmaxf(v[0].x, v[0].y);
#pragma unroll
        for (int i = 1; i < 8; ++i) mx = fmaxf(fmaxf(mx, v[i].x), v[i].y);
        mx = max_xor32(mx);
        const float m_new = fmaxf(m_run, mx);
        const float corr = (m_run == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);    // first tile: nothing to rescale
        f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2_t d2 = v[i] - m_new;
            const f32x2_t p2 = {__builtin_amdgcn_exp2f(d2.x), __builtin_amdgcn_exp2f(d2.y)};
            sacc[2 * i] = p2.x; sacc[2 * i + 1] = p2.y;
            ps2 += p2;
        }
        float ps = ps2.x + ps2.y;
        ps = add_xor32(ps);
        l_run = l_run * corr + ps;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2_t r = f32x2_t{o[t][i], o[t][i + 1]} * corr;
                o[t][i] = r.x; o[t][i + 1] = r.y;
            }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4_t ph, pl;
            split8(make_float4(sacc[8 * s2], sacc[8 * s2 + 1], sacc[8 * s2 + 2], sacc[8 * s2 + 3]),
                   make_float4(sacc[8 * s2 + 4], sacc[8 * s2 + 5], sacc[8 * s2 + 6], sacc[8 * s2 + 7]), ph, pl);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.vl[s2 * 2 + t], *(bf16x8_t*)&ph, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.vh[s2 * 2 + t], *(bf16x8_t*)&pl, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.vh[s2 * 2 + t], *(bf16x8_t*)&ph, o[t], 0, 0, 0);
            }
        }
    };
    // two operand sets, ping-pong: the next tile's 16 KiB per wave are in flight under this tile's products and softmax
    KV fa, fb;
    load(fa, 0);
    int k0 = 0;
    for (; k0 + 64 <= kmin; k0 += 64) {                 // tiles wholly inside every lane's key bound: no mask
        load(fb, (k0 >> 5) + 1);
        step(fa, k0, std::false_type{});
        load(fa, (k0 >> 5) + 2);
        step(fb, k0 + 32, std::false_type{});
    }
    for (; k0 < kmax; k0 += 64) {
        load(fb, (k0 >> 5) + 1);
        step(fa, k0, std::true_type{});
        if (k0 + 32 >= kmax) break;
        load(fa, (k0 >> 5) + 2);
        step(fb, k0 + 32, std::true_type{});
    }
    if (q < T) {
        const float inv = 1.0f / l_run;
        const long ob = ((long)b * T + q) * d + head * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int col = head * 64 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * g;
                if (att_lo) {                                  // fragment-packed planes for the pre-split o_proj GEMM
                    const size_t at = xpack_off(b * T + q, col, d);
                    split1(o[t][i] * inv, ((uint16_t*)att)[at], att_lo[at]);
                } else att[ob - head * 64 + col] = o[t][i] * inv;
            }
    }
}

// fused attention of one layer: K / V packed once into the (otherwise unused) score buffer, then the packed kernel
static void launch_codec_attn(MttsCodec* k, hipStream_t st, const float* qkv, float* att, const int* d_lens, int B, int T, int d,
                              int heads, uint16_t* att_lo) {
    const float scale = 1.0f / sqrtf(64.f);
    if (!k->attn_packed) {
        hipLaunchKernelGGL(codec_attn_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, st, qkv, att, d_lens, T, d, scale, att_lo);
        return;
    }
    const int tiles = (T + 31) / 32;
    const size_t plane = (size_t)B * heads * tiles * 2048;               // 4 planes x 2 B <= B heads T ldT x 4 B of k->scores
    hipLaunchKernelGGL(attn_pack_kv_kernel, dim3(tiles, heads, B), dim3(64), 0, st, qkv, (uint16_t*)k->scores, plane, T, d, tiles);
    const int nqb = (T + 127) / 128, groups = B * heads;
    hipLaunchKernelGGL(codec_attn_packed_kernel, dim3((unsigned)(((groups + 7) / 8) * 8 * nqb)), dim3(256), 0, st, qkv,
                       (const uint16_t*)k->scores, plane, tiles, att, d_lens, T, d, scale, att_lo, heads, nqb, groups);
}

// One pre-LN transformer layer (OmniWhisperTransformerLayer, modules.py:187-205) on x [B*T][d].
static int transformer_layer(MttsCodec* k, hipStream_t st, const std::string& p, float* x, float* tmp, float* qkv, float* att,
                             int B, int T, int d, int heads, int ffn, const int* d_lens) {
    const int rows = B * T, hd = d / heads;
    const int ldT = (T + 15) / 16 * 16;
    NEED(ln1w, p + "ln1.w", d); NEED(ln1b, p + "ln1.b", d);
    NEED(wqkv, p + "qkv.w", (size_t)3 * d * d); NEED(bqkv, p + "qkv.b", 3 * d);
    NEED(wo, p + "o.w", (size_t)d * d); NEED(bo, p + "o.b", d);
    NEED(ln2w, p + "ln2.w", d); NEED(ln2b, p + "ln2.b", d);
    NEED(w1, p + "fc1.w", (size_t)ffn * d); NEED(b1, p + "fc1.b", ffn);
    NEED(w2, p + "fc2.w", (size_t)d * ffn); NEED(b2, p + "fc2.b", d);
    if (g_gemm_split && hd == 64 && planes_ok(k, d, d) && planes_ok(k, ffn, ffn)) {
        // decode direction, pre-split operands: every activation that only feeds a GEMM is written as bf16 hi / lo planes
        // by its producer (same buffers: hi plane first, lo plane `rows` rows further), weights are split once
        uint16_t* tlo = (uint16_t*)tmp + pad32(rows) * d;
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ln1w, ln1b, tmp, rows, d, 1e-5f,
                           (const int*)nullptr, T, (long)d, tlo);
        TRYC(gemm_planes(k, st, tmp, rows, wqkv, qkv, rows, 3 * d, d, d, 3 * d, bqkv, 0, nullptr, nullptr, 0, false, 0));
        launch_codec_attn(k, st, qkv, att, d_lens, B, T, d, heads, (uint16_t*)att + pad32(rows) * d);
        TRYC(gemm_planes(k, st, att, rows, wo, x, rows, d, d, d, d, bo, 0, nullptr, x, d, false, 0));      // x += out_proj(att)
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ln2w, ln2b, tmp, rows, d, 1e-5f,
                           (const int*)nullptr, T, (long)d, tlo);
        TRYC(gemm_planes(k, st, tmp, rows, w1, k->big, rows, ffn, d, d, ffn, b1, 1, nullptr, nullptr, 0, true, rows));
        TRYC(gemm_planes(k, st, k->big, rows, w2, x, rows, d, ffn, ffn, d, b2, 0, nullptr, x, d, false, 0));  // x += fc2(gelu(fc1))
        return 0;
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ln1w, ln1b, tmp, rows, d, 1e-5f,
                       (const int*)nullptr, T, (long)d);
    gemm_f32(st, false, tmp, wqkv, qkv, rows, 3 * d, d, d, d, 3 * d, bqkv);
    if (g_gemm_split && hd == 64) {
        // decode direction: one fused launch, the score matrix stays on chip
        launch_codec_attn(k, st, qkv, att, d_lens, B, T, d, heads, nullptr);
        gemm_f32(st, false, att, wo, x, rows, d, d, d, d, d, bo, 0, nullptr, x, d);          // x += out_proj(att)
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ln2w, ln2b, tmp, rows, d, 1e-5f,
                           (const int*)nullptr, T, (long)d);
        gemm_f32(st, false, tmp, w1, k->big, rows, ffn, d, d, d, ffn, b1, 1);
        gemm_f32(st, false, k->big, w2, x, rows, d, ffn, ffn, ffn, d, b2, 0, nullptr, x, d);  // x += fc2(gelu(fc1))
        return 0;
    }
    // S[b,h] = (q k^T) * hd^-0.5   (the reference scales q before the product, modules.py:131)
    gemm_f32(st, false, qkv, qkv + d, k->scores, T, T, hd, 3 * d, 3 * d, ldT, nullptr, 0, nullptr, nullptr, 0, 0,
             1.0f / sqrtf((float)hd), B * heads, heads, (long)T * 3 * d, hd, (long)T * 3 * d, hd, (long)heads * T * ldT,
             (long)T * ldT);
    hipLaunchKernelGGL(softmax_mask_kernel, dim3((unsigned)(((long)B * heads * T + 3) / 4)), dim3(256), 0, st, k->scores,
                       d_lens, heads, T, ldT, (long)B * heads * T);
    // O[b,:,h] = P[b,h] V[b,:,h]
    gemm_f32(st, true, k->scores, qkv + 2 * d, att, T, hd, T, ldT, 3 * d, d, nullptr, 0, nullptr, nullptr, 0, 0, 1.f,
             B * heads, heads, (long)heads * T * ldT, (long)T * ldT, (long)T * 3 * d, hd, (long)T * d, hd);
    gemm_f32(st, false, att, wo, x, rows, d, d, d, d, d, bo, 0, nullptr, x, d);          // x += out_proj(att)
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ln2w, ln2b, tmp, rows, d, 1e-5f,
                       (const int*)nullptr, T, (long)d);
    gemm_f32(st, false, tmp, w1, k->big, rows, ffn, d, d, d, ffn, b1, 1);
    gemm_f32(st, false, k->big, w2, x, rows, d, ffn, ffn, ffn, d, b2, 0, nullptr, x, d);  // x += fc2(gelu(fc1))
    return 0;
}

// codes: device int64 [nq][B][T]; host_lens int32[B]; wav: device f32 [B][T*up*2*hop]
static int detokenize_async(MttsCodec* k, const int64_t* dev_codes, const int32_t* host_lens, int32_t B, int32_t T,
                            float* dev_wav, void* stream);

// Synchronous form: returns after the waveform is complete and the code indices were validated.
extern "C" int32_t mtts_codec_detokenize(MttsCodec* k, const int64_t* dev_codes, const int32_t* host_lens, int32_t B,
                                         int32_t T, float* dev_wav, void* stream) {
    int r = detokenize_async(k, dev_codes, host_lens, B, T, dev_wav, stream);
    if (r) return r;
    int herr = 0;
    CHK(hipMemcpyAsync(&herr, k->d_err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CHK(hipStreamSynchronize((hipStream_t)stream));
    if (herr) { hipMemset(k->d_err, 0, 4); return cfail(MTTS_EINVAL, "code index outside the codebook"); }
    return MTTS_OK;
}

// Asynchronous form for overlapping the codec with the decode loop on another HIP stream: only enqueues.
// mtts_codec_check() later synchronises the stream and reports a bad code index.
extern "C" int32_t mtts_codec_detokenize_async(MttsCodec* k, const int64_t* dev_codes, const int32_t* host_lens, int32_t B,
                                               int32_t T, float* dev_wav, void* stream) {
    return detokenize_async(k, dev_codes, host_lens, B, T, dev_wav, stream);
}
extern "C" int32_t mtts_codec_check(MttsCodec* k, void* stream) {
    if (!k) return cfail(MTTS_EINVAL, "null codec");
    CHK(hipSetDevice(k->device));
    int herr = 0;
    CHK(hipMemcpyAsync(&herr, k->d_err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CHK(hipStreamSynchronize((hipStream_t)stream));
    if (herr) { hipMemset(k->d_err, 0, 4); return cfail(MTTS_EINVAL, "code index outside the codebook"); }
    return MTTS_OK;
}

static int detokenize_async(MttsCodec* k, const int64_t* dev_codes, const int32_t* host_lens, int32_t B, int32_t T,
                            float* dev_wav, void* stream) {
    if (!k || !dev_codes || !host_lens || !dev_wav || B < 1 || T < 1) return cfail(MTTS_EINVAL, "bad argument");
    const MttsCodecConfig& c = k->c;
    if (T > c.adapter_max_pos) return cfail(MTTS_EINVAL, "window of %d codes exceeds adapter_max_pos %d", T, c.adapter_max_pos);
    if (T * c.up_stride > c.dec_max_pos) return cfail(MTTS_EINVAL, "window too long for the acoustic decoder");
    CHK(hipSetDevice(k->device));
    hipStream_t st = (hipStream_t)stream;
    g_gemm_split = k->split_decode;
    {
        int r = ensure_workspace(k, B, T);
        if (r) return r;
    }
    std::vector<int>& lens = k->h_lens;
    std::vector<int>& lens4 = k->h_lens4;
    CHK(hipStreamSynchronize(st));        // the previous call's async length upload must be done before the host copy is reused
    lens.resize(B);
    lens4.resize(B);
    for (int b = 0; b < B; ++b) {
        if (host_lens[b] < 0 || host_lens[b] > T) return cfail(MTTS_EINVAL, "length %d out of range", host_lens[b]);
        lens[b] = host_lens[b];
        lens4[b] = host_lens[b] * c.up_stride;
    }
    CHK(hipMemcpyAsync(k->d_lens, lens.data(), B * 4, hipMemcpyHostToDevice, st));
    CHK(hipMemcpyAsync(k->d_lens4, lens4.data(), B * 4, hipMemcpyHostToDevice, st));
    if (!k->d_cbs) {
        std::vector<float*> cbs(c.nq);
        for (int q = 0; q < c.nq; ++q) {
            NEED(cb, "rvq.codebook." + std::to_string(q), (size_t)c.codebook_size * c.rvq_dim);
            cbs[q] = cb;
        }
        CHK(hipMalloc((void**)&k->d_cbs, c.nq * sizeof(float*)));
        CHK(hipMemcpy(k->d_cbs, cbs.data(), c.nq * sizeof(float*), hipMemcpyHostToDevice));
    }
    const int rows = B * T, da = c.adapter_dim, dd = c.dec_dim, Q = c.quant_out_dim;
    float *A = k->bufA, *Bb = k->bufB, *Cc = k->bufC, *D = k->bufD;
    // C1: RVQ decode + output_proj (weight norm folded at bind time)
    NEED(rvq_w, "rvq.out.w", (size_t)Q * c.rvq_dim); NEED(rvq_b, "rvq.out.b", Q);
    hipLaunchKernelGGL(rvq_gather_kernel, dim3(rows), dim3(128), 0, st, dev_codes, (const float* const*)k->d_cbs, A,
                       c.nq, rows, c.rvq_dim, c.codebook_size, k->d_err);
    // reference layout of codes is [nq][B][T]; rows index b*T+t within each q plane
    gemm_f32(st, false, A, rvq_w, Bb, rows, Q, c.rvq_dim, c.rvq_dim, c.rvq_dim, Q, rvq_b);
    // C2: post_rvq_adapter
    NEED(ap_w, "adapter.proj.w", (size_t)da * Q); NEED(ap_b, "adapter.proj.b", da);
    NEED(a_pe, "adapter.pe", (size_t)c.adapter_max_pos * da);
    gemm_f32(st, false, Bb, ap_w, A, rows, da, Q, Q, Q, da, ap_b, 0, nullptr, a_pe, da, T);     // + PE[t]
    for (int n = 0; n < c.adapter_layers; ++n) {
        int r = transformer_layer(k, st, "adapter.layers." + std::to_string(n) + ".", A, Cc, Bb, D, B, T, da,
                                  c.adapter_heads, c.adapter_ffn, k->d_lens);
        if (r) return r;
    }
    NEED(aln_w, "adapter.ln.w", da); NEED(aln_b, "adapter.ln.b", da);
    NEED(ao_w, "adapter.out.w", (size_t)Q * da); NEED(ao_b, "adapter.out.b", Q);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, A, aln_w, aln_b, Cc, rows, da, 1e-5f,
                       (const int*)k->d_lens, T, (long)da);
    gemm_f32(st, false, Cc, ao_w, Bb, rows, Q, da, da, da, Q, ao_b);
    // C3: upsample ConvTranspose1d(k = s = up): GEMM to [rows][up*dd] == token-major [rows*up][dd]
    const int up = c.up_stride, T4 = T * up, rows4 = B * T4;
    NEED(up_w, "up.w", (size_t)up * dd * Q);
    NEED(d_pe, "dec.pe", (size_t)c.dec_max_pos * dd);
    gemm_f32(st, false, Bb, up_w, A, rows, up * dd, Q, Q, Q, up * dd);
    // C4: acoustic decoder: + PE (row index within the window = m % T4), A viewed as [rows4][dd]
    hipLaunchKernelGGL(add_pe_kernel, dim3((unsigned)(((long)rows4 * dd + 255) / 256)), dim3(256), 0, st, A, d_pe,
                       (long)rows4 * dd, T4, dd);
    for (int n = 0; n < c.dec_layers; ++n) {
        int r = transformer_layer(k, st, "dec.layers." + std::to_string(n) + ".", A, Cc, Bb, D, B, T4, dd, c.dec_heads,
                                  c.dec_ffn, k->d_lens4);
        if (r) return r;
    }
    NEED(dln_w, "dec.ln.w", dd); NEED(dln_b, "dec.ln.b", dd);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows4 + 3) / 4), dim3(256), 0, st, A, dln_w, dln_b, Cc, rows4, dd, 1e-5f,
                       (const int*)k->d_lens4, T4, (long)dd);
    NEED(dc1_w, "dec.deconv1.w", (size_t)3 * dd * dd); NEED(dc1_b, "dec.deconv1.b", dd);
    NEED(dc2_w, "dec.deconv2.w", (size_t)3 * c.mel_bins * dd); NEED(dc2_b, "dec.deconv2.b", c.mel_bins);
    gemm_f32(st, false, Cc, dc1_w, Bb, rows4, 3 * dd, dd, dd, dd, 3 * dd);
    const int P1 = 2 * T4 + 1, T8 = 2 * T4, mel = c.mel_bins;
    hipLaunchKernelGGL(deconv_s2_kernel, dim3(P1, B), dim3(256), 0, st, Bb, dc1_b, A, T4, dd);
    gemm_f32(st, false, A, dc2_w, Bb, B * P1, 3 * mel, dd, dd, dd, 3 * mel);
    hipLaunchKernelGGL(deconv_s1_kernel, dim3(T8, B), dim3(128), 0, st, Bb, dc2_b, Cc, P1, T8, mel);
    // C5: Vocos backbone
    const int rows8 = B * T8, vd = c.voc_dim, vi = c.voc_inter;
    NEED(ve_w, "voc.embed.w", (size_t)vd * 7 * mel); NEED(ve_b, "voc.embed.b", vd);
    NEED(vn_w, "voc.norm.w", vd); NEED(vn_b, "voc.norm.b", vd);
    hipLaunchKernelGGL(im2col7_kernel, dim3(T8, B), dim3(256), 0, st, Cc, Bb, T8, mel);
    gemm_f32(st, false, Bb, ve_w, D, rows8, vd, 7 * mel, 7 * mel, 7 * mel, vd, ve_b);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows8 + 3) / 4), dim3(256), 0, st, D, vn_w, vn_b, A, rows8, vd, 1e-6f,
                       (const int*)nullptr, T8, (long)vd);
    for (int n = 0; n < c.voc_layers; ++n) {
        const std::string p = "voc.blocks." + std::to_string(n) + ".";
        NEED(dw_w, p + "dw.w", (size_t)7 * vd); NEED(dw_b, p + "dw.b", vd);
        NEED(ln_w, p + "ln.w", vd); NEED(ln_b, p + "ln.b", vd);
        NEED(p1_w, p + "pw1.w", (size_t)vi * vd); NEED(p1_b, p + "pw1.b", vi);
        NEED(p2_w, p + "pw2.w", (size_t)vd * vi); NEED(p2_b, p + "pw2.b", vd);
        NEED(gam, p + "gamma", vd);
        if (planes_ok(k, vd, vd) && planes_ok(k, vi, vi)) {
            if (vd == 512)
                if (k->dw_rows == 4)
                    hipLaunchKernelGGL(dwconv_ln512_kernel<4>, dim3((rows8 + 15) / 16), dim3(256), 0, st, A, dw_w, dw_b, ln_w, ln_b, Cc, B, T8,
                                       1e-6f, (uint16_t*)Cc + pad32(rows8) * vd);
                else
                    hipLaunchKernelGGL(dwconv_ln512_kernel<8>, dim3((rows8 + 31) / 32), dim3(256), 0, st, A, dw_w, dw_b, ln_w, ln_b, Cc, B, T8,
                                       1e-6f, (uint16_t*)Cc + pad32(rows8) * vd);
            else
                hipLaunchKernelGGL(dwconv_ln_kernel, dim3((rows8 + 3) / 4), dim3(256), 0, st, A, dw_w, dw_b, ln_w, ln_b, Cc, B, T8,
                                   vd, 1e-6f, (uint16_t*)Cc + pad32(rows8) * vd);
            if (vd == 512 && vi == 4096 && k->fused_pw_rows != 0) {
                // pw1 -> GELU -> pw2 in one launch: the 4096-wide intermediate stays on the CU.  Either the whole call, or
                // (auto) its full rounds of 256 blocks with the last, partial round left to the two launches below.
                long rows_f = fused_pw_pays(rows8, k->fused_pw_rows) ? rows8 : 0;
                if (!rows_f && k->fused_pw_rows < 0) rows_f = ((rows8 + 63) / 64 / 256) * 256 * 64;
                if (rows_f) {
                    uint16_t *w1p = nullptr, *w2p = nullptr;
                    TRYC(weight_planes(k, st, p1_w, vi, vd, &w1p));
                    TRYC(weight_planes_perm(k, st, p2_w, (long)vd * vi, &w2p));
                    launch_vocos_pw_fused(st, (const uint16_t*)Cc, pad32(rows8) * (long)vd, w1p, p1_b, w2p, (long)vd * vi, p2_b, gam, A, (int)rows_f);
                    if (rows_f < rows8) {
                        const int rem = rows8 - (int)rows_f;
                        TRYC(gemm_planes(k, st, Cc, rows8, p1_w, k->big, rem, vi, vd, vd, vi, p1_b, 1, nullptr, nullptr, 0, true, rows8, rows_f, rows8));
                        TRYC(gemm_planes(k, st, k->big, rows8, p2_w, A, rem, vd, vi, vi, vd, p2_b, 0, gam, A, vd, false, 0, rows_f, rows8));
                    }
                    continue;
                }
            }
            TRYC(gemm_planes(k, st, Cc, rows8, p1_w, k->big, rows8, vi, vd, vd, vi, p1_b, 1, nullptr, nullptr, 0, true, rows8));
            TRYC(gemm_planes(k, st, k->big, rows8, p2_w, A, rows8, vd, vi, vi, vd, p2_b, 0, gam, A, vd, false, 0));   // h += gamma * pw2(..)
            continue;
        }
        hipLaunchKernelGGL(dwconv_ln_kernel, dim3((rows8 + 3) / 4), dim3(256), 0, st, A, dw_w, dw_b, ln_w, ln_b, Cc, B, T8,
                           vd, 1e-6f);
        gemm_f32(st, false, Cc, p1_w, k->big, rows8, vi, vd, vd, vd, vi, p1_b, 1);
        gemm_f32(st, false, k->big, p2_w, A, rows8, vd, vi, vi, vi, vd, p2_b, 0, gam, A, vd);   // h += gamma * pw2(..)
    }
    NEED(fl_w, "voc.final_ln.w", vd); NEED(fl_b, "voc.final_ln.b", vd);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows8 + 3) / 4), dim3(256), 0, st, A, fl_w, fl_b, Cc, rows8, vd, 1e-6f,
                       (const int*)nullptr, T8, (long)vd);
    // C6: ISTFT head
    const int nfft = c.n_fft, nb = nfft / 2 + 1, ldsp = (2 * nb + 15) / 16 * 16;
    NEED(hd_w, "voc.head.w", (size_t)2 * nb * vd); NEED(hd_b, "voc.head.b", 2 * nb);
    NEED(basis, "istft.basis", (size_t)ldsp * nfft); NEED(win, "istft.window", nfft);
    gemm_f32(st, false, Cc, hd_w, Bb, rows8, 2 * nb, vd, vd, vd, 2 * nb, hd_b);
    hipLaunchKernelGGL(istft_prep_kernel, dim3(rows8), dim3(256), 0, st, Bb, D, (long)rows8, nb, 2 * nb, ldsp);
    gemm_f32(st, true, D, basis, A, rows8, nfft, ldsp, ldsp, nfft, nfft);
    const long nsamp = (long)T8 * c.hop;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((nsamp + 255) / 256), B), dim3(256), 0, st, A, win, dev_wav, T8,
                       nfft, c.hop);
    CHK(hipGetLastError());
    return MTTS_OK;
}

// ------------------------------------------------------------------------------------
// Encode: 16 kHz waveform chunk (<= 30 s) -> RVQ codes.  Replaces XY_Tokenizer.inference_tokenize
// (reference model.py:55-101); the mel front-end runs on the device (the reference bounces the
// waveform device->host->device through a CPU feature extractor, model.py:66-74).
// dev_wav f32 [B][nsamp] (zero padded), host_lens int32[B] valid samples; dev_codes int64 [nq][B][Tc]
// with Tc = mel_frames / (2*down_pool) = 375; host_code_lens int32[B] out.
// ------------------------------------------------------------------------------------
static int run_audio_encoder(MttsCodec* k, hipStream_t st, const std::string& p, int B, float* A, float* Bb, float* Cc,
                             float* D) {
    const MttsCodecConfig& c = k->c;
    const int d = c.enc_dim, mel = c.mel_bins, T = c.mel_frames, T2 = T / 2;
    NEED(c1w, p + "conv1.w", (size_t)d * 3 * mel); NEED(c1b, p + "conv1.b", d);
    NEED(c2w, p + "conv2.w", (size_t)d * 3 * d); NEED(c2b, p + "conv2.b", d);
    NEED(pe, "enc.pe", (size_t)c.enc_max_pos * d);
    hipLaunchKernelGGL(im2col_kernel, dim3(T, B), dim3(256), 0, st, k->melbuf, Bb, T, T, mel, 3, 1, 1);
    gemm_f32(st, false, Bb, c1w, A, B * T, d, 3 * mel, 3 * mel, 3 * mel, d, c1b, 1);
    hipLaunchKernelGGL(im2col_kernel, dim3(T2, B), dim3(256), 0, st, A, Bb, T, T2, d, 3, 2, 1);
    gemm_f32(st, false, Bb, c2w, A, B * T2, d, 3 * d, 3 * d, 3 * d, d, c2b, 1, nullptr, pe, d, T2);   // GELU then + PE[t]
    for (int n = 0; n < c.enc_layers; ++n) {
        int r = transformer_layer(k, st, p + "layers." + std::to_string(n) + ".", A, Cc, Bb, D, B, T2, d, c.enc_heads,
                                  c.enc_ffn, k->d_lens2);
        if (r) return r;
    }
    return 0;
}

extern "C" int32_t mtts_codec_tokenize(MttsCodec* k, const float* dev_wav, const int32_t* host_lens, int32_t B,
                                       int32_t nsamp, int64_t* dev_codes, int32_t* host_code_lens, void* stream) {
    if (!k || !dev_wav || !host_lens || !dev_codes || !host_code_lens || B < 1 || nsamp < 1) return cfail(MTTS_EINVAL, "bad argument");
    const MttsCodecConfig& c = k->c;
    const int T = c.mel_frames, hop = c.mel_hop, N = T * hop;
    if (nsamp > N) return cfail(MTTS_EINVAL, "chunk of %d samples exceeds %d", nsamp, N);
    if (T % (2 * c.down_pool)) return cfail(MTTS_EINVAL, "mel_frames must divide by 2*down_pool");
    CHK(hipSetDevice(k->device));
    hipStream_t st = (hipStream_t)stream;
    g_gemm_split = 0;            // code ids come out of an argmin: exact f32 products
    const int Tc = T / (2 * c.down_pool), T2 = T / 2;
    {
        int r = ensure_workspace(k, B, std::max(Tc, 1));
        if (r) return r;
    }
    std::vector<int> l2(B), l4(B);
    for (int b = 0; b < B; ++b) {
        if (host_lens[b] < 0 || host_lens[b] > nsamp) return cfail(MTTS_EINVAL, "length %d out of range", host_lens[b]);
        const int mel_len = (host_lens[b] + hop - 1) / hop;       // attention_mask[:, ::hop].sum()
        l2[b] = mel_len / 2;
        l4[b] = l2[b] / c.down_pool;
        host_code_lens[b] = l4[b];
    }
    CHK(hipMemcpyAsync(k->d_lens2, l2.data(), B * 4, hipMemcpyHostToDevice, st));
    CHK(hipMemcpyAsync(k->d_lens4, l4.data(), B * 4, hipMemcpyHostToDevice, st));
    float *A = k->bufA, *Bb = k->bufB, *Cc = k->bufC, *D = k->bufD, *E = k->bufE;
    const int nfft = c.mel_n_fft, nb = nfft / 2 + 1, ldri = (2 * nb + 15) / 16 * 16, ldp = (nb + 15) / 16 * 16;
    const int mel = c.mel_bins, d = c.enc_dim;
    NEED(mwin, "mel.window", nfft); NEED(dft, "mel.dft", (size_t)nfft * ldri); NEED(fb, "mel.fb", (size_t)ldp * mel);
    // log-mel (feature_extractor.py:78-104)
    hipLaunchKernelGGL(mel_frames_kernel, dim3(T, B), dim3(256), 0, st, dev_wav, mwin, Bb, nsamp, N, T, nfft, hop);
    gemm_f32(st, true, Bb, dft, Cc, B * T, ldri, nfft, nfft, ldri, ldri);
    hipLaunchKernelGGL(power_kernel, dim3(B * T), dim3(256), 0, st, Cc, Bb, nb, ldri, ldp);
    gemm_f32(st, true, Bb, fb, k->melbuf, B * T, mel, ldp, ldp, mel, mel);
    hipLaunchKernelGGL(logmel_max_kernel, dim3(B), dim3(256), 0, st, k->melbuf, k->melmax, (long)T * mel);
    hipLaunchKernelGGL(logmel_norm_kernel, dim3((unsigned)(((long)B * T * mel + 255) / 256)), dim3(256), 0, st, k->melbuf,
                       k->melmax, (long)T * mel, (long)B * T * mel);
    const int rows2 = B * T2;
    NEED(pe, "enc.pe", (size_t)c.enc_max_pos * d);
    // semantic encoder -> semantic adapter -> E[:, 0:d]
    {
        int r = run_audio_encoder(k, st, "sem.", B, A, Bb, Cc, D);
        if (r) return r;
        NEED(lw, "sem.ln.w", d); NEED(lb, "sem.ln.b", d);
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows2 + 3) / 4), dim3(256), 0, st, A, lw, lb, Cc, rows2, d, 1e-5f,
                           (const int*)k->d_lens2, T2, (long)d);
        hipLaunchKernelGGL(add_pe_kernel, dim3((unsigned)(((long)rows2 * d + 255) / 256)), dim3(256), 0, st, Cc, pe,
                           (long)rows2 * d, T2, d);
        for (int n = 0; n < c.sem_adapter_layers; ++n) {
            r = transformer_layer(k, st, "semad.layers." + std::to_string(n) + ".", Cc, A, Bb, D, B, T2, d, c.enc_heads,
                                  c.enc_ffn, k->d_lens2);
            if (r) return r;
        }
        NEED(aw, "semad.ln.w", d); NEED(ab, "semad.ln.b", d);
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows2 + 3) / 4), dim3(256), 0, st, Cc, aw, ab, E, rows2, d, 1e-5f,
                           (const int*)k->d_lens2, T2, (long)2 * d);
    }
    // acoustic encoder -> E[:, d:2d]
    {
        int r = run_audio_encoder(k, st, "aco.", B, A, Bb, Cc, D);
        if (r) return r;
        NEED(lw, "aco.ln.w", d); NEED(lb, "aco.ln.b", d);
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows2 + 3) / 4), dim3(256), 0, st, A, lw, lb, E + d, rows2, d, 1e-5f,
                           (const int*)k->d_lens2, T2, (long)2 * d);
    }
    // pre_rvq_adapter
    {
        NEED(pw, "prervq.proj.w", (size_t)d * 2 * d); NEED(pb, "prervq.proj.b", d);
        gemm_f32(st, false, E, pw, A, rows2, d, 2 * d, 2 * d, 2 * d, d, pb, 0, nullptr, pe, d, T2);
        for (int n = 0; n < c.pre_rvq_layers; ++n) {
            int r = transformer_layer(k, st, "prervq.layers." + std::to_string(n) + ".", A, Cc, Bb, D, B, T2, d, c.enc_heads,
                                      c.enc_ffn, k->d_lens2);
            if (r) return r;
        }
        NEED(lw, "prervq.ln.w", d); NEED(lb, "prervq.ln.b", d);
        hipLaunchKernelGGL(layernorm_kernel, dim3((rows2 + 3) / 4), dim3(256), 0, st, A, lw, lb, Cc, rows2, d, 1e-5f,
                           (const int*)k->d_lens2, T2, (long)d);
    }
    // ResidualDownConv (modules.py:452-477): rows of P consecutive frames
    const int P = c.down_pool, di = d * P, rows4 = B * Tc;
    {
        NEED(gw, "down.gate.w", (size_t)di * di); NEED(uw, "down.up.w", (size_t)di * di); NEED(dw, "down.down.w", (size_t)di * di);
        NEED(lw, "down.ln.w", di); NEED(lb, "down.ln.b", di);
        gemm_f32(st, false, Cc, gw, A, rows4, di, di, di, di, di);
        gemm_f32(st, false, Cc, uw, Bb, rows4, di, di, di, di, di);
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)(((long)rows4 * di + 255) / 256)), dim3(256), 0, st, A, Bb,
                           (long)rows4 * di);
        gemm_f32(st, false, A, dw, D, rows4, di, di, di, di, di, nullptr, 0, nullptr, Cc, di);
        hipLaunchKernelGGL(layernorm_wide_kernel, dim3(rows4), dim3(256), 0, st, D, lw, lb, A, di, 1e-5f);
    }
    // ResidualVQ.forward, eval branch
    {
        const int R = c.rvq_dim, K = c.codebook_size;
        NEED(iw, "rvq.in.w", (size_t)R * di); NEED(ib, "rvq.in.b", R);
        gemm_f32(st, false, A, iw, Bb, rows4, R, di, di, di, R, ib);          // residual lives in Bb
        for (int q = 0; q < c.nq; ++q) {
            NEED(cb, "rvq.codebook." + std::to_string(q), (size_t)K * R);
            NEED(cc, "rvq.cc." + std::to_string(q), K);
            gemm_f32(st, false, Bb, cb, Cc, rows4, K, R, R, R, K);
            hipLaunchKernelGGL(vq_argmin_update_kernel, dim3(rows4), dim3(256), 0, st, Cc, cb, cc, Bb,
                               dev_codes + (long)q * rows4, (const int*)k->d_lens4, Tc, K, R);
        }
    }
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(st));
    return MTTS_OK;
}

// tuning hook: time gemm_b3t_kernel alone on synthetic operands.  flags: 1 GELU, 2 gamma, 4 residual, 8 planes out
// (the decoder's combinations: 0, 4, 6, 9); tile_code 0 = the production choice.  avg_us over `iters` launches (HIP events).
__global__ void fill_lcg_kernel(float* p, long n, unsigned seed, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
extern "C" int32_t mtts_k_gemm_planes_bench(int32_t M, int32_t N, int32_t K, int32_t flags, int32_t tile_code, int32_t iters,
                                            float* avg_us, int32_t* code_used) {
    if (M < 1 || N < 1 || K < 64 || K % 64 || N % 4 || iters < 1 || !avg_us) return cfail(MTTS_EINVAL, "gemm_planes_bench: bad argument");
    if (flags != 0 && flags != 4 && flags != 6 && flags != 9) return cfail(MTTS_EINVAL, "gemm_planes_bench: no epilogue for flags %d", flags);
    const long Mp = pad32(M), Np = pad32(N);
    float *a32 = nullptr, *w32 = nullptr, *vecs = nullptr, *res = nullptr, *c = nullptr;
    uint16_t *ap = nullptr, *wp = nullptr;
    CHK(hipMalloc((void**)&a32, (size_t)M * K * 4)); CHK(hipMalloc((void**)&w32, (size_t)N * K * 4));
    CHK(hipMalloc((void**)&ap, (size_t)Mp * K * 4)); CHK(hipMalloc((void**)&wp, (size_t)Np * K * 4));
    CHK(hipMalloc((void**)&vecs, (size_t)N * 8)); CHK(hipMalloc((void**)&res, (size_t)M * N * 4)); CHK(hipMalloc((void**)&c, (size_t)Mp * N * 4));
    CHK(hipMemset(ap, 0, (size_t)Mp * K * 4)); CHK(hipMemset(wp, 0, (size_t)Np * K * 4));
    auto fill = [&](float* p, long n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_lcg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, p, n, seed, sc); };
    fill(a32, (long)M * K, 1u, 1.0f); fill(w32, (long)N * K, 2u, 0.05f); fill(vecs, 2L * N, 3u, 1.0f); fill(res, (long)M * N, 4u, 1.0f);
    hipLaunchKernelGGL(split_pack_w_kernel, dim3((unsigned)(((long)M * K + 255) / 256)), dim3(256), 0, nullptr, a32, ap, ap + Mp * K, M, K);
    hipLaunchKernelGGL(split_pack_w_kernel, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, nullptr, w32, wp, wp + Np * K, N, K);
    const bool planes = flags & 8;
    GemmF32Args g{nullptr, nullptr, c, vecs, (flags & 2) ? vecs + N : nullptr, (flags & 4) ? res : nullptr, M, N, K, (long)K, (long)K, (long)N,
                  (long)N, 0, 1.f, flags & 1, 1, 0, 0, 0, 0, 0, 0, ap, ap + Mp * K, wp, wp + Np * K,
                  planes ? (uint16_t*)c : nullptr, planes ? (uint16_t*)c + Mp * N : nullptr};
    const int code = tile_code ? tile_code : b3t_choose(M, N, K, flags & 1);
    if (code_used) *code_used = code;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    TRYC(b3t_launch(nullptr, g, code));
    TRYC(b3t_launch(nullptr, g, code));
    CHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) TRYC(b3t_launch(nullptr, g, code));
    CHK(hipEventRecord(e1, nullptr));
    CHK(hipEventSynchronize(e1));
    CHK(hipGetLastError());
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.f / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a32); hipFree(w32); hipFree(ap); hipFree(wp); hipFree(vecs); hipFree(res); hipFree(c);
    return MTTS_OK;
}

// unit-test entry: C[M,N] = A[M,K] W[N,K]^T + bias (fp32)
extern "C" int32_t mtts_k_gemm_f32(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N,
                                   int32_t K, int32_t act, void* stream) {
    if (!A || !W || !C || M < 1 || N < 1 || K < 1 || (K % 4)) return cfail(MTTS_EINVAL, "gemm_f32: K must be a multiple of 4");
    g_gemm_split = (act >> 8) & 1;              // act | 0x100: the bf16x3 kernel (test / tuning hook)
    gemm_f32((hipStream_t)stream, false, A, W, C, M, N, K, K, K, N, bias, act & 0xff);
    g_gemm_split = 0;
    CHK(hipGetLastError());
    return MTTS_OK;
}
