// fp32 execution of the AsteroidTTS decode path (`inference.py --dtype fp32`, reference inference.py:27-40,
// generation_utils.py:15-18): fp32 weights, fp32 arithmetic everywhere, fp32 K/V pages, fp32 logits.
//
// This is the strict-parity mode: with no bf16 rounding points the only difference from the reference's CPU run is
// fp32 summation order (~1e-7 relative), so greedy ids are compared with NO margin gate (tests/test_engine_gpu.py).
// It mirrors the same reference code as the bf16 kernels:
//   embedding sum      modeling_asteroid.py:244-248       RMSNorm   modeling_qwen3.py:59-64
//   q/k norm + RoPE    modeling_qwen3.py:148-170,251-252  eager attention (fp32 softmax)  :185-208
//   SwiGLU             :81-83                             8 tied heads  modeling_asteroid.py:412
// The kernels are plain and HBM-bound on the fp32 weights (6.9 GB per decode step at the ASSUMED dims); the tuned path
// is the bf16 one.  Products and sums that the reference does as separate fp32 operations are kept separate
// (__fmul_rn / __fadd_rn) where an FMA contraction would otherwise change the rounding.
//
// `--dtype fp16` (the third value inference.py offers) runs the SAME kernels with `h16` set: weights arrive as fp16 values
// (held in fp32, exact) and every tensor the reference materialises in fp16 -- each Linear output, each elementwise
// result, the attention scores / probabilities / output -- is rounded to fp16 where torch's fp16 CPU kernels round it
// (r16 below; the same points at which the bf16 kernels round to bf16).  Accumulations stay fp32, as torch's do.
#include <hip/hip_fp16.h>

#include <algorithm>

#include "common.h"

__device__ __forceinline__ float r16(float v, int h16) { return h16 ? __half2float(__float2half_rn(v)) : v; }

__device__ __forceinline__ float block_sum_256f(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float block_max_256f(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return t;
}

// x[r] = sum_c emb_c[tok_c] (zeros, then += in channel order); xn = w * (x * rsqrt(mean(x^2) + eps)).  grid R, block 256.
__global__ __launch_bounds__(256) void f32_embed_norm_kernel(const int32_t* __restrict__ tokens, const RowMeta* __restrict__ meta,
                                                             const float* const* __restrict__ tables, const float* __restrict__ norm_w,
                                                             float* __restrict__ x, float* __restrict__ xn, int H, float eps, int h16) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const bool active = meta[r].seq >= 0;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        float v = 0.f;
        if (active)
            for (int c = 0; c < 8; ++c) v = r16(__fadd_rn(v, tables[c][(size_t)tokens[r * 8 + c] * H + i]), h16);
        x[(size_t)r * H + i] = v;
        ss += v * v;
    }
    const float inv = rsqrtf(block_sum_256f(ss, sh) / (float)H + eps);
    for (int i = threadIdx.x; i < H; i += 256) xn[(size_t)r * H + i] = r16(norm_w[i] * r16(__fmul_rn(x[(size_t)r * H + i], inv), h16), h16);
}

// x += y; xn = RMSNorm(x) * w; rows flagged `last` also keep xn in hlast[seq].  grid R, block 256.
__global__ __launch_bounds__(256) void f32_resid_norm_kernel(const float* __restrict__ y, float* __restrict__ x,
                                                             const float* __restrict__ norm_w, float* __restrict__ xn,
                                                             float* __restrict__ hlast, const RowMeta* __restrict__ meta, int H, float eps, int h16) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const RowMeta m = meta[r];
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        const float v = r16(__fadd_rn(x[(size_t)r * H + i], y[(size_t)r * H + i]), h16);
        x[(size_t)r * H + i] = v;
        ss += v * v;
    }
    const float inv = rsqrtf(block_sum_256f(ss, sh) / (float)H + eps);
    for (int i = threadIdx.x; i < H; i += 256) {
        const float o = r16(norm_w[i] * r16(__fmul_rn(x[(size_t)r * H + i], inv), h16), h16);
        xn[(size_t)r * H + i] = o;
        if (hlast && m.seq >= 0 && m.last) hlast[(size_t)m.seq * H + i] = o;
    }
}

// Y[r][n] = sum_k X[r][k] W[n][k] for r < R <= 8: one wave per CPW consecutive output columns, lanes across K
// (each W row is read once, 16 B per lane per step); X rows come from L1/L2.  grid ceil(N / (4*CPW)), block 256.
#define F32_CPW 4
__global__ __launch_bounds__(256) void f32_gemv_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                       float* __restrict__ Y, int R, int N, int K, long ldy, int h16) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + wave) * F32_CPW;
    if (n0 >= N) return;
    float acc[F32_CPW][8];
#pragma unroll
    for (int c = 0; c < F32_CPW; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[c][r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        f32x4_t w[F32_CPW];
#pragma unroll
        for (int c = 0; c < F32_CPW; ++c) {
            const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
            w[c] = (n0 + c < N) ? __builtin_nontemporal_load((const f32x4_t*)(W + (size_t)(n0 + c) * K + k)) : z;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R) {
                const float4 xv = *(const float4*)(X + (size_t)r * K + k);
#pragma unroll
                for (int c = 0; c < F32_CPW; ++c)
                    acc[c][r] += w[c].x * xv.x + w[c].y * xv.y + w[c].z * xv.z + w[c].w * xv.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < F32_CPW; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R) {
                const float s = wave_sum(acc[c][r]);
                if (lane == 0 && n0 + c < N) Y[(size_t)r * ldy + n0 + c] = r16(s, h16);
            }
        }
}

// q/k per-head RMSNorm + RoPE, K/V into the fp32 pages.  qkv row layout: q heads | k heads | v heads.
// K and V pages: [kvh][page][64 tokens][128] fp32.  grid (R, ceil(heads/4)), block 256 (one head per wave).
__global__ __launch_bounds__(256) void f32_qkv_post_kernel(const float* __restrict__ qkv, int ldq, const RowMeta* __restrict__ meta,
                                                           const float* __restrict__ qnw, const float* __restrict__ knw,
                                                           const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                           float* __restrict__ qbuf, float* __restrict__ kcache, float* __restrict__ vcache,
                                                           const int32_t* __restrict__ page_table, int max_pages, int total_pages,
                                                           int nq, int nkv, float eps, int h16) {
    const int r = blockIdx.x, h = blockIdx.y * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (h >= nq + 2 * nkv) return;
    const RowMeta m = meta[r];
    if (m.seq < 0) return;
    const float* src = qkv + (size_t)r * ldq + (size_t)h * MTTS_HD;
    float a = src[l], b = src[l + 64];
    const int page = page_table[(size_t)m.seq * max_pages + (m.pos >> 6)], tok = m.pos & 63;
    if (h >= nq + nkv) {
        float* dst = vcache + (((size_t)(h - nq - nkv) * total_pages + page) * MTTS_PAGE + tok) * MTTS_HD;
        dst[l] = a;
        dst[l + 64] = b;
        return;
    }
    const float* nw = h < nq ? qnw : knw;
    const float inv = rsqrtf(wave_sum(a * a + b * b) / (float)MTTS_HD + eps);
    a = r16(nw[l] * r16(__fmul_rn(a, inv), h16), h16);
    b = r16(nw[l + 64] * r16(__fmul_rn(b, inv), h16), h16);
    const float c = rope_cos[(size_t)m.pos * 64 + l], s = rope_sin[(size_t)m.pos * 64 + l];
    // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)
    const float o1 = r16(__fadd_rn(r16(__fmul_rn(a, c), h16), r16(__fmul_rn(-b, s), h16)), h16);
    const float o2 = r16(__fadd_rn(r16(__fmul_rn(b, c), h16), r16(__fmul_rn(a, s), h16)), h16);
    float* dst = h < nq ? qbuf + ((size_t)r * nq + h) * MTTS_HD
                        : kcache + (((size_t)(h - nq) * total_pages + page) * MTTS_PAGE + tok) * MTTS_HD;
    dst[l] = o1;
    dst[l + 64] = o2;
}

// Eager attention for one (row, head): scores = (q.k) * scale, softmax in fp32, out = P.V.  The row's scores live in
// the scratch row `sc` [len].  grid (nq, R), block 256: a wave takes every 4th token for q.k (lanes across d, one
// coalesced 512-byte K row per step); P.V with thread = (d, half of the tokens).
__global__ __launch_bounds__(256) void f32_attn_kernel(const float* __restrict__ qbuf, const float* __restrict__ kcache,
                                                       const float* __restrict__ vcache, const int32_t* __restrict__ page_table,
                                                       const RowMeta* __restrict__ meta, float* __restrict__ scores,
                                                       float* __restrict__ out, int max_pages, int total_pages, int nq, int nkv,
                                                       float scale, int Lmax, int h16) {
    __shared__ float sh[4];
    __shared__ float part[2][MTTS_HD];
    const int h = blockIdx.x, r = blockIdx.y;
    const RowMeta m = meta[r];
    if (m.seq < 0) return;
    const int len = m.pos + 1, kvh = h / (nq / nkv);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t* pt = page_table + (size_t)m.seq * max_pages;
    float* sc = scores + ((size_t)r * nq + h) * Lmax;
    const float2 q = *(const float2*)(qbuf + ((size_t)r * nq + h) * MTTS_HD + 2 * lane);
    float mx = -INFINITY;
    for (int t = wave; t < len; t += 4) {
        const float* kr = kcache + (((size_t)kvh * total_pages + pt[t >> 6]) * MTTS_PAGE + (t & 63)) * MTTS_HD;
        const float2 kv = *(const float2*)(kr + 2 * lane);
        const float s = r16(__fmul_rn(r16(wave_sum(q.x * kv.x + q.y * kv.y), h16), scale), h16);
        if (lane == 0) sc[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max_256f(mx, sh);            // (every lane of a wave holds the wave's maximum)
    float sm = 0.f;
    for (int t = threadIdx.x; t < len; t += 256) {
        const float e = expf(sc[t] - mx);
        sc[t] = e;
        sm += e;
    }
    sm = block_sum_256f(sm, sh);
    const int d = threadIdx.x & 127, half = threadIdx.x >> 7;
    float acc = 0.f;
    for (int t = half; t < len; t += 2) {
        const float p = r16(sc[t] / sm, h16);
        acc += p * vcache[(((size_t)kvh * total_pages + pt[t >> 6]) * MTTS_PAGE + (t & 63)) * MTTS_HD + d];
    }
    part[half][d] = acc;
    __syncthreads();
    if (half == 0) out[((size_t)r * nq + h) * MTTS_HD + d] = r16(part[0][d] + part[1][d], h16);
}

// act[r][i] = silu(gate) * up, gate = gu[r][i], up = gu[r][I + i]
__global__ void f32_swiglu_kernel(const float* __restrict__ gu, float* __restrict__ act, int R, int I, int h16) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * I) return;
    const size_t r = idx / I, i = idx % I;
    const float g = gu[r * 2 * I + i], u = gu[r * 2 * I + I + i];
    act[idx] = r16(__fmul_rn(r16(g / (1.0f + expf(-g)), h16), u), h16);
}
// fp16 mode, rows through the MFMA GEMM: round its fp32 outputs in place
__global__ void f32_round16_kernel(float* __restrict__ y, int R, int N, long ldy) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * N) return;
    float* p = y + (idx / N) * ldy + idx % N;
    *p = r16(*p, 1);
}

void launch_f32_embed_norm(const int32_t* tokens, const RowMeta* meta, const float* const* tables, const float* norm_w, float* x,
                           float* xn, int R, int H, float eps, int h16, hipStream_t st) {
    hipLaunchKernelGGL(f32_embed_norm_kernel, dim3(R), dim3(256), 0, st, tokens, meta, tables, norm_w, x, xn, H, eps, h16);
}
void launch_f32_resid_norm(const float* y, float* x, const float* norm_w, float* xn, float* hlast, const RowMeta* meta, int R,
                           int H, float eps, int h16, hipStream_t st) {
    hipLaunchKernelGGL(f32_resid_norm_kernel, dim3(R), dim3(256), 0, st, y, x, norm_w, xn, hlast, meta, H, eps, h16);
}
void mtts_gemm_f32_exact(hipStream_t st, const float* A, const float* W, float* C, int M, int N, int K, long ldc);   // codec.hip
// Decode rows (one dialogue each; `gemv`) go through the GEMV kernel 8 rows per launch whatever the batch size, prefill
// passes through the exact-f32 MFMA GEMM whatever their row count: which kernel -- hence which fp32 summation order -- a
// dialogue's numbers come from must not depend on how many other dialogues share its batch.
void launch_f32_linear(const float* W, const float* X, float* Y, int R, int N, int K, long ldy, int h16, bool gemv, hipStream_t st) {
    if (gemv) {
        for (int r0 = 0; r0 < R; r0 += 8)
            hipLaunchKernelGGL(f32_gemv_kernel, dim3((N + 4 * F32_CPW - 1) / (4 * F32_CPW)), dim3(256), 0, st, W, X + (size_t)r0 * K,
                               Y + (size_t)r0 * ldy, std::min(8, R - r0), N, K, ldy, h16);
        return;
    }
    mtts_gemm_f32_exact(st, X, W, Y, R, N, K, ldy);
    if (h16) {
        const size_t total = (size_t)R * N;
        hipLaunchKernelGGL(f32_round16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Y, R, N, ldy);
    }
}
void launch_f32_qkv_post(const float* qkv, int ldq, const RowMeta* meta, const float* qnw, const float* knw, const float* cosb,
                         const float* sinb, float* qbuf, float* kcache, float* vcache, const int32_t* page_table, int max_pages,
                         int total_pages, int R, int nq, int nkv, float eps, int h16, hipStream_t st) {
    hipLaunchKernelGGL(f32_qkv_post_kernel, dim3(R, (nq + 2 * nkv + 3) / 4), dim3(256), 0, st, qkv, ldq, meta, qnw, knw, cosb, sinb,
                       qbuf, kcache, vcache, page_table, max_pages, total_pages, nq, nkv, eps, h16);
}
void launch_f32_attn(const float* qbuf, const float* kcache, const float* vcache, const int32_t* page_table, const RowMeta* meta,
                     float* scores, float* out, int R, int max_pages, int total_pages, int nq, int nkv, float scale, int Lmax,
                     int h16, hipStream_t st) {
    hipLaunchKernelGGL(f32_attn_kernel, dim3(nq, R), dim3(256), 0, st, qbuf, kcache, vcache, page_table, meta, scores, out,
                       max_pages, total_pages, nq, nkv, scale, Lmax, h16);
}
void launch_f32_swiglu(const float* gu, float* act, int R, int I, int h16, hipStream_t st) {
    const size_t total = (size_t)R * I;
    hipLaunchKernelGGL(f32_swiglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gu, act, R, I, h16);
}
