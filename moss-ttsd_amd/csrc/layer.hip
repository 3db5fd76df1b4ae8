// Elementwise / normalisation kernels of one decoder layer, each fused with the
// split-K reduction of the GEMM in front of it.  All rounding points follow the
// reference's CPU bf16 execution (oracle/asteroid_oracle.py lists them).
#include "common.h"

// block-wide sum over 256 threads
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------------------
// K1 + first RMSNorm.  reference modeling_asteroid.py:244-248: zeros, then
// `+= Emb_c(ids[...,c])` for c = 0..7 in the weight dtype (8 bf16 roundings),
// followed by layer 0's input_layernorm (modeling_qwen3.py:59-64).
// grid = R rows, block 256.  Writes the residual stream x[R][H] (bf16) and the
// normalised activations in X-fragment layout.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_norm_kernel(
    const int32_t* __restrict__ tokens /*[R][8]*/, const RowMeta* __restrict__ meta,
    const uint16_t* const* __restrict__ tables /*[8]*/, const uint16_t* __restrict__ norm_w,
    uint16_t* __restrict__ x, uint16_t* __restrict__ xn_packed, int H, float eps) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const bool active = meta[r].seq >= 0;
    // the 8 row addresses are block-uniform; each thread then owns 8 consecutive elements per 2048-wide chunk,
    // so the gather is 8 independent 16-byte loads in flight (one per channel), summed in channel order
    const uint16_t* row[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) row[c] = tables[c] + (size_t)(active ? tokens[r * 8 + c] : 0) * H;
    constexpr int MAXC = 4;                       // H <= 8192
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int cc = 0; cc < MAXC; ++cc) {
        const int i0 = cc * 2048 + threadIdx.x * 8;
        if (i0 < H) {
            u32x4_t e[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) e[c] = *(const u32x4_t*)(row[c] + i0);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[cc][j] = 0.f;
            if (active) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    v[cc][0] = rbf(v[cc][0] + bflo(e[c].x)); v[cc][1] = rbf(v[cc][1] + bfhi(e[c].x));
                    v[cc][2] = rbf(v[cc][2] + bflo(e[c].y)); v[cc][3] = rbf(v[cc][3] + bfhi(e[c].y));
                    v[cc][4] = rbf(v[cc][4] + bflo(e[c].z)); v[cc][5] = rbf(v[cc][5] + bfhi(e[c].z));
                    v[cc][6] = rbf(v[cc][6] + bflo(e[c].w)); v[cc][7] = rbf(v[cc][7] + bfhi(e[c].w));
                }
            }
            u32x4_t xo;
            xo.x = pack2(v[cc][0], v[cc][1]); xo.y = pack2(v[cc][2], v[cc][3]);
            xo.z = pack2(v[cc][4], v[cc][5]); xo.w = pack2(v[cc][6], v[cc][7]);
            *(u32x4_t*)(x + (size_t)r * H + i0) = xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[cc][j] * v[cc][j];
        }
    }
    float tot = block_sum_256(ss, sh);
    float inv = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
    for (int cc = 0; cc < MAXC; ++cc) {
        const int i0 = cc * 2048 + threadIdx.x * 8;
        if (i0 < H) {
            const u32x4_t w = *(const u32x4_t*)(norm_w + i0);
            u32x4_t y;
            y.x = pack2(bflo(w.x) * rbf(v[cc][0] * inv), bfhi(w.x) * rbf(v[cc][1] * inv));
            y.y = pack2(bflo(w.y) * rbf(v[cc][2] * inv), bfhi(w.y) * rbf(v[cc][3] * inv));
            y.z = pack2(bflo(w.z) * rbf(v[cc][4] * inv), bfhi(w.z) * rbf(v[cc][5] * inv));
            y.w = pack2(bflo(w.w) * rbf(v[cc][6] * inv), bfhi(w.w) * rbf(v[cc][7] * inv));
            *(u32x4_t*)(xn_packed + xpack_off(r, i0, H)) = y;
        }
    }
}

// ---------------------------------------------------------------------------
// [split-K reduce] + residual add + RMSNorm (modeling_qwen3.py:299-324, :59-64).
//   y = bf16(sum_ks partial)            (the Linear's bf16 output)
//   x = bf16(x + y)                     (residual, bf16 add)
//   xn = bf16(w * bf16(x * rsqrt(mean(x^2)+eps)))
// xn goes to the X-fragment layout for the next GEMM; rows flagged `last` also
// store xn row-major into hlast[seq] (final norm -> LM heads).
// grid = R, block 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resid_norm_kernel(
    const float* __restrict__ partial, int ksplit, int Npad, uint16_t* __restrict__ x,
    const uint16_t* __restrict__ norm_w, uint16_t* __restrict__ xn_packed, uint16_t* __restrict__ hlast,
    const RowMeta* __restrict__ meta, int H, float eps) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const size_t kstride = (size_t)MTTS_PFCAP * Npad;
    // each thread owns 8 consecutive elements per 2048-wide chunk (16-byte loads/stores; the 8
    // elements are one 16-byte group of the X-fragment layout); values stay in registers.
    constexpr int MAXC = 4;                       // H <= 8192
    float v[MAXC][8];
    u32x4_t nw[MAXC];                             // norm weights: loaded with everything else, used after the sum
    const RowMeta m = meta[r];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i0 = c * 2048 + threadIdx.x * 8;
        if (i0 < H) {
            const float* p0 = partial + (size_t)r * Npad + i0;
            const u32x4_t xo = *(const u32x4_t*)(x + (size_t)r * H + i0);
            nw[c] = *(const u32x4_t*)(norm_w + i0);
            // split-K slabs: up to 8 x 32 bytes per thread in flight at once (a data-dependent loop would pay
            // one memory round trip per slab); the sum keeps the fixed order k = 0, 1, 2, ...
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            for (int k0 = 0; k0 < ksplit; k0 += 8) {
                float4 ta[8], tb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* pk = p0 + (size_t)min(k0 + j, ksplit - 1) * kstride;
                    ta[j] = *(const float4*)pk;
                    tb[j] = *(const float4*)(pk + 4);
                }
                if (k0 == 0) { a = ta[0]; b = tb[0]; }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j < ksplit && k0 + j > 0) {
                        a.x += ta[j].x; a.y += ta[j].y; a.z += ta[j].z; a.w += ta[j].w;
                        b.x += tb[j].x; b.y += tb[j].y; b.z += tb[j].z; b.w += tb[j].w;
                    }
            }
            v[c][0] = rbf(bflo(xo.x) + rbf(a.x)); v[c][1] = rbf(bfhi(xo.x) + rbf(a.y));
            v[c][2] = rbf(bflo(xo.y) + rbf(a.z)); v[c][3] = rbf(bfhi(xo.y) + rbf(a.w));
            v[c][4] = rbf(bflo(xo.z) + rbf(b.x)); v[c][5] = rbf(bfhi(xo.z) + rbf(b.y));
            v[c][6] = rbf(bflo(xo.w) + rbf(b.z)); v[c][7] = rbf(bfhi(xo.w) + rbf(b.w));
            u32x4_t xn;
            xn.x = pack2(v[c][0], v[c][1]); xn.y = pack2(v[c][2], v[c][3]);
            xn.z = pack2(v[c][4], v[c][5]); xn.w = pack2(v[c][6], v[c][7]);
            *(u32x4_t*)(x + (size_t)r * H + i0) = xn;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[c][j] * v[c][j];
        }
    }
    float tot = block_sum_256(ss, sh);
    float inv = 1.0f / sqrtf(tot / (float)H + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i0 = c * 2048 + threadIdx.x * 8;
        if (i0 < H) {
            const u32x4_t w = nw[c];
            u32x4_t y;
            y.x = pack2(bflo(w.x) * rbf(v[c][0] * inv), bfhi(w.x) * rbf(v[c][1] * inv));
            y.y = pack2(bflo(w.y) * rbf(v[c][2] * inv), bfhi(w.y) * rbf(v[c][3] * inv));
            y.z = pack2(bflo(w.z) * rbf(v[c][4] * inv), bfhi(w.z) * rbf(v[c][5] * inv));
            y.w = pack2(bflo(w.w) * rbf(v[c][6] * inv), bfhi(w.w) * rbf(v[c][7] * inv));
            if (xn_packed) *(u32x4_t*)(xn_packed + xpack_off(r, i0, H)) = y;
            if (hlast && m.seq >= 0 && m.last) *(u32x4_t*)(hlast + (size_t)m.seq * H + i0) = y;
        }
    }
}

// ---------------------------------------------------------------------------
// QKV epilogue: split-K reduce -> bf16; per-head q/k RMSNorm (modeling_qwen3.py
// :251-252); RoPE in bf16 with three roundings (:148-170); q to qbuf, k/v into
// the paged cache (replaces DynamicCache.update's torch.cat, :258-259).
//   cache layout per layer: [kvh][page][16 KiB]  (a sequence's consecutive pages of one head are contiguous)
//   K page layout: [d/8][token 0..63][8]   (token-major inner: the
//   score kernel reads one token per lane with no cross-lane reduction)
//   V page layout: [page][kvh][token pair 0..31][d 0..127][2]  (P.V as v_dot2c against packed p pairs)
// grid = (R, ceil((nq + 2*nkv)/4)), block 256: one head per wave (lane l owns d = l and d = l+64).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qkv_post_kernel(
    const float* __restrict__ partial, int ksplit, int Npad, const RowMeta* __restrict__ meta,
    const uint16_t* __restrict__ qnorm_w, const uint16_t* __restrict__ knorm_w,
    const uint16_t* __restrict__ rope_cos, const uint16_t* __restrict__ rope_sin,
    uint16_t* __restrict__ qbuf /*[R][nq][128]*/, uint16_t* __restrict__ kcache, uint16_t* __restrict__ vcache,
    const int32_t* __restrict__ page_table, int max_pages, int total_pages, int nq, int nkv, float eps) {
    const int r = blockIdx.x, h = blockIdx.y * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (h >= nq + 2 * nkv) return;                       // whole wave
    const int col = h * MTTS_HD;
    const size_t kstride = (size_t)MTTS_PFCAP * Npad;
    const float* p0 = partial + (size_t)r * Npad + col + l;
    // all loads go out before anything is waited for: the split-K slabs (fixed summation order), then the
    // row's position -> page / RoPE row.  Idle rows (seq < 0) run on clamped indices and store nothing.
    float a = 0.f, b = 0.f;
    const RowMeta m = meta[r];
    const bool live = m.seq >= 0;
    const int pos = live ? m.pos : 0;
    const int page = page_table[(size_t)(live ? m.seq : 0) * max_pages + (pos >> 6)];
    const float c = bf2f(rope_cos[(size_t)pos * 64 + l]);
    const float s = bf2f(rope_sin[(size_t)pos * 64 + l]);
    for (int k0 = 0; k0 < ksplit; k0 += 8) {
        float ta[8], tb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* pk = p0 + (size_t)min(k0 + j, ksplit - 1) * kstride;
            ta[j] = pk[0];
            tb[j] = pk[64];
        }
        if (k0 == 0) { a = ta[0]; b = tb[0]; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < ksplit && k0 + j > 0) { a += ta[j]; b += tb[j]; }
    }
    a = rbf(a);
    b = rbf(b);
    if (!live) return;
    const int tok = m.pos & 63;
    if (h >= nq + nkv) {   // V head: no norm, no rope
        const int kvh = h - nq - nkv;
        // V page layout [token pair][d][2]: element (tok, d) at ((tok>>1)*128 + d)*2 + (tok&1)
        uint16_t* dst = vcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD) + (size_t)(tok >> 1) * (MTTS_HD * 2) + (tok & 1);
        dst[2 * l] = f2bf(a);
        dst[2 * (l + 64)] = f2bf(b);
        return;
    }
    const uint16_t* nw = (h < nq) ? qnorm_w : knorm_w;
    float ss = wave_sum(a * a + b * b);
    float inv = 1.0f / sqrtf(ss / (float)MTTS_HD + eps);
    a = rbf(bf2f(nw[l]) * rbf(a * inv));
    b = rbf(bf2f(nw[l + 64]) * rbf(b * inv));
    // q_embed = q*cos + rotate_half(q)*sin ; rotate_half = cat(-x2, x1)
    float o1 = rbf(rbf(a * c) + rbf(-b * s));
    float o2 = rbf(rbf(b * c) + rbf(a * s));
    if (h < nq) {
        uint16_t* dst = qbuf + ((size_t)r * nq + h) * MTTS_HD;
        dst[l] = f2bf(o1);
        dst[l + 64] = f2bf(o2);
    } else {
        const int kvh = h - nq;
        uint16_t* base = kcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD);
        // element (tok, d) at ((d/8)*64 + tok)*8 + d%8
        base[(((l >> 3) * 64) + tok) * 8 + (l & 7)] = f2bf(o1);
        base[((((l + 64) >> 3) * 64) + tok) * 8 + (l & 7)] = f2bf(o2);
    }
}

// Standalone RMSNorm over rows (unit-test entry point mtts_k_rmsnorm).
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                           uint16_t* __restrict__ y, int n, float eps) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = bf2f(x[(size_t)r * n + i]);
        ss += v * v;
    }
    float tot = block_sum_256(ss, sh);
    float inv = 1.0f / sqrtf(tot / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        float v = bf2f(x[(size_t)r * n + i]);
        y[(size_t)r * n + i] = f2bf(bf2f(w[i]) * rbf(v * inv));
    }
}

void launch_embed_norm(const int32_t* tokens, const RowMeta* meta, const uint16_t* const* tables, const void* norm_w,
                       void* x, void* xn_packed, int R, int H, float eps, hipStream_t st) {
    // rows that are not running carry seq < 0 in their RowMeta: the kernels read no separate stop flag
    hipLaunchKernelGGL(embed_norm_kernel, dim3(R), dim3(256), 0, st, tokens, meta, tables, (const uint16_t*)norm_w,
                       (uint16_t*)x, (uint16_t*)xn_packed, H, eps);
}
void launch_resid_norm(const float* partial, int ksplit, int Npad, void* x, const void* norm_w, void* xn_packed,
                       void* hlast, const RowMeta* meta, int R, int H, float eps, hipStream_t st) {
    hipLaunchKernelGGL(resid_norm_kernel, dim3(R), dim3(256), 0, st, partial, ksplit, Npad, (uint16_t*)x,
                       (const uint16_t*)norm_w, (uint16_t*)xn_packed, (uint16_t*)hlast, meta, H, eps);
}
void launch_qkv_post(const float* partial, int ksplit, int Npad, const RowMeta* meta, const void* qnw, const void* knw,
                     const void* cosb, const void* sinb, void* qbuf, void* kcache, void* vcache,
                     const int32_t* page_table, int max_pages, int total_pages, int R, int nq, int nkv, float eps,
                     hipStream_t st) {
    hipLaunchKernelGGL(qkv_post_kernel, dim3(R, (nq + 2 * nkv + 3) / 4), dim3(256), 0, st, partial, ksplit, Npad, meta,
                       (const uint16_t*)qnw, (const uint16_t*)knw, (const uint16_t*)cosb, (const uint16_t*)sinb,
                       (uint16_t*)qbuf, (uint16_t*)kcache, (uint16_t*)vcache, page_table, max_pages, total_pages, nq, nkv, eps);
}
void launch_rmsnorm_rows(const void* x, const void* w, void* y, int rows, int n, float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(rows), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)w,
                       (uint16_t*)y, n, eps);
}

// Measurement aid: pseudo-random bf16 values in (-1, 1) (profiling runs jump to a long context without replaying it;
// timing HBM-bound kernels on zeros would flatter them).
__global__ void fill_random_bf16_kernel(uint16_t* __restrict__ p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float v = ((float)(h & 0xffffff) / 8388608.0f) - 1.0f;
        p[i] = f2bf(v);
    }
}
void launch_fill_random_bf16(void* p, size_t n, uint32_t seed, hipStream_t st) {
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(4096), dim3(256), 0, st, (uint16_t*)p, n, seed);
}

// Page-table edits handed over as launch arguments (engine.hip: pool_flush).
struct PageEdits { int32_t n; int32_t idx[31]; int32_t val[31]; };
__global__ void set_pages_kernel(int32_t* __restrict__ table, PageEdits ed) {
    if ((int)threadIdx.x < ed.n) table[ed.idx[threadIdx.x]] = ed.val[threadIdx.x];
}
void launch_set_pages(int32_t* table, const PageEdits& ed, hipStream_t st) {
    hipLaunchKernelGGL(set_pages_kernel, dim3(1), dim3(64), 0, st, table, ed);
}

// Unit-test helper (mtts_k_paged_attn_decode): row-major K / V [seq][Lmax][nkv][128] bf16 -> the paged cache layouts
// (K page [d/8][token][8], V page [token pair][d][2]) through an arbitrary page table.
__global__ void pack_kv_pages_kernel(const uint16_t* __restrict__ K, const uint16_t* __restrict__ V, uint16_t* __restrict__ kcache,
                                     uint16_t* __restrict__ vcache, const int32_t* __restrict__ page_table, const int32_t* __restrict__ lens,
                                     int Lmax, int nkv, int max_pages, int total_pages) {
    const int seq = blockIdx.z, kvh = blockIdx.y, tok = blockIdx.x, d = threadIdx.x;
    if (tok >= lens[seq]) return;
    const int page = page_table[(size_t)seq * max_pages + (tok >> 6)], t = tok & 63;
    const size_t src = (((size_t)seq * Lmax + tok) * nkv + kvh) * MTTS_HD + d;
    const size_t base = ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD);
    kcache[base + (((d >> 3) * 64) + t) * 8 + (d & 7)] = K[src];
    vcache[base + ((size_t)(t >> 1) * MTTS_HD + d) * 2 + (t & 1)] = V[src];
}
void launch_pack_kv_pages(const void* K, const void* V, void* kcache, void* vcache, const int32_t* page_table, const int32_t* lens,
                          int S, int Lmax, int nkv, int max_pages, int total_pages, hipStream_t st) {
    hipLaunchKernelGGL(pack_kv_pages_kernel, dim3(Lmax, nkv, S), dim3(MTTS_HD), 0, st, (const uint16_t*)K, (const uint16_t*)V,
                       (uint16_t*)kcache, (uint16_t*)vcache, page_table, lens, Lmax, nkv, max_pages, total_pages);
}
// bf16 [n] -> fp32 [n] (exact): lets the unit-test entry points feed bf16 Linear outputs to kernels that read fp32 slabs
__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ a, float* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = bf2f(a[i]);
}
void launch_bf16_to_f32(const void* a, float* b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint16_t*)a, b, n);
}
// X-fragment layout [rows/32 tiles][K/16][lane][8] -> row-major [R][K]
__global__ void unpack_rows_kernel(const uint16_t* __restrict__ packed, uint16_t* __restrict__ out, int R, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * K) return;
    const int r = (int)(i / K), k = (int)(i % K);
    out[i] = packed[xpack_off(r, k, K)];
}
void launch_unpack_rows(const void* packed, void* out, int R, int K, hipStream_t st) {
    const size_t n = (size_t)R * K;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint16_t*)packed, (uint16_t*)out, R, K);
}
