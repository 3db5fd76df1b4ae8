// Skinny bf16 GEMM for the decode step: Y[32 rows, N] = X[32, K] * W[N, K]^T.
//
// Replaces every nn.Linear on the decode path (q/k/v/o_proj, gate/up/down_proj:
// transformers modeling_qwen3.py:81-83,223-238; the 8 tied lm_heads: reference
// modeling_asteroid.py:412).  The weight is streamed exactly once per call and
// is the only HBM traffic that matters (X is 128 KiB and lives in L2).
//
// Layout: W and X are stored in MFMA-fragment order (common.h wpack_off /
// xpack_off) so that every wave-instruction is one contiguous KiB.  One block
// owns one 32-row N-tile; its waves split the K range and reduce through LDS.
// With v_mfma_f32_32x32x16_bf16, A = W tile (32 n x 16 k), B = X^T (16 k x 32
// rows): D[n][row], lane holds column `row = lane&31` and 16 n values.
#include "common.h"

enum { EPI_PARTIAL = 0, EPI_BF16 = 1, EPI_SILU = 2 };

// grid = (N/32, ksplit); block = WAVES*64.  MB = row tiles (of 32) that share the weight stream.
// kt_per_split: k-tiles (of 16) per blockIdx.y; kt_per_wave: per wave inside that.
template <int WAVES, int EPI, int MB>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(
    const u32x4_t* __restrict__ Wp, const u32x4_t* __restrict__ Xp, int KT, int kt_per_split,
    int kt_per_wave, float* __restrict__ partial, uint16_t* __restrict__ out, int Npad, int n_valid) {
    __shared__ float red[WAVES][16][64];
    const int nt = blockIdx.x, ks = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int kt0 = ks * kt_per_split + wave * kt_per_wave;
    int kt1 = min(min(kt0 + kt_per_wave, (ks + 1) * kt_per_split), KT);
    f32x16_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
    const u32x4_t* wp = Wp + ((size_t)nt * KT + kt0) * 64 + lane;
    const u32x4_t* xp = Xp + (size_t)kt0 * 64 + lane;
    const size_t xtile = (size_t)KT * 64;            // one 32-row activation tile, in 16-byte units
    constexpr int U = (MB == 1) ? 8 : 4;
    int n = kt1 - kt0;
    int i = 0;
    for (; i + U <= n; i += U) {
        u32x4_t a[U], b[U][MB];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(wp + (size_t)(i + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) b[u][mb] = xp[(size_t)(i + u) * 64 + mb * xtile];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a[u], *(bf16x8_t*)&b[u][mb], acc[mb], 0, 0, 0);
    }
    for (; i < n; ++i) {
        u32x4_t a = __builtin_nontemporal_load(wp + (size_t)i * 64);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            u32x4_t b = xp[(size_t)i * 64 + mb * xtile];
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a, *(bf16x8_t*)&b, acc[mb], 0, 0, 0);
        }
    }
    const int l2 = threadIdx.x & 63;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        if (mb) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[mb][r];
        __syncthreads();
        for (int q = threadIdx.x >> 6; q < 4; q += WAVES) {      // q: register quad 4q..4q+3
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = red[0][4 * q + j][l2];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) s += red[w][4 * q + j][l2];
                v[j] = s;
            }
            const int row = mb * 32 + (l2 & 31);           // activation row (sequence / token)
            const int nl = 8 * q + 4 * (l2 >> 5);          // first of 4 consecutive n in the tile
            const int n0 = nt * 32 + nl;
            if (EPI == EPI_PARTIAL) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(partial + ((size_t)ks * MTTS_PFCAP + row) * Npad + n0) = o;
            } else if (EPI == EPI_BF16) {
                // row-major [rows][Npad] bf16 (logits; columns >= n_valid are padding): rows are 64-byte aligned, a lane's four
                // consecutive columns go out as one 8-byte store
                uint16_t* o = out + (size_t)row * Npad + n0;
                if (n0 + 3 < n_valid) *(u32x2_t*)o = u32x2_t{pack2(v[0], v[1]), pack2(v[2], v[3])};
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n0 + j < n_valid) o[j] = f2bf(v[j]);
                }
            } else {
                // rows interleaved gate,up,gate,up: SwiGLU (modeling_qwen3.py:81-83) with the
                // reference's bf16 rounding points: gate, up -> bf16; silu(gate) -> bf16; product -> bf16.
                // Output goes straight into the X-fragment layout of the down projection (K = Npad/2).
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float g = rbf(v[2 * j]), u = rbf(v[2 * j + 1]);
                    float a = rbf(g / (1.0f + expf(-g)));
                    int idx = (n0 >> 1) + j;
                    out[xpack_off(row, idx, Npad >> 1)] = f2bf(a * u);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Small-batch decode GEMM (1..SMALL_RP rows): the same weight stream and MFMA tile as gemm_skinny_kernel, but
//   * the B operand (activations) is built by a PROLOGUE into LDS, compact [k/8][SMALL_RP rows][8 bf16]:
//       PRO_NORM     residual + split-K slabs of the previous Linear -> x', RMSNorm(x')*w   (replaces resid_norm_kernel)
//       PRO_COMBINE  sum of the P.V chunk partials -> attention output                       (replaces attn_combine_kernel)
//       PRO_ROWS     copy of a row-major activation (the SwiGLU output)
//     every block redoes it (a few KiB per row) while its first weight tiles are in flight;
//   * only the live rows are stored.
// The prologues repeat the arithmetic of the kernels they replace, operation for operation and in the same order, so
// the small path gives the same bits as the general path (tested).  MFMA lanes whose row is >= SMALL_RP feed zeros.
// grid = (N/32, ksplit), block = WAVES*64 (WAVES 4 or 8: the norm prologue works in groups of 256 threads).
// ---------------------------------------------------------------------------------------------------
enum { EPI_SILU_RM = 3 };      // SwiGLU, row-major bf16 [rows][N/2] output

template <int WAVES, int EPI, int PRO>
__global__ __launch_bounds__(WAVES * 64, 4) void gemv_small_kernel(
    const u32x4_t* __restrict__ Wp, int KT, int kt_per_split, int kt_per_wave, float* __restrict__ partial,
    uint16_t* __restrict__ out, int Npad, int n_valid, SmallPro pr) {
    __shared__ float red[WAVES][16][64];
    __shared__ float sh[WAVES / 4][4];
    extern __shared__ __attribute__((aligned(16))) u32x4_t xs[];       // [(kt - kt_base)*2 + half][SMALL_RP]
    const int nt = blockIdx.x, ks = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kb0 = ks * kt_per_split;                                 // this block's k-tile range
    int kt0 = kb0 + wave * kt_per_wave;
    int kt1 = min(min(kt0 + kt_per_wave, (ks + 1) * kt_per_split), KT);
    const int n = max(kt1 - kt0, 0);
    constexpr int U = 8;
    const u32x4_t* wp = Wp + ((size_t)nt * KT + kt0) * 64 + lane;
    // the first weight tiles go out before the prologue: its round trips overlap theirs
    u32x4_t a[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < n) a[u] = __builtin_nontemporal_load(wp + (size_t)u * 64);
    const int kt_base = (PRO == PRO_NORM) ? 0 : kb0;
    if (PRO == PRO_NORM) {
        const int H = KT * 16;
        // Two passes over the row with x' parked (as bf16, exactly what the residual stream stores) in the LDS slot
        // that will hold xn: nothing but the running sum of squares lives in registers across the block-wide sum, so
        // the kernel keeps two 512-thread blocks per CU (a register-resident row cost 200 VGPRs = one block).
        constexpr int G = WAVES / 4;                                   // rows handled at a time (256 threads each)
        const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
        const size_t kstride = (size_t)MTTS_PFCAP * pr.slab_npad;
        const bool writer = nt == 0 && ks == 0 && pr.x_out != nullptr;
        for (int r0 = 0; r0 < pr.rows; r0 += G) {
            const int r = r0 + g;
            const bool act = r < pr.rows;
            float ss = 0.f;
            // the norm weights of this thread's first chunk travel with the row's loads (one round trip, not two)
            const u32x4_t nw0 = (t * 8 < H) ? *(const u32x4_t*)(pr.norm_w + t * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll 1
            for (int i0 = t * 8; i0 < H; i0 += 2048) {
                if (!act) break;
                const u32x4_t xo = *(const u32x4_t*)(pr.x_in + (size_t)r * H + i0);
                float v[8];
                if (pr.ksplit > 0) {
                    const float* p0 = pr.slabs + (size_t)r * pr.slab_npad + i0;
                    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
                    for (int k0 = 0; k0 < pr.ksplit; k0 += 4) {        // slabs summed in the fixed order k = 0, 1, 2, ...
                        float4 ta[4], tb[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float* pk = p0 + (size_t)min(k0 + j, pr.ksplit - 1) * kstride;
                            ta[j] = *(const float4*)pk;
                            tb[j] = *(const float4*)(pk + 4);
                        }
                        if (k0 == 0) { sa = ta[0]; sb = tb[0]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + j < pr.ksplit && k0 + j > 0) {
                                sa.x += ta[j].x; sa.y += ta[j].y; sa.z += ta[j].z; sa.w += ta[j].w;
                                sb.x += tb[j].x; sb.y += tb[j].y; sb.z += tb[j].z; sb.w += tb[j].w;
                            }
                    }
                    v[0] = rbf(bflo(xo.x) + rbf(sa.x)); v[1] = rbf(bfhi(xo.x) + rbf(sa.y));
                    v[2] = rbf(bflo(xo.y) + rbf(sa.z)); v[3] = rbf(bfhi(xo.y) + rbf(sa.w));
                    v[4] = rbf(bflo(xo.z) + rbf(sb.x)); v[5] = rbf(bfhi(xo.z) + rbf(sb.y));
                    v[6] = rbf(bflo(xo.w) + rbf(sb.z)); v[7] = rbf(bfhi(xo.w) + rbf(sb.w));
                } else {                                               // first layer: x is the embedding sum itself
                    v[0] = bflo(xo.x); v[1] = bfhi(xo.x); v[2] = bflo(xo.y); v[3] = bfhi(xo.y);
                    v[4] = bflo(xo.z); v[5] = bfhi(xo.z); v[6] = bflo(xo.w); v[7] = bfhi(xo.w);
                }
                u32x4_t xn;
                xn.x = pack2(v[0], v[1]); xn.y = pack2(v[2], v[3]); xn.z = pack2(v[4], v[5]); xn.w = pack2(v[6], v[7]);
                if (writer) *(u32x4_t*)(pr.x_out + (size_t)r * H + i0) = xn;
                xs[(size_t)(i0 >> 3) * SMALL_RP + r] = xn;
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
            }
            // block_sum_256 of resid_norm_kernel, per group of 256 threads
            ss = wave_sum(ss);
            if ((t & 63) == 0) sh[g][t >> 6] = ss;
            __syncthreads();
            const float tot = sh[g][0] + sh[g][1] + sh[g][2] + sh[g][3];
            __syncthreads();
            const float inv = 1.0f / sqrtf(tot / (float)H + pr.eps);
#pragma unroll 1
            for (int i0 = t * 8; i0 < H; i0 += 2048) {
                if (!act) break;
                const u32x4_t w = (i0 == t * 8) ? nw0 : *(const u32x4_t*)(pr.norm_w + i0);
                const u32x4_t xv = xs[(size_t)(i0 >> 3) * SMALL_RP + r];
                u32x4_t y;
                y.x = pack2(bflo(w.x) * rbf(bflo(xv.x) * inv), bfhi(w.x) * rbf(bfhi(xv.x) * inv));
                y.y = pack2(bflo(w.y) * rbf(bflo(xv.y) * inv), bfhi(w.y) * rbf(bfhi(xv.y) * inv));
                y.z = pack2(bflo(w.z) * rbf(bflo(xv.z) * inv), bfhi(w.z) * rbf(bfhi(xv.z) * inv));
                y.w = pack2(bflo(w.w) * rbf(bflo(xv.w) * inv), bfhi(w.w) * rbf(bfhi(xv.w) * inv));
                xs[(size_t)(i0 >> 3) * SMALL_RP + r] = y;
            }
        }
    } else if (PRO == PRO_COMBINE) {
        // element (r, k = head*128 + d) of this block's K slice: chunks summed in order, as attn_combine_kernel does
        const int k_lo = kb0 * 16, k_hi = min((ks + 1) * kt_per_split, KT) * 16, len = k_hi - k_lo;
        for (int idx = threadIdx.x; idx < pr.rows * len; idx += WAVES * 64) {
            const int r = idx / len, k = k_lo + idx % len;
            const int h = k >> 7, d = k & 127;
            // the first 8 chunk partials are fetched without waiting for the row's chunk count (the slots exist for
            // every row; what lies beyond a row's last chunk is simply not added): one round trip instead of two
            const float* p = pr.opart + ((size_t)r * pr.nq + h) * pr.nchunks_max * MTTS_HD + d;
            float t0[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t0[j] = p[(size_t)min(j, pr.nchunks_max - 1) * MTTS_HD];
            const RowMeta m = pr.meta[r];
            const int npages = m.seq >= 0 ? (m.pos + 1 + MTTS_PAGE - 1) / MTTS_PAGE : 0;
            const int nch = (npages + pr.pages_per_chunk - 1) / pr.pages_per_chunk;
            float sacc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nch) sacc += t0[j];
            for (int c0 = 8; c0 < nch; c0 += 8) {
                float tt[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tt[j] = p[(size_t)min(c0 + j, nch - 1) * MTTS_HD];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < nch) sacc += tt[j];
            }
            ((uint16_t*)xs)[((size_t)((k - k_lo) >> 3) * SMALL_RP + r) * 8 + (k & 7)] = f2bf(sacc);
        }
    } else if (PRO == PRO_ROWS) {
        const int K = KT * 16;
        const int g_lo = kb0 * 2, g_hi = min((ks + 1) * kt_per_split, KT) * 2, len = g_hi - g_lo;     // 16-byte groups
        for (int idx = threadIdx.x; idx < pr.rows * len; idx += WAVES * 64) {
            const int r = idx / len, gi = idx % len;
            xs[(size_t)gi * SMALL_RP + r] = *(const u32x4_t*)(pr.xrows + (size_t)r * K + (size_t)(g_lo + gi) * 8);
        }
    }
    __syncthreads();
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int row = lane & 31;
    const bool has = row < SMALL_RP;
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    // lane's B fragment of k-tile kt: 16-byte group (kt - kt_base)*2 + (lane >> 5), its row
    const u32x4_t* xl = xs + (size_t)((kt0 - kt_base) * 2 + (lane >> 5)) * SMALL_RP + (has ? row : 0);
    // the tiles fetched before the prologue (all of them for the shapes whose waves own <= 8)
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < n) {
            const u32x4_t t = xl[(size_t)u * 2 * SMALL_RP];
            const u32x4_t bw = has ? t : zero;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a[u], *(bf16x8_t*)&bw, acc, 0, 0, 0);
        }
    int i = U;
    for (; i + U <= n; i += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(wp + (size_t)(i + u) * 64);
        u32x4_t b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4_t t = xl[(size_t)(i + u) * 2 * SMALL_RP];
            b[u] = has ? t : zero;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a[u], *(bf16x8_t*)&b[u], acc, 0, 0, 0);
    }
    for (; i < n; ++i) {
        const u32x4_t aw = __builtin_nontemporal_load(wp + (size_t)i * 64);
        const u32x4_t t = xl[(size_t)i * 2 * SMALL_RP];
        const u32x4_t bw = has ? t : zero;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&aw, *(bf16x8_t*)&bw, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const int l2 = lane;
    if ((l2 & 31) >= pr.rows) return;                          // only live rows are stored
    for (int q = wave; q < 4; q += WAVES) {                    // q: register quad 4q..4q+3
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = red[0][4 * q + j][l2];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) s += red[w][4 * q + j][l2];
            v[j] = s;
        }
        const int orow = l2 & 31;
        const int nl = 8 * q + 4 * (l2 >> 5);
        const int n0 = nt * 32 + nl;
        if (EPI == EPI_PARTIAL) {
            *(float4*)(partial + ((size_t)ks * MTTS_PFCAP + orow) * Npad + n0) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (EPI == EPI_BF16) {
            uint16_t* o = out + (size_t)orow * Npad + n0;
            if (n0 + 3 < n_valid) *(u32x2_t*)o = u32x2_t{pack2(v[0], v[1]), pack2(v[2], v[3])};
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n0 + j < n_valid) o[j] = f2bf(v[j]);
            }
        } else {                                               // EPI_SILU_RM: gate,up interleaved rows -> [rows][Npad/2]
            uint32_t w2;
            {
                float g0 = rbf(v[0]), u0 = rbf(v[1]), g1 = rbf(v[2]), u1 = rbf(v[3]);
                float a0 = rbf(g0 / (1.0f + expf(-g0))), a1 = rbf(g1 / (1.0f + expf(-g1)));
                w2 = pack2(a0 * u0, a1 * u1);
            }
            *(uint32_t*)(out + (size_t)orow * (Npad >> 1) + (n0 >> 1)) = w2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Prefill GEMM: Y[R rows, N] = X[R, K] * W[N, K]^T for R up to MTTS_PFCAP rows per pass (compute-bound side of
// the path: 2*R*N*K flops on the matrix cores against one read of W).  Same fragment layouts as the skinny
// kernel, so W and X fragments are contiguous KiB loads straight into MFMA operands (no LDS): a block of
// 4 waves (2 x 2) owns 128 rows x 128 columns, a wave 64 x 64 = 2 x 2 accumulators; the two waves that share a
// W (or X) fragment hit the same lines in L1.  Same epilogues and rounding points as the skinny kernel.
// grid = (ceil(N/128), ceil(R/128), ksplit).
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_tile_kernel(
    const u32x4_t* __restrict__ Wp, const u32x4_t* __restrict__ Xp, int KT, int kt_per_split, int ntiles, int rtiles,
    float* __restrict__ partial, uint16_t* __restrict__ out, int Npad, int n_valid) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nt0 = blockIdx.x * 4 + (wave & 1) * 2;       // first of this wave's two 32-column tiles
    const int rt0 = blockIdx.y * 4 + (wave >> 1) * 2;      // first of its two 32-row tiles
    if (nt0 >= ntiles || rt0 >= rtiles) return;
    const bool n1 = nt0 + 1 < ntiles, r1 = rt0 + 1 < rtiles;
    const int ks = blockIdx.z;
    const int kt0 = ks * kt_per_split, kt1 = min(kt0 + kt_per_split, KT);
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const size_t xtile = (size_t)KT * 64;
    const u32x4_t* w0 = Wp + ((size_t)nt0 * KT + kt0) * 64 + lane;
    const u32x4_t* w1 = Wp + ((size_t)(n1 ? nt0 + 1 : nt0) * KT + kt0) * 64 + lane;
    const u32x4_t* x0 = Xp + (size_t)rt0 * xtile + (size_t)kt0 * 64 + lane;
    const u32x4_t* x1 = Xp + (size_t)(r1 ? rt0 + 1 : rt0) * xtile + (size_t)kt0 * 64 + lane;
    constexpr int U = 8;
    const int n = kt1 - kt0;
    int i = 0;
    for (; i + U <= n; i += U) {
        u32x4_t a0[U], a1[U], b0[U], b1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a0[u] = w0[(size_t)(i + u) * 64];
            a1[u] = w1[(size_t)(i + u) * 64];
            b0[u] = x0[(size_t)(i + u) * 64];
            b1[u] = x1[(size_t)(i + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a0[u], *(bf16x8_t*)&b0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a0[u], *(bf16x8_t*)&b1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a1[u], *(bf16x8_t*)&b0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a1[u], *(bf16x8_t*)&b1[u], acc[1][1], 0, 0, 0);
        }
    }
    for (; i < n; ++i) {
        const u32x4_t a0 = w0[(size_t)i * 64], a1 = w1[(size_t)i * 64], b0 = x0[(size_t)i * 64], b1 = x1[(size_t)i * 64];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a0, *(bf16x8_t*)&b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a0, *(bf16x8_t*)&b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a1, *(bf16x8_t*)&b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&a1, *(bf16x8_t*)&b1, acc[1][1], 0, 0, 0);
    }
    // D[n][row]: lane holds column row = lane&31 and n = (i&3) + 8*(i>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a && !n1) break;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b && !r1) break;
            const int row = (rt0 + b) * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                const int n0 = (nt0 + a) * 32 + 8 * q + 4 * (lane >> 5);
                if (EPI == EPI_PARTIAL) {
                    *(float4*)(partial + ((size_t)ks * MTTS_PFCAP + row) * Npad + n0) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (EPI == EPI_BF16) {
                    uint16_t* o = out + (size_t)row * Npad + n0;
                    if (n0 + 3 < n_valid) *(u32x2_t*)o = u32x2_t{pack2(v[0], v[1]), pack2(v[2], v[3])};
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (n0 + j < n_valid) o[j] = f2bf(v[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float g = rbf(v[2 * j]), u = rbf(v[2 * j + 1]);
                        float s = rbf(g / (1.0f + expf(-g)));
                        out[xpack_off(row, (n0 >> 1) + j, Npad >> 1)] = f2bf(s * u);
                    }
                }
            }
        }
    }
}

// Pack a row-major bf16 matrix [rows][cols] into fragment order inside a packed
// buffer of rows_pad rows (pre-zeroed by the caller).  Source row s lands on packed row
// s*row_mul + row_off: (1,off) places q/k/v or the 7 speech heads one after another,
// (2,0)/(2,1) interleaves gate and up rows for the fused SwiGLU epilogue.
__global__ void pack_weight_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                   int rows, int cols, int rows_pad, int row_mul, int row_off) {
    const int KT = cols >> 4;
    size_t total = (size_t)rows_pad * cols / 8;          // 16-byte groups
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        // g indexes the packed array in units of 8 elements: [nt][kt][lane]
        int lane = (int)(g & 63);
        size_t t = g >> 6;
        int kt = (int)(t % KT);
        int nt = (int)(t / KT);
        int p = nt * 32 + (lane & 31) - row_off;
        int k = kt * 16 + 8 * (lane >> 5);
        if (p < 0 || (p % row_mul) != 0) continue;
        int srow = p / row_mul;
        if (srow >= rows) continue;
        *(u32x4_t*)(dst + g * 8) = *(const u32x4_t*)(src + (size_t)srow * cols + k);
    }
}

// Row-major activations [R][K] bf16 -> X-fragment layout, `tiles` row tiles (rows >= R are zero).
__global__ void pack_rows_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int R, int K, int tiles) {
    int per_tile = (K >> 4) * 64;
    int total = per_tile * tiles;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        int tile = g / per_tile, gi = g % per_tile;
        int lane = gi & 63, kt = gi >> 6;
        int r = tile * 32 + (lane & 31), k = kt * 16 + 8 * (lane >> 5);
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (r < R) v = *(const u32x4_t*)(src + (size_t)r * K + k);
        *(u32x4_t*)(dst + (size_t)g * 8) = v;
    }
}

// Sum split-K partials and round to bf16, row-major output (unit-test epilogue).
__global__ void reduce_partial_bf16_kernel(const float* __restrict__ partial, uint16_t* __restrict__ out,
                                           int ksplit, int Npad, int n_valid, int R) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * n_valid) return;
    int r = idx / n_valid, n = idx % n_valid;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += partial[((size_t)k * MTTS_PFCAP + r) * Npad + n];
    out[idx] = f2bf(s);
}

struct GemmPlan {
    int waves, ksplit, kt_per_split, kt_per_wave;
};

// Choose the decomposition so that the grid is >= ~256 blocks where the shape allows
// and every wave gets >= 4 k-tiles.
static GemmPlan plan_gemm(int Npad, int K, int want_ksplit) {
    GemmPlan p;
    int KT = K / 16, ntiles = Npad / 32;
    int ks = want_ksplit;
    if (ks <= 0) {
        ks = 1;
        while (ntiles * ks < 256 && ks < 8 && KT / (ks * 2) >= 16) ks *= 2;
    }
    p.ksplit = ks;
    p.kt_per_split = (KT + ks - 1) / ks;
    int w = 8;
    while (w > 1 && p.kt_per_split / w < 4) w >>= 1;
    // (very wide GEMMs -- head 0 -- used to run faster with 4 waves; with 8-byte logit stores 8 waves win: 115 vs 126 us)
    p.waves = w;
    p.kt_per_wave = (p.kt_per_split + w - 1) / w;
    return p;
}

template <int EPI, int MB>
static void launch_gemm_epi(const GemmPlan& p, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                            float* partial, uint16_t* out, hipStream_t st) {
    dim3 grid(Npad / 32, p.ksplit);
    int KT = K / 16;
#define MTTS_GEMM_CASE(WV)                                                                            \
    case WV:                                                                                          \
        hipLaunchKernelGGL((gemm_skinny_kernel<WV, EPI, MB>), grid, dim3(WV * 64), 0, st,             \
                           (const u32x4_t*)Wp, (const u32x4_t*)Xp, KT, p.kt_per_split, p.kt_per_wave, \
                           partial, out, Npad, n_valid);                                              \
        break;
    switch (p.waves) {
        MTTS_GEMM_CASE(8)
        MTTS_GEMM_CASE(4)
        MTTS_GEMM_CASE(2)
        MTTS_GEMM_CASE(1)
    }
#undef MTTS_GEMM_CASE
}

template <int EPI>
static void launch_gemm_mb(int mb, const GemmPlan& p, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                           float* partial, uint16_t* out, hipStream_t st) {
    if (mb <= 1) launch_gemm_epi<EPI, 1>(p, Wp, Xp, K, Npad, n_valid, partial, out, st);
    else if (mb == 2) launch_gemm_epi<EPI, 2>(p, Wp, Xp, K, Npad, n_valid, partial, out, st);
    else launch_gemm_epi<EPI, 4>(p, Wp, Xp, K, Npad, n_valid, partial, out, st);     // 3 tiles run as 4 (zero tile)
}

// mb = number of 32-row activation tiles (1..4) that share the weight stream
void launch_gemm(int epi, int mb, const GemmPlan& p, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                 float* partial, uint16_t* out, hipStream_t st) {
    if (epi == EPI_PARTIAL) launch_gemm_mb<EPI_PARTIAL>(mb, p, Wp, Xp, K, Npad, n_valid, partial, out, st);
    else if (epi == EPI_BF16) launch_gemm_mb<EPI_BF16>(mb, p, Wp, Xp, K, Npad, n_valid, partial, out, st);
    else launch_gemm_mb<EPI_SILU>(mb, p, Wp, Xp, K, Npad, n_valid, partial, out, st);
}


// Small-batch launch.  The plan's waves are raised to >= 4 (the norm prologue works in groups of 256 threads).
// Returns the dynamic LDS bytes it needs, or -1 when the shape does not fit (the caller then takes the general path).
int mtts_small_lds_bytes(const GemmPlan& p, int K, int pro) {
    const int kts = pro == PRO_NORM ? K / 16 : p.kt_per_split;
    return kts * 2 * SMALL_RP * 16;
}
template <int EPI, int PRO>
static void launch_small_epi(const GemmPlan& p0, const void* Wp, int K, int Npad, int n_valid, float* partial, uint16_t* out,
                             const SmallPro& pr, hipStream_t st) {
    GemmPlan p = p0;
    if (p.waves < 4) { p.waves = 4; p.kt_per_wave = (p.kt_per_split + 3) / 4; }
    dim3 grid(Npad / 32, p.ksplit);
    const int KT = K / 16;
    const size_t lds = (size_t)mtts_small_lds_bytes(p, K, PRO);
    if (p.waves == 8)
        hipLaunchKernelGGL((gemv_small_kernel<8, EPI, PRO>), grid, dim3(512), lds, st, (const u32x4_t*)Wp, KT, p.kt_per_split,
                           p.kt_per_wave, partial, out, Npad, n_valid, pr);
    else
        hipLaunchKernelGGL((gemv_small_kernel<4, EPI, PRO>), grid, dim3(256), lds, st, (const u32x4_t*)Wp, KT, p.kt_per_split,
                           p.kt_per_wave, partial, out, Npad, n_valid, pr);
}
// the combinations the decode step uses
void launch_gemv_small(int epi, int pro, const GemmPlan& p, const void* Wp, int K, int Npad, int n_valid, float* partial,
                       uint16_t* out, const SmallPro& pr, hipStream_t st) {
    if (epi == EPI_PARTIAL && pro == PRO_NORM) launch_small_epi<EPI_PARTIAL, PRO_NORM>(p, Wp, K, Npad, n_valid, partial, out, pr, st);
    else if (epi == EPI_PARTIAL && pro == PRO_COMBINE) launch_small_epi<EPI_PARTIAL, PRO_COMBINE>(p, Wp, K, Npad, n_valid, partial, out, pr, st);
    else if (epi == EPI_PARTIAL && pro == PRO_ROWS) launch_small_epi<EPI_PARTIAL, PRO_ROWS>(p, Wp, K, Npad, n_valid, partial, out, pr, st);
    else if (epi == EPI_SILU_RM && pro == PRO_NORM) launch_small_epi<EPI_SILU_RM, PRO_NORM>(p, Wp, K, Npad, n_valid, partial, out, pr, st);
    else if (epi == EPI_BF16 && pro == PRO_NORM) launch_small_epi<EPI_BF16, PRO_NORM>(p, Wp, K, Npad, n_valid, partial, out, pr, st);
}

// Tiled launch for R > 128 rows (prefill passes).  ksplit only where the grid would leave most CUs idle.
int mtts_tile_ksplit(int Npad, int K, int R) {
    int blocks = ((Npad + 127) / 128) * ((R + 127) / 128), ks = 1;
    while (blocks * ks < 512 && ks < 8 && (K / 16) / (ks * 2) >= 16) ks *= 2;
    return ks;
}
void launch_gemm_tile(int epi, int R, int ksplit, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                      float* partial, uint16_t* out, hipStream_t st) {
    const int KT = K / 16, ntiles = Npad / 32, rtiles = (R + 31) / 32;
    const int kps = (KT + ksplit - 1) / ksplit;
    dim3 grid((ntiles + 3) / 4, (rtiles + 3) / 4, ksplit);
#define MTTS_TILE_CASE(E)                                                                                     \
    hipLaunchKernelGGL((gemm_tile_kernel<E>), grid, dim3(256), 0, st, (const u32x4_t*)Wp, (const u32x4_t*)Xp, \
                       KT, kps, ntiles, rtiles, partial, out, Npad, n_valid)
    if (epi == EPI_PARTIAL) MTTS_TILE_CASE(EPI_PARTIAL);
    else if (epi == EPI_BF16) MTTS_TILE_CASE(EPI_BF16);
    else MTTS_TILE_CASE(EPI_SILU);
#undef MTTS_TILE_CASE
}

GemmPlan mtts_plan_gemm(int Npad, int K, int want_ksplit) { return plan_gemm(Npad, K, want_ksplit); }
GemmPlan mtts_plan_gemm_forced(int Npad, int K, int ksplit, int waves) {
    GemmPlan p;
    int KT = K / 16;
    p.ksplit = ksplit;
    p.kt_per_split = (KT + ksplit - 1) / ksplit;
    p.waves = waves;
    p.kt_per_wave = (p.kt_per_split + waves - 1) / waves;
    return p;
}

void launch_pack_weight(const void* src, void* dst, int rows, int cols, int rows_pad, int row_mul, int row_off, hipStream_t st) {
    size_t total = (size_t)rows_pad * cols / 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, rows,
                       cols, rows_pad, row_mul, row_off);
}
void launch_pack_rows(const void* src, void* dst, int R, int K, int tiles, hipStream_t st) {
    int total = (K / 16) * 64 * tiles;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const uint16_t*)src,
                       (uint16_t*)dst, R, K, tiles);
}
void launch_reduce_partial_bf16(const float* partial, void* out, int ksplit, int Npad, int n_valid, int R, hipStream_t st) {
    int total = R * n_valid;
    hipLaunchKernelGGL(reduce_partial_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, partial,
                       (uint16_t*)out, ksplit, Npad, n_valid, R);
}
