// Paged-KV attention for single-token queries (decode rows and prefill rows alike).
//
// Replaces eager_attention_forward (transformers modeling_qwen3.py:185-208) with
// GQA (repeat_kv :173) over the paged cache.  The reference's eager path rounds
//   s = bf16(bf16(q.k) * scaling);  p = bf16(softmax_fp32(s));  o = bf16(p @ V)
// so the kernel keeps those three rounding points: the probabilities need the
// row-wide max and sum BEFORE they are rounded, hence two streaming passes:
//   pass A (attn_scores): K stream -> bf16 scores + per-page (max, sumexp)
//   pass B (attn_pv):     V stream x rounded probabilities -> fp32 partial O
//   combine:              sum of partial O over chunks -> bf16, X-fragment layout
// K and V are each read exactly once; scores are 2 B per (head, token) of scratch.
//
// HBM bound.  Algorithmic bytes per (row, layer) = len * nkv * 128 * 2 B * 2.  Since round 3 the decode passes read
// every COMPLETE page in a lossless 13-bit form ("sealed pages", below): 13/16 of those bytes, the same bits out.
// Two rules shape the decode kernels: a wave's loads return IN ORDER, so whatever is small and needed first (q, the
// softmax statistics, a page's scores, the fused epilogue's operands, a sealed page's dictionary) is requested before
// what is large; and nothing on the way may wait for "all loads" -- which the compiler's wait counts do after a JOIN of
// code paths with different numbers of loads in flight: hence one straight-line body per page form.
#include "common.h"
#include <cstdlib>
// (tuning aid, tools/ab_build.sh: waves per SIMD the sealed-page kernels are compiled for; 0 = the compiler's choice --
//  5 was measured and lost, profiles/r03_kv_pack_ab.txt item 4)
#ifndef MTTS_PK_WAVES
#define MTTS_PK_WAVES 0
#endif
#if MTTS_PK_WAVES
#define PK_OCC(PK) __attribute__((amdgpu_waves_per_eu((PK) ? MTTS_PK_WAVES : 1, (PK) ? MTTS_PK_WAVES : 8)))
#else
#define PK_OCC(PK)
#endif

// ---------------------------------------------------------------------------------------------------
// Sealed pages (common.h: MTTS_PKU).  A bf16 value is [sign | exponent 8 | mantissa 7]; over the 128 values one lane
// reads from a page (a K row; 32 tokens x 4 dims of V) the LOW byte (exponent bit 0 + mantissa) is incompressible, but
// the high byte without its sign (exponent bits 7..1) takes only a handful of distinct values.  A sealed page stores the
// low bytes, a 4-bit code per value (sign, index) and the lane's dictionary of up to 8 such bytes: 13 bits per value,
// exact.  Unpacking one 16-byte unit (8 values) is 12 VALU instructions: the dictionary look-up is a v_perm_b32 whose
// selector is the masked code word, the sign is and-ed back in, two more v_perm interleave high and low bytes.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t u4c(const u32x4_t& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
// unit j (0..15, compile-time) of the lane's page from its 13 packed registers -> the 4 dwords of the bf16 form
__device__ __forceinline__ void pk_unit(const u32x4_t* pk, int j, uint32_t w[4]) {
    const uint32_t L0 = u4c(pk[j >> 1], 2 * (j & 1)), L1 = u4c(pk[j >> 1], 2 * (j & 1) + 1);
    const uint32_t N = u4c(pk[8 + (j >> 2)], j & 3);
    const uint32_t lo = pk[12].x, hi = pk[12].y;
    const uint32_t HA = ((N & 0x08080808u) << 4) | __builtin_amdgcn_perm(hi, lo, N & 0x07070707u);
    const uint32_t HB = (N & 0x80808080u) | __builtin_amdgcn_perm(hi, lo, (N >> 4) & 0x07070707u);
    w[0] = __builtin_amdgcn_perm(HA, L0, 0x05010400u);
    w[1] = __builtin_amdgcn_perm(HA, L0, 0x07030602u);
    w[2] = __builtin_amdgcn_perm(HB, L1, 0x05010400u);
    w[3] = __builtin_amdgcn_perm(HB, L1, 0x07030602u);
}

// i-th unit to request so that unit j's operands (low (j >> 1), nibbles 8 + (j >> 2), dictionary 12) are complete as
// early as possible: 12, 8, 0, 1, 9, 2, 3, 10, 4, 5, 11, 6, 7
__host__ __device__ constexpr int pk_order(int i) { return i == 0 ? 12 : (i - 1) % 3 == 0 ? 8 + (i - 1) / 3 : 2 * ((i - 1) / 3) + (i - 1) % 3 - 1; }
// One lane's share of a page (16 units of 16 B in registers) -> its sealed form (13 units); `spare` = third dword of unit 12.
__device__ bool seal_encode(const u32x4_t (&v)[16], u32x4_t* __restrict__ pk, uint32_t spare, bool unfit) {
    unsigned long long b0 = 0ull, b1 = 0ull;                  // which of the 128 possible high bytes occur
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t e = (u4c(v[j], c) >> (8 + 16 * h)) & 0x7fu;
                if (e & 64u) b1 |= 1ull << (e & 63u);
                else b0 |= 1ull << e;
            }
    const int n0 = __popcll(b0), n = n0 + __popcll(b1);
    if (n > 8 || unfit) {                                     // does not fit: readers take the bf16 page
        pk[12 * 64] = u32x4_t{0u, 0u, spare, 1u};
        return false;
    }
    unsigned long long dict = 0ull;                           // entry i = the i-th smallest high byte present
    {
        unsigned long long t0 = b0, t1 = b1;
        for (int i = 0; i < n; ++i) {
            uint32_t e;
            if (t0) { e = __ffsll((long long)t0) - 1; t0 &= t0 - 1; }
            else { e = 64 + __ffsll((long long)t1) - 1; t1 &= t1 - 1; }
            dict |= (unsigned long long)e << (8 * i);
        }
    }
    auto code = [&](uint32_t half) -> uint32_t {              // 16-bit value -> sign << 3 | rank of its high byte
        const uint32_t e = (half >> 8) & 0x7fu;
        const uint32_t rank = (e & 64u) ? n0 + __popcll(b1 & ((1ull << (e & 63u)) - 1ull)) : __popcll(b0 & ((1ull << e) - 1ull));
        return ((half >> 12) & 8u) | rank;
    };
    uint32_t lowp[32], nib[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t N = 0u;
#pragma unroll
        for (int g = 0; g < 2; ++g) {                         // group A = dwords 0,1 of the unit, B = dwords 2,3
            uint32_t L = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t half = (u4c(v[j], 2 * g + (k >> 1)) >> (16 * (k & 1))) & 0xffffu;
                L |= (half & 0xffu) << (8 * k);
                N |= code(half) << (8 * k + 4 * g);
            }
            lowp[2 * j + g] = L;
        }
        nib[j] = N;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) pk[u * 64] = u32x4_t{lowp[4 * u], lowp[4 * u + 1], lowp[4 * u + 2], lowp[4 * u + 3]};
#pragma unroll
    for (int u = 0; u < 4; ++u) pk[(8 + u) * 64] = u32x4_t{nib[4 * u], nib[4 * u + 1], nib[4 * u + 2], nib[4 * u + 3]};
    pk[12 * 64] = u32x4_t{(uint32_t)dict, (uint32_t)(dict >> 32), spare, 0u};
    return true;
}
// The format hook's plain form: the lane's values as they are.
__device__ bool seal_lane(const u32x4_t* __restrict__ raw, u32x4_t* __restrict__ pk) {
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = raw[j * 64];
    return seal_encode(v, pk, 0u, false);
}
// K pages: a lane is a token's 128 dims, and what spreads their magnitudes is the dim (k_norm's weight, the RoPE
// frequency), not the token.  So each dim is first divided by a power of two chosen from the page itself (the mean
// exponent of that dim over the 64 tokens goes to 125), the shifts are kept in the page (lane l: dims 2l, 2l+1, in the
// spare dword of unit 12) and the READER multiplies q by the same powers of two: k 2^-s . q 2^s is the same fp32 product,
// bit for bit, as long as every rescaled exponent stays normal -- a lane (or a q) where it would not is flagged and
// the page is read as bf16.  `lds` = 8 KiB + 128 B of this wave's own.
__device__ bool seal_lane_k(const u32x4_t* __restrict__ raw, u32x4_t* __restrict__ pk, uint8_t* __restrict__ lds, int lane) {
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = raw[j * 64];
    __builtin_amdgcn_wave_barrier();                          // (a previous page's reads of `lds` are done: same wave, in order)
    uint32_t* row = (uint32_t*)(lds + lane * 128);            // this token's 128 exponents, dim order
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            const uint32_t a = u4c(v[j], c), b = u4c(v[j], c + 1);
            row[2 * j + (c >> 1)] = ((a >> 7) & 0xffu) | (((a >> 23) & 0xffu) << 8) | (((b >> 7) & 0xffu) << 16) | (((b >> 23) & 0xffu) << 24);
        }
    __builtin_amdgcn_wave_barrier();
    uint32_t a0 = 0u, a1 = 0u, n0 = 0u, n1 = 0u;              // dims 2 lane, 2 lane + 1: sum / count of the tokens' non-zero exponent fields
    for (int t = 0; t < 64; ++t) {
        const uint32_t e = *(const uint16_t*)(lds + t * 128 + 2 * lane);
        a0 += e & 0xffu; n0 += (e & 0xffu) ? 1u : 0u;
        a1 += e >> 8; n1 += (e >> 8) ? 1u : 0u;
    }
    // the dim's MEAN exponent goes to 125 (the mean, not the maximum: one outlying token -- the first of a dialogue --
    // must not push the dim's other 63 values away from the other dims')
    const int s0 = n0 ? min(max((int)((a0 + n0 / 2) / n0) - 125, -127), 127) : 0;
    const int s1 = n1 ? min(max((int)((a1 + n1 / 2) / n1) - 125, -127), 127) : 0;
    int8_t* sv = (int8_t*)(lds + 8192);
    sv[2 * lane] = (int8_t)s0;
    sv[2 * lane + 1] = (int8_t)s1;
    __builtin_amdgcn_wave_barrier();
    bool unfit = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t sh = *(const uint16_t*)(sv + 8 * j + 2 * c);
            uint32_t out = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t b = (w[c] >> (16 * h)) & 0xffffu;
                const int sft = (int)(int8_t)((sh >> (8 * h)) & 0xffu);
                const int e = (int)((b >> 7) & 0xffu);
                if (e == 0) {
                    if ((b & 0x7fu) && sft) unfit = true;     // a denormal cannot be rescaled exactly
                } else {
                    const int e2 = e - sft;
                    if (e == 255 ? sft != 0 : (e2 < 1 || e2 > 254)) unfit = true;
                    else b = (b - ((uint32_t)sft << 7)) & 0xffffu;
                }
                out |= b << (16 * h);
            }
            w[c] = out;
        }
        v[j] = u32x4_t{w[0], w[1], w[2], w[3]};
    }
    return seal_encode(v, pk, ((uint32_t)s0 & 0xffu) | (((uint32_t)s1 & 0xffu) << 8), unfit);
}
// V pages: a lane is 4 neighbouring dims of the 32 tokens of one parity (lane = 32 * sub + dl: token pair 2 it + sub in
// unit it, dims 4 dl .. 4 dl + 3, halves = the pair's two tokens), and what spreads its magnitudes most is the TOKEN (V has
// no norm: a dialogue's first tokens, sinks, loud and quiet frames).  So each token is first divided by a power of two
// taken from the page: s[t] = (rounded mean exponent of token t's 128 values) - (the smallest such mean in the page),
// rounded down to even, >= 0, and the reader multiplies the token's PROBABILITY by 2^s[t] before the dot products: p 2^s . v 2^-s is the same
// fp32 product bit for bit (p <= 1 and s >= 0: p 2^s cannot underflow; a denormal p with s != 0, or a value whose
// rescaled exponent would leave the normal range, sends the page to its bf16 form).  Lane t keeps s[t] in the low byte
// of its spare dword (the reader's lane t is the one that computes token t's probability).  `lds` = 8 KiB + 320 B.
__device__ bool seal_lane_v(const u32x4_t* __restrict__ raw, u32x4_t* __restrict__ pk, uint8_t* __restrict__ lds, int lane) {
    u32x4_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = raw[j * 64];
    const int sub = lane >> 5;
    __builtin_amdgcn_wave_barrier();
    uint32_t* row = (uint32_t*)(lds + lane * 128);            // entry 2 it + h: this lane's 4 dims of token 4 it + 2 sub + h
#pragma unroll
    for (int it = 0; it < 16; ++it)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t sum = 0u, cnt = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t e = (u4c(v[it], c) >> (7 + 16 * h)) & 0xffu;
                sum += e;
                cnt += e ? 1u : 0u;
            }
            row[2 * it + h] = sum | (cnt << 16);
        }
    __builtin_amdgcn_wave_barrier();
    int mean;
    {                                                         // lane t gathers token t from the 32 lanes of its parity
        const int t = lane, st = (t >> 1) & 1, j = 2 * (t >> 2) + (t & 1);
        uint32_t tot = 0u, cn = 0u;
        for (int d = 0; d < 32; ++d) {
            const uint32_t w = ((const uint32_t*)(lds + (st * 32 + d) * 128))[j];
            tot += w & 0xffffu;
            cn += w >> 16;
        }
        mean = cn ? (int)((tot + cn / 2) / cn) : 0;          // 0: an all-zero token
    }
    int* means = (int*)(lds + 8192);
    int8_t* sv = (int8_t*)(lds + 8192 + 256);
    means[lane] = mean;
    __builtin_amdgcn_wave_barrier();
    int ref = 255;
    for (int t = 0; t < 64; ++t) { const int m = means[t]; if (m > 0) ref = min(ref, m); }
    const int s_own = mean > 0 ? min((mean - ref) & ~1, 126) : 0;     // even: the dictionary's entries are PAIRS of binades
    sv[lane] = (int8_t)s_own;
    __builtin_amdgcn_wave_barrier();
    bool unfit = false;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int s0 = sv[4 * it + 2 * sub], s1 = sv[4 * it + 2 * sub + 1];
        uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t out = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t b = (w[c] >> (16 * h)) & 0xffffu;
                const int sft = h ? s1 : s0;
                const int e = (int)((b >> 7) & 0xffu);
                if (e == 0) {
                    if ((b & 0x7fu) && sft) unfit = true;
                } else {
                    const int e2 = e - sft;
                    if (e == 255 ? sft != 0 : e2 < 1) unfit = true;
                    else b = (b - ((uint32_t)sft << 7)) & 0xffffu;
                }
                out |= b << (16 * h);
            }
            w[c] = out;
        }
        v[it] = u32x4_t{w[0], w[1], w[2], w[3]};
    }
    return seal_encode(v, pk, (uint32_t)s_own & 0xffu, unfit);
}
// counters per layer: {K pages sealed, K pages with a lane that did not fit, the same for V} (the engine's read policy)
__device__ __forceinline__ void seal_count(bool fit, unsigned long long* __restrict__ cnt, int layer, int wave, int lane) {
    const bool all = !__any(!fit);
    if (cnt && lane == 0) atomicAdd(cnt + layer * 4 + wave * 2 + (all ? 0 : 1), 1ull);
}

// Seal the pages that the rows of a forward pass have just completed ((pos & 63) == 63), for every layer and kv head, at
// the end of the pass (decode step -- it is in the step's graph -- or prefill pass alike).  A block looks at 64 rows and
// seals what it finds, the K wave and the V wave side by side: grid = (ceil(R / 64), nkv, L), block 128.  (One block
// per ROW was the first form: 7 168 blocks of 300 registers and 8 KiB of LDS that look at their row and leave cost
// 15 us in every decode step; 224 blocks cost 2-3, and a batch whose rows all complete a page in the same step pays
// 32 pages in a row once in 64 steps.)
__global__ __launch_bounds__(128) void kv_seal_scan_kernel(const u32x4_t* __restrict__ kcache, const u32x4_t* __restrict__ vcache,
                                                           u32x4_t* __restrict__ kpack, u32x4_t* __restrict__ vpack,
                                                           const int32_t* __restrict__ page_table, const RowMeta* __restrict__ meta, int R,
                                                           int max_pages, int total_pages, size_t raw_layer, size_t pk_layer,
                                                           unsigned long long* __restrict__ cnt) {
    __shared__ __attribute__((aligned(16))) uint8_t klds[8192 + 128];
    __shared__ __attribute__((aligned(16))) uint8_t vlds[8192 + 320];
    const int kvh = blockIdx.y, layer = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 64 + lane;
    RowMeta m{-1, 0, 0, 0};
    if (r < R) m = meta[r];
    unsigned long long todo = __ballot(m.seq >= 0 && (m.pos & 63) == 63);
    while (todo) {
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int seq = __shfl(m.seq, i, 64), pos = __shfl(m.pos, i, 64);
        const int page = page_table[(size_t)seq * max_pages + (pos >> 6)];
        const size_t pi = (size_t)kvh * total_pages + page;
        const bool fit = wave ? seal_lane_v(vcache + layer * raw_layer + pi * (MTTS_PAGE * MTTS_HD / 8) + lane,
                                            vpack + layer * pk_layer + pi * (MTTS_PKU * 64) + lane, vlds, lane)
                              : seal_lane_k(kcache + layer * raw_layer + pi * (MTTS_PAGE * MTTS_HD / 8) + lane,
                                            kpack + layer * pk_layer + pi * (MTTS_PKU * 64) + lane, klds, lane);
        seal_count(fit, cnt, layer, wave, lane);
    }
}
// Every physical page (measurement hook: after the caches were filled behind the engine's back).  grid = (pages, nkv, L)
__global__ __launch_bounds__(128) void kv_seal_all_kernel(const u32x4_t* __restrict__ kcache, const u32x4_t* __restrict__ vcache,
                                                          u32x4_t* __restrict__ kpack, u32x4_t* __restrict__ vpack, int total_pages,
                                                          size_t raw_layer, size_t pk_layer, unsigned long long* __restrict__ cnt) {
    __shared__ __attribute__((aligned(16))) uint8_t klds[8192 + 128];
    __shared__ __attribute__((aligned(16))) uint8_t vlds[8192 + 320];
    const int kvh = blockIdx.y, layer = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t pi = (size_t)kvh * total_pages + blockIdx.x;
    const bool fit = wave ? seal_lane_v(vcache + layer * raw_layer + pi * (MTTS_PAGE * MTTS_HD / 8) + lane,
                                        vpack + layer * pk_layer + pi * (MTTS_PKU * 64) + lane, vlds, lane)
                          : seal_lane_k(kcache + layer * raw_layer + pi * (MTTS_PAGE * MTTS_HD / 8) + lane,
                                        kpack + layer * pk_layer + pi * (MTTS_PKU * 64) + lane, klds, lane);
    seal_count(fit, cnt, layer, wave, lane);
}
// Test hook: `npages` pages in a row (bf16 form, 16 KiB each) -> their sealed forms (13 KiB each).  grid = pages, block 64
__global__ __launch_bounds__(64) void kv_seal_pages_kernel(const u32x4_t* __restrict__ raw, u32x4_t* __restrict__ pk, int as_k) {
    __shared__ __attribute__((aligned(16))) uint8_t klds[8192 + 320];
    const u32x4_t* src = raw + (size_t)blockIdx.x * (MTTS_PAGE * MTTS_HD / 8) + threadIdx.x;
    u32x4_t* dst = pk + (size_t)blockIdx.x * (MTTS_PKU * 64) + threadIdx.x;
    if (as_k == 1) seal_lane_k(src, dst, klds, threadIdx.x);
    else if (as_k == 2) seal_lane_v(src, dst, klds, threadIdx.x);
    else seal_lane(src, dst);
}
void launch_kv_seal_pages(const void* raw, void* pk, int npages, int as_k, hipStream_t st) {
    hipLaunchKernelGGL(kv_seal_pages_kernel, dim3(npages), dim3(64), 0, st, (const u32x4_t*)raw, (u32x4_t*)pk, as_k);
}
// Debug hook: how many of the complete pages of the live sequences are sealed / had a lane that did not fit.
// grid = (slots, nkv, L), block 64; out = {K pages, K pages not sealed, V pages, V pages not sealed}
__global__ __launch_bounds__(64) void kv_pack_count_kernel(const u32x4_t* __restrict__ kpack, const u32x4_t* __restrict__ vpack,
                                                           const int32_t* __restrict__ page_table, const int32_t* __restrict__ complete,
                                                           int max_pages, int total_pages, size_t pk_layer, unsigned long long* __restrict__ out) {
    const int b = blockIdx.x, kvh = blockIdx.y, layer = blockIdx.z, lane = threadIdx.x;
    unsigned long long n = 0, dk = 0, dv = 0;
    for (int pg = 0; pg < complete[b]; ++pg) {
        const size_t pi = (size_t)kvh * total_pages + page_table[(size_t)b * max_pages + pg];
        const u32x4_t fk = kpack[layer * pk_layer + pi * (MTTS_PKU * 64) + 12 * 64 + lane];
        const u32x4_t fv = vpack[layer * pk_layer + pi * (MTTS_PKU * 64) + 12 * 64 + lane];
        n += 1;
        dk += __any(fk.w != 0u) ? 1 : 0;
        dv += __any(fv.w != 0u) ? 1 : 0;
    }
    if (lane == 0 && n) {
        atomicAdd(out + 0, n); atomicAdd(out + 1, dk); atomicAdd(out + 2, n); atomicAdd(out + 3, dv);
    }
}
void launch_kv_pack_count(const void* kpack, const void* vpack, const int32_t* page_table, const int32_t* complete, int B, int max_pages,
                          int total_pages, int nkv, int L, unsigned long long* out, hipStream_t st) {
    const size_t pk_layer = (size_t)total_pages * nkv * (MTTS_PKU * 64);
    hipLaunchKernelGGL(kv_pack_count_kernel, dim3(B, nkv, L), dim3(64), 0, st, (const u32x4_t*)kpack, (const u32x4_t*)vpack, page_table,
                       complete, max_pages, total_pages, pk_layer, out);
}
void launch_kv_seal_rows(const void* kcache, const void* vcache, void* kpack, void* vpack, const int32_t* page_table,
                         const RowMeta* meta, int R, int max_pages, int total_pages, int nkv, int L, unsigned long long* cnt, hipStream_t st) {
    const size_t raw_layer = (size_t)total_pages * nkv * (MTTS_PAGE * MTTS_HD / 8), pk_layer = (size_t)total_pages * nkv * (MTTS_PKU * 64);
    hipLaunchKernelGGL(kv_seal_scan_kernel, dim3((R + 63) / 64, nkv, L), dim3(128), 0, st, (const u32x4_t*)kcache, (const u32x4_t*)vcache,
                       (u32x4_t*)kpack, (u32x4_t*)vpack, page_table, meta, R, max_pages, total_pages, raw_layer, pk_layer, cnt);
}
void launch_kv_seal_all(const void* kcache, const void* vcache, void* kpack, void* vpack, int total_pages, int nkv, int L,
                        unsigned long long* cnt, hipStream_t st) {
    const size_t raw_layer = (size_t)total_pages * nkv * (MTTS_PAGE * MTTS_HD / 8), pk_layer = (size_t)total_pages * nkv * (MTTS_PKU * 64);
    hipLaunchKernelGGL(kv_seal_all_kernel, dim3(total_pages, nkv, L), dim3(128), 0, st, (const u32x4_t*)kcache, (const u32x4_t*)vcache,
                       (u32x4_t*)kpack, (u32x4_t*)vpack, total_pages, raw_layer, pk_layer, cnt);
}


// grid = (ceil(pages/4), nkv, R); block 256 = 4 waves, one page (64 tokens) per wave,
// one token per lane: K page is [d/8][token][8] so lane t's 16 loads are 16 B each
// and every wave-instruction covers one contiguous KiB.
// Sum of the qkv GEMM's split-K slabs for head column `col` of row r: lane l gets d = l and d = l + 64 (all loads
// in flight at once, fixed summation order), rounded to bf16 like the Linear's output.
__device__ __forceinline__ void fuse_reduce(const QkvFuse& f, int r, int col, int lane, float& a, float& b) {
    const size_t kstride = (size_t)MTTS_PFCAP * f.Npad;
    const float* p0 = f.partial + (size_t)r * f.Npad + col + lane;
    a = 0.f; b = 0.f;
    for (int k0 = 0; k0 < f.ksplit; k0 += 8) {
        float ta[8], tb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* pk = p0 + (size_t)min(k0 + j, f.ksplit - 1) * kstride;
            ta[j] = pk[0];
            tb[j] = pk[64];
        }
        if (k0 == 0) { a = ta[0]; b = tb[0]; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < f.ksplit && k0 + j > 0) { a += ta[j]; b += tb[j]; }
    }
    a = rbf(a);
    b = rbf(b);
}

// The same in two steps for up to 8 slabs (the usual split): request now, add later -- so that the requests can go out
// BEFORE a page's loads (loads return in order) and the sums run while the page is in flight.
__device__ __forceinline__ void fuse_reduce_request(const QkvFuse& f, int r, int col, int lane, float (&ta)[8], float (&tb)[8]) {
    const size_t kstride = (size_t)MTTS_PFCAP * f.Npad;
    const float* p0 = f.partial + (size_t)r * f.Npad + col + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* pk = p0 + (size_t)min(j, f.ksplit - 1) * kstride;
        ta[j] = pk[0];
        tb[j] = pk[64];
    }
}
__device__ __forceinline__ void fuse_reduce_sum(const QkvFuse& f, const float (&ta)[8], const float (&tb)[8], float& a, float& b) {
    a = ta[0]; b = tb[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
        if (j < f.ksplit) { a += ta[j]; b += tb[j]; }
    a = rbf(a);
    b = rbf(b);
}

// per-head RMSNorm + RoPE of one q or k head held as (a = x[l], b = x[l+64]) across a wave (qkv_post_kernel's math).
// The norm weights and the RoPE row are loaded before the slabs are waited for (one memory round trip, not two).
struct FuseVec { float w0, w1, c, s; };
__device__ __forceinline__ FuseVec fuse_load_vec(const QkvFuse& f, const uint16_t* __restrict__ nw, int pos, int lane) {
    FuseVec v;
    v.w0 = bf2f(nw[lane]);
    v.w1 = bf2f(nw[lane + 64]);
    v.c = bf2f(f.rope_cos[(size_t)pos * 64 + lane]);
    v.s = bf2f(f.rope_sin[(size_t)pos * 64 + lane]);
    return v;
}
__device__ __forceinline__ void fuse_norm_rope(const QkvFuse& f, const FuseVec& v, float a, float b, float& o1, float& o2) {
    const float ss = wave_sum(a * a + b * b);
    const float inv = 1.0f / sqrtf(ss / (float)MTTS_HD + f.eps);
    a = rbf(v.w0 * rbf(a * inv));
    b = rbf(v.w1 * rbf(b * inv));
    o1 = rbf(rbf(a * v.c) + rbf(-b * v.s));
    o2 = rbf(rbf(b * v.c) + rbf(a * v.s));
}

// FUSED (decode rows): q comes from the qkv GEMM's slabs (reduce, RMSNorm, RoPE done here, one head per wave), and the
// block whose pages hold position `pos` also produces the new K row, writes it to the cache and uses it from LDS.
// PK: pages before the one that receives this step's token are complete, hence sealed: 13 loads per lane instead of 16.
// The kernel body exists twice, once per page form, each straight-line from its loads to its dot products: with ONE
// body behind a join of "13 or 16 loads in flight" the compiler's wait counts collapse to "wait for everything", and the
// point of the load order (q, then the page in the order the dot products consume it) is that the first products start
// while the last units are still in flight.
template <int G, bool FUSED, bool PK>
__global__ __launch_bounds__(256) PK_OCC(PK) void attn_scores_kernel(
    const uint16_t* __restrict__ qbuf, u32x4_t* __restrict__ kcache, const int32_t* __restrict__ page_table,
    const RowMeta* __restrict__ meta, uint16_t* __restrict__ scores, float* __restrict__ stats, int max_pages,
    int total_pages, int nq, int nkv, float scale, QkvFuse f, const u32x4_t* __restrict__ kpack) {
    __shared__ __attribute__((aligned(16))) uint32_t qs[G][MTTS_HD / 2];   // bf16 pairs, as stored
    __shared__ __attribute__((aligned(16))) uint16_t knew[MTTS_HD];
    __shared__ __attribute__((aligned(16))) uint32_t qw[PK ? 4 : 1][G][MTTS_HD / 2];   // q times the sealed page's per-dim powers of two
    const int r = blockIdx.z, kvh = blockIdx.y;
    const RowMeta m = meta[r];
    if (m.seq < 0) return;
    const int len = m.pos + 1;
    const int npages = (len + MTTS_PAGE - 1) / MTTS_PAGE;
    if ((int)blockIdx.x * 4 >= npages) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pg = blockIdx.x * 4 + wave;
    const int own_pg = m.pos >> 6;                    // page that receives this step's token
    const int Lmax = max_pages * MTTS_PAGE;
    // q first (separate q/k/v epilogue): loads return in order, so behind the page q would arrive last
    uint32_t qreg[(G * MTTS_HD / 2 + 255) / 256];
    if (!FUSED) {
#pragma unroll
        for (int i = 0; i < (G * MTTS_HD / 2 + 255) / 256; ++i) {
            const int idx = threadIdx.x + 256 * i;
            qreg[i] = idx < G * MTTS_HD / 2 ? ((const uint32_t*)qbuf)[((size_t)r * nq + kvh * G) * (MTTS_HD / 2) + idx] : 0u;
        }
    }
    // fused epilogue: this wave's head (q head `wave`, or the new K row in the block that owns the step's page) has its
    // split-K slabs, norm weights and RoPE row requested here, before the page, for the same reason
    const bool own = FUSED && (own_pg >> 2) == (int)blockIdx.x;
    const bool pre = FUSED && f.ksplit <= 8 && wave < G + (own ? 1 : 0);
    float pta[8], ptb[8];
    FuseVec pfv{0.f, 0.f, 0.f, 0.f};
    if (pre) {
        const bool isk = wave == G;
        pfv = fuse_load_vec(f, isk ? f.knorm_w : f.qnorm_w, m.pos, lane);
        fuse_reduce_request(f, r, (isk ? nq + kvh : kvh * G + wave) * MTTS_HD, lane, pta, ptb);
    }
    const int page = pg < npages ? page_table[(size_t)m.seq * max_pages + pg] : 0;
    const u32x4_t* kp = kcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD / 8) + lane;

    // q (and, in the block that holds the step's own page, the new K row) into LDS; then the block barrier
    auto stage_q = [&]() {
        if (FUSED) {
            for (int hh = wave; hh < G + (own ? 1 : 0); hh += 4) {
                const bool isk = hh == G;
                float a, b, o1, o2;
                FuseVec fv;
                if (pre && hh == wave) {                  // requested before the page
                    fv = pfv;
                    fuse_reduce_sum(f, pta, ptb, a, b);
                } else {
                    fv = fuse_load_vec(f, isk ? f.knorm_w : f.qnorm_w, m.pos, lane);
                    fuse_reduce(f, r, (isk ? nq + kvh : kvh * G + hh) * MTTS_HD, lane, a, b);
                }
                fuse_norm_rope(f, fv, a, b, o1, o2);
                if (!isk) {
                    ((uint16_t*)qs[hh])[lane] = f2bf(o1);
                    ((uint16_t*)qs[hh])[lane + 64] = f2bf(o2);
                } else {
                    knew[lane] = f2bf(o1);
                    knew[lane + 64] = f2bf(o2);
                    const int opage = page_table[(size_t)m.seq * max_pages + own_pg];
                    uint16_t* base = (uint16_t*)kcache + ((size_t)kvh * total_pages + opage) * (MTTS_PAGE * MTTS_HD);
                    const int tok = m.pos & 63;        // element (tok, d) at ((d/8)*64 + tok)*8 + d%8
                    base[(((lane >> 3) * 64) + tok) * 8 + (lane & 7)] = f2bf(o1);
                    base[((((lane + 64) >> 3) * 64) + tok) * 8 + (lane & 7)] = f2bf(o2);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < (G * MTTS_HD / 2 + 255) / 256; ++i) {
                const int idx = threadIdx.x + 256 * i;
                if (idx < G * MTTS_HD / 2) (&qs[0][0])[idx] = qreg[i];      // heads kvh*G .. +G-1 are contiguous in qbuf
            }
        }
        __syncthreads();                              // (waits for LDS only: the page loads keep flying across it)
    };
    // rounding points, the page's scores and its softmax statistics
    auto finish = [&](const float* acc) {
        const int tok = pg * MTTS_PAGE + lane;
        const bool valid = tok < len;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int h = kvh * G + g;
            float sc = rbf(rbf(acc[g]) * scale);
            if (valid) scores[((size_t)r * nq + h) * Lmax + tok] = f2bf(sc);
            float mx = wave_max(valid ? sc : -INFINITY);
            float e = valid ? expf(sc - mx) : 0.f;
            float sm = wave_sum(e);
            if (lane == 0) {
                float* st = stats + (((size_t)r * nq + h) * max_pages + pg) * 2;
                st[0] = mx;
                st[1] = sm;
            }
        }
    };
    // v_dot2c_f32_bf16: two bf16 products per lane-op, fp32 accumulate.  q is read as a wave-uniform (broadcast) 16-byte
    // LDS word per 8 dims.  `kv` = the bf16 page (16 units per lane), already requested.
    auto raw_dots = [&](u32x4_t (&kv)[16]) {
        if (FUSED && pg == own_pg && lane == (m.pos & 63)) {      // this lane's token is the new one: take its K from LDS
#pragma unroll
            for (int j = 0; j < 16; ++j) kv[j] = *(const u32x4_t*)&knew[8 * j];
        }
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4_t q = *(const u32x4_t*)&qs[g][4 * j];
                acc[g] = dot2bf(kv[j].x, q.x, acc[g]);
                acc[g] = dot2bf(kv[j].y, q.y, acc[g]);
                acc[g] = dot2bf(kv[j].z, q.z, acc[g]);
                acc[g] = dot2bf(kv[j].w, q.w, acc[g]);
            }
            // (every chain ends here: otherwise the optimiser runs the heads one after the other and keeps more alive)
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc[g]));
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the q reads from being hoisted into 100s of VGPRs
        }
        finish(acc);
    };

    if (PK && pg < own_pg) {                          // (wave-uniform) a complete page: its sealed form
        u32x4_t pk[MTTS_PKU];
        const u32x4_t* pp = kpack + ((size_t)kvh * total_pages + page) * (MTTS_PKU * 64) + lane;
        // in the order the dot products need them: the dictionary, then per 4 units of work their nibble unit and their
        // two low-byte units
#pragma unroll
        for (int j = 0; j < MTTS_PKU; ++j) pk[pk_order(j)] = __builtin_nontemporal_load(pp + pk_order(j) * 64);
        stage_q();
        // the page's K is k 2^-s per dim: this wave's q becomes q 2^s (exact, or flagged)
        bool qok = true;
        const uint32_t sh = pk[12].z;                 // shifts of dims 2 lane, 2 lane + 1
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t qd = qs[g][lane];
            uint32_t out = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t b = (qd >> (16 * h)) & 0xffffu;
                const int sft = (int)(int8_t)((sh >> (8 * h)) & 0xffu);
                if ((b & 0x7fffu) && sft) {
                    const int e = (int)((b >> 7) & 0xffu), e2 = e + sft;
                    if (e == 0 || e == 255 || e2 < 1 || e2 > 254) qok = false;
                    else b = (b + ((uint32_t)sft << 7)) & 0xffffu;
                }
                out |= b << (16 * h);
            }
            qw[PK ? wave : 0][g][lane] = out;
        }
        __builtin_amdgcn_wave_barrier();
        if (__any(pk[12].w != 0u || !qok)) {          // a row of this page (or this q) did not fit: take the bf16 page
            u32x4_t kv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) kv[j] = __builtin_nontemporal_load(kp + j * 64);
            raw_dots(kv);
            return;
        }
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w[4];
            pk_unit(pk, j, w);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4_t q = *(const u32x4_t*)&qw[PK ? wave : 0][g][4 * j];
                acc[g] = dot2bf(w[0], q.x, acc[g]);
                acc[g] = dot2bf(w[1], q.y, acc[g]);
                acc[g] = dot2bf(w[2], q.z, acc[g]);
                acc[g] = dot2bf(w[3], q.w, acc[g]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" : "+v"(acc[g]));
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        finish(acc);
    } else {                                          // the page being filled (or an engine without sealed pages): bf16
        u32x4_t kv[16];                              // (a wave past the row's last page reads page 0 and drops it: no branch,
#pragma unroll                                    //  so the wait counts below stay exact)
        for (int j = 0; j < 16; ++j) kv[j] = __builtin_nontemporal_load(kp + j * 64);
        stage_q();
        if (pg >= npages) return;
        raw_dots(kv);
    }
}

#ifndef PV_WAVES
#define PV_WAVES 4            // waves per pass-B block (ATT_PB / PV_WAVES pages each)
#endif
// grid = (ceil(pages/ATT_PB), nkv, R); block PV_WAVES x 64.  V page is [token pair][d][2]: one dword holds
// (v[2i][d], v[2i+1][d]) so that P.V is a v_dot2c_f32_bf16 against the packed (already bf16-rounded,
// hence exact) probability pair: out[d] += p[2i]*v[2i][d] + p[2i+1]*v[2i+1][d].  Lane l covers
// d = 4*(l&31).. of token pair 2*it + (l>>5): one contiguous KiB per wave instruction.
// FUSED (decode rows): the wave whose pages hold position `pos` reduces the new V row from the qkv GEMM's slabs,
// writes it to the cache and patches it into the page it has just loaded.
// SF: the (max, sumexp) pairs of the row's first 64 pages are requested BEFORE the V page (loads return in order: behind
// 13-16 KiB of page they would arrive last, and the statistics would be reduced after the page has landed instead of
// while it is in flight).
template <int G, bool FUSED, bool PK, bool SF>
__global__ __launch_bounds__(PV_WAVES * 64) PK_OCC(PK) void attn_pv_kernel(
    const uint16_t* __restrict__ scores, const float* __restrict__ stats, u32x4_t* __restrict__ vcache,
    const int32_t* __restrict__ page_table, const RowMeta* __restrict__ meta, float* __restrict__ opart,
    int max_pages, int total_pages, int nchunks_max, int nq, int nkv, QkvFuse f, const u32x4_t* __restrict__ vpack) {
    __shared__ float red[PV_WAVES][G][MTTS_HD];
    __shared__ uint16_t pbuf[PV_WAVES][G][MTTS_PAGE];
    __shared__ __attribute__((aligned(16))) uint16_t vnew[MTTS_HD];
    const int r = blockIdx.z, kvh = blockIdx.y, chunk = blockIdx.x;
    const RowMeta m = meta[r];
    if (m.seq < 0) return;
    const int len = m.pos + 1;
    const int npages = (len + MTTS_PAGE - 1) / MTTS_PAGE;
    if (chunk * ATT_PB >= npages) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Lmax = max_pages * MTTS_PAGE;
    const int sub = lane >> 5, dl = lane & 31;
    float2 s0[G];
    if (SF) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float* st = stats + ((size_t)r * nq + kvh * G + g) * max_pages * 2;
            s0[g] = lane < npages ? *(const float2*)(st + 2 * lane) : float2{-INFINITY, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // V loads of this wave's first page go out before the softmax statistics are reduced
    u32x4_t vv[16];
    int pg = chunk * ATT_PB + wave * (ATT_PB / PV_WAVES);
    const int own_pg = m.pos >> 6;
    bool packed = false;
    const u32x4_t* vp = nullptr;
    uint16_t sraw[G];                                 // this lane's token of the page: its score per head
    auto load_page = [&]() {                          // complete pages are sealed: 13 loads per lane instead of 16
        if (SF) {                                     // the page's scores first: two bytes that would otherwise queue behind the page
            const int tok = pg * MTTS_PAGE + lane;
#pragma unroll
            for (int g = 0; g < G; ++g) sraw[g] = tok < len ? scores[((size_t)r * nq + kvh * G + g) * Lmax + tok] : (uint16_t)0;
            __builtin_amdgcn_sched_barrier(0);
        }
        const int page = page_table[(size_t)m.seq * max_pages + pg];
        vp = vcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD / 8) + lane;
        packed = PK && pg < own_pg;
        if (packed) {
            const u32x4_t* pp = vpack + ((size_t)kvh * total_pages + page) * (MTTS_PKU * 64) + lane;
#pragma unroll
            for (int it = 0; it < MTTS_PKU; ++it) vv[pk_order(it)] = __builtin_nontemporal_load(pp + pk_order(it) * 64);
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) vv[it] = __builtin_nontemporal_load(vp + it * 64);
        }
    };
    // fused epilogue: the wave that owns the step's page requests the new V row's split-K slabs before its page (SF)
    const bool own_wave = FUSED && own_pg >= pg && own_pg < pg + ATT_PB / PV_WAVES;       // one wave per (row, kv head)
    const bool pre_v = SF && own_wave && f.ksplit <= 8;
    float vta[8], vtb[8];
    if (pre_v) fuse_reduce_request(f, r, (nq + nkv + kvh) * MTTS_HD, lane, vta, vtb);
    if (pg < npages) load_page();
    if (own_wave) {
        float a, b;
        if (pre_v) fuse_reduce_sum(f, vta, vtb, a, b);
        else fuse_reduce(f, r, (nq + nkv + kvh) * MTTS_HD, lane, a, b);
        vnew[lane] = f2bf(a);
        vnew[lane + 64] = f2bf(b);
        const int page = page_table[(size_t)m.seq * max_pages + own_pg];
        const int tok = m.pos & 63;          // element (tok, d) at ((tok>>1)*128 + d)*2 + (tok&1)
        uint16_t* dst = (uint16_t*)vcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD) + (size_t)(tok >> 1) * (MTTS_HD * 2) + (tok & 1);
        dst[2 * lane] = f2bf(a);
        dst[2 * (lane + 64)] = f2bf(b);
    }
    // row-wide softmax statistics from the per-page (max, sumexp) pairs
    float M[G], S[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float* st = stats + ((size_t)r * nq + kvh * G + g) * max_pages * 2;
        float mx = -INFINITY;
        if (SF) mx = fmaxf(mx, s0[g].x);
        for (int p = lane + (SF ? 64 : 0); p < npages; p += 64) mx = fmaxf(mx, st[2 * p]);
        mx = wave_max(mx);
        float sm = 0.f;
        if (SF && lane < npages) sm += s0[g].y * expf(s0[g].x - mx);
        for (int p = lane + (SF ? 64 : 0); p < npages; p += 64) sm += st[2 * p + 1] * expf(st[2 * p] - mx);
        sm = wave_sum(sm);
        M[g] = mx;
        S[g] = sm;
    }
    float acc[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = 0.f;
#pragma unroll 1
    for (int pp = 0; pp < ATT_PB / PV_WAVES; ++pp, ++pg) {
        if (pg >= npages) break;
        if (pp > 0) load_page();
        if (own_wave && pg == own_pg) {
            // the page was loaded before the new row reached the cache: patch the token's half of its pair
            __builtin_amdgcn_wave_barrier();
            const int tp = (m.pos & 63) >> 1;             // token pair inside the page
            if (sub == (tp & 1)) {
                const uint32_t w01 = ((const uint32_t*)vnew)[2 * dl], w23 = ((const uint32_t*)vnew)[2 * dl + 1];
                const uint32_t nv[4] = {w01 & 0xffffu, w01 >> 16, w23 & 0xffffu, w23 >> 16};   // d = 4*dl .. 4*dl+3
                const bool hi = m.pos & 1;
#pragma unroll
                for (int it = 0; it < 16; ++it)
                    if (it == (tp >> 1)) {
                        vv[it].x = hi ? (vv[it].x & 0xffffu) | (nv[0] << 16) : (vv[it].x & 0xffff0000u) | nv[0];
                        vv[it].y = hi ? (vv[it].y & 0xffffu) | (nv[1] << 16) : (vv[it].y & 0xffff0000u) | nv[1];
                        vv[it].z = hi ? (vv[it].z & 0xffffu) | (nv[2] << 16) : (vv[it].z & 0xffff0000u) | nv[2];
                        vv[it].w = hi ? (vv[it].w & 0xffffu) | (nv[3] << 16) : (vv[it].w & 0xffff0000u) | nv[3];
                    }
            }
        }
        // lane t rounds the probability of token pg*64+t once (bf16, as the reference stores it);
        // the V loop reads pairs back from LDS (same wave: LDS ops are ordered).
        uint16_t pun[G];                              // the probabilities as the reference rounds them
        bool pok = true;
        {
            const int tok = pg * MTTS_PAGE + lane;
            // a sealed V page holds token t's values divided by 2^s[t] (lane t keeps s[t]): its probability takes the 2^s[t]
            const int sv = (PK && packed) ? (int)(vv[12].z & 0xffu) : 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float p = 0.f;
                if (tok < len) {
                    float s = bf2f(SF ? sraw[g] : scores[((size_t)r * nq + kvh * G + g) * Lmax + tok]);
                    p = expf(s - M[g]) / S[g];
                }
                uint32_t pb = f2bf(p);
                pun[g] = (uint16_t)pb;
                if (sv && (pb & 0x7fffu)) {
                    if (((pb >> 7) & 0xffu) == 0u) pok = false;       // a denormal probability cannot be rescaled exactly
                    else pb += (uint32_t)sv << 7;                     // p <= 1, s <= 127: the exponent field stays <= 254
                }
                pbuf[wave][g][lane] = (uint16_t)pb;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (PK && packed && __any(vv[12].w != 0u || !pok)) {  // a lane did not fit the sealed form (or a probability its scale): the bf16 page
            packed = false;
#pragma unroll
            for (int it = 0; it < 16; ++it) vv[it] = __builtin_nontemporal_load(vp + it * 64);
#pragma unroll
            for (int g = 0; g < G; ++g) pbuf[wave][g][lane] = pun[g];
            __builtin_amdgcn_wave_barrier();
        }
        if (PK && packed) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                uint32_t w[4];
                pk_unit(vv, it, w);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t pp2 = ((const uint32_t*)&pbuf[wave][g][0])[it * 2 + sub];
                    acc[g][0] = dot2bf(w[0], pp2, acc[g][0]);
                    acc[g][1] = dot2bf(w[1], pp2, acc[g][1]);
                    acc[g][2] = dot2bf(w[2], pp2, acc[g][2]);
                    acc[g][3] = dot2bf(w[3], pp2, acc[g][3]);
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t pp2 = ((const uint32_t*)&pbuf[wave][g][0])[it * 2 + sub];
                    acc[g][0] = dot2bf(vv[it].x, pp2, acc[g][0]);
                    acc[g][1] = dot2bf(vv[it].y, pp2, acc[g][1]);
                    acc[g][2] = dot2bf(vv[it].z, pp2, acc[g][2]);
                    acc[g][3] = dot2bf(vv[it].w, pp2, acc[g][3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[g][i];
            v = add_xor32(v);
            if (sub == 0) red[wave][g][dl * 4 + i] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < G * MTTS_HD; i += PV_WAVES * 64) {
        int g = i / MTTS_HD, d = i % MTTS_HD;
        float v = red[0][g][d];
#pragma unroll
        for (int w = 1; w < PV_WAVES; ++w) v += red[w][g][d];
        opart[(((size_t)r * nq + kvh * G + g) * nchunks_max + chunk) * MTTS_HD + d] = v;
    }
}

// grid = (R, ceil(nq/4)), block 512 = 4 heads x 128 dims.  The chunk partials are loaded 8 at a time (independent loads; a loop that adds as it
// loads would pay one memory round trip per chunk) and summed in chunk order.
__global__ __launch_bounds__(512) void attn_combine_kernel(const float* __restrict__ opart, const RowMeta* __restrict__ meta,
                                                           uint16_t* __restrict__ out_packed, int nchunks_max, int nq,
                                                           int pages_per_chunk) {
    const int r = blockIdx.x, h = blockIdx.y * 4 + (threadIdx.x >> 7), d = threadIdx.x & 127;
    if (h >= nq) return;
    const RowMeta m = meta[r];
    float s = 0.f;
    const int npages = m.seq >= 0 ? (m.pos + 1 + MTTS_PAGE - 1) / MTTS_PAGE : 0;
    const int nch = (npages + pages_per_chunk - 1) / pages_per_chunk;
    const float* p = opart + ((size_t)r * nq + h) * nchunks_max * MTTS_HD + d;
    for (int c0 = 0; c0 < nch; c0 += 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = p[(size_t)min(c0 + j, nch - 1) * MTTS_HD];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < nch) s += t[j];
    }
    out_packed[xpack_off(r, h * MTTS_HD + d, nq * MTTS_HD)] = f2bf(s);
}

// ---------------------------------------------------------------------------------------------------
// Prefill attention: the 32 rows of an activation tile are consecutive positions of ONE dialogue (the host
// pads every dialogue's prompt to whole tiles), so a tile shares its K/V pages: one block streams a page once
// for 32 query rows instead of once per row, with the products on the matrix cores.  Same rounding points,
// same scratch (scores, per-page statistics, chunk partials) and the same combine kernel as the decode path.
//   v_mfma_f32_32x32x16_bf16: A lane l = (row l&31, k 8*(l>>5)..+8), B lane l = (col l&31, k 8*(l>>5)..+8),
//   D lane l = col l&31, rows (i&3) + 8*(i>>2) + 4*(l>>5) for i = 0..15.
// ---------------------------------------------------------------------------------------------------

// scores: D[token][row] = K page (A: tokens x d) . Q^T (B: d x rows).  grid = (ceil(pages/4), nkv, tiles),
// block 256 = 4 waves, one page per wave.  The K page layout [d/8][token][8] is the A operand as stored.
template <int G>
__global__ __launch_bounds__(256) void attn_prefill_scores_kernel(
    const uint16_t* __restrict__ qbuf, const u32x4_t* __restrict__ kcache, const int32_t* __restrict__ page_table,
    const RowMeta* __restrict__ meta, uint16_t* __restrict__ scores, float* __restrict__ stats, int max_pages,
    int total_pages, int nq, int nkv, float scale) {
    const int tile = blockIdx.z, kvh = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = tile * MTTS_MAXR + (lane & 31);
    const RowMeta m0 = meta[tile * MTTS_MAXR];          // a tile with any live row has its first row live
    if (m0.seq < 0) return;
    const RowMeta mr = meta[row];
    const int pos = mr.seq >= 0 ? mr.pos : -1;           // idle rows see no token
    int maxpos = pos;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) maxpos = max(maxpos, __shfl_xor(maxpos, o, 64));
    const int npages = (maxpos + 1 + MTTS_PAGE - 1) / MTTS_PAGE;
    const int pg = blockIdx.x * 4 + wave;
    if (pg >= npages) return;
    const int page = page_table[(size_t)m0.seq * max_pages + pg];
    const u32x4_t* kp = kcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD / 8);
    // A fragments: token half th (32 tokens), d step ks (16 dims): 16-byte group (2*ks + (lane>>5)) of token th*32 + (lane&31)
    u32x4_t ka[2][8];
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) ka[th][ks] = kp[(2 * ks + (lane >> 5)) * 64 + th * 32 + (lane & 31)];
    const int Lmax = max_pages * MTTS_PAGE;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int h = kvh * G + g;
        const u32x4_t* qp = (const u32x4_t*)(qbuf + ((size_t)row * nq + h) * MTTS_HD);
        u32x4_t qb[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qb[ks] = qp[2 * ks + (lane >> 5)];
        float mx = -INFINITY;
        f32x16_t acc[2];
#pragma unroll
        for (int th = 0; th < 2; ++th) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[th][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                acc[th] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&ka[th][ks], *(bf16x8_t*)&qb[ks], acc[th], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tok = pg * MTTS_PAGE + th * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const float sc = rbf(rbf(acc[th][i]) * scale);
                const bool valid = tok <= pos;
                if (valid) scores[((size_t)row * nq + h) * Lmax + tok] = f2bf(sc);
                acc[th][i] = valid ? sc : -INFINITY;
                mx = fmaxf(mx, acc[th][i]);
            }
        }
        mx = max_xor32(mx);           // lanes l and l^32 hold the two halves of a row's tokens
        float sm = 0.f;
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm += (acc[th][i] > -INFINITY) ? expf(acc[th][i] - mx) : 0.f;
        sm = add_xor32(sm);
        if (lane < 32 && mr.seq >= 0) {
            float* st = stats + (((size_t)row * nq + h) * max_pages + pg) * 2;
            st[0] = mx;
            st[1] = sm;
        }
    }
}

// P.V: D[d][row] = V^T (A: d x tokens) . P^T (B: tokens x rows), probabilities rounded to bf16 first.
// grid = (ceil(pages/(4*ATT_PF)), nkv, tiles), block 256: each wave owns ATT_PF pages and all 128 head dims and
// writes its own chunk partial (chunks of ATT_PF pages for prefill rows), so there is no cross-wave reduction.
// The V page layout [token pair][d][2] gives the A operand as four dwords per lane (8 consecutive tokens of one d).
template <int G>
__global__ __launch_bounds__(256) void attn_prefill_pv_kernel(
    const uint16_t* __restrict__ scores, const float* __restrict__ stats, const uint32_t* __restrict__ vcache,
    const int32_t* __restrict__ page_table, const RowMeta* __restrict__ meta, float* __restrict__ opart,
    int max_pages, int total_pages, int nchunks_pf, int nq, int nkv) {
    constexpr int GP = G > 2 ? 2 : G;                    // heads per sweep over the pages (accumulator registers)
    const int tile = blockIdx.z, kvh = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
    const int chunk = blockIdx.x * 4 + wave;
    const int row = tile * MTTS_MAXR + (lane & 31);
    const RowMeta m0 = meta[tile * MTTS_MAXR];
    if (m0.seq < 0) return;
    const RowMeta mr = meta[row];
    const int pos = mr.seq >= 0 ? mr.pos : -1;
    int maxpos = pos;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) maxpos = max(maxpos, __shfl_xor(maxpos, o, 64));
    const int npages = (maxpos + 1 + MTTS_PAGE - 1) / MTTS_PAGE;
    if (chunk * ATT_PF >= npages) return;
    const int rpages = (pos + 1 + MTTS_PAGE - 1) / MTTS_PAGE;   // this row's own pages
    const int Lmax = max_pages * MTTS_PAGE;
    const int pg_end = min(npages, (chunk + 1) * ATT_PF);
#pragma unroll 1
    for (int g0 = 0; g0 < G; g0 += GP) {
        // row-wide softmax statistics: the two lanes of a row take alternate pages, four loads in flight
        float M[GP], S[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) {
            const float* st = stats + ((size_t)row * nq + kvh * G + g0 + g) * max_pages * 2;
            float mx = -INFINITY;
            for (int p = half; p < rpages; p += 8) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = st[2 * min(p + 2 * j, rpages - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, t[j]);
            }
            mx = max_xor32(mx);
            float sm = 0.f;
            for (int p = half; p < rpages; p += 8) {
                float2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = *(const float2*)(st + 2 * min(p + 2 * j, rpages - 1));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p + 2 * j < rpages) sm += t[j].y * expf(t[j].x - mx);
            }
            sm = add_xor32(sm);
            M[g] = mx;
            S[g] = sm;
        }
        f32x16_t acc[GP][4];
#pragma unroll
        for (int g = 0; g < GP; ++g)
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[g][db][i] = 0.f;
#pragma unroll 1
        for (int pg = chunk * ATT_PF; pg < pg_end; ++pg) {
            const int page = page_table[(size_t)m0.seq * max_pages + pg];
            const uint32_t* vp = vcache + ((size_t)kvh * total_pages + page) * (MTTS_PAGE * MTTS_HD / 2) + (lane & 31);
            // the page's score words for this lane's rows/tokens go out first, then V
            u32x4_t sv[4][GP];
#pragma unroll
            for (int ts = 0; ts < 4; ++ts) {
                const int tok0 = pg * MTTS_PAGE + ts * 16 + 8 * half;
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    sv[ts][g] = u32x4_t{0u, 0u, 0u, 0u};
                    if (tok0 <= pos) sv[ts][g] = *(const u32x4_t*)(scores + ((size_t)row * nq + kvh * G + g0 + g) * Lmax + tok0);
                }
            }
#pragma unroll
            for (int ts = 0; ts < 4; ++ts) {                 // 16 tokens per MFMA
                const int t0 = ts * 16 + 8 * half;           // this lane's 8 tokens inside the page
                u32x4_t va[4];
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const uint32_t* q = vp + db * 32 + (size_t)(t0 / 2) * MTTS_HD;
                    va[db].x = q[0];
                    va[db].y = q[MTTS_HD];
                    va[db].z = q[2 * MTTS_HD];
                    va[db].w = q[3 * MTTS_HD];
                }
                const int tok0 = pg * MTTS_PAGE + t0;
#pragma unroll
                for (int g = 0; g < GP; ++g) {
                    const uint32_t sw[4] = {sv[ts][g].x, sv[ts][g].y, sv[ts][g].z, sv[ts][g].w};
                    uint32_t pw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p0 = (tok0 + 2 * j <= pos) ? expf(bflo(sw[j]) - M[g]) / S[g] : 0.f;
                        const float p1 = (tok0 + 2 * j + 1 <= pos) ? expf(bfhi(sw[j]) - M[g]) / S[g] : 0.f;
                        pw[j] = pack2(p0, p1);
                    }
                    u32x4_t pb = {pw[0], pw[1], pw[2], pw[3]};
#pragma unroll
                    for (int db = 0; db < 4; ++db)
                        acc[g][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&va[db], *(bf16x8_t*)&pb, acc[g][db], 0, 0, 0);
                }
            }
        }
        if (mr.seq >= 0) {
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                float* o = opart + (((size_t)row * nq + kvh * G + g0 + g) * nchunks_pf + chunk) * MTTS_HD + 4 * half;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v4 = make_float4(acc[g][db][4 * q], acc[g][db][4 * q + 1], acc[g][db][4 * q + 2], acc[g][db][4 * q + 3]);
                        *(float4*)(o + db * 32 + 8 * q) = v4;
                    }
            }
        }
    }
}

static bool pv_stats_first() {          // MTTS_PV_STATS_FIRST=0: the round-1 order (page loads first)
    static int v = -1;
    if (v < 0) {
        const char* g = getenv("MTTS_PV_STATS_FIRST");
        v = (g && atoi(g) == 0) ? 0 : 1;
    }
    return v != 0;
}
template <int G>
static void launch_attn_g(const void* qbuf, void* kcache, void* vcache, const int32_t* page_table,
                          const RowMeta* meta, void* scores, float* stats, float* opart, void* out_packed, int R,
                          int pages_bound, int max_pages, int total_pages, int nchunks_max, int nq, int nkv, float scale,
                          const QkvFuse* fuse, int phase, hipStream_t st, const KvPack* pack) {
    if (phase == 11) {   // prefill tiles (32 consecutive positions of one dialogue per tile)
        dim3 ga((pages_bound + 3) / 4, nkv, R / MTTS_MAXR);
        hipLaunchKernelGGL((attn_prefill_scores_kernel<G>), ga, dim3(256), 0, st, (const uint16_t*)qbuf, (const u32x4_t*)kcache,
                           page_table, meta, (uint16_t*)scores, stats, max_pages, total_pages, nq, nkv, scale);
        return;
    }
    const int nchunks_pf = (max_pages + ATT_PF - 1) / ATT_PF;      // prefill rows: chunks of ATT_PF pages
    if (phase == 12) {
        dim3 gb((pages_bound + 4 * ATT_PF - 1) / (4 * ATT_PF), nkv, R / MTTS_MAXR);
        hipLaunchKernelGGL((attn_prefill_pv_kernel<G>), gb, dim3(256), 0, st, (const uint16_t*)scores, (const float*)stats,
                           (const uint32_t*)vcache, page_table, meta, opart, max_pages, total_pages, nchunks_pf, nq, nkv);
        return;
    }
    if (phase == 13) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3(R, (nq + 3) / 4), dim3(512), 0, st, (const float*)opart, meta,
                           (uint16_t*)out_packed, nchunks_pf, nq, ATT_PF);
        return;
    }
    const QkvFuse f = fuse ? *fuse : QkvFuse{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f};
    const u32x4_t* kpk = pack ? (const u32x4_t*)pack->k : nullptr;
    const u32x4_t* vpk = pack ? (const u32x4_t*)pack->v : nullptr;
#define MTTS_SC(FU, PK)                                                                                                  \
    hipLaunchKernelGGL((attn_scores_kernel<G, FU, PK>), ga, dim3(256), 0, st, (const uint16_t*)qbuf, (u32x4_t*)kcache, \
                       page_table, meta, (uint16_t*)scores, stats, max_pages, total_pages, nq, nkv, scale, f, kpk)
#define MTTS_PV2(FU, PK, SF)                                                                                                        \
    hipLaunchKernelGGL((attn_pv_kernel<G, FU, PK, SF>), gb, dim3(PV_WAVES * 64), 0, st, (const uint16_t*)scores, (const float*)stats, \
                       (u32x4_t*)vcache, page_table, meta, opart, max_pages, total_pages, nchunks_max, nq, nkv, f, vpk)
#define MTTS_PV(FU, PK)                        \
    do {                                       \
        if (pv_stats_first()) MTTS_PV2(FU, PK, true); \
        else MTTS_PV2(FU, PK, false);          \
    } while (0)
    if (phase == 0 || phase == 1) {
        dim3 ga((pages_bound + 3) / 4, nkv, R);
        if (fuse) { if (kpk) MTTS_SC(true, true); else MTTS_SC(true, false); }
        else { if (kpk) MTTS_SC(false, true); else MTTS_SC(false, false); }
    }
    if (phase == 0 || phase == 2) {
        dim3 gb((pages_bound + ATT_PB - 1) / ATT_PB, nkv, R);
        if (fuse) { if (vpk) MTTS_PV(true, true); else MTTS_PV(true, false); }
        else { if (vpk) MTTS_PV(false, true); else MTTS_PV(false, false); }
    }
#undef MTTS_SC
#undef MTTS_PV
#undef MTTS_PV2
    if (phase == 0 || phase == 3)
        hipLaunchKernelGGL(attn_combine_kernel, dim3(R, (nq + 3) / 4), dim3(512), 0, st, (const float*)opart, meta,
                           (uint16_t*)out_packed, nchunks_max, nq, ATT_PB);
}

// phase 1/2/3 = scores / P.V / combine for decode-style rows (one dialogue per row), 11/12/13 = the same for prefill
// tiles.  `fuse` (decode rows only) moves the q/k/v epilogue into phases 1 and 2: no qkv_post launch before them.
int launch_attn(const void* qbuf, void* kcache, void* vcache, const int32_t* page_table,
                const RowMeta* meta, void* scores, float* stats, float* opart, void* out_packed, int R,
                int pages_bound, int max_pages, int total_pages, int nchunks_max, int nq, int nkv, float scale,
                const QkvFuse* fuse, int phase, hipStream_t st, const KvPack* pack) {
    int G = nq / nkv;
#define MTTS_ATT(GG)                                                                                              \
    launch_attn_g<GG>(qbuf, kcache, vcache, page_table, meta, scores, stats, opart, out_packed, R, pages_bound,   \
                      max_pages, total_pages, nchunks_max, nq, nkv, scale, fuse, phase, st, pack)
    if (G == 1) MTTS_ATT(1);
    else if (G == 2) MTTS_ATT(2);
    else if (G == 4) MTTS_ATT(4);
    else return -1;
#undef MTTS_ATT
    return 0;
}
