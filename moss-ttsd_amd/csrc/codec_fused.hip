// Vocos ConvNeXt block, the two pointwise convolutions in ONE launch:
//     h += gamma * (W2 . GELU(W1 . xn + b1) + b2)          (reference nn/modules.py:1135-1154, after dwconv + LayerNorm)
// with the 4096-wide intermediate never leaving the CU.  The two-launch form (codec.hip: gemm_b3t_kernel twice) writes
// 393 MB of hi / lo planes per launch of 24 000 rows and reads them straight back.
//
// Same arithmetic as the two launches: every fp32 product is three bf16 MFMAs (a_hi b_lo + a_lo b_hi + a_hi b_hi, in that
// order, fp32 accumulate), GELU by the same erf approximation, the same hi / lo split of the intermediate.
//
// Block = 4 waves (one per SIMD) x 64 rows.  LDS (all 160 KiB): the block's input rows as hi / lo planes in MFMA-fragment
// order (128 KiB, loaded once) + one 128-column chunk of the GELU'd intermediate as hi / lo B-operand fragments (32 KiB).
// Per chunk: GEMM 1 -- wave w computes the 32 intermediate columns 32 (4c + w).. for both row tiles (K = 512: W1 fragments
// from L2, X fragments from LDS); its bias + GELU + split epilogue is issued one chunk LATER, in eight units inside the next
// chunk's GEMM-1 MFMA stream, and parks the result in LDS; barrier; GEMM 2 -- wave w accumulates its 128 output columns x
// 64 rows over the chunk's 128 intermediate columns (W2 fragments from L2, the chunk from LDS); barrier.
// 192 + 192 MFMAs per wave and chunk.
//
// No shuffle between the two GEMMs: a GEMM-1 accumulator tile D[j][row] holds, in lane (row, half h), the columns
// j = 32 jt + 8 q + 4 h + r (register 4 q + r).  W2 is packed with its K index PERMUTED so that k-step 2 jt + ks, lane half h,
// element e means j = 32 jt + 16 ks + 8 (e >> 2) + 4 h + (e & 3): then registers 8 ks .. 8 ks + 7 of the tile, after GELU and
// split, ARE the lane's B-operand fragment of that k-step.
#include "common.h"

typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define FVD 512                 // Vocos width (xy_tokenizer_config.yaml: vocos dim)
#define FVI 4096                // intermediate width
#define FKT1 (FVD / 16)         // k-steps of GEMM 1
#define FKT2 (FVI / 16)         // k-steps of GEMM 2 (over the whole intermediate)
#define FCH 128                 // intermediate columns per chunk

__device__ __forceinline__ void fsplit2(float x, float y, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {x, y};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ f32x2_t fgelu2(f32x2_t v) {          // codec.hip: gelu_fast2 (erf by Abramowitz-Stegun 7.1.26)
    const f32x2_t x = v * 0.70710678118654752440f;
    const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2_t den = ax * 0.3275911f + 1.0f;
    const f32x2_t t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2_t poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const f32x2_t ex = {__expf(-ax.x * ax.x), __expf(-ax.y * ax.y)};
    const f32x2_t e = 1.0f - poly * ex;
    const f32x2_t se = {copysignf(e.x, x.x), copysignf(e.y, x.y)};
    return 0.5f * v * (1.0f + se);
}

#define MFMA3(acc, ah, al, bh, bl)                                                                          \
    do {                                                                                                    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&(ah), *(bf16x8_t*)&(bl), acc, 0, 0, 0);  \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&(al), *(bf16x8_t*)&(bh), acc, 0, 0, 0);  \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&(ah), *(bf16x8_t*)&(bh), acc, 0, 0, 0);  \
    } while (0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void vocos_pw_fused_kernel(
    const u32x4_t* __restrict__ Xh, const u32x4_t* __restrict__ Xl, const u32x4_t* __restrict__ W1h,
    const u32x4_t* __restrict__ W1l, const float* __restrict__ b1, const u32x4_t* __restrict__ W2h,
    const u32x4_t* __restrict__ W2l, const float* __restrict__ b2, const float* __restrict__ gamma, float* __restrict__ h, int M) {
    __shared__ u32x4_t Xs[2][2][FKT1][64];      // [plane][row tile][k-step][lane]                    128 KiB
    __shared__ u32x4_t Ys[2][2][8][64];         // [plane][row tile][k-step of the chunk][lane]        32 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, hh = lane >> 5;
    const int rtiles = (M + 31) / 32, rt0 = blockIdx.x * 2;
    // ---- the block's 64 input rows -> LDS (a tile past the end re-reads the last one; its rows are never stored)
    // (all 32 loads of a thread in flight before the first store: a load-store loop would pay 16 memory round trips here,
    //  and with one block per CU nothing else runs meanwhile)
    {
        u32x4_t xv[2][2 * FKT1 * 64 / 256];
#pragma unroll
        for (int it = 0; it < 2 * FKT1 * 64 / 256; ++it) {
            const int i = threadIdx.x + it * 256;
            const int rt = i / (FKT1 * 64), rem = i % (FKT1 * 64);
            const size_t src = (size_t)min(rt0 + rt, rtiles - 1) * (FKT1 * 64) + rem;
            xv[0][it] = __builtin_nontemporal_load(Xh + src);
            xv[1][it] = __builtin_nontemporal_load(Xl + src);
        }
#pragma unroll
        for (int it = 0; it < 2 * FKT1 * 64 / 256; ++it) {
            const int i = threadIdx.x + it * 256;
            const int rt = i / (FKT1 * 64), rem = i % (FKT1 * 64);
            (&Xs[0][rt][0][0])[rem] = xv[0][it];
            (&Xs[1][rt][0][0])[rem] = xv[1][it];
        }
    }
    __syncthreads();
    f32x16_t acc2[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[t][rt][i] = 0.f;
    const u32x4_t* w2h = W2h + (size_t)(4 * wave) * FKT2 * 64 + lane;       // this wave's four output tiles, k-step 0
    const u32x4_t* w2l = W2l + (size_t)(4 * wave) * FKT2 * 64 + lane;
    constexpr int U = 2;            // k-steps per register set (4: measured no better -- 384 registers)
    struct S1 { u32x4_t ah[U], al[U], bh[2][U], bl[2][U]; };
    struct S2 { u32x4_t ah[4], al[4], bh[2], bl[2]; };
    // fragment loads: W1 / W2 from global memory (L2), X / the chunk from LDS
    auto load1 = [&](S1& f, int jt, int k0) {
        const u32x4_t* w1h = W1h + ((size_t)jt * FKT1 + k0) * 64 + lane;
        const u32x4_t* w1l = W1l + ((size_t)jt * FKT1 + k0) * 64 + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f.ah[u] = w1h[(size_t)u * 64];
            f.al[u] = w1l[(size_t)u * 64];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { f.bh[rt][u] = Xs[0][rt][k0 + u][lane]; f.bl[rt][u] = Xs[1][rt][k0 + u][lane]; }
        }
    };
    auto load2g = [&](S2& f, int ks) {            // ks: k-step over the whole intermediate (8 c + kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) { f.ah[t] = w2h[((size_t)t * FKT2 + ks) * 64]; f.al[t] = w2l[((size_t)t * FKT2 + ks) * 64]; }
    };
    auto load2l = [&](S2& f, int kk) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { f.bh[rt] = Ys[0][rt][kk][lane]; f.bl[rt] = Ys[1][rt][kk][lane]; }
    };
    S1 fa, fb;
    S2 ga, gb;
    f32x16_t acc1[2], accp[2];                     // GEMM-1 tile being accumulated / the finished one awaiting its epilogue
    auto comp1 = [&](S1& f) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) MFMA3(acc1[rt], f.ah[u], f.al[u], f.bh[rt][u], f.bl[rt][u]);
    };
    auto comp2 = [&](S2& f) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) MFMA3(acc2[t][rt], f.ah[t], f.al[t], f.bh[rt], f.bl[rt]);
    };
    float4 bq[4];
    uint32_t fh[4], fl[4];
    // The epilogue of one value pair of the finished tile (bias + GELU + hi / lo split; fgelu2 / fsplit2 cut into six
    // slices of at most seven VALU instructions).  Unit u = (rt, ks, qq), half = its first / second value pair; registers
    // 8 ks .. 8 ks + 7 of a tile are the B fragment of k-step 2 w + ks of the chunk, stored when its last pair is done.
    f32x2_t ev, ex, eax, et, ep, ee;
    auto epi_slice = [&](int u, int half, int sl) {
        const int rt = u >> 2, ks = (u >> 1) & 1, qq = u & 1, q = 2 * ks + qq;
        if (sl == 0) {
            const float bx = half ? bq[q].z : bq[q].x, by = half ? bq[q].w : bq[q].y;
            ev = f32x2_t{accp[rt][4 * q + 2 * half] + bx, accp[rt][4 * q + 2 * half + 1] + by};
            ex = ev * 0.70710678118654752440f;
            eax = f32x2_t{fabsf(ex.x), fabsf(ex.y)};
        } else if (sl == 1) {
            const f32x2_t den = eax * 0.3275911f + 1.0f;
            et = f32x2_t{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            ee = (eax * -1.44269504088896340736f) * eax;            // exp(-ax^2) = exp2(ee)
        } else if (sl == 2) {
            ee = f32x2_t{__builtin_amdgcn_exp2f(ee.x), __builtin_amdgcn_exp2f(ee.y)};
            ep = (1.061405429f * et - 1.453152027f) * et + 1.421413741f;
        } else if (sl == 3) {
            ep = ((ep * et - 0.284496736f) * et + 0.254829592f) * et;
            ee = 1.0f - ep * ee;
        } else if (sl == 4) {
            const f32x2_t se = {copysignf(ee.x, ex.x), copysignf(ee.y, ex.y)};
            ev = 0.5f * ev * (1.0f + se);
        } else if (sl == 5) {
            fsplit2(ev.x, ev.y, fh[2 * qq + half], fl[2 * qq + half]);
        } else if (sl == 6 && qq && half) {
            Ys[0][rt][2 * wave + ks][lane] = u32x4_t{fh[0], fh[1], fh[2], fh[3]};
            Ys[1][rt][2 * wave + ks][lane] = u32x4_t{fl[0], fl[1], fl[2], fl[3]};
        }
    };
    // 12 MFMAs of one fragment set; with EPI one epilogue slice is issued behind each of the first seven, pinned there by
    // scheduling barriers: an MFMA holds the matrix pipe 32 cycles, a VALU instruction issues in 4, so the slices run in
    // its shadow instead of between the two GEMMs.
    auto comp1e = [&](S1& f, int u, int half) {
        int sl = 0;
#pragma unroll
        for (int uu = 0; uu < U; ++uu)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.ah[uu], *(bf16x8_t*)&f.bl[rt][uu], acc1[rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                epi_slice(u, half, sl++);
                __builtin_amdgcn_sched_barrier(0);
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.al[uu], *(bf16x8_t*)&f.bh[rt][uu], acc1[rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                epi_slice(u, half, sl++);
                __builtin_amdgcn_sched_barrier(0);
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t*)&f.ah[uu], *(bf16x8_t*)&f.bh[rt][uu], acc1[rt], 0, 0, 0);
            }
    };
    // GEMM 1 of chunk `cc` (K = 512 in 8 rounds of 4 k-steps), carrying the epilogue of the PREVIOUS chunk's tile (one unit
    // per round).  `fa` holds the chunk's first fragments on entry; `pf2` = k-step of W2 to prefetch at the end.
    auto gemm1 = [&](int cc, bool epi, int pf2) {
        const int jt = 4 * cc + wave;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[rt][i] = 0.f;
#pragma unroll
        for (int it = 0; it < FKT1 / (2 * U); ++it) {
            const int k = it * 2 * U;
            load1(fb, jt, k + U);
            __builtin_amdgcn_sched_barrier(0);
            if (epi) comp1e(fa, it, 0);
            else comp1(fa);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < FKT1 / (2 * U)) load1(fa, jt, k + 2 * U);
            else load2g(ga, pf2);                  // phase 2's first W2 fragments fly under the last MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (epi) comp1e(fb, it, 1);
            else comp1(fb);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int NCH = FVI / FCH;
    load1(fa, wave, 0);
    gemm1(0, false, 0);
    for (int c = 0; c < NCH; ++c) {
        // ---- stage A: the tile of chunk c gets its epilogue while the tile of chunk c + 1 is being accumulated
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) accp[rt] = acc1[rt];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *(const float4*)(b1 + 32 * (4 * c + wave) + 8 * q + 4 * hh);
        if (c + 1 < NCH) {
            load1(fa, 4 * (c + 1) + wave, 0);
            gemm1(c + 1, true, 8 * c);
        } else {
            load2g(ga, 8 * c);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int sl = 0; sl < 7; ++sl) epi_slice(u, half, sl);
        }
        __syncthreads();
        // ---- stage B: 128 output columns x 64 rows += chunk c (K = 128)
        load2l(ga, 0);
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            load2g(gb, 8 * c + kk + 1); load2l(gb, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            comp2(ga);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < 8) { load2g(ga, 8 * c + kk + 2); load2l(ga, kk + 2); }
            __builtin_amdgcn_sched_barrier(0);
            comp2(gb);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // ---- h += gamma * (acc + b2), in place (every element is read and written by one lane)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int nb = (4 * wave + t) * 32 + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nb + 8 * q;
            const float4 bb = *(const float4*)(b2 + n), gg = *(const float4*)(gamma + n);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int m = (rt0 + rt) * 32 + (lane & 31);
                if (m < M) {
                    float4* p = (float4*)(h + (size_t)m * FVD + n);
                    const float4 r = *p;
                    *p = make_float4((acc2[t][rt][4 * q] + bb.x) * gg.x + r.x, (acc2[t][rt][4 * q + 1] + bb.y) * gg.y + r.y,
                                     (acc2[t][rt][4 * q + 2] + bb.z) * gg.z + r.z, (acc2[t][rt][4 * q + 3] + bb.w) * gg.w + r.w);
                }
            }
        }
    }
}

// W2 [512][4096] fp32 -> fragment-packed bf16 hi / lo planes with the K index permuted as the kernel's phase 2 expects
__global__ void split_pack_w2perm_kernel(const float* __restrict__ w, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)FVD * FVI) return;
    const int n = (int)(i / FVI), j = (int)(i % FVI);
    const int jt = j >> 5, jj = j & 31;
    const int ks = (jj >> 4) & 1, eh = (jj >> 3) & 1, hh = (jj >> 2) & 1, r = jj & 3;
    const size_t at = (((size_t)(n >> 5) * FKT2 + (2 * jt + ks)) * 64 + (n & 31) + 32 * hh) * 8 + 4 * eh + r;
    uint32_t h2, l2;
    fsplit2(w[i], 0.f, h2, l2);
    hi[at] = (uint16_t)h2;
    lo[at] = (uint16_t)l2;
}

void launch_split_pack_w2perm(hipStream_t st, const float* w2, uint16_t* hi, uint16_t* lo) {
    const long n = (long)FVD * FVI;
    hipLaunchKernelGGL(split_pack_w2perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w2, hi, lo);
}

// xn planes: hi plane, lo plane `x_plane_elems` bf16 elements further (codec.hip: pad32(M) * 512); w1 planes / w2 planes
// likewise `w_plane_elems` apart (4096 * 512 elements each).
void launch_vocos_pw_fused(hipStream_t st, const uint16_t* xn_planes, long x_plane_elems, const uint16_t* w1_planes,
                           const float* b1, const uint16_t* w2perm_planes, long w_plane_elems, const float* b2,
                           const float* gamma, float* h, int M) {
    hipLaunchKernelGGL(vocos_pw_fused_kernel, dim3((M + 63) / 64), dim3(256), 0, st, (const u32x4_t*)xn_planes,
                       (const u32x4_t*)(xn_planes + x_plane_elems), (const u32x4_t*)w1_planes,
                       (const u32x4_t*)(w1_planes + w_plane_elems), b1, (const u32x4_t*)w2perm_planes,
                       (const u32x4_t*)(w2perm_planes + w_plane_elems), b2, gamma, h, M);
}
