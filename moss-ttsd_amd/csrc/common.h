// Shared device helpers for the mtts HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#define MTTS_WAVE 64
#define MTTS_MAXR 32          // rows per MFMA N-tile (one activation fragment tile)
#define MTTS_RCAP 128         // dialogue slots; rows of a decode pass (up to 4 activation tiles share one weight stream)
#define MTTS_PFCAP 2048       // rows of a prefill pass (tiled GEMM); also the row stride of the split-K slabs
#define MTTS_PAGE 64          // tokens per KV page
#define MTTS_HD 128           // head_dim the kernels are written for
#define ATT_PB 8              // KV pages per pass-B chunk, decode rows (4 waves x 2 pages)
#define ATT_PF 2              // KV pages per pass-B chunk, prefill tiles (one wave)

// bf16 bit pattern <-> fp32.  Round-to-nearest-even, NaN stays NaN, inf stays inf.
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {   // c + a.lo*b.lo + a.hi*b.hi
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// Cross-lane moves without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: ~100 cycles per dependent step):
// DPP reaches lane ^ 1, ^ 2 (quad_perm), ^ 4 (row_shl:4 / row_shr:4 on alternate banks) and ^ 8 (row_ror:8) in one VALU
// instruction (two for ^ 4).  tools/dpp_probe.hip checks the four partner patterns and the max sequence on the GPU.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_f<0xB1>(v, v); }                  // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_f<0x4E>(v, v); }                  // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor4(float v) { return dpp_f<0x114, 0xf, 0xa>(dpp_f<0x104, 0xf, 0x5>(v, v), v); }
__device__ __forceinline__ float lane_xor8(float v) { return dpp_f<0x128>(v, v); }                 // row_ror:8
// v[l] + v[l ^ 32] and v[l] + v[l ^ 16] in every lane: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper
// half (the odd rows) of one copy with the lower half (the even rows) of the other, so the two results are {lo, lo} and
// {hi, hi} and their sum is the pair sum in both halves (same two operands as v + __shfl_xor(v, 32): same bits).
typedef unsigned int u32x2s_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_xor32(float v) {
    const u32x2s_t t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}
__device__ __forceinline__ float max_xor32(float v) {
    const u32x2s_t t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ float add_xor16(float v) {
    const u32x2s_t t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}
// Sum over the wave, in every lane: the xor butterfly 32, 16, 8, 4, 2, 1 of rounds 1-2 -- the SAME tree of additions
// (the sums are bit-identical) -- without the LDS crossbar: lane swaps for the two upper levels, DPP for the lower four.
__device__ __forceinline__ float wave_sum(float v) {
    v = add_xor32(v);
    v = add_xor16(v);
    v += lane_xor8(v);
    v += lane_xor4(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}
// Max over the wave, in every lane (order does not matter for a max): rows of 16 by DPP, then row_bcast15 / row_bcast31
// bring the whole wave's max to lane 63, which is broadcast.
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v, v));
    v = fmaxf(v, dpp_f<0x4E>(v, v));
    v = fmaxf(v, dpp_f<0x141>(v, v));                 // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v, v));                 // row_mirror
    v = fmaxf(v, dpp_f<0x142, 0xa>(v, v));            // row_bcast15 into rows 1, 3
    v = fmaxf(v, dpp_f<0x143, 0xc>(v, v));            // row_bcast31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Packed "fragment" layouts.  A matrix [rows32][K] that feeds
// v_mfma_f32_32x32x16_bf16 is stored as [row_tile][k/16][lane][8] with
// lane = (row%32) + 32*((k%16)/8), element = k%8: one wave-instruction reads one
// contiguous KiB.  Element offset helpers (in bf16 elements):
__host__ __device__ __forceinline__ size_t wpack_off(int n, int k, int KT) {
    return ((((size_t)(n >> 5) * KT + (k >> 4)) * 64) + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
}
// activations: [row tile r/32][k/16][lane][8]; K = row length of the matrix
__host__ __device__ __forceinline__ size_t xpack_off(int r, int k, int K) {
    return (size_t)(r >> 5) * 32 * K + (((size_t)(k >> 4) * 64) + (r & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
}

// Decode rows: the q/k/v epilogue (split-K reduce, per-head RMSNorm, RoPE, cache write) runs inside the attention
// kernels instead of a launch of its own; they need the qkv GEMM's slabs and the layer's small vectors.
struct QkvFuse {
    const float* partial;      // [ksplit][MTTS_PFCAP][Npad] fp32
    int ksplit, Npad;
    const uint16_t *qnorm_w, *knorm_w, *rope_cos, *rope_sin;
    float eps;
};

// Sealed KV pages (attn.hip: kv_seal): a COMPLETE page also exists in a 13-bit lossless form, 13 sixteen-byte units per
// lane instead of 16 -- units 0..7 the low bytes of the lane's 128 values, 8..11 a 4-bit code per value (sign + index into
// the lane's dictionary), unit 12 = {dictionary of up to 8 distinct "bf16 high byte without the sign" values, 0, flag};
// flag != 0: the lane's values needed more than 8 entries, the page is read in its bf16 form.
#define MTTS_PKU 13
struct KvPack { void* k; void* v; };             // this layer's sealed K / V pages ([kv head][page][13][64][16 B]), or null

// Per-row metadata of one forward pass.
struct RowMeta {
    int32_t seq;     // sequence slot (page-table row), -1 = inactive row
    int32_t pos;     // position of this token = its index among the sequence's real tokens
    int32_t last;    // 1 if logits are wanted for this row
    int32_t pad;
};

// ---- small-batch decode path (1..SMALL_RP dialogues) --------------------------------------------------------------
// With a handful of rows the activation is a few KiB, so the work of the small kernels between two GEMMs (split-K
// reduce + residual + RMSNorm; the sum of the P.V chunk partials) is cheap enough to be redone by every GEMM block as
// a prologue that leaves the MFMA B operand in LDS: three launches per layer disappear (gemm.hip: gemv_small_kernel).
#define SMALL_RP 4
enum { PRO_NONE = 0, PRO_NORM = 1, PRO_COMBINE = 2, PRO_ROWS = 3 };
struct SmallPro {
    int rows;                  // live rows (<= SMALL_RP); only these are stored
    // PRO_NORM: x' = bf16(x + bf16(sum of slabs)); xn = RMSNorm(x') * w   (resid_norm_kernel's arithmetic, same order)
    const uint16_t* x_in;      // [rows][H] residual stream
    uint16_t* x_out;           // x' goes here (the other ping-pong buffer; written by block (0,0)); may be null
    const float* slabs;        // [ksplit][MTTS_PFCAP][slab_npad] fp32 split-K partials of the previous Linear; ksplit 0: none
    int ksplit, slab_npad;
    const uint16_t* norm_w;
    float eps;
    // PRO_COMBINE: attention output = bf16(sum over pass-B chunks of the fp32 partials)   (attn_combine_kernel's arithmetic)
    const float* opart;        // [(row*nq + head)][nchunks_max][128]
    const RowMeta* meta;
    int nchunks_max, nq, pages_per_chunk;
    // PRO_ROWS: the activation is row-major bf16 [rows][K]
    const uint16_t* xrows;
};
