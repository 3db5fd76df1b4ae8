// Multi-codebook sampler and the delay-pattern state machine of the decode loop.
//
// sample_kernel   : reference modeling_asteroid.py:123-138 (logit masks, HF
//                   RepetitionPenalty/Temperature/TopK/TopP processors in that
//                   order, then argmax or a multinomial draw).
// update_kernel   : reference modeling_asteroid.py:139-169 (EOS flush via
//                   needs_additional_steps, teacher forcing of the delayed prompt
//                   tail, finished-row padding, stopping criteria).
//
// Draw definition (torch.multinomial's stream cannot be reproduced): Philox4x32-10,
// key = seed, counter = (step, row, channel, 0), u = (x0 >> 8) * 2^-24; kept tokens
// are walked in descending score and the first whose running sum of
// exp(score - max) exceeds u * total is taken.  oracle/asteroid_oracle.py
// (sample_from_scores) states the same rule.
#include "common.h"
#include "../../include/mtts.h"


struct SeqState {           // one per sequence slot, device resident.  Every dialogue carries its own clock so
                            // that slots can be refilled while others are mid-flight (continuous batching).
    int32_t nas;            // needs_additional_steps
    int32_t unfinished;
    int32_t kv_len;         // real tokens already in the KV cache
    int32_t step;           // decode steps this dialogue has run (= rows it generated)
    int32_t base_length;    // T-7 (padded slots) of its prompt
    int32_t max_length;     // HF max_length in padded slots
    int32_t row_id;         // Philox counter word 1 (batch row index in mtts_generate, 0 for scheduled dialogues)
    int32_t active;         // slot holds a dialogue that still steps
    uint64_t seed;          // Philox key
};

struct LoopState {          // one per engine, device resident
    int32_t step;           // decode steps executed so far by the engine
    int32_t done;           // no active row is unfinished
    int32_t continuous;     // 1: a finished row leaves the batch at once (scheduler); 0: it keeps emitting the
                            //    reference's finished-row padding until the whole batch is done (mtts_generate)
    int32_t B;
    int32_t error;          // sticky device-side error
    int32_t gen_cap;        // rows of generated-token storage per slot
    int32_t forced_draw;    // forced replay: 1 (2: cut-off rows too) = the forced row replaces the step's raw draw BEFORE the state machine
                            //    (replay of a sampled reference run); 0 = it replaces the state machine's output
    int32_t logits_f32;     // the logits buffers hold fp32 (MTTS_DTYPE_F32 engine) instead of bf16
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// order-preserving map fp32 -> uint32 (larger float = larger key)
__device__ __forceinline__ uint32_t fkey(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Processed score of token i (everything except top-k / top-p, which act on the set).
struct LogitsPtr { const void* p; int f32; };     // one row of logits: bf16 bit patterns, or fp32 in the fp32 engine
__device__ __forceinline__ float proc_score(LogitsPtr logits, int i, int mask_id,
                                            const uint32_t* __restrict__ bitmap, float penalty, float temperature) {
    if (i == mask_id) return -INFINITY;
    float s = logits.f32 ? ((const float*)logits.p)[i] : bf2f(((const uint16_t*)logits.p)[i]);
    if (penalty > 0.f && (bitmap[i >> 5] >> (i & 31)) & 1u) s = (s < 0.f) ? s * penalty : s / penalty;
    if (temperature > 0.f) s = s / temperature;
    return s;
}

// ---------------------------------------------------------------------------
// Sampler = three short kernels per step.
//   A sample_scan_kernel    (big vocab only, grid (NS,B)): per-slice argmax + 11-bit
//                            radix histogram of the processed scores (global, per row)
//   B sample_collect_kernel (big vocab only, grid (NS,B)): bin b0 that holds the k-th
//                            largest score; every token in bins >= b0 is appended to the
//                            row's candidate list (a superset of the top-k set)
//   C sample_final_kernel   (grid (8,B)): candidates -> LDS, bitonic sort by
//                            (score asc, id asc), exact top-k threshold, top-p cut on
//                            the ascending cumulative softmax, Philox draw.
// Channels 1..7 (1025 tokens) skip A/B: kernel C reads the logits straight into LDS.
// Draw order: kept tokens from the highest score down (ties: higher id first); the
// first whose running sum of exp(score - max) exceeds u * total wins.
// ---------------------------------------------------------------------------
#define SAMP_T 256
#define SAMP_CAND 4096
#define SAMP_NS 32

struct SampleScratch {
    uint32_t* hist;        // [32][2048]
    float* slice_val;      // [32][SAMP_NS]
    int32_t* slice_idx;    // [32][SAMP_NS]
    float* cand_val;       // [32][SAMP_CAND]
    int32_t* cand_idx;     // [32][SAMP_CAND]
    uint32_t* cand_n;      // [rows]
    int32_t* overflow;     // [rows] set by the final kernel when a row needs the full-vocabulary path
    float* full_val;       // [rows][full_cap]   full-vocabulary path: level-0 bin (uint16) of every token
    int32_t* full_idx;     // [rows][full_cap]   full-vocabulary path: key of every token
    uint32_t* nuc_cnt;     // [rows][2048] level-0 histogram (count) left by the collect kernel for the full-vocabulary kernel
    unsigned long long* nuc_mass;   // [rows][2048] ... and mass (exp(s - max) * 2^45, exact integer sums)
};

struct SampleCtx {         // resolved per (row, channel)
    LogitsPtr lg;
    const uint32_t* bm;
    int V, mask_id, step, c;
    float penalty, temp;
    uint64_t seed;
    uint32_t row_id;
};

__device__ __forceinline__ bool sample_ctx(SampleCtx& x, int b, int c_in, const uint16_t* logits0,
                                           const uint16_t* logits17, int V0, int Vs, int Vs_pad,
                                           const uint32_t* bitmaps, int bm_words, const MttsSamplerCfg* cfgs,
                                           const LoopState* ls, const SeqState* seqs, int single_vocab, int single_mask,
                                           int single_step, int single_channel) {
    if (single_vocab > 0) {            // unit-test entry: one logits matrix [rows][vocab]
        x.c = single_channel; x.step = single_step; x.V = single_vocab; x.mask_id = single_mask;
        x.seed = 0; x.row_id = (uint32_t)b;
        x.lg = LogitsPtr{logits0 + (size_t)b * x.V, 0};
        x.bm = bitmaps ? bitmaps + (size_t)b * bm_words : nullptr;
    } else {
        if (ls->done) return false;
        const SeqState sq = seqs[b];
        // finished rows get padding from the state machine; a row that finished AT max_length without a flush behind it
        // (active == 2) is still evaluated, as in the reference: see update_kernel
        if (!sq.active || !(sq.unfinished || sq.active == 2)) return false;
        x.c = c_in; x.step = sq.step;
        x.seed = sq.seed; x.row_id = (uint32_t)sq.row_id;
        x.V = (x.c == 0) ? V0 : Vs;
        {
            // channel-0 rows are padded to a multiple of 32 tokens (aligned rows: the head GEMM stores 8 bytes per lane)
            const size_t off = (x.c == 0) ? (size_t)b * ((V0 + 31) & ~31) : ((size_t)b * 7 + (x.c - 1)) * Vs_pad;
            const char* base = (const char*)((x.c == 0) ? logits0 : logits17);
            x.lg = LogitsPtr{base + off * (ls->logits_f32 ? 4 : 2), ls->logits_f32};
        }
        // modeling_asteroid.py:124-128 (hard-coded ids 1024 / 152694 as in the reference)
        x.mask_id = -1;
        if (x.c != 0 && x.step >= x.c) x.mask_id = 1024;
        if (x.c == 0 && x.step <= 6) x.mask_id = 152694;
        x.bm = bitmaps ? bitmaps + ((size_t)b * 8 + x.c) * bm_words : nullptr;
    }
    const MttsSamplerCfg cfg = cfgs[x.c];
    x.penalty = (x.bm && cfg.repetition_penalty > 0.f) ? cfg.repetition_penalty : 0.f;
    x.temp = cfg.temperature;
    return true;
}

__device__ __forceinline__ void argmax_merge(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

// block-wide argmax (lowest index wins ties); result valid in every thread
__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* shf, int* shi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
    const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shf[wid] = bv; shi[wid] = bi; }
    __syncthreads();
    bv = shf[0]; bi = shi[0];
    for (int w = 1; w < nw; ++w) argmax_merge(bv, bi, shf[w], shi[w]);
    __syncthreads();
}

__global__ __launch_bounds__(SAMP_T) void sample_scan_kernel(
    const uint16_t* __restrict__ logits0, int V0, const uint32_t* __restrict__ bitmaps, int bm_words,
    const MttsSamplerCfg* __restrict__ cfgs, const LoopState* __restrict__ ls, const SeqState* __restrict__ seqs,
    SampleScratch sc, int single_vocab,
    int single_mask, int single_step, int single_channel) {
    __shared__ uint32_t hist[2048];
    __shared__ float shf[SAMP_T / 64];
    __shared__ int shi[SAMP_T / 64];
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    SampleCtx x;
    if (!sample_ctx(x, b, 0, logits0, nullptr, V0, 0, 0, bitmaps, bm_words, cfgs, ls, seqs, single_vocab, single_mask,
                    single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    const bool want_hist = cfg.do_sample && cfg.top_k > 0 && cfg.top_k < x.V;
    for (int i = tid; i < 2048; i += SAMP_T) hist[i] = 0;
    __syncthreads();
    const int per = (x.V + SAMP_NS - 1) / SAMP_NS;
    const int i0 = slice * per, i1 = min(x.V, i0 + per);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = i0 + tid; i < i1; i += SAMP_T) {
        float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp);
        argmax_merge(bv, bi, s, i);
        if (want_hist) atomicAdd(&hist[fkey(s) >> 21], 1u);
    }
    block_argmax(bv, bi, shf, shi);
    if (tid == 0) { sc.slice_val[b * SAMP_NS + slice] = bv; sc.slice_idx[b * SAMP_NS + slice] = bi; }
    if (want_hist)
        for (int i = tid; i < 2048; i += SAMP_T)
            if (hist[i]) atomicAdd(&sc.hist[(size_t)b * 2048 + i], hist[i]);
}

// ---- helpers of the full-vocabulary path (sample_full_kernel below; its first pass runs inside sample_collect_kernel)
#define SAMP_FT 1024
#define NUC_BINS 2048
#define NUC_SCALE 35184372088832.0          // 2^45
typedef unsigned long long u64;

struct NucSel {            // membership of the current candidate set, in (key, id) order
    uint32_t kmin;         // survivors of top-k: key >= kmin (and finite)
    uint32_t kp;           // kept by top-p: key > kp, or key == kp and id >= idp
    int idp;
};
__device__ __forceinline__ float unfkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ bool nuc_in(const NucSel& z, uint32_t key, int id) {
    return key > 0x007fffffu /* fkey(-inf) */ && key >= z.kmin && (key > z.kp || (key == z.kp && id >= z.idp));
}
__device__ __forceinline__ int nuc_bin(uint32_t key, float lo) {
    const int b = (int)((unfkey(key) - lo) * ((float)(NUC_BINS - 1) / 32.0f));      // monotone in the score
    return min(max(b, 0), NUC_BINS - 1);
}

__device__ __forceinline__ u64 nuc_q(uint32_t key, float smax) { return (u64)((double)expf(unfkey(key) - smax) * NUC_SCALE); }

// Pass A for tokens i0, i0 + step, ... < i1 of one row: processed score -> key + level-0 bin (scratch, read by every later
// pass) and the level-0 histogram in LDS (cnt / mass: 2048 entries each, zeroed by the caller).
__device__ __forceinline__ void nuc_pass_a(const SampleCtx& x, float smax, float lo, uint32_t* __restrict__ keys,
                                           uint16_t* __restrict__ bins, unsigned int* cnt, u64* mass, int i0, int i1, int step) {
    for (int i = i0; i < i1; i += step) {
        const float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp) + 0.0f;      // (-0 -> +0: one key per value)
        const uint32_t key = fkey(s);
        keys[i] = key;
        if (s > -INFINITY) {
            const int d = nuc_bin(key, lo);
            bins[i] = (uint16_t)d;
            atomicAdd(&cnt[d], 1u);
            atomicAdd(&mass[d], nuc_q(key, smax));
        } else bins[i] = 0xffffu;
    }
}


// Bin of the k-th largest score in a 2048-bin histogram of fkey(score) >> 21 (LDS or global), SAMP_T threads:
// thread t owns bins 8t..8t+7; suffix sums over lanes and waves locate the bin.  Result in every thread.
__device__ __forceinline__ int topk_bin(const uint32_t* hist, uint32_t k, uint32_t* wsum, int* sh_b0) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t* h = hist + tid * 8;
    uint32_t cnt[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { cnt[j] = h[j]; mine += cnt[j]; }
    uint32_t suf = mine;                      // inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    if (lane == 0) wsum[wid] = suf;
    if (tid == 0) *sh_b0 = 0;
    __syncthreads();
    uint32_t above_waves = 0;
    for (int w = wid + 1; w < SAMP_T / 64; ++w) above_waves += wsum[w];
    const uint32_t above = above_waves + suf - mine;      // scores in bins strictly above this thread's
    if (above < k && above + mine >= k) {
        uint32_t run = above;
        for (int j = 7; j >= 0; --j) {
            run += cnt[j];
            if (run >= k) { *sh_b0 = tid * 8 + j; break; }
        }
    }
    __syncthreads();
    return *sh_b0;
}

__global__ __launch_bounds__(SAMP_T) void sample_collect_kernel(
    const uint16_t* __restrict__ logits0, int V0, const uint32_t* __restrict__ bitmaps, int bm_words,
    const MttsSamplerCfg* __restrict__ cfgs, const LoopState* __restrict__ ls, const SeqState* __restrict__ seqs,
    SampleScratch sc, int full_cap, int single_vocab,
    int single_mask, int single_step, int single_channel) {
    __shared__ uint32_t wsum[SAMP_T / 64];
    __shared__ int sh_b0;
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    SampleCtx x;
    if (!sample_ctx(x, b, 0, logits0, nullptr, V0, 0, 0, bitmaps, bm_words, cfgs, ls, seqs, single_vocab, single_mask,
                    single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    if (!cfg.do_sample) return;
    // no top-k (every finite score is a candidate) or a top_k beyond the candidate buffer: nothing to collect, the row
    // goes to the full-vocabulary kernel (the final kernel flags it when it sees more than SAMP_CAND candidates)
    if (!(cfg.top_k > 0 && cfg.top_k < x.V) || cfg.top_k > SAMP_CAND) {
        // ... whose first pass over the row (keys, level-0 bins, level-0 histogram) is done HERE, by the row's 32 blocks
        __shared__ unsigned int a_cnt[2048];
        __shared__ unsigned long long a_mass[2048];
        float smax = sc.slice_val[b * SAMP_NS + (lane & (SAMP_NS - 1))];
        smax = wave_max(smax);
        for (int i = tid; i < 2048; i += SAMP_T) { a_cnt[i] = 0; a_mass[i] = 0; }
        __syncthreads();
        const int per = (x.V + SAMP_NS - 1) / SAMP_NS;
        const int i0 = slice * per, i1 = min(x.V, i0 + per);
        nuc_pass_a(x, smax, smax - 32.0f, (uint32_t*)(sc.full_idx + (size_t)b * full_cap), (uint16_t*)(sc.full_val + (size_t)b * full_cap),
                   a_cnt, a_mass, i0 + tid, i1, SAMP_T);
        __syncthreads();
        for (int i = tid; i < 2048; i += SAMP_T)
            if (a_cnt[i]) { atomicAdd(&sc.nuc_cnt[(size_t)b * 2048 + i], a_cnt[i]); atomicAdd(&sc.nuc_mass[(size_t)b * 2048 + i], a_mass[i]); }
        if (slice == 0 && tid == 0) { sc.cand_n[b] = SAMP_CAND + 1; sc.overflow[b] = 2; }
        return;
    }
    const int b0 = topk_bin(sc.hist + (size_t)b * 2048, (uint32_t)cfg.top_k, wsum, &sh_b0);
    const uint32_t ninf_key = fkey(-INFINITY);
    const int per = (x.V + SAMP_NS - 1) / SAMP_NS;
    const int i0 = slice * per, i1 = min(x.V, i0 + per);
    for (int i = i0 + tid; i < i1; i += SAMP_T) {
        float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp);
        uint32_t key = fkey(s);
        if ((int)(key >> 21) >= b0 && key > ninf_key) {
            uint32_t slot = atomicAdd(&sc.cand_n[b], 1u);
            if (slot < SAMP_CAND) { sc.cand_val[(size_t)b * SAMP_CAND + slot] = s; sc.cand_idx[(size_t)b * SAMP_CAND + slot] = i; }
        }
    }
}

// block-wide exclusive prefix sum of one double per thread (NT threads); returns (exclusive, total).
// Sums run in fp64: a nucleus over the whole 152 k vocabulary has terms of 1e-5 of the total, and the draw must
// land on the same token as the oracle's fp64 cumulative sum.
template <int NT>
__device__ __forceinline__ double block_excl_scan(double v, double* sh, double& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) sh[wid] = inc;
    __syncthreads();
    double base = 0.0, tot = 0.0;
    for (int w = 0; w < NT / 64; ++w) { if (w < wid) base += sh[w]; tot += sh[w]; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

// Everything after the candidates are known: sort, HF top-k / top-p cuts, Philox draw.
// val/idx: n candidate (score, id) pairs (LDS for the fast path, global memory for the full-vocabulary path),
// capacity >= next_pow2(n).  NT threads.  Returns the chosen token in every thread.
template <int NT>
__device__ int finish_sample(float* __restrict__ cval, int* __restrict__ cidx, int n, const MttsSamplerCfg& cfg, float smax,
                             uint32_t step, uint32_t b, uint32_t c, uint64_t seed, double* shd, int* sh_i) {
    const int tid = threadIdx.x;
    // bitonic sort of P = next_pow2(n) slots by (score asc, id asc); padding sorts last
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += NT) { cval[i] = INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    for (int kk = 2; kk <= P; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P; t += NT) {
                int ixj = t ^ j;
                if (ixj > t) {
                    bool up = ((t & kk) == 0);
                    float a = cval[t], bb = cval[ixj];
                    int ai = cidx[t], bi2 = cidx[ixj];
                    bool gt = (a > bb) || (a == bb && ai > bi2);
                    if (gt == up) { cval[t] = bb; cval[ixj] = a; cidx[t] = bi2; cidx[ixj] = ai; }
                }
            }
            __syncthreads();
        }
    }
    // top-k (HF TopKLogitsWarper): drop scores < k-th largest; ties with the k-th stay
    int f0 = 0;
    if (cfg.top_k > 0 && cfg.top_k < n) {
        const float thr = cval[n - cfg.top_k];
        int cnt = 0;
        for (int i = tid; i < n; i += NT) cnt += (cval[i] < thr) ? 1 : 0;
        double tot;
        (void)block_excl_scan<NT>((double)cnt, shd, tot);
        f0 = (int)tot;
    }
    // top-p (HF TopPLogitsWarper): ascending cumulative softmax over the survivors; drop cum <= 1-p,
    // always keep the last (most probable) one.  Thread t owns a contiguous run of the kept range.
    const int m = n - f0;                                  // survivors of top-k
    const int per = (m + NT - 1) / NT;
    int first_keep = f0;
    if (cfg.top_p > 0.f && cfg.top_p < 1.0f) {
        const int a0 = f0 + tid * per, a1 = min(n, a0 + per);
        double loc = 0.0;
        for (int i = a0; i < a1; ++i) loc += (double)expf(cval[i] - smax);
        double tot;
        double base = block_excl_scan<NT>(loc, shd, tot);
        int drop = 0;
        double run = base;
        for (int i = a0; i < a1; ++i) {
            run += (double)expf(cval[i] - smax);
            if (i < n - 1 && (float)(run / tot) <= cfg.one_minus_top_p) drop++;
        }
        double dtot;
        (void)block_excl_scan<NT>((double)drop, shd, dtot);
        first_keep = f0 + (int)dtot;          // cum is monotone: the dropped ones are a prefix
    }
    // draw: walk kept tokens from the top (position n-1 down to first_keep)
    const int nk = n - first_keep;
    const int perk = (nk + NT - 1) / NT;
    const int r0 = tid * perk, r1 = min(nk, r0 + perk);    // ranks from the top
    double loc = 0.0;
    for (int r = r0; r < r1; ++r) loc += (double)expf(cval[n - 1 - r] - smax);
    double tot;
    double base = block_excl_scan<NT>(loc, shd, tot);
    uint32_t rnd[4];
    philox4x32_10(step, b, c, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
    const double u = (double)((float)(rnd[0] >> 8) * (1.0f / 16777216.0f));
    const double target = u * tot;
    if (tid == 0) sh_i[1] = 0x7fffffff;
    __syncthreads();
    double run = base;
    for (int r = r0; r < r1; ++r) {
        run += (double)expf(cval[n - 1 - r] - smax);
        if (run > target) { atomicMin(&sh_i[1], r); break; }
    }
    __syncthreads();
    int r = sh_i[1];
    if (r >= nk) r = nk - 1;                    // u*tot rounding: fall back to the last kept token
    return cidx[n - 1 - r];
}

__global__ __launch_bounds__(SAMP_T) void sample_final_kernel(
    const uint16_t* __restrict__ logits0, const uint16_t* __restrict__ logits17, int V0, int Vs, int Vs_pad,
    const uint32_t* __restrict__ bitmaps, int bm_words, const MttsSamplerCfg* __restrict__ cfgs,
    const LoopState* __restrict__ ls, const SeqState* __restrict__ seqs, uint64_t seed, int32_t* __restrict__ decisions,
    int32_t* __restrict__ err, SampleScratch sc, int big_channel0, int single_vocab, int single_mask, int single_step, int single_channel) {
    __shared__ float cval[SAMP_CAND];
    __shared__ int cidx[SAMP_CAND];
    __shared__ float shf[SAMP_T / 64];
    __shared__ double shd[SAMP_T / 64];
    __shared__ int shi[SAMP_T / 64];
    __shared__ int sh_i[4];
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wsum[SAMP_T / 64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    SampleCtx x;
    if (!sample_ctx(x, b, blockIdx.x, logits0, logits17, V0, Vs, Vs_pad, bitmaps, bm_words, cfgs, ls, seqs, single_vocab,
                    single_mask, single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    const bool big = single_vocab > 0 ? (single_vocab > SAMP_CAND) : (x.c == 0 && big_channel0);
    const int out_slot = b * 8 + x.c;
    int n = 0;
    float smax; int amax;
    if (big) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        if (tid < SAMP_NS) { bv = sc.slice_val[b * SAMP_NS + tid]; bi = sc.slice_idx[b * SAMP_NS + tid]; }
        block_argmax(bv, bi, shf, shi);
        smax = bv; amax = bi;
        if (cfg.do_sample) {
            n = (int)sc.cand_n[b];
            const int nn = min(n, SAMP_CAND);
            for (int i = tid; i < nn; i += SAMP_T) { cval[i] = sc.cand_val[(size_t)b * SAMP_CAND + i]; cidx[i] = sc.cand_idx[(size_t)b * SAMP_CAND + i]; }
        }
        __syncthreads();
        // reset the per-row scratch for the next step
        for (int i = tid; i < 2048; i += SAMP_T) sc.hist[(size_t)b * 2048 + i] = 0;
        if (tid == 0) sc.cand_n[b] = 0;
    } else {
        // small vocabulary (channels 1..7): scores straight from the logits.  With top_k set, an LDS histogram
        // first finds the radix bin of the k-th largest score and only that bin and the ones above it are kept
        // (a superset of the top-k set, typically ~2k entries): the sort below then runs on 64-128 slots, not 2048.
        float bv = -INFINITY; int bi = 0x7fffffff;
        const bool prefilter = cfg.do_sample && cfg.top_k > 0 && cfg.top_k < x.V;
        if (tid == 0) sh_i[0] = 0;
        if (prefilter)
            for (int i = tid; i < 2048; i += SAMP_T) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < x.V; i += SAMP_T) {
            float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp);
            argmax_merge(bv, bi, s, i);
            if (prefilter) atomicAdd(&hist[fkey(s) >> 21], 1u);
            else if (cfg.do_sample && s > -INFINITY) {
                int slot = atomicAdd(&sh_i[0], 1);
                if (slot < SAMP_CAND) { cval[slot] = s; cidx[slot] = i; }
            }
        }
        if (prefilter) {
            __syncthreads();
            const int b0 = topk_bin(hist, (uint32_t)cfg.top_k, wsum, &sh_i[2]);
            const uint32_t ninf_key = fkey(-INFINITY);
            for (int i = tid; i < x.V; i += SAMP_T) {
                float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp);
                uint32_t key = fkey(s);
                if ((int)(key >> 21) >= b0 && key > ninf_key) {
                    int slot = atomicAdd(&sh_i[0], 1);
                    if (slot < SAMP_CAND) { cval[slot] = s; cidx[slot] = i; }
                }
            }
        }
        block_argmax(bv, bi, shf, shi);
        smax = bv; amax = bi;
        n = sh_i[0];
    }
    if (!cfg.do_sample) {
        if (tid == 0) decisions[out_slot] = amax;
        return;
    }
    if (n > SAMP_CAND || n == 0) {               // too many candidates: hand the row to the full-vocabulary kernel
        if (tid == 0) { decisions[out_slot] = amax; if (n > SAMP_CAND && sc.overflow[b] == 0) sc.overflow[b] = 1; }
        return;
    }
    const int pick = finish_sample<SAMP_T>(cval, cidx, n, cfg, smax, (uint32_t)x.step, x.row_id, (uint32_t)x.c,
                                           single_vocab > 0 ? seed : x.seed, shd, sh_i);
    if (tid == 0) decisions[out_slot] = pick;
}

// ---------------------------------------------------------------------------------------------------------------
// Full-vocabulary path (sampling without top_k, or a top_k beyond the candidate buffer): the same HF rules and the same
// draw as finish_sample, WITHOUT sorting the row.  In the order (score asc, id asc) every quantity the rules need is
// a position found by counting or by mass: the k-th largest score (top-k threshold), the longest prefix whose
// cumulative softmax mass is <= 1 - top_p, and the token at which the running mass from the top exceeds u * total.
// Each is located by radix selection: a 2048-bin histogram (count + mass) over a linear binning of the score, then the
// three digits (11 | 11 | 10 bits) of the order-preserving key inside the crossing bin, then -- among tokens of EQUAL
// score -- two 9-bit digits of the token id.  One 1024-thread block per row, ~11 passes over the row's keys (L2
// resident), against the 171 global-memory bitonic passes over 2^18 slots this replaces (17 ms -> tens of us).
// Masses are exact integers (exp(s - max) * 2^45, uint64): sums do not depend on the order of the LDS atomics, so the
// pick is reproducible; against the fp64 sums of the in-block path / the oracle they differ by < 1e-8 of the total.
// ---------------------------------------------------------------------------------------------------------------
struct NucShared {
    unsigned int cnt0[NUC_BINS];      // level-0 histogram (linear score bins) of the CURRENT candidate set
    u64 mass0[NUC_BINS];
    unsigned int cnt[NUC_BINS];       // digit histograms of the passes below
    u64 mass[NUC_BINS];
    u64 wm[SAMP_FT / 64];
    unsigned int wc[SAMP_FT / 64];
    int cross;
    unsigned int out_cnt, grp_cnt;
    u64 out_mass, grp_mass;
};
// One digit pass over the row, restricted to the tokens of level-0 bin `binA` that are members of `z`.
// level 1..3: key digits (higher digits == `prefix`); 4, 5: id digits (9 bits each) of the tokens whose key == `prefix`.
__device__ void nuc_hist(NucShared& S, const uint32_t* __restrict__ keys, const uint16_t* __restrict__ bins, int V, const NucSel& z,
                         float smax, int level, int binA, uint32_t prefix, int idhi) {
    for (int i = threadIdx.x; i < NUC_BINS; i += SAMP_FT) { S.cnt[i] = 0; S.mass[i] = 0; }
    __syncthreads();
    // The bins are read 8 per 16-byte load, four loads in flight per thread (a load-test-loop pays one L2 round trip per
    // token); almost every group holds no token of bin `binA` and leaves after one packed compare.
    const int G = (V + 7) >> 3;                           // (bins[V .. 8G) hold 0xffff: sample_full_kernel pads them)
    const u32x4_t* bv = (const u32x4_t*)bins;
    const uint32_t pat = (uint32_t)binA * 0x00010001u;
    auto member = [&](int i) {
        const uint32_t key = keys[i];
        if (!nuc_in(z, key, i)) return;
        int d;
        if (level == 1) d = key >> 21;
        else if (level == 2) { if ((key >> 21) != prefix) return; d = (key >> 10) & 2047; }
        else if (level == 3) { if ((key >> 10) != prefix) return; d = key & 1023; }
        else {
            if (key != prefix) return;
            if (level == 4) d = i >> 9;
            else { if ((i >> 9) != idhi) return; d = i & 511; }
        }
        atomicAdd(&S.cnt[d], 1u);
        if (level <= 3) atomicAdd(&S.mass[d], nuc_q(key, smax));
    };
    for (int g0 = threadIdx.x; g0 < G; g0 += SAMP_FT * 4) {
        u32x4_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * SAMP_FT;
            w[u] = g < G ? bv[g] : u32x4_t{~0u, ~0u, ~0u, ~0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + u * SAMP_FT;
            const uint32_t ww[4] = {w[u].x ^ pat, w[u].y ^ pat, w[u].z ^ pat, w[u].w ^ pat};
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) any |= (ww[k] - 0x00010001u) & ~ww[k] & 0x80008000u;    // a zero half-word somewhere?
            if (!any) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((ww[k] & 0xffffu) == 0) member(8 * g + 2 * k);
                if ((ww[k] >> 16) == 0) member(8 * g + 2 * k + 1);
            }
        }
    }
    __syncthreads();
}

// Walk the bins of (cnt, mass) in ascending (asc) or descending order and find the first one at which `crossed(count so far
// incl. the bin, mass so far incl. the bin)` holds (it is monotone).  `bc` / `bm` = count / mass in front of the walk; on
// return they hold the totals in front of the crossing bin, S.grp_cnt / S.grp_mass that bin's own content.  Returns the
// bin, or -1 (then S.wc / S.wm hold the per-wave totals of the walk).
template <typename F>
__device__ int nuc_cross(NucShared& S, const unsigned int* cnt, const u64* mass, bool asc, unsigned int& bc, u64& bm, F crossed) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = 2 * tid, b0 = asc ? p0 : NUC_BINS - 1 - p0, b1 = asc ? p0 + 1 : NUC_BINS - 2 - p0;
    const unsigned int c0 = cnt[b0], c1 = cnt[b1];
    const u64 m0 = mass[b0], m1 = mass[b1];
    unsigned int ic = c0 + c1;
    u64 im = m0 + m1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int tc = __shfl_up(ic, o, 64);
        const u64 tm = __shfl_up(im, o, 64);
        if (lane >= o) { ic += tc; im += tm; }
    }
    if (tid == 0) S.cross = 0x7fffffff;
    __syncthreads();
    if (lane == 63) { S.wc[wid] = ic; S.wm[wid] = im; }
    __syncthreads();
    unsigned int ec = bc + ic - (c0 + c1);
    u64 em = bm + im - (m0 + m1);
    for (int w = 0; w < wid; ++w) { ec += S.wc[w]; em += S.wm[w]; }
    const bool x0 = (c0 > 0) && crossed(ec + c0, em + m0);
    const bool x1 = (c1 > 0) && crossed(ec + c0 + c1, em + m0 + m1);
    if (x0) atomicMin(&S.cross, p0);
    else if (x1) atomicMin(&S.cross, p0 + 1);
    __syncthreads();
    const int px = S.cross;
    if (px == p0) { S.out_cnt = ec; S.out_mass = em; S.grp_cnt = c0; S.grp_mass = m0; }
    if (px == p0 + 1) { S.out_cnt = ec + c0; S.out_mass = em + m0; S.grp_cnt = c1; S.grp_mass = m1; }
    __syncthreads();
    if (px == 0x7fffffff) return -1;
    bc = S.out_cnt;
    bm = S.out_mass;
    return asc ? px : NUC_BINS - 1 - px;
}

// Locate the crossing of `crossed` down to one key value.  false: the walk never crosses.  Else `key` = the score group (all
// member tokens of that key), bc / bm = what lies in front of the group, gc / gm = the group's count / mass; binA = its level-0
// bin, c_bin / m_bin = what lies in front of that bin.
template <typename F>
__device__ bool nuc_select_key(NucShared& S, const uint32_t* keys, const uint16_t* bins, int V, const NucSel& z, float smax, bool asc,
                               unsigned int& bc, u64& bm, F crossed, uint32_t& key, unsigned int& gc, u64& gm, int& binA,
                               unsigned int& c_bin, u64& m_bin) {
    binA = nuc_cross(S, S.cnt0, S.mass0, asc, bc, bm, crossed);
    if (binA < 0) return false;
    c_bin = bc;
    m_bin = bm;
    nuc_hist(S, keys, bins, V, z, smax, 1, binA, 0, 0);
    const int d1 = nuc_cross(S, S.cnt, S.mass, asc, bc, bm, crossed);
    nuc_hist(S, keys, bins, V, z, smax, 2, binA, (uint32_t)d1, 0);
    const int d2 = nuc_cross(S, S.cnt, S.mass, asc, bc, bm, crossed);
    nuc_hist(S, keys, bins, V, z, smax, 3, binA, ((uint32_t)d1 << 11) | (uint32_t)d2, 0);
    const int d3 = nuc_cross(S, S.cnt, S.mass, asc, bc, bm, crossed);
    key = ((uint32_t)d1 << 21) | ((uint32_t)d2 << 10) | (uint32_t)d3;
    gc = S.grp_cnt;
    gm = S.grp_mass;
    return true;
}

// Among the member tokens of key `key` (level-0 bin binA): the id at rank `rank` (0-based) in ascending / descending id order.
__device__ int nuc_select_id(NucShared& S, const uint32_t* keys, const uint16_t* bins, int V, const NucSel& z, float smax, int binA,
                             uint32_t key, bool asc, unsigned int rank) {
    unsigned int bc = 0;
    u64 bm = 0;
    auto by_rank = [rank](unsigned int c, u64) { return c > rank; };
    nuc_hist(S, keys, bins, V, z, smax, 4, binA, key, 0);
    const int hi = nuc_cross(S, S.cnt, S.mass, asc, bc, bm, by_rank);
    nuc_hist(S, keys, bins, V, z, smax, 5, binA, key, hi);
    const int low = nuc_cross(S, S.cnt, S.mass, asc, bc, bm, by_rank);
    return (hi << 9) | low;
}

// The level-0 histogram after `dc` tokens / `dm` mass have left from the bottom of the order, the last of them inside bin
// binA (c_bin / m_bin = what lay in front of that bin).
__device__ void nuc_drop_prefix(NucShared& S, int binA, unsigned int dc, u64 dm, unsigned int c_bin, u64 m_bin) {
    __syncthreads();
    for (int i = threadIdx.x; i < binA; i += SAMP_FT) { S.cnt0[i] = 0; S.mass0[i] = 0; }
    if (threadIdx.x == 0) { S.cnt0[binA] -= dc - c_bin; S.mass0[binA] -= dm - m_bin; }
    __syncthreads();
}

__global__ __launch_bounds__(SAMP_FT) void sample_full_kernel(
    const uint16_t* __restrict__ logits0, int V0, const uint32_t* __restrict__ bitmaps, int bm_words,
    const MttsSamplerCfg* __restrict__ cfgs, const LoopState* __restrict__ ls, const SeqState* __restrict__ seqs,
    uint64_t seed, int32_t* __restrict__ decisions, SampleScratch sc, int full_cap, int single_vocab, int single_mask, int single_step,
    int single_channel) {
    __shared__ NucShared S;
    __shared__ float shf[SAMP_FT / 64];
    __shared__ int shi[SAMP_FT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int flag = sc.overflow[b];
    if (!flag) return;
    SampleCtx x;
    if (!sample_ctx(x, b, 0, logits0, nullptr, V0, 0, 0, bitmaps, bm_words, cfgs, ls, seqs, single_vocab, single_mask,
                    single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    uint16_t* bins = (uint16_t*)(sc.full_val + (size_t)b * full_cap);  // level-0 bin of every token (0xffff: -inf)
    uint32_t* keys = (uint32_t*)(sc.full_idx + (size_t)b * full_cap);  // order-preserving key of every processed score
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (tid < SAMP_NS) { bv = sc.slice_val[b * SAMP_NS + tid]; bi = sc.slice_idx[b * SAMP_NS + tid]; }
    block_argmax(bv, bi, shf, shi);
    const float smax = bv;
    const float lo = smax - 32.0f;
    const int V = x.V;
    // ---- level-0 histogram: left by the collect kernel's 32 blocks per row (flag 2), or built here (a row that overflowed
    // the candidate buffer although its top_k fits: many tokens tie with the k-th score)
    if (flag == 2) {
        unsigned int* gc_ = sc.nuc_cnt + (size_t)b * NUC_BINS;
        u64* gm_ = sc.nuc_mass + (size_t)b * NUC_BINS;
        for (int i = tid; i < NUC_BINS; i += SAMP_FT) { S.cnt0[i] = gc_[i]; S.mass0[i] = gm_[i]; gc_[i] = 0; gm_[i] = 0; }
        __syncthreads();
    } else {
        for (int i = tid; i < NUC_BINS; i += SAMP_FT) { S.cnt0[i] = 0; S.mass0[i] = 0; }
        __syncthreads();
        nuc_pass_a(x, smax, lo, keys, bins, S.cnt0, S.mass0, tid, V, SAMP_FT);
        __syncthreads();
    }
    if (tid < 8 && V + tid < ((V + 7) & ~7)) bins[V + tid] = 0xffffu;   // the digit passes read the bins 8 at a time
    __syncthreads();
    unsigned int n = 0;
    u64 tot = 0;
    {
        unsigned int bc = 0; u64 bm = 0;
        auto never = [](unsigned int, u64) { return false; };
        (void)nuc_cross(S, S.cnt0, S.mass0, true, bc, bm, never);      // no crossing: the walk's per-wave sums are the totals
        for (int w = 0; w < SAMP_FT / 64; ++w) { n += S.wc[w]; tot += S.wm[w]; }
        __syncthreads();
    }
    int pick = bi;
    if (n > 0) {
        NucSel z{0u, 0u, 0};
        int binA; unsigned int c_bin; u64 m_bin;
        // ---- top-k (HF TopKLogitsWarper): scores below the k-th largest go; ties with it stay
        if (cfg.top_k > 0 && (unsigned int)cfg.top_k < n) {
            const unsigned int idx0 = n - (unsigned int)cfg.top_k;       // ascending position of the k-th largest
            unsigned int bc = 0, gc; u64 bm = 0, gm; uint32_t kk;
            auto at = [idx0](unsigned int c, u64) { return c > idx0; };
            nuc_select_key(S, keys, bins, V, z, smax, true, bc, bm, at, kk, gc, gm, binA, c_bin, m_bin);
            z.kmin = kk;
            n -= bc;                                        // everything in front of the threshold's group is dropped
            tot -= bm;
            nuc_drop_prefix(S, binA, bc, bm, c_bin, m_bin);
        }
        // ---- top-p (HF TopPLogitsWarper): longest ascending prefix with cumulative softmax <= 1 - top_p goes
        if (cfg.top_p > 0.f && cfg.top_p < 1.0f) {
            const double dtot = (double)tot;
            const float omp = cfg.one_minus_top_p;
            auto beyond = [dtot, omp](unsigned int, u64 m) { return !((float)((double)m / dtot) <= omp); };
            unsigned int bc = 0, gc; u64 bm = 0, gm; uint32_t kp;
            if (nuc_select_key(S, keys, bins, V, z, smax, true, bc, bm, beyond, kp, gc, gm, binA, c_bin, m_bin)) {
                // tokens of equal score leave in ascending id order: t of the group's gc go (t < gc: the group crosses)
                const u64 q = gm / gc;
                unsigned int t = 0, hi = gc;                // largest t with cum(bm + t q) <= 1 - p
                while (t < hi) {
                    const unsigned int mid = (t + hi + 1) >> 1;
                    if (!beyond(0u, bm + (u64)mid * q)) t = mid; else hi = mid - 1;
                }
                int idp = 0;
                if (t > 0) idp = nuc_select_id(S, keys, bins, V, z, smax, binA, kp, true, t);      // the (t+1)-th smallest id stays
                z.kp = kp;
                z.idp = idp;
                n -= bc + t;
                tot -= bm + (u64)t * q;
                nuc_drop_prefix(S, binA, bc + t, bm + (u64)t * q, c_bin, m_bin);
            }
        }
        // ---- draw: walk the kept tokens from the top; the first whose running mass exceeds u * total
        uint32_t rnd[4];
        const uint64_t sd = single_vocab > 0 ? seed : x.seed;
        philox4x32_10((uint32_t)x.step, x.row_id, (uint32_t)x.c, 0u, (uint32_t)sd, (uint32_t)(sd >> 32), rnd);
        const double u = (double)((float)(rnd[0] >> 8) * (1.0f / 16777216.0f));
        const double target = u * ((double)tot / NUC_SCALE);
        auto past = [target](unsigned int, u64 m) { return (double)m / NUC_SCALE > target; };
        unsigned int bc = 0, gc; u64 bm = 0, gm; uint32_t kd;
        if (nuc_select_key(S, keys, bins, V, z, smax, false, bc, bm, past, kd, gc, gm, binA, c_bin, m_bin)) {
            const u64 q = gm / gc;
            unsigned int j = 1, hi = gc;                    // smallest j with mass(bm + j q) past the target (ties: higher id first)
            while (j < hi) {
                const unsigned int mid = (j + hi) >> 1;
                if (past(0u, bm + (u64)mid * q)) hi = mid; else j = mid + 1;
            }
            pick = nuc_select_id(S, keys, bins, V, z, smax, binA, kd, false, j - 1);
        }
    }
    if (tid == 0) { decisions[b * 8 + x.c] = pick; sc.overflow[b] = 0; }
}

// One block; thread b handles sequence slot b.  Restates modeling_asteroid.py:139-169 with a per-dialogue clock.
// gen / dec_log / forced: [slot][gen_cap][8]; tf_tail: [slot][7][8] = tf_inputs[:, base_length + s, :].
__global__ void update_kernel(const int32_t* __restrict__ decisions, int32_t* __restrict__ dec_log,
                              const int32_t* __restrict__ forced, const int32_t* __restrict__ tf_tail,
                              int32_t* __restrict__ gen, int32_t* __restrict__ cur_tokens,
                              SeqState* __restrict__ seqs, RowMeta* __restrict__ meta, uint32_t* __restrict__ bitmaps,
                              int bm_words, LoopState* __restrict__ ls, int eos, int spad, int sp_lo, int sp_hi) {
    __shared__ int any_unfinished;
    if (ls->done) return;
    const int b = threadIdx.x, B = ls->B;
    const int cap = ls->gen_cap;
    if (b == 0) any_unfinished = 0;
    __syncthreads();
    if (b < B) {
        SeqState s = seqs[b];
        if (!s.active || s.step >= cap) {
            meta[b].seq = -1;
            if (s.active && s.step >= cap) { s.active = 0; s.unfinished = 0; seqs[b] = s; }
        } else {
            const int step = s.step;
            int tok[8];
            // The reference evaluates EVERY row of a static batch at every step (finished ones too) and tests their
            // channel-0 pick for "not a speech token" (:140-141) before it overwrites the row with padding (:155-158).
            // A row that was finished by max_length (needs_additional_steps still -1) is thereby resurrected for a
            // 7-step flush as soon as its pick is a non-speech token (`unfinished | nas > 0`, :168), while the rest of
            // the batch is still running.  Such rows are kept in the forward pass (active == 2); rows finished by EOS
            // have nas == 0, can never come back, and are skipped for good.
            const bool linger = s.active == 2;
#pragma unroll
            for (int c = 0; c < 8; ++c) tok[c] = (s.unfinished || linger) ? decisions[b * 8 + c] : 0;
            const size_t slot = ((size_t)b * cap + step) * 8;
            const bool as_draw = forced && ls->forced_draw;
            if (as_draw) {          // the log keeps the raw draws; the reference's history drives the state machine
                // (mode 2, tests of chained resurrections: the forced row is also the raw draw of a cut-off row that
                //  the reference keeps evaluating -- a real reference run does not record those draws)
                const bool take = s.unfinished != 0 || (linger && ls->forced_draw == 2);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    dec_log[slot + c] = tok[c];
                    const int f = forced[slot + c];
                    if (f >= 0 && take) tok[c] = f;
                }
            }
            // :140-141
            const bool speech = tok[0] >= sp_lo && tok[0] < sp_hi;
            if ((s.unfinished || linger) && !speech && s.nas < 0) s.nas = 7;
            // :143-145 teacher forcing of the delayed prompt tail (first 7 steps)
            if (step < 7) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c >= step + 1) tok[c] = tf_tail[(b * 7 + step) * 8 + c];
            }
            // :148-153 EOS flush
            if (s.nas > 0 && s.nas < 7) {
                tok[0] = eos;
#pragma unroll
                for (int c = 1; c < 8; ++c)
                    if (s.nas < 8 - c) tok[c] = spad;
            }
            // :155-158 finished rows
            if (!s.unfinished) {
                tok[0] = eos;
#pragma unroll
                for (int c = 1; c < 8; ++c) tok[c] = spad;
            }
            if (dec_log && !as_draw) {
#pragma unroll
                for (int c = 0; c < 8; ++c) dec_log[slot + c] = tok[c];
            }
            if (forced && !as_draw) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    int f = forced[slot + c];
                    if (f >= 0) tok[c] = f;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                gen[slot + c] = tok[c];
                cur_tokens[b * 8 + c] = tok[c];
                uint32_t* bm = bitmaps + ((size_t)b * 8 + c) * bm_words;
                bm[tok[c] >> 5] |= 1u << (tok[c] & 31);       // history for the repetition penalty
            }
            // :165-168
            if (s.nas > 0) s.nas -= 1;
            const int new_len = s.base_length + step + 1;
            const bool stopping = (new_len >= s.max_length) || (tok[0] == eos) || (s.nas == 0);
            s.unfinished = (s.unfinished && !stopping) ? 1 : 0;
            if (s.nas > 0) s.unfinished = 1;
            s.step = step + 1;
            if (ls->continuous) {
                if (!s.unfinished) s.active = 0;                     // scheduler mode: the slot is free again
            } else {
                s.active = (!s.unfinished && s.nas < 0) ? 2 : 1;     // static batch: finished at max_length, may come back
            }
            // the forward that follows appends this token to the cache at position kv_len
            meta[b].seq = (s.unfinished || s.active == 2) ? b : -1;
            meta[b].pos = s.kv_len;
            meta[b].last = 1;
            s.kv_len += 1;
            seqs[b] = s;
            if (s.unfinished) atomicOr(&any_unfinished, 1);
        }
    }
    __syncthreads();
    if (b == 0) {
        ls->step = ls->step + 1;
        if (!any_unfinished) ls->done = 1;
    }
}

static SampleScratch g_dummy_scratch;
void launch_sample(const void* logits0, const void* logits17, int V0, int Vs, int Vs_pad, const uint32_t* bitmaps, int bm_words,
                   const MttsSamplerCfg* cfgs, const LoopState* ls, const SeqState* seqs, uint64_t seed, int32_t* decisions,
                   int32_t* err, int B, const SampleScratch& sc, int ch0_sampled, int full_cap, hipStream_t st) {
    const int big0 = V0 > SAMP_CAND ? 1 : 0;
    if (big0) {
        hipLaunchKernelGGL(sample_scan_kernel, dim3(SAMP_NS, B), dim3(SAMP_T), 0, st, (const uint16_t*)logits0, V0,
                           bitmaps, bm_words, cfgs, ls, seqs, sc, 0, 0, 0, 0);
        if (ch0_sampled)
            hipLaunchKernelGGL(sample_collect_kernel, dim3(SAMP_NS, B), dim3(SAMP_T), 0, st, (const uint16_t*)logits0, V0,
                               bitmaps, bm_words, cfgs, ls, seqs, sc, full_cap, 0, 0, 0, 0);
    }
    hipLaunchKernelGGL(sample_final_kernel, dim3(8, B), dim3(SAMP_T), 0, st, (const uint16_t*)logits0,
                       (const uint16_t*)logits17, V0, Vs, Vs_pad, bitmaps, bm_words, cfgs, ls, seqs, seed, decisions, err, sc,
                       big0, 0, 0, 0, 0);
    if (big0 && ch0_sampled)
        hipLaunchKernelGGL(sample_full_kernel, dim3(B), dim3(SAMP_FT), 0, st, (const uint16_t*)logits0, V0, bitmaps,
                           bm_words, cfgs, ls, seqs, seed, decisions, sc, full_cap, 0, 0, 0, 0);
}
void launch_sample_single(const void* logits, int rows, int vocab, const uint32_t* bitmap, int bm_words,
                          const MttsSamplerCfg* cfgs8, int mask_id, uint64_t seed, int step, int channel,
                          int32_t* decisions, int32_t* err, const SampleScratch& sc, int full_cap, hipStream_t st) {
    if (vocab > SAMP_CAND) {
        hipLaunchKernelGGL(sample_scan_kernel, dim3(SAMP_NS, rows), dim3(SAMP_T), 0, st, (const uint16_t*)logits, vocab,
                           bitmap, bm_words, cfgs8, (const LoopState*)nullptr, (const SeqState*)nullptr, sc, vocab, mask_id, step, channel);
        hipLaunchKernelGGL(sample_collect_kernel, dim3(SAMP_NS, rows), dim3(SAMP_T), 0, st, (const uint16_t*)logits, vocab,
                           bitmap, bm_words, cfgs8, (const LoopState*)nullptr, (const SeqState*)nullptr, sc, full_cap, vocab, mask_id, step, channel);
    }
    hipLaunchKernelGGL(sample_final_kernel, dim3(1, rows), dim3(SAMP_T), 0, st, (const uint16_t*)logits,
                       (const uint16_t*)nullptr, vocab, vocab, vocab, bitmap, bm_words, cfgs8, (const LoopState*)nullptr,
                       (const SeqState*)nullptr, seed, decisions, err, sc, 1, vocab, mask_id, step, channel);
    if (vocab > SAMP_CAND)
        hipLaunchKernelGGL(sample_full_kernel, dim3(rows), dim3(SAMP_FT), 0, st, (const uint16_t*)logits, vocab, bitmap,
                           bm_words, cfgs8, (const LoopState*)nullptr, (const SeqState*)nullptr, seed, decisions, sc, full_cap, vocab,
                           mask_id, step, channel);
}
void launch_update(const int32_t* decisions, int32_t* dec_log, const int32_t* forced, const int32_t* tf_tail,
                   int32_t* gen, int32_t* cur_tokens, SeqState* seqs, RowMeta* meta, uint32_t* bitmaps, int bm_words,
                   LoopState* ls, int eos, int spad, int sp_lo, int sp_hi, hipStream_t st) {
    hipLaunchKernelGGL(update_kernel, dim3(1), dim3(MTTS_RCAP), 0, st, decisions, dec_log, forced, tf_tail, gen,
                       cur_tokens, seqs, meta, bitmaps, bm_words, ls, eos, spad, sp_lo, sp_hi);
}

// Un-shift the delay pattern on the device (reference generation_utils.py:416-425):
// codes[c][b][f - first] = gen[f + c][b][c] (- speech offset on channel 0), frames first..first+n-1.
__global__ void export_codes_kernel(const int32_t* __restrict__ gen, int64_t* __restrict__ codes, int B, int first, int n,
                                    int speech_offset, int clamp_hi, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * B * n) return;
    const int t = i % n, b = (i / n) % B, c = i / (n * B);
    int v = gen[((size_t)b * cap + (first + t + c)) * 8 + c];
    if (c == 0) v -= speech_offset;
    v = min(max(v, 0), clamp_hi);          // flushed / padded frames carry 1024 or EOS: keep the gather in range
    codes[i] = v;
}
void launch_export_codes(const int32_t* gen, int64_t* codes, int B, int first, int n, int speech_offset, int clamp_hi,
                         int cap, hipStream_t st) {
    int total = 8 * B * n;
    hipLaunchKernelGGL(export_codes_kernel, dim3((total + 255) / 256), dim3(256), 0, st, gen, codes, B, first, n,
                       speech_offset, clamp_hi, cap);
}
