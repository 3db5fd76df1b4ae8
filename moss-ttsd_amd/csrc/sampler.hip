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
    float* full_val;       // [rows][full_cap]
    int32_t* full_idx;     // [rows][full_cap]
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
    SampleScratch sc, int single_vocab,
    int single_mask, int single_step, int single_channel) {
    __shared__ uint32_t wsum[SAMP_T / 64];
    __shared__ int sh_b0;
    const int b = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    SampleCtx x;
    if (!sample_ctx(x, b, 0, logits0, nullptr, V0, 0, 0, bitmaps, bm_words, cfgs, ls, seqs, single_vocab, single_mask,
                    single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    if (!cfg.do_sample) return;
    int b0 = 0;                                   // no top-k: every finite score is a candidate
    if (cfg.top_k > 0 && cfg.top_k < x.V) {
        b0 = topk_bin(sc.hist + (size_t)b * 2048, (uint32_t)cfg.top_k, wsum, &sh_b0);
    }
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
        if (tid == 0) { decisions[out_slot] = amax; if (n > SAMP_CAND) sc.overflow[b] = 1; }
        return;
    }
    const int pick = finish_sample<SAMP_T>(cval, cidx, n, cfg, smax, (uint32_t)x.step, x.row_id, (uint32_t)x.c,
                                           single_vocab > 0 ? seed : x.seed, shd, sh_i);
    if (tid == 0) decisions[out_slot] = pick;
}

// Full-vocabulary path (no top_k, or more than SAMP_CAND candidates): same rules on the whole row, sorted in
// global memory by one 1024-thread block per row.  Runs only for rows the final kernel flagged.
#define SAMP_FT 1024
__global__ __launch_bounds__(SAMP_FT) void sample_full_kernel(
    const uint16_t* __restrict__ logits0, int V0, const uint32_t* __restrict__ bitmaps, int bm_words,
    const MttsSamplerCfg* __restrict__ cfgs, const LoopState* __restrict__ ls, const SeqState* __restrict__ seqs,
    uint64_t seed, int32_t* __restrict__ decisions, SampleScratch sc, int full_cap, int single_vocab, int single_mask, int single_step,
    int single_channel) {
    __shared__ float shf[SAMP_FT / 64];
    __shared__ double shd[SAMP_FT / 64];
    __shared__ int shi[SAMP_FT / 64];
    __shared__ int sh_i[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!sc.overflow[b]) return;
    SampleCtx x;
    if (!sample_ctx(x, b, 0, logits0, nullptr, V0, 0, 0, bitmaps, bm_words, cfgs, ls, seqs, single_vocab, single_mask,
                    single_step, single_channel)) return;
    const MttsSamplerCfg cfg = cfgs[x.c];
    float* val = sc.full_val + (size_t)b * full_cap;
    int* idx = sc.full_idx + (size_t)b * full_cap;
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (tid < SAMP_NS) { bv = sc.slice_val[b * SAMP_NS + tid]; bi = sc.slice_idx[b * SAMP_NS + tid]; }
    block_argmax(bv, bi, shf, shi);
    const float smax = bv;
    // compact the finite scores (ascending id per thread chunk; order is fixed by the sort anyway)
    if (tid == 0) sh_i[0] = 0;
    __syncthreads();
    for (int i = tid; i < x.V; i += SAMP_FT) {
        const float s = proc_score(x.lg, i, x.mask_id, x.bm, x.penalty, x.temp);
        if (s > -INFINITY) { const int slot = atomicAdd(&sh_i[0], 1); val[slot] = s; idx[slot] = i; }
    }
    __syncthreads();
    const int n = sh_i[0];
    __syncthreads();
    int pick = bi;
    if (n > 0) pick = finish_sample<SAMP_FT>(val, idx, n, cfg, smax, (uint32_t)x.step, x.row_id, (uint32_t)x.c,
                                            single_vocab > 0 ? seed : x.seed, shd, sh_i);
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
                               bitmaps, bm_words, cfgs, ls, seqs, sc, 0, 0, 0, 0);
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
                           bitmap, bm_words, cfgs8, (const LoopState*)nullptr, (const SeqState*)nullptr, sc, vocab, mask_id, step, channel);
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
