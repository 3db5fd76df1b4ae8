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
// are walked in ascending token id and the first whose running sum of
// exp(score - max) exceeds u * total is taken.  oracle/asteroid_oracle.py
// (sample_from_scores) states the same rule.
#include "common.h"
#include "../../include/mtts.h"

#define SAMP_THREADS 1024
#define SAMP_CAP 2048     // max candidates that survive top-k (ties included)

struct SeqState {           // one per sequence slot, device resident
    int32_t nas;            // needs_additional_steps
    int32_t unfinished;
    int32_t kv_len;         // real tokens already in the KV cache
    int32_t pad;
};

struct LoopState {          // one per engine, device resident (mirrored to pinned host memory)
    int32_t step;           // decode steps executed so far (= generated rows)
    int32_t done;           // all rows finished
    int32_t base_length;    // T-7 (padded slots)
    int32_t max_length;     // HF max_length in padded slots
    int32_t tf_len;         // T
    int32_t B;
    int32_t error;          // sticky device-side error (e.g. candidate overflow)
    int32_t pad;
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
__device__ __forceinline__ float proc_score(const uint16_t* __restrict__ logits, int i, int mask_id,
                                            const uint32_t* __restrict__ bitmap, float penalty, float temperature) {
    if (i == mask_id) return -INFINITY;
    float s = bf2f(logits[i]);
    if (penalty > 0.f && (bitmap[i >> 5] >> (i & 31)) & 1u) s = (s < 0.f) ? s * penalty : s / penalty;
    if (temperature > 0.f) s = s / temperature;
    return s;
}

// grid = (B, 8); block 1024.  One block handles one (row, channel).
__global__ __launch_bounds__(SAMP_THREADS) void sample_kernel(
    const uint16_t* __restrict__ logits0 /*[32][V0]*/, const uint16_t* __restrict__ logits17 /*[32][7][Vs_pad]*/,
    int V0, int Vs, int Vs_pad, const uint32_t* __restrict__ bitmaps /*[B][8][bm_words]*/, int bm_words,
    const MttsSamplerCfg* __restrict__ cfgs /*[8]*/, const LoopState* __restrict__ ls, uint64_t seed,
    int32_t* __restrict__ decisions /*[B][8]*/, int32_t* __restrict__ err, int single_vocab, int single_mask,
    int single_step, int single_channel) {
    __shared__ uint32_t hist[2048];
    __shared__ float cval[SAMP_CAP];
    __shared__ int cidx[SAMP_CAP];
    __shared__ float shf[SAMP_THREADS / 64];
    __shared__ int shi[SAMP_THREADS / 64];
    __shared__ uint32_t sh_u[4];
    __shared__ float sh_f[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    int c, step, V, mask_id;
    const uint16_t* lg;
    if (single_vocab > 0) {            // unit-test entry: one logits matrix [rows][vocab]
        c = single_channel; step = single_step; V = single_vocab; mask_id = single_mask;
        lg = logits0 + (size_t)b * V;
    } else {
        if (ls->done) return;
        c = blockIdx.y; step = ls->step;
        V = (c == 0) ? V0 : Vs;
        lg = (c == 0) ? logits0 + (size_t)b * V0 : logits17 + ((size_t)b * 7 + (c - 1)) * Vs_pad;
        // modeling_asteroid.py:124-128 (hard-coded ids 1024 / 152694 as in the reference)
        mask_id = -1;
        if (c != 0 && step >= c) mask_id = 1024;
        if (c == 0 && step <= 6) mask_id = 152694;
    }
    const MttsSamplerCfg cfg = cfgs[c];
    const uint32_t* bm = nullptr;
    if (bitmaps) bm = (single_vocab > 0) ? bitmaps + (size_t)b * bm_words : bitmaps + ((size_t)b * 8 + c) * bm_words;
    const float penalty = (bm && cfg.repetition_penalty > 0.f) ? cfg.repetition_penalty : 0.f;
    const float temp = cfg.temperature;

    // ---- pass 1: argmax (lowest index wins ties) --------------------------------
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = tid; i < V; i += SAMP_THREADS) {
        float s = proc_score(lg, i, mask_id, bm, penalty, temp);
        if (s > best || (s == best && i < besti)) { best = s; besti = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { shf[wid] = best; shi[wid] = besti; }
    __syncthreads();
    if (tid == 0) {
        float bv = shf[0]; int bi = shi[0];
        for (int w = 1; w < SAMP_THREADS / 64; ++w)
            if (shf[w] > bv || (shf[w] == bv && shi[w] < bi)) { bv = shf[w]; bi = shi[w]; }
        sh_f[0] = bv; shi[0] = bi;
    }
    __syncthreads();
    const float smax = sh_f[0];
    const int amax = shi[0];
    if (!cfg.do_sample) {
        if (tid == 0) decisions[b * 8 + c] = amax;
        return;
    }

    // ---- top-k threshold: exact k-th largest score by 3-pass radix select ----------
    uint32_t thr_key = 0;                        // keep everything by default
    int k = cfg.top_k;
    if (k > 0 && k < V) {
        uint32_t prefix = 0, pmask = 0;
        int remaining = k;
        const int shifts[3] = {21, 10, 0};
        const int bits[3] = {11, 11, 10};
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const int nb = 1 << bits[pass];
            for (int i = tid; i < 2048; i += SAMP_THREADS) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < V; i += SAMP_THREADS) {
                uint32_t key = fkey(proc_score(lg, i, mask_id, bm, penalty, temp));
                if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
            }
            __syncthreads();
            if (tid == 0) {                       // walk bins from the top
                int rem = remaining, bsel = 0;
                for (int bin = nb - 1; bin >= 0; --bin) {
                    int cnt = (int)hist[bin];
                    if (cnt >= rem) { bsel = bin; break; }
                    rem -= cnt;
                }
                sh_u[0] = (uint32_t)bsel;
                sh_u[1] = (uint32_t)rem;
            }
            __syncthreads();
            prefix |= sh_u[0] << shifts[pass];
            pmask |= (uint32_t)(nb - 1) << shifts[pass];
            remaining = (int)sh_u[1];
            __syncthreads();
        }
        thr_key = prefix;
    }

    // ---- collect survivors (score >= k-th value, not -inf) -------------------------
    if (tid == 0) sh_u[2] = 0;
    __syncthreads();
    const uint32_t ninf_key = fkey(-INFINITY);
    for (int i0 = 0; i0 < V; i0 += SAMP_THREADS) {
        int i = i0 + tid;
        bool keep = false;
        float s = 0.f;
        if (i < V) {
            s = proc_score(lg, i, mask_id, bm, penalty, temp);
            uint32_t key = fkey(s);
            keep = key >= thr_key && key > ninf_key;
        }
        if (keep) {
            uint32_t slot = atomicAdd(&sh_u[2], 1u);
            if (slot < SAMP_CAP) { cval[slot] = s; cidx[slot] = i; }
        }
    }
    __syncthreads();
    int n = (int)sh_u[2];
    if (n > SAMP_CAP) {                        // loud failure: flag + argmax so that the loop stays defined
        if (tid == 0) { atomicExch(err, 1); decisions[b * 8 + c] = amax; }
        return;
    }
    // sort survivors by (score asc, token id asc): bitonic over SAMP_CAP slots padded with +inf
    for (int i = n + tid; i < SAMP_CAP; i += SAMP_THREADS) { cval[i] = INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    for (int kk = 2; kk <= SAMP_CAP; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < SAMP_CAP; t += SAMP_THREADS) {
                int ixj = t ^ j;
                if (ixj > t) {
                    bool up = ((t & kk) == 0);
                    float a = cval[t], bb = cval[ixj];
                    int ai = cidx[t], bi = cidx[ixj];
                    bool gt = (a > bb) || (a == bb && ai > bi);
                    if (gt == up) { cval[t] = bb; cval[ixj] = a; cidx[t] = bi; cidx[ixj] = ai; }
                }
            }
            __syncthreads();
        }
    }
    // ---- top-p: ascending cumulative softmax, drop cum <= 1-p, keep the last one --------
    // (single thread walks <= SAMP_CAP sorted survivors: n is ~top_k in practice)
    if (tid == 0) {
        int first_keep = 0;
        if (cfg.top_p > 0.f && cfg.top_p < 1.0f) {
            float tot = 0.f;
            for (int i = 0; i < n; ++i) tot += expf(cval[i] - smax);
            float cum = 0.f;
            const float lim = cfg.one_minus_top_p;   // float32(1.0 - top_p) computed in double by the host, as HF does
            for (int i = 0; i < n - 1; ++i) {
                cum += expf(cval[i] - smax) / tot;
                if (cum <= lim) first_keep = i + 1; else break;
            }
        }
        sh_u[3] = (uint32_t)first_keep;
    }
    __syncthreads();
    const int first_keep = (int)sh_u[3];
    // ---- draw: inverse CDF over kept tokens in ascending token id ------------------------
    // re-sort kept survivors by token id (kept = [first_keep, n)): mark dropped ones +inf id
    for (int t = tid; t < SAMP_CAP; t += SAMP_THREADS) {
        if (t < first_keep || t >= n) { cidx[t] = 0x7fffffff; cval[t] = -INFINITY; }
    }
    __syncthreads();
    for (int kk = 2; kk <= SAMP_CAP; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < SAMP_CAP; t += SAMP_THREADS) {
                int ixj = t ^ j;
                if (ixj > t) {
                    bool up = ((t & kk) == 0);
                    int ai = cidx[t], bi = cidx[ixj];
                    if ((ai > bi) == up) {
                        float a = cval[t]; cval[t] = cval[ixj]; cval[ixj] = a;
                        cidx[t] = bi; cidx[ixj] = ai;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const int nk = n - first_keep;
        float kmax = -INFINITY;
        for (int i = 0; i < nk; ++i) kmax = fmaxf(kmax, cval[i]);
        double tot = 0.0;
        for (int i = 0; i < nk; ++i) tot += (double)expf(cval[i] - kmax);
        uint32_t rnd[4];
        philox4x32_10((uint32_t)step, (uint32_t)b, (uint32_t)c, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        const double u = (double)((float)(rnd[0] >> 8) * (1.0f / 16777216.0f));
        const double target = u * tot;
        double cum = 0.0;
        int pick = cidx[nk - 1];
        for (int i = 0; i < nk; ++i) {
            cum += (double)expf(cval[i] - kmax);
            if (cum > target) { pick = cidx[i]; break; }
        }
        decisions[b * 8 + c] = pick;
    }
}

// One block; thread b handles sequence b.  Restates modeling_asteroid.py:139-169.
// gen: [max_steps][32][8] generated rows; tf_tail: [32][7][8] last 7 prompt slots
// (tf_inputs[:, base_length + s, :]); forced: optional [max_steps][32][8] (-1 = none).
__global__ void update_kernel(const int32_t* __restrict__ decisions, int32_t* __restrict__ dec_log,
                              const int32_t* __restrict__ forced, const int32_t* __restrict__ tf_tail,
                              int32_t* __restrict__ gen, int32_t* __restrict__ cur_tokens,
                              SeqState* __restrict__ seqs, RowMeta* __restrict__ meta, uint32_t* __restrict__ bitmaps,
                              int bm_words, LoopState* __restrict__ ls, LoopState* __restrict__ host_ls,
                              int eos, int spad, int sp_lo, int sp_hi, int max_steps) {
    __shared__ int any_unfinished;
    if (ls->done) return;
    const int b = threadIdx.x, B = ls->B, step = ls->step;
    if (b == 0) any_unfinished = 0;
    __syncthreads();
    if (b < B) {
        SeqState s = seqs[b];
        int tok[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) tok[c] = decisions[b * 8 + c];
        // :140-141
        const bool speech = tok[0] >= sp_lo && tok[0] < sp_hi;
        if (!speech && s.nas < 0) s.nas = 7;
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
        if (dec_log) {
#pragma unroll
            for (int c = 0; c < 8; ++c) dec_log[((size_t)step * MTTS_MAXR + b) * 8 + c] = tok[c];
        }
        if (forced) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                int f = forced[((size_t)step * MTTS_MAXR + b) * 8 + c];
                if (f >= 0) tok[c] = f;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            gen[((size_t)step * MTTS_MAXR + b) * 8 + c] = tok[c];
            cur_tokens[b * 8 + c] = tok[c];
            uint32_t* bm = bitmaps + ((size_t)b * 8 + c) * bm_words;
            bm[tok[c] >> 5] |= 1u << (tok[c] & 31);       // history for the repetition penalty
        }
        // :165-168
        if (s.nas > 0) s.nas -= 1;
        const int new_len = ls->base_length + step + 1;
        const bool stopping = (new_len >= ls->max_length) || (tok[0] == eos) || (s.nas == 0);
        s.unfinished = (s.unfinished && !stopping) ? 1 : 0;
        if (s.nas > 0) s.unfinished = 1;
        // the forward that follows appends this token to the cache at position kv_len
        meta[b].seq = s.unfinished ? b : -1;
        meta[b].pos = s.kv_len;
        meta[b].last = 1;
        s.kv_len += 1;
        seqs[b] = s;
        if (s.unfinished) atomicOr(&any_unfinished, 1);
    }
    __syncthreads();
    if (b == 0) {
        ls->step = step + 1;
        if (!any_unfinished || step + 1 >= max_steps) ls->done = 1;
        if (host_ls) { host_ls->step = ls->step; host_ls->done = ls->done; host_ls->error = ls->error; }
    }
}

void launch_sample(const void* logits0, const void* logits17, int V0, int Vs, int Vs_pad, const uint32_t* bitmaps, int bm_words,
                   const MttsSamplerCfg* cfgs, const LoopState* ls, uint64_t seed, int32_t* decisions, int32_t* err,
                   int B, hipStream_t st) {
    hipLaunchKernelGGL(sample_kernel, dim3(B, 8), dim3(SAMP_THREADS), 0, st, (const uint16_t*)logits0,
                       (const uint16_t*)logits17, V0, Vs, Vs_pad, bitmaps, bm_words, cfgs, ls, seed, decisions, err, 0, 0, 0, 0);
}
void launch_sample_single(const void* logits, int rows, int vocab, const uint32_t* bitmap, int bm_words,
                          const MttsSamplerCfg* cfgs8, int mask_id, uint64_t seed, int step, int channel,
                          int32_t* decisions, int32_t* err, hipStream_t st) {
    hipLaunchKernelGGL(sample_kernel, dim3(rows, 1), dim3(SAMP_THREADS), 0, st, (const uint16_t*)logits,
                       (const uint16_t*)nullptr, vocab, vocab, vocab, bitmap, bm_words, cfgs8, (const LoopState*)nullptr, seed,
                       decisions, err, vocab, mask_id, step, channel);
}
void launch_update(const int32_t* decisions, int32_t* dec_log, const int32_t* forced, const int32_t* tf_tail,
                   int32_t* gen, int32_t* cur_tokens, SeqState* seqs, RowMeta* meta, uint32_t* bitmaps, int bm_words,
                   LoopState* ls, LoopState* host_ls, int eos, int spad, int sp_lo, int sp_hi, int max_steps,
                   hipStream_t st) {
    hipLaunchKernelGGL(update_kernel, dim3(1), dim3(MTTS_MAXR), 0, st, decisions, dec_log, forced, tf_tail, gen,
                       cur_tokens, seqs, meta, bitmaps, bm_words, ls, host_ls, eos, spad, sp_lo, sp_hi, max_steps);
}
