/*
 * mtts.h -- C ABI of libmtts.so, the MI355X-native engine under the
 * MOSS-TTSD hot path (AsteroidTTSInstruct decode + XY_Tokenizer decode).
 *
 * The reference has no FFI of its own: its boundary is the Python surface
 *   model.generate(input_ids[B,T,8], attention_mask[B,T])   generation_utils.py:406-409
 *   CustomMixin._sample                                      modeling_asteroid.py:53-197
 *   AsteroidTTSInstruct.forward (inference branch)           modeling_asteroid.py:337-380,411-426
 *   spt.decode(codes_list)                                   XY_Tokenizer/xy_tokenizer/model.py:195-256
 * Each entry point below names the reference code it replaces.  The Python
 * mirror of that surface (moss-ttsd_amd/modeling_asteroid.py etc.) binds these
 * symbols with ctypes; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions: extern "C"; plain pointers and sizes; every `dev_` pointer is
 * device memory of the engine's GPU (caller-owned: torch tensors); `host_`
 * pointers are host memory; `stream` is a hipStream_t passed as void* (NULL =
 * default stream).  Every function returns 0 on success or a negative
 * MTTS_E* code; mtts_last_error() gives the message of the calling thread's
 * last failure.  One engine = one device; an engine is not thread-safe.
 * No C++ exception crosses this boundary.
 */
#ifndef MTTS_H
#define MTTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTTS_OK 0
#define MTTS_EINVAL (-1)      /* bad argument / unsupported shape */
#define MTTS_EHIP (-2)        /* HIP runtime error */
#define MTTS_ENOMEM (-3)      /* KV page pool or workspace exhausted */
#define MTTS_ESTATE (-4)      /* call sequence error (e.g. decode before prefill) */

#define MTTS_CHANNELS 8

/* Fields of AsteroidTTSConfig / Qwen3Config the path reads
 * (modeling_asteroid.py:17-28; transformers Qwen3Config). */
typedef struct MttsConfig {
    int32_t vocab_size;            /* channel-0 vocabulary */
    int32_t hidden_size;
    int32_t intermediate_size;
    int32_t num_hidden_layers;
    int32_t num_attention_heads;
    int32_t num_key_value_heads;
    int32_t head_dim;              /* must be 128 */
    int32_t channels;              /* must be 8 */
    int32_t speech_vocab_size;     /* 1025 */
    int32_t speech_pad_token;      /* 1024 */
    int32_t speech_range_lo;       /* speech_token_range[0] */
    int32_t speech_range_hi;       /* speech_token_range[1] */
    int32_t eos_token_id;
    int32_t max_position;          /* rows in the RoPE table */
    float rms_norm_eps;
    int32_t max_batch;             /* sequences resident at once (1..128: up to 4 MFMA row tiles share each weight stream) */
    int32_t max_seq_len;           /* real tokens one sequence may reach (width of a page-table row) */
    int32_t kv_pool_pages;         /* 64-token KV pages in the pool shared by all sequences; 0 = max_batch x pages(max_seq_len),
                                      i.e. every slot can reach max_seq_len at once.  A smaller pool serves short dialogues
                                      with less memory: pages are taken on demand and returned when a dialogue finishes */
    int32_t dtype;                 /* MTTS_DTYPE_BF16 (0, the reference default) or MTTS_DTYPE_F32 (`inference.py --dtype fp32`,
                                      inference.py:27-40): fp32 weights, arithmetic, K/V pages and logits -- the strict-parity mode */
} MttsConfig;
#define MTTS_DTYPE_BF16 0
#define MTTS_DTYPE_F32 1
#define MTTS_DTYPE_F16 2   /* `--dtype fp16`: the MTTS_DTYPE_F32 kernels with an fp16 rounding point wherever the reference's fp16
                              run materialises a tensor; weights / RoPE tables are bound as fp32 tensors holding fp16 values */

/* generation_config.layers[i] / do_samples[i] (modeling_asteroid.py:95-106).
 * A field <= 0 (or top_k == 0) means "processor absent". */
typedef struct MttsSamplerCfg {
    int32_t do_sample;
    int32_t top_k;
    float top_p;
    float one_minus_top_p;   /* (float)(1.0 - (double)top_p): HF compares cum <= 1 - top_p with the scalar cast to fp32 */
    float temperature;
    float repetition_penalty;
} MttsSamplerCfg;

typedef struct MttsEngine MttsEngine;

const char* mtts_last_error(void);
int32_t mtts_version(void);

/* ---- engine lifetime ---------------------------------------------------- */
int32_t mtts_engine_create(const MttsConfig* cfg, int32_t device, MttsEngine** out);
int32_t mtts_engine_destroy(MttsEngine* e);

/* Weight binding: state-dict name (reference naming, e.g.
 * "model.language_model.layers.3.self_attn.q_proj.weight",
 * "model.embedding_list.0.weight", "model.language_model.norm.weight") ->
 * bf16 row-major device tensor.  The engine re-lays the matrix out into its
 * MFMA-fragment order in its own memory; the caller's tensor may be freed
 * after the call returns (stream-ordered).  Replaces from_pretrained's
 * parameter materialisation (generation_utils.py:18). */
int32_t mtts_bind_weight(MttsEngine* e, const char* name, const void* dev_bf16,
                         int64_t rows, int64_t cols, void* stream);
/* (an MTTS_DTYPE_F32 engine takes fp32 row-major tensors through the same calls: dev_bf16 / dev_cos_bf16 / dev_sin_bf16
 * then point to floats, and the engine keeps plain copies) */
/* RoPE table cos|sin, bf16 [max_position][64] each, computed by the host the
 * way Qwen3RotaryEmbedding does (fp32 -> bf16). */
int32_t mtts_bind_rope(MttsEngine* e, const void* dev_cos_bf16, const void* dev_sin_bf16,
                       int32_t rows, void* stream);
int32_t mtts_weights_ready(MttsEngine* e);   /* 0 when every tensor is bound */

/* ---- generation (replaces model.generate -> CustomMixin._sample) --------- */
/* host_input_ids int64 [B,T,8], host_attention_mask uint8 [B,T] (left padded,
 * 1 = real).  max_length = HF generation_config.max_length (T + max_new_tokens).
 * sampler[8], seed: sampling stream (Philox4x32-10).
 * Prefills the first T-7 slots, then runs the decode loop on the device until
 * every row is finished.  out: host_out_ids int64 [B, out_capacity, 8] receives
 * [B, T-7+G, 8]; *out_len = T-7+G.  G can exceed max_length - (T-7), exactly as in the reference
 * (modeling_asteroid.py:140-141,165-168): a dialogue whose EOS falls within 7 steps of max_length still runs its
 * delay-pattern flush, and while it does, a row that was cut off by max_length is resurrected for a flush of its own
 * as soon as its channel-0 pick is not a speech token (up to 14 steps past max_length); resurrections can chain, each
 * further row adding up to 6 steps, so size out_capacity as max_length + 6 * B + 8.  The engine has room for the whole
 * chain where max_seq_len allows and for 14 steps at least; a chain that outruns the room returns MTTS_ESTATE.
 * host_forced (verification hook, may be NULL): int64 [B, forced_len, 8] full
 * sequences; when given, every step's own decision is written to
 * host_decisions int64 [G,B,8] and the forced row is appended instead. */
int32_t mtts_generate(MttsEngine* e, const int64_t* host_input_ids, const uint8_t* host_attention_mask,
                      int32_t B, int32_t T, int32_t max_length,
                      const MttsSamplerCfg* sampler, uint64_t seed,
                      int64_t* host_out_ids, int32_t out_capacity, int32_t* out_len,
                      const int64_t* host_forced, int32_t forced_len, int64_t* host_decisions,
                      void* stream);

/* Step-level API (bench + tests): same state as mtts_generate, driven by the host. */
int32_t mtts_begin(MttsEngine* e, const int64_t* host_input_ids, const uint8_t* host_attention_mask,
                   int32_t B, int32_t T, int32_t max_length,
                   const MttsSamplerCfg* sampler, uint64_t seed, void* stream);   /* prefill */
int32_t mtts_step(MttsEngine* e, int32_t n_steps, void* stream);   /* n x (sample+update, forward); async */
int32_t mtts_sync_state(MttsEngine* e, int32_t* steps_done, int32_t* all_finished, void* stream);
int32_t mtts_read_generated(MttsEngine* e, int64_t* host_gen, int32_t capacity_steps, int32_t* n_steps);
/* last forward's logits: bf16 bits, channel 0 [B,vocab_size], channels 1..7 [7,B,speech_vocab_size] */
int32_t mtts_read_logits(MttsEngine* e, uint16_t* host_logits0, uint16_t* host_logits17, void* stream);
/* the same for an MTTS_DTYPE_F32 engine: fp32 logits */
int32_t mtts_read_logits_f32(MttsEngine* e, float* host_logits0, float* host_logits17, void* stream);
/* ---- continuous batching: slots are refilled while other dialogues are mid-flight ---------
 * mtts_sched_open: B empty slots, gen_cap rows of token storage each.  mtts_slot_submit: prefill ONE delay-shifted
 * prompt (host int64 [T][8], no padding) into an empty slot; its Philox stream is (seed; step, 0, channel).
 * mtts_step advances every occupied slot; a finished dialogue leaves the batch at once.  mtts_slot_states:
 * int32 [B][4] = (active, unfinished, rows generated, tokens cached).  mtts_slot_read: its rows int64 [steps][8]. */
int32_t mtts_sched_open(MttsEngine* e, int32_t B, int32_t gen_cap, const MttsSamplerCfg* sampler, void* stream);
int32_t mtts_slot_submit(MttsEngine* e, int32_t slot, const int64_t* host_ids, int32_t T, int32_t max_length, uint64_t seed,
                         void* stream);
/* the same with an explicit Philox row id: the dialogue draws from (seed; step, row_id, channel) -- what row `row_id` of a
 * static batch with that seed draws (mtts_slot_submit = row_id 0) */
int32_t mtts_slot_submit_row(MttsEngine* e, int32_t slot, const int64_t* host_ids, int32_t T, int32_t max_length, uint64_t seed,
                             int32_t row_id, void* stream);
/* Philox row ids of the rows of the NEXT mtts_begin / mtts_generate (consumed by it; default 0..B-1): one rank's share of
 * a sharded batch draws what its rows would draw inside the whole batch (host_row_ids int32 [n], n = that call's B). */
int32_t mtts_set_row_ids(MttsEngine* e, const int32_t* host_row_ids, int32_t n);
int32_t mtts_slot_states(MttsEngine* e, int32_t* host_state, void* stream);
int32_t mtts_slot_read(MttsEngine* e, int32_t slot, int64_t* host_rows, int32_t capacity_steps, int32_t* n_steps);
int32_t mtts_read_seq_state(MttsEngine* e, int32_t* host_nas, int32_t* host_unfinished, int32_t* host_kv_len, void* stream);
/* Evict the dialogue in `slot` (scheduler mode): it leaves the batch now and its KV pages return to the pool; the host
 * re-submits it later (tokens are a function of prompt and seed only, so the re-run reproduces them).  Used when
 * mtts_step reports MTTS_ENOMEM: the pool cannot grow every resident dialogue by another page. */
int32_t mtts_slot_evict(MttsEngine* e, int32_t slot, void* stream);
/* KV pool occupancy: pages in the pool, pages free, pages per page-table row. */
int32_t mtts_kv_pool_state(MttsEngine* e, int32_t* total_pages, int32_t* free_pages, int32_t* max_pages_per_seq);
/* host_table int32 [max_batch][max_pages_per_seq] (the device page table as the host tracks it), host_n_pages int32 [max_batch]. */
int32_t mtts_read_page_table(MttsEngine* e, int32_t* host_table, int32_t* host_n_pages);
/* Forced replay mode of mtts_generate's verification hook: 0 (default) the forced row replaces the state machine's
 * output (replay of a greedy run); 1 it replaces the step's raw draw before the state machine, host_decisions receives
 * the raw draws (replay of a SAMPLED reference run: the state follows the reference's history); 2 = as 1, and the forced
 * row is also the raw draw of rows that max_length has cut off and that the reference keeps evaluating
 * (modeling_asteroid.py:140-141: tests that script chained finished-row resurrections). */
int32_t mtts_set_forced_mode(MttsEngine* e, int32_t as_draw);
/* verification hook: the page table as the device holds it once everything queued on `stream` has run
 * (host_table int32 [max_batch][max_pages_per_seq]); compare with mtts_read_page_table. */
int32_t mtts_debug_read_device_page_table(MttsEngine* e, int32_t* host_table, void* stream);
/* Frames first..first+n-1 as codec codes int64 [8][B][n] on the device (delay pattern undone, generation_utils.py:416-425);
 * `stream` must be ordered after the steps that produced frame first+n+6. */
int32_t mtts_export_codes(MttsEngine* e, int32_t first, int32_t n, int64_t* dev_codes, void* stream);
/* name of / time spent in the dominant decode kernel since the last reset, measured with hipEvents
 * on the launch stream (bench.py's roofline leg). */
int32_t mtts_profile_enable(MttsEngine* e, int32_t on);
int32_t mtts_profile_read(MttsEngine* e, int32_t which, double* total_ms, int64_t* launches, int64_t* bytes);

/* measurement hooks (bench / profiling only, never on the product path) */
int32_t mtts_debug_set_kv_len(MttsEngine* e, int32_t kv_len);   /* also fills the KV pages with pseudo-random bf16 */
/* sealed KV pages (csrc/attn.hip kv_seal): out6 = {complete K pages of the live dialogues x kv heads x
 * layers, of which NOT sealed (a lane's 128 values held more than 8 distinct high bytes: read as bf16), the same for V,
 * number of layers whose K reads / V reads currently take the sealed pages (the read policy switches a layer back to bf16
 * reads when more than 1 in 8 of its pages did not seal)}.  MTTS_ESTATE when the engine keeps none (fp32 / fp16 engines,
 * MTTS_KV_PACK=0). */
int32_t mtts_debug_kv_pack_stats(MttsEngine* e, int64_t* out6);
/* train of `iters` launches of one decode attention pass (1 scores, 2 P.V) at the current state; when the product would
 * run the fused q/k/v epilogue at this size the train does too and overwrites the current position's K/V rows:
 * call it only when no further step follows */
int32_t mtts_k_attn_bench(MttsEngine* e, int32_t phase, int32_t iters, float* avg_ms, int64_t* bytes_per_launch);
/* waves > 0: skinny decode GEMM (32 rows) with that decomposition; waves < 0: the tiled prefill GEMM on -waves rows */
int32_t mtts_k_gemm_bench(int32_t N, int32_t K, int32_t epi, int32_t ksplit, int32_t waves, int32_t copies,
                          int32_t iters, float* avg_us);

/* ---- per-kernel entry points (unit tests; device pointers) ---------------- */
/* Y[M,N] = X[M,K] * W[N,K]^T, bf16 in, fp32 accumulate, bf16 out.  M <= 128: skinny decode kernel; 128 < M <= 512:
 * tiled prefill kernel. */
int32_t mtts_k_gemm_bf16(const void* dev_w, const void* dev_x, void* dev_y,
                         int32_t M, int32_t N, int32_t K, int32_t ksplit, void* stream);
/* RMSNorm (Qwen3RMSNorm, modeling_qwen3.py:59-64): x,w bf16 -> y bf16, rows x n. */
int32_t mtts_k_rmsnorm(const void* dev_x, const void* dev_w, void* dev_y,
                       int32_t rows, int32_t n, float eps, void* stream);
/* q/k/v epilogue of one new token per row (per-head q/k RMSNorm, RoPE, K/V page write: transformers modeling_qwen3.py
 * :148-170,251-259): dev_qkv bf16 [R][(nq+2*nkv)*128] = the q|k|v Linear outputs, host_pos int32 [R], norm weights bf16
 * [128], RoPE tables bf16 [rows][64].  Out (bf16): dev_q [R][nq][128], dev_k / dev_v [R][nkv][128] as read back from the
 * cache pages they were written to. */
int32_t mtts_k_rope_kvwrite(const void* dev_qkv, const int32_t* host_pos, const void* dev_qnorm, const void* dev_knorm,
                            const void* dev_cos, const void* dev_sin, int32_t R, int32_t nq, int32_t nkv, float eps,
                            void* dev_q, void* dev_k, void* dev_v, void* stream);
/* Decode attention, one query token per row, over a paged KV cache (eager_attention_forward, modeling_qwen3.py:185-208,
 * with its three bf16 rounding points): dev_q bf16 [R][nq][128]; dev_k / dev_v bf16 [R][Lmax][nkv][128] (row r holds
 * host_lens[r] tokens, the query is the last one); host_page_table int32 [R][ceil(Lmax/64)] = any permutation of the
 * pool's page numbers (NULL: consecutive).  dev_out bf16 [R][nq*128].  R <= 32. */
int32_t mtts_k_paged_attn_decode(const void* dev_q, const void* dev_k, const void* dev_v, const int32_t* host_lens,
                                 const int32_t* host_page_table, int32_t R, int32_t Lmax, int32_t nq, int32_t nkv,
                                 void* dev_out, void* stream);
/* The sealed page format on its own: dev_pages = npages x 16 KiB (a page as the cache holds it: [16 units][64 lanes][16 B]),
 * dev_sealed = npages x 13 KiB ([13 units][64 lanes][16 B]: units 0-7 the low bytes of the lane's 128 values in order,
 * 8-11 one code nibble per value (byte 4j+k: low nibble = value 8j+k, high nibble = value 8j+4+k; code = sign << 3 | index),
 * unit 12 = 8 dictionary bytes ((bf16 >> 8) & 0x7f, ascending), a 32-bit spare, a 32-bit flag: != 0 = the lane did not fit
 * and the rest of its sealed data is undefined).  Lossless: value = sign << 15 | dictionary[index] << 8 | low byte.
 * as_k = 0: the values as they are.  as_k = 1 (K pages: lane = token, value i = dim i): dim d is first divided
 * by 2^s[d], s[d] = (rounded mean of the non-zero exponent fields of dim d over the page's 64 tokens) - 125 (0 for an
 * all-zero dim; clamped to -127..127), lane l keeps s[2l], s[2l+1] as int8 in the low 16 bits of its spare; k = stored value
 * x 2^s[d] exactly (a lane with a denormal, inf / NaN or an exponent that would leave 1..254 is flagged instead).
 * as_k = 2 (V pages: lane = 32 * sub + dl, value 8 it + 2 c + h = token 4 it + 2 sub + h, dim 4 dl + c): token t is first
 * divided by 2^s[t], s[t] = (rounded mean of the non-zero exponent fields of token t's 128 values) - (the smallest such
 * mean of the page), rounded down to even, 0..126 (0 for an all-zero token); lane t keeps s[t] in the low byte of its spare; the same flags. */
int32_t mtts_k_kv_seal(const void* dev_pages, int32_t npages, void* dev_sealed, int32_t as_k, void* stream);
/* One sampler call on fp32-from-bf16 logits (HF processors + engine draw). */
int32_t mtts_k_sample(const void* dev_logits_bf16, int32_t rows, int32_t vocab,
                      const void* dev_history_bitmap, const MttsSamplerCfg* cfg,
                      int32_t mask_id, uint64_t seed, int32_t step, int32_t channel,
                      int32_t* dev_tokens, void* stream);

/* ======================================================================================
 * XY_Tokenizer decode path (codes -> 24 kHz waveform), fp32.
 * Replaces XY_Tokenizer.inference_detokenize (XY_Tokenizer/xy_tokenizer/model.py:104-128);
 * the 30 s-window / 20 s-stride scheduling of XY_Tokenizer.decode (model.py:195-256) stays
 * in the host mirror, which calls mtts_codec_detokenize once per window batch.
 * ====================================================================================== */
typedef struct MttsCodecConfig {       /* decode-side fields of xy_tokenizer_config.yaml */
    int32_t nq, codebook_size, rvq_dim, quant_out_dim;
    int32_t adapter_layers, adapter_dim, adapter_heads, adapter_ffn, adapter_max_pos;
    int32_t up_stride;
    int32_t dec_layers, dec_dim, dec_heads, dec_ffn, dec_max_pos, mel_bins;
    int32_t voc_dim, voc_inter, voc_layers, n_fft, hop;
    /* encode side (voice-clone prompts) */
    int32_t mel_n_fft, mel_hop, mel_frames;
    int32_t enc_layers, enc_dim, enc_heads, enc_ffn, enc_max_pos;
    int32_t sem_adapter_layers, pre_rvq_layers, down_pool;
} MttsCodecConfig;

typedef struct MttsCodec MttsCodec;
const char* mtts_codec_last_error(void);
int32_t mtts_codec_create(const MttsCodecConfig* cfg, int32_t device, MttsCodec** out);
int32_t mtts_codec_destroy(MttsCodec* c);
/* Bind one fp32 tensor by role name (role <- reference state-dict key and the
 * re-layout applied); the engine copies it. */
int32_t mtts_codec_bind(MttsCodec* c, const char* role, const float* dev_f32, int64_t n, void* stream);
/* dev_codes int64 [nq][B][T] (T <= 375), host_lens int32 [B]; dev_wav f32 [B][T*1920]. Synchronous. */
int32_t mtts_codec_detokenize(MttsCodec* c, const int64_t* dev_codes, const int32_t* host_lens, int32_t B, int32_t T,
                              float* dev_wav, void* stream);
/* Enqueue-only form (overlap with the decode loop on another stream); mtts_codec_check() synchronises and
 * reports a code index outside the codebook. */
int32_t mtts_codec_detokenize_async(MttsCodec* c, const int64_t* dev_codes, const int32_t* host_lens, int32_t B, int32_t T,
                                    float* dev_wav, void* stream);
int32_t mtts_codec_check(MttsCodec* c, void* stream);
/* Encode one chunk (<= 30 s of 16 kHz audio) per row.  Replaces XY_Tokenizer.inference_tokenize
 * (XY_Tokenizer/xy_tokenizer/model.py:55-101), log-mel included.  dev_wav f32 [B][nsamp] zero padded,
 * host_lens int32 [B]; dev_codes int64 [nq][B][375]; host_code_lens int32 [B] (out).  Synchronous. */
int32_t mtts_codec_tokenize(MttsCodec* c, const float* dev_wav, const int32_t* host_lens, int32_t B, int32_t nsamp,
                            int64_t* dev_codes, int32_t* host_code_lens, void* stream);
/* unit test: C[M,N] = act(A[M,K] * W[N,K]^T + bias); act 0 none, 1 GELU(erf); act | 0x100: the decode direction's
 * bf16x3 kernel instead of the exact-f32 MFMA */
int32_t mtts_k_gemm_f32(const float* dev_a, const float* dev_w, const float* dev_bias, float* dev_c,
                        int32_t M, int32_t N, int32_t K, int32_t act, void* stream);
/* Tuning hook: time the decoder's pre-split bf16x3 GEMM (gemm_b3t_kernel) alone on synthetic operands.
 * flags: 1 GELU | 2 gamma | 4 residual | 8 fragment-packed output (the decoder uses 0, 4, 6, 9); tile_code 0 = the
 * production choice for the shape, else NA NB U OCC as decimal digits.  No reference counterpart. */
int32_t mtts_k_gemm_planes_bench(int32_t M, int32_t N, int32_t K, int32_t flags, int32_t tile_code, int32_t iters,
                                 float* avg_us, int32_t* code_used);

#ifdef __cplusplus
}
#endif
#endif /* MTTS_H */
