// Host side of libmtts.so: engine object, weight binding, KV page pool, the
// prefill / decode-step orchestration and the C ABI of include/mtts.h.
//
// Mirrors (file:line in /root/reference):
//   AsteroidTTSInstruct.forward inference branch   modeling_asteroid.py:337-380,411-426
//   AsteroidTTSModel._prepare_multi_modal_inputs    modeling_asteroid.py:235-250
//   CustomMixin._sample                             modeling_asteroid.py:83-169
// and, third-party, transformers Qwen3Model.forward (models/qwen3/modeling_qwen3.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mtts.h"
#include "common.h"

// ---- kernels' launchers (other translation units) ---------------------------
struct GemmPlan { int waves, ksplit, kt_per_split, kt_per_wave; };
enum { EPI_PARTIAL = 0, EPI_BF16 = 1, EPI_SILU = 2 };
GemmPlan mtts_plan_gemm(int Npad, int K, int want_ksplit);
GemmPlan mtts_plan_gemm_forced(int Npad, int K, int ksplit, int waves);
void launch_gemm(int epi, int mb, const GemmPlan& p, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                 float* partial, uint16_t* out, hipStream_t st);
int mtts_tile_ksplit(int Npad, int K, int R);
void launch_gemm_tile(int epi, int R, int ksplit, const void* Wp, const void* Xp, int K, int Npad, int n_valid,
                      float* partial, uint16_t* out, hipStream_t st);
int mtts_small_lds_bytes(const GemmPlan& p, int K, int pro);
enum { EPI_SILU_RM = 3 };
void launch_gemv_small(int epi, int pro, const GemmPlan& p, const void* Wp, int K, int Npad, int n_valid, float* partial,
                       uint16_t* out, const SmallPro& pr, hipStream_t st);
void launch_pack_weight(const void* src, void* dst, int rows, int cols, int rows_pad, int row_mul, int row_off, hipStream_t st);
void launch_pack_rows(const void* src, void* dst, int R, int K, int tiles, hipStream_t st);
void launch_reduce_partial_bf16(const float* partial, void* out, int ksplit, int Npad, int n_valid, int R, hipStream_t st);
void launch_embed_norm(const int32_t* tokens, const RowMeta* meta, const uint16_t* const* tables, const void* norm_w,
                       void* x, void* xn_packed, int R, int H, float eps, hipStream_t st);
void launch_resid_norm(const float* partial, int ksplit, int Npad, void* x, const void* norm_w, void* xn_packed,
                       void* hlast, const RowMeta* meta, int R, int H, float eps, hipStream_t st);
void launch_qkv_post(const float* partial, int ksplit, int Npad, const RowMeta* meta, const void* qnw, const void* knw,
                     const void* cosb, const void* sinb, void* qbuf, void* kcache, void* vcache,
                     const int32_t* page_table, int max_pages, int total_pages, int R, int nq, int nkv, float eps,
                     hipStream_t st);
void launch_rmsnorm_rows(const void* x, const void* w, void* y, int rows, int n, float eps, hipStream_t st);
void launch_fill_random_bf16(void* p, size_t n, uint32_t seed, hipStream_t st);
int launch_attn(const void* qbuf, void* kcache, void* vcache, const int32_t* page_table,
                const RowMeta* meta, void* scores, float* stats, float* opart, void* out_packed, int R,
                int pages_bound, int max_pages, int total_pages, int nchunks_max, int nq, int nkv, float scale,
                const QkvFuse* fuse, int phase, hipStream_t st, const KvPack* pack = nullptr);
void launch_kv_seal_rows(const void* kcache, const void* vcache, void* kpack, void* vpack, const int32_t* page_table,
                         const RowMeta* meta, int R, int max_pages, int total_pages, int nkv, int L, unsigned long long* cnt, hipStream_t st);
void launch_kv_seal_all(const void* kcache, const void* vcache, void* kpack, void* vpack, int total_pages, int nkv, int L,
                        unsigned long long* cnt, hipStream_t st);
void launch_kv_seal_pages(const void* raw, void* pk, int npages, int as_k, hipStream_t st);
void launch_kv_pack_count(const void* kpack, const void* vpack, const int32_t* page_table, const int32_t* complete, int B, int max_pages,
                          int total_pages, int nkv, int L, unsigned long long* out, hipStream_t st);
struct SeqState { int32_t nas, unfinished, kv_len, step, base_length, max_length, row_id, active; uint64_t seed; };
struct LoopState { int32_t step, done, continuous, B, error, gen_cap, forced_draw, logits_f32; };
struct SampleScratch { uint32_t* hist; float* slice_val; int32_t* slice_idx; float* cand_val; int32_t* cand_idx; uint32_t* cand_n; int32_t* overflow; float* full_val; int32_t* full_idx; uint32_t* nuc_cnt; unsigned long long* nuc_mass; };
#define SAMP_CAND 4096
#define SAMP_NS 32
void launch_sample(const void* logits0, const void* logits17, int V0, int Vs, int Vs_pad, const uint32_t* bitmaps,
                   int bm_words, const MttsSamplerCfg* cfgs, const LoopState* ls, const SeqState* seqs, uint64_t seed,
                   int32_t* decisions, int32_t* err, int B, const SampleScratch& sc, int ch0_sampled, int full_cap,
                   hipStream_t st);
void launch_sample_single(const void* logits, int rows, int vocab, const uint32_t* bitmap, int bm_words,
                          const MttsSamplerCfg* cfgs8, int mask_id, uint64_t seed, int step, int channel,
                          int32_t* decisions, int32_t* err, const SampleScratch& sc, int full_cap, hipStream_t st);
static int alloc_scratch(SampleScratch& sc, int rows, int vocab);
static int full_cap_for(int vocab) { int p = 1; while (p < vocab) p <<= 1; return vocab > SAMP_CAND ? p : 0; }
static void free_scratch(SampleScratch& sc);
void launch_update(const int32_t* decisions, int32_t* dec_log, const int32_t* forced, const int32_t* tf_tail,
                   int32_t* gen, int32_t* cur_tokens, SeqState* seqs, RowMeta* meta, uint32_t* bitmaps, int bm_words,
                   LoopState* ls, int eos, int spad, int sp_lo, int sp_hi, hipStream_t st);

void launch_export_codes(const int32_t* gen, int64_t* codes, int B, int first, int n, int speech_offset, int clamp_hi,
                         int cap, hipStream_t st);
void launch_f32_embed_norm(const int32_t* tokens, const RowMeta* meta, const float* const* tables, const float* norm_w, float* x,
                           float* xn, int R, int H, float eps, int h16, hipStream_t st);
void launch_f32_resid_norm(const float* y, float* x, const float* norm_w, float* xn, float* hlast, const RowMeta* meta, int R,
                           int H, float eps, int h16, hipStream_t st);
void launch_f32_linear(const float* W, const float* X, float* Y, int R, int N, int K, long ldy, int h16, bool gemv, hipStream_t st);
void launch_f32_qkv_post(const float* qkv, int ldq, const RowMeta* meta, const float* qnw, const float* knw, const float* cosb,
                         const float* sinb, float* qbuf, float* kcache, float* vcache, const int32_t* page_table, int max_pages,
                         int total_pages, int R, int nq, int nkv, float eps, int h16, hipStream_t st);
void launch_f32_attn(const float* qbuf, const float* kcache, const float* vcache, const int32_t* page_table, const RowMeta* meta,
                     float* scores, float* out, int R, int max_pages, int total_pages, int nq, int nkv, float scale, int Lmax,
                     int h16, hipStream_t st);
void launch_f32_swiglu(const float* gu, float* act, int R, int I, int h16, hipStream_t st);
#define MTTS_PF32CAP 256       // rows of a prefill pass in the fp32 engine (bounds its fp32 score scratch)
struct PageEdits { int32_t n; int32_t idx[31]; int32_t val[31]; };     // page-table entries handed over as launch arguments
void launch_set_pages(int32_t* table, const PageEdits& ed, hipStream_t st);
#define FLUSH_STEPS 7          // a dialogue whose EOS falls within 7 steps of max_length still runs its delay-pattern flush (modeling_asteroid.py:165-168)
#define LINGER_STEPS 14        // static batch: a row finished BY max_length can be resurrected for a flush while another row's flush is still running (sampler.hip: update_kernel), so a batch runs up to 6 + 8 steps past max_length after ONE resurrection
// Resurrections chain: a resurrected row's own 7-step flush keeps the batch alive, and every step of it re-tests the other
// cut-off rows (modeling_asteroid.py:140-141,168), so each further row can add up to 6 more steps: 6 * B + 8 bounds the run.
// Storage (token rows, KV pages, RoPE rows) is sized for that bound where the engine's max_seq_len leaves room, and for
// LINGER_STEPS at least; a chain that outruns the room is reported (MTTS_ESTATE), never truncated silently.
static inline int linger_bound(int B) { return 6 * B + 8; }

// ---- errors -------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t _e = (x);                                                               \
        if (_e != hipSuccess) return fail(MTTS_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

struct Layer {
    void *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr;     // packed
    void *ln_in = nullptr, *ln_post = nullptr, *qn = nullptr, *kn = nullptr; // bf16 vectors
    int bound = 0;
};

enum { PROF_SCORES = 0, PROF_PV = 1, PROF_GEMM = 2, PROF_STEP = 3, PROF_N = 4 };

struct MttsEngine {
    MttsConfig cfg;
    int device = 0;
    int H, I, L, nq, nkv, V0, Vs, Vs_pad, V0_pad, qkv_rows;
    std::vector<Layer> layers;
    void* emb[8] = {nullptr};          // row-major tables (gather)
    void* head0 = nullptr;             // packed [V0_pad][H]
    void* heads17 = nullptr;           // packed [7*Vs_pad][H]
    void* final_norm = nullptr;
    void *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_rows = 0;
    int emb_bound = 0, norm_bound = 0;
    const uint16_t** d_tables = nullptr;
    // plans
    GemmPlan p_qkv, p_o, p_gu, p_d, p_h0, p_h17;
    // workspaces
    float* partial = nullptr;
    float* partial2 = nullptr;          // small-batch path: o_proj / down_proj slabs (the qkv slabs stay in `partial`)
    void *x2 = nullptr, *act_rm = nullptr;   // small-batch path: second residual buffer (ping-pong), row-major SwiGLU output
    int small_rows = SMALL_RP;          // decode batches up to this many dialogues take the small-batch path (0 = off)
    void *x = nullptr, *xn = nullptr, *attn_p = nullptr, *act_p = nullptr, *qbuf = nullptr, *hlast = nullptr, *xh = nullptr;
    void *logits0 = nullptr, *logits17 = nullptr, *join_logits0 = nullptr, *join_logits17 = nullptr;
    void* scores = nullptr;
    float *stats = nullptr, *opart = nullptr;
    // kv
    void *kcache = nullptr, *vcache = nullptr;
    size_t layer_stride = 0;           // elements per layer in each cache
    int total_pages = 0, max_pages = 0, nchunks_max = 0;
    int32_t* d_page_table = nullptr;
    std::vector<int32_t> h_page_table;  // [slot][max_pages]: pages a slot owns, in position order
    // KV page pool: pages are handed out on demand as a dialogue's length crosses a page boundary and come back
    // when it finishes (free list = stack; initial order ascending, or shuffled by MTTS_PAGE_SHUFFLE for the tests)
    std::vector<int32_t> free_pages;
    std::vector<int32_t> n_pages;       // pages each slot owns
    std::vector<char> slot_live;        // host's view: the slot holds a dialogue that may still step
    PageEdits pending_edits;            // table entries not yet on the device
    int forced_draw = 0;
    std::vector<int32_t> next_row_ids;  // Philox row ids of the next mtts_begin (mtts_set_row_ids); empty = 0..B-1
    // ---- MTTS_DTYPE_F32 engine (f32path.hip): plain fp32 copies of everything, no packed layouts ----
    bool f32 = false;
    int h16 = 0;                        // MTTS_DTYPE_F16: the fp32 engine with fp16 rounding points (f32path.hip: r16)
    struct LayerF32 { float *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr, *ln_in = nullptr, *ln_post = nullptr, *qn = nullptr, *kn = nullptr; };
    std::vector<LayerF32> lf;
    float* embf[8] = {nullptr};
    const float** d_tables_f = nullptr;
    float *final_norm_f = nullptr, *rope_cos_f = nullptr, *rope_sin_f = nullptr;
    float *kcache_f = nullptr, *vcache_f = nullptr;
    float *xf = nullptr, *xnf = nullptr, *qkvf = nullptr, *qbuf_f = nullptr, *attnf = nullptr, *yf = nullptr, *guf = nullptr,
          *actf = nullptr, *hlast_f = nullptr, *scores_f = nullptr;
    // generation state
    SeqState* d_seqs = nullptr;
    RowMeta* d_meta = nullptr;          // decode rows
    LoopState* d_ls = nullptr;
    LoopState* h_ls = nullptr;          // pinned mirror
    SeqState* h_seqs = nullptr;         // pinned mirror of d_seqs (mtts_sync_state)
    int32_t *d_decisions = nullptr, *d_cur = nullptr, *d_gen = nullptr, *d_declog = nullptr, *d_forced = nullptr,
            *d_tf = nullptr;
    uint32_t* d_bitmaps = nullptr;
    int bm_words = 0;
    MttsSamplerCfg* d_scfg = nullptr;
    SampleScratch sscr = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int ch0_sampled = 0;
    int32_t* d_pf_tokens = nullptr;     // prefill staging
    RowMeta* d_pf_meta = nullptr;
    size_t pf_cap_rows = 0;
    int gen_cap = 0;
    // current run
    int B = 0, T = 0, base_length = 0, max_length = 0, max_steps = 0, steps_issued = 0;
    bool continuous = false;
    std::vector<int> join_step;         // engine step at which each slot's dialogue joined
    std::vector<int> n_real;
    int max_real = 0;
    uint64_t seed = 0;
    bool began = false, has_forced = false;
    // decode-step graphs: one captured step per (rows, KV page bound, ...) key, replayed by mtts_step
    struct StepGraph { int B, pages, forced, ch0; hipGraphExec_t exec; };
    std::vector<StepGraph> graphs;
    hipStream_t cap_stream = nullptr;
    bool use_graphs = true;
    // sealed KV pages (attn.hip: kv_seal): every COMPLETE page also kept in a lossless 13-bit form the decode attention
    // reads instead of the bf16 page (MTTS_KV_PACK=0: off, no second pool)
    int kv_pack = 1;
    void *kpack = nullptr, *vpack = nullptr;
    size_t pk_layer_stride = 0;         // bytes per layer of a sealed pool
    // read policy: a page that did not seal costs a wasted sealed read + the bf16 read (29 units instead of 16), so a
    // layer whose K (or V) pages stop sealing (more than 1 in 8 since mtts_begin) goes back to bf16 reads; the sealer
    // counts per layer {K sealed, K not, V sealed, V not}, mtts_sync_state looks at the counts (MTTS_KV_PACK=2: no policy)
    unsigned long long *d_seal_cnt = nullptr, *h_seal_cnt = nullptr;
    std::vector<char> pack_k_on, pack_v_on;
    int pack_min_work = 512;            // sealed reads from this many rows x KV pages up (MTTS_KV_PACK_MIN)
    int fuse_qkv_max = 2560;            // decode: q/k/v epilogue inside the attention kernels while rows x KV pages <= this (round 3: 1024 -> 2560 = 32 rows x 80 pages, once its loads go out before the page's: wins at 32 x 64, loses at 64 x 64)
    int pf_mfma_pages = 0;              // prefill attention: tile-sharing MFMA kernels from this many KV pages up (0 = always; a dialogue's numerics must not depend on its batch)
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[PROF_N];
    int64_t prof_bytes[PROF_N] = {0, 0, 0, 0};
};

static hipStream_t S(void* s) { return (hipStream_t)s; }

const char* mtts_last_error(void) { return g_err; }
int32_t mtts_version(void) { return 200; }

template <typename T>
static int dalloc(T** p, size_t n, bool zero = true) {
    HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
    if (zero) HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
    return 0;
}
#define TRY(x)             \
    do {                   \
        int _r = (x);      \
        if (_r) return _r; \
    } while (0)

static int alloc_scratch(SampleScratch& sc, int rows, int vocab) {
    const int fc = full_cap_for(vocab);
    sc.full_val = nullptr; sc.full_idx = nullptr; sc.nuc_cnt = nullptr; sc.nuc_mass = nullptr;
    TRY(dalloc(&sc.overflow, (size_t)rows));
    if (fc) {            // full-vocabulary path (sampling without top_k): per token a key and a level-0 bin, per row the level-0 histogram
        TRY(dalloc(&sc.full_val, (size_t)rows * fc, false));
        TRY(dalloc(&sc.full_idx, (size_t)rows * fc, false));
        TRY(dalloc(&sc.nuc_cnt, (size_t)rows * 2048));
        TRY(dalloc(&sc.nuc_mass, (size_t)rows * 2048));
    }
    TRY(dalloc(&sc.hist, (size_t)rows * 2048));
    TRY(dalloc(&sc.slice_val, (size_t)rows * SAMP_NS));
    TRY(dalloc(&sc.slice_idx, (size_t)rows * SAMP_NS));
    TRY(dalloc(&sc.cand_val, (size_t)rows * SAMP_CAND));
    TRY(dalloc(&sc.cand_idx, (size_t)rows * SAMP_CAND));
    TRY(dalloc(&sc.cand_n, (size_t)rows));
    return 0;
}
static void free_scratch(SampleScratch& sc) {
    hipFree(sc.hist); hipFree(sc.slice_val); hipFree(sc.slice_idx); hipFree(sc.cand_val); hipFree(sc.cand_idx); hipFree(sc.cand_n);
    hipFree(sc.overflow);
    if (sc.full_val) { hipFree(sc.full_val); hipFree(sc.full_idx); hipFree(sc.nuc_cnt); hipFree(sc.nuc_mass); }
}

// ---- KV page pool -----------------------------------------------------------------------------------------------
// Free list = stack whose top is the lowest page number (a fresh engine hands pages out in ascending order).
// MTTS_PAGE_SHUFFLE=<seed> (test hook) shuffles it, so that page tables are arbitrary permutations.
static void pool_reset(MttsEngine* e) {
    e->free_pages.resize(e->total_pages);
    for (int i = 0; i < e->total_pages; ++i) e->free_pages[i] = e->total_pages - 1 - i;
    if (const char* g = getenv("MTTS_PAGE_SHUFFLE")) {
        uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(g);
        for (int i = e->total_pages - 1; i > 0; --i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(e->free_pages[i], e->free_pages[(x >> 33) % (uint64_t)(i + 1)]);
        }
    }
    std::fill(e->n_pages.begin(), e->n_pages.end(), 0);
    e->pending_edits.n = 0;
}
// table entries reach the device as launch arguments of a one-wave kernel on the caller's stream: ordered with the
// steps around it, and no host buffer has to outlive the call
static int pool_flush(MttsEngine* e, hipStream_t st) {
    if (e->pending_edits.n) {
        launch_set_pages(e->d_page_table, e->pending_edits, st);
        e->pending_edits.n = 0;
        HIPCHK(hipGetLastError());
    }
    return 0;
}
// make slot b own at least `need` pages; MTTS_ENOMEM when the pool runs dry (nothing is taken back)
static int pool_grow(MttsEngine* e, int b, int need, hipStream_t st) {
    if (need > e->max_pages) return fail(MTTS_ENOMEM, "slot %d needs %d KV pages, a sequence holds at most %d (max_seq_len %d)", b, need, e->max_pages, e->cfg.max_seq_len);
    while (e->n_pages[b] < need) {
        if (e->free_pages.empty()) return fail(MTTS_ENOMEM, "KV page pool exhausted (%d pages of %d tokens): slot %d needs page %d", e->total_pages, MTTS_PAGE, b, e->n_pages[b]);
        const int page = e->free_pages.back();
        e->free_pages.pop_back();
        const int at = b * e->max_pages + e->n_pages[b]++;
        e->h_page_table[at] = page;
        // one batch is written by the lanes of ONE store (set_pages_kernel): an index must not appear twice in it
        int dup = -1;
        for (int i = 0; i < e->pending_edits.n; ++i) if (e->pending_edits.idx[i] == at) dup = i;
        if (dup >= 0) { e->pending_edits.val[dup] = page; continue; }
        if (e->pending_edits.n == 31) TRY(pool_flush(e, st));
        e->pending_edits.idx[e->pending_edits.n] = at;
        e->pending_edits.val[e->pending_edits.n++] = page;
    }
    return 0;
}
// every launch that could touch the slot's pages must have been issued before (stream order protects the rest:
// the next owner's writes are enqueued after them)
static void pool_release(MttsEngine* e, int b) {
    for (int i = e->n_pages[b] - 1; i >= 0; --i) e->free_pages.push_back(e->h_page_table[(size_t)b * e->max_pages + i]);
    e->n_pages[b] = 0;
    // table entries of this slot that never reached the device (a grow that ended in MTTS_ENOMEM) are void now
    int k = 0;
    for (int i = 0; i < e->pending_edits.n; ++i)
        if (e->pending_edits.idx[i] / e->max_pages != b) {
            e->pending_edits.idx[k] = e->pending_edits.idx[i];
            e->pending_edits.val[k++] = e->pending_edits.val[i];
        }
    e->pending_edits.n = k;
}

// ---- MTTS_DTYPE_F32 engine: allocation, binding, forward (kernels: f32path.hip) -------------------------------------
static int create_f32(MttsEngine* e) {
    const size_t H = e->H, I = e->I, D = MTTS_HD;
    e->lf.resize(e->L);
    for (auto& l : e->lf) {
        TRY(dalloc(&l.wqkv, (size_t)e->qkv_rows * H, false));
        TRY(dalloc(&l.wo, H * e->nq * D, false));
        TRY(dalloc(&l.wgu, 2 * I * H, false));
        TRY(dalloc(&l.wd, H * I, false));
        TRY(dalloc(&l.ln_in, H)); TRY(dalloc(&l.ln_post, H)); TRY(dalloc(&l.qn, D)); TRY(dalloc(&l.kn, D));
    }
    TRY(dalloc(&e->embf[0], (size_t)e->V0 * H, false));
    for (int ch = 1; ch < 8; ++ch) TRY(dalloc(&e->embf[ch], (size_t)e->Vs * H, false));
    TRY(dalloc(&e->final_norm_f, H));
    TRY(dalloc(&e->d_tables_f, 8));
    HIPCHK(hipMemcpy((void*)e->d_tables_f, e->embf, 8 * sizeof(void*), hipMemcpyHostToDevice));
    const size_t P = MTTS_PF32CAP;
    TRY(dalloc(&e->xf, P * H)); TRY(dalloc(&e->xnf, P * H)); TRY(dalloc(&e->yf, P * H));
    TRY(dalloc(&e->qkvf, P * e->qkv_rows)); TRY(dalloc(&e->qbuf_f, P * e->nq * D)); TRY(dalloc(&e->attnf, P * e->nq * D));
    TRY(dalloc(&e->guf, P * 2 * I)); TRY(dalloc(&e->actf, P * I));
    TRY(dalloc(&e->hlast_f, (size_t)MTTS_RCAP * H));
    TRY(dalloc((float**)&e->logits0, (size_t)MTTS_RCAP * e->V0_pad));
    TRY(dalloc((float**)&e->logits17, (size_t)MTTS_RCAP * 7 * e->Vs_pad));
    return 0;
}

static int forward_rows_f32(MttsEngine* e, const int32_t* d_tokens, const RowMeta* d_meta, int R, int heads, hipStream_t st) {
    const int H = e->H, I = e->I, nq = e->nq, nkv = e->nkv;
    const float eps = e->cfg.rms_norm_eps, scale = 1.0f / sqrtf((float)MTTS_HD);
    const int Lmax = e->max_pages * MTTS_PAGE, h16 = e->h16;
    const bool dec = heads == 1;                       // decode rows: GEMV in groups of 8 (batch-independent numerics)
    launch_f32_embed_norm(d_tokens, d_meta, e->d_tables_f, e->lf[0].ln_in, e->xf, e->xnf, R, H, eps, h16, st);
    for (int n = 0; n < e->L; ++n) {
        auto& l = e->lf[n];
        float* kc = e->kcache_f + e->layer_stride * n;
        float* vc = e->vcache_f + e->layer_stride * n;
        launch_f32_linear(l.wqkv, e->xnf, e->qkvf, R, e->qkv_rows, H, e->qkv_rows, h16, dec, st);
        launch_f32_qkv_post(e->qkvf, e->qkv_rows, d_meta, l.qn, l.kn, e->rope_cos_f, e->rope_sin_f, e->qbuf_f, kc, vc,
                            e->d_page_table, e->max_pages, e->total_pages, R, nq, nkv, eps, h16, st);
        launch_f32_attn(e->qbuf_f, kc, vc, e->d_page_table, d_meta, e->scores_f, e->attnf, R, e->max_pages, e->total_pages, nq,
                        nkv, scale, Lmax, h16, st);
        launch_f32_linear(l.wo, e->attnf, e->yf, R, H, nq * MTTS_HD, H, h16, dec, st);
        launch_f32_resid_norm(e->yf, e->xf, l.ln_post, e->xnf, nullptr, d_meta, R, H, eps, h16, st);
        launch_f32_linear(l.wgu, e->xnf, e->guf, R, 2 * I, H, 2 * I, h16, dec, st);
        launch_f32_swiglu(e->guf, e->actf, R, I, h16, st);
        launch_f32_linear(l.wd, e->actf, e->yf, R, H, I, H, h16, dec, st);
        const bool lastl = n == e->L - 1;
        launch_f32_resid_norm(e->yf, e->xf, lastl ? e->final_norm_f : e->lf[n + 1].ln_in, e->xnf, lastl ? e->hlast_f : nullptr,
                              d_meta, R, H, eps, h16, st);
    }
    if (heads) {
        // decode rows are the dialogues themselves (row b = slot b); after a prefill the last tokens' states are in hlast
        const float* xin = heads == 1 ? e->xnf : e->hlast_f;
        launch_f32_linear(e->embf[0], xin, (float*)e->logits0, e->B, e->V0, H, e->V0_pad, h16, true, st);
        for (int c = 1; c < 8; ++c)
            launch_f32_linear(e->embf[c], xin, (float*)e->logits17 + (size_t)(c - 1) * e->Vs_pad, e->B, e->Vs, H, 7 * e->Vs_pad, h16, true, st);
    }
    HIPCHK(hipGetLastError());
    return MTTS_OK;
}

int32_t mtts_engine_create(const MttsConfig* c, int32_t device, MttsEngine** out) {
    if (!c || !out) return fail(MTTS_EINVAL, "null argument");
    if (c->head_dim != MTTS_HD) return fail(MTTS_EINVAL, "head_dim must be 128 (got %d)", c->head_dim);
    if (c->channels != 8) return fail(MTTS_EINVAL, "channels must be 8");
    if (c->hidden_size % 16 || c->intermediate_size % 16) return fail(MTTS_EINVAL, "hidden/intermediate must be multiples of 16");
    if (c->hidden_size > 8192) return fail(MTTS_EINVAL, "hidden_size > 8192 not built (resid_norm keeps a row in registers)");
    if (c->num_attention_heads % c->num_key_value_heads) return fail(MTTS_EINVAL, "bad GQA ratio");
    int G = c->num_attention_heads / c->num_key_value_heads;
    if (G != 1 && G != 2 && G != 4) return fail(MTTS_EINVAL, "GQA group %d not built (1,2,4)", G);
    if (c->max_batch < 1 || c->max_batch > MTTS_RCAP) return fail(MTTS_EINVAL, "max_batch must be 1..128");
    if (c->vocab_size <= 152694 || c->speech_vocab_size <= 1024)
        return fail(MTTS_EINVAL, "vocab too small for the reference's hard-coded mask ids 152694 / 1024");
    HIPCHK(hipSetDevice(device));
    MttsEngine* e = new MttsEngine();
    e->cfg = *c;
    e->device = device;
    if (const char* g = getenv("MTTS_GRAPHS")) e->use_graphs = atoi(g) != 0;
    if (const char* g = getenv("MTTS_FUSE_QKV_MAX")) e->fuse_qkv_max = atoi(g);
    if (const char* g = getenv("MTTS_KV_PACK")) e->kv_pack = atoi(g);
    if (const char* g = getenv("MTTS_KV_PACK_MIN")) e->pack_min_work = atoi(g);
    if (const char* g = getenv("MTTS_PREFILL_MFMA_PAGES")) e->pf_mfma_pages = atoi(g);
    if (const char* g = getenv("MTTS_SMALL_ROWS")) e->small_rows = std::min(std::max(atoi(g), 0), SMALL_RP);
    e->H = c->hidden_size; e->I = c->intermediate_size; e->L = c->num_hidden_layers;
    e->nq = c->num_attention_heads; e->nkv = c->num_key_value_heads;
    e->V0 = c->vocab_size; e->Vs = c->speech_vocab_size;
    e->V0_pad = round_up(e->V0, 32); e->Vs_pad = round_up(e->Vs, 32);
    e->qkv_rows = (e->nq + 2 * e->nkv) * MTTS_HD;
    e->layers.resize(e->L);
    const int H = e->H, I = e->I;
    if (c->dtype != MTTS_DTYPE_BF16 && c->dtype != MTTS_DTYPE_F32 && c->dtype != MTTS_DTYPE_F16)
        return fail(MTTS_EINVAL, "dtype %d not built (bf16 = 0, fp32 = 1, fp16 = 2)", c->dtype);
    e->f32 = c->dtype != MTTS_DTYPE_BF16;
    e->h16 = c->dtype == MTTS_DTYPE_F16;
    if (e->f32) TRY(create_f32(e));
    // packed weights (zeroed: padding rows must be zero)
    if (!e->f32)
    for (auto& l : e->layers) {
        TRY(dalloc((uint16_t**)&l.wqkv, (size_t)e->qkv_rows * H));
        TRY(dalloc((uint16_t**)&l.wo, (size_t)H * e->nq * MTTS_HD));
        TRY(dalloc((uint16_t**)&l.wgu, (size_t)2 * I * H));
        TRY(dalloc((uint16_t**)&l.wd, (size_t)round_up(H, 32) * I));
        TRY(dalloc((uint16_t**)&l.ln_in, (size_t)H));
        TRY(dalloc((uint16_t**)&l.ln_post, (size_t)H));
        TRY(dalloc((uint16_t**)&l.qn, (size_t)MTTS_HD));
        TRY(dalloc((uint16_t**)&l.kn, (size_t)MTTS_HD));
    }
    if (!e->f32) {
    TRY(dalloc((uint16_t**)&e->emb[0], (size_t)e->V0 * H, false));
    for (int ch = 1; ch < 8; ++ch) TRY(dalloc((uint16_t**)&e->emb[ch], (size_t)e->Vs * H, false));
    TRY(dalloc((uint16_t**)&e->head0, (size_t)e->V0_pad * H));
    TRY(dalloc((uint16_t**)&e->heads17, (size_t)7 * e->Vs_pad * H));
    TRY(dalloc((uint16_t**)&e->final_norm, (size_t)H));
    TRY(dalloc(&e->d_tables, 8));
    HIPCHK(hipMemcpy((void*)e->d_tables, e->emb, 8 * sizeof(void*), hipMemcpyHostToDevice));
    }
    // plans
    e->p_qkv = mtts_plan_gemm(e->qkv_rows, H, 0);
    e->p_o = mtts_plan_gemm(round_up(H, 32), e->nq * MTTS_HD, 0);
    e->p_gu = mtts_plan_gemm(2 * I, H, 1);
    e->p_d = mtts_plan_gemm(round_up(H, 32), I, 0);
    e->p_h0 = mtts_plan_gemm(e->V0_pad, H, 1);
    e->p_h17 = mtts_plan_gemm(7 * e->Vs_pad, H, 1);
    // activations hold a whole prefill pass (MTTS_PFCAP rows); split-K slabs: up to 8 of [MTTS_PFCAP][Npad] fp32
    size_t pmax = (size_t)8 * std::max(e->qkv_rows, round_up(H, 32));
    if (!e->f32) {
    TRY(dalloc(&e->partial, pmax * MTTS_PFCAP));
    TRY(dalloc(&e->partial2, (size_t)8 * round_up(H, 32) * MTTS_PFCAP));
    TRY(dalloc((uint16_t**)&e->x2, (size_t)MTTS_MAXR * H));
    TRY(dalloc((uint16_t**)&e->act_rm, (size_t)MTTS_MAXR * I));
    TRY(dalloc((uint16_t**)&e->x, (size_t)MTTS_PFCAP * H));
    TRY(dalloc((uint16_t**)&e->xn, (size_t)MTTS_PFCAP * H));
    TRY(dalloc((uint16_t**)&e->xh, (size_t)MTTS_RCAP * H));
    TRY(dalloc((uint16_t**)&e->hlast, (size_t)MTTS_RCAP * H));
    TRY(dalloc((uint16_t**)&e->attn_p, (size_t)MTTS_PFCAP * e->nq * MTTS_HD));
    TRY(dalloc((uint16_t**)&e->act_p, (size_t)MTTS_PFCAP * I));
    TRY(dalloc((uint16_t**)&e->qbuf, (size_t)MTTS_PFCAP * e->nq * MTTS_HD));
    TRY(dalloc((uint16_t**)&e->logits0, (size_t)MTTS_RCAP * e->V0_pad));        // channel-0 rows padded to 32 tokens
    TRY(dalloc((uint16_t**)&e->logits17, (size_t)MTTS_RCAP * 7 * e->Vs_pad));
    TRY(dalloc((uint16_t**)&e->join_logits0, (size_t)MTTS_MAXR * e->V0_pad));
    TRY(dalloc((uint16_t**)&e->join_logits17, (size_t)MTTS_MAXR * 7 * e->Vs_pad));
    }
    // KV pool
    e->max_pages = (c->max_seq_len + MTTS_PAGE - 1) / MTTS_PAGE + 1;
    e->total_pages = c->kv_pool_pages > 0 ? c->kv_pool_pages : e->max_pages * c->max_batch;
    if (c->kv_pool_pages < 0) return fail(MTTS_EINVAL, "kv_pool_pages must be >= 0");
    e->nchunks_max = (e->max_pages + ATT_PB - 1) / ATT_PB;
    e->layer_stride = (size_t)e->total_pages * e->nkv * MTTS_PAGE * MTTS_HD;
    if (e->f32) {
        TRY(dalloc(&e->kcache_f, e->layer_stride * e->L));
        TRY(dalloc(&e->vcache_f, e->layer_stride * e->L));
        TRY(dalloc(&e->scores_f, (size_t)MTTS_PF32CAP * e->nq * e->max_pages * MTTS_PAGE, false));
    } else {
        TRY(dalloc((uint16_t**)&e->kcache, e->layer_stride * e->L));
        TRY(dalloc((uint16_t**)&e->vcache, e->layer_stride * e->L));
        if (e->kv_pack) {
            // the second pool is an optimisation: when the card cannot hold it beside everything else (a very large
            // kv_pool_pages), the engine runs on bf16 pages alone instead of failing
            e->pk_layer_stride = (size_t)e->total_pages * e->nkv * MTTS_PKU * 64 * 16;
            size_t free_b = 0, total_b = 0;
            const size_t need = 2 * e->pk_layer_stride * e->L;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need + ((size_t)8 << 30) > free_b) {
                fprintf(stderr, "mtts: sealed KV pages off: %.1f GB needed, %.1f GB free (8 GB kept for the run)\n", need / 1e9, free_b / 1e9);
                e->kv_pack = 0;
            }
        }
        if (e->kv_pack) {
            TRY(dalloc((uint8_t**)&e->kpack, e->pk_layer_stride * e->L));
            TRY(dalloc((uint8_t**)&e->vpack, e->pk_layer_stride * e->L));
            TRY(dalloc(&e->d_seal_cnt, (size_t)e->L * 4));
            HIPCHK(hipHostMalloc((void**)&e->h_seal_cnt, (size_t)e->L * 4 * sizeof(unsigned long long)));
            memset(e->h_seal_cnt, 0, (size_t)e->L * 4 * sizeof(unsigned long long));
            e->pack_k_on.assign(e->L, 1);
            e->pack_v_on.assign(e->L, 1);
        }
    }
    TRY(dalloc(&e->d_page_table, (size_t)c->max_batch * e->max_pages));
    e->h_page_table.assign((size_t)c->max_batch * e->max_pages, 0);
    e->n_pages.assign(c->max_batch, 0);
    e->slot_live.assign(c->max_batch, 0);
    e->pending_edits.n = 0;
    pool_reset(e);
    if (!e->f32) {
    TRY(dalloc((uint16_t**)&e->scores, (size_t)MTTS_PFCAP * e->nq * e->max_pages * MTTS_PAGE));
    TRY(dalloc(&e->stats, (size_t)MTTS_PFCAP * e->nq * e->max_pages * 2));
    TRY(dalloc(&e->opart, (size_t)MTTS_PFCAP * e->nq * ((e->max_pages + ATT_PF - 1) / ATT_PF) * MTTS_HD));   // prefill chunking is the finer one
    }
    // state
    TRY(dalloc(&e->d_seqs, MTTS_RCAP));
    TRY(dalloc(&e->d_meta, MTTS_RCAP));
    TRY(dalloc(&e->d_ls, 1));
    HIPCHK(hipHostMalloc((void**)&e->h_ls, sizeof(LoopState)));
    memset(e->h_ls, 0, sizeof(LoopState));
    HIPCHK(hipHostMalloc((void**)&e->h_seqs, MTTS_RCAP * sizeof(SeqState)));
    TRY(dalloc(&e->d_decisions, MTTS_RCAP * 8));
    TRY(dalloc(&e->d_cur, MTTS_RCAP * 8));
    TRY(dalloc(&e->d_tf, MTTS_RCAP * 7 * 8));
    e->bm_words = (e->V0 + 31) / 32;
    TRY(dalloc(&e->d_bitmaps, (size_t)MTTS_RCAP * 8 * e->bm_words));
    TRY(dalloc(&e->d_scfg, 8));
    TRY(alloc_scratch(e->sscr, c->max_batch, e->V0));
    *out = e;
    return MTTS_OK;
}

static void drop_graphs(MttsEngine* e) {
    for (auto& g : e->graphs) hipGraphExecDestroy(g.exec);
    e->graphs.clear();
}

int32_t mtts_engine_destroy(MttsEngine* e) {
    if (!e) return MTTS_OK;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    for (auto& l : e->layers) {
        hipFree(l.wqkv); hipFree(l.wo); hipFree(l.wgu); hipFree(l.wd);
        hipFree(l.ln_in); hipFree(l.ln_post); hipFree(l.qn); hipFree(l.kn);
    }
    for (int c = 0; c < 8; ++c) hipFree(e->emb[c]);
    for (auto& l : e->lf) { hipFree(l.wqkv); hipFree(l.wo); hipFree(l.wgu); hipFree(l.wd); hipFree(l.ln_in); hipFree(l.ln_post); hipFree(l.qn); hipFree(l.kn); }
    for (int c = 0; c < 8; ++c) hipFree(e->embf[c]);
    {
        void* fp[] = {(void*)e->d_tables_f, e->final_norm_f, e->rope_cos_f, e->rope_sin_f, e->kcache_f, e->vcache_f, e->xf, e->xnf, e->qkvf,
                      e->qbuf_f, e->attnf, e->yf, e->guf, e->actf, e->hlast_f, e->scores_f};
        for (void* q : fp) if (q) hipFree(q);
    }
    void* ptrs[] = {e->partial2, e->x2, e->act_rm, e->head0, e->heads17, e->final_norm, e->rope_cos, e->rope_sin, (void*)e->d_tables, e->partial, e->x,
                    e->xn, e->xh, e->hlast, e->attn_p, e->act_p, e->qbuf, e->logits0, e->logits17, e->join_logits0, e->join_logits17, e->scores, e->stats,
                    e->opart, e->kcache, e->vcache, e->kpack, e->vpack, e->d_page_table, e->d_seqs, e->d_meta, e->d_ls, e->d_decisions,
                    e->d_cur, e->d_gen, e->d_declog, e->d_forced, e->d_tf, e->d_bitmaps, e->d_scfg, e->d_pf_tokens,
                    e->d_pf_meta};
    for (void* p : ptrs) if (p) hipFree(p);
    if (e->h_ls) hipHostFree(e->h_ls);
    if (e->h_seqs) hipHostFree(e->h_seqs);
    if (e->h_seal_cnt) hipHostFree(e->h_seal_cnt);
    if (e->d_seal_cnt) hipFree(e->d_seal_cnt);
    free_scratch(e->sscr);
    drop_graphs(e);
    if (e->cap_stream) hipStreamDestroy(e->cap_stream);
    for (int w = 0; w < PROF_N; ++w) for (auto& pr : e->ev[w]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    delete e;
    return MTTS_OK;
}

static bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// fp32 engine: plain row-major copies (q|k|v rows one after another, gate rows then up rows)
static int bind_weight_f32(MttsEngine* e, const char* name_c, const float* src, int64_t rows, int64_t cols, hipStream_t st) {
    std::string name(name_c);
    const int64_t H = e->H, I = e->I, D = MTTS_HD;
    auto put = [&](float* dst, int64_t r, int64_t c) -> int {
        const bool vec = (c == 1) && ((rows == r && cols == 1) || (rows == 1 && cols == r));
        if (!vec && !(rows == r && cols == c)) return fail(MTTS_EINVAL, "%s: expected [%lld,%lld] got [%lld,%lld]", name_c, (long long)r, (long long)c, (long long)rows, (long long)cols);
        HIPCHK(hipMemcpyAsync(dst, src, (size_t)r * c * 4, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    int ch = -1;
    if (sscanf(name_c, "model.embedding_list.%d.weight", &ch) == 1) {
        if (ch < 0 || ch > 7) return fail(MTTS_EINVAL, "bad channel in %s", name_c);
        TRY(put(e->embf[ch], ch == 0 ? e->V0 : e->Vs, H));
        e->emb_bound |= 1 << ch;
        return MTTS_OK;
    }
    if (name == "model.language_model.norm.weight") { TRY(put(e->final_norm_f, H, 1)); e->norm_bound = 1; return MTTS_OK; }
    if (name.find("lm_heads.") == 0 || name == "model.language_model.embed_tokens.weight") return MTTS_OK;
    int n = -1;
    char rest[128];
    if (sscanf(name_c, "model.language_model.layers.%d.%127s", &n, rest) == 2) {
        if (n < 0 || n >= e->L) return fail(MTTS_EINVAL, "layer index out of range in %s", name_c);
        auto& l = e->lf[n];
        Layer& lb = e->layers[n];
        std::string r(rest);
        if (r == "input_layernorm.weight") { TRY(put(l.ln_in, H, 1)); lb.bound |= 1; }
        else if (r == "post_attention_layernorm.weight") { TRY(put(l.ln_post, H, 1)); lb.bound |= 2; }
        else if (r == "self_attn.q_norm.weight") { TRY(put(l.qn, D, 1)); lb.bound |= 4; }
        else if (r == "self_attn.k_norm.weight") { TRY(put(l.kn, D, 1)); lb.bound |= 8; }
        else if (r == "self_attn.q_proj.weight") { TRY(put(l.wqkv, e->nq * D, H)); lb.bound |= 16; }
        else if (r == "self_attn.k_proj.weight") { TRY(put(l.wqkv + (size_t)e->nq * D * H, e->nkv * D, H)); lb.bound |= 32; }
        else if (r == "self_attn.v_proj.weight") { TRY(put(l.wqkv + (size_t)(e->nq + e->nkv) * D * H, e->nkv * D, H)); lb.bound |= 64; }
        else if (r == "self_attn.o_proj.weight") { TRY(put(l.wo, H, e->nq * D)); lb.bound |= 128; }
        else if (r == "mlp.gate_proj.weight") { TRY(put(l.wgu, I, H)); lb.bound |= 256; }
        else if (r == "mlp.up_proj.weight") { TRY(put(l.wgu + (size_t)I * H, I, H)); lb.bound |= 512; }
        else if (r == "mlp.down_proj.weight") { TRY(put(l.wd, H, I)); lb.bound |= 1024; }
        else return fail(MTTS_EINVAL, "unknown tensor %s", name_c);
        return MTTS_OK;
    }
    return fail(MTTS_EINVAL, "unknown tensor %s", name_c);
}

int32_t mtts_bind_weight(MttsEngine* e, const char* name_c, const void* src, int64_t rows, int64_t cols, void* stream) {
    if (!e || !name_c || !src) return fail(MTTS_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->device));
    drop_graphs(e);
    hipStream_t st = S(stream);
    if (e->f32) return bind_weight_f32(e, name_c, (const float*)src, rows, cols, st);
    std::string name(name_c);
    const int H = e->H, I = e->I, D = MTTS_HD;
    auto expect = [&](int64_t r, int64_t c) { return rows == r && cols == c; };
    auto copyvec = [&](void* dst, int64_t n) -> int {
        if (!(rows == n && cols == 1) && !(rows == 1 && cols == n)) return fail(MTTS_EINVAL, "%s: expected vector of %lld", name_c, (long long)n);
        HIPCHK(hipMemcpyAsync(dst, src, n * 2, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    int ch = -1;
    if (sscanf(name_c, "model.embedding_list.%d.weight", &ch) == 1 && ends_with(name, ".weight") && name.find("embedding_list") != std::string::npos) {
        if (ch < 0 || ch > 7) return fail(MTTS_EINVAL, "bad channel in %s", name_c);
        int64_t V = ch == 0 ? e->V0 : e->Vs;
        if (!expect(V, H)) return fail(MTTS_EINVAL, "%s: expected [%lld,%d] got [%lld,%lld]", name_c, (long long)V, H, (long long)rows, (long long)cols);
        HIPCHK(hipMemcpyAsync(e->emb[ch], src, (size_t)V * H * 2, hipMemcpyDeviceToDevice, st));
        // the head is tied to the embedding (modeling_asteroid.py:315-317)
        if (ch == 0) launch_pack_weight(src, e->head0, V, H, e->V0_pad, 1, 0, st);
        else launch_pack_weight(src, e->heads17, V, H, 7 * e->Vs_pad, 1, (ch - 1) * e->Vs_pad, st);
        e->emb_bound |= 1 << ch;
        return MTTS_OK;
    }
    if (name == "model.language_model.norm.weight") { TRY(copyvec(e->final_norm, H)); e->norm_bound = 1; return MTTS_OK; }
    if (name.find("lm_heads.") == 0 || name == "model.language_model.embed_tokens.weight") return MTTS_OK;  // tied / unused
    int n = -1;
    char rest[128];
    if (sscanf(name_c, "model.language_model.layers.%d.%127s", &n, rest) == 2) {
        if (n < 0 || n >= e->L) return fail(MTTS_EINVAL, "layer index out of range in %s", name_c);
        Layer& l = e->layers[n];
        std::string r(rest);
        if (r == "input_layernorm.weight") { TRY(copyvec(l.ln_in, H)); l.bound |= 1; }
        else if (r == "post_attention_layernorm.weight") { TRY(copyvec(l.ln_post, H)); l.bound |= 2; }
        else if (r == "self_attn.q_norm.weight") { TRY(copyvec(l.qn, D)); l.bound |= 4; }
        else if (r == "self_attn.k_norm.weight") { TRY(copyvec(l.kn, D)); l.bound |= 8; }
        else if (r == "self_attn.q_proj.weight") {
            if (!expect(e->nq * D, H)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wqkv, rows, H, e->qkv_rows, 1, 0, st); l.bound |= 16;
        } else if (r == "self_attn.k_proj.weight") {
            if (!expect(e->nkv * D, H)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wqkv, rows, H, e->qkv_rows, 1, e->nq * D, st); l.bound |= 32;
        } else if (r == "self_attn.v_proj.weight") {
            if (!expect(e->nkv * D, H)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wqkv, rows, H, e->qkv_rows, 1, (e->nq + e->nkv) * D, st); l.bound |= 64;
        } else if (r == "self_attn.o_proj.weight") {
            if (!expect(H, e->nq * D)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wo, rows, e->nq * D, round_up(H, 32), 1, 0, st); l.bound |= 128;
        } else if (r == "mlp.gate_proj.weight") {
            if (!expect(I, H)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wgu, rows, H, 2 * I, 2, 0, st); l.bound |= 256;
        } else if (r == "mlp.up_proj.weight") {
            if (!expect(I, H)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wgu, rows, H, 2 * I, 2, 1, st); l.bound |= 512;
        } else if (r == "mlp.down_proj.weight") {
            if (!expect(H, I)) return fail(MTTS_EINVAL, "%s: bad shape", name_c);
            launch_pack_weight(src, l.wd, rows, I, round_up(H, 32), 1, 0, st); l.bound |= 1024;
        } else return fail(MTTS_EINVAL, "unknown tensor %s", name_c);
        HIPCHK(hipGetLastError());
        return MTTS_OK;
    }
    return fail(MTTS_EINVAL, "unknown tensor %s", name_c);
}

int32_t mtts_bind_rope(MttsEngine* e, const void* cosb, const void* sinb, int32_t rows, void* stream) {
    if (!e || !cosb || !sinb || rows < 1) return fail(MTTS_EINVAL, "bad rope table");
    HIPCHK(hipSetDevice(e->device));
    drop_graphs(e);
    if (e->f32) {           // fp32 tables [rows][64], as Qwen3RotaryEmbedding leaves them before the cast to the model dtype
        if (e->rope_cos_f) { hipFree(e->rope_cos_f); hipFree(e->rope_sin_f); }
        TRY(dalloc(&e->rope_cos_f, (size_t)rows * 64, false));
        TRY(dalloc(&e->rope_sin_f, (size_t)rows * 64, false));
        HIPCHK(hipMemcpyAsync(e->rope_cos_f, cosb, (size_t)rows * 256, hipMemcpyDeviceToDevice, S(stream)));
        HIPCHK(hipMemcpyAsync(e->rope_sin_f, sinb, (size_t)rows * 256, hipMemcpyDeviceToDevice, S(stream)));
        e->rope_rows = rows;
        return MTTS_OK;
    }
    if (e->rope_cos) { hipFree(e->rope_cos); hipFree(e->rope_sin); }
    TRY(dalloc((uint16_t**)&e->rope_cos, (size_t)rows * 64, false));
    TRY(dalloc((uint16_t**)&e->rope_sin, (size_t)rows * 64, false));
    HIPCHK(hipMemcpyAsync(e->rope_cos, cosb, (size_t)rows * 128, hipMemcpyDeviceToDevice, S(stream)));
    HIPCHK(hipMemcpyAsync(e->rope_sin, sinb, (size_t)rows * 128, hipMemcpyDeviceToDevice, S(stream)));
    e->rope_rows = rows;
    return MTTS_OK;
}

int32_t mtts_weights_ready(MttsEngine* e) {
    if (!e) return fail(MTTS_EINVAL, "null engine");
    if (e->emb_bound != 0xff) return fail(MTTS_ESTATE, "embedding tables missing (mask %x)", e->emb_bound);
    if (!e->norm_bound) return fail(MTTS_ESTATE, "final norm missing");
    if (!e->rope_rows) return fail(MTTS_ESTATE, "rope table missing");
    for (int n = 0; n < e->L; ++n)
        if (e->layers[n].bound != 2047) return fail(MTTS_ESTATE, "layer %d incomplete (mask %x)", n, e->layers[n].bound);
    return MTTS_OK;
}

// ---- profiling helpers ------------------------------------------------------------
static void prof_begin(MttsEngine* e, int which, hipStream_t st, hipEvent_t* a) {
    if (!e->prof) return;
    hipEvent_t s0, s1;
    hipEventCreate(&s0); hipEventCreate(&s1);
    hipEventRecord(s0, st);
    e->ev[which].push_back({s0, s1});
    *a = s1;
}
static void prof_end(MttsEngine* e, hipStream_t st, hipEvent_t a) {
    if (!e->prof) return;
    hipEventRecord(a, st);
}

// ---- decode step for 1..SMALL_RP dialogues: six launches per layer instead of nine ---------------------------------
// qkv GEMM [prologue: residual + slabs + input norm] -> scores -> P.V -> o_proj [prologue: chunk sum] ->
// gate/up + SwiGLU [prologue: residual + slabs + post-attention norm] -> down_proj; the heads take the final norm as
// their prologue.  The residual stream alternates between two buffers (a prologue's block (0,0) writes x' while the
// other blocks still read x); the qkv slabs live in `partial` (the fused attention epilogue reads them), the o_proj /
// down_proj slabs in `partial2`.  Same arithmetic as forward_rows, operation for operation.
// a new run: sealed reads everywhere, counts from zero
static int pack_policy_reset(MttsEngine* e, hipStream_t st) {
    if (!e->d_seal_cnt) return 0;
    HIPCHK(hipMemsetAsync(e->d_seal_cnt, 0, (size_t)e->L * 4 * sizeof(unsigned long long), st));
    bool changed = false;
    for (int n = 0; n < e->L; ++n) { changed |= !e->pack_k_on[n] || !e->pack_v_on[n]; e->pack_k_on[n] = 1; e->pack_v_on[n] = 1; }
    if (changed) drop_graphs(e);
    return 0;
}
// this layer's sealed pools, each null where the read policy (or MTTS_KV_PACK=0) says bf16 pages
static KvPack layer_pack(MttsEngine* e, int n, int pages_bound) {
    KvPack pk{nullptr, nullptr};
    // few rows x pages: the passes are latency-bound and the unpack sits on the critical path (B=1 at 2 k: +4 %; break-even
    // at 8 rows x 64 pages, -6 % at 16 x 64: profiles/r03_kv_pack_ab.txt)
    if (e->B * pages_bound < e->pack_min_work) return pk;
    if (e->kpack && e->pack_k_on[n]) pk.k = (uint8_t*)e->kpack + e->pk_layer_stride * n;
    if (e->vpack && e->pack_v_on[n]) pk.v = (uint8_t*)e->vpack + e->pk_layer_stride * n;
    return pk;
}
static bool small_path_fits(MttsEngine* e) {
    const int H = e->H;
    if (H > 8192 || H % 8 || e->I % 8) return false;
    const int lim = 64 * 1024 - 33 * 1024;            // default dynamic-LDS budget next to the kernel's static 32.1 KiB
    return mtts_small_lds_bytes(e->p_qkv, H, PRO_NORM) <= lim && mtts_small_lds_bytes(e->p_d, e->I, PRO_ROWS) <= lim &&
           mtts_small_lds_bytes(e->p_o, e->nq * MTTS_HD, PRO_COMBINE) <= lim;
}
static int forward_small(MttsEngine* e, const RowMeta* d_meta, int pages_bound, hipStream_t st, int64_t kv_tokens_hint) {
    const int H = e->H, I = e->I, nq = e->nq, nkv = e->nkv, Hp = round_up(H, 32), R = MTTS_MAXR;
    const float eps = e->cfg.rms_norm_eps;
    const float scale = 1.0f / sqrtf((float)MTTS_HD);
    uint16_t* xa = (uint16_t*)e->x;                    // embed_norm has left the embedding sum here
    uint16_t* xb = (uint16_t*)e->x2;
    SmallPro base{};
    base.rows = e->B; base.eps = eps; base.slab_npad = Hp;
    for (int n = 0; n < e->L; ++n) {
        Layer& l = e->layers[n];
        uint16_t* kc = (uint16_t*)e->kcache + e->layer_stride * n;
        uint16_t* vc = (uint16_t*)e->vcache + e->layer_stride * n;
        SmallPro pq = base;                            // input norm (+ the previous layer's down_proj slabs)
        pq.x_in = xa; pq.x_out = xb; pq.slabs = e->partial2; pq.ksplit = n ? e->p_d.ksplit : 0; pq.norm_w = (const uint16_t*)l.ln_in;
        launch_gemv_small(EPI_PARTIAL, PRO_NORM, e->p_qkv, l.wqkv, H, e->qkv_rows, e->qkv_rows, e->partial, nullptr, pq, st);
        const bool fused = e->B * pages_bound <= e->fuse_qkv_max;
        const QkvFuse fz{e->partial, e->p_qkv.ksplit, e->qkv_rows, (const uint16_t*)l.qn, (const uint16_t*)l.kn,
                         (const uint16_t*)e->rope_cos, (const uint16_t*)e->rope_sin, eps};
        const KvPack pk = layer_pack(e, n, pages_bound);
        if (!fused)
            launch_qkv_post(e->partial, e->p_qkv.ksplit, e->qkv_rows, d_meta, l.qn, l.kn, e->rope_cos, e->rope_sin, e->qbuf,
                            kc, vc, e->d_page_table, e->max_pages, e->total_pages, R, nq, nkv, eps, st);
        for (int phase = 1; phase <= 2; ++phase) {
            hipEvent_t ev = nullptr;
            prof_begin(e, phase == 1 ? PROF_SCORES : PROF_PV, st, &ev);
            if (launch_attn(e->qbuf, kc, vc, e->d_page_table, d_meta, e->scores, e->stats, e->opart, e->attn_p, R,
                            pages_bound, e->max_pages, e->total_pages, e->nchunks_max, nq, nkv, scale,
                            fused ? &fz : nullptr, phase, st, (pk.k || pk.v) ? &pk : nullptr))
                return fail(MTTS_EINVAL, "attention group size not built");
            prof_end(e, st, ev);
        }
        if (e->prof) {
            e->prof_bytes[PROF_SCORES] += kv_tokens_hint * nkv * MTTS_HD * 2;
            e->prof_bytes[PROF_PV] += kv_tokens_hint * nkv * MTTS_HD * 2;
        }
        SmallPro po = base;                            // o_proj: the chunk partials of P.V are summed in its prologue
        po.opart = e->opart; po.meta = d_meta; po.nchunks_max = e->nchunks_max; po.nq = nq; po.pages_per_chunk = ATT_PB;
        launch_gemv_small(EPI_PARTIAL, PRO_COMBINE, e->p_o, l.wo, nq * MTTS_HD, Hp, Hp, e->partial2, nullptr, po, st);
        SmallPro pg = base;                            // post-attention norm (+ the o_proj slabs)
        pg.x_in = xb; pg.x_out = xa; pg.slabs = e->partial2; pg.ksplit = e->p_o.ksplit; pg.norm_w = (const uint16_t*)l.ln_post;
        launch_gemv_small(EPI_SILU_RM, PRO_NORM, e->p_gu, l.wgu, H, 2 * I, 2 * I, nullptr, (uint16_t*)e->act_rm, pg, st);
        SmallPro pd = base;
        pd.xrows = (const uint16_t*)e->act_rm;
        launch_gemv_small(EPI_PARTIAL, PRO_ROWS, e->p_d, l.wd, I, Hp, Hp, e->partial2, nullptr, pd, st);
    }
    if (e->kpack)                                      // rows whose token completed a KV page: seal it (all layers)
        launch_kv_seal_rows(e->kcache, e->vcache, e->kpack, e->vpack, e->d_page_table, d_meta, R, e->max_pages, e->total_pages, nkv, e->L, e->d_seal_cnt, st);
    SmallPro ph = base;                                // final norm (+ the last down_proj slabs) in front of the 8 heads
    ph.x_in = xa; ph.x_out = nullptr; ph.slabs = e->partial2; ph.ksplit = e->p_d.ksplit; ph.norm_w = (const uint16_t*)e->final_norm;
    // (same plan as the general path: a different K partition over the waves would change the fp32 sums, and with them
    //  the bit-identity of a dialogue's logits across batch sizes; 4 waves would be 1 % faster at B=1)
    launch_gemv_small(EPI_BF16, PRO_NORM, e->p_h0, e->head0, H, e->V0_pad, e->V0, nullptr, (uint16_t*)e->logits0, ph, st);
    launch_gemv_small(EPI_BF16, PRO_NORM, e->p_h17, e->heads17, H, 7 * e->Vs_pad, 7 * e->Vs_pad, nullptr, (uint16_t*)e->logits17, ph, st);
    HIPCHK(hipGetLastError());
    return MTTS_OK;
}

// ---- one forward pass: R rows = decode rows (<= MTTS_RCAP, one dialogue each) or a prefill pass (<= MTTS_PFCAP) ----
// heads: 0 none, 1 from xn (rows are sequences: decode), 2 from hlast (end of prefill)
static int forward_rows(MttsEngine* e, const int32_t* d_tokens, const RowMeta* d_meta, int R, int pages_bound,
                        int heads, hipStream_t st, int64_t kv_tokens_hint) {
    const int H = e->H, I = e->I, nq = e->nq, nkv = e->nkv;
    const float eps = e->cfg.rms_norm_eps;
    const float scale = 1.0f / sqrtf((float)MTTS_HD);
    const int Hp = round_up(H, 32);
    const int mb = (R + 31) / 32;                    // activation row tiles sharing each weight stream
    // prefill passes always take the tiled GEMM with a split-K that depends on the shape only (chosen for a
    // 1024-row pass: good from one dialogue's prompt up to a full pass): a prompt's hidden states then do not
    // depend on how many rows (other dialogues) share its pass
    const bool tiled = heads != 1;
    if (e->f32) return forward_rows_f32(e, d_tokens, d_meta, heads == 1 ? e->B : R, heads, st);
    launch_embed_norm(d_tokens, d_meta, e->d_tables, e->layers[0].ln_in, e->x, e->xn, R, H, eps, st);
    if (heads == 1 && e->B <= e->small_rows && small_path_fits(e)) return forward_small(e, d_meta, pages_bound, st, kv_tokens_hint);
    for (int n = 0; n < e->L; ++n) {
        Layer& l = e->layers[n];
        uint16_t* kc = (uint16_t*)e->kcache + e->layer_stride * n;
        uint16_t* vc = (uint16_t*)e->vcache + e->layer_stride * n;
        const int ks_qkv = tiled ? mtts_tile_ksplit(e->qkv_rows, H, 1024) : e->p_qkv.ksplit;
        const int ks_o = tiled ? mtts_tile_ksplit(Hp, nq * MTTS_HD, 1024) : e->p_o.ksplit;
        const int ks_d = tiled ? mtts_tile_ksplit(Hp, I, 1024) : e->p_d.ksplit;
        if (tiled) launch_gemm_tile(EPI_PARTIAL, R, ks_qkv, l.wqkv, e->xn, H, e->qkv_rows, e->qkv_rows, e->partial, nullptr, st);
        else launch_gemm(EPI_PARTIAL, mb, e->p_qkv, l.wqkv, e->xn, H, e->qkv_rows, e->qkv_rows, e->partial, nullptr, st);
        // decode rows (one dialogue each): the q/k/v epilogue runs inside the attention kernels; prefill passes
        // need every K/V row of the pass in the cache before any of its attention runs, so they keep the launch
        // (fused where it pays: every attention block repeats the q epilogue, which costs more than the saved launch
        // once rows x pages is large -- break-even between 32 x 64 and 64 x 64 rows x pages since round 3; the results are bit-identical)
        const bool fused = heads == 1 && e->B * pages_bound <= e->fuse_qkv_max;
        const QkvFuse fz{e->partial, ks_qkv, e->qkv_rows, (const uint16_t*)l.qn, (const uint16_t*)l.kn,
                         (const uint16_t*)e->rope_cos, (const uint16_t*)e->rope_sin, eps};
        if (!fused)
            launch_qkv_post(e->partial, ks_qkv, e->qkv_rows, d_meta, l.qn, l.kn, e->rope_cos, e->rope_sin, e->qbuf,
                            kc, vc, e->d_page_table, e->max_pages, e->total_pages, R, nq, nkv, eps, st);
        // decode rows are one dialogue each (phases 1,2); prefill tiles are 32 consecutive positions of one
        // dialogue and share their K/V pages (phases 11,12,13: chunks of ATT_PF pages)
        const int ph0 = (heads != 1 && pages_bound >= e->pf_mfma_pages) ? 10 : 0;
        // decode rows read complete pages in their sealed form; prefill rows (a page may be completed by the pass
        // itself) read the bf16 pages
        const KvPack pk = layer_pack(e, n, pages_bound);
        for (int phase = 1; phase <= 3; ++phase) {
            hipEvent_t ev = nullptr;
            if (phase < 3) prof_begin(e, phase == 1 ? PROF_SCORES : PROF_PV, st, &ev);
            if (launch_attn(e->qbuf, kc, vc, e->d_page_table, d_meta, e->scores, e->stats, e->opart, e->attn_p, R,
                            pages_bound, e->max_pages, e->total_pages, e->nchunks_max, nq, nkv, scale,
                            fused ? &fz : nullptr, ph0 + phase, st, (heads == 1 && (pk.k || pk.v)) ? &pk : nullptr))
                return fail(MTTS_EINVAL, "attention group size not built");
            if (phase < 3) prof_end(e, st, ev);
        }
        if (e->prof) {   // algorithmic bytes: one K (or V) row of 128 bf16 per kv head per cached token
            e->prof_bytes[PROF_SCORES] += kv_tokens_hint * nkv * MTTS_HD * 2;
            e->prof_bytes[PROF_PV] += kv_tokens_hint * nkv * MTTS_HD * 2;
        }
        if (tiled) launch_gemm_tile(EPI_PARTIAL, R, ks_o, l.wo, e->attn_p, nq * MTTS_HD, Hp, Hp, e->partial, nullptr, st);
        else launch_gemm(EPI_PARTIAL, mb, e->p_o, l.wo, e->attn_p, nq * MTTS_HD, Hp, Hp, e->partial, nullptr, st);
        launch_resid_norm(e->partial, ks_o, Hp, e->x, l.ln_post, e->xn, nullptr, d_meta, R, H, eps, st);
        if (tiled) {
            launch_gemm_tile(EPI_SILU, R, 1, l.wgu, e->xn, H, 2 * I, 2 * I, nullptr, (uint16_t*)e->act_p, st);
            launch_gemm_tile(EPI_PARTIAL, R, ks_d, l.wd, e->act_p, I, Hp, Hp, e->partial, nullptr, st);
        } else {
            launch_gemm(EPI_SILU, mb, e->p_gu, l.wgu, e->xn, H, 2 * I, 2 * I, nullptr, (uint16_t*)e->act_p, st);
            launch_gemm(EPI_PARTIAL, mb, e->p_d, l.wd, e->act_p, I, Hp, Hp, e->partial, nullptr, st);
        }
        const bool lastl = (n == e->L - 1);
        const void* nw = lastl ? e->final_norm : e->layers[n + 1].ln_in;
        launch_resid_norm(e->partial, ks_d, Hp, e->x, nw, e->xn, lastl ? e->hlast : nullptr, d_meta, R, H, eps, st);
    }
    if (e->kpack)                                      // rows whose token completed a KV page: seal it (all layers)
        launch_kv_seal_rows(e->kcache, e->vcache, e->kpack, e->vpack, e->d_page_table, d_meta, R, e->max_pages, e->total_pages, nkv, e->L, e->d_seal_cnt, st);
    if (heads) {
        const void* xin = e->xn;
        int hmb = mb;
        if (heads == 2) {
            hmb = (e->B + 31) / 32;
            launch_pack_rows(e->hlast, e->xh, e->B, H, hmb, st);
            xin = e->xh;
        }
        launch_gemm(EPI_BF16, hmb, e->p_h0, e->head0, xin, H, e->V0_pad, e->V0, nullptr, (uint16_t*)e->logits0, st);
        launch_gemm(EPI_BF16, hmb, e->p_h17, e->heads17, xin, H, 7 * e->Vs_pad, 7 * e->Vs_pad, nullptr, (uint16_t*)e->logits17, st);
    }
    HIPCHK(hipGetLastError());
    return MTTS_OK;
}

// generated-token storage [slot][gen_cap][8] (+ decision log and forced rows of the same shape)
static int ensure_gen_storage(MttsEngine* e, int steps) {
    if (steps <= e->gen_cap) return 0;
    if (e->d_gen) { hipFree(e->d_gen); hipFree(e->d_declog); hipFree(e->d_forced); }
    drop_graphs(e);                      // captured steps hold the old pointers
    e->gen_cap = steps;
    const size_t n = (size_t)e->cfg.max_batch * steps * 8;
    TRY(dalloc(&e->d_gen, n));
    TRY(dalloc(&e->d_declog, n));
    TRY(dalloc(&e->d_forced, n, false));
    return 0;
}

// ---- begin: parse prompt, allocate pages, prefill --------------------------------------
int32_t mtts_begin(MttsEngine* e, const int64_t* ids, const uint8_t* mask, int32_t B, int32_t T, int32_t max_length,
                   const MttsSamplerCfg* sampler, uint64_t seed, void* stream) {
    if (!e || !ids || !mask || !sampler) return fail(MTTS_EINVAL, "null argument");
    TRY(mtts_weights_ready(e));
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = S(stream);
    if (B < 1 || B > e->cfg.max_batch) return fail(MTTS_EINVAL, "batch %d exceeds max_batch %d", B, e->cfg.max_batch);
    if (T < 8) return fail(MTTS_EINVAL, "T must be >= 8 (delay pattern adds 7 slots)");
    const int base = T - 7;
    if (max_length <= base) return fail(MTTS_EINVAL, "max_length %d leaves no room to generate (prompt slots %d)", max_length, base);
    // a dialogue whose EOS falls within 7 steps of max_length keeps stepping until its flush is through
    // (`unfinished | needs_additional_steps > 0`, modeling_asteroid.py:165-168)
    int max_steps = max_length - base + LINGER_STEPS;          // the least the engine must have room for; widened below
    e->B = B; e->T = T; e->base_length = base; e->max_length = max_length;
    e->seed = seed; e->steps_issued = 0; e->has_forced = false;
    e->n_real.assign(B, 0);
    e->max_real = 0;
    // attention_mask must be the left-padded form rpadding() produces (generation_utils.py:221-237)
    std::vector<int> pad(B, 0);
    for (int b = 0; b < B; ++b) {
        int p = 0;
        while (p < base && !mask[(size_t)b * T + p]) ++p;
        for (int t = p; t < base; ++t)
            if (!mask[(size_t)b * T + t]) return fail(MTTS_EINVAL, "attention_mask of row %d is not left-padded", b);
        if (p >= base) return fail(MTTS_EINVAL, "row %d has no real token in the first T-7 slots", b);
        pad[b] = p;
        e->n_real[b] = base - p;
        e->max_real = std::max(e->max_real, base - p);
    }
    // pages: every earlier run has been synchronised by its caller or is ordered before us on `st`; the prompts'
    // pages are taken now, the rest on demand as the dialogues grow (issue_steps)
    for (int b = 0; b < e->cfg.max_batch; ++b) { pool_release(e, b); e->slot_live[b] = b < B; }
    e->pending_edits.n = 0;            // edits queued by a run that failed (MTTS_ENOMEM before its flush) belong to released pages
    TRY(pack_policy_reset(e, st));
    for (int b = 0; b < B; ++b) {
        const int need = (e->n_real[b] + max_steps + MTTS_PAGE - 1) / MTTS_PAGE;
        if (need > e->max_pages) return fail(MTTS_ENOMEM, "row %d needs %d KV pages, a sequence holds at most %d (max_seq_len %d)", b, need, e->max_pages, e->cfg.max_seq_len);
        TRY(pool_grow(e, b, (e->n_real[b] + MTTS_PAGE - 1) / MTTS_PAGE, st));
    }
    TRY(pool_flush(e, st));
    if (e->max_real + max_steps > e->rope_rows)
        return fail(MTTS_EINVAL, "rope table has %d rows, need %d", e->rope_rows, e->max_real + max_steps);
    // room for chained resurrections (linger_bound) as far as the page-table width and the RoPE table allow
    max_steps = std::max(max_steps, std::min(max_length - base + linger_bound(B),
                                             std::min(e->max_pages * MTTS_PAGE, e->rope_rows) - e->max_real));
    e->max_steps = max_steps;
    // generation buffers
    TRY(ensure_gen_storage(e, max_steps));
    // flattened prefill rows; every dialogue starts on a 32-row tile boundary so that a tile holds consecutive
    // positions of one dialogue (the prefill attention kernels share K/V pages across a tile); filler rows are idle
    size_t Mtot = 0;
    for (int b = 0; b < B; ++b) Mtot += (size_t)round_up(e->n_real[b], MTTS_MAXR);
    size_t Mpad = (Mtot + MTTS_RCAP - 1) / MTTS_RCAP * MTTS_RCAP;
    std::vector<int32_t> toks(Mpad * 8, 0);
    std::vector<RowMeta> metas(Mpad, RowMeta{-1, 0, 0, 0});
    size_t r = 0;
    for (int b = 0; b < B; r = (r + MTTS_MAXR - 1) / MTTS_MAXR * MTTS_MAXR, ++b)
        for (int i = 0; i < e->n_real[b]; ++i, ++r) {
            const int64_t* src = ids + ((size_t)b * T + pad[b] + i) * 8;
            for (int c = 0; c < 8; ++c) {
                int64_t t = src[c];
                int64_t V = c == 0 ? e->V0 : e->Vs;
                if (t < 0 || t >= V) return fail(MTTS_EINVAL, "token %lld out of range on channel %d", (long long)t, c);
                toks[r * 8 + c] = (int32_t)t;
            }
            metas[r] = RowMeta{b, i, i == e->n_real[b] - 1 ? 1 : 0, 0};
        }
    if (Mpad > e->pf_cap_rows) {
        if (e->d_pf_tokens) { hipFree(e->d_pf_tokens); hipFree(e->d_pf_meta); }
        TRY(dalloc(&e->d_pf_tokens, Mpad * 8, false));
        TRY(dalloc(&e->d_pf_meta, Mpad, false));
        e->pf_cap_rows = Mpad;
    }
    HIPCHK(hipMemcpyAsync(e->d_pf_tokens, toks.data(), Mpad * 8 * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->d_pf_meta, metas.data(), Mpad * sizeof(RowMeta), hipMemcpyHostToDevice, st));
    // history bitmaps (HF repetition penalty sees the whole channel incl. pads: modeling_asteroid.py:129)
    {
        std::vector<uint32_t> bm((size_t)MTTS_RCAP * 8 * e->bm_words, 0u);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < base; ++t)
                for (int c = 0; c < 8; ++c) {
                    int64_t tk = ids[((size_t)b * T + t) * 8 + c];
                    if (tk >= 0 && tk < (int64_t)e->bm_words * 32) bm[((size_t)b * 8 + c) * e->bm_words + (tk >> 5)] |= 1u << (tk & 31);
                }
        HIPCHK(hipMemcpyAsync(e->d_bitmaps, bm.data(), bm.size() * 4, hipMemcpyHostToDevice, st));
        // teacher-forcing tail tf_inputs[:, base+s, :] for s = 0..6 (modeling_asteroid.py:143-145)
        std::vector<int32_t> tf((size_t)MTTS_RCAP * 7 * 8, 0);
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < 7; ++s)
                for (int c = 0; c < 8; ++c) {
                    int64_t tk = ids[((size_t)b * T + base + s) * 8 + c];
                    if (tk < 0 || tk >= (c == 0 ? e->V0 : e->Vs)) return fail(MTTS_EINVAL, "token %lld out of range on channel %d (delayed tail)", (long long)tk, c);
                    tf[((size_t)b * 7 + s) * 8 + c] = (int32_t)tk;
                }
        HIPCHK(hipMemcpyAsync(e->d_tf, tf.data(), tf.size() * 4, hipMemcpyHostToDevice, st));
        std::vector<SeqState> ss(MTTS_RCAP, SeqState{-1, 0, 0, 0, 0, 0, 0, 0, 0});
        if (!e->next_row_ids.empty() && (int)e->next_row_ids.size() != B)
            return fail(MTTS_EINVAL, "mtts_set_row_ids gave %d ids, the batch has %d rows", (int)e->next_row_ids.size(), B);
        for (int b = 0; b < B; ++b)
            ss[b] = SeqState{-1, 1, e->n_real[b], 0, base, max_length, e->next_row_ids.empty() ? b : e->next_row_ids[b], 1, seed};
        e->next_row_ids.clear();
        HIPCHK(hipMemcpyAsync(e->d_seqs, ss.data(), ss.size() * sizeof(SeqState), hipMemcpyHostToDevice, st));
        LoopState ls{0, 0, 0, B, 0, e->gen_cap, e->forced_draw, e->f32 ? 1 : 0};
        e->continuous = false;
        e->join_step.assign(MTTS_RCAP, 0);
        HIPCHK(hipMemcpyAsync(e->d_ls, &ls, sizeof(ls), hipMemcpyHostToDevice, st));
        *e->h_ls = ls;
        std::vector<RowMeta> dm(MTTS_RCAP, RowMeta{-1, 0, 0, 0});
        HIPCHK(hipMemcpyAsync(e->d_meta, dm.data(), dm.size() * sizeof(RowMeta), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->d_scfg, sampler, 8 * sizeof(MttsSamplerCfg), hipMemcpyHostToDevice, st));
        e->ch0_sampled = sampler[0].do_sample ? 1 : 0;
        HIPCHK(hipMemsetAsync(e->sscr.hist, 0, (size_t)e->cfg.max_batch * 2048 * 4, st));
        HIPCHK(hipMemsetAsync(e->sscr.cand_n, 0, (size_t)e->cfg.max_batch * 4, st));
        HIPCHK(hipMemsetAsync(e->sscr.overflow, 0, (size_t)e->cfg.max_batch * 4, st));
        HIPCHK(hipStreamSynchronize(st));   // host vectors above go out of scope
    }
    // prefill: chunks of 32 flattened tokens; K/V of a chunk are written before its attention runs
    const int pages_bound = (e->max_real + MTTS_PAGE - 1) / MTTS_PAGE;
    const size_t pfcap = e->f32 ? MTTS_PF32CAP : MTTS_PFCAP;
    for (size_t off = 0; off < Mpad; off += pfcap) {
        const int rows = (int)std::min<size_t>(pfcap, Mpad - off);     // multiple of MTTS_RCAP
        const bool lastc = off + rows >= Mpad;
        TRY(forward_rows(e, e->d_pf_tokens + off * 8, e->d_pf_meta + off, rows, pages_bound, lastc ? 2 : 0, st, 0));
    }
    e->began = true;
    return MTTS_OK;
}

// One decode step: sample from the previous logits, advance the per-dialogue state machine, run the stack.
// Nothing in it depends on the host (step counters, lengths and the stop flag live on the device), so the same
// launches repeat until the KV page bound grows: they are captured once per bound into a hipGraph and replayed --
// a dependent kernel costs ~1.5 us inside a graph against ~2.9 us launched on a stream (tools/latency_probe.hip).
static int step_body(MttsEngine* e, int pages_bound, hipStream_t st, int64_t kvtok) {
    launch_sample(e->logits0, e->logits17, e->V0, e->Vs, e->Vs_pad, e->d_bitmaps, e->bm_words, e->d_scfg, e->d_ls,
                  e->d_seqs, e->seed, e->d_decisions, &e->d_ls->error, e->B, e->sscr, e->ch0_sampled,
                  full_cap_for(e->V0), st);
    launch_update(e->d_decisions, e->d_declog, e->has_forced ? e->d_forced : nullptr, e->d_tf, e->d_gen, e->d_cur,
                  e->d_seqs, e->d_meta, e->d_bitmaps, e->bm_words, e->d_ls, e->cfg.eos_token_id,
                  e->cfg.speech_pad_token, e->cfg.speech_range_lo, e->cfg.speech_range_hi, st);
    return forward_rows(e, e->d_cur, e->d_meta, round_up(e->B, 32), pages_bound, 1, st, kvtok);
}

static int step_graph(MttsEngine* e, int pages, hipGraphExec_t* out) {
    const int forced = e->has_forced ? 1 : 0;
    for (auto& g : e->graphs)
        if (g.B == e->B && g.pages == pages && g.forced == forced && g.ch0 == e->ch0_sampled) {
            *out = g.exec;
            return MTTS_OK;
        }
    if (e->graphs.size() >= 256) drop_graphs(e);
    if (!e->cap_stream) HIPCHK(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed));
    int rc = step_body(e, pages, e->cap_stream, 0);
    hipError_t ce = hipStreamEndCapture(e->cap_stream, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    HIPCHK(ce);
    hipGraphExec_t exec = nullptr;
    hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    HIPCHK(ie);
    e->graphs.push_back({e->B, pages, forced, e->ch0_sampled, exec});
    *out = exec;
    return MTTS_OK;
}

static int issue_steps(MttsEngine* e, int n, hipStream_t st) {
    for (int i = 0; i < n; ++i) {
        if (e->steps_issued >= e->max_steps) break;
        // this step appends one token per live dialogue at position n_real + steps_issued: take its page if that
        // crosses a page boundary (the host's view of "live" lags the device by at most one poll: a dialogue that
        // has just finished may get one page it never writes, returned with the others)
        int len_bound = 1;
        for (int b = 0; b < e->B; ++b) {
            if (!e->slot_live[b]) continue;
            const int len = e->n_real[b] + e->steps_issued + 1;
            len_bound = std::max(len_bound, len);
            TRY(pool_grow(e, b, (len + MTTS_PAGE - 1) / MTTS_PAGE, st));
        }
        TRY(pool_flush(e, st));
        const int pages_bound = (len_bound + MTTS_PAGE - 1) / MTTS_PAGE;
        if (e->use_graphs && !e->prof) {
            // the attention grids are sized by the page bound: round it up to a whole pass-B chunk so that one
            // graph serves ATT_PB pages (512 steps); blocks past a row's last page exit at once
            hipGraphExec_t exec = nullptr;
            TRY(step_graph(e, std::min(round_up(pages_bound, ATT_PB), e->max_pages), &exec));
            HIPCHK(hipGraphLaunch(exec, st));
        } else {
            hipEvent_t ev = nullptr;
            prof_begin(e, PROF_STEP, st, &ev);
            // rough KV token count for the profile's byte figure: every row at its current length
            int64_t kvtok = 0;
            for (int b = 0; b < e->B; ++b) if (e->slot_live[b]) kvtok += e->n_real[b] + e->steps_issued + 1;
            TRY(step_body(e, pages_bound, st, kvtok));
            prof_end(e, st, ev);
        }
        e->steps_issued++;
    }
    return MTTS_OK;
}

int32_t mtts_step(MttsEngine* e, int32_t n_steps, void* stream) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    HIPCHK(hipSetDevice(e->device));
    return issue_steps(e, n_steps, S(stream));
}

int32_t mtts_sync_state(MttsEngine* e, int32_t* steps_done, int32_t* all_finished, void* stream) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->h_ls, e->d_ls, sizeof(LoopState), hipMemcpyDeviceToHost, S(stream)));
    // (both copies ride the caller's stream into pinned memory: no null-stream copy that would serialise with the codec leg)
    if (!e->continuous) HIPCHK(hipMemcpyAsync(e->h_seqs, e->d_seqs, e->B * sizeof(SeqState), hipMemcpyDeviceToHost, S(stream)));
    if (e->d_seal_cnt) HIPCHK(hipMemcpyAsync(e->h_seal_cnt, e->d_seal_cnt, (size_t)e->L * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    if (e->d_seal_cnt && e->kv_pack == 1) {
        bool changed = false;
        for (int n = 0; n < e->L; ++n) {
            const unsigned long long* c = e->h_seal_cnt + (size_t)n * 4;
            const bool k_on = c[0] + c[1] < 16 || c[1] * 8 <= c[0] + c[1], v_on = c[2] + c[3] < 16 || c[3] * 8 <= c[2] + c[3];
            changed |= (k_on != (bool)e->pack_k_on[n]) || (v_on != (bool)e->pack_v_on[n]);
            e->pack_k_on[n] = k_on; e->pack_v_on[n] = v_on;
        }
        if (changed) drop_graphs(e);                 // the captured steps hold the old choice of kernels
    }
    if (e->h_ls->error) return fail(MTTS_EINVAL, "device sampler error %d: more than 4096 candidate tokens (set top_k so that the k-th score's radix bin holds <= 4096 tokens)", e->h_ls->error);
    if (!e->continuous) {
        // static batch (mtts_generate semantics): a finished row only emits padding from here on and never touches
        // its KV pages again; everything issued so far has completed, so its pages go back to the pool
        const SeqState* ss = e->h_seqs;
        for (int b = 0; b < e->B; ++b)
            if (e->slot_live[b] && ss[b].step > 0 && !ss[b].unfinished && ss[b].active != 2) { e->slot_live[b] = 0; pool_release(e, b); }
    }
    if (steps_done) *steps_done = e->h_ls->step;
    if (all_finished) *all_finished = e->h_ls->done;
    return MTTS_OK;
}

static int read_rows(MttsEngine* e, const int32_t* d_src, int64_t* host, int capacity_steps, int* n_steps) {
    int steps = e->h_ls->step;
    if (steps > capacity_steps) return fail(MTTS_EINVAL, "output buffer holds %d steps, need %d", capacity_steps, steps);
    std::vector<int32_t> tmp((size_t)std::max(steps, 1) * 8);
    for (int b = 0; b < e->B; ++b) {
        if (steps) HIPCHK(hipMemcpy(tmp.data(), d_src + (size_t)b * e->gen_cap * 8, (size_t)steps * 8 * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < steps; ++s)
            for (int c = 0; c < 8; ++c) host[((size_t)s * e->B + b) * 8 + c] = tmp[(size_t)s * 8 + c];
    }
    if (n_steps) *n_steps = steps;
    return MTTS_OK;
}

int32_t mtts_read_generated(MttsEngine* e, int64_t* host_gen, int32_t capacity_steps, int32_t* n_steps) {
    if (!e || !e->began || !host_gen) return fail(MTTS_ESTATE, "nothing generated");
    HIPCHK(hipSetDevice(e->device));
    TRY(mtts_sync_state(e, nullptr, nullptr, nullptr));
    return read_rows(e, e->d_gen, host_gen, capacity_steps, n_steps);
}

int32_t mtts_read_logits_f32(MttsEngine* e, float* l0, float* l17, void* stream) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    if (!e->f32) return fail(MTTS_ESTATE, "bf16 engine: use mtts_read_logits");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(S(stream)));
    if (l0) HIPCHK(hipMemcpy2D(l0, (size_t)e->V0 * 4, e->logits0, (size_t)e->V0_pad * 4, (size_t)e->V0 * 4, e->B, hipMemcpyDeviceToHost));
    if (l17) {
        std::vector<float> tmp((size_t)MTTS_RCAP * 7 * e->Vs_pad);
        HIPCHK(hipMemcpy(tmp.data(), e->logits17, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < 7; ++c)
            for (int b = 0; b < e->B; ++b)
                memcpy(l17 + ((size_t)c * e->B + b) * e->Vs, tmp.data() + ((size_t)b * 7 + c) * e->Vs_pad, (size_t)e->Vs * 4);
    }
    return MTTS_OK;
}

int32_t mtts_read_logits(MttsEngine* e, uint16_t* l0, uint16_t* l17, void* stream) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    if (e->f32) return fail(MTTS_ESTATE, "fp32 engine: use mtts_read_logits_f32");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(S(stream)));
    if (l0) HIPCHK(hipMemcpy2D(l0, (size_t)e->V0 * 2, e->logits0, (size_t)e->V0_pad * 2, (size_t)e->V0 * 2, e->B, hipMemcpyDeviceToHost));
    if (l17) {
        std::vector<uint16_t> tmp((size_t)MTTS_RCAP * 7 * e->Vs_pad);
        HIPCHK(hipMemcpy(tmp.data(), e->logits17, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (int c = 0; c < 7; ++c)
            for (int b = 0; b < e->B; ++b)
                memcpy(l17 + ((size_t)c * e->B + b) * e->Vs, tmp.data() + ((size_t)b * 7 + c) * e->Vs_pad, (size_t)e->Vs * 2);
    }
    return MTTS_OK;
}

int32_t mtts_generate(MttsEngine* e, const int64_t* ids, const uint8_t* mask, int32_t B, int32_t T, int32_t max_length,
                      const MttsSamplerCfg* sampler, uint64_t seed, int64_t* out, int32_t out_capacity, int32_t* out_len,
                      const int64_t* forced, int32_t forced_len, int64_t* decisions, void* stream) {
    if (!out || !out_len) return fail(MTTS_EINVAL, "null output");
    TRY(mtts_begin(e, ids, mask, B, T, max_length, sampler, seed, stream));
    hipStream_t st = S(stream);
    const int base = e->base_length;
    if (forced) {
        if (!decisions) return fail(MTTS_EINVAL, "forced replay needs host_decisions");
        std::vector<int32_t> f((size_t)e->cfg.max_batch * e->gen_cap * 8, -1);
        for (int s = 0; s < e->max_steps && base + s < forced_len; ++s)
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < 8; ++c) {
                    int64_t tk = forced[((size_t)b * forced_len + base + s) * 8 + c];
                    if (tk < 0 || tk >= (c == 0 ? e->V0 : e->Vs)) return fail(MTTS_EINVAL, "forced token %lld out of range on channel %d", (long long)tk, c);
                    f[((size_t)b * e->gen_cap + s) * 8 + c] = (int32_t)tk;
                }
        HIPCHK(hipMemcpy(e->d_forced, f.data(), f.size() * 4, hipMemcpyHostToDevice));
        e->has_forced = true;
        e->max_steps = std::min(e->max_steps, forced_len - base);
    }
    // run ahead of the device in small batches; once every dialogue has finished the device marks all rows idle and
    // the steps still in flight do no attention and change no state
    int done = 0, steps = 0;
    while (!done && e->steps_issued < e->max_steps) {
        TRY(issue_steps(e, 8, st));
        TRY(mtts_sync_state(e, &steps, &done, stream));
    }
    TRY(mtts_sync_state(e, &steps, &done, stream));
    if (!forced && !done)
        return fail(MTTS_ESTATE, "the batch is still flushing after %d steps (%d past max_length): chained finished-row "
                    "resurrections (modeling_asteroid.py:140-141,168) outran the room max_seq_len %d leaves; raise it by %d",
                    steps, steps - (max_length - base), e->cfg.max_seq_len, linger_bound(B) - LINGER_STEPS);
    const int total = base + steps;
    if (total > out_capacity) return fail(MTTS_EINVAL, "out_capacity %d < %d", out_capacity, total);
    std::vector<int64_t> gen((size_t)std::max(steps, 1) * B * 8);
    int ns = 0;
    TRY(read_rows(e, e->d_gen, gen.data(), steps, &ns));
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < base; ++t)
            for (int c = 0; c < 8; ++c) out[((size_t)b * out_capacity + t) * 8 + c] = ids[((size_t)b * T + t) * 8 + c];
        for (int s = 0; s < steps; ++s)
            for (int c = 0; c < 8; ++c) out[((size_t)b * out_capacity + base + s) * 8 + c] = gen[((size_t)s * B + b) * 8 + c];
    }
    if (decisions) TRY(read_rows(e, e->d_declog, decisions, steps, &ns));
    *out_len = total;
    return MTTS_OK;
}

// Per-sequence loop state (needs_additional_steps, unfinished, tokens in the KV cache) after the steps issued so far.
int32_t mtts_read_seq_state(MttsEngine* e, int32_t* host_nas, int32_t* host_unfinished, int32_t* host_kv_len, void* stream) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    HIPCHK(hipSetDevice(e->device));
    std::vector<SeqState> ss(MTTS_RCAP);
    HIPCHK(hipMemcpyAsync(ss.data(), e->d_seqs, ss.size() * sizeof(SeqState), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    for (int b = 0; b < e->B; ++b) {
        if (host_nas) host_nas[b] = ss[b].nas;
        if (host_unfinished) host_unfinished[b] = ss[b].unfinished;
        if (host_kv_len) host_kv_len[b] = ss[b].kv_len;
    }
    return MTTS_OK;
}

// ---- continuous batching: per-slot dialogues ---------------------------------------------------------
// mtts_sched_open(e, B, sampler): B slots, all empty.  mtts_slot_submit(): prefill ONE dialogue into an empty slot
// while the others keep their state.  mtts_step() then advances every occupied slot by one frame; a dialogue that
// finishes leaves the batch at once (no finished-row padding) and its slot can be refilled.
int32_t mtts_sched_open(MttsEngine* e, int32_t B, int32_t gen_cap, const MttsSamplerCfg* sampler, void* stream) {
    if (!e || !sampler) return fail(MTTS_EINVAL, "null argument");
    TRY(mtts_weights_ready(e));
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = S(stream);
    if (B < 1 || B > e->cfg.max_batch) return fail(MTTS_EINVAL, "batch %d exceeds max_batch %d", B, e->cfg.max_batch);
    if (gen_cap < 8) return fail(MTTS_EINVAL, "gen_cap too small");
    TRY(ensure_gen_storage(e, gen_cap));
    e->B = B; e->steps_issued = 0; e->has_forced = false; e->continuous = true;
    e->max_steps = 1 << 30;
    e->n_real.assign(B, 0);
    e->join_step.assign(MTTS_RCAP, 0);
    e->max_real = 0;
    HIPCHK(hipStreamSynchronize(st));
    for (int b = 0; b < e->cfg.max_batch; ++b) { pool_release(e, b); e->slot_live[b] = 0; }
    e->pending_edits.n = 0;
    TRY(pack_policy_reset(e, st));
    std::vector<SeqState> ss(MTTS_RCAP, SeqState{-1, 0, 0, 0, 0, 0, 0, 0, 0});
    HIPCHK(hipMemcpyAsync(e->d_seqs, ss.data(), ss.size() * sizeof(SeqState), hipMemcpyHostToDevice, st));
    LoopState ls{0, 0, 1, B, 0, e->gen_cap, 0, e->f32 ? 1 : 0};
    e->forced_draw = 0;
    HIPCHK(hipMemcpyAsync(e->d_ls, &ls, sizeof(ls), hipMemcpyHostToDevice, st));
    *e->h_ls = ls;
    std::vector<RowMeta> dm(MTTS_RCAP, RowMeta{-1, 0, 0, 0});
    HIPCHK(hipMemcpyAsync(e->d_meta, dm.data(), dm.size() * sizeof(RowMeta), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->d_scfg, sampler, 8 * sizeof(MttsSamplerCfg), hipMemcpyHostToDevice, st));
    e->ch0_sampled = sampler[0].do_sample ? 1 : 0;
    HIPCHK(hipMemsetAsync(e->sscr.hist, 0, (size_t)e->cfg.max_batch * 2048 * 4, st));
    HIPCHK(hipMemsetAsync(e->sscr.cand_n, 0, (size_t)e->cfg.max_batch * 4, st));
    HIPCHK(hipMemsetAsync(e->sscr.overflow, 0, (size_t)e->cfg.max_batch * 4, st));
    HIPCHK(hipStreamSynchronize(st));
    e->began = true;
    return MTTS_OK;
}

// host_ids int64 [T][8] (one delay-shifted prompt, no padding), max_length in its own padded-slot units (T + max_new).
int32_t mtts_slot_submit(MttsEngine* e, int32_t slot, const int64_t* ids, int32_t T, int32_t max_length, uint64_t seed,
                         void* stream) {
    return mtts_slot_submit_row(e, slot, ids, T, max_length, seed, 0, stream);
}

// Philox row ids for the rows of the NEXT mtts_begin / mtts_generate (consumed by it): row b draws from counter
// (step, host_row_ids[b], channel, 0).  Default 0..B-1.  Lets one rank's share of a sharded batch, or one chunk of a
// large one, draw exactly what its rows would draw inside the whole batch.
int32_t mtts_set_row_ids(MttsEngine* e, const int32_t* host_row_ids, int32_t n) {
    if (!e || (n > 0 && !host_row_ids) || n < 0 || n > MTTS_RCAP) return fail(MTTS_EINVAL, "set_row_ids: bad argument");
    e->next_row_ids.assign(host_row_ids, host_row_ids + n);
    return MTTS_OK;
}

int32_t mtts_slot_submit_row(MttsEngine* e, int32_t slot, const int64_t* ids, int32_t T, int32_t max_length, uint64_t seed,
                             int32_t row_id, void* stream) {
    if (!e || !e->began || !e->continuous || !ids) return fail(MTTS_ESTATE, "mtts_sched_open has not run");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = S(stream);
    if (slot < 0 || slot >= e->B) return fail(MTTS_EINVAL, "slot %d out of range", slot);
    if (T < 8) return fail(MTTS_EINVAL, "T must be >= 8");
    const int base = T - 7, n = base;
    if (max_length <= base) return fail(MTTS_EINVAL, "max_length leaves no room to generate");
    const int max_new = max_length - base + FLUSH_STEPS;      // incl. a delay-pattern flush that starts at max_length
    if (max_new > e->gen_cap) return fail(MTTS_EINVAL, "dialogue may run %d steps, slot storage holds %d", max_new, e->gen_cap);
    if ((n + max_new + MTTS_PAGE - 1) / MTTS_PAGE > e->max_pages) return fail(MTTS_ENOMEM, "dialogue needs more KV pages than a sequence may hold");
    if (n + max_new > e->rope_rows) return fail(MTTS_EINVAL, "rope table too short");
    HIPCHK(hipStreamSynchronize(st));
    {   // the slot must be empty
        SeqState cur;
        HIPCHK(hipMemcpy(&cur, e->d_seqs + slot, sizeof(cur), hipMemcpyDeviceToHost));
        if (cur.active) return fail(MTTS_ESTATE, "slot %d is occupied", slot);
    }
    // admission: the prompt's pages plus the page its first step may open must be free now
    if (e->slot_live[slot]) { e->slot_live[slot] = 0; pool_release(e, slot); }      // finished, not collected yet
    if ((int)e->free_pages.size() < (n + 1 + MTTS_PAGE - 1) / MTTS_PAGE)
        return fail(MTTS_ENOMEM, "KV page pool: %d pages free, the prompt needs %d", (int)e->free_pages.size(), (n + 1 + MTTS_PAGE - 1) / MTTS_PAGE);
    TRY(pool_grow(e, slot, (n + MTTS_PAGE - 1) / MTTS_PAGE, st));
    TRY(pool_flush(e, st));
    const size_t Mpad = ((size_t)n + MTTS_RCAP - 1) / MTTS_RCAP * MTTS_RCAP;
    std::vector<int32_t> toks(Mpad * 8, 0);
    std::vector<RowMeta> metas(Mpad, RowMeta{-1, 0, 0, 0});
    std::vector<uint32_t> bm((size_t)8 * e->bm_words, 0u);
    std::vector<int32_t> tf(7 * 8, 0);
    for (int i = 0; i < T; ++i)
        for (int c = 0; c < 8; ++c) {
            int64_t t = ids[(size_t)i * 8 + c];
            if (t < 0 || t >= (c == 0 ? e->V0 : e->Vs)) return fail(MTTS_EINVAL, "token %lld out of range on channel %d", (long long)t, c);
            if (i < n) { toks[(size_t)i * 8 + c] = (int32_t)t; bm[(size_t)c * e->bm_words + (t >> 5)] |= 1u << (t & 31); }
            else tf[(i - n) * 8 + c] = (int32_t)t;
        }
    for (int i = 0; i < n; ++i) metas[i] = RowMeta{slot, i, i == n - 1 ? 1 : 0, 0};
    if (Mpad > e->pf_cap_rows) {
        if (e->d_pf_tokens) { hipFree(e->d_pf_tokens); hipFree(e->d_pf_meta); }
        TRY(dalloc(&e->d_pf_tokens, Mpad * 8, false));
        TRY(dalloc(&e->d_pf_meta, Mpad, false));
        e->pf_cap_rows = Mpad;
    }
    HIPCHK(hipMemcpy(e->d_pf_tokens, toks.data(), Mpad * 8 * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_pf_meta, metas.data(), Mpad * sizeof(RowMeta), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_bitmaps + (size_t)slot * 8 * e->bm_words, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_tf + (size_t)slot * 7 * 8, tf.data(), tf.size() * 4, hipMemcpyHostToDevice));
    const int pages_bound = (n + MTTS_PAGE - 1) / MTTS_PAGE;
    const size_t pfcap = e->f32 ? MTTS_PF32CAP : MTTS_PFCAP;
    for (size_t off = 0; off < Mpad; off += pfcap)
        TRY(forward_rows(e, e->d_pf_tokens + off * 8, e->d_pf_meta + off, (int)std::min<size_t>(pfcap, Mpad - off),
                         pages_bound, 0, st, 0));
    // logits of the dialogue's last prompt token only: heads on a one-row activation tile, copied into its slot
    // (the other slots' logits belong to dialogues that are mid-flight)
    if (e->f32) {            // fp32 engine: the GEMV writes the slot's logits rows directly
        const float* xin = e->hlast_f + (size_t)slot * e->H;
        launch_f32_linear(e->embf[0], xin, (float*)e->logits0 + (size_t)slot * e->V0_pad, 1, e->V0, e->H, e->V0_pad, e->h16, true, st);
        for (int c = 1; c < 8; ++c)
            launch_f32_linear(e->embf[c], xin, (float*)e->logits17 + (size_t)slot * 7 * e->Vs_pad + (size_t)(c - 1) * e->Vs_pad, 1, e->Vs,
                              e->H, 7 * e->Vs_pad, e->h16, true, st);
    } else {
    HIPCHK(hipMemsetAsync(e->xh, 0, (size_t)MTTS_MAXR * e->H * 2, st));
    launch_pack_rows((const uint16_t*)e->hlast + (size_t)slot * e->H, e->xh, 1, e->H, 1, st);
    launch_gemm(EPI_BF16, 1, e->p_h0, e->head0, e->xh, e->H, e->V0_pad, e->V0, nullptr, (uint16_t*)e->join_logits0, st);
    launch_gemm(EPI_BF16, 1, e->p_h17, e->heads17, e->xh, e->H, 7 * e->Vs_pad, 7 * e->Vs_pad, nullptr, (uint16_t*)e->join_logits17, st);
    HIPCHK(hipMemcpyAsync((uint16_t*)e->logits0 + (size_t)slot * e->V0_pad, e->join_logits0, (size_t)e->V0 * 2, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync((uint16_t*)e->logits17 + (size_t)slot * 7 * e->Vs_pad, e->join_logits17, (size_t)7 * e->Vs_pad * 2, hipMemcpyDeviceToDevice, st));
    }
    SeqState ns{-1, 1, n, 0, base, max_length, row_id, 1, seed};
    HIPCHK(hipMemcpyAsync(e->d_seqs + slot, &ns, sizeof(ns), hipMemcpyHostToDevice, st));
    int32_t zero = 0;
    HIPCHK(hipMemcpyAsync(&e->d_ls->done, &zero, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    e->n_real[slot] = n - e->steps_issued;          // so that n_real + steps_issued is this dialogue's current length
    e->join_step[slot] = e->steps_issued;
    e->slot_live[slot] = 1;
    return MTTS_OK;
}

// host_state int32 [B][4] = (active, unfinished, steps generated, tokens in cache)
int32_t mtts_slot_states(MttsEngine* e, int32_t* host_state, void* stream) {
    if (!e || !e->began || !host_state) return fail(MTTS_ESTATE, "engine not started");
    HIPCHK(hipSetDevice(e->device));
    std::vector<SeqState> ss(MTTS_RCAP);
    HIPCHK(hipMemcpyAsync(ss.data(), e->d_seqs, ss.size() * sizeof(SeqState), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    for (int b = 0; b < e->B; ++b) {
        host_state[b * 4 + 0] = ss[b].active; host_state[b * 4 + 1] = ss[b].unfinished;
        host_state[b * 4 + 2] = ss[b].step; host_state[b * 4 + 3] = ss[b].kv_len;
        // a dialogue that has left the batch (scheduler mode) gives its pages back: the stream is idle, nothing
        // in flight reads them
        if (e->continuous && e->slot_live[b] && !ss[b].active) { e->slot_live[b] = 0; pool_release(e, b); }
    }
    return MTTS_OK;
}

int32_t mtts_slot_evict(MttsEngine* e, int32_t slot, void* stream) {
    if (!e || !e->began || !e->continuous) return fail(MTTS_ESTATE, "mtts_sched_open has not run");
    if (slot < 0 || slot >= e->B) return fail(MTTS_EINVAL, "slot %d out of range", slot);
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(S(stream)));
    SeqState cur;
    HIPCHK(hipMemcpy(&cur, e->d_seqs + slot, sizeof(cur), hipMemcpyDeviceToHost));
    cur.active = 0; cur.unfinished = 0;
    HIPCHK(hipMemcpy(e->d_seqs + slot, &cur, sizeof(cur), hipMemcpyHostToDevice));
    RowMeta idle{-1, 0, 0, 0};
    HIPCHK(hipMemcpy(e->d_meta + slot, &idle, sizeof(idle), hipMemcpyHostToDevice));
    if (e->slot_live[slot]) { e->slot_live[slot] = 0; pool_release(e, slot); }
    return MTTS_OK;
}

int32_t mtts_kv_pool_state(MttsEngine* e, int32_t* total_pages, int32_t* free_pages, int32_t* max_pages_per_seq) {
    if (!e) return fail(MTTS_EINVAL, "null engine");
    if (total_pages) *total_pages = e->total_pages;
    if (free_pages) *free_pages = (int32_t)e->free_pages.size();
    if (max_pages_per_seq) *max_pages_per_seq = e->max_pages;
    return MTTS_OK;
}

int32_t mtts_read_page_table(MttsEngine* e, int32_t* host_table, int32_t* host_n_pages) {
    if (!e || !host_table || !host_n_pages) return fail(MTTS_EINVAL, "null argument");
    memcpy(host_table, e->h_page_table.data(), e->h_page_table.size() * 4);
    memcpy(host_n_pages, e->n_pages.data(), e->n_pages.size() * 4);
    return MTTS_OK;
}

// verification hook: the page table as the DEVICE holds it (after everything queued on `stream`)
int32_t mtts_debug_read_device_page_table(MttsEngine* e, int32_t* host_table, void* stream) {
    if (!e || !host_table) return fail(MTTS_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->device));
    TRY(pool_flush(e, S(stream)));
    HIPCHK(hipMemcpyAsync(host_table, e->d_page_table, e->h_page_table.size() * 4, hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return MTTS_OK;
}

int32_t mtts_set_forced_mode(MttsEngine* e, int32_t as_draw) {
    if (!e) return fail(MTTS_EINVAL, "null engine");
    if (as_draw < 0 || as_draw > 2) return fail(MTTS_EINVAL, "forced mode must be 0, 1 or 2");
    e->forced_draw = as_draw;
    return MTTS_OK;
}

// generated rows of one slot: host_rows int64 [steps][8]
int32_t mtts_slot_read(MttsEngine* e, int32_t slot, int64_t* host_rows, int32_t capacity_steps, int32_t* n_steps) {
    if (!e || !e->began || !host_rows || slot < 0 || slot >= e->B) return fail(MTTS_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    SeqState cur;
    HIPCHK(hipMemcpy(&cur, e->d_seqs + slot, sizeof(cur), hipMemcpyDeviceToHost));
    if (cur.step > capacity_steps) return fail(MTTS_EINVAL, "buffer holds %d steps, need %d", capacity_steps, cur.step);
    std::vector<int32_t> tmp((size_t)std::max(cur.step, 1) * 8);
    if (cur.step) HIPCHK(hipMemcpy(tmp.data(), e->d_gen + (size_t)slot * e->gen_cap * 8, (size_t)cur.step * 8 * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < cur.step * 8; ++i) host_rows[i] = tmp[i];
    if (n_steps) *n_steps = cur.step;
    return MTTS_OK;
}

// Frames first..first+n-1 of every sequence as codec codes int64 [8][B][n] on the device (delay pattern undone,
// channel-0 offset removed): lets the codec decode windows while the decode loop is still running.  The caller
// orders `stream` after the steps that produced frame first+n+6 (event / same stream).
int32_t mtts_export_codes(MttsEngine* e, int32_t first, int32_t n, int64_t* dev_codes, void* stream) {
    if (!e || !e->began || !dev_codes) return fail(MTTS_ESTATE, "nothing generated");
    if (first < 0 || n < 1 || first + n + 7 > e->steps_issued) return fail(MTTS_EINVAL, "frames %d..%d need %d issued steps, have %d", first, first + n - 1, first + n + 7, e->steps_issued);
    HIPCHK(hipSetDevice(e->device));
    launch_export_codes(e->d_gen, dev_codes, e->B, first, n, e->cfg.speech_range_lo, e->cfg.speech_vocab_size - 2, e->gen_cap, S(stream));
    HIPCHK(hipGetLastError());
    return MTTS_OK;
}

int32_t mtts_profile_enable(MttsEngine* e, int32_t on) {
    if (!e) return fail(MTTS_EINVAL, "null engine");
    e->prof = on != 0;
    for (int w = 0; w < PROF_N; ++w) {
        for (auto& pr : e->ev[w]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
        e->ev[w].clear();
        e->prof_bytes[w] = 0;
    }
    return MTTS_OK;
}

int32_t mtts_profile_read(MttsEngine* e, int32_t which, double* total_ms, int64_t* launches, int64_t* bytes) {
    if (!e || which < 0 || which >= PROF_N) return fail(MTTS_EINVAL, "bad profile slot");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    double tot = 0;
    for (auto& pr : e->ev[which]) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)e->ev[which].size();
    if (bytes) *bytes = e->prof_bytes[which];
    return MTTS_OK;
}

// ---- per-kernel entry points --------------------------------------------------------
// device buffers of a test hook: freed on every return path
struct HookBufs {
    std::vector<void*> p;
    ~HookBufs() { for (void* q : p) if (q) hipFree(q); }
    template <typename T>
    int get(T** out, size_t n, bool zero = true) {
        int rc = dalloc(out, n, zero);
        if (!rc) p.push_back((void*)*out);
        return rc;
    }
};
int32_t mtts_k_gemm_bf16(const void* w, const void* x, void* y, int32_t M, int32_t N, int32_t K, int32_t ksplit, void* stream) {
    if (!w || !x || !y || M < 1 || M > MTTS_PFCAP || K % 16 || N < 1) return fail(MTTS_EINVAL, "gemm: need 1<=M<=MTTS_PFCAP, K%%16==0");
    hipStream_t st = S(stream);
    int Npad = round_up(N, 32);
    void *wp = nullptr, *xp = nullptr;
    float* part = nullptr;
    GemmPlan p = mtts_plan_gemm(Npad, K, ksplit);
    HookBufs hb;
    TRY(hb.get((uint16_t**)&wp, (size_t)Npad * K));
    // M <= 128: skinny kernel (decode); above: tiled kernel (prefill), ksplit as given or its own choice
    const bool tiled = M > MTTS_RCAP;
    const int ks = tiled ? (ksplit > 0 ? ksplit : mtts_tile_ksplit(Npad, K, M)) : p.ksplit;
    TRY(hb.get((uint16_t**)&xp, (size_t)MTTS_PFCAP * K));
    TRY(hb.get(&part, (size_t)ks * MTTS_PFCAP * Npad));
    launch_pack_weight(w, wp, N, K, Npad, 1, 0, st);
    const int tiles = (M + 31) / 32;
    launch_pack_rows(x, xp, M, K, tiles == 3 ? 4 : tiles, st);
    if (tiled) launch_gemm_tile(EPI_PARTIAL, M, ks, wp, xp, K, Npad, Npad, part, nullptr, st);
    else launch_gemm(EPI_PARTIAL, tiles, p, wp, xp, K, Npad, Npad, part, nullptr, st);
    launch_reduce_partial_bf16(part, y, ks, Npad, N, M, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return MTTS_OK;
}

int32_t mtts_k_rmsnorm(const void* x, const void* w, void* y, int32_t rows, int32_t n, float eps, void* stream) {
    if (!x || !w || !y || rows < 1 || n < 1) return fail(MTTS_EINVAL, "rmsnorm: bad argument");
    launch_rmsnorm_rows(x, w, y, rows, n, eps, S(stream));
    HIPCHK(hipGetLastError());
    return MTTS_OK;
}

int32_t mtts_k_sample(const void* logits, int32_t rows, int32_t vocab, const void* bitmap, const MttsSamplerCfg* cfg,
                      int32_t mask_id, uint64_t seed, int32_t step, int32_t channel, int32_t* dev_tokens, void* stream) {
    if (!logits || !cfg || !dev_tokens || rows < 1 || vocab < 1 || channel < 0 || channel > 7) return fail(MTTS_EINVAL, "sample: bad argument");
    hipStream_t st = S(stream);
    MttsSamplerCfg h[8];
    for (int i = 0; i < 8; ++i) h[i] = *cfg;
    MttsSamplerCfg* d = nullptr;
    int32_t *err = nullptr, *dec = nullptr;
    if (rows > MTTS_RCAP) return fail(MTTS_EINVAL, "sample: at most 128 rows");
    HookBufs hb;
    TRY(hb.get(&d, 8));
    TRY(hb.get(&err, 1));
    TRY(hb.get(&dec, (size_t)rows * 8));
    HIPCHK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    SampleScratch sc;
    struct ScratchGuard { SampleScratch* s; ~ScratchGuard() { free_scratch(*s); } } sg{&sc};
    sc = SampleScratch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    TRY(alloc_scratch(sc, rows, vocab));
    launch_sample_single(logits, rows, vocab, (const uint32_t*)bitmap, (vocab + 31) / 32, d, mask_id, seed, step, channel, dec, err, sc, full_cap_for(vocab), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int32_t> hd((size_t)rows * 8);
    int32_t herr = 0;
    HIPCHK(hipMemcpy(hd.data(), dec, hd.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> outv(rows);
    for (int r = 0; r < rows; ++r) outv[r] = hd[(size_t)r * 8 + channel];
    HIPCHK(hipMemcpy(dev_tokens, outv.data(), rows * 4, hipMemcpyHostToDevice));
    if (herr) return fail(MTTS_EINVAL, "sample: more than 4096 candidate tokens");
    return MTTS_OK;
}

// ---- per-kernel entry points for attention and RoPE / cache write (unit tests) ----------------------------------------
void launch_pack_kv_pages(const void* K, const void* V, void* kcache, void* vcache, const int32_t* page_table, const int32_t* lens,
                          int S, int Lmax, int nkv, int max_pages, int total_pages, hipStream_t st);
void launch_bf16_to_f32(const void* a, float* b, size_t n, hipStream_t st);
void launch_unpack_rows(const void* packed, void* out, int R, int K, hipStream_t st);

// q/k/v epilogue of one token per row (qkv_post_kernel): dev_qkv bf16 [R][(nq+2*nkv)*128] = the three Linears' outputs,
// host_pos int32 [R] positions, dev_qnorm / dev_knorm bf16 [128], dev_cos / dev_sin bf16 [rope_rows][64].
// Outputs bf16: dev_q [R][nq][128] (normed + rotated), dev_k [R][nkv][128] (normed + rotated, read back from the K page it was
// written to), dev_v [R][nkv][128] (read back from the V page).  Every row is its own sequence (page table = one page each).
int32_t mtts_k_rope_kvwrite(const void* dev_qkv, const int32_t* host_pos, const void* dev_qnorm, const void* dev_knorm,
                            const void* dev_cos, const void* dev_sin, int32_t R, int32_t nq, int32_t nkv, float eps,
                            void* dev_q, void* dev_k, void* dev_v, void* stream) {
    if (!dev_qkv || !host_pos || !dev_q || !dev_k || !dev_v || R < 1 || R > MTTS_RCAP || nq < 1 || nkv < 1) return fail(MTTS_EINVAL, "rope_kvwrite: bad argument");
    hipStream_t st = S(stream);
    const int N = (nq + 2 * nkv) * MTTS_HD;
    int maxpos = 0;
    for (int r = 0; r < R; ++r) { if (host_pos[r] < 0) return fail(MTTS_EINVAL, "negative position"); maxpos = std::max(maxpos, host_pos[r]); }
    const int max_pages = maxpos / MTTS_PAGE + 1, total_pages = R * max_pages;
    float* slab = nullptr; RowMeta* meta = nullptr; int32_t* pt = nullptr; uint16_t *kc = nullptr, *vc = nullptr;
    HookBufs hb;
    TRY(hb.get(&slab, (size_t)MTTS_PFCAP * N));
    TRY(hb.get(&meta, R)); TRY(hb.get(&pt, (size_t)R * max_pages));
    TRY(hb.get(&kc, (size_t)total_pages * nkv * MTTS_PAGE * MTTS_HD)); TRY(hb.get(&vc, (size_t)total_pages * nkv * MTTS_PAGE * MTTS_HD));
    std::vector<RowMeta> hm(R);
    std::vector<int32_t> hpt((size_t)R * max_pages);
    for (int r = 0; r < R; ++r) { hm[r] = RowMeta{r, host_pos[r], 1, 0}; for (int p = 0; p < max_pages; ++p) hpt[(size_t)r * max_pages + p] = r * max_pages + p; }
    HIPCHK(hipMemcpy(meta, hm.data(), R * sizeof(RowMeta), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pt, hpt.data(), hpt.size() * 4, hipMemcpyHostToDevice));
    launch_bf16_to_f32(dev_qkv, slab, (size_t)R * N, st);
    launch_qkv_post(slab, 1, N, meta, dev_qnorm, dev_knorm, dev_cos, dev_sin, dev_q, kc, vc, pt, max_pages, total_pages, R, nq, nkv, eps, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    // read the written K / V rows back out of their pages
    std::vector<uint16_t> hk((size_t)total_pages * nkv * MTTS_PAGE * MTTS_HD), hv(hk.size()), ok((size_t)R * nkv * MTTS_HD), ov(ok.size());
    HIPCHK(hipMemcpy(hk.data(), kc, hk.size() * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hv.data(), vc, hv.size() * 2, hipMemcpyDeviceToHost));
    for (int r = 0; r < R; ++r)
        for (int h = 0; h < nkv; ++h)
            for (int d = 0; d < MTTS_HD; ++d) {
                const int page = r * max_pages + host_pos[r] / MTTS_PAGE, t = host_pos[r] % MTTS_PAGE;
                const size_t base = ((size_t)h * total_pages + page) * (MTTS_PAGE * MTTS_HD);
                ok[((size_t)r * nkv + h) * MTTS_HD + d] = hk[base + (((d >> 3) * 64) + t) * 8 + (d & 7)];
                ov[((size_t)r * nkv + h) * MTTS_HD + d] = hv[base + ((size_t)(t >> 1) * MTTS_HD + d) * 2 + (t & 1)];
            }
    HIPCHK(hipMemcpy(dev_k, ok.data(), ok.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dev_v, ov.data(), ov.size() * 2, hipMemcpyHostToDevice));
    return MTTS_OK;
}

// Decode attention of one query token per row over a paged cache (attn_scores / attn_pv / attn_combine, the launches of
// a decode step): dev_q bf16 [R][nq][128]; dev_k / dev_v bf16 [R][Lmax][nkv][128] row-major (row r uses its first
// host_lens[r] tokens; its query sits at position host_lens[r]-1); host_page_table int32 [R][pages] with
// pages = ceil(Lmax/64): any permutation of 0..R*pages-1 (NULL = consecutive).  dev_out bf16 [R][nq*128].
int32_t mtts_k_paged_attn_decode(const void* dev_q, const void* dev_k, const void* dev_v, const int32_t* host_lens,
                                 const int32_t* host_page_table, int32_t R, int32_t Lmax, int32_t nq, int32_t nkv,
                                 void* dev_out, void* stream) {
    if (!dev_q || !dev_k || !dev_v || !host_lens || !dev_out || R < 1 || R > MTTS_MAXR || Lmax < 1 || nq < 1 || nkv < 1 || nq % nkv)
        return fail(MTTS_EINVAL, "paged_attn_decode: bad argument (1..32 rows)");
    hipStream_t st = S(stream);
    const int max_pages = (Lmax + MTTS_PAGE - 1) / MTTS_PAGE, total_pages = R * max_pages, nch = (max_pages + ATT_PB - 1) / ATT_PB;
    std::vector<int32_t> hpt((size_t)R * max_pages);
    std::vector<char> seen(total_pages, 0);
    for (size_t i = 0; i < hpt.size(); ++i) {
        hpt[i] = host_page_table ? host_page_table[i] : (int32_t)i;
        if (hpt[i] < 0 || hpt[i] >= total_pages || seen[hpt[i]]) return fail(MTTS_EINVAL, "page table must be a permutation of 0..%d", total_pages - 1);
        seen[hpt[i]] = 1;
    }
    std::vector<RowMeta> hm(MTTS_MAXR, RowMeta{-1, 0, 0, 0});
    int pages_bound = 1;
    for (int r = 0; r < R; ++r) {
        if (host_lens[r] < 1 || host_lens[r] > Lmax) return fail(MTTS_EINVAL, "row %d: length %d outside 1..%d", r, host_lens[r], Lmax);
        hm[r] = RowMeta{r, host_lens[r] - 1, 1, 0};
        pages_bound = std::max(pages_bound, (host_lens[r] + MTTS_PAGE - 1) / MTTS_PAGE);
    }
    RowMeta* meta = nullptr; int32_t *pt = nullptr, *lens = nullptr; uint16_t *kc = nullptr, *vc = nullptr, *scores = nullptr, *outp = nullptr;
    float *stats = nullptr, *opart = nullptr;
    const size_t cache_n = (size_t)total_pages * nkv * MTTS_PAGE * MTTS_HD;
    HookBufs hb;
    TRY(hb.get(&meta, MTTS_MAXR)); TRY(hb.get(&pt, hpt.size())); TRY(hb.get(&lens, R));
    TRY(hb.get(&kc, cache_n)); TRY(hb.get(&vc, cache_n));
    TRY(hb.get(&scores, (size_t)MTTS_MAXR * nq * max_pages * MTTS_PAGE));
    TRY(hb.get(&stats, (size_t)MTTS_MAXR * nq * max_pages * 2));
    TRY(hb.get(&opart, (size_t)MTTS_MAXR * nq * nch * MTTS_HD));
    TRY(hb.get(&outp, (size_t)MTTS_MAXR * nq * MTTS_HD));
    HIPCHK(hipMemcpy(meta, hm.data(), hm.size() * sizeof(RowMeta), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pt, hpt.data(), hpt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(lens, host_lens, R * 4, hipMemcpyHostToDevice));
    launch_pack_kv_pages(dev_k, dev_v, kc, vc, pt, lens, R, Lmax, nkv, max_pages, total_pages, st);
    // like the engine: complete pages are read in their sealed form unless MTTS_KV_PACK=0
    KvPack pk{nullptr, nullptr};
    const char* g = getenv("MTTS_KV_PACK");
    if (!g || atoi(g) != 0) {
        uint8_t *kp = nullptr, *vp = nullptr;
        const size_t pk_n = (size_t)total_pages * nkv * MTTS_PKU * 64 * 16;
        TRY(hb.get(&kp, pk_n)); TRY(hb.get(&vp, pk_n));
        launch_kv_seal_all(kc, vc, kp, vp, total_pages, nkv, 1, nullptr, st);
        pk = KvPack{kp, vp};
    }
    const float scale = 1.0f / sqrtf((float)MTTS_HD);
    if (launch_attn(dev_q, kc, vc, pt, meta, scores, stats, opart, outp, MTTS_MAXR, pages_bound, max_pages, total_pages, nch, nq, nkv,
                    scale, nullptr, 0, st, pk.k ? &pk : nullptr))
        return fail(MTTS_EINVAL, "attention group size not built (1, 2, 4)");
    launch_unpack_rows(outp, dev_out, R, nq * MTTS_HD, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return MTTS_OK;
}

// Test hook for the sealed page format (attn.hip: seal_lane): `npages` bf16 pages of 16 KiB -> sealed pages of 13 KiB.
extern "C" int32_t mtts_k_kv_seal(const void* dev_pages, int32_t npages, void* dev_sealed, int32_t as_k, void* stream) {
    if (!dev_pages || !dev_sealed || npages < 1) return fail(MTTS_EINVAL, "kv_seal: bad argument");
    launch_kv_seal_pages(dev_pages, dev_sealed, npages, as_k, S(stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(S(stream)));
    return MTTS_OK;
}

// Debug hook: out6 = {complete K pages of the live sequences (x kv heads x layers), of which not sealed (a lane did not
// fit: read as bf16), the same two numbers for V, layers whose K / V reads currently use the sealed pages}.
// MTTS_ESTATE when the engine runs without sealed pages.
extern "C" int32_t mtts_debug_kv_pack_stats(MttsEngine* e, int64_t* out6) {
    if (!e || !out6) return fail(MTTS_EINVAL, "null argument");
    if (!e->kpack) return fail(MTTS_ESTATE, "the engine keeps no sealed pages (fp32 / fp16 engine, or MTTS_KV_PACK=0)");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    std::vector<SeqState> ss(MTTS_RCAP);
    HIPCHK(hipMemcpy(ss.data(), e->d_seqs, ss.size() * sizeof(SeqState), hipMemcpyDeviceToHost));
    std::vector<int32_t> complete(e->B, 0);
    for (int b = 0; b < e->B; ++b)
        if (e->slot_live[b]) complete[b] = std::min(ss[b].kv_len >> 6, e->n_pages[b]);
    HookBufs hb;
    int32_t* dc = nullptr; unsigned long long* dout = nullptr;
    TRY(hb.get(&dc, e->B)); TRY(hb.get(&dout, 4));
    HIPCHK(hipMemcpy(dc, complete.data(), e->B * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout, 0, 32));
    launch_kv_pack_count(e->kpack, e->vpack, e->d_page_table, dc, e->B, e->max_pages, e->total_pages, e->nkv, e->L, dout, nullptr);
    HIPCHK(hipGetLastError());
    unsigned long long h[4];
    HIPCHK(hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out6[i] = (int64_t)h[i];
    out6[4] = out6[5] = 0;
    for (int n = 0; n < e->L; ++n) { out6[4] += e->pack_k_on[n]; out6[5] += e->pack_v_on[n]; }
    return MTTS_OK;
}

// Measurement hook (bench/profiling only): pretend every live sequence already holds `kv_len` tokens.
// The cache content is whatever the pages hold; used to reach a long context without replaying it
// when collecting PMC counters.
extern "C" int32_t mtts_debug_set_kv_len(MttsEngine* e, int32_t kv_len) {
    if (!e || !e->began) return fail(MTTS_ESTATE, "mtts_begin has not run");
    if (e->f32) return fail(MTTS_EINVAL, "measurement hook of the bf16 engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const int cap = std::min(e->max_pages * MTTS_PAGE, e->rope_rows);          // positions the pages and the RoPE table hold
    const int limit = cap - MTTS_PAGE;
    if (kv_len < 1 || kv_len > limit) return fail(MTTS_EINVAL, "kv_len %d outside 1..%d", kv_len, limit);
    e->max_steps = std::min(e->max_steps, e->steps_issued + cap - kv_len);   // stay inside the pages and the RoPE table
    std::vector<SeqState> ss(MTTS_RCAP);
    HIPCHK(hipMemcpy(ss.data(), e->d_seqs, ss.size() * sizeof(SeqState), hipMemcpyDeviceToHost));
    for (int b = 0; b < e->B; ++b) { ss[b].kv_len = kv_len; e->n_real[b] = kv_len - e->steps_issued; }
    e->max_real = kv_len - e->steps_issued;
    HIPCHK(hipMemcpy(e->d_seqs, ss.data(), ss.size() * sizeof(SeqState), hipMemcpyHostToDevice));
    launch_fill_random_bf16(e->kcache, e->layer_stride * e->L, 0x1234u, nullptr);
    launch_fill_random_bf16(e->vcache, e->layer_stride * e->L, 0x9876u, nullptr);
    if (e->kpack) launch_kv_seal_all(e->kcache, e->vcache, e->kpack, e->vpack, e->total_pages, e->nkv, e->L, nullptr, nullptr);
    HIPCHK(hipDeviceSynchronize());
    return MTTS_OK;
}

// Measurement hook: `iters` back-to-back launches of one attention pass (phase 1 = scores, 2 = PV) at the
// engine's CURRENT decode state, cycling over the layers' caches; average duration from two HIP events on the
// launch stream.  (Events around a single launch also time the launch gap, which rocprof's kernel duration
// does not; a train of launches does not have that bias.)
extern "C" int32_t mtts_k_attn_bench(MttsEngine* e, int32_t phase, int32_t iters, float* avg_ms, int64_t* bytes_per_launch) {
    if (!e || !e->began || (phase != 1 && phase != 2) || iters < 1 || !avg_ms) return fail(MTTS_EINVAL, "attn_bench: bad argument");
    if (e->f32) return fail(MTTS_EINVAL, "measurement hook of the bf16 engine");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const float scale = 1.0f / sqrtf((float)MTTS_HD);
    const int R = round_up(e->B, 32);
    const int len_bound = e->max_real + e->steps_issued + 1;
    const int pages_bound = (len_bound + MTTS_PAGE - 1) / MTTS_PAGE;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int n) {
        for (int i = 0; i < n; ++i) {
            const int layer = i % e->L;
            uint16_t* kc = (uint16_t*)e->kcache + e->layer_stride * layer;
            uint16_t* vc = (uint16_t*)e->vcache + e->layer_stride * layer;
            // the product's decode launch: q/k/v epilogue fused (the slabs are whatever the last step left there)
            const QkvFuse fz{e->partial, e->p_qkv.ksplit, e->qkv_rows, (const uint16_t*)e->layers[layer].qn,
                             (const uint16_t*)e->layers[layer].kn, (const uint16_t*)e->rope_cos, (const uint16_t*)e->rope_sin,
                             e->cfg.rms_norm_eps};
            const KvPack pk = layer_pack(e, layer, pages_bound);
            launch_attn(e->qbuf, kc, vc, e->d_page_table, e->d_meta, e->scores, e->stats, e->opart, e->attn_p, R, pages_bound,
                        e->max_pages, e->total_pages, e->nchunks_max, e->nq, e->nkv, scale, e->B * pages_bound <= e->fuse_qkv_max ? &fz : nullptr, phase, nullptr,
                        (pk.k || pk.v) ? &pk : nullptr);
        }
    };
    run(e->L);                               // warm-up
    hipEventRecord(e0, nullptr);
    run(iters);
    hipEventRecord(e1, nullptr);
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_ms = ms / iters;
    if (bytes_per_launch) {
        std::vector<RowMeta> m(MTTS_RCAP);
        HIPCHK(hipMemcpy(m.data(), e->d_meta, m.size() * sizeof(RowMeta), hipMemcpyDeviceToHost));
        int64_t tok = 0;
        for (int b = 0; b < R; ++b) if (m[b].seq >= 0) tok += m[b].pos + 1;
        *bytes_per_launch = tok * e->nkv * MTTS_HD * 2;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return MTTS_OK;
}

// Tuning hook (not part of the product path): average time of one skinny-GEMM launch over `copies`
// distinct weight buffers (so that no launch finds its weights in L2 / Infinity Cache).
extern "C" int32_t mtts_k_gemm_bench(int32_t N, int32_t K, int32_t epi, int32_t ksplit, int32_t waves, int32_t copies,
                                     int32_t iters, float* avg_us) {
    if (N % 32 || K % 16 || copies < 1 || iters < 1 || !avg_us) return fail(MTTS_EINVAL, "gemm_bench: bad argument");
    // waves < 0: the tiled prefill kernel on -waves rows (<= MTTS_PFCAP), split-K as given
    const int tile_rows = waves < 0 ? -waves : 0;
    if (tile_rows > MTTS_PFCAP) return fail(MTTS_EINVAL, "gemm_bench: at most MTTS_PFCAP rows");
    GemmPlan p = (ksplit > 0 && waves > 0) ? mtts_plan_gemm_forced(N, K, ksplit, waves) : mtts_plan_gemm(N, K, tile_rows ? std::max(ksplit, 1) : ksplit);
    if (tile_rows) p.ksplit = std::max(ksplit, 1);
    std::vector<uint16_t*> w(copies, nullptr);
    for (auto& q : w) { TRY(dalloc(&q, (size_t)N * K, false)); HIPCHK(hipMemset(q, 0x3c, (size_t)N * K * 2)); }
    uint16_t *x = nullptr, *out = nullptr;
    float* part = nullptr;
    TRY(dalloc(&x, (size_t)MTTS_PFCAP * K, false));
    HIPCHK(hipMemset(x, 0x3c, (size_t)MTTS_PFCAP * K * 2));
    TRY(dalloc(&out, (size_t)MTTS_PFCAP * N));
    TRY(dalloc(&part, (size_t)p.ksplit * MTTS_PFCAP * N));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&](int i) {
        if (tile_rows) launch_gemm_tile(epi, tile_rows, p.ksplit, w[i % copies], x, K, N, N, part, out, nullptr);
        else launch_gemm(epi, 1, p, w[i % copies], x, K, N, N, part, out, nullptr);
    };
    for (int i = 0; i < copies; ++i) go(i);
    HIPCHK(hipDeviceSynchronize());
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) go(i);
    hipEventRecord(e1, nullptr);
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *avg_us = ms * 1000.f / iters;
    for (auto q : w) hipFree(q);
    hipFree(x); hipFree(out); hipFree(part);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return MTTS_OK;
}
