"""CPU oracle for the AsteroidTTS autoregressive hot path (numpy).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`moss-ttsd_amd/`) may
import this module; it is used by `tests/`, by `__graft_entry__.smoke()` and by
the `cpu_baseline` leg of `bench.py`, and only as the checker.

It restates, with explicit rounding points, what the reference computes when
`AsteroidTTSInstruct` runs on CPU with eager attention:

  * 8-channel embedding sum ............ reference modeling_asteroid.py:235-250
  * Qwen3 decoder stack ................ transformers (pinned 4.53.2 by the
    reference's requirements.txt:3; third-party, not under /root/reference)
    models/qwen3/modeling_qwen3.py: RMSNorm :59-64, MLP :81-83, RoPE :126-170,
    eager attention :185-208, attention block :241-280, layer :299-324
  * 8 tied LM heads .................... reference modeling_asteroid.py:412
  * the `_sample` decode loop .......... reference modeling_asteroid.py:83-169
  * HF logits processors (third-party, transformers generation/logits_process.py:
    RepetitionPenalty, Temperature, TopK, TopP) instantiated at
    modeling_asteroid.py:97-106

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4).  The oracle is pinned by fixtures generated here by importing
the reference itself (tests/golden/make_golden.py -> tests/golden/ar_*.npz):
per-step logits, greedy token matrices and processor outputs.

Rounding model ("bf16" mode): every nn.Linear output, every elementwise bf16
op and the attention probabilities are rounded to bf16 exactly where torch's
CPU bf16 kernels round; accumulations are fp32.  In "fp32" mode no rounding
happens.  Summation ORDER inside a dot product is not part of the contract
(torch's own CPU and GPU back-ends differ there).

Sampling: torch.multinomial's stream cannot be reproduced, so sampled channels
use the engine's documented stream instead: Philox4x32-10 keyed by the seed,
counter (step, row, channel, 0), 24-bit uniform u, inverse CDF over the kept
tokens from the highest score down (ties: higher token id first).  The kept set and its probabilities follow
the HF processors exactly.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def round_bf16(x):
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(F32)
    return np.where(np.isfinite(x), out, x)


def _ident(x):
    return np.asarray(x, dtype=F32)


def round_fp16(x):
    """fp32 -> nearest-even fp16 (overflow -> inf, as torch's cast), returned as fp32."""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


# --------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011), the engine's sampling stream.
# --------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key):
    """counter: 4 python ints (uint32), key: 2 ints -> 4 uint32 outputs."""
    c = [int(v) & 0xFFFFFFFF for v in counter]
    k = [int(v) & 0xFFFFFFFF for v in key]
    for _ in range(10):
        p0 = _PH_M0 * c[0]
        p1 = _PH_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + _PH_W0) & 0xFFFFFFFF, (k[1] + _PH_W1) & 0xFFFFFFFF]
    return c


def philox_uniform(seed, step, row, channel):
    x = philox4x32((step, row, channel, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[0]
    return F32((x >> 8) * (1.0 / 16777216.0))


# --------------------------------------------------------------------------
# HF logits processors (semantics of transformers generation/logits_process.py)
# --------------------------------------------------------------------------
def proc_repetition_penalty(history, scores, penalty):
    """history int64 [B,n]; scores fp32 [B,V] (copy returned)."""
    out = scores.copy()
    for b in range(scores.shape[0]):
        idx = history[b]
        s = scores[b, idx]
        out[b, idx] = np.where(s < 0, s * F32(penalty), s / F32(penalty)).astype(F32)
    return out


def proc_temperature(scores, t):
    return (scores / F32(t)).astype(F32)


def proc_top_k(scores, k):
    k = min(int(k), scores.shape[-1])
    kth = np.sort(scores, axis=-1)[:, -k][:, None]
    out = scores.copy()
    out[scores < kth] = -np.inf
    return out


def softmax_f32(x):
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp((x - m).astype(F32)).astype(F32)
    return (e / np.sum(e, axis=-1, keepdims=True, dtype=F32)).astype(F32)


def proc_top_p(scores, p, min_keep=1):
    order = np.argsort(scores, axis=-1, kind="stable")            # ascending, stable
    srt = np.take_along_axis(scores, order, axis=-1)
    cum = np.cumsum(softmax_f32(srt), axis=-1, dtype=F32)
    remove = cum <= F32(1.0 - p)
    remove[:, -min_keep:] = False
    out = scores.copy()
    rm = np.zeros_like(remove)
    np.put_along_axis(rm, order, remove, axis=-1)
    out[rm] = -np.inf
    return out


def apply_processors(history, logits, layer_cfg):
    """Order fixed by reference modeling_asteroid.py:99-106."""
    s = logits.astype(F32)
    if layer_cfg.get("repetition_penalty") is not None:
        s = proc_repetition_penalty(history, s, layer_cfg["repetition_penalty"])
    if layer_cfg.get("temperature") is not None:
        s = proc_temperature(s, layer_cfg["temperature"])
    if layer_cfg.get("top_k") is not None:
        s = proc_top_k(s, layer_cfg["top_k"])
    if layer_cfg.get("top_p") is not None:
        s = proc_top_p(s, layer_cfg["top_p"])
    return s


def sample_from_scores(scores, seed, step, channel, row0=0):
    """Engine-defined draw: inverse CDF over kept tokens from the highest score down
    (ties: higher token id first)."""
    B = scores.shape[0]
    out = np.zeros(B, dtype=np.int64)
    for b in range(B):
        s = scores[b]
        kept = np.nonzero(s > -np.inf)[0]
        kept = kept[np.lexsort((kept, s[kept]))[::-1]]     # (score asc, id asc) reversed
        e = np.exp((s[kept] - s[kept].max()).astype(F32)).astype(np.float64)
        cum = np.cumsum(e)
        u = float(philox_uniform(seed, step, row0 + b, channel))
        j = int(np.searchsorted(cum, u * cum[-1], side="right"))
        out[b] = kept[min(j, len(kept) - 1)]
    return out


# --------------------------------------------------------------------------
# The model
# --------------------------------------------------------------------------
class AsteroidOracle:
    def __init__(self, cfg, weights, dtype="bf16"):
        self.cfg = cfg
        self.w = weights
        self.dtype = dtype
        # "fp16" (`inference.py --dtype fp16`, reference inference.py:27-40): the same rounding points with an fp16 cast;
        # the weights arrive as fp32 and are cast like `model.to(torch.float16)` casts them
        self.r = {"bf16": round_bf16, "fp16": round_fp16}.get(dtype, _ident)
        if dtype == "fp16":
            self.w = {k: round_fp16(v) for k, v in weights.items()}
        self.neg = {"bf16": F32(-3.3895313892515355e38), "fp16": F32(-65504.0)}.get(dtype, np.finfo(F32).min)
        D = cfg["head_dim"]
        # RoPE frequencies: modeling_qwen3.py compute_default_rope_parameters
        self.inv_freq = (1.0 / (F32(cfg["rope_theta"]) ** (np.arange(0, D, 2, dtype=F32) / F32(D)))).astype(F32)
        # `attn_weights * scaling` on a bf16 CPU tensor multiplies in fp32 by the
        # python scalar and rounds once (verified against the reference fixtures).
        self.scale = F32(D ** -0.5)
        self.reset()

    # ---- pieces ----------------------------------------------------------
    def reset(self):
        self.K = [None] * self.cfg["num_hidden_layers"]
        self.V = [None] * self.cfg["num_hidden_layers"]

    def embed_sum(self, ids):
        """reference modeling_asteroid.py:244-248 (sequential adds in weight dtype)."""
        acc = np.zeros(ids.shape[:2] + (self.cfg["hidden_size"],), dtype=F32)
        for c in range(self.cfg["channels"]):
            acc = self.r(acc + self.w[f"model.embedding_list.{c}.weight"][ids[..., c]])
        return acc

    def rmsnorm(self, x, w):
        x = x.astype(F32)
        var = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
        y = x * (F32(1.0) / np.sqrt(var + F32(self.cfg["rms_norm_eps"]), dtype=F32))
        return self.r(w * self.r(y))

    def linear(self, x, w):
        return self.r(np.matmul(x.astype(F32), w.T.astype(F32)))

    def rope(self, pos):
        f = pos[..., None].astype(F32) * self.inv_freq           # [B,S,D/2]
        emb = np.concatenate([f, f], axis=-1)
        return self.r(np.cos(emb, dtype=F32)), self.r(np.sin(emb, dtype=F32))

    def apply_rope(self, x, cos, sin):
        """x [B,h,S,D]; bf16 elementwise with three roundings."""
        h = x.shape[-1] // 2
        rot = np.concatenate([-x[..., h:], x[..., :h]], axis=-1)
        return self.r(self.r(x * cos[:, None]) + self.r(rot * sin[:, None]))

    # ---- forward ---------------------------------------------------------
    def forward(self, ids, positions, key_mask, all_positions=False):
        """ids [B,S,8]; positions [B,S] int; key_mask [B,L_total] (1 = real token)
        over every cache slot including the S new ones.  Returns logits list
        (8 x [B,V_c] fp32 for the last position, or [B,S,V_c])."""
        x = self.forward_hidden(ids, positions, key_mask)
        if not all_positions:
            x = x[:, -1]
        return self.heads(x)

    def heads(self, x):
        """reference modeling_asteroid.py:412 (8 tied heads), logits .float() (:123)."""
        return [self.linear(x, self.w[f"model.embedding_list.{c}.weight"]).astype(F32)
                for c in range(self.cfg["channels"])]

    def forward_hidden(self, ids, positions, key_mask):
        """Embedding sum + decoder stack + final norm -> [B,S,H]."""
        cfg = self.cfg
        B, S, _ = ids.shape
        nq, nkv, D = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
        g = nq // nkv
        x = self.embed_sum(ids)
        cos, sin = self.rope(positions)
        past = 0 if self.K[0] is None else self.K[0].shape[2]
        L = past + S
        # additive mask [B,1,S,L]: causal over absolute slots + key padding
        qslot = past + np.arange(S)
        allow = (np.arange(L)[None, :] <= qslot[:, None])[None] & (key_mask[:, None, :L] > 0)
        amask = np.where(allow, F32(0), self.neg)[:, None]
        for n in range(cfg["num_hidden_layers"]):
            p = f"model.language_model.layers.{n}."
            hn = self.rmsnorm(x, self.w[p + "input_layernorm.weight"])
            q = self.linear(hn, self.w[p + "self_attn.q_proj.weight"]).reshape(B, S, nq, D)
            k = self.linear(hn, self.w[p + "self_attn.k_proj.weight"]).reshape(B, S, nkv, D)
            v = self.linear(hn, self.w[p + "self_attn.v_proj.weight"]).reshape(B, S, nkv, D)
            q = self.rmsnorm(q, self.w[p + "self_attn.q_norm.weight"]).transpose(0, 2, 1, 3)
            k = self.rmsnorm(k, self.w[p + "self_attn.k_norm.weight"]).transpose(0, 2, 1, 3)
            v = v.transpose(0, 2, 1, 3)
            q = self.apply_rope(q, cos, sin)
            k = self.apply_rope(k, cos, sin)
            if self.K[n] is None:
                self.K[n], self.V[n] = k, v
            else:
                self.K[n] = np.concatenate([self.K[n], k], axis=2)
                self.V[n] = np.concatenate([self.V[n], v], axis=2)
            Kr = np.repeat(self.K[n], g, axis=1)
            Vr = np.repeat(self.V[n], g, axis=1)
            s = self.r(np.matmul(q, Kr.transpose(0, 1, 3, 2)))
            s = self.r(s * self.scale)
            s = self.r(s + amask)
            pr = self.r(softmax_f32(s))
            o = self.r(np.matmul(pr, Vr)).transpose(0, 2, 1, 3).reshape(B, S, nq * D)
            x = self.r(x + self.linear(o, self.w[p + "self_attn.o_proj.weight"]))
            hn = self.rmsnorm(x, self.w[p + "post_attention_layernorm.weight"])
            gt = self.linear(hn, self.w[p + "mlp.gate_proj.weight"])
            up = self.linear(hn, self.w[p + "mlp.up_proj.weight"])
            act = self.r(gt / (F32(1) + np.exp(-gt, dtype=F32)))
            x = self.r(x + self.linear(self.r(act * up), self.w[p + "mlp.down_proj.weight"]))
        return self.rmsnorm(x, self.w["model.language_model.norm.weight"])

    # ---- the decode loop (reference modeling_asteroid.py:83-169) ----------
    def generate(self, input_ids, attention_mask, max_length, layers=None,
                 do_samples=None, seed=0, return_logits=False, max_steps=None, forced=None, forced_as_draw=False):
        """input_ids int64 [B,T,8], attention_mask [B,T].  `max_length` is HF's
        generation_config.max_length (counts the T-7 kept prompt slots + new
        tokens).  layers/do_samples mirror generation_config.layers/do_samples.
        forced (test hook): int64 [B,G,8] continuation; each step's own decision is
        recorded, then the forced row is appended instead (teacher-forced replay);
        returns (ids, decisions[steps,B,8]).  forced_as_draw (replay of a SAMPLED reference run): the forced row
        replaces the step's raw draw BEFORE the state machine (which then follows the reference's history, not this
        run's own draws); the recorded decisions are the raw draws.  forced_as_draw="all": the forced row is also the
        raw pick of rows that are already finished (the reference keeps evaluating every row, :140-141) -- lets a test
        script chained finished-row resurrections."""
        cfg = self.cfg
        C = cfg["channels"]
        eos, pad = cfg["eos_token_id"], cfg["speech_pad_token"]
        lo, hi = cfg["speech_token_range"]
        self.reset()
        B, T, _ = input_ids.shape
        tf_inputs = input_ids
        ids = input_ids[:, :-(C - 1)].copy()
        mask = (np.asarray(attention_mask)[:, :-(C - 1)] > 0).astype(np.int64)
        base_length = ids.shape[1]
        unfinished = np.ones(B, dtype=np.int64)
        nas = -np.ones(B, dtype=np.int64)
        layers = layers or [{} for _ in range(C)]
        do_samples = do_samples or [False] * C
        logits_log, decisions = [], []
        self.last_margins = []          # relative top-2 gap of every decision, [steps][B,C]
        self.last_scores = []           # processed scores [steps][C] of [B,V_c] when self.keep_scores is set
        step = 0
        first = True
        while True:
            # HF 4.53.2 prepare_inputs_for_generation: positions = cumsum(mask)-1, pads -> 1
            pos_all = np.cumsum(mask, axis=1) - 1
            pos_all[mask == 0] = 1
            if first:
                # long prompts: the prompt is fed in chunks of query rows (each row's arithmetic is its own, so the
                # chunking only bounds the [S,L] score temporaries)
                ch = int(getattr(self, "prefill_chunk", 0) or 0)
                if ch and ids.shape[1] > ch:
                    for a in range(0, ids.shape[1], ch):
                        e = min(a + ch, ids.shape[1])
                        logits = self.forward(ids[:, a:e], pos_all[:, a:e], mask[:, :e])
                else:
                    logits = self.forward(ids, pos_all, mask)
                first = False
            else:
                logits = self.forward(ids[:, -1:], pos_all[:, -1:], mask)
            cur = ids.shape[1]
            for c in range(C):
                if c != 0 and cur + 1 > tf_inputs.shape[1] - 7 + c:
                    logits[c][:, 1024] = -np.inf
                if c == 0 and cur + 1 <= tf_inputs.shape[1]:
                    logits[c][:, 152694] = -np.inf
            if return_logits:
                logits_log.append([l.copy() for l in logits])
            nxt = np.zeros((B, C), dtype=np.int64)
            mg = np.zeros((B, C), dtype=F32)
            for c in range(C):
                sc = apply_processors(ids[..., c], logits[c], layers[c])
                if getattr(self, "keep_scores", False):
                    if c == 0:
                        self.last_scores.append([])
                    self.last_scores[-1].append(sc.copy())
                if do_samples[c]:
                    nxt[:, c] = sample_from_scores(sc, seed, step, c)
                else:
                    nxt[:, c] = np.argmax(sc, axis=-1)
                top2 = np.partition(sc, -2, axis=-1)[:, -2:]
                mg[:, c] = (top2[:, 1] - top2[:, 0]) / np.maximum(np.abs(top2[:, 1]), F32(1e-6))
            self.last_margins.append(mg)
            if forced is not None and forced_as_draw:
                decisions.append(nxt.copy())
                # the draws of rows the reference has already finished are not in its output (padding is): those rows
                # keep their own picks, unless the test scripts them too (forced_as_draw == "all")
                fr = forced[:, ids.shape[1]]
                take = np.ones(B, dtype=bool) if forced_as_draw == "all" else (unfinished == 1)
                nxt = np.where(take[:, None], fr, nxt)
            is_speech = (nxt[:, 0] >= lo) & (nxt[:, 0] < hi)
            nas[(~is_speech) & (nas < 0)] = C - 1
            if cur + 1 <= tf_inputs.shape[1]:
                i = cur + 1 - base_length
                nxt[:, i:] = tf_inputs[:, cur, i:]
            m = (nas > 0) & (nas < 7)
            if m.any():
                nxt[m, 0] = eos
                for i in range(1, C):
                    nxt[m & (nas < C - i), i] = pad
            for i in range(C):
                pd = eos if i == 0 else pad
                nxt[:, i] = nxt[:, i] * unfinished + pd * (1 - unfinished)
            if forced is not None and not forced_as_draw:
                decisions.append(nxt.copy())
                nxt = forced[:, ids.shape[1]].copy()
            ids = np.concatenate([ids, nxt[:, None, :]], axis=1)
            mask = np.concatenate([mask, np.ones((B, 1), dtype=np.int64)], axis=1)
            nas = np.where(nas > 0, nas - 1, nas)
            stopping = (ids.shape[1] >= max_length) | (ids[:, -1, 0] == eos) | (nas == 0)
            unfinished = unfinished & (~stopping).astype(np.int64)
            unfinished = unfinished | (nas > 0).astype(np.int64)
            step += 1
            if unfinished.max() == 0:
                break
            if max_steps is not None and step >= max_steps:
                break
            if forced is not None and ids.shape[1] >= forced.shape[1]:
                break
        if forced is not None:
            return ids, np.stack(decisions), logits_log
        if return_logits:
            return ids, logits_log
        return ids


# --------------------------------------------------------------------------
# Glue restated for the tests (reference generation_utils.py:416-428, :240-249)
# --------------------------------------------------------------------------
def unshift_outputs(outputs, start, channels=8, offset=151665):
    out = outputs[:, start:]
    seq_len = out.shape[1] - channels + 1
    speech = np.zeros((out.shape[0], seq_len, channels), dtype=np.int64)
    for j in range(channels):
        speech[..., j] = out[:, j:seq_len + j, j]
    speech[..., 0] -= offset
    return speech


def find_max_valid_positions(C, invalid=1024):
    vals = C[:, :, 1]
    m = vals != invalid
    has = m.any(axis=1)
    last = C.shape[1] - 1 - np.argmax(m[:, ::-1], axis=1)
    return np.where(has, last, -1)
